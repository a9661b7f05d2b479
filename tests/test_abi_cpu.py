"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads without a GPU and exports every
symbol include/madtp_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from madtp_amd import build
    return build.build(verbose=False)


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "madtp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(madtp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/madtp_hip.h but not exported"


def test_binding_matches_header(lib_path):
    from madtp_amd import hip
    assert hip.exported_symbols() == _header_symbols()
    lib = hip.load()
    assert lib.madtp_abi_version() == hip.ABI_VERSION
    assert lib.madtp_strerror(-2) == b"unsupported shape"


def test_argument_validation_without_gpu(lib_path):
    # argument checks run on the host before any launch: usable as a no-GPU smoke of the error contract
    from madtp_amd import hip
    lib = hip.load()
    assert lib.madtp_gemm(0, 0, 0, 0, 0, 1, 1, 64, 64, 64, 1, 0, 1, 1, 0, 1.0, 1.0, 0) == -1      # null pointers
    assert lib.madtp_gemm(16, 16, 0, 0, 16, 4, 4, 40, 40, 40, 4, 0, 1, 1, 0, 1.0, 1.0, 0) == -2   # K*2 % 128 != 0
    assert lib.madtp_gemm(16, 16, 0, 0, 16, 4, 4, 64, 64, 64, 4, 0, 7, 1, 0, 1.0, 1.0, 0) == -3   # dtype
    assert lib.madtp_gemm(8, 16, 0, 0, 16, 4, 4, 64, 64, 64, 4, 0, 1, 1, 0, 1.0, 1.0, 0) == -4    # alignment
    assert lib.madtp_token_select(16, 0, 16, 16, 16, 16, 2, 8, 0) == -2                       # k < 1


def test_product_path_refuses_cpu_tensors(lib_path):
    import torch
    from madtp_amd import hip
    with pytest.raises(RuntimeError):
        hip.gemm(torch.zeros(4, 64), torch.zeros(128, 64))
