import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keep_freed_memory_in_the_heap():
    """The CPU oracle allocates and frees hundreds of multi-MB activations per forward; glibc serves those with mmap / munmap,
    so every one of them page-faults afresh - in the build container first-touch memory runs at ~1 GB/s and the suite spent
    5-15 min of SYSTEM time there.  Serve them from the brk heap and never trim it (this process: mallopt; children: the
    MALLOC_* environment variables).  Test processes only."""
    os.environ.setdefault("MALLOC_MMAP_MAX_", "0")
    os.environ.setdefault("MALLOC_TRIM_THRESHOLD_", str(1 << 40))
    os.environ.setdefault("MALLOC_TOP_PAD_", str(256 << 20))
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-4, 0)                 # M_MMAP_MAX
        libc.mallopt(-1, (1 << 31) - 1)     # M_TRIM_THRESHOLD
        libc.mallopt(-2, 256 << 20)         # M_TOP_PAD
    except OSError:
        pass


_keep_freed_memory_in_the_heap()
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
