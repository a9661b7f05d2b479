"""Host-side runtime shared by the module mirrors: precision mode, prepared-weight cache, small helpers.

Precision modes (SURVEY.md section 7 "hard parts"):
  "fp32" - parity mode: every GEMM runs on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32), activations f32.
           Kept-token index sets match the fp32 reference eager path.
  "bf16" - fast mode: GEMM operands are bf16 (f32 accumulate, MFMA 16x16x32), the residual stream, LayerNorm
           statistics, softmax, the alignment logits x.sd^T and every pruning score stay f32.
"""
import threading

import torch

from . import hip

_state = threading.local()
_DEFAULT = "bf16"


def set_precision(mode: str):
    if mode not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    _state.mode = mode


def get_precision() -> str:
    return getattr(_state, "mode", _DEFAULT)


def compute_dtype():
    return torch.float32 if get_precision() == "fp32" else torch.bfloat16


class precision:
    """context manager: with precision('fp32'): ..."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_precision()
        set_precision(self.mode)

    def __exit__(self, *a):
        set_precision(self.prev)


class Lin:
    """A Linear prepared for madtp_gemm: weight padded to a multiple of 128 rows in the compute dtype, f32 bias."""
    __slots__ = ("w", "b", "n")

    def __init__(self, w, b, n):
        self.w, self.b, self.n = w, b, n


def _pad_rows(w):
    n = w.shape[0]
    npad = (n + 127) // 128 * 128
    if npad == n:
        return w.contiguous()
    out = torch.zeros((npad, w.shape[1]), device=w.device, dtype=w.dtype)
    out[:n] = w
    return out


def prepare_linear(weights, biases, dtype):
    """weights: list of [out_i, in] tensors concatenated along out (fused projections); biases likewise (or None)."""
    w = torch.cat([t.detach().reshape(t.shape[0], -1).float() for t in weights], 0) if len(weights) > 1 \
        else weights[0].detach().reshape(weights[0].shape[0], -1).float()
    n = w.shape[0]
    w = _pad_rows(w)
    if dtype == torch.bfloat16:
        w = hip.cast_bf16(w.contiguous())
    b = None
    if biases is not None and all(bb is not None for bb in biases):
        b = torch.cat([bb.detach().float() for bb in biases], 0).contiguous() if len(biases) > 1 \
            else biases[0].detach().float().contiguous()
    return Lin(w, b, n)


class PreparedCache:
    """Per-module cache of prepared weights, invalidated when a parameter is modified in place, re-assigned,
    moved, or the precision mode changes (load_state_dict bumps Parameter._version)."""

    def __init__(self):
        self._store = {}

    def get(self, key, params, builder):
        sig = (get_precision(),) + tuple((p.data_ptr(), p._version, p.device) if p is not None else None for p in params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        val = builder()
        self._store[key] = (sig, val)
        return val


def lin_of(cache, key, linears, dtype=None):
    """Prepared (optionally fused) projection for a list of nn.Linear-like modules (weight/bias attributes)."""
    dtype = dtype or compute_dtype()
    params = []
    for l in linears:
        params += [l.weight, l.bias]
    return cache.get((key, dtype), params,
                     lambda: prepare_linear([l.weight for l in linears], [l.bias for l in linears], dtype))


_SIDE = {}


def side_stream(device=None):
    """One auxiliary HIP stream per device for work the main stream does not depend on (the deferred att_ft sum of the
    vision encoder runs there, under the text encoder's latency-bound kernels)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev)
    return _SIDE[dev]


def require_gpu(t, name="input"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: madtp_amd modules run only on an MI355X through the HIP "
                           "kernels (no CPU / eager fallback).  Use oracle/ for CPU checking.")
    return t


def as_f32_contig(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
