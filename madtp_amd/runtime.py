"""Host-side runtime shared by the module mirrors: precision mode, prepared-weight cache, small helpers.

Precision modes (SURVEY.md section 7 "hard parts"):
  "fp32"  - parity mode: every GEMM runs on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32), activations f32.
            Kept-token index sets match the fp32 reference eager path.
  "f16x3" - fp32-ACCURATE mode on the f16 MFMA: every Linear runs as three f16 MFMA products of f16-split operands
            (csrc/common.h: x = P0 + 2^-11 P1, w 2^s = Q0 + Q1; exact partial products, f32 accumulation, ~2^-23
            relative error per product - the rounding class of an f32 dot product) at 3/16 of the exact-f32 MFMA cost;
            attention, LayerNorm, the alignment logits and all pruning scores are the fp32 mode's kernels.  Kept-token
            sets match the reference like the fp32 mode (tests/test_model_parity_gpu.py); GEMM operands are torch.float16
            tensors holding the planes side by side ([M, 2K] activations, [N, 3K] weights).
            RANGE: the activation planes carry no prescale (weights get a power of two), so a GEMM input must stay inside the
            f16 range: |x| >= 65504 becomes inf (then NaN) and |x| below ~6e-5 loses relative accuracy.  The BLIP / CLIP
            activations on this path (LayerNorm outputs, GELU(fc1), attention contexts, CLIP's q_map inputs) are O(1e-3 .. 1e2);
            a model whose residual stream leaves that range has to run in the "fp32" mode, which has no such limit.
  "bf16"  - fast mode: GEMM operands are bf16 (f32 accumulate, MFMA 16x16x32), the residual stream, LayerNorm
            statistics, softmax, the alignment logits x.sd^T and every pruning score stay f32.
  "f16"   - the same fast mode with IEEE f16 operands (round 4): identical kernels, layouts and MFMA rate
            (v_mfma_f32_16x16x32_f16), 11 significand bits instead of 8.  Weights are stored as w * 2^s (accumulator scale 2^-s)
            like the split planes; activations carry no prescale, so the RANGE note of "f16x3" applies - but a value that leaves
            the f16 range raises the library's range flag (MADTP_E_RANGE at the layer's host read of k, hip.range_status())
            instead of turning into NaNs.  Operands live in torch.bfloat16-typed containers (hip.set_lp_format).
"""
import os
import threading
import weakref

import torch

from . import hip

_state = threading.local()
_DEFAULT = "bf16"
MODES = ("fp32", "f16x3", "bf16", "f16")
_CDT = {"fp32": torch.float32, "f16x3": torch.float16, "bf16": torch.bfloat16, "f16": torch.bfloat16}  # "f16": a 2-byte container


def set_precision(mode: str):
    if mode not in MODES:
        raise ValueError(f"precision must be one of {MODES}")
    _state.mode = mode
    hip.set_lp_format(hip.F16 if mode == "f16" else hip.BF16)
    hip.set_score_fast(mode in ("bf16", "f16") and _SCORE_FAST)  # process-wide, like the element format: the modes of concurrent threads agree


_SCORE_FAST = os.environ.get("MADTP_SCORE_FAST", "1") != "0"  # A/B switch: the fast modes on the reference score arithmetic
hip.set_score_fast(_DEFAULT in ("bf16", "f16") and _SCORE_FAST)  # (remembered by the binding until the library is loaded)


def get_precision() -> str:
    return getattr(_state, "mode", _DEFAULT)


# The autograd route (madtp_amd/backward.py) of the mirrors: taken in the "fp32" mode, and - opt-in, round 5 - in the "f16x3" mode,
# whose backward runs every GEMM as three f16 MFMA products as well (MADTP_TRAIN_F16X3=1, or `with runtime.training_f16x3():`).
# Opt-in because an f16x3 forward without torch.no_grad() is otherwise an inference call.
_TRAIN_X3 = os.environ.get("MADTP_TRAIN_F16X3", "0") == "1"


# Round 6: `--amp` (compress_nlvr_dtp.py:46-53: torch.cuda.amp.autocast around the forward, GradScaler around the backward).  Opt-in,
# MADTP_TRAIN_AMP=1 or `with runtime.training_amp():` - the autograd route in the FAST modes: every GEMM of the training forward, of dgrad
# and of wgrad takes its operands rounded to the mode's 2-byte format (bf16, or IEEE f16 - the format autocast uses on GPUs) with f32
# accumulation, parameters / gradients / LayerNorm / softmax / pruning scores stay f32, as under autocast.  Gradients are approximate by
# construction (operand rounding 2^-9 / 2^-12); with f16 operands tiny gradients underflow, which is what the reference's GradScaler is
# for - a caller's scaled loss simply rides through the backward.
_TRAIN_AMP = os.environ.get("MADTP_TRAIN_AMP", "0") == "1"


def autograd_precision():
    """True when a forward in the current precision mode builds an autograd graph (given grad mode and inputs that need one)."""
    m = get_precision()
    return (m == "fp32" or (m == "f16x3" and getattr(_state, "train_x3", _TRAIN_X3))
            or (m in ("bf16", "f16") and getattr(_state, "train_amp", _TRAIN_AMP)))


class training_amp:
    """context manager: forwards of the fast modes (bf16 / f16) inside build autograd graphs whose GEMMs run on 2-byte operands - the
    counterpart of the reference's `torch.cuda.amp.autocast()` (compress_nlvr_dtp.py:46-53)"""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = getattr(_state, "train_amp", _TRAIN_AMP)
        _state.train_amp = bool(self.on)

    def __exit__(self, *a):
        _state.train_amp = self.prev


class training_f16x3:
    """context manager: forwards of the f16x3 mode inside build autograd graphs (the backward's GEMMs run as f16x3 too)"""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = getattr(_state, "train_x3", _TRAIN_X3)
        _state.train_x3 = bool(self.on)

    def __exit__(self, *a):
        _state.train_x3 = self.prev


def set_encoder_call_preference(v):
    """Thread-local override of MADTP_ENCODER_CALL's "auto" rule (None = no override).  The in-flight workers of
    madtp_amd.pipeline set True: with several host threads the per-layer Python path makes the forwards' progress depend on GIL
    hand-overs (measured: 17-22 k images/s from run to run), the encoder-level C loops hold no GIL (21-23 k)."""
    _state.encoder_call = v


def encoder_call_preference():
    return getattr(_state, "encoder_call", None)


def compute_dtype():
    """torch dtype of the GEMM operands: float32, bfloat16, or float16 = f16-split planes (mode "f16x3")."""
    return _CDT[get_precision()]


def attn_dtype():
    """dtype of q/k/v and the context of the attention kernels (the f16x3 mode keeps attention on the exact-f32 kernels)."""
    return torch.bfloat16 if get_precision() in ("bf16", "f16") else torch.float32


def dtype_code():
    """MADTP_F32 / MADTP_BF16 / MADTP_F16S of the current mode (the `dtype` field of the layer weight structs)."""
    return hip.dt_code(compute_dtype())


def to_compute(x):
    """f32 [..., K] with contiguous rows -> the current mode's GEMM operand (itself, a bf16 copy, or f16-split planes)."""
    cdt = compute_dtype()
    if cdt == torch.float32:
        return x
    x = x if x.is_contiguous() else x.contiguous()
    return hip.split_f16(x) if cdt == torch.float16 else hip.cast_bf16(x)


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
    """A Linear prepared for madtp_gemm: weight padded to a multiple of 128 rows in the compute dtype (float16: the three
    f16 planes of w * 2^s side by side, the tensor tagged with its accumulator scale 2^-s), f32 bias.
    log2_scale / age: the s of an f16-split weight and how many re-preparations it has served (training: see prepare_linear)."""
    __slots__ = ("w", "b", "n", "log2_scale", "age")

    def __init__(self, w, b, n, log2_scale=None, age=0):
        self.w, self.b, self.n, self.log2_scale, self.age = w, b, n, log2_scale, age


def param_step_count(p):
    """how many optimizer steps have updated parameter p in this process (the per-parameter update epoch, see _on_optimizer_step)"""
    return getattr(p, "_madtp_steps", 0)


def _pad_rows(w):
    n = w.shape[0]
    npad = (n + 127) // 128 * 128
    if npad == n:
        return w.contiguous()
    out = torch.zeros((npad, w.shape[1]), device=w.device, dtype=w.dtype)
    out[:n] = w
    return out


# An f16-split weight is stored as w * 2^s with max|w| 2^s in (2^13, 2^14]: s comes from a host read of max|w|.  In TRAINING every
# optimizer step re-prepares every weight (their versions change) - ~400 host reads per step.  The scale only has to keep
# max|w| 2^s inside the f16 range with the low plane clear of the subnormals, which a slowly moving weight does for many steps: a
# re-preparation of the SAME cache entry reuses the previous s up to SCALE_REUSE times (two binades of headroom above 2^14; a
# weight that outgrows them raises the library's range flag - split_f16_weight_kernel / weight_planes_kernel - instead of going
# wrong silently).  Reuse is for OPTIMIZER STEPS only (small moves): the cache hands the previous value to the builder only when
# every changed parameter was changed by an optimizer step since (param_step_count); a load_state_dict / copy_ over the same storage
# (a checkpoint over random-init weights can be orders of magnitude away) takes a fresh scale from max|w|.
SCALE_REUSE = 64


def prepare_linear(weights, biases, dtype, prev=None):
    """weights: list of [out_i, in] tensors concatenated along out (fused projections); biases likewise (or None).
    prev: the Lin this one replaces in its cache entry (same parameters, an older version) or None."""
    w = torch.cat([t.detach().reshape(t.shape[0], -1).float() for t in weights], 0) if len(weights) > 1 \
        else weights[0].detach().reshape(weights[0].shape[0], -1).float()
    n = w.shape[0]
    w = _pad_rows(w)
    s, age = None, 0
    if dtype == torch.bfloat16:
        w = hip.cast_lp_weight(w.contiguous())
    elif dtype == torch.float16:
        if prev is not None and prev.log2_scale is not None and prev.age < SCALE_REUSE and prev.w.shape[0] == w.shape[0]:
            s, age = prev.log2_scale, prev.age + 1
        w = hip.split_f16_weight(w.contiguous(), log2_scale=s)
        s = w._madtp_log2_scale
    b = None
    if biases is not None and all(bb is not None for bb in biases):
        b = torch.cat([bb.detach().float() for bb in biases], 0).contiguous() if len(biases) > 1 \
            else biases[0].detach().float().contiguous()
    return Lin(w, b, n, s, age)


class PreparedCache:
    """Per-module cache of prepared weights, invalidated when a parameter is modified in place, re-assigned,
    moved, or the precision mode changes (load_state_dict bumps Parameter._version)."""

    MAX_ENTRIES = 32

    def __init__(self):
        self._store = {}
        self._lock = threading.RLock()
        self.shared = False  # set by pipeline.shared_replica: several module replicas (host threads, streams) read this cache

    # copy.deepcopy(model) / pickling a module: the copy starts with an EMPTY cache (prepared tensors are derived data keyed by the
    # original's parameter storage; the lock is not copyable)
    def __deepcopy__(self, memo):
        return PreparedCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def get(self, key, params, builder):
        sig = ((get_precision(), _UPDATE_EPOCH[0]),) + tuple((p.data_ptr(), p._version, p.device, getattr(p, "_madtp_steps", 0)) if p is not None else None
                                                             for p in params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with self._lock:  # replicas that share this cache (pipeline.shared_replica) may miss at the same time: one of them builds
            return self._build(key, sig, builder)

    def _build(self, key, sig, builder):
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        # a builder that takes the entry's previous value (same key and storage, older parameter versions) may reuse parts of it (the
        # f16-split scale) - only across optimizer steps: every parameter whose version moved has a step count that moved too, and no
        # hand edit (parameters_updated) happened in between
        prev = hit[1] if (hit is not None and hit[0][0] == sig[0] and len(hit[0]) == len(sig)
                          and all((a is None) == (b is None) and (a is None or (a[0] == b[0] and a[2] == b[2] and (a[1] == b[1] or a[3] != b[3])))
                                  for a, b in zip(hit[0][1:], sig[1:]))
                          and any(a is not None and a[3] != b[3] for a, b in zip(hit[0][1:], sig[1:]))) else None
        val = builder(prev) if getattr(builder, "_takes_prev", False) else builder()
        if self.shared and torch.cuda.is_available():
            # other host threads will use `val` on THEIR streams: its preparation kernels (casts, splits) must have completed,
            # not merely been enqueued on this thread's stream (rare: once per weight and precision mode)
            torch.cuda.current_stream().synchronize()
        self._store.pop(key, None)
        self._store[key] = (sig, val)
        while len(self._store) > self.MAX_ENTRIES:  # bounded: keys may embed id() of caller tensors (a fresh space_dict per call)
            self._store.pop(next(iter(self._store)))
        return val


def lin_of(cache, key, linears, dtype=None):
    """Prepared (optionally fused) projection for a list of nn.Linear-like modules (weight/bias attributes)."""
    dtype = dtype or compute_dtype()
    params = []
    for l in linears:
        params += [l.weight, l.bias]
    def build(prev=None):
        return prepare_linear([l.weight for l in linears], [l.bias for l in linears], dtype, prev if isinstance(prev, Lin) else None)
    build._takes_prev = True
    return cache.get((key, dtype), params, build)


# Parameter RE-ASSIGNMENT (`blk.mlp.fc1.weight = nn.Parameter(...)`, `load_state_dict(assign=True)`) or a replaced sub-module
# (`blk.mlp.fc1 = nn.Linear(...)`) swaps the Parameter objects, which the per-module lists of collected parameters below would
# keep missing (their signatures only see in-place changes of the objects they hold).  torch calls these hooks for EVERY
# registration in the process, so they only count registrations ON modules of a mirror model that has collected its parameters
# (own_modules(), called where a layer collects its list): a host application that builds other modules per request, or on
# another thread, does not make the layers re-walk their parameters.
_PARAM_EPOCH = [0]
_OWNED = weakref.WeakSet()


def own_modules(root):
    """Marks root and its sub-modules as holders of collected parameters (see above)."""
    for m in root.modules():
        _OWNED.add(m)


def _on_parameter_registration(module, name, param):
    if module in _OWNED:
        _PARAM_EPOCH[0] += 1


def _on_module_registration(module, name, submodule):
    if module in _OWNED:
        _PARAM_EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_on_parameter_registration)
torch.nn.modules.module.register_module_module_registration_hook(_on_module_registration)


def param_epoch():
    return _PARAM_EPOCH[0]


# In-place parameter UPDATES are normally seen through Parameter._version (every cache signature below carries it) - but not all of
# them bump it: torch.optim's fused=True implementations (torch._fused_adamw_ and friends) leave _version untouched (checked on this
# build: a foreach step takes it 0 -> 2, a fused step 0 -> 0), and neither do edits through `.data`.  A stale prepared weight would
# silently keep multiplying with the old values while LayerNorm parameters (read in place) moved on.  So every optimizer step bumps
# a step count ON THE PARAMETERS OF THAT OPTIMIZER (a global torch.optim step post-hook walks its param_groups; `_madtp_steps`), which
# all signatures carry next to the version - scoped per parameter (round 6): a second optimizer, a second model or a frozen inference
# model in the same process keeps its prepared weights and its forward/backward consistency check.  Code that edits `.data` by hand
# calls parameters_updated(), which bumps the process-wide epoch (nothing says WHICH parameters moved).
_UPDATE_EPOCH = [0]


def _on_optimizer_step(optimizer, args, kwargs):
    for group in optimizer.param_groups:
        for p in group["params"]:
            p._madtp_steps = getattr(p, "_madtp_steps", 0) + 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook
    _register_step_hook(_on_optimizer_step)
except ImportError:  # (a torch without global optimizer hooks: version counters only)
    pass


def update_epoch():
    return _UPDATE_EPOCH[0]


def parameters_updated():
    """Tell the prepared-weight caches that parameters changed through a path that bumps no version counter (`.data` edits)."""
    _UPDATE_EPOCH[0] += 1


class EncoderWeights:
    """The weight structs of all layers of an encoder for the encoder-level calls (hip.vit_encoder / hip.bert_encoder),
    revalidated per forward with ONE pass over the flattened parameter list (version counters and data pointers) instead of
    a PreparedCache lookup per layer - that was ~0.1 ms of host time in front of the first launch of an encoder."""

    def __init__(self):
        self.flat, self.sig, self.val, self.epoch = None, None, None, -1

    def invalidate(self):
        self.flat = self.sig = self.val = None

    def get(self, layers):
        import ctypes
        if self.flat is None or self.epoch != _PARAM_EPOCH[0]:
            self.epoch = _PARAM_EPOCH[0]
            for l in layers:
                l._weights()  # (collects l._madtp_params)
            self.flat = [p for l in layers for p in l.__dict__["_madtp_params"]]
            self.sig = None
        sig = (get_precision(), _UPDATE_EPOCH[0], tuple([p._version for p in self.flat]), tuple([p.data_ptr() for p in self.flat]),
               tuple([getattr(p, "_madtp_steps", 0) for p in self.flat]))
        if sig != self.sig:
            ws = [l._weights() for l in layers]
            arr = (ctypes.c_void_p * len(ws))(*[ctypes.addressof(w) for w in ws])
            self.sig, self.val = sig, (ws, arr)
        return self.val


_SIDE = {}
_SIDE_HANDLES = {}  # MaskedStream handles of side streams that follow a CU-masked main stream


def side_stream(device=None):
    """One auxiliary HIP stream per (device, current main stream) for work the main stream does not depend on (the deferred
    att_ft sum of the vision encoder runs there, under the text encoder's latency-bound kernels).  Keyed by the main stream so
    that forwards in flight on different streams (madtp_amd.pipeline) do not serialise on one side stream."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _SIDE:
        info = hip.masked_stream_info(key[1])
        if info is not None:  # the main stream owns a slice of the chip (hip.MaskedStream): its side work stays on the same CUs
            h = hip.MaskedStream(info[0], info[1], device=torch.device("cuda", dev), sq_cost=info[2], small_tile=info[3])
            _SIDE_HANDLES[key] = h
            _SIDE[key] = h.stream
        else:
            _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def require_gpu(t, name="input"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: madtp_amd modules run only on an MI355X through the HIP "
                           "kernels (no CPU / eager fallback).  Use oracle/ for CPU checking.")
    return t


def f32_ptr(p, name="parameter"):
    """data_ptr() of a parameter the kernels read as contiguous f32 (LayerNorm gamma / beta): model.half() / .bfloat16() or a
    non-contiguous parameter must fail loudly instead of being misread."""
    if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
        raise TypeError(f"{name}: the HIP kernels read it as a contiguous float32 GPU tensor, got {p.dtype} "
                        f"{'contiguous' if p.is_contiguous() else 'strided'} on {p.device} - keep the module in float32")
    return p.data_ptr()


def as_f32_contig(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ---- dropout / DropPath of the training forward (round 5) ---------------------------------------------------------------------------
# Active when a module is in .train() mode, its rate is > 0 and the forward takes the autograd route (madtp_amd/backward.py).  Masks are
# counter-based (csrc/backward.hip: Philox4x32-10 over (seed, site, element)): every layer call draws a fresh `base` from the
# process-wide counter, its sites are base * 32 + a per-site code; the backward regenerates them.  set_dropout_seed(seed) also resets
# the counter, so two runs with the same seed and the same sequence of calls drop the same elements.
_DROP_STATE = {"seed": None, "counter": 0}
_DROP_LOCK = __import__("threading").Lock()


def set_dropout_seed(seed):
    with _DROP_LOCK:
        _DROP_STATE["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
        _DROP_STATE["counter"] = 0


def next_dropout_base():
    """-> (seed, base): a fresh site base for one layer call of the training forward"""
    with _DROP_LOCK:
        if _DROP_STATE["seed"] is None:
            _DROP_STATE["seed"] = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        b = _DROP_STATE["counter"]
        _DROP_STATE["counter"] = b + 1
        return _DROP_STATE["seed"], b


_WARNED_DROPOUT = [False]


def warn_no_dropout(module):
    """kept for callers of earlier rounds: the training forwards now apply the reference's dropout / DropPath in module.train() mode
    (madtp_amd/backward.py); nothing to warn about."""
    return None
