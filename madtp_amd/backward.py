"""Backward of the pruned ViT block (first half): what `loss.backward()` does through
models/vit.py Block.forward (:183-207) in the reference's compression training (compress_nlvr_dtp.py:46-58, loss_fdt
blip_nlvr.py:84-98), as hand-written gfx950 kernels behind a torch.autograd.Function.

Scope: the block (VitBlockFunction) and, at the end of this file, the whole VisionTransformer.forward (both outputs), fp32
("parity") arithmetic, gradient-checked against the reference's own .grad (tests/golden/blockgrad_*.npz, encgrad_*.npz) and
autograd through the CPU oracle.  PyTorch is plumbing (buffers, the autograd graph
edge); every arithmetic op is a kernel of csrc/backward.hip or madtp_gemm:
  * dgrad  dX = dY W      -> madtp_gemm(dY, W^T)          (exact-f32 MFMA; W^T by madtp_transpose_pad)
  * wgrad  dW = dY^T X    -> madtp_gemm(dY^T, X^T)        (both operands transposed and zero-padded to 32 rows of M)
  * bias / LayerNorm parameter gradients: fixed-order column reductions
  * attention backward with recomputed probabilities, gather/merge backward, importance-score backward (the merge weights
    w = I / (sum_dropped I + 1e-8) carry gradient into the attention probabilities and the alignment logits; the top-k
    INDICES do not, exactly as torch.topk in vit.py:153-155).
The forward's intermediates are recomputed here from (x, token_attn) with the forward's own kernels and the forward's pruning
decision k (activation recomputation: nothing but the two inputs is kept alive between forward and backward).
"""
import math

import os

import torch

from . import hip
from .hip import _check, _p, _stream, load


def _pad(n, m):
    return (n + m - 1) // m * m


def transpose_pad(src, rows_pad, cols_pad):
    """src f32 [R, C] (row-major view, unit column stride) -> [cols_pad, rows_pad] with dst[c, r] = src[r, c], zero padding."""
    R, C = src.shape
    dst = torch.empty((cols_pad, rows_pad), device=src.device, dtype=torch.float32)
    _check(load().madtp_transpose_pad(_p(src), src.stride(0), R, C, _p(dst), rows_pad, rows_pad, cols_pad, _stream()),
           "madtp_transpose_pad")
    return dst


def transpose_split(src, rows_pad, cols_pad, weight, colsum=None):
    """src f32 [R, C] -> f16-split planes [cols_pad, 2 rows_pad] of src^T (madtp_transpose_split): activation format, or the weight
    format at scale 1 (tagged like hip.split_f16_weight(..., log2_scale=0)).  colsum (True / False): -> (planes, column sums of src
    [C] or None) from the same pass."""
    R, C = src.shape
    dst = torch.empty((cols_pad, 2 * rows_pad), device=src.device, dtype=torch.float16)
    cs = torch.empty((C,), device=src.device, dtype=torch.float32) if colsum else None
    ws = torch.empty(((rows_pad + 63) // 64 * C,), device=src.device, dtype=torch.float32) if colsum else None
    _check(load().madtp_transpose_split(_p(src), src.stride(0), R, C, _p(dst), rows_pad, cols_pad, 1 if weight else 0, _p(cs), _p(ws),
                                        _stream()), "madtp_transpose_split")
    if weight:
        dst._madtp_w_scale, dst._madtp_log2_scale = 1.0, 0
    return dst if colsum is None else (dst, cs)


# ---- precision of the backward's GEMMs --------------------------------------------------------------------------------------
# "fp32": exact-f32 MFMA (1/16 of the f16 rate).  "f16x3" (round 5): every product - the recomputed forward, dgrad, wgrad - as
# three f16 MFMA products of f16-split operands (common.h), the arithmetic the f16x3 FORWARD runs, at 3/16 of the f16 rate.
#   * the recomputed forward uses the forward's own operand formats and kernels (same planes, same dispatch: same bits, so the
#     recomputed pruning scores select the forward's kept set);
#   * a gradient operand dY travels in the ACTIVATION format (P0 + 2^-11 P1: absolute error ~2^-36, so gradients of 1e-6 keep five
#     digits without a loss scale; the reference trains under autocast + GradScaler, compress_nlvr_dtp.py:46-53, whose scale - if a
#     caller applies one - simply rides along);
#   * weights^T travel in the weight format with their per-tensor power-of-two scale (prepared once per parameter version);
#     wgrad's X^T in the weight format with scale 1 (O(1) activations: ~3e-8 absolute, no host read of max|X|).
_CACHE = {}


def _x3():
    from . import runtime
    return runtime.get_precision() == "f16x3"


def _lp():
    """the autocast-style route (runtime.training_amp): the fast modes' 2-byte GEMM operands in the training forward and backward"""
    from . import runtime
    return runtime.get_precision() in ("bf16", "f16") and runtime.autograd_precision()  # (a fast mode AND the amp opt-in)


def _lp_w(w32):
    """the 2-byte copy (bf16, or f16 of w 2^s tagged with its accumulator scale) of a cached f32 weight / transposed weight, attached
    to that tensor object (one cast per parameter version and element format)"""
    hit = getattr(w32, "_madtp_lp_w", None)
    if hit is None or hit[0] != hip.lp_format():
        hit = (hip.lp_format(), hip.cast_lp_weight(w32.contiguous()))
        w32._madtp_lp_w = hit
    return hit[1]


def _mode():
    from . import runtime
    return runtime.get_precision()


def _in_mode(mode):
    """context of a backward(): the autograd engine runs it on a worker thread whose thread-local precision mode is the default -
    the forward's mode is recorded on ctx and re-established here"""
    from . import runtime
    if mode in ("bf16", "f16"):  # a graph built in a fast mode was built under runtime.training_amp(): the backward runs under it too
        import contextlib
        st = contextlib.ExitStack()
        st.enter_context(runtime.precision(mode))
        st.enter_context(runtime.training_amp())
        return st
    return runtime.precision(mode)


def _check_mode(what):
    from . import runtime
    if runtime.get_precision() in ("bf16", "f16") and runtime.autograd_precision():
        return  # runtime.training_amp(): approximate gradients on 2-byte operands, asked for
    if runtime.get_precision() not in ("fp32", "f16x3"):
        raise NotImplementedError(f"{what} is built for the fp32-accurate precision modes (runtime.precision('fp32') or 'f16x3'); "
                                  f"current mode: {runtime.get_precision()}")


def _cached(key, tensors, build):
    """build() memoised per (key, identity and version of the source tensors): padded / concatenated / transposed / split copies of
    parameters are made once per parameter VERSION (an optimizer step bumps it) instead of in every backward."""
    from .runtime import update_epoch, param_step_count
    # (version, per-parameter optimizer step count, process-wide epoch of hand edits: fused optimizers bump no version, runtime.py)
    ver = tuple((t.data_ptr(), (t._version, param_step_count(t), update_epoch()), tuple(t.shape)) for t in tensors)
    hit = _CACHE.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    # (the entry holds its sources alive, so the same addresses and shapes ARE the same parameters at an older version: a builder
    #  that asks for it gets the previous value - the f16-split scale of a weight is reused across OPTIMIZER STEPS, runtime.SCALE_REUSE:
    #  every source that changed was changed by a step; a load_state_dict / copy_ / hand edit prepares from scratch)
    same = (hit is not None and len(hit[0]) == len(ver) and all(a[0] == b[0] and a[2] == b[2] for a, b in zip(hit[0], ver))
            and all(a[1] == b[1] or (a[1][1] != b[1][1] and a[1][2] == b[1][2]) for a, b in zip(hit[0], ver)))
    val = build(hit[1] if same else None) if getattr(build, "_takes_prev", False) else build()
    if len(_CACHE) > 4096:
        _CACHE.clear()
    # the entry holds the source tensors: while it lives their storage cannot be recycled for another tensor at the same address
    # (a key of addresses + versions alone would alias a freed model's weights with a new model's)
    _CACHE[key] = (ver, val, tuple(t.detach() for t in tensors))
    return val


def _planes(w32, prev=None):
    """f16-split weight planes of a (padded) f32 weight tensor, attached to it (the tensor itself is a cached object).  prev: the
    tensor this one replaces in its cache entry - its scale is reused (no host read of max|w|) up to runtime.SCALE_REUSE times."""
    if w32.dtype == torch.float16:  # already the planes (_x3_weights)
        return w32
    pl = getattr(w32, "_madtp_x3_planes", None)
    if pl is None:
        from .runtime import SCALE_REUSE
        s, age = None, 0
        pp = getattr(prev, "_madtp_x3_planes", None) if prev is not None else None
        if pp is not None and getattr(prev, "_madtp_x3_age", SCALE_REUSE) < SCALE_REUSE and pp.shape == (w32.shape[0], 2 * w32.shape[1]):
            s, age = pp._madtp_log2_scale, prev._madtp_x3_age + 1
        pl = hip.split_f16_weight(w32, log2_scale=s)
        w32._madtp_x3_planes, w32._madtp_x3_age = pl, age
    return pl


def _versioned(make):
    """builder for _cached: a (padded / concatenated / transposed) f32 weight copy that, in the f16x3 mode, carries its split planes
    from the start - prepared with the previous version's scale"""
    def build(prev=None):
        t = make()
        if _x3():
            _planes(t, prev)
        return t
    build._takes_prev = True
    return build


class _X3W:
    """both operand forms of one (possibly fused) weight in the f16x3 mode: planes [pad128(N), 2K] (forward) and planes_t
    [pad128(K), 2 pad64(N)] (dgrad), with their common power-of-two scale"""
    __slots__ = ("planes", "planes_t", "s", "age", "sig")


def _x3_weights(key, ws):
    """f16-split planes of the weight(s) ws (f32 [N_i, K], concatenated along N) and of the transpose, made by ONE kernel per part
    (madtp_weight_planes) into buffers that are allocated - zero padding included - once per cache entry and rewritten in place for
    every new parameter version; the scale is the previous version's up to runtime.SCALE_REUSE times (else one host read of max|w|)."""
    def build(prev=None):
        from .runtime import SCALE_REUSE
        Ns, K = tuple(int(w.shape[0]) for w in ws), int(ws[0].shape[1])
        Nt = sum(Ns)
        Ntp = _pad(Nt, 64)
        o = _X3W()
        o.sig = (Ns, K)
        if isinstance(prev, _X3W) and prev.sig == o.sig:
            o.planes, o.planes_t = prev.planes, prev.planes_t
            o.s, o.age = (prev.s, prev.age + 1) if prev.age < SCALE_REUSE else (None, 0)
        else:
            dev = ws[0].device
            o.planes = torch.zeros((_pad(Nt, 128), 2 * K), device=dev, dtype=torch.float16)
            o.planes_t = torch.zeros((_pad(K, 128), 2 * Ntp), device=dev, dtype=torch.float16)
            o.s, o.age = None, 0
        if o.s is None:
            amax = float(torch.stack([w.detach().abs().max() for w in ws]).max())
            o.s = 0
            if amax > 0 and amax == amax and amax != float("inf"):
                o.s = max(-100, min(100, 14 - math.ceil(math.log2(amax))))
        row = 0
        for w in ws:
            wd = w.detach()
            if wd.stride(1) != 1:
                wd = wd.contiguous()
            _check(load().madtp_weight_planes(_p(wd), wd.stride(0), int(wd.shape[0]), K, float(2.0 ** o.s), _p(o.planes), row, _p(o.planes_t),
                                              Ntp, row, _stream()), "madtp_weight_planes")
            row += int(wd.shape[0])
        for t in (o.planes, o.planes_t):
            t._madtp_w_scale, t._madtp_log2_scale = float(2.0 ** -o.s), o.s
        o.planes._madtp_t, o.planes._madtp_n = o.planes_t, Nt
        return o
    build._takes_prev = True
    return _cached(key, list(ws), build)


def _x3_ok(w):
    """the single-pass weight preparation takes 2-D f32 weights whose K feeds an f16x3 GEMM (K % 64 == 0) and whose N keeps the
    transposed planes' 4-element groups aligned"""
    return _x3() and w.dim() == 2 and w.dtype == torch.float32 and w.shape[1] % 64 == 0 and w.shape[0] % 4 == 0


def _gemm(a, w, bias=None, n=None, residual=None, out_dtype=torch.float32):
    """act-free Linear of the recomputed forward: a f32 [M, K] @ w f32 [Npad, K]^T (+ bias, + residual) -> f32 [M, n]."""
    if _x3() and a.shape[1] % 64 == 0:
        return hip.gemm(hip.split_f16(a.contiguous()), _planes(w), bias, residual, out_dtype=torch.float32, n=n)
    if _lp() and a.shape[1] % 64 == 0 and w.dtype == torch.float32:
        return hip.gemm(hip.cast_bf16(a.contiguous()), _lp_w(w), bias, residual, out_dtype=torch.float32, n=n)
    return hip.gemm(a, w, bias, residual, out_dtype=torch.float32, n=n)


def _attention(q, k, v, B, H, Nq, Nk, scale, **kw):
    """the recomputed forward's attention: the f16x3 mode's three-product kernels where the forward ran them (and on the amp route:
    q / k / v are f32 there, the three-product kernels are the fast ones for f32 storage)"""
    return hip.attention(q, k, v, B, H, Nq, Nk, scale, split=_x3() or _lp(), **kw)


def dgrad(dy, weight, residual=None):
    """dX[M, K] = dY[M, N] @ W[N, K] (nn.Linear weight layout) [+ residual[M, K]: the other branch of a residual connection]."""
    x3 = _x3()
    if x3 and weight.dtype == torch.float16:  # the planes of a fused projection (_cat_wb): its W^T planes ride along
        wtp, N, K = weight._madtp_t, weight._madtp_n, weight.shape[1] // 2
    elif x3 and _x3_ok(weight):
        wtp, (N, K) = _x3_weights(("x3w", weight.data_ptr(), tuple(weight.shape)), [weight]).planes_t, weight.shape
    else:
        wtp, (N, K) = None, weight.shape
    lp = _lp() and not x3
    Np = _pad(N, 64 if (x3 or lp) else 32)  # the GEMM's reduction length (slabs of 32 f32 / 64 f16): zero columns for e.g. the 100 dictionary columns
    if Np != N:
        dyp = torch.zeros((dy.shape[0], Np), device=dy.device, dtype=torch.float32)
        dyp[:, :N] = dy
        dy = dyp
    # [Kpad, Np]: the GEMM's "weight" with K' = N contiguous - once per parameter version (was: one transpose per backward call)
    if wtp is not None:
        return hip.gemm(hip.split_f16(dy.contiguous()), wtp, None, residual, out_dtype=torch.float32, n=K)
    wt = _cached(("wt", weight.data_ptr(), Np, x3), [weight], _versioned(lambda: transpose_pad(weight, Np, _pad(K, 128))))
    if x3:
        return hip.gemm(hip.split_f16(dy.contiguous()), _planes(wt), None, residual, out_dtype=torch.float32, n=K)
    if lp:
        return hip.gemm(hip.cast_bf16(dy.contiguous()), _lp_w(wt), None, residual, out_dtype=torch.float32, n=K)
    return hip.gemm(dy, wt, n=K, out_dtype=torch.float32, residual=residual)


def _wgrad_splits(M, N, K):
    """K ranges of the split-K weight-gradient product (0: the plain GEMM dispatch): the [N, K] output has (N/256)(K/256) tiles of
    the big kernel - enough ranges to give every CU one unit, each at least four 64-row slabs long."""
    if os.environ.get("MADTP_WGRAD_SPLITK", "1") == "0" or M < 2048 or N < 64 or K < 256 or K % 8:
        return 0
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    return max(1, min(32, 256 // tiles, (M // 64) // 4))


def wgrad(dy, x, bias=False):
    """dW[N, K] = dY[M, N]^T @ X[M, K]; bias: -> (dW, db) with db[N] = the column sums of dY (the f16x3 route takes them from the
    pass that transposes dY)."""
    M, N = dy.shape
    K = x.shape[1]
    x3 = _x3()
    Mp = _pad(M, 64 if x3 else 32)
    if x3:  # dY^T in the activation format (tiny values keep their digits), X^T in the weight format at scale 1
        S = _wgrad_splits(M, N, K)
        Mp = _pad(M, 128 * S) if S else Mp
        a, db = transpose_split(dy, Mp, N, False, colsum=bias)
        w = transpose_split(x, Mp, _pad(K, 128), True)
        if S:  # long K (every token row of the batch), few output tiles: split-K partials on the 256x256 ping-pong kernel
            part = torch.empty((S, N, K), device=dy.device, dtype=torch.float32)
            _check(load().madtp_gemm_splitk_pp(_p(a), _p(w), _p(part), N, K, Mp, 2 * Mp, 2 * Mp, S, 1.0, _stream()), "madtp_gemm_splitk_pp")
            if S == 1:
                dw = part[0]
            else:
                dw = torch.empty((N, K), device=dy.device, dtype=torch.float32)
                _check(load().madtp_splitk_sum(_p(part), S, N * K, _p(dw), _stream()), "madtp_splitk_sum")
        else:
            dw = hip.gemm(a, w, n=K, out_dtype=torch.float32)
        return (dw, db) if bias else dw
    if _lp():
        Mp = _pad(M, 64)
    dyt = transpose_pad(dy, Mp, N)             # [N, Mp]
    xt = transpose_pad(x, Mp, _pad(K, 128))    # [Kpad, Mp]
    if _lp():  # amp route: both operands rounded to the mode's 2-byte format (X^T as the "weight" operand at scale 1)
        dw = hip.gemm(hip.cast_bf16(dyt), hip.cast_bf16(xt), n=K, out_dtype=torch.float32)
    else:
        dw = hip.gemm(dyt, xt, n=K, out_dtype=torch.float32)
    return (dw, colsum(dy)) if bias else dw


def colsum(dy):
    M, N = dy.shape
    out = torch.empty((N,), device=dy.device, dtype=torch.float32)
    P = 64 if M >= 4096 else (16 if M >= 256 else 1)  # the row-chunk count csrc/backward.hip col_reduce uses for M rows
    part = torch.empty((P * N,), device=dy.device, dtype=torch.float32)
    _check(load().madtp_colsum(_p(dy), dy.stride(0), M, N, _p(out), _p(part), _stream()), "madtp_colsum")
    return out


def act_fwd(u, act):
    g = torch.empty_like(u)
    _check(load().madtp_act_fwd_bwd(_p(u), None, _p(g), None, u.numel(), act, _stream()), "madtp_act_fwd_bwd")
    return g


def act_bwd(u, dg, act):
    du = torch.empty_like(u)
    _check(load().madtp_act_fwd_bwd(_p(u), _p(dg), None, _p(du), u.numel(), act, _stream()), "madtp_act_fwd_bwd")
    return du


def dropout(x, p, seed, site, residual=None, per_sample=0):
    """madtp_dropout: residual + x * keep / (1 - p) with the counter-based mask of (seed, site); per_sample > 0: one draw per run of
    per_sample elements (DropPath).  The backward of the site is the same call on dY (without the residual)."""
    y = torch.empty_like(x)
    _check(load().madtp_dropout(_p(x), _p(residual), _p(y), x.numel(), int(per_sample), float(p), int(seed), int(site), _stream()),
           "madtp_dropout")
    return y


def attention_train(q, k, v, B, H, Nq, Nk, scale, drop, key_mask=None, mask_qk=None, scores=False):
    """madtp_attention_train: the training forward's attention with attention_probs dropout drop = (p, seed, site) -> (out [B*Nq, H*64],
    side or None) like hip.attention (exact f32; P is scratch)."""
    p, seed, site = drop
    out = torch.empty((B * Nq, H * 64), device=q.device, dtype=torch.float32)
    P = torch.empty((B * H * Nq * Nk,), device=q.device, dtype=torch.float32)
    cs = p0 = on = None
    if scores:
        cs = torch.empty((B, (Nq + 15) // 16, Nk), device=q.device, dtype=torch.float32)
        p0 = torch.empty((B, H, Nk), device=q.device, dtype=torch.float32)
        on = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    if k.stride(0) != v.stride(0):
        raise RuntimeError("attention_train: k and v must share a row stride")
    _check(load().madtp_attention_train(_p(q), q.stride(0), _p(k), _p(v), k.stride(0), _p(key_mask), _p(mask_qk),
                                        mask_qk.stride(0) if mask_qk is not None else 0, _p(P), _p(out), out.stride(0), _p(cs), _p(p0), _p(on),
                                        B, H, Nq, Nk, float(scale), float(p), int(seed), int(site), _stream()), "madtp_attention_train")
    return out, ((cs, p0, on) if scores else None)


def layernorm_bwd(x2d, gamma, dy2d, eps, add=None):
    rows, dim = x2d.shape
    dx = torch.empty_like(x2d)
    dgamma = torch.empty((dim,), device=x2d.device, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    ws = torch.empty((2 * rows + 64 * dim,), device=x2d.device, dtype=torch.float32)
    _check(load().madtp_layernorm_bwd(_p(x2d), _p(gamma), _p(dy2d), _p(add), _p(dx), _p(dgamma), _p(dbeta), _p(ws), rows, dim,
                                      float(eps), _stream()), "madtp_layernorm_bwd")
    return dx, dgamma, dbeta


def token_gather_bwd(dy, x_attn, dst_pos, merge_w, k):
    B, N, dim = x_attn.shape
    dx = torch.empty_like(x_attn)
    dw = torch.empty((B, N - 1), device=x_attn.device, dtype=torch.float32)
    _check(load().madtp_token_gather_bwd(_p(dy), _p(x_attn), _p(dst_pos), _p(merge_w), _p(dx), _p(dw), B, N, k, dim, _stream()),
           "madtp_token_gather_bwd")
    return dx, dw


def token_score_bwd(dw, score, dst_pos, merge_w, side, token_attn, B, H, N):
    cs, p0, on = side
    tp, ldr, ldb, K = hip._ta_view(token_attn)
    dev = dw.device
    da = torch.empty((B, N), device=dev, dtype=torch.float32)
    dp0 = torch.empty((B, H, N), device=dev, dtype=torch.float32)
    dnrm = torch.empty((B, H, N), device=dev, dtype=torch.float32)
    dta = torch.empty((B, N - 1, K), device=dev, dtype=torch.float32)
    _check(load().madtp_token_score_bwd(_p(dw), _p(score), _p(dst_pos), _p(merge_w), _p(cs), cs.shape[1], _p(p0), _p(on), tp, ldr,
                                        ldb, K, _p(da), _p(dp0), _p(dnrm), _p(dta), B, H, N, _stream()), "madtp_token_score_bwd")
    return da, dp0, dnrm, dta


def attention_bwd(qkv, dout, out, B, H, N, scale, dnrm=None, da=None, dp0=None, key_mask=None, mask_qk=None, dp_out=None, drop=None):
    """qkv f32 [B*N, 3*H*64] (fused projection), dout / out [B*N, H*64] -> dqkv [B*N, 3*H*64].  key_mask: additive f32 [B,N]
    over the keys (the BERT layers' padding mask) or None.  drop: (p, seed, site) of the forward's attention_probs dropout or None."""
    dp, dseed, dsite = drop if drop is not None else (0.0, 0, 0)
    D = H * 64
    dqkv = torch.empty_like(qkv)
    lib = load()
    nbytes = lib.madtp_attention_bwd_workspace(B, H, N)
    ws = torch.empty((nbytes,), device=qkv.device, dtype=torch.uint8)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
    _check(lib.madtp_attention_bwd(_p(q), _p(k), _p(v), qkv.stride(0), _p(key_mask), _p(mask_qk),
                                   mask_qk.stride(0) if mask_qk is not None else 0, _p(dout), dout.stride(0), _p(out), out.stride(0), _p(dnrm),
                                   _p(da), _p(dp0), _p(dq), _p(dk), _p(dv), dqkv.stride(0), _p(ws), nbytes, _p(dp_out), B, H, N, float(scale),
                                   float(dp), int(dseed), int(dsite), _stream()), "madtp_attention_bwd")
    return dqkv


def attention_bwd_cross(q, kv, dout, B, H, Nq, Nk, scale, key_mask=None, drop=None):
    """Cross-attention backward: q f32 [B*Nq, D], kv f32 [B*Nk, 2D] (fused [k|v] projection of the encoder tokens), dout [B*Nq, D]
    -> (dq [B*Nq, D], dkv [B*Nk, 2D]).  drop: (p, seed, site) of the forward's attention_probs dropout or None."""
    dp, dseed, dsite = drop if drop is not None else (0.0, 0, 0)
    D = H * 64
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    lib = load()
    nbytes = lib.madtp_attention_bwd_cross_workspace(B, H, Nq, Nk)
    ws = torch.empty((nbytes,), device=q.device, dtype=torch.uint8)
    k, v = kv[:, :D], kv[:, D:]
    dk, dv = dkv[:, :D], dkv[:, D:]
    _check(lib.madtp_attention_bwd_cross(_p(q), q.stride(0), _p(k), _p(v), kv.stride(0), _p(key_mask), _p(dout), dout.stride(0),
                                         _p(dq), dq.stride(0), _p(dk), _p(dv), dkv.stride(0), _p(ws), nbytes, B, H, Nq, Nk,
                                         float(scale), float(dp), int(dseed), int(dsite), _stream()), "madtp_attention_bwd_cross")
    return dq, dkv


def _pad_w(w):
    w = w.detach()
    n = w.shape[0]
    npad = _pad(n, 128)
    if npad != n:
        wp = torch.zeros((npad, w.shape[1]), device=w.device, dtype=torch.float32)
        wp[:n] = w
        w = wp
    return w.contiguous()


def _f32_lin(linear):
    """(weight padded to 128 rows, bias) of an nn.Linear for the recomputed forward's GEMM (cached per parameter version)."""
    b = None if linear.bias is None else linear.bias.detach().contiguous()
    if _x3_ok(linear.weight):
        return _x3_weights(("x3w", linear.weight.data_ptr(), tuple(linear.weight.shape)), [linear.weight]).planes, b
    w = _cached(("pw", linear.weight.data_ptr(), _x3()), [linear.weight], _versioned(lambda: _fresh(_pad_w(linear.weight))))
    return w, b


def _f32_wb(w, b):
    """(weight padded to 128 rows, bias) of a Linear given as tensors (cached per parameter version)."""
    if _x3_ok(w):
        return _x3_weights(("x3w", w.data_ptr(), tuple(w.shape)), [w]).planes, (None if b is None else b.detach().contiguous())
    wp = _cached(("pw", w.data_ptr(), _x3()), [w], _versioned(lambda: _fresh(_pad_w(w))))
    return wp, (None if b is None else b.detach().contiguous())


def _fresh(t):
    """a tensor OBJECT of the cache's own (attributes such as the split planes hang on it; never the parameter's storage view)"""
    return t.detach().view(t.shape) if t.requires_grad else t.view(t.shape)


def _cat_wb(tag, linears):
    """[w0; w1; ...] and [b0; b1; ...] of Linears that run as one fused projection (cached per parameter version)."""
    ws = [l.weight for l in linears]
    if all(_x3_ok(x) for x in ws):
        w = _x3_weights(("x3cat", tag) + tuple(x.data_ptr() for x in ws), ws).planes
        return w, torch.cat([l.bias.detach() for l in linears], 0).contiguous()
    w = _cached(("cat", tag, _x3()) + tuple(x.data_ptr() for x in ws), ws, _versioned(lambda: torch.cat([x.detach() for x in ws], 0).contiguous()))
    b = torch.cat([l.bias.detach() for l in linears], 0).contiguous()
    return w, b


class _BlockParts:
    """The modules / tensors of a pruned ViT block under one set of names: BLIP's Block (models/vit.py:106-207) and CLIP's
    ResidualAttentionBlock (clip/model.py:174-261: fused in_proj, QuickGELU, LayerNorm eps 1e-5) run the same layer call."""

    def __init__(self, blk):
        if hasattr(blk, "ln_1"):  # CLIP
            self.norm1, self.norm2 = blk.ln_1, blk.ln_2
            self.qkv_w, self.qkv_b = blk.attn.in_proj_weight, blk.attn.in_proj_bias
            self.proj, self.fc1, self.fc2 = blk.attn.out_proj, blk.mlp.c_fc, blk.mlp.c_proj
            self.H, self.scale, self.act = blk.n_head, (blk.d_model // blk.n_head) ** -0.5, hip.ACT_QUICK_GELU
            self.names = {"norm1": "ln_1", "norm2": "ln_2", "qkv_w": "attn.in_proj_weight", "qkv_b": "attn.in_proj_bias",
                          "proj": "attn.out_proj", "fc1": "mlp.c_fc", "fc2": "mlp.c_proj"}
        else:
            self.norm1, self.norm2 = blk.norm1, blk.norm2
            self.qkv_w, self.qkv_b = blk.attn.qkv.weight, blk.attn.qkv.bias
            self.proj, self.fc1, self.fc2 = blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2
            self.H, self.scale, self.act = blk.attn.num_heads, blk.attn.scale, hip.ACT_GELU
            self.names = {"norm1": "norm1", "norm2": "norm2", "qkv_w": "attn.qkv.weight", "qkv_b": "attn.qkv.bias",
                          "proj": "attn.proj", "fc1": "mlp.fc1", "fc2": "mlp.fc2"}

    def order(self):
        n = self.names
        return (n["norm1"] + ".weight", n["norm1"] + ".bias", n["qkv_w"], n["qkv_b"], n["proj"] + ".weight", n["proj"] + ".bias",
                n["norm2"] + ".weight", n["norm2"] + ".bias", n["fc1"] + ".weight", n["fc1"] + ".bias", n["fc2"] + ".weight",
                n["fc2"] + ".bias")

    def params(self):
        return [self.norm1.weight, self.norm1.bias, self.qkv_w, self.qkv_b, self.proj.weight, self.proj.bias, self.norm2.weight,
                self.norm2.bias, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias]


def _note_versions(ctx, params):
    """the backward reads the module's CURRENT parameters (W^T planes per parameter version) next to activations of the forward: an
    in-place update between the two (optimizer.step, load_state_dict, an EMA) would pair them wrongly - native autograd raises
    'modified by an inplace operation' there, so do these Functions"""
    ctx.param_versions = _versions_of(params)


def _versions_of(params):
    """(process-wide epoch of hand edits, then per parameter: version and optimizer step count).  The step count is per PARAMETER
    (runtime._on_optimizer_step): a step of another optimizer - a second model, a GAN's other half, a teacher - between this
    module's forward and backward does not touch it."""
    from .runtime import update_epoch, param_step_count
    return (update_epoch(),) + tuple((p._version, param_step_count(p)) if p is not None else -1 for p in params)


def _check_versions(ctx, params, what):
    now = _versions_of(params)
    if now != ctx.param_versions:
        raise RuntimeError(f"{what}: a parameter was modified in place between forward and backward (versions {ctx.param_versions} -> "
                           f"{now}); back-propagate before the optimizer step / load_state_dict, as native autograd requires")


def _second_backward_guard(ctx, what):
    """The first backward releases the kept activations (ctx.saved); a second one over a retained graph recomputes the layer - which is
    the same forward only WITHOUT dropout / DropPath (the recompute draws no masks).  In model.train() with active rates that would be
    gradients of a different forward than the one that produced the loss: refuse instead."""
    if ctx.bwd_done and ctx.had_drop:
        raise RuntimeError(f"{what}: a second backward over a retained graph of a forward that applied dropout / DropPath is not supported "
                           "(the kept activations and masks were released by the first backward); run the forward again")
    ctx.bwd_done = True


_WARNED_IMPLICIT = [False]


def _warn_implicit(wants):
    """the autograd route was taken only because the parameters have requires_grad = True (the nn.Module default), not because an
    input asks for a gradient: say so once - a forgotten torch.no_grad() otherwise becomes a silent slowdown (no encoder-level call,
    activations kept)"""
    if wants or _WARNED_IMPLICIT[0]:
        return
    _WARNED_IMPLICIT[0] = True
    import warnings
    warnings.warn("madtp_amd: a forward in grad mode takes the autograd route (hand-written backward, activations kept) because the "
                  "module's parameters require grad; wrap inference in torch.no_grad() to get the fused inference path", RuntimeWarning,
                  stacklevel=3)


def _save_forward():
    """MADTP_TRAIN_SAVE (default 1): the training forward of a block is composed from the single kernels and KEEPS its intermediates
    for the backward (round 5) - instead of running the fused layer call and recomputing the layer in the backward (0: the
    round-4 scheme, a third less activation memory, one more forward's worth of kernels per step)."""
    # (the amp route always keeps its forward: a recompute on other arithmetic than the fused fast-mode layer call could take another
    #  pruning decision than the forward whose output shape the graph already has)
    return os.environ.get("MADTP_TRAIN_SAVE", "1") != "0" or _lp()


def _site(base, code):
    """site id of dropout site `code` of the layer call that drew `base` (runtime.next_dropout_base)"""
    return base * 32 + code


def _block_drop(blk):
    """(p, seed, base) of a ViT block's DropPath in this forward, or None (eval mode / rate 0)"""
    p = float(getattr(blk, "drop_path_rate", 0.0) or 0.0)
    if not blk.training or p <= 0.0:
        return None
    from .runtime import next_dropout_base
    seed, base = next_dropout_base()
    return (p, seed, base)


def _layer_drop(layer):
    """(p_hidden, p_attn, seed, base) of a BERT layer's dropouts in this forward, or None"""
    if not layer.training:
        return None
    ph = float(layer.output.dropout.p)
    pa = float(layer.attention.self.dropout.p)
    if ph <= 0.0 and pa <= 0.0:
        return None
    from .runtime import next_dropout_base
    seed, base = next_dropout_base()
    return (ph, pa, seed, base)


class _Saved:
    """intermediates of a block's forward, by name"""
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _vit_block_fwd(blk, x, token_attn, temperature, k, mask_qk=None, max_keep=0, drop=None):
    """Block.forward from single kernels (the forward's own: same operand formats and dispatch in the f16x3 mode), keeping every
    intermediate the backward reads.  k: the forward's pruning decision (recompute), or None: decide it here - k = max_b count read
    on the host, vit.py:145, kept unless k <= max_keep or fewer than two tokens would go (madtp_vit_block_keep's rule)."""
    P = _BlockParts(blk)
    B, N, D = x.shape
    H, scale = P.H, P.scale
    M = B * N
    eps1, eps2 = P.norm1.eps, P.norm2.eps
    x2 = x.reshape(M, D)
    h1, _ = hip.layernorm(x2, P.norm1.weight.detach(), P.norm1.bias.detach(), eps1)
    wq, bq = _f32_wb(P.qkv_w, P.qkv_b)
    wp, bp = _f32_lin(P.proj)
    w1, b1 = _f32_lin(P.fc1)
    qkv = _gemm(h1, wq, bq, n=3 * D, out_dtype=torch.float32)
    prune = temperature > 0 if k is None else k > 0
    out, side = _attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, scale, scores=prune, mask_qk=mask_qk)
    if drop is not None:  # x + drop_path(attn(norm1(x))), vit.py:186: one draw per sample
        x_attn = dropout(_gemm(out, wp, bp, n=D, out_dtype=torch.float32), drop[0], drop[1], _site(drop[2], 0), residual=x2, per_sample=N * D)
    else:
        x_attn = _gemm(out, wp, bp, residual=x2, n=D, out_dtype=torch.float32)
    score = dst_pos = merge_w = info = None
    if prune and k is None:
        score, thr, count, kk = hip.token_score_sync(side, token_attn, temperature, B, H, N)
        info = {"k": kk, "score": score, "threshold": thr, "count": count, "pruned": False, "indices": None, "indices_sort": None}
        k = 0 if (kk <= max_keep or (N - 1 - kk) <= 1) else kk
    elif prune:
        score, _, _, _ = hip.token_score(side, token_attn, temperature, B, H, N)
    k = k or 0
    if k > 0:
        indices, indices_sort, dst_pos, merge_w = hip.token_select(score, k)
        if info is not None:
            info.update(pruned=True, indices=indices, indices_sort=indices_sort)
        y0 = hip.token_gather(x_attn.view(B, N, D), dst_pos, merge_w, k)
    else:
        y0 = x_attn.view(B, N, D)
    N2 = y0.shape[1]
    y02 = y0.reshape(B * N2, D)
    h2, _ = hip.layernorm(y02, P.norm2.weight.detach(), P.norm2.bias.detach(), eps2)
    u = _gemm(h2, w1, b1, n=P.fc1.weight.shape[0], out_dtype=torch.float32)
    g = act_fwd(u, P.act)
    return _Saved(k=k, info=info, h1=h1, qkv=qkv, out=out, side=side, x_attn=x_attn, score=score, dst_pos=dst_pos, merge_w=merge_w,
                  y02=y02, h2=h2, u=u, g=g, drop=drop)


def vit_block_forward_saved(blk, x, token_attn, temperature, max_keep=0, mask_qk=None):
    """-> (y [B,N',D], info, saved): Block.forward for training, see _save_forward()."""
    P = _BlockParts(blk)
    B, _, D = x.shape
    drop = _block_drop(blk)
    s = _vit_block_fwd(blk, x, token_attn, temperature, None, mask_qk=mask_qk, max_keep=max_keep, drop=drop)
    w2, b2 = _f32_lin(P.fc2)
    if drop is not None:  # x + drop_path(mlp(norm2(x))), vit.py:205
        y = dropout(_gemm(s.g, w2, b2, n=D, out_dtype=torch.float32), drop[0], drop[1], _site(drop[2], 1), residual=s.y02,
                    per_sample=(s.y02.shape[0] // B) * D)
    else:
        y = _gemm(s.g, w2, b2, residual=s.y02, n=D, out_dtype=torch.float32)
    return y.view(B, -1, D), s.info, s


def vit_block_backward(blk, x, token_attn, temperature, k, dy, mask_qk=None, dp_out=None, saved=None):
    """Gradients of Block.forward (vit.py:183-207; clip/model.py:236-261 for CLIP's block) at (x [B,N,D], token_attn [B,N-1,K])
    for the output gradient dy [B,N',D], with the forward's pruning decision k (0 = the layer was not pruned); mask_qk: the
    additive [N,N] attention mask of CLIP's text tower or None.  saved: the forward's intermediates (vit_block_forward_saved), else
    the forward is recomputed from (x, token_attn, k).  Returns (dx, dtoken_attn or None, {parameter name: grad})."""
    P = _BlockParts(blk)
    nm = P.names
    B, N, D = x.shape
    H, scale = P.H, P.scale
    M = B * N
    eps1, eps2 = P.norm1.eps, P.norm2.eps
    x2 = x.reshape(M, D)
    s = saved if saved is not None else _vit_block_fwd(blk, x, token_attn, temperature, k, mask_qk=mask_qk)
    h1, qkv, out, side, x_attn, score, dst_pos, merge_w = s.h1, s.qkv, s.out, s.side, s.x_attn, s.score, s.dst_pos, s.merge_w
    y02, h2, u, g = s.y02, s.h2, s.u, s.g
    M2 = y02.shape[0]
    N2 = M2 // B
    # ---- backward ----
    grads = {}
    drop = getattr(s, "drop", None)
    dy2 = dy.reshape(M2, D).contiguous().float()
    dyb = dropout(dy2, drop[0], drop[1], _site(drop[2], 1), per_sample=N2 * D) if drop is not None else dy2  # the MLP branch's share
    dg = dgrad(dyb, P.fc2.weight.detach())               # y = y0 + drop_path(g W2^T + b2)
    grads[nm["fc2"] + ".weight"], grads[nm["fc2"] + ".bias"] = wgrad(dyb, g, bias=True)
    du = act_bwd(u, dg, P.act)
    dh2 = dgrad(du, P.fc1.weight.detach())
    grads[nm["fc1"] + ".weight"], grads[nm["fc1"] + ".bias"] = wgrad(du, h2, bias=True)
    dy0, grads[nm["norm2"] + ".weight"], grads[nm["norm2"] + ".bias"] = layernorm_bwd(y02, P.norm2.weight.detach(), dh2, eps2, add=dy2)
    dta = None
    dnrm = da = dp0 = None
    if k > 0:
        dx_attn, dw = token_gather_bwd(dy0.view(B, N2, D), x_attn.view(B, N, D), dst_pos, merge_w, k)
        da, dp0, dnrm, dta = token_score_bwd(dw, score, dst_pos, merge_w, side, token_attn, B, H, N)
        dxa2 = dx_attn.view(M, D)
    else:
        dxa2 = dy0
    dpb = dropout(dxa2.contiguous(), drop[0], drop[1], _site(drop[2], 0), per_sample=N * D) if drop is not None else dxa2
    dout = dgrad(dpb, P.proj.weight.detach())           # x_attn = x + drop_path(out Wp^T + bp)
    grads[nm["proj"] + ".weight"], grads[nm["proj"] + ".bias"] = wgrad(dpb, out, bias=True)
    dqkv = attention_bwd(qkv, dout, out, B, H, N, scale, dnrm, da, dp0, mask_qk=mask_qk, dp_out=dp_out)
    dh1 = dgrad(dqkv, P.qkv_w.detach())
    if P.qkv_b is not None:
        grads[nm["qkv_w"]], grads[nm["qkv_b"]] = wgrad(dqkv, h1, bias=True)
    else:
        grads[nm["qkv_w"]] = wgrad(dqkv, h1)
    dx2, grads[nm["norm1"] + ".weight"], grads[nm["norm1"] + ".bias"] = layernorm_bwd(x2, P.norm1.weight.detach(), dh1, eps1, add=dxa2)
    return dx2.view(B, N, D), dta, grads


class VitBlockFunction(torch.autograd.Function):
    """Block.forward (BLIP Block or CLIP ResidualAttentionBlock) with a hand-written backward.  Inputs after (blk, temperature,
    max_keep): x, token_attn (or None), then the block's 12 parameters in _BlockParts.order() (passed so that autograd routes
    their gradients; the kernels read them from the module)."""

    @staticmethod
    def forward(ctx, blk, temperature, max_keep, x, token_attn, *params):
        ctx.mode = _mode()
        _note_versions(ctx, params)
        prune = temperature > 0
        ctx.saved = None
        ctx.had_drop, ctx.bwd_done = bool(blk.training and float(getattr(blk, "drop_path_rate", 0.0) or 0.0) > 0.0), False
        if _save_forward() or (blk.training and float(getattr(blk, "drop_path_rate", 0.0) or 0.0) > 0.0):  # (DropPath: the kept forward only)
            if getattr(blk, "attn_mask", None) is not None and getattr(blk, "_mask_dev", None) is None:
                blk._weights()  # (creates the device copy of CLIP's text attention mask)
            mask = getattr(blk, "_mask_dev", None) if getattr(blk, "attn_mask", None) is not None else None
            y, info, ctx.saved = vit_block_forward_saved(blk, x, token_attn, temperature if prune else 0, max_keep=max_keep, mask_qk=mask)
        else:
            y, info = hip.vit_block(blk._weights(), x, token_attn, temperature if prune else 0, max_keep=max_keep)
        blk.last_prune = info
        ctx.blk, ctx.temperature = blk, float(temperature)
        ctx.k = int(info["indices"].shape[1]) if (info is not None and info.get("pruned")) else 0
        ctx.save_for_backward(x, token_attn if token_attn is not None else x.new_empty(0))
        ctx.has_ta = token_attn is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ta = ctx.saved_tensors
        ta = ta if ctx.has_ta else None
        _check_versions(ctx, _BlockParts(ctx.blk).params(), "Block backward")
        _second_backward_guard(ctx, "Block backward")
        mask = getattr(ctx.blk, "_mask_dev", None) if getattr(ctx.blk, "attn_mask", None) is not None else None
        # (the full [ctx, ctx] mask: the kernels read its leading [N, N] corner through the row stride, clip/mock.py:309-310)
        dp_out = None
        att = getattr(ctx.blk, "attn", None)
        if getattr(att, "_hook_attn_gradients", False):  # Block.forward(register_hook=True): vit.py:88-90, the Grad-CAM hook
            att._hook_attn_gradients = False
            B_, N_ = x.shape[0], x.shape[1]
            dp_out = torch.empty((B_, att.num_heads, N_, N_), device=x.device, dtype=torch.float32)
        with torch.no_grad(), _in_mode(ctx.mode):
            dx, dta, grads = vit_block_backward(ctx.blk, x, ta, ctx.temperature, ctx.k, dy, mask_qk=mask, dp_out=dp_out, saved=ctx.saved)
            ctx.saved = None
        if dp_out is not None:
            att.save_attn_gradients(dp_out)
        pg = [grads.get(name) for name in _BlockParts(ctx.blk).order()]
        if ctx.has_ta and dta is None:
            dta = torch.zeros_like(ta)
        return (None, None, None, dx, dta if ctx.has_ta else None) + tuple(pg)


def block_forward_with_grad(blk, x, temperature, token_attn, max_keep=0):
    """Block.forward under autograd (called by madtp_amd.vit.Block.forward when gradients are required)."""
    _check_mode("the block backward")
    _warn_implicit(x.requires_grad or (token_attn is not None and token_attn.requires_grad))
    if token_attn is not None and not token_attn.is_contiguous():
        token_attn = token_attn.contiguous()
    return VitBlockFunction.apply(blk, temperature, max_keep, x, token_attn, *_BlockParts(blk).params())


# ---------------------------------------------------------------------------------------------------------------------------
# MED layer (models/med.py BertLayer.forward :393-462): self-attention with the padding mask, output LayerNorm, Reduce_token on
# the post-LN tokens (:345-391 - the same importance score and merge rule as the ViT block), in mode 'multimodal' cross-attention
# to the image tokens, FFN with post-LayerNorm.  Same recipe as vit_block_backward: recompute the forward from (hidden, token_attn) with the forward's own fp32
# kernels and the forward's k, then walk the graph backwards.

_MED_PARAMS = ("attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
               "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight",
               "attention.output.dense.bias", "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
               "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight", "output.dense.bias",
               "output.LayerNorm.weight", "output.LayerNorm.bias")
_MED_CROSS_PARAMS = ("crossattention.self.query.weight", "crossattention.self.query.bias", "crossattention.self.key.weight",
                     "crossattention.self.key.bias", "crossattention.self.value.weight", "crossattention.self.value.bias",
                     "crossattention.output.dense.weight", "crossattention.output.dense.bias",
                     "crossattention.output.LayerNorm.weight", "crossattention.output.LayerNorm.bias")


def _nlvr_cross_params(layer):
    names = []
    for br in ("self0", "self1"):
        for nm in ("query", "key", "value"):
            names += [f"crossattention.{br}.{nm}.weight", f"crossattention.{br}.{nm}.bias"]
    names += ["crossattention.output.dense0.weight", "crossattention.output.dense0.bias", "crossattention.output.dense1.weight",
              "crossattention.output.dense1.bias"]
    if layer.crossattention.output.merge:
        names += ["crossattention.output.merge_layer.weight", "crossattention.output.merge_layer.bias"]
    return tuple(names + ["crossattention.output.LayerNorm.weight", "crossattention.output.LayerNorm.bias"])


def _layer_param_names(layer, cross):
    if not cross:
        return _MED_PARAMS
    return _MED_PARAMS + (_nlvr_cross_params(layer) if layer.variant == "nlvr" else _MED_CROSS_PARAMS)


def _med_params_of(layer, cross):
    mods = dict(layer.named_parameters())
    return [mods[n] for n in _layer_param_names(layer, cross)]


def _cross_branch_fwd(sm, y02, enc2, B, H, L2, Nk, scale, key_mask, drop=None):
    """One cross-attention branch (BertSelfAttention with encoder_hidden_states): -> (cq, ckv, cctx, wckv).  drop: (p, seed, site) of
    the attention_probs dropout or None."""
    D = H * 64
    wcq, bcq = _f32_lin(sm.query)
    wckv, bckv = _cat_wb("ckv", [sm.key, sm.value])  # [2D, Denc]
    cq = _gemm(y02, wcq, bcq, n=D, out_dtype=torch.float32)
    ckv = _gemm(enc2, wckv, bckv, n=2 * D, out_dtype=torch.float32)
    if drop is not None:
        cctx, _ = attention_train(cq, ckv[:, :D], ckv[:, D:], B, H, L2, Nk, scale, drop, key_mask=key_mask)
    else:
        cctx, _ = _attention(cq, ckv[:, :D], ckv[:, D:], B, H, L2, Nk, scale, add_mask=key_mask)
    return cq, ckv, cctx, wckv


def _cross_branch_bwd(prefix, sm, grads, dcctx, cq, ckv, wckv, y02, enc2, B, H, L2, Nk, scale, key_mask, dy0_acc, drop=None):
    """Backward of one cross-attention branch; -> (dy0_acc + its contribution to the layer tokens, d encoder tokens [B*Nk, Denc])."""
    D = H * 64
    dcq, dckv = attention_bwd_cross(cq, ckv, dcctx, B, H, L2, Nk, scale, key_mask=key_mask, drop=drop)
    dy0 = dgrad(dcq, sm.query.weight.detach(), residual=dy0_acc)
    grads[prefix + "query.weight"], grads[prefix + "query.bias"] = wgrad(dcq, y02, bias=True)
    denc = dgrad(dckv, wckv)
    gw, gb = wgrad(dckv, enc2, bias=True)
    for i, nm in enumerate(("key", "value")):
        grads[prefix + nm + ".weight"] = gw[i * D:(i + 1) * D]
        grads[prefix + nm + ".bias"] = gb[i * D:(i + 1) * D]
    return dy0, denc


def _med_layer_fwd(layer, hidden, mask2d, token_attn, temperature, k, enc=None, enc_masks=None, causal=None, drop=None):
    """BertLayer.forward (MED / NLVR) from single kernels, keeping every intermediate the backward reads.  k: the forward's pruning
    decision (recompute), or None: decide it here (k = max_b count on the host, med.py:374-375; kept unless k < 1 or fewer than two
    tokens would go - madtp_bert_layer's rule)."""
    B, L, D = hidden.shape
    sa, so = layer.attention.self, layer.attention.output
    H, scale = sa.num_attention_heads, 1.0 / math.sqrt(sa.attention_head_size)
    M = B * L
    h2 = hidden.reshape(M, D)
    twin = isinstance(enc, (list, tuple))
    wqkv, bqkv = _cat_wb("qkv", [sa.query, sa.key, sa.value])  # [3D, D]
    wo, bo = _f32_lin(so.dense)
    wi, bi = _f32_lin(layer.intermediate.dense)
    wout, bout = _f32_lin(layer.output.dense)
    qkv = _gemm(h2, wqkv, bqkv, n=3 * D, out_dtype=torch.float32)
    prune = temperature > 0 if k is None else k > 0
    # drop = (p_hidden, p_attn, seed, base): the layer's dropouts in .train() mode (med.py:55,111,244,323); sites: 0 self-attention
    # probabilities, 1 self-output, 2 / 3 cross-attention probabilities (branch 0 / 1), 4 cross-output, 5 FFN output
    ph, pa = (drop[0], drop[1]) if drop is not None else (0.0, 0.0)
    adrop = (lambda code: (pa, drop[2], _site(drop[3], code))) if pa > 0.0 else (lambda code: None)
    hdrop = (lambda t, res, code: dropout(t, ph, drop[2], _site(drop[3], code), residual=res))
    if pa > 0.0:
        ctx, side = attention_train(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, L, L, scale, adrop(0), key_mask=mask2d, mask_qk=causal,
                                    scores=prune)
    else:
        ctx, side = _attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, L, L, scale, add_mask=mask2d, scores=prune,
                               mask_qk=causal)   # causal [L,L]: the decoder's mask (med.py:752-786), never together with pruning
    if ph > 0.0:
        a0 = hdrop(_gemm(ctx, wo, bo, n=D, out_dtype=torch.float32), h2, 1)   # LayerNorm(dropout(dense(ctx)) + input), med.py:246-250
    else:
        a0 = _gemm(ctx, wo, bo, residual=h2, n=D, out_dtype=torch.float32)
    ao, _ = hip.layernorm(a0, so.LayerNorm.weight.detach(), so.LayerNorm.bias.detach(), so.LayerNorm.eps)
    score = dst_pos = merge_w = info = mask_out = None
    if prune and k is None:
        score, thr, count, kk = hip.token_score_sync(side, token_attn, temperature, B, H, L)
        info = {"k": kk, "score": score, "threshold": thr, "count": count, "pruned": False, "indices": None, "indices_sort": None}
        k = 0 if (kk < 1 or (L - 1 - kk) <= 1) else kk
    elif prune:
        score, _, _, _ = hip.token_score(side, token_attn, temperature, B, H, L)
    k = k or 0
    if k > 0:
        indices, indices_sort, dst_pos, merge_w = hip.token_select(score, k)
        if info is not None:
            info.update(pruned=True, indices=indices, indices_sort=indices_sort)
            if mask2d is not None:  # the pruned sequence's additive mask (nlvr_encoder.py:451-452 / med.py:386-389)
                mask_out = (hip.mask_gather(mask2d, indices_sort, k) if getattr(layer, "variant", "") == "nlvr"
                            else hip.mask_gather(mask2d, indices, k, indices_sort))
        y0 = hip.token_gather(ao.view(B, L, D), dst_pos, merge_w, k)
    else:
        y0 = ao.view(B, L, D)
    L2 = y0.shape[1]
    M2 = B * L2
    y02 = y0.reshape(M2, D)
    br = enc2s = sms = ems = c0 = d01 = None
    Nk = De = 0
    if enc is not None:
        ca, co = layer.crossattention, layer.crossattention.output
        encs = list(enc) if twin else [enc]
        sms = [ca.self0, ca.self1] if twin else [ca.self]
        ems = list(enc_masks) if (twin and enc_masks is not None) else [None] * len(encs)
        Nk, De = encs[0].shape[1], encs[0].shape[2]
        enc2s = [e.reshape(B * Nk, De).contiguous().float() for e in encs]
        br = [_cross_branch_fwd(sm, y02, e2, B, H, L2, Nk, scale, em, drop=adrop(2 + i)) for i, (sm, e2, em) in enumerate(zip(sms, enc2s, ems))]
        if twin:
            w0, b0 = _f32_lin(co.dense0)
            w1, b1 = _f32_lin(co.dense1)
            d0 = _gemm(br[0][2], w0, b0, n=D, out_dtype=torch.float32)
            d1 = _gemm(br[1][2], w1, b1, n=D, out_dtype=torch.float32)
            if co.merge:
                d01 = torch.cat([d0, d1], 1).contiguous()  # (a copy: the operand layout of merge_layer)
                wm, bm = _f32_lin(co.merge_layer)
                c0 = (hdrop(_gemm(d01, wm, bm, n=D, out_dtype=torch.float32), y02, 4) if ph > 0.0   # nlvr_encoder.py:259-271
                      else _gemm(d01, wm, bm, residual=y02, n=D, out_dtype=torch.float32))
            else:
                c0 = hdrop(((d0 + d1) * 0.5).contiguous(), y02, 4) if ph > 0.0 else (d0 + d1) * 0.5 + y02
        else:
            wcd, bcd = _f32_lin(co.dense)
            c0 = (hdrop(_gemm(br[0][2], wcd, bcd, n=D, out_dtype=torch.float32), y02, 4) if ph > 0.0
                  else _gemm(br[0][2], wcd, bcd, residual=y02, n=D, out_dtype=torch.float32))
        x2, _ = hip.layernorm(c0, co.LayerNorm.weight.detach(), co.LayerNorm.bias.detach(), co.LayerNorm.eps)
    else:
        x2 = y02
    F = layer.intermediate.dense.weight.shape[0]
    u = _gemm(x2, wi, bi, n=F, out_dtype=torch.float32)
    gl = act_fwd(u, hip.ACT_GELU)
    f0 = (hdrop(_gemm(gl, wout, bout, n=D, out_dtype=torch.float32), x2, 5) if ph > 0.0                # med.py:326-328
          else _gemm(gl, wout, bout, residual=x2, n=D, out_dtype=torch.float32))
    return _Saved(drop=drop, k=k, info=info, mask_out=mask_out, wqkv=wqkv, qkv=qkv, ctx=ctx, side=side, a0=a0, ao=ao, score=score, dst_pos=dst_pos,
                  merge_w=merge_w, y02=y02, br=br, enc2s=enc2s, sms=sms, ems=ems, Nk=Nk, De=De, c0=c0, d01=d01, x2=x2, u=u, gl=gl, f0=f0)


def med_layer_forward_saved(layer, hidden, mask2d, token_attn, temperature, enc=None, enc_masks=None, causal=None):
    """-> (y [B,L',D], mask_out or None, info, saved): BertLayer.forward for training, see _save_forward()."""
    s = _med_layer_fwd(layer, hidden, mask2d, token_attn, temperature, None, enc, enc_masks, causal, drop=_layer_drop(layer))
    ln2 = layer.output.LayerNorm
    y, _ = hip.layernorm(s.f0, ln2.weight.detach(), ln2.bias.detach(), ln2.eps)
    return y.view(hidden.shape[0], -1, hidden.shape[2]), s.mask_out, s.info, s


def med_layer_backward(layer, hidden, mask2d, token_attn, temperature, k, dy, enc=None, enc_masks=None, causal=None, saved=None):
    """Gradients of a BertLayer - MED (med.py:393-462) or NLVR (nlvr_encoder.py:484-554) - at (hidden [B,L,D], token_attn
    [B,L-1,K]) for the output gradient dy [B,L',D]; mask2d: additive key mask [B,L] or None; k: the forward's pruning decision
    (0 = not pruned).  enc: None (mode 'text'), the encoder tokens [B,Nk,Denc] (MED, mode 'multimodal': one cross-attention between
    the pruning step and the FFN, encoder mask ignored, med.py:197-199) or a list of two (NLVR: twin branches self0 / self1 whose
    outputs are averaged, or merged by merge_layer from layer 6 on, nlvr_encoder.py:259-266; enc_masks: their additive key masks
    [B,Nk] or None, applied as nlvr_encoder.py:196-198 does).  saved: the forward's intermediates (med_layer_forward_saved), else the
    forward is recomputed.  Returns (dhidden, dtoken_attn or None, denc (None / tensor / list), {parameter name: grad})."""
    B, L, D = hidden.shape
    sa, so = layer.attention.self, layer.attention.output
    H, scale = sa.num_attention_heads, 1.0 / math.sqrt(sa.attention_head_size)
    M = B * L
    h2 = hidden.reshape(M, D)
    twin = isinstance(enc, (list, tuple))
    s = saved if saved is not None else _med_layer_fwd(layer, hidden, mask2d, token_attn, temperature, k, enc, enc_masks, causal)
    wqkv, qkv, ctx, side, a0, ao, score, dst_pos, merge_w = s.wqkv, s.qkv, s.ctx, s.side, s.a0, s.ao, s.score, s.dst_pos, s.merge_w
    y02, br, enc2s, sms, ems, Nk, De, c0, d01, x2, u, gl, f0 = s.y02, s.br, s.enc2s, s.sms, s.ems, s.Nk, s.De, s.c0, s.d01, s.x2, s.u, s.gl, s.f0
    M2 = y02.shape[0]
    L2 = M2 // B
    if enc is not None:
        co = layer.crossattention.output
    # ---- backward ----
    grads = {}
    ln2 = layer.output.LayerNorm
    dy2 = dy.reshape(M2, D).contiguous().float()
    df0, grads["output.LayerNorm.weight"], grads["output.LayerNorm.bias"] = layernorm_bwd(f0, ln2.weight.detach(), dy2, ln2.eps)
    # the forward's dropouts (see _med_layer_fwd): the branch behind a hidden dropout receives mask o d / (1 - p), the residual all of d
    drop = getattr(s, "drop", None)
    ph, pa = (drop[0], drop[1]) if drop is not None else (0.0, 0.0)
    hd = (lambda t, code: dropout(t.contiguous(), ph, drop[2], _site(drop[3], code))) if ph > 0.0 else (lambda t, code: t)
    adrop = (lambda code: (pa, drop[2], _site(drop[3], code))) if pa > 0.0 else (lambda code: None)
    dfb = hd(df0, 5)
    dgl = dgrad(dfb, layer.output.dense.weight.detach())
    grads["output.dense.weight"], grads["output.dense.bias"] = wgrad(dfb, gl, bias=True)
    du = act_bwd(u, dgl, hip.ACT_GELU)
    dx2 = dgrad(du, layer.intermediate.dense.weight.detach(), residual=df0)   # f0 = x2 + ffn(x2)
    grads["intermediate.dense.weight"], grads["intermediate.dense.bias"] = wgrad(du, x2, bias=True)
    denc = None
    if enc is not None:
        P = "crossattention."
        dc0, grads[P + "output.LayerNorm.weight"], grads[P + "output.LayerNorm.bias"] = layernorm_bwd(
            c0, co.LayerNorm.weight.detach(), dx2, co.LayerNorm.eps)          # c0 = y0 + combine(branches)
        dcb = hd(dc0, 4)
        if twin:
            if co.merge:
                dd01 = dgrad(dcb, co.merge_layer.weight.detach())             # [M2, 2D]
                grads[P + "output.merge_layer.weight"], grads[P + "output.merge_layer.bias"] = wgrad(dcb, d01, bias=True)
                dds = [dd01[:, :D].contiguous(), dd01[:, D:].contiguous()]
            else:
                half = dcb * 0.5
                dds = [half, half]
            dy0, dencs = dc0, []
            for i, (sm, dd) in enumerate(zip(sms, dds)):
                dn = (co.dense0, co.dense1)[i]
                cq, ckv, cctx, wckv = br[i]
                dcctx = dgrad(dd, dn.weight.detach())
                grads[P + f"output.dense{i}.weight"], grads[P + f"output.dense{i}.bias"] = wgrad(dd, cctx, bias=True)
                dy0, de = _cross_branch_bwd(P + f"self{i}.", sm, grads, dcctx, cq, ckv, wckv, y02, enc2s[i], B, H, L2, Nk, scale,
                                            ems[i], dy0, drop=adrop(2 + i))
                dencs.append(de.view(B, Nk, De))
            denc = dencs
        else:
            cq, ckv, cctx, wckv = br[0]
            dcctx = dgrad(dcb, co.dense.weight.detach())
            grads[P + "output.dense.weight"], grads[P + "output.dense.bias"] = wgrad(dcb, cctx, bias=True)
            dy0, de = _cross_branch_bwd(P + "self.", sms[0], grads, dcctx, cq, ckv, wckv, y02, enc2s[0], B, H, L2, Nk, scale, None, dc0,
                                        drop=adrop(2))
            denc = de.view(B, Nk, De)
    else:
        dy0 = dx2
    dta = dnrm = da = dp0 = None
    if k > 0:
        dao3, dw = token_gather_bwd(dy0.view(B, L2, D), ao.view(B, L, D), dst_pos, merge_w, k)
        da, dp0, dnrm, dta = token_score_bwd(dw, score, dst_pos, merge_w, side, token_attn, B, H, L)
        dao = dao3.view(M, D)
    else:
        dao = dy0
    ln1 = so.LayerNorm
    da0, grads["attention.output.LayerNorm.weight"], grads["attention.output.LayerNorm.bias"] = layernorm_bwd(
        a0, ln1.weight.detach(), dao.contiguous(), ln1.eps)
    dab = hd(da0, 1)
    dctx = dgrad(dab, so.dense.weight.detach())                               # a0 = hidden + dropout(ctx Wo^T + bo)
    grads["attention.output.dense.weight"], grads["attention.output.dense.bias"] = wgrad(dab, ctx, bias=True)
    dqkv = attention_bwd(qkv, dctx, ctx, B, H, L, scale, dnrm, da, dp0, key_mask=mask2d, mask_qk=causal, drop=adrop(0))
    dh = dgrad(dqkv, wqkv, residual=da0)
    gw, gb = wgrad(dqkv, h2, bias=True)
    for i, nm in enumerate(("query", "key", "value")):
        grads[f"attention.self.{nm}.weight"] = gw[i * D:(i + 1) * D]
        grads[f"attention.self.{nm}.bias"] = gb[i * D:(i + 1) * D]
    return dh.view(B, L, D), dta, denc, grads


class MedLayerFunction(torch.autograd.Function):
    """BertLayer.forward (MED: modes 'text' / 'multimodal'; NLVR: twin cross-attention) with a hand-written backward.  Inputs after
    (layer, temperature, mask2d, enc_masks): hidden, token_attn (or None), enc0, enc1 (encoder tokens [B,Nk,Denc] or None; MED uses
    enc0 only), then the layer's parameters in _layer_param_names order.  Returns (layer output, new additive mask [B,L'])."""

    @staticmethod
    def forward(ctx, layer, temperature, mask2d, enc_masks, causal, hidden, token_attn, enc0, enc1, *params):
        ctx.mode = _mode()
        _note_versions(ctx, params)
        prune = temperature > 0
        cross = enc0 is not None
        twin = enc1 is not None
        from .runtime import to_compute
        # encoder tokens as the layer's GEMM operand: f32 in the fp32 mode, f16-split planes in the f16x3 mode
        flat = lambda e: to_compute(e.reshape(-1, e.shape[-1]).contiguous().float())
        em = enc_masks if enc_masks is not None else (None, None)
        ctx.causal = causal
        ctx.saved = None
        ctx.had_drop, ctx.bwd_done = bool(layer.training and (layer.output.dropout.p > 0 or layer.attention.self.dropout.p > 0)), False
        if _save_forward() or (layer.training and (layer.output.dropout.p > 0 or layer.attention.self.dropout.p > 0)):  # (dropout: the
            enc = ([enc0, enc1] if twin else enc0) if cross else None                                                  # kept forward only)
            y, mask_out, info, ctx.saved = med_layer_forward_saved(layer, hidden, mask2d, token_attn, temperature if prune else 0, enc,
                                                                   em if twin else None, causal)
        else:
            w = layer._weights()
            if causal is not None:  # a copy of the cached struct with this call's causal mask (as BertLayer._forward does)
                w = hip.BertLayerW.from_buffer_copy(w)
                w.self_mask_qk, w.ld_self_mask_qk = causal.data_ptr(), causal.stride(0)
            y, mask_out, info, _ = hip.bert_layer(w, hidden, mask2d, token_attn, temperature if prune else 0, cross,
                                                  flat(enc0) if cross else None, flat(enc1) if twin else None,
                                                  enc0.shape[1] if cross else 0, em[0] if twin else None, em[1] if twin else None)
        layer.last_prune = info
        ctx.layer, ctx.temperature, ctx.cross, ctx.twin = layer, float(temperature), cross, twin
        ctx.k = int(info["indices"].shape[1]) if (info is not None and info.get("pruned")) else 0
        ctx.has_ta = token_attn is not None
        ctx.has_mask = mask2d is not None
        ctx.enc_masks = em if twin else None
        e = hidden.new_empty(0)
        ctx.save_for_backward(hidden, token_attn if token_attn is not None else e, mask2d if mask2d is not None else e,
                              enc0 if cross else e, enc1 if twin else e)
        if mask_out is None:
            mask_out = mask2d if mask2d is not None else hidden.new_zeros((hidden.shape[0], y.shape[1]))
        ctx.mark_non_differentiable(mask_out)
        return y, mask_out

    @staticmethod
    def backward(ctx, dy, _dmask):
        hidden, ta, mask2d, enc0, enc1 = ctx.saved_tensors
        ta = ta if ctx.has_ta else None
        _check_versions(ctx, _med_params_of(ctx.layer, ctx.cross), "BertLayer backward")
        _second_backward_guard(ctx, "BertLayer backward")
        enc = ([enc0, enc1] if ctx.twin else enc0) if ctx.cross else None
        with torch.no_grad(), _in_mode(ctx.mode):
            dh, dta, denc, grads = med_layer_backward(ctx.layer, hidden, mask2d if ctx.has_mask else None, ta, ctx.temperature,
                                                      ctx.k, dy, enc, ctx.enc_masks, ctx.causal, saved=ctx.saved)
            ctx.saved = None
        if ctx.has_ta and dta is None:
            dta = torch.zeros_like(ta)
        de0, de1 = (denc if ctx.twin else (denc, None)) if ctx.cross else (None, None)
        return (None, None, None, None, None, dh, dta if ctx.has_ta else None, de0, de1) + tuple(
            grads.get(n) for n in _layer_param_names(ctx.layer, ctx.cross))


def med_layer_forward_with_grad(layer, hidden, mask2d, temperature, token_attn, enc=None, enc_masks=None, causal=None):
    """BertLayer (MED or NLVR) under autograd -> (output, new additive mask [B,L'] or None); enc: the encoder tokens of mode
    'multimodal' (MED: a tensor; NLVR: a list of two, with enc_masks their additive key masks [B,Nk] or None) or None for mode
    'text'; fp32 mode only."""
    _check_mode("the BERT layer backward")
    _warn_implicit(hidden.requires_grad or (token_attn is not None and token_attn.requires_grad))
    if token_attn is not None and not token_attn.is_contiguous():
        token_attn = token_attn.contiguous()
    twin = isinstance(enc, (list, tuple))
    e0, e1 = (enc[0], enc[1]) if twin else (enc, None)
    y, mask_out = MedLayerFunction.apply(layer, temperature, mask2d, enc_masks if twin else None, causal, hidden, token_attn, e0, e1,
                                         *_med_params_of(layer, enc is not None))
    return y, (mask_out if mask2d is not None else None)


# ---------------------------------------------------------------------------------------------------------------------------
# The whole pruned ViT under autograd (fp32 mode): patch embedding + CLS / position (vit.py:283-289), the query model's logits
# and att_ft (models/utils.py:165-178), the twelve blocks (VitBlockFunction) and the final LayerNorm (vit.py:309).  Both outputs
# of VisionTransformer.forward carry a graph: the image tokens and sd_img_ft_all (the sum of the layers' att_ft, consumed by the
# training drivers' alignment loss).

class PatchTokensFunction(torch.autograd.Function):
    """x = cat(cls, conv(img)) + pos_embed[:, :N] (vit.py:283-289; conv = im2col + GEMM, madtp_amd.vit.PatchEmbed).
    Gradients for the projection weight / bias, cls_token and pos_embed; none for the image."""

    @staticmethod
    def forward(ctx, vit, img, w, b, cls, pos):
        ctx.mode = _mode()
        patches, np_ = vit.patch_embed.run(img)
        B = img.shape[0]
        x = hip.assemble_tokens(patches, cls, pos, B, np_)
        ctx.vit, ctx.np_, ctx.patch = vit, np_, vit.patch_embed.patch_size[0]
        ctx.has_b = b is not None
        ctx.save_for_backward(img, w, pos)
        return x

    @staticmethod
    def backward(ctx, dx):
        img, w, pos = ctx.saved_tensors
        B, N, D = dx.shape
        with torch.no_grad(), _in_mode(ctx.mode):
            dx = dx.contiguous().float()
            dtok = colsum(dx.view(B, N * D)).view(N, D)          # sum over the batch: d pos_embed[:, :N] (row 0 = d cls_token too)
            dpos = torch.zeros_like(pos)
            dpos[0, :N] = dtok
            dcls = dtok[0].reshape(1, 1, D).clone()
            dpatch = dx[:, 1:, :].reshape(B * ctx.np_, D).contiguous()
            cols = hip.patchify(img.contiguous().float(), ctx.patch, torch.float32)
            dw, db = wgrad(dpatch, cols, bias=True) if ctx.has_b else (wgrad(dpatch, cols), None)
            dw = dw.view_as(w)
        return None, None, dw, db, dcls, dpos


def att_ft_bwd(inner, q, dA, sd_dim, dinner, dq):
    """madtp_att_ft_bwd: adds the att_ft branch's gradient to dinner [B,n,K] and dq [B,n,D] (both dense f32, in place)."""
    B, n, K = inner.shape
    D = q.shape[-1]
    ws = torch.empty((2 * B * K * n,), device=inner.device, dtype=torch.float32)
    _check(load().madtp_att_ft_bwd(_p(inner), _p(q), _p(dA), 1.0 / math.sqrt(sd_dim), _p(dinner), _p(dq), _p(ws), B, n, K, D, _stream()),
           "madtp_att_ft_bwd")


class QueryModelFunction(torch.autograd.Function):
    """Query_model.forward(return_token_att=True) (models/utils.py:147-183): x [B,N,D] (row 0 = CLS, not used), sd [K,Dsd] ->
    (token_att [B,N-1,K] raw logits, att_ft [B,K,Dsd]); with a q_map (CLIP, map_func=True: q = Linear(ft) before the logits,
    :160-163) its weight and bias follow as inputs.  Forward = the inference path's own kernels (madtp_amd.utils.Query_model);
    backward: the logits' gradient (from token_attn's use in the blocks and from att_ft's softmax over tokens) goes through dgrad /
    wgrad on the exact-f32 GEMM, plus the direct att_ft term W^T dA for the (mapped) tokens."""

    @staticmethod
    def forward(ctx, qm, x, sd, *qmap):
        ctx.mode = _mode()
        ta, att_ft, _ = qm(x[:, 1:, :], sd, return_token_att=True)
        ta = ta.contiguous()
        ctx.save_for_backward(x, sd, ta, *qmap)
        ctx.sd_dim, ctx.has_map = qm.att_dim, len(qmap) > 0
        return ta, (att_ft.clone() if att_ft is not None else x.new_zeros((x.shape[0], sd.shape[0], sd.shape[1])))

    @staticmethod
    def backward(ctx, dta, datt):
        x, sd, ta = ctx.saved_tensors[:3]
        B, N, D = x.shape
        n, K = N - 1, sd.shape[0]
        with torch.no_grad(), _in_mode(ctx.mode):
            ft = x[:, 1:, :].contiguous()
            if ctx.has_map:
                wm, bm = ctx.saved_tensors[3], ctx.saved_tensors[4]
                wmp, bmp = _f32_wb(wm, bm)
                q = _gemm(ft.view(B * n, D), wmp, bmp, n=wm.shape[0], out_dtype=torch.float32).view(B, n, -1)
            else:
                q = ft
            dinner = dta.contiguous().float().clone() if dta is not None else torch.zeros_like(ta)
            dq = torch.zeros_like(q)
            if datt is not None:
                att_ft_bwd(ta, q, datt.contiguous().float(), ctx.sd_dim, dinner, dq)
            d2 = dinner.view(B * n, K)
            dq2 = dgrad(d2, sd.detach(), residual=dq.view(B * n, -1))
            dsd = wgrad(d2, q.reshape(B * n, -1)) if ctx.needs_input_grad[2] else None
            gmap = ()
            if ctx.has_map:
                gmap = wgrad(dq2, ft.view(B * n, D), bias=True)
                dq2 = dgrad(dq2, wm.detach())
            dx = None
            if ctx.needs_input_grad[1]:
                dx = torch.zeros_like(x)
                dx[:, 1:, :] = dq2.view(B, n, D)
        return (None, dx, dsd) + gmap


class LayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.mode = _mode()
        ctx.eps = float(eps)
        ctx.save_for_backward(x, gamma)
        y, _ = hip.layernorm(x.contiguous(), gamma.detach(), beta.detach(), eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        D = x.shape[-1]
        with torch.no_grad(), _in_mode(ctx.mode):
            dx, dg, db = layernorm_bwd(x.reshape(-1, D).contiguous(), gamma.detach(), dy.reshape(-1, D).contiguous().float(), ctx.eps)
        return dx.view_as(x), dg, db, None


def vit_forward_with_grad(vit, img, space_dict, temperature, register_blk=-1):
    """VisionTransformer.forward (vit.py:281-310) under autograd -> (x, sd_img_ft_all); called by madtp_amd.vit when gradients
    are required in the fp32 mode."""
    _check_mode("the ViT backward")
    if img.requires_grad:
        raise NotImplementedError("gradients with respect to the image are not built (the drivers never ask for them)")
    pe = vit.patch_embed.proj
    x = PatchTokensFunction.apply(vit, img, pe.weight, pe.bias, vit.cls_token, vit.pos_embed)
    token_num = x.shape[-2]
    reduce_num = int((token_num - 1) // vit.depth)
    sd_all = None
    for i, blk in enumerate(vit.blocks):
        if space_dict is not None:
            token_attn, sd_ft = QueryModelFunction.apply(vit.img_query_model, x, space_dict)  # :297-298
            sd_all = sd_ft if sd_all is None else sd_all + sd_ft                              # :300-303
            x = blk(x, register_blk == i, reduce_num, temperature, token_attn)  # :304 (Block.forward routes to VitBlockFunction)
        else:
            x = blk(x, register_blk == i)
    return LayerNormFunction.apply(x, vit.norm.weight, vit.norm.bias, vit.norm.eps), sd_all


# ---------------------------------------------------------------------------------------------------------------------------
# The text side and the task head under autograd (fp32 mode): embeddings, the encoder's layer loop (query model + layer
# Functions), a Linear with activation - what BLIP_NLVR.forward needs besides the ViT for loss.backward() on its logits.

class EmbeddingsFunction(torch.autograd.Function):
    """BertEmbeddings.forward (med.py:63-86, absolute positions): LayerNorm(word[ids] + pos[:L]).  Gradients for the two tables and
    the LayerNorm; the word table's is a scatter-add over the ids (index_add_)."""

    @staticmethod
    def forward(ctx, ids, word, pos, gamma, beta, eps):
        ctx.mode = _mode()
        y, _ = hip.bert_embed(ids.contiguous(), word, pos, gamma, beta, eps)
        ctx.eps = float(eps)
        ctx.save_for_backward(ids, word, pos, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, word, pos, gamma = ctx.saved_tensors
        B, L = ids.shape
        D = word.shape[1]
        with torch.no_grad(), _in_mode(ctx.mode):
            e = (word.detach()[ids] + pos.detach()[:L]).reshape(B * L, D).contiguous()   # the LayerNorm's input, recomputed
            de, dg, db = layernorm_bwd(e, gamma.detach(), dy.reshape(B * L, D).contiguous().float(), ctx.eps)
            dpos = torch.zeros_like(pos)
            dpos[:L] = colsum(de.view(B, L * D)).view(L, D)
            dword = torch.zeros_like(word).index_add_(0, ids.reshape(-1), de)
        return None, dword, dpos, dg, db, None


class DropoutFunction(torch.autograd.Function):
    """y = x o mask / (1 - p) with the counter-based mask of (seed, site) (madtp_dropout); the backward applies the same mask."""

    @staticmethod
    def forward(ctx, x, p, seed, site):
        ctx.args = (float(p), int(seed), int(site))
        return dropout(x.contiguous().float(), p, seed, site)

    @staticmethod
    def backward(ctx, dy):
        p, seed, site = ctx.args
        return dropout(dy.contiguous().float(), p, seed, site), None, None, None


def module_dropout(module, p, x):
    """nn.Dropout(p) of a mirror module on the HIP path: identity in eval mode / p = 0, else a fresh site of the process-wide counter"""
    if not module.training or p <= 0.0:
        return x
    from .runtime import next_dropout_base
    seed, base = next_dropout_base()
    return DropoutFunction.apply(x, p, seed, _site(base, 0))


class LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b) on the exact-f32 GEMM (x [M,K], W [N,K])."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        ctx.mode = _mode()
        N = w.shape[0]
        wp, bb = _f32_wb(w, b)
        x = x.contiguous().float()
        u = _gemm(x, wp, bb, n=N, out_dtype=torch.float32)
        ctx.act, ctx.has_b = act, b is not None
        ctx.save_for_backward(x, w, u)
        return act_fwd(u, act) if act != hip.ACT_NONE else u

    @staticmethod
    def backward(ctx, dy):
        x, w, u = ctx.saved_tensors
        with torch.no_grad(), _in_mode(ctx.mode):
            dy = dy.contiguous().float()
            du = act_bwd(u, dy, ctx.act) if ctx.act != hip.ACT_NONE else dy
            dx = dgrad(du, w.detach()) if ctx.needs_input_grad[0] else None
            dw = wgrad(du, x) if ctx.needs_input_grad[1] else None
            db = colsum(du) if ctx.has_b else None
        return dx, dw, db, None


def bert_encoder_forward_with_grad(enc, hidden_states, attention_mask, space_dict, temperature, encoder_hidden_states,
                                   encoder_attention_mask, mode, always_query):
    """The layer loop of BertEncoder.forward (med.py:509-571 / nlvr_encoder.py:600-660) under autograd: the text query model as a
    QueryModelFunction, the layers route themselves to MedLayerFunction.  -> (hidden_states, attention_mask, sd_txt_ft_all)."""
    sd_all = None
    reduce_num = int((hidden_states.shape[-2] - 1) // enc.config.num_hidden_layers)
    for layer_module in enc.layer:
        layer_module.__dict__.pop("_kv_pre", None)
        token_attn = None
        if space_dict is not None or always_query:
            if space_dict is None:
                raise TypeError("nlvr_encoder.BertEncoder calls txt_query_model unconditionally (:608): space_dict must be given")
            token_attn, sd_ft = QueryModelFunction.apply(enc.txt_query_model, hidden_states, space_dict)
            sd_all = sd_ft if sd_all is None else sd_all + sd_ft
        t = temperature if space_dict is not None else 0
        if enc.layer_cls.variant == "nlvr":
            outs = layer_module(hidden_states, attention_mask, space_dict, None, encoder_hidden_states, encoder_attention_mask, None,
                                False, mode=mode, token_attn=token_attn, reduce_num=reduce_num, temperature=t)
        else:
            outs = layer_module(hidden_states, attention_mask, None, encoder_hidden_states, encoder_attention_mask, None, False,
                                mode=mode, space_dict=space_dict, token_attn=token_attn, reduce_num=reduce_num, temperature=t)
        hidden_states, attention_mask = outs[0], outs[-1]
    return hidden_states, attention_mask, sd_all


# ---------------------------------------------------------------------------------------------------------------------------
# CLIP's vision tower (clip/model.py:275-313) under autograd: conv1 (no bias) + class / positional embedding, ln_pre, the blocks
# (they route themselves to VitBlockFunction / QueryModelFunction), ln_post on the class token, x @ proj.

class ClipPatchTokensFunction(torch.autograd.Function):
    """tok = cat(class_embedding, conv1(img)) + positional_embedding (clip/model.py:293-297; conv1 = im2col + GEMM)."""

    @staticmethod
    def forward(ctx, img, w, cls, pos, patch):
        ctx.mode = _mode()
        cols = hip.patchify(img, patch, torch.float32)
        wp, _ = _f32_wb(w.reshape(w.shape[0], -1), None)
        patches = _gemm(cols, wp, None, out_dtype=torch.float32, n=w.shape[0])
        B = img.shape[0]
        ctx.np_, ctx.patch = patches.shape[0] // B, patch
        ctx.save_for_backward(img, w)
        return hip.assemble_tokens(patches, cls.detach().contiguous(), pos.detach().contiguous(), B, ctx.np_)

    @staticmethod
    def backward(ctx, dx):
        img, w = ctx.saved_tensors
        B, N, D = dx.shape
        with torch.no_grad(), _in_mode(ctx.mode):
            dx = dx.contiguous().float()
            dpos = colsum(dx.view(B, N * D)).view(N, D)
            dcls = dpos[0].clone()
            dpatch = dx[:, 1:, :].reshape(B * ctx.np_, D).contiguous()
            cols = hip.patchify(img, ctx.patch, torch.float32)
            dw = wgrad(dpatch, cols).view_as(w)
        return None, dw, dcls, dpos, None


def clip_vision_forward_with_grad(vt, img, space_dict, temperature, max_keep):
    """clip.model.VisionTransformer.forward under autograd -> (features [B, output_dim], sd_img_ft_all); fp32 mode."""
    _check_mode("the CLIP backward")
    tok = ClipPatchTokensFunction.apply(img, vt.conv1.weight, vt.class_embedding, vt.positional_embedding, vt.patch_size)
    tok = LayerNormFunction.apply(tok, vt.ln_pre.weight, vt.ln_pre.bias, vt.ln_pre.eps)
    xs = tok.permute(1, 0, 2)
    sd_all = None
    if space_dict is not None:
        xs, _, _, sd_all, _ = vt.transformer(xs, space_dict, temperature, None, max_keep)
    else:
        xs = vt.transformer(xs)[0]
    cls = xs.permute(1, 0, 2)[:, 0, :].contiguous()
    cls = LayerNormFunction.apply(cls, vt.ln_post.weight, vt.ln_post.bias, vt.ln_post.eps)
    if vt.proj is not None:
        cls = LinearFunction.apply(cls, vt.proj.t().contiguous(), None, hip.ACT_NONE)  # x @ proj (:311-312)
    return cls, sd_all
