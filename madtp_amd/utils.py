"""Mirror of the reference's models/utils.py hot-path helpers (vector_gather :13-33, Query_model :109-183) with
the same names, arguments and return values, executed by the HIP kernels."""
import math

import torch
from torch import nn

from . import hip
from .runtime import PreparedCache, compute_dtype, prepare_linear, require_gpu, to_compute


def vector_gather(vectors, indices):
    """models/utils.py:13-33 - Tensor[N,L,D], indices Tensor[N,K] or [N] -> Tensor[N,K,D] or [N,D]."""
    require_gpu(vectors, "vectors")
    N, L, D = vectors.shape
    squeeze = False
    if indices.ndim == 1:
        squeeze = True
        indices = indices.unsqueeze(-1)
    N2, K = indices.shape
    assert N == N2
    out = hip.vector_gather(vectors.float().contiguous(), indices.to(torch.int64).contiguous())
    return out.squeeze(1) if squeeze else out


def full_rows_of(ft):
    """If ft is x[:,1:,:] of a contiguous [B,N,D] f32 tensor, return x as a [B*N, D] alias (no copy), else None.
    Lets the alignment GEMM run over the token buffer in place instead of copying the patch slice."""
    if ft.dim() != 3 or ft.dtype != torch.float32:
        return None
    B, n, D = ft.shape
    if ft.stride(2) != 1 or ft.stride(1) != D or (B > 1 and ft.stride(0) != (n + 1) * D) or ft.storage_offset() < D:
        return None  # (with one sample the batch stride is meaningless: a permuted (N,1,C) tensor reports stride(0) = C)
    return torch.as_strided(ft, (B * (n + 1), D), (D, 1), ft.storage_offset() - D)


class DeferredAttFt:
    """Fast-mode bookkeeping for an encoder's `sd_ft_all += sd_ft` (vit.py:297-303, nlvr_encoder.py:608-613): the layers
    only record their (logits, token rows) pair - both stay alive as ordinary tensors - and finish() sums all the
    layers' att_ft in ONE kernel (madtp_query_att_ft_multi) instead of a 39 MB read-modify-write per layer."""

    def __init__(self, sd_dim, exact=False):
        self.pairs, self.sd_dim, self.exact = [], sd_dim, exact

    def add(self, token_att, ft):
        self.pairs.append((token_att, ft))

    def finish(self, pending=None):
        """pending (optional list): run the sum on the auxiliary stream and append a handle whose .sync() makes the
        current stream wait for it; the returned tensor must not be consumed (or freed) before that."""
        if not self.pairs:
            return None
        if pending is None:
            out = hip.query_att_ft_multi(self.pairs, sd_dim=self.sd_dim, exact=self.exact)
            self.pairs = []
            return out
        from .runtime import side_stream
        main, side = torch.cuda.current_stream(), side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = hip.query_att_ft_multi(self.pairs, sd_dim=self.sd_dim, exact=self.exact)
        out.record_stream(main)
        pending.append(_SideWork(side, self.pairs))  # the handle keeps the layers' tensors alive until the wait
        self.pairs = []
        return out


class _SideWork:
    def __init__(self, stream, keep):
        self.stream, self.keep = stream, keep

    def sync(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        self.keep = None


class Query_model(nn.Module):
    """models/utils.py:109-183.  forward(ft, sd, mask=None, return_token_att=False, temperature=1) ->
    (token_att[B,n,K] raw logits, att_ft[B,K,sd_dim], sd).  The logits x.sd^T always run on the exact-f32 MFMA
    (they feed the pruning threshold); att_ft uses the fused softmax-over-tokens kernel."""

    def __init__(self, ft_dim, sd_dim, temperature=1, att_func_type='softmax', pool_type='sum', map_func=False):
        super().__init__()
        assert att_func_type in ['softmax', 'sigmoid', 'sparsemax']
        assert pool_type in ['mean', 'max', 'sum']
        self.att_func_type = att_func_type
        self.pool_type = pool_type
        self.att_dim = sd_dim
        self.temperature = temperature
        self.map_func = map_func
        if self.map_func:
            self.q_map = nn.Sequential(nn.Linear(ft_dim, sd_dim))
        self._cache = PreparedCache()
        self.compute_att_ft = True  # att_ft only feeds the training loss (blip_nlvr.py:86-96); eval callers may clear

    def deferred(self):
        """A DeferredAttFt for an encoder loop, or None when att_ft is not computed.  Fast mode: one bf16-MFMA launch over all
        layers; parity modes: one launch of the exact-f32 kernel that keeps the reference's per-layer summation order
        (bit-identical to accumulating layer by layer).  With a q_map (CLIP: every block has its own query model) the layers'
        mapped q tensors are kept alive by the DeferredAttFt until that launch."""
        if self.compute_att_ft:
            cdt = compute_dtype()  # fp32: the exact-f32 kernel in layer order; f16x3: f16-split operands; bf16: the fast kernel
            return DeferredAttFt(self.att_dim, exact=True if cdt == torch.float32 else ("split" if cdt == torch.float16 else False))
        return None

    MAX_ENTRIES = 128  # the pruning kernels keep one dictionary row per lane pair (token_score: K <= 128; att_ft: K <= 112)

    def _dictionary(self, sd):
        """prepared dictionary operands: (f32 [128, dim] Lin, split planes or None)"""
        if sd.shape[0] > self.MAX_ENTRIES:
            raise NotImplementedError(
                f"space_dict with {sd.shape[0]} entries: the HIP pruning kernels hold at most {self.MAX_ENTRIES} dictionary "
                "entries (112 when att_ft is computed); every reference config uses sd_num = 100 (configs/*.yaml)")
        sdl = self._cache.get(("sd", id(sd)), [sd], lambda: prepare_linear([sd], None, torch.float32))
        split = None
        if compute_dtype() == torch.bfloat16 and sdl.w.shape[0] == 128:
            def _split():
                hi = hip.cast_bf16_plain(sdl.w)  # (bf16 planes in both fast modes: the logits kernel splits x into bf16 too)
                lo = hip.cast_bf16_plain((sdl.w - hi.float()).contiguous())
                return hi, lo
            split = self._cache.get(("sd_split", id(sd)), [sd], _split)
        elif compute_dtype() == torch.float16 and sdl.w.shape[0] == 128:
            def _split16():
                q = hip.split_f16_weight(sdl.w)
                d = sdl.w.shape[1]
                return q[:, :d].contiguous(), q[:, d:].contiguous(), q._madtp_w_scale
            split = self._cache.get(("sd_split16", id(sd)), [sd], _split16)
        return sdl, split

    def encoder_args(self, sd, B, dim, device):
        """Operands of the query model for an encoder-level call (hip.vit_encoder / hip.bert_encoder): -> (qargs, deferred)
        where deferred=True means att_ft is NOT accumulated by the call (fast mode: one launch over all layers afterwards).
        None when this query model cannot run inside the encoder call (q_map)."""
        if self.map_func:
            return None, False
        sdl, split = self._dictionary(sd)
        K = sd.shape[0]
        if sdl.w.shape[0] != 128:
            # the encoder-level calls carve their per-layer logits slabs (and the deferred att_ft segments) at a row stride of
            # 128 floats; a dictionary of more than 128 entries (padded to 256+) stays on the per-layer path, which sizes its
            # logits as [rows, roundup(K, 128)]
            return None, False
        qa = {"sd_w": sdl.w, "K": K, "sd_dim": self.att_dim, "att_ft": None, "stats_ws": None}
        if split is not None:
            qa.update(sd_hi=split[0], sd_lo=split[1], split_dtype=hip.split_code(split[0]),
                      sd_scale=split[2] if len(split) > 2 else 1.0)
        # att_ft of all layers in ONE launch after the call ("bf16": fast-mode kernel, "exact": parity arithmetic and order)
        deferred = False
        if self.compute_att_ft:
            deferred = "bf16" if (split is not None and split[0].dtype == torch.bfloat16) else \
                ("split" if compute_dtype() == torch.float16 else "exact")
        return qa, deferred

    def forward(self, ft, sd, mask=None, return_token_att=False, temperature=1, acc_ft=None, defer=None):
        """acc_ft (extension): running sum tensor to accumulate att_ft into (the encoders' `sd_ft_all += sd_ft`).
        defer (extension): a DeferredAttFt - att_ft is not computed here (returned as None) but summed by defer.finish()."""
        require_gpu(ft, "ft")
        if not return_token_att:
            raise NotImplementedError("Query_model(return_token_att=False) returns the normalised attention weights; "
                                      "no reference call site on the pruned forward path uses it")
        B, n, D = ft.shape
        K = sd.shape[0]
        sdl, split = self._dictionary(sd)  # fast mode: bf16 hi/lo planes; f16x3 mode: f16 planes Q0 / Q1 of sd * 2^s
        if self.map_func:
            # CLIP: q = q_map(ft) (clip/model.py:188, models/utils.py:160-163).  Mapped over ALL rows of the token buffer
            # when ft is x[:,1:,:] of a contiguous tensor (the CLS row is computed and ignored), then the same
            # logits / att_ft kernels run on q.
            cdt = compute_dtype()
            qm = self._cache.get(("qmap", cdt), [self.q_map[0].weight, self.q_map[0].bias],
                                 lambda: prepare_linear([self.q_map[0].weight], [self.q_map[0].bias], cdt))
            rows = full_rows_of(ft)
            off = 1
            if rows is None:
                rows, off = ft.float().contiguous().view(B * n, D), 0
            q = hip.gemm(to_compute(rows.contiguous()), qm.w, qm.b, out_dtype=torch.float32, n=qm.n).view(B, n + off, -1)
            if off == 1:
                want = self.compute_att_ft and defer is None
                token_att, att_ft = hip.query_model(q, sdl.w, K, att_ft=acc_ft, want_att_ft=want,
                                                    sd_dim=self.att_dim, sd_split=split)
                if self.compute_att_ft and not want:  # the mapped q stays alive in the DeferredAttFt until its one launch
                    defer.add(token_att, q[:, 1:, :])
                    return token_att, None, sd
                return token_att, (att_ft if self.compute_att_ft else acc_ft), sd
            rows, ftq = q.view(B * n, -1), q
        else:
            rows = full_rows_of(ft)
            off = 1
            ftq = ft
            if rows is not None:
                # fast path: ft is x[:,1:,:] of a contiguous token buffer -> one C call (logits GEMM + att_ft)
                want = self.compute_att_ft and not (defer is not None and (split is not None or defer.exact))
                token_att, att_ft = hip.query_model(rows.view(B, n + 1, D), sdl.w, K, att_ft=acc_ft, want_att_ft=want,
                                                    sd_dim=self.att_dim, sd_split=split)
                if self.compute_att_ft and not want:
                    defer.add(token_att, ft)
                    return token_att, None, sd
                return token_att, (att_ft if self.compute_att_ft else acc_ft), sd
            ftq = ft.float().contiguous()
            rows, off = ftq.view(B * n, D), 0
        kp = sdl.w.shape[0]
        full = hip.gemm(rows, sdl.w, n=kp)  # [rows, 128], zero weight rows beyond K
        token_att = full.view(B, n + off, kp)[:, off:, :K]
        att_ft = acc_ft
        if self.compute_att_ft and defer is not None:
            # (callers with a DeferredAttFt discard the returned att_ft: this layer's share must go through it as well)
            defer.add(token_att, ftq[:, off:, :] if off else ftq)
            return token_att, None, sd
        if self.compute_att_ft:
            att_ft = hip.query_att_ft(token_att, ftq, out=acc_ft, sd_dim=self.att_dim,
                                      fast=split is not None and split[0].dtype == torch.bfloat16)
        return token_att, att_ft, sd
