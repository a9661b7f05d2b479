"""Mirror of the reference's BERT text encoders with token pruning:
     variant 'med'  - models/med.py           (BertLayer :332-467, BertEncoder :470-598, BertModel :686-929)
     variant 'nlvr' - models/nlvr_encoder.py  (twin cross-attention; BertLayer :385-559, BertEncoder :562-687)
Same sub-module / parameter names (checkpoint keys) and forward() signatures; the arithmetic is enqueued on the
hand-written gfx950 kernels.  madtp_amd/med.py and madtp_amd/nlvr_encoder.py export the two variants under the
reference's class names.
"""
import json
import math
import os

import torch
import torch.nn as nn

from . import hip
from .runtime import autograd_precision as _autograd_precision, EncoderWeights, PreparedCache, param_epoch, own_modules, get_precision, encoder_call_preference, f32_ptr, attn_dtype, compute_dtype, dtype_code, lin_of, require_gpu, as_f32_contig, to_compute
from .utils import Query_model


class BertConfig:
    """Minimal stand-in for transformers.BertConfig (the reference loads configs/med_config.json with it)."""
    _defaults = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                     intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                     initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0, position_embedding_type="absolute",
                     add_cross_attention=False, encoder_width=768, chunk_size_feed_forward=0, evaluate=True,
                     output_attentions=False, output_hidden_states=False, use_return_dict=True, is_decoder=False)

    def __init__(self, **kw):
        for k, v in {**self._defaults, **kw}.items():
            setattr(self, k, v)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))

    @classmethod
    def med_default(cls):
        """configs/med_config.json of the reference."""
        return cls(vocab_size=30524, encoder_width=768, add_cross_attention=True)


_ENC_STORE = []
# MADTP_KV_AHEAD=1: project the encoder tokens to every layer's cross-attention [k|v] in one GEMM per branch before the layer
# loop instead of inside each layer (the reference's data flow, default).  Measured neutral on the NLVR2 headline (2 x 183 us
# instead of 24 x 17.8 us of GEMM, but the strided K/V rows cost the 24 cross-attention kernels 6.7 -> 8.1 us each and the two
# chip-filling GEMMs no longer leave room for the vision encoder's deferred att_ft kernel on the auxiliary stream).
_KV_AHEAD = os.environ.get("MADTP_KV_AHEAD", "0") == "1"
# MADTP_ENCODER_CALL (see madtp_amd/vit.py): "auto" = madtp_bert_encoder for small batches only (up to ENCODER_CALL_MAX_SAMPLES
# samples: the text side has ~20-35 rows per sample, so the criterion is the batch), "1" always, "0" one call per BertLayer
_ENCODER_CALL = {"0": False, "1": True}.get(os.environ.get("MADTP_ENCODER_CALL", "auto"), "auto")
ENCODER_CALL_MAX_SAMPLES = 32


def _use_encoder_call(B, flag):
    if flag == "auto" and encoder_call_preference() is not None:
        return bool(encoder_call_preference())
    return (B <= ENCODER_CALL_MAX_SAMPLES) if flag == "auto" else bool(flag)


def _cast(x2d):
    return to_compute(x2d)


class BertEmbeddings(nn.Module):
    """med.py:43-86 / nlvr_encoder.py:43-86 (absolute positions; no token-type embeddings in BLIP's MED)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.position_embedding_type = getattr(config, "position_embedding_type", "absolute")
        self.config = config

    def forward(self, input_ids=None, position_ids=None, inputs_embeds=None, past_key_values_length=0):
        if input_ids is None or position_ids is not None:
            raise NotImplementedError("only input_ids with the default (consecutive) positions are implemented")
        require_gpu(input_ids, "input_ids")
        if past_key_values_length:
            # med.py:66-67: position_ids[:, past : past + L] - the same kernel on the position table from row `past` on
            p0 = int(past_key_values_length)
            if p0 + input_ids.shape[1] > self.position_embeddings.weight.shape[0]:
                raise ValueError("past_key_values_length + sequence length exceeds max_position_embeddings")
            cdt = compute_dtype()
            y32, ylp = hip.bert_embed(input_ids.contiguous(), self.word_embeddings.weight, self.position_embeddings.weight[p0:],
                                      self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps,
                                      lp=None if cdt == torch.float32 else cdt)
            return y32
        if torch.is_grad_enabled() and _autograd_precision() and any(p.requires_grad for p in self.parameters()):
            from .backward import EmbeddingsFunction, module_dropout  # training use: gradients for the two tables and the LayerNorm
            y = EmbeddingsFunction.apply(input_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                                         self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
            return module_dropout(self, float(self.dropout.p), y)  # med.py:85 (identity in eval mode)
        cdt = compute_dtype()
        y32, ylp = hip.bert_embed(input_ids.contiguous(), self.word_embeddings.weight, self.position_embeddings.weight,
                                  self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps,
                                  lp=None if cdt == torch.float32 else cdt)
        if ylp is not None:  # the first layer takes the compute-dtype copy from the embedding LayerNorm (no cast launch)
            y32._madtp_lp = (ylp, y32._version)
        return y32


class _LinHolder:
    """weight / bias of a derived (not registered) Linear for runtime.lin_of"""
    __slots__ = ("weight", "bias")

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


class BertSelfAttention(nn.Module):
    """med.py:89-236 / nlvr_encoder.py:88-237."""

    def __init__(self, config, is_cross_attention):
        super().__init__()
        self.config = config
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        if self.attention_head_size != 64:
            raise ValueError("the gfx950 attention kernels are built for head_dim 64")
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        kv_in = config.encoder_width if is_cross_attention else config.hidden_size
        self.key = nn.Linear(kv_in, self.all_head_size)
        self.value = nn.Linear(kv_in, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.position_embedding_type = getattr(config, "position_embedding_type", "absolute")
        self.save_attention = False
        self.attention_map = None
        self.cls_attn = None
        self.score_side = None
        self.is_cross_attention = is_cross_attention
        self._cache = PreparedCache()

    def save_attn_gradients(self, g): self.attn_gradients = g
    def get_attn_gradients(self): return self.attn_gradients
    def save_attention_map(self, m): self.attention_map = m
    def get_attention_map(self): return self.attention_map
    def save_cls_attn(self, c): self.cls_attn = c
    def get_cls_attn(self): return self.cls_attn


class BertSelfOutput(nn.Module):
    """med.py:239-250; nlvr_encoder.py:240-271 (twin / merge)."""

    def __init__(self, config, twin=False, merge=False):
        super().__init__()
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if twin:
            self.dense0 = nn.Linear(config.hidden_size, config.hidden_size)
            self.dense1 = nn.Linear(config.hidden_size, config.hidden_size)
        else:
            self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        if merge:
            self.merge_layer = nn.Linear(config.hidden_size * 2, config.hidden_size)
            self.merge = True
        else:
            self.merge = False
        self.twin = twin
        self._cache = PreparedCache()

def _fused_twin_output(co, fold_merge):
    """[W0 | W1] along K for the twin cross-attention output (nlvr_encoder.py:259-266).  Average layers: bias b0+b1,
    the 0.5 goes into the GEMM epilogue scale.  Merge layers (fast mode only): merge_layer folded in,
    Wm[:, :D] W0 | Wm[:, D:] W1 and bias Wm[:, :D] b0 + Wm[:, D:] b1 + bm, computed once in f32 on the f32 GEMM."""
    from .runtime import prepare_linear
    W0, W1 = co.dense0.weight.detach().float(), co.dense1.weight.detach().float()
    b0, b1 = co.dense0.bias.detach().float(), co.dense1.bias.detach().float()
    if fold_merge:
        Wm, bm = co.merge_layer.weight.detach().float(), co.merge_layer.bias.detach().float()
        D = W0.shape[0]

        def mm(a, b_t):  # a [m,k] @ b_t[n,k]^T on the exact-f32 kernel
            bt = b_t.contiguous()
            pad = (-bt.shape[0]) % 128
            if pad:
                bt = torch.cat([bt, torch.zeros(pad, bt.shape[1], device=bt.device)], 0)
            return hip.gemm(a.contiguous(), bt.contiguous(), n=b_t.shape[0])

        Wl, Wr = Wm[:, :D].contiguous(), Wm[:, D:].contiguous()
        W0f = mm(Wl, W0.t().contiguous())  # Wl @ W0
        W1f = mm(Wr, W1.t().contiguous())
        bias = mm(b0[None, :], Wl)[0] + mm(b1[None, :], Wr)[0] + bm
        Wcat = torch.cat([W0f, W1f], dim=1)
    else:
        Wcat = torch.cat([W0, W1], dim=1)
        bias = b0 + b1
    return prepare_linear([Wcat], [bias], compute_dtype())


class BertAttention(nn.Module):
    """med.py:253-299 / nlvr_encoder.py:274-349."""

    def __init__(self, config, is_cross_attention=False, layer_num=-1, twin_cross=False):
        super().__init__()
        self.twin = is_cross_attention and twin_cross
        if self.twin:
            self.self0 = BertSelfAttention(config, is_cross_attention)
            self.self1 = BertSelfAttention(config, is_cross_attention)
            self.output = BertSelfOutput(config, twin=True, merge=layer_num >= 6)
        else:
            self.self = BertSelfAttention(config, is_cross_attention)
            self.output = BertSelfOutput(config)
        self.pruned_heads = set()
        self._cache = PreparedCache()


class BertIntermediate(nn.Module):
    """med.py:302-315."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act != "gelu":
            raise ValueError("only the erf-GELU of med_config.json is implemented")


class BertOutput(nn.Module):
    """med.py:318-329."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class _BertLayerBase(nn.Module):
    variant = "med"

    def __init__(self, config, layer_num):
        super().__init__()
        self.config = config
        self.chunk_size_feed_forward = config.chunk_size_feed_forward
        self.seq_len_dim = 1
        self.attention = BertAttention(config)
        self.layer_num = layer_num
        if self.config.add_cross_attention:
            self.crossattention = BertAttention(config, is_cross_attention=True, layer_num=layer_num,
                                                twin_cross=self.variant == "nlvr")
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self._cache = PreparedCache()
        self.last_prune = None
        self._enc_cache = None  # (key, compute-dtype copies of encoder_hidden_states) shared across layers by the encoder

    # ---- pruning -------------------------------------------------------------------------------------------
    def Reduce_token(self, x, reduce_num, temperature=0, self_attn=None, cls_attn=None, token_attn=None, mask=None):
        """med.py:345-391 / nlvr_encoder.py:400-454.  x: FULL [B,L,D] f32 (row 0 = [ENC]/CLS kept); mask: additive
        f32 [B,L].  Returns (x', mask')."""
        B, L, D = x.shape
        n = L - 1
        sa = self.attention.self
        score, thr, count, kmax = hip.token_score(sa.score_side, token_attn, temperature, B, sa.num_attention_heads, L)
        k = int(kmax.item())
        info = {"k": k, "score": score, "threshold": thr, "count": count, "pruned": False, "indices": None,
                "indices_sort": None}
        self.last_prune = info
        if k < 1 or (n - k) <= 1:
            return x, mask
        indices, indices_sort, dst_pos, merge_w = hip.token_select(score, k)
        info.update(pruned=True, indices=indices, indices_sort=indices_sort)
        y = hip.token_gather(x, dst_pos, merge_w, k)
        if mask is not None:
            if self.variant == "nlvr":
                mask = hip.mask_gather(mask, indices_sort, k)  # nlvr_encoder.py:452: indices_sort[:, :k+1]
            else:
                # med.py:377,388-390: topk(k+1, sorted=False) - kept tokens keep their own mask, the merged slot takes
                # the mask of the (k+1)-th ranked token (torch-CPU places it last)
                mask = hip.mask_gather(mask, indices, k, order2=indices_sort)
        return y, mask

    def _apply(self, fn, recurse=True):
        self.__dict__.pop("_madtp_params", None)
        return super()._apply(fn, recurse)

    # ---- forward --------------------------------------------------------------------------------------------
    def _head_mask_vec(self, head_mask, device):
        """a layer's head_mask ([H] or [1,H,1,1], med.py:215-217 via get_head_mask) as the f32 GPU vector [H], kept per caller tensor
        and version (it keys the prepared, head-scaled value projections)"""
        H = self.attention.self.num_attention_heads
        if head_mask.numel() != H:
            raise NotImplementedError(f"head_mask with {tuple(head_mask.shape)}: one factor per head ([{H}] or [1,{H},1,1]) is supported")
        store = self.__dict__.setdefault("_hm_vec", {})
        key = (head_mask.data_ptr(), head_mask._version, tuple(head_mask.shape), head_mask.dtype, str(head_mask.device))
        hit = store.get(key)
        if hit is None:
            if len(store) >= 4:
                store.clear()
            hit = store[key] = (head_mask, head_mask.detach().reshape(H).to(device=device, dtype=torch.float32).contiguous())
        return hit[1]

    def _scaled_value(self, sm, hm):
        """head_mask (med.py:215-217: attention_probs_dropped * head_mask, a constant per head) as a scaling of the VALUE projection of
        head h by hm[h]: P_h (hm_h V_h) = hm_h (P_h V_h) - the context, and the head-importance norms the pruning score takes from it
        (:229-231), are the reference's; the probabilities themselves (and cls_attn's first factor) stay unmasked, as there.  A
        weight / bias holder for lin_of (one-time weight preparation per head-mask tensor version; not a parameter)."""
        store = self.__dict__.setdefault("_hm_store", {})
        wv, bv = sm.value.weight, sm.value.bias
        sig = (hm.data_ptr(), hm._version, wv.data_ptr(), wv._version, bv.data_ptr(), bv._version)
        hit = store.get(id(sm))
        if hit is None or hit[0] != sig:
            rows = hm.repeat_interleave(sm.attention_head_size)
            hit = (sig, _LinHolder((wv.detach().float() * rows[:, None]).contiguous(), (bv.detach().float() * rows).contiguous()), hm)
            store[id(sm)] = hit
        return hit[1]

    def _weights(self, hm=None):
        """madtp_bert_layer_w for the layer-level C entry points (hm: a head-mask vector [H] on the GPU, see _scaled_value)."""
        sa, ao = self.attention.self, self.attention.output
        has_cross = hasattr(self, "crossattention")
        # the Parameter objects whose (data_ptr, version, device) key the prepared weights; collected once per module
        # (walking the sub-modules costs ~0.1 ms per layer call - visible in the launch-bound small-batch regime) and
        # dropped by _apply() (.to / .half / ...), which may replace Parameter objects
        params = self.__dict__.get("_madtp_params")
        if params is None or self.__dict__.get("_madtp_params_epoch") != param_epoch():
            self.__dict__["_madtp_params_epoch"] = param_epoch()
            own_modules(self)
            params = [sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias,
                      ao.dense.weight, ao.dense.bias, ao.LayerNorm.weight, ao.LayerNorm.bias, self.intermediate.dense.weight,
                      self.intermediate.dense.bias, self.output.dense.weight, self.output.dense.bias,
                      self.output.LayerNorm.weight, self.output.LayerNorm.bias]
            if has_cross:
                params += list(self.crossattention.parameters())
            self.__dict__["_madtp_params"] = params

        sfx = "" if hm is None else "_hm"
        val = (lambda sm: sm.value) if hm is None else (lambda sm: self._scaled_value(sm, hm))

        def build():
            keep = []

            def L(cache, key, mods):
                l = lin_of(cache, key, mods)
                keep.append(l)
                return hip.lin_struct(l)

            w = hip.BertLayerW()
            w.qkv = L(sa._cache, "qkv" + sfx, [sa.query, sa.key, val(sa)])
            w.attn_out = L(ao._cache, "d", [ao.dense])
            w.ln_att_g, w.ln_att_b = f32_ptr(ao.LayerNorm.weight, "LayerNorm parameter"), f32_ptr(ao.LayerNorm.bias, "LayerNorm parameter")
            w.cross, w.variant_nlvr, w.has_merge = 0, int(self.variant == "nlvr"), 0
            if has_cross:
                ca = self.crossattention
                co = ca.output
                if ca.twin:
                    w.cross = 2
                    for br, sm in enumerate((ca.self0, ca.self1)):
                        w.cq[br] = L(sm._cache, "q", [sm.query])
                        w.ckv[br] = L(sm._cache, "kv" + sfx, [sm.key, val(sm)])
                    w.cdense[0] = L(co._cache, "d0", [co.dense0])
                    w.cdense[1] = L(co._cache, "d1", [co.dense1])
                    if co.merge:
                        w.has_merge = 1
                        w.merge = L(co._cache, "mg", [co.merge_layer])
                    # fused twin projections (one-time weight preparation; see madtp_bert_layer_w in the header)
                    bf = compute_dtype() == torch.bfloat16
                    if not co.merge or bf:
                        w.cq_fused = L(ca._cache, "qq", [ca.self0.query, ca.self1.query])
                        fl = _fused_twin_output(co, fold_merge=co.merge)
                        keep.append(fl)
                        w.cdense_fused = hip.lin_struct(fl)
                        w.fused_twin = 2 if co.merge else 1
                else:
                    w.cross = 1
                    w.cq[0] = L(ca.self._cache, "q", [ca.self.query])
                    w.ckv[0] = L(ca.self._cache, "kv" + sfx, [ca.self.key, val(ca.self)])
                    w.cdense[0] = L(co._cache, "d", [co.dense])
                w.ln_cross_g, w.ln_cross_b = f32_ptr(co.LayerNorm.weight, "LayerNorm parameter"), f32_ptr(co.LayerNorm.bias, "LayerNorm parameter")
            w.inter = L(self._cache, "inter", [self.intermediate.dense])
            w.out = L(self._cache, "out", [self.output.dense])
            w.ln_out_g, w.ln_out_b = f32_ptr(self.output.LayerNorm.weight, "LayerNorm parameter"), f32_ptr(self.output.LayerNorm.bias, "LayerNorm parameter")
            w.eps, w.scale = self.output.LayerNorm.eps, 1.0 / math.sqrt(sa.attention_head_size)
            w.heads, w.dim = sa.num_attention_heads, sa.all_head_size
            w.dtype = dtype_code()
            return (w, keep)

        if hm is not None:
            mods = [sa] + ([self.crossattention.self0, self.crossattention.self1] if (has_cross and self.crossattention.twin)
                           else ([self.crossattention.self] if has_cross else []))
            held = [self._scaled_value(sm, hm) for sm in mods]
            return self._cache.get("w_hm", params + [t for h in held for t in (h.weight, h.bias)], build)[0]
        return self._cache.get("w", params, build)[0]

    def _forward(self, hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                 past_key_value, output_attentions, mode, token_attn, temperature):
        require_gpu(hidden_states, "hidden_states")
        hidden = as_f32_contig(hidden_states)
        B, L, D = hidden.shape
        hm = None
        if head_mask is not None:  # med.py:215-217; the reference's get_head_mask hands a layer [1,H,1,1] (or [H])
            hm = self._head_mask_vec(head_mask, hidden.device)
        if past_key_value is not None or output_attentions:
            # the rest of med.py:393-407's signature: the layer composed from the single kernels (an inference call; the pruned
            # encoder forward above never takes these arguments)
            return self._forward_general(hidden, attention_mask, hm, encoder_hidden_states, encoder_attention_mask, past_key_value,
                                         output_attentions, mode, token_attn, temperature)
        mask2d = causal = None
        if attention_mask is not None:
            if attention_mask.dim() == 4 and attention_mask.shape[2] == L and attention_mask.shape[3] == L and L > 1:
                # decoder mask (med.py:752-786): (1 - causal[L,L] * padding[B,L]) * -10000.  The kernels take the two factors
                # as additive masks [B,L] + [L,L] (masked entries underflow to exactly 0 either way).  BertModel.forward attaches
                # them; for a foreign tensor the padding part is the last query row (it sees every key causally) and the causal
                # part sample 0's mask with its padding removed.
                parts = getattr(attention_mask, "_madtp_causal", None)
                if parts is None:
                    mask2d = as_f32_contig(attention_mask[:, 0, L - 1, :])
                    causal = as_f32_contig(attention_mask[0, 0] - mask2d[0][None, :])
                else:
                    mask2d, causal = parts
                if temperature > 0:
                    raise NotImplementedError("token pruning inside the causal decoder is not a reference code path")
            elif attention_mask.dim() != 4 or attention_mask.shape[2] != 1:
                raise NotImplementedError("only padding masks [B,1,1,L] and decoder masks [B,1,L,L] are supported")
            else:
                mask2d = as_f32_contig(attention_mask[:, 0, 0, :])
        prune = temperature > 0
        if prune and mask2d is None:
            raise ValueError("attention_mask is required when temperature > 0 (med.py:424)")
        cross = mode == 'multimodal'
        # (a call that arrives with a pre-projected K/V cache - rank_answer / teacher-forced decoding against an EncoderKVCache -
        #  has no encoder tokens to differentiate through: it is an inference call whatever the parameters' requires_grad says)
        if hm is not None and torch.is_grad_enabled() and (hidden.requires_grad or (token_attn is not None and token_attn.requires_grad)):
            raise NotImplementedError("head_mask is an inference argument here: the hand-written backward has no head-mask term")
        if hm is None and torch.is_grad_enabled() and _autograd_precision() and "_kv_pre" not in self.__dict__:
            encs = (list(encoder_hidden_states) if isinstance(encoder_hidden_states, (list, tuple)) else [encoder_hidden_states]) \
                if (cross and encoder_hidden_states is not None) else []
            if (hidden.requires_grad or (token_attn is not None and token_attn.requires_grad) or any(e.requires_grad for e in encs)
                    or any(p.requires_grad for p in self.parameters())):
                # training / compression use: the layer as an autograd.Function around the same kernels
                # (madtp_amd/backward.py)
                from .backward import med_layer_forward_with_grad
                if cross:
                    assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
                enc_arg = enc_masks = None
                if cross and self.variant == "nlvr":
                    enc_arg = [as_f32_contig(e) for e in encoder_hidden_states]
                    if encoder_attention_mask is not None:
                        enc_masks = (self._enc_mask2d(encoder_attention_mask[0]), self._enc_mask2d(encoder_attention_mask[1]))
                elif cross:
                    enc_arg = encoder_hidden_states
                y, mask_out = med_layer_forward_with_grad(self, hidden, mask2d, temperature if prune else 0, token_attn, enc_arg,
                                                          enc_masks, causal)
                if mask_out is not None and causal is None:
                    attention_mask = mask_out[:, None, None, :]
                return (y, None, attention_mask)
        enc0 = enc1 = em0 = em1 = None
        Nk = 0
        pre = self.__dict__.pop("_kv_pre", None)  # (kv cache of THIS layer, Nk, int32 index [B]) set by the encoder for this call
        if cross and pre is not None:
            Nk = pre[1]
            # pre-projected K/V: the encoder tokens are not needed, their padding masks still are (nlvr_encoder.py:162,
            # 193-195 applies encoder_attention_mask in cross-attention; med.py:197-199 drops it)
            if self.variant == "nlvr" and encoder_attention_mask is not None:
                em0, em1 = self._enc_mask2d(encoder_attention_mask[0]), self._enc_mask2d(encoder_attention_mask[1])
        elif cross:
            assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
            if self.variant == "nlvr":
                Nk = encoder_hidden_states[0].shape[1]
                enc0, enc1 = self._enc_operand(encoder_hidden_states[0]), self._enc_operand(encoder_hidden_states[1])
                em0, em1 = self._enc_mask2d(encoder_attention_mask[0]), self._enc_mask2d(encoder_attention_mask[1])
            else:
                Nk = encoder_hidden_states.shape[1]
                enc0 = self._enc_operand(encoder_hidden_states)
        w = self._weights(hm)
        if causal is not None:  # a copy of the cached struct with this call's causal mask (kept alive by `causal` below)
            w = hip.BertLayerW.from_buffer_copy(w)
            w.self_mask_qk, w.ld_self_mask_qk = causal.data_ptr(), causal.stride(0)
        # one library call: self-attention + output LayerNorm, importance score / threshold / count (med.py:408-418,
        # 347-371), host read of k, [prune att + mask], cross-attention, FFN
        # (fast mode: the bf16 copy of the layer output, emitted by the output LayerNorm, rides along on the returned tensor
        #  so the next layer does not cast its input again; it is only trusted for the very same, unmodified tensor)
        lp = getattr(hidden_states, "_madtp_lp", None)
        if lp is not None and (lp[1] != hidden._version or hidden is not hidden_states or lp[0].shape[:-1] != hidden.shape[:-1]
                               or compute_dtype() == torch.float32 or lp[0].dtype != compute_dtype()):
            lp = None
        y, mask_out, self.last_prune, ylp = hip.bert_layer(w, hidden, mask2d, token_attn, temperature if prune else 0, cross, enc0,
                                                           enc1, Nk, em0, em1, hidden_lp=lp[0] if lp else None,
                                                           kv_pre=((pre[0] if isinstance(pre[0], tuple) else (pre[0], None))
                                                                   if (cross and pre) else (None, None)),
                                                           kv_index=pre[2] if (cross and pre) else None)
        if ylp is not None:
            y._madtp_lp = (ylp, y._version)
        if mask_out is not None:
            attention_mask = mask_out[:, None, None, :]
        return (y, None, attention_mask)  # present_key_value is not kept (encoder use, use_cache=False)

    def _forward_general(self, hidden, attention_mask, hm, encoder_hidden_states, encoder_attention_mask, past_key_value,
                         output_attentions, mode, token_attn, temperature):
        """BertLayer.forward with `past_key_value` and / or `output_attentions=True` (med.py:393-462; nlvr_encoder.py:484-554),
        composed from the single kernels in the current precision mode.  Returns the reference's tuple: (layer_output,
        [self-attention probabilities [B,H,L,Lk], cross-attention probabilities [B,H,L',Nk] (NLVR: one per twin branch, :531-543) if
        output_attentions], present_key_value = (k, v) f32 [B,H,Lk,64] (med.py:174), attention_mask [B,1,1,L'] or the caller's).
        past_key_value = (k, v) [B,H,Lp,64]: the cached keys / values go in front of this call's (med.py:164-168) - the queries
        are the L new tokens, the key mask covers Lp + L keys; pruning needs the square probabilities (the reference's Reduce_token
        dereferences cls_attn = None otherwise, :229,:424) and is refused together with a cache."""
        from .runtime import lin_of as _lin
        if torch.is_grad_enabled() and (hidden.requires_grad or (token_attn is not None and token_attn.requires_grad)):
            raise NotImplementedError("past_key_value / output_attentions are inference arguments here (no hand-written backward)")
        B, L, D = hidden.shape
        sa, so = self.attention.self, self.attention.output
        H, hd = sa.num_attention_heads, sa.attention_head_size
        scale = 1.0 / math.sqrt(hd)
        cdt, adt = compute_dtype(), attn_dtype()
        split = cdt == torch.float16
        prune = temperature > 0
        Lp = 0
        if past_key_value is not None:
            if prune:
                raise NotImplementedError("token pruning against cached keys is not a reference code path (Reduce_token needs the "
                                          "square attention map, med.py:229,424)")
            Lp = int(past_key_value[0].shape[2])
        Lk = Lp + L
        mask2d = causal = None
        if attention_mask is not None:
            if attention_mask.dim() != 4 or attention_mask.shape[-1] != Lk:
                raise ValueError(f"attention_mask {tuple(attention_mask.shape)}: expected [B,1,1 or L,{Lk}]")
            if attention_mask.shape[2] == 1:
                mask2d = as_f32_contig(attention_mask[:, 0, 0, :])
            elif attention_mask.shape[2] == L and Lp == 0:
                parts = getattr(attention_mask, "_madtp_causal", None)
                if parts is None:
                    mask2d = as_f32_contig(attention_mask[:, 0, L - 1, :])
                    causal = as_f32_contig(attention_mask[0, 0] - mask2d[0][None, :])
                else:
                    mask2d, causal = parts
            else:
                raise NotImplementedError("a [B,1,L,Lk] mask together with cached keys: feed the new tokens one at a time")
        if prune and mask2d is None:
            raise ValueError("attention_mask is required when temperature > 0 (med.py:424)")
        to_f32 = (lambda t: t if t.dtype == torch.float32 else hip.lp_to_f32(t))
        val = (lambda sm: sm.value) if hm is None else (lambda sm: self._scaled_value(sm, hm))
        sfx = "" if hm is None else "_hm"
        h2 = hidden.view(B * L, D)
        h_c = to_compute(h2)
        qkv = _lin(sa._cache, "qkv" + sfx, [sa.query, sa.key, val(sa)])
        y = hip.gemm(h_c, qkv.w, qkv.b, n=qkv.n, out_dtype=adt)  # [B*L, 3D]: q | k | v (v scaled per head under a head mask)
        q, k, v = y[:, :D], y[:, D:2 * D], y[:, 2 * D:]
        v_cache = v if hm is None else hip.gemm(h_c, *(lambda l: (l.w, l.b))(_lin(sa._cache, "v_raw", [sa.value])), n=D, out_dtype=adt)
        if Lp:
            def rows(t):  # [B,H,Lp,64] -> [B, Lp, D] in the attention kernels' dtype
                r = as_f32_contig(require_gpu(t, "past_key_value")).permute(0, 2, 1, 3).reshape(B, Lp, D).contiguous()
                return r if adt == torch.float32 else hip.cast_bf16(r)
            pk, pv = rows(past_key_value[0]), rows(past_key_value[1])
            k = torch.cat([pk, k.reshape(B, L, D)], 1).view(B * Lk, D)
            v_cache = torch.cat([pv, v_cache.reshape(B, L, D)], 1).view(B * Lk, D)
            if hm is None:
                v = v_cache
            else:  # the cache holds UNSCALED values (the reference masks the probabilities): scale the cached rows like the new ones
                pvs = hip.gemm(to_compute(as_f32_contig(pv if pv.dtype == torch.float32 else hip.lp_to_f32(pv)).view(B * Lp, D)),
                               *(lambda l: (l.w, l.b))(self._head_scale_lin(hm, D)), n=D, out_dtype=adt)
                v = torch.cat([pvs.view(B, Lp, D), v.reshape(B, L, D)], 1).view(B * Lk, D)
        o, side = hip.attention(q, k, v, B, H, L, Lk, scale, add_mask=mask2d, scores=prune, mask_qk=causal, split=split)
        present = (to_f32(k).reshape(B, Lk, H, hd).permute(0, 2, 1, 3), to_f32(v_cache).reshape(B, Lk, H, hd).permute(0, 2, 1, 3))
        extra = ()
        if output_attentions:
            extra = (hip.attention_probs_x(to_f32(q).contiguous(), to_f32(k).contiguous(), B, H, L, Lk, scale, key_mask=mask2d,
                                           mask_qk=causal),)
        dl = _lin(so._cache, "d", [so.dense])
        a0 = hip.gemm(o if o.dtype == cdt else to_compute(o), dl.w, dl.b, residual=h2, out_dtype=torch.float32, n=dl.n)
        att = hip.layernorm(a0, so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.eps)[0].view(B, L, D)  # med.py:246-250
        if prune:
            sa.score_side = side
            att, mask2d = self.Reduce_token(att, 0, temperature, token_attn=token_attn, mask=mask2d)  # med.py:421-437
            attention_mask = mask2d[:, None, None, :]
        L2 = att.shape[1]
        a2 = att.reshape(B * L2, D)
        if mode == 'multimodal':
            assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
            ca, co = self.crossattention, self.crossattention.output
            if ca.twin:
                em = encoder_attention_mask if encoder_attention_mask is not None else (None, None)
                branches = [(ca.self0, encoder_hidden_states[0], self._enc_mask2d(em[0])),
                            (ca.self1, encoder_hidden_states[1], self._enc_mask2d(em[1]))]
            else:
                branches = [(ca.self, encoder_hidden_states, None)]  # (med.py:197-199: the encoder mask is not applied)
            a_c = to_compute(a2)
            ctxs, cps = [], []
            for sm, e, emask in branches:
                Nk = e.shape[1]
                ql = _lin(sm._cache, "q", [sm.query])
                kvl = _lin(sm._cache, "kv" + sfx, [sm.key, val(sm)])
                cq = hip.gemm(a_c, ql.w, ql.b, n=ql.n, out_dtype=adt)
                ckv = hip.gemm(self._enc_operand(e), kvl.w, kvl.b, n=kvl.n, out_dtype=adt)
                c, _ = hip.attention(cq, ckv[:, :D], ckv[:, D:], B, H, L2, Nk, scale, add_mask=emask, split=split)
                ctxs.append(c if c.dtype == cdt else to_compute(c))
                if output_attentions:
                    cps.append(hip.attention_probs_x(to_f32(cq).contiguous(), to_f32(ckv[:, :D]).contiguous(), B, H, L2, Nk, scale,
                                                     key_mask=emask))
            if ca.twin:  # nlvr_encoder.py:259-271
                d0l, d1l = _lin(co._cache, "d0", [co.dense0]), _lin(co._cache, "d1", [co.dense1])
                d0 = hip.gemm(ctxs[0], d0l.w, d0l.b, n=d0l.n, out_dtype=torch.float32)
                d1 = hip.gemm(ctxs[1], d1l.w, d1l.b, n=d1l.n, out_dtype=torch.float32)
                if co.merge:
                    ml = _lin(co._cache, "mg", [co.merge_layer])
                    c0 = hip.gemm(to_compute(torch.cat([d0, d1], 1).contiguous()), ml.w, ml.b, residual=a2, out_dtype=torch.float32, n=ml.n)
                else:
                    c0 = hip.add_scale(hip.add_scale(d0, d1, 0.5), a2.contiguous(), 1.0)
            else:
                cl = _lin(co._cache, "d", [co.dense])
                c0 = hip.gemm(ctxs[0], cl.w, cl.b, residual=a2, out_dtype=torch.float32, n=cl.n)
            a2 = hip.layernorm(c0, co.LayerNorm.weight, co.LayerNorm.bias, co.LayerNorm.eps)[0]
            extra = extra + tuple(cps)
        il, ol = _lin(self._cache, "inter", [self.intermediate.dense]), _lin(self._cache, "out", [self.output.dense])
        mid = hip.gemm(to_compute(a2), il.w, il.b, act=hip.ACT_GELU, n=il.n)                                   # med.py:312-315
        f0 = hip.gemm(mid, ol.w, ol.b, residual=a2, out_dtype=torch.float32, n=ol.n)                            # :326-328
        yout = hip.layernorm(f0, self.output.LayerNorm.weight, self.output.LayerNorm.bias, self.output.LayerNorm.eps)[0]
        return (yout.view(B, L2, D),) + extra + (present, attention_mask)

    def _head_scale_lin(self, hm, D):
        """diag(head-mask per column) as a prepared Linear: scales cached (unscaled) value rows like the scaled value projection"""
        store = self.__dict__.setdefault("_hm_store", {})
        sig = (hm.data_ptr(), hm._version)
        hit = store.get("diag")
        if hit is None or hit[0] != sig:
            rows = hm.repeat_interleave(D // hm.numel())
            hit = (sig, _LinHolder(torch.diag(rows).contiguous(), torch.zeros(D, device=hm.device)))
            store["diag"] = hit
        from .runtime import lin_of as _lin
        return _lin(self._cache, "hm_diag", [hit[1]])

    def _enc_operand(self, enc):
        """compute-dtype 2-D copy of an encoder tensor, shared by the 12 layers that receive the same tensor object
        (the entry keeps `enc` alive, so identity + version cannot alias a recycled allocation)."""
        dt = compute_dtype()
        for src, ver, d, val in _ENC_STORE:
            if src is enc and ver == enc._version and d == dt:
                return val
        lp = getattr(enc, "_madtp_lp", None)  # compute-dtype copy emitted by the producer (the ViT's final LayerNorm)
        if (lp is not None and lp[1] == enc._version and dt != torch.float32 and lp[0].dtype == dt and lp[0].is_contiguous()
                and lp[0].shape[:-1] == enc.shape[:-1]):
            return lp[0].view(-1, lp[0].shape[-1])
        e = as_f32_contig(enc)
        val = _cast(e.view(-1, e.shape[-1]))
        _ENC_STORE.append((enc, enc._version, dt, val))
        while len(_ENC_STORE) > 4:
            _ENC_STORE.pop(0)
        return val

    @staticmethod
    def _enc_mask2d(m):
        if m is None:
            return None
        return as_f32_contig(m[:, 0, 0, :]) if m.dim() == 4 else as_f32_contig(m)

    def feed_forward_chunk(self, attention_output):
        raise NotImplementedError("fused into forward()")


class MedBertLayer(_BertLayerBase):
    """models/med.py BertLayer :332-467."""
    variant = "med"

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, mode=None, space_dict=None,
                token_attn=None, reduce_num=0, temperature=0):
        return self._forward(hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                             past_key_value, output_attentions, mode, token_attn, temperature)


class NlvrBertLayer(_BertLayerBase):
    """models/nlvr_encoder.py BertLayer :385-559 (note the different positional order: space_dict is arg 3)."""
    variant = "nlvr"

    def forward(self, hidden_states, attention_mask=None, space_dict=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, mode=None, token_attn=None,
                reduce_num=0, temperature=0):
        return self._forward(hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                             past_key_value, output_attentions, mode, token_attn, temperature)


class _Out(tuple):
    """BaseModelOutput* stand-in: tuple with attribute access."""

    def __new__(cls, last_hidden_state):
        o = super().__new__(cls, (last_hidden_state,))
        o.last_hidden_state = last_hidden_state
        o.past_key_values = None
        o.hidden_states = None
        o.attentions = None
        o.cross_attentions = None
        o.pooler_output = None
        return o


class _BertEncoderBase(nn.Module):
    layer_cls = MedBertLayer

    def __init__(self, config, sd_dim=768):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([self.layer_cls(config, i) for i in range(config.num_hidden_layers)])
        self.gradient_checkpointing = False
        self.txt_query_model = Query_model(ft_dim=config.hidden_size, sd_dim=sd_dim, temperature=1,
                                           att_func_type='sparsemax', pool_type='max')
        self._cache = PreparedCache()

    def _run(self, hidden_states, attention_mask, space_dict, temperature, encoder_hidden_states, encoder_attention_mask,
             mode, always_query):
        sd_txt_ft_all = None
        cache = self.__dict__.pop("_kv_cache", None)  # EncoderKVCache for THIS call (MedBertModel.forward(encoder_kv_cache=...))
        if torch.is_grad_enabled() and _autograd_precision() and cache is None:
            encs = [] if encoder_hidden_states is None else (list(encoder_hidden_states) if isinstance(encoder_hidden_states, (list, tuple))
                                                              else [encoder_hidden_states])
            if (hidden_states.requires_grad or (space_dict is not None and space_dict.requires_grad)
                    or any(e.requires_grad for e in encs) or any(p.requires_grad for p in self.parameters())):
                from .backward import bert_encoder_forward_with_grad  # (the layer loop under autograd)
                self.__dict__.pop("_prepared_weights", None)
                h, _, sd_all = bert_encoder_forward_with_grad(self, hidden_states, attention_mask, space_dict, temperature,
                                                              encoder_hidden_states, encoder_attention_mask, mode, always_query)
                return _Out(h), sd_all
        if (_use_encoder_call(hidden_states.shape[0], _ENCODER_CALL) and not _KV_AHEAD
                and all(type(l) is self.layer_cls for l in self.layer)):
            out = self._run_encoder_call(hidden_states, attention_mask, space_dict, temperature, encoder_hidden_states,
                                         encoder_attention_mask, mode, always_query, cache)
            if out is not None:
                return out
        defer = self.txt_query_model.deferred() if space_dict is not None else None
        reduce_num = int((hidden_states.shape[-2] - 1) // self.config.num_hidden_layers)
        ahead = None
        if cache is None and mode == 'multimodal' and encoder_hidden_states is not None and _KV_AHEAD:
            ahead = self._project_encoder_tokens(encoder_hidden_states)
        for i, layer_module in enumerate(self.layer):
            layer_module.__dict__.pop("_kv_pre", None)
            if cache is not None and mode == 'multimodal':
                layer_module._kv_pre = (cache.kv[i], cache.Nk, cache.index)
            elif ahead is not None:
                kvs, Nk, width = ahead
                layer_module._kv_pre = (tuple(kv[:, i * width:(i + 1) * width] for kv in kvs) if len(kvs) == 2
                                        else kvs[0][:, i * width:(i + 1) * width], Nk, None)
            token_attn = None
            if space_dict is not None or always_query:
                if space_dict is None:
                    raise TypeError("nlvr_encoder.BertEncoder calls txt_query_model unconditionally (:608): "
                                    "space_dict must be given")
                token_attn, sd_txt_ft_all, _ = self.txt_query_model(hidden_states[:, 1:, :], space_dict,
                                                                    return_token_att=True, temperature=temperature,
                                                                    acc_ft=sd_txt_ft_all, defer=defer)
            t = temperature if space_dict is not None else 0
            if self.layer_cls.variant == "nlvr":
                outs = layer_module(hidden_states, attention_mask, space_dict, None, encoder_hidden_states,
                                    encoder_attention_mask, None, False, mode=mode, token_attn=token_attn,
                                    reduce_num=reduce_num, temperature=t)
            else:
                outs = layer_module(hidden_states, attention_mask, None, encoder_hidden_states, encoder_attention_mask,
                                    None, False, mode=mode, space_dict=space_dict, token_attn=token_attn,
                                    reduce_num=reduce_num, temperature=t)
            hidden_states = outs[0]
            attention_mask = outs[-1]
        if defer is not None and defer.pairs:
            sd_txt_ft_all = defer.finish()
        return _Out(hidden_states), sd_txt_ft_all


    def _run_options(self, hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask, past_key_values,
                     use_cache, output_attentions, output_hidden_states, mode, space_dict, temperature, always_query):
        """BertEncoder.forward's layer loop WITH the arguments the pruned-encoder call never passes (med.py:478-598;
        nlvr_encoder.py:570-687): head_mask[i] and past_key_values[i] go to layer i (:509-510, :554-566), use_cache collects every
        layer's present_key_value (`layer_outputs[-2]`, :571-572), output_attentions the self- / cross-attention probabilities
        (:573-575), output_hidden_states the layer inputs and the final output (:507-508, :577-578).  One layer call per iteration on
        the composed path (BertLayer._forward_general); returns (BaseModelOutput-like tuple with .last_hidden_state,
        .past_key_values, .hidden_states, .attentions, .cross_attentions, sd_txt_ft_all)."""
        all_hidden = () if output_hidden_states else None
        all_self = () if output_attentions else None
        all_cross = () if (output_attentions and mode == 'multimodal') else None
        next_cache = () if use_cache else None
        sd_txt_ft_all = None
        require_gpu(hidden_states, "hidden_states")
        for i, layer_module in enumerate(self.layer):
            layer_module.__dict__.pop("_kv_pre", None)
            if output_hidden_states:
                all_hidden = all_hidden + (hidden_states,)
            token_attn = None
            if space_dict is not None or always_query:
                if space_dict is None:
                    raise TypeError("nlvr_encoder.BertEncoder calls txt_query_model unconditionally (:608): space_dict must be given")
                token_attn, sd_txt_ft_all, _ = self.txt_query_model(hidden_states[:, 1:, :], space_dict, return_token_att=True,
                                                                    temperature=temperature, acc_ft=sd_txt_ft_all)
            t = temperature if space_dict is not None else 0
            lhm = head_mask[i] if head_mask is not None else None
            hm = layer_module._head_mask_vec(lhm, hidden_states.device) if lhm is not None else None
            outs = layer_module._forward_general(as_f32_contig(hidden_states), attention_mask, hm, encoder_hidden_states,
                                                 encoder_attention_mask, past_key_values[i] if past_key_values is not None else None,
                                                 bool(output_attentions), mode, token_attn, t)
            hidden_states, attention_mask = outs[0], outs[-1]
            if use_cache:
                next_cache = next_cache + (outs[-2],)
            if output_attentions:
                all_self = all_self + (outs[1],)
                if all_cross is not None:
                    all_cross = all_cross + ((outs[2] if len(outs) == 5 else tuple(outs[2:-2])),)
        if output_hidden_states:
            all_hidden = all_hidden + (hidden_states,)
        out = _Out(hidden_states)
        out.past_key_values, out.hidden_states, out.attentions, out.cross_attentions = next_cache, all_hidden, all_self, all_cross
        return out, sd_txt_ft_all

    def _apply(self, fn, recurse=True):
        self.__dict__.pop("_enc_weights", None)  # .to() / .half() may replace Parameter objects
        return super()._apply(fn, recurse)

    def _encoder_weights(self):
        ew = self.__dict__.get("_enc_weights")
        if ew is None:
            ew = self.__dict__["_enc_weights"] = EncoderWeights()
        return ew.get(list(self.layer))

    def prepare_encoder_call(self, batch=None):
        """Optional hint (extension): validate / build the layers' weight structs NOW - a task model calls it before it runs
        the vision encoder, so that the ~0.1 ms pass over this encoder's ~500 parameters overlaps GPU work instead of sitting
        between the embedding kernel and the first layer.  Consumed by the next forward of this encoder."""
        if (batch is None or _use_encoder_call(batch, _ENCODER_CALL)) and _ENCODER_CALL and not _KV_AHEAD \
                and all(type(l) is self.layer_cls for l in self.layer):
            self.__dict__["_prepared_weights"] = (param_epoch(), get_precision(), self._encoder_weights())

    def _run_encoder_call(self, hidden_states, attention_mask, space_dict, temperature, encoder_hidden_states,
                          encoder_attention_mask, mode, always_query, cache):
        """The layer loop of _run() as ONE library call (madtp_bert_encoder): the same kernels in the same order as the
        per-layer path without the return to Python between layers.  None: fall back to the per-layer path."""
        require_gpu(hidden_states, "hidden_states")
        query = space_dict is not None or always_query
        if query and space_dict is None:
            raise TypeError("nlvr_encoder.BertEncoder calls txt_query_model unconditionally (:608): space_dict must be given")
        hidden = as_f32_contig(hidden_states)
        B, L, D = hidden.shape
        mask2d = None
        if attention_mask is not None:
            if attention_mask.dim() == 4 and attention_mask.shape[2] == L and L > 1:
                return None  # decoder (causal) mask: the per-layer path carries it (madtp_bert_layer_w.self_mask_qk)
            if attention_mask.dim() != 4 or attention_mask.shape[2] != 1:
                raise NotImplementedError("only padding masks [B,1,1,L] and decoder masks [B,1,L,L] are supported")
            mask2d = as_f32_contig(attention_mask[:, 0, 0, :])
        t = temperature if space_dict is not None else 0
        if t > 0 and mask2d is None:
            raise ValueError("attention_mask is required when temperature > 0 (med.py:424)")
        qargs, deferred = (None, False)
        if query:
            qargs, deferred = self.txt_query_model.encoder_args(space_dict, B, D, hidden.device)
            if qargs is None:
                return None
        l0 = self.layer[0]
        cross = mode == 'multimodal'
        enc0 = enc1 = em0 = em1 = None
        kv0 = kv1 = kv_index = None
        Nk = 0
        nlvr = self.layer_cls.variant == "nlvr"
        if cross and cache is not None:
            Nk, kv0, kv_index = cache.Nk, list(cache.kv), cache.index
            if nlvr:
                return None
        elif cross:
            assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
            if nlvr:
                Nk = encoder_hidden_states[0].shape[1]
                enc0, enc1 = l0._enc_operand(encoder_hidden_states[0]), l0._enc_operand(encoder_hidden_states[1])
                em0, em1 = l0._enc_mask2d(encoder_attention_mask[0]), l0._enc_mask2d(encoder_attention_mask[1])
            else:
                Nk = encoder_hidden_states.shape[1]
                enc0 = l0._enc_operand(encoder_hidden_states)
        lp = getattr(hidden_states, "_madtp_lp", None)
        if lp is not None and (lp[1] != hidden._version or hidden is not hidden_states or lp[0].shape[:-1] != hidden.shape[:-1]
                               or compute_dtype() == torch.float32 or lp[0].dtype != compute_dtype()):
            lp = None
        ws = self.__dict__.pop("_prepared_weights", None)  # validated by prepare_encoder_call() earlier in this forward
        if ws is not None and ws[:2] == (param_epoch(), get_precision()):
            ws = ws[2]
        else:  # no hint, or parameters re-assigned / precision switched since the hint
            ws = self._encoder_weights()
        kv_ld = 0
        if kv0 is not None:
            for tns in kv0:
                if tns.stride(-1) != 1 or (kv_ld and kv_ld != tns.stride(0)):
                    raise RuntimeError("encoder_kv_cache tensors must be row-major with one common row stride")
                kv_ld = tns.stride(0)
        # device-side lengths (madtp_bert_encoder_async, opt-in with the ViT's: MADTP_ENCODER_SYNC_FREE=1): no host read of k
        # between the layers; shapes it takes: hip.bert_encoder_sync_free_ok
        from . import vit as _vit
        sync_free = _vit._SYNC_FREE and query and t > 0 and hip.bert_encoder_sync_free_ok(B, L, Nk, True, qargs, mask2d)
        run = hip.bert_encoder(ws, hidden, lp[0] if lp else None, mask2d, qargs, t, cross, enc0, enc1, Nk, em0, em1,
                               kv_pre0=kv0, kv_pre1=kv1, kv_index=kv_index, kv_ld=kv_ld, sync_free=sync_free)
        for l, layer in enumerate(self.layer):
            layer.last_prune = run.info(l, t if query else 0)
            layer.__dict__.pop("_kv_pre", None)
        sd_txt_ft_all = None
        qm = self.txt_query_model
        if query and qm.compute_att_ft:
            if deferred:
                K = space_dict.shape[0]
                segs = []
                for l in range(len(self.layer)):
                    n_in = run.n_in(l)
                    xp = hidden.data_ptr() if l == 0 else run.ptr(l - 1, "y")
                    segs.append((run.ptr(l, "logits") + 128 * 4, xp + D * 4, n_in - 1, 128, n_in * 128, D, n_in * D))
                sd_txt_ft_all = hip.query_att_ft_multi_ptrs(segs, B, K, D, hidden.device, sd_dim=qm.att_dim, exact={"exact": True, "split": "split"}.get(deferred, False))
            else:
                sd_txt_ft_all = qargs["att_ft"]
        self._last_run = run
        out = run.output(len(self.layer) - 1)
        if run.lp_dtype is not None:  # compute-dtype copy of the last layer's output (emitted by its output LayerNorm)
            nl = len(self.layer) - 1
            out._madtp_lp = (run.view(nl, "y_lp", run.lp_dtype, B, out.shape[1], (2 * D if run.lp_dtype == torch.float16 else D)),
                             out._version)
        return _Out(out), sd_txt_ft_all

    def _project_encoder_tokens(self, encoder_hidden_states):
        """The cross-attention [k|v] projections of ALL layers in one GEMM per branch: the layers' fused key|value weights
        stacked along N ([layers*2*hidden, hidden], prepared once), the encoder tokens read once.  The reference projects
        inside every layer (med.py:178-179, nlvr_encoder.py:177-178); the values are the same (each output column is the
        same dot product), but 24 one-round launches of 240 tiles become 2 launches of 2880.  Layer i reads the column
        slice [i*2*hidden, (i+1)*2*hidden) through kv_ld.  -> ([kv per branch], Nk, 2*hidden) or None."""
        encs = list(encoder_hidden_states) if isinstance(encoder_hidden_states, (list, tuple)) else [encoder_hidden_states]
        if not all(getattr(l, "has_cross", hasattr(l, "crossattention")) for l in self.layer):
            return None
        kvs = []
        for br, enc in enumerate(encs):
            mods = []
            for l in self.layer:
                ca = l.crossattention
                sm = (ca.self0, ca.self1)[br] if ca.twin else ca.self
                mods += [sm.key, sm.value]
            lin = lin_of(self._cache, ("kv_all", br), mods)
            kvs.append(hip.gemm(self.layer[0]._enc_operand(enc), lin.w, lin.b, n=lin.n, out_dtype=attn_dtype()))
        return kvs, encs[0].shape[1], 2 * self.config.hidden_size


class MedBertEncoder(_BertEncoderBase):
    """models/med.py BertEncoder :470-598."""
    layer_cls = MedBertLayer

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_values=None, use_cache=None, output_attentions=False,
                output_hidden_states=False, return_dict=True, mode='multimodal', space_dict=None, temperature=0):
        if head_mask is not None or past_key_values is not None or use_cache or output_attentions or output_hidden_states:
            return self._run_options(hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                                     past_key_values, use_cache, output_attentions, output_hidden_states, mode, space_dict,
                                     temperature, always_query=False)
        return self._run(hidden_states, attention_mask, space_dict, temperature, encoder_hidden_states,
                         encoder_attention_mask, mode, always_query=False)


class NlvrBertEncoder(_BertEncoderBase):
    """models/nlvr_encoder.py BertEncoder :562-687."""
    layer_cls = NlvrBertLayer

    def forward(self, hidden_states, attention_mask=None, space_dict=None, temperature=0, head_mask=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None, use_cache=None,
                output_attentions=False, output_hidden_states=False, return_dict=True, mode='multimodal'):
        if head_mask is not None or past_key_values is not None or use_cache or output_attentions or output_hidden_states:
            return self._run_options(hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                                     past_key_values, use_cache, output_attentions, output_hidden_states, mode, space_dict,
                                     temperature, always_query=True)
        return self._run(hidden_states, attention_mask, space_dict, temperature, encoder_hidden_states,
                         encoder_attention_mask, mode, always_query=True)


class _BertModelBase(nn.Module):
    encoder_cls = MedBertEncoder

    def __init__(self, config, add_pooling_layer=True, sd_dim=768):
        super().__init__()
        if add_pooling_layer:
            raise NotImplementedError("BLIP builds its text encoders with add_pooling_layer=False")
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = self.encoder_cls(config, sd_dim)
        self.pooler = None
        if not getattr(config, "evaluate", True):
            self.apply(self._init_weights)

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def get_extended_attention_mask(self, attention_mask, input_shape=None, device=None, is_decoder=False):
        """med.py:728-786: (1 - m) * -10000 broadcast as [B,1,1,L]; is_decoder (2-D mask): m = causal[L,L] * padding[B,L] as
        [B,1,L,L] (:752-768), with its two additive factors attached for the layers' kernels (`_madtp_causal`)."""
        if is_decoder and attention_mask.dim() == 2:
            B, L = attention_mask.shape
            ids = torch.arange(L, device=attention_mask.device)
            causal = (ids[None, :] <= ids[:, None])                               # [L(query), L(key)]
            pad = attention_mask != 0
            ext = torch.where(causal[None, None, :, :] & pad[:, None, None, :], 0.0, -10000.0)
            ext._madtp_causal = (torch.where(pad, 0.0, -10000.0).contiguous(), torch.where(causal, 0.0, -10000.0).contiguous())
            return ext
        if attention_mask.dim() == 3:
            ext = attention_mask[:, None, :, :]
        elif attention_mask.dim() == 2:
            ext = attention_mask[:, None, None, :]
        else:
            raise ValueError("Wrong shape for input_ids (shape {}) or attention_mask (shape {})".format(
                input_shape, attention_mask.shape))
        if not ext.dtype.is_floating_point:  # {0,1} masks: the same values (up to the sign of zero) in two launches instead of three
            return torch.where(ext != 0, 0.0, -10000.0)
        return (1.0 - ext.to(torch.float32)) * -10000.0

    def invert_attention_mask(self, m):
        return self.get_extended_attention_mask(m)

    def _run(self, input_ids, attention_mask, space_dict, temperature, encoder_embeds, encoder_hidden_states,
             encoder_attention_mask, mode, inputs_embeds=None, is_decoder=False, past_len=0):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds (pre-embedded text WITHOUT the position embeddings / LayerNorm, "
                                      "med.py:63-86) is not on the pruned forward path; pass input_ids or encoder_embeds")
        if input_ids is not None:
            batch_size, seq_length = input_ids.size()
            device = input_ids.device
        elif encoder_embeds is not None:
            batch_size, seq_length = encoder_embeds.size()[:-1]
            device = encoder_embeds.device
        else:
            raise ValueError("You have to specify either input_ids or inputs_embeds or encoder_embeds")
        if attention_mask is None:
            attention_mask = torch.ones((batch_size, seq_length + past_len), device=device)  # med.py:842-843
        if past_len and is_decoder and attention_mask.dim() == 2:
            # med.py:752-786 with a prefix: a new token sees every cached position and, causally, the new ones; for ONE new token that
            # is the padding mask over past + 1 keys
            if seq_length != 1:
                raise NotImplementedError("is_decoder with past_key_values: feed one new token per call (as generate does)")
            ext = self.get_extended_attention_mask(attention_mask, (batch_size, seq_length), device, False)
        else:
            ext = self.get_extended_attention_mask(attention_mask, (batch_size, seq_length), device, is_decoder)
        if encoder_hidden_states is not None:
            if isinstance(encoder_hidden_states, list):
                # (None entry: no padding in that image's tokens - the same values as an all-ones mask without its launches)
                enc_ext = [None if m is None else self.invert_attention_mask(m) for m in encoder_attention_mask]
            elif encoder_attention_mask is None:
                enc_ext = None
            else:
                enc_ext = self.invert_attention_mask(encoder_attention_mask)
        else:
            enc_ext = None
        emb = self.embeddings(input_ids=input_ids, past_key_values_length=past_len) if encoder_embeds is None else encoder_embeds
        return emb, ext, enc_ext


class MedBertModel(_BertModelBase):
    """models/med.py BertModel :686-929."""
    encoder_cls = MedBertEncoder

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, inputs_embeds=None,
                encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=False,
                mode='multimodal', space_dict=None, temperature=0, encoder_kv_cache=None):
        """encoder_kv_cache (extension): an EncoderKVCache - the layers' cross-attention then reads the cached [k|v]
        projections of encoder block index[b] for sample b instead of projecting encoder_hidden_states (which may be None)."""
        if position_ids is not None:
            raise NotImplementedError("position_ids: only the default consecutive positions are implemented")
        if is_decoder and temperature > 0:
            raise NotImplementedError("token pruning inside the causal decoder is not a reference code path")
        past_len = int(past_key_values[0][0].shape[2]) if past_key_values is not None else 0  # med.py:838
        emb, ext, enc_ext = self._run(input_ids, attention_mask, space_dict, temperature, encoder_embeds,
                                      encoder_hidden_states, encoder_attention_mask, mode, inputs_embeds, is_decoder, past_len)
        options = (head_mask is not None or past_key_values is not None or use_cache or output_attentions or output_hidden_states)
        if options:
            # the rest of med.py:803-929's signature (incremental decoding through the reference's own past_key_values / use_cache
            # protocol, attention outputs, head masks): the per-layer composed path of the encoder
            if encoder_kv_cache is not None:
                raise NotImplementedError("encoder_kv_cache together with past_key_values / output_attentions / head_mask")
            n_layers = len(self.encoder.layer)
            hms = None
            if head_mask is not None:  # transformers get_head_mask (med.py:875): [H] -> every layer, [layers, H] -> one row per layer
                hms = [head_mask] * n_layers if head_mask.dim() == 1 else [head_mask[i] for i in range(n_layers)]
            return self.encoder(emb, attention_mask=ext, head_mask=hms, encoder_hidden_states=encoder_hidden_states,
                                encoder_attention_mask=enc_ext, past_key_values=past_key_values, use_cache=use_cache,
                                output_attentions=bool(output_attentions), output_hidden_states=bool(output_hidden_states),
                                mode=mode, space_dict=space_dict, temperature=temperature)
        if encoder_kv_cache is not None:
            if self.encoder.layer_cls.variant != "med":
                raise NotImplementedError("encoder_kv_cache is wired for the single-cross-attention (MED) layers")
            self.encoder._kv_cache = encoder_kv_cache
        out, sd_txt_ft = self.encoder(emb, attention_mask=ext, encoder_hidden_states=encoder_hidden_states,
                                      encoder_attention_mask=enc_ext, mode=mode, space_dict=space_dict,
                                      temperature=temperature)
        return out, sd_txt_ft


class EncoderKVCache:
    """Cross-attention [k|v] projections of a set of encoder token blocks, one tensor [blocks*Nk, 2*hidden] per layer in the
    compute dtype (bit-identical to what each layer would project itself), plus the block index of every sample of the next
    forward (int32 [B], set with .select()).  At 288 GB per GPU the projections of a whole retrieval test set stay resident
    (7 MB per 190-token image): re-ranking projects every image once instead of once per (query, candidate) pair."""

    def __init__(self, kv, Nk):
        self.kv, self.Nk, self.index = kv, Nk, None

    def select(self, index):
        self.index = index.to(torch.int32).contiguous()
        return self

    @staticmethod
    def build(bert_model, enc):
        """enc: [blocks, Nk, hidden] f32 GPU tensor (e.g. the padded image tokens of the evaluation set)."""
        enc = as_f32_contig(require_gpu(enc, "enc"))
        blocks, Nk, D = enc.shape
        a = _cast(enc.view(blocks * Nk, D))
        kv = []
        for l in bert_model.encoder.layer:
            sm = l.crossattention.self
            lin = lin_of(sm._cache, "kv", [sm.key, sm.value])
            kv.append(hip.gemm(a, lin.w, lin.b, n=lin.n, out_dtype=attn_dtype()))  # the dtype the attention kernels read
        return EncoderKVCache(kv, Nk)


class BertPredictionHeadTransform(nn.Module):
    """med.py:616-631: dense, erf-GELU, LayerNorm."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        if config.hidden_act != "gelu":
            raise ValueError("only the erf-GELU of med_config.json is implemented")
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    """med.py:634-651: transform + decoder Linear(hidden -> vocab, bias=False) with the separate output bias linked to it."""

    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias  # :648 (the same Parameter under both state-dict names)
        self._cache = PreparedCache()


class BertOnlyMLMHead(nn.Module):
    """med.py:654-661."""

    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)


VOCAB_PAD = 8  # the vocabulary GEMM writes ceil(V / 8) * 8 columns (30524 -> 30528): 16-byte vector stores on every row


class _LMOut:
    """CausalLMOutputWithCrossAttentions stand-in (med.py:1046-1060)."""

    def __init__(self, loss, logits):
        self.loss, self.logits = loss, logits
        self.past_key_values = self.hidden_states = self.attentions = self.cross_attentions = None


class BertLMHeadModel(nn.Module):
    """models/med.py BertLMHeadModel :933-1094, teacher-forced use (labels / logits of whole sequences: BLIP_VQA.rank_answer,
    blip_vqa.py:156-203) and beam-search generation (`generate`, madtp_amd/generation.py; no past_key_values: every step re-runs the
    prefix)."""

    def __init__(self, config, sd_dim=768):
        super().__init__()
        self.config = config
        self.bert = MedBertModel(config, add_pooling_layer=False, sd_dim=sd_dim)
        self.cls = BertOnlyMLMHead(config)
        if not getattr(config, "evaluate", True):
            self.apply(self.bert._init_weights)
        self.tie_weights()

    def tie_weights(self):
        """transformers PreTrainedModel.tie_weights (config.tie_word_embeddings, default True): the LM head's output embedding IS
        the input embedding matrix (one Parameter under two state-dict names, as in every BLIP checkpoint)."""
        if getattr(self.config, "tie_word_embeddings", True):
            self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def set_output_embeddings(self, new_embeddings):
        self.cls.predictions.decoder = new_embeddings

    def _vocab_linear(self):
        """decoder weight padded to 128 rows in the compute dtype + the output bias padded likewise, N = ceil(V / 8) * 8."""
        from .runtime import Lin, prepare_linear
        pr = self.cls.predictions

        def build():
            lin = prepare_linear([pr.decoder.weight], None, compute_dtype())
            V = pr.decoder.weight.shape[0]
            b = torch.zeros(lin.w.shape[0], device=pr.bias.device, dtype=torch.float32)
            b[:V] = pr.bias.detach().float()
            return Lin(lin.w, b, (V + VOCAB_PAD - 1) // VOCAB_PAD * VOCAB_PAD)
        return pr._cache.get(("vocab", compute_dtype()), [pr.decoder.weight, pr.bias], build)

    def prediction_scores(self, sequence_output):
        """self.cls(sequence_output) med.py:1027: [B, L, D] f32 -> f32 view [B, L, V] of a [B, L, ceil(V/8)*8] buffer."""
        pr = self.cls.predictions
        B, L, D = sequence_output.shape
        x = as_f32_contig(sequence_output).view(B * L, D)
        tl = lin_of(pr._cache, "tdense", [pr.transform.dense])
        h = hip.gemm(to_compute(x), tl.w, tl.b, out_dtype=torch.float32, act=hip.ACT_GELU, n=tl.n)
        cdt = compute_dtype()
        h32, hlp = hip.layernorm(h, pr.transform.LayerNorm.weight, pr.transform.LayerNorm.bias, pr.transform.LayerNorm.eps,
                                 lp=None if cdt == torch.float32 else cdt)
        vl = self._vocab_linear()
        logits = hip.gemm(h32 if hlp is None else hlp, vl.w, vl.b, out_dtype=torch.float32, n=vl.n)
        V = pr.decoder.weight.shape[0]
        return logits.view(B, L, vl.n)[:, :, :V], logits.view(B, L, vl.n)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, inputs_embeds=None,
                encoder_hidden_states=None, encoder_attention_mask=None, labels=None, past_key_values=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, return_logits=False, is_decoder=True,
                reduction='mean', mode='multimodal', space_dict=None, temperature=0, train=False, encoder_kv_cache=None):
        """encoder_kv_cache (extension, as MedBertModel.forward): cached cross-attention [k|v] of the encoder states - sample b
        attends to block index[b], so rank_answer does not tile / re-project the question states per candidate."""
        outputs, sd_txt_ft = self.bert(input_ids, attention_mask=attention_mask, position_ids=position_ids, head_mask=head_mask,
                                       inputs_embeds=inputs_embeds, encoder_hidden_states=encoder_hidden_states,
                                       encoder_attention_mask=encoder_attention_mask, past_key_values=past_key_values,
                                       use_cache=use_cache, output_attentions=output_attentions,
                                       output_hidden_states=output_hidden_states, is_decoder=is_decoder, mode=mode,
                                       space_dict=space_dict, temperature=temperature, encoder_kv_cache=encoder_kv_cache)
        if torch.is_grad_enabled() and _autograd_precision() and outputs[0].requires_grad:
            # training use: the LM head as autograd Functions on the exact-f32 GEMM, the label-smoothed
            # next-token cross-entropy of :1033-1042 as torch ops on the [B (L-1), V] scores
            from .backward import LayerNormFunction, LinearFunction
            import torch.nn.functional as F
            pr = self.cls.predictions
            B, L, D = outputs[0].shape
            h = LinearFunction.apply(outputs[0].reshape(B * L, D), pr.transform.dense.weight, pr.transform.dense.bias, hip.ACT_GELU)
            h = LayerNormFunction.apply(h, pr.transform.LayerNorm.weight, pr.transform.LayerNorm.bias, pr.transform.LayerNorm.eps)
            scores = LinearFunction.apply(h, pr.decoder.weight, pr.bias, hip.ACT_NONE).view(B, L, -1)
            if return_logits:
                return scores[:, :-1, :].contiguous()
            lm_loss = None
            if labels is not None:
                V = scores.shape[-1]
                lm_loss = F.cross_entropy(scores[:, :-1, :].reshape(-1, V), labels[:, 1:].reshape(-1).to(torch.int64),
                                          reduction=reduction, label_smoothing=0.1)
                if reduction == 'none':
                    lm_loss = lm_loss.view(B, -1).sum(1)
            if not (return_dict if return_dict is not None else True):
                return ((lm_loss, scores) if lm_loss is not None else (scores,))
            out = _LMOut(lm_loss, scores)
            return (out, sd_txt_ft) if train else out
        scores, padded = self.prediction_scores(outputs[0])
        if return_logits:
            return scores[:, :-1, :].contiguous()  # :1029-1030
        lm_loss = None
        if labels is not None:
            # :1033-1042: shifted next-token cross-entropy, label_smoothing 0.1; per-sequence sums from one kernel
            V = scores.shape[-1]
            per_seq = hip.lm_loss(padded, labels.to(torch.int64).contiguous(), V, 0.1)
            if reduction == 'none':
                lm_loss = per_seq
            elif reduction == 'sum':
                lm_loss = per_seq.sum()
            else:  # 'mean' over the non-ignored target tokens
                lm_loss = per_seq.sum() / (labels[:, 1:] != -100).sum().clamp(min=1)
        if not (return_dict if return_dict is not None else True):
            return ((lm_loss, scores) if lm_loss is not None else (scores,))
        out = _LMOut(lm_loss, scores)
        # med.py:1046-1060: the encoder's optional outputs ride along (the reference's incremental-decoding protocol: past_key_values
        # in, use_cache=True -> out.past_key_values back)
        out.past_key_values, out.hidden_states = getattr(outputs, "past_key_values", None), getattr(outputs, "hidden_states", None)
        out.attentions, out.cross_attentions = getattr(outputs, "attentions", None), getattr(outputs, "cross_attentions", None)
        return (out, sd_txt_ft) if train else out

    def prepare_inputs_for_generation(self, input_ids, past=None, attention_mask=None, **model_kwargs):
        """med.py:1071-1089: with a cache only the last token is fed; the attention mask covers the whole prefix."""
        if attention_mask is None:
            attention_mask = input_ids.new_ones(input_ids.shape)
        if past is not None:
            input_ids = input_ids[:, -1:]
        return {"input_ids": input_ids, "attention_mask": attention_mask, "past_key_values": past,
                "encoder_hidden_states": model_kwargs.get("encoder_hidden_states", None),
                "encoder_attention_mask": model_kwargs.get("encoder_attention_mask", None), "is_decoder": True}

    def _reorder_cache(self, past, beam_idx):
        """med.py:1091-1094"""
        return tuple(tuple(past_state.index_select(0, beam_idx) for past_state in layer_past) for layer_past in past)

    def generate(self, input_ids, max_length=20, min_length=0, num_beams=1, eos_token_id=None, pad_token_id=None,
                 repetition_penalty=1.0, length_penalty=1.0, early_stopping=False, do_sample=False, encoder_hidden_states=None,
                 encoder_attention_mask=None, **unused):
        """The beam-search use of transformers' `generate` at the reference's call sites (models/blip_vqa.py:134-140,
        models/blip.py:189-196): input_ids [B, t0] prompt, encoder_hidden_states ALREADY repeated num_beams times per item
        ([B * num_beams, N, D], as the reference passes them) -> int64 [B, <= max_length].  Every step runs the decoder over the
        whole prefix (med.py prepare_inputs_for_generation :1071-1089: attention_mask = ones, is_decoder=True); the encoder
        states are projected to every layer's cross-attention [k|v] ONCE per item and beam j of item b reads block b.
        encoder_attention_mask is accepted and ignored: MED cross-attention drops the encoder mask (med.py:197-199)."""
        top_p = unused.pop("top_p", 1.0) if do_sample else None
        top_k = unused.pop("top_k", 50) if do_sample else None  # (transformers: config.top_k = 50 unless the caller passes one)
        generator = unused.pop("generator", None)
        # keyword arguments of transformers' generate that would change the result must not be dropped silently
        for key, val in unused.items():
            if key == "num_return_sequences" and val == 1:
                continue
            if key in ("top_p", "top_k", "temperature") and not do_sample:  # sampling-only knobs
                continue
            if key == "use_cache":
                continue
            if key == "attention_mask" and (val is None or bool((val != 0).all())):
                continue  # an all-ones prompt mask is the default prepare_inputs_for_generation builds (med.py:1075-1077)
            raise TypeError(f"generate(): unsupported argument {key}={val!r} (beam search of the reference's call sites only: "
                            "num_beams, max_length, min_length, eos / pad ids, repetition_penalty, length_penalty, early_stopping)")
        if do_sample:
            num_beams = 1  # models/blip.py:175-186: one sampled sequence per image (num_return_sequences = 1)
        elif num_beams < 2:
            raise NotImplementedError("greedy search: the reference's call sites use num_beams = 3 or do_sample = True")
        if eos_token_id is None or pad_token_id is None or encoder_hidden_states is None:
            raise ValueError("generate: eos_token_id, pad_token_id and encoder_hidden_states are required")
        from . import generation
        require_gpu(encoder_hidden_states, "encoder_hidden_states")
        dev = encoder_hidden_states.device
        B = input_ids.shape[0]
        if encoder_hidden_states.shape[0] != B * num_beams:
            raise ValueError("generate: encoder_hidden_states must hold num_beams copies per prompt (repeat_interleave, "
                             "as models/blip_vqa.py:128 / models/blip.py:165 pass them)")
        ehs = encoder_hidden_states.reshape(B, num_beams, *encoder_hidden_states.shape[1:])
        if not bool((ehs == ehs[:, :1]).all()):
            raise ValueError("generate: the num_beams copies of an item's encoder_hidden_states differ - they are projected once "
                             "per item here (the reference repeats ONE image num_beams times, models/blip.py:165)")
        cache = EncoderKVCache.build(self.bert, encoder_hidden_states[::num_beams].contiguous())
        sel = cache.select(torch.arange(B, device=dev).repeat_interleave(num_beams))
        V = self.cls.predictions.decoder.weight.shape[0]

        def step_full(ids):
            outputs, _ = self.bert(ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                                   is_decoder=True, mode='multimodal', encoder_kv_cache=sel)
            _, padded = self.prediction_scores(outputs[0][:, -1:, :].contiguous())
            return padded[:, 0, :]

        # Incremental decoding (med.py:1071-1094; round 5): one new token per beam and step through madtp_bert_decode_step against
        # the layers' self-attention K/V cache [layers, rows, Lmax, 2 D]; the prompt is fed token by token (<= 4 tokens at the
        # reference's call sites), the beams' re-ordering is a gather over the cache rows.  MADTP_DECODE_CACHE=0 (or a decoder
        # whose layers the step does not take) re-runs the whole prefix every step.
        rows = B * num_beams
        enc = self.bert.encoder
        layers = list(enc.layer)
        use_cache = (os.environ.get("MADTP_DECODE_CACHE", "1") != "0" and max_length <= 256
                     and all(type(l) is enc.layer_cls and getattr(l, "has_cross", hasattr(l, "crossattention")) for l in layers))
        state = {"t": 0, "cache": None}
        item_index = sel.index[::num_beams].contiguous()

        def step_cached(ids, beam_src=None):
            emb = self.bert.embeddings
            D = emb.word_embeddings.weight.shape[1]
            if state["cache"] is None:
                state["cache"] = torch.zeros((len(layers), rows, max_length, 2 * D), device=dev, dtype=attn_dtype())
                state["spare"] = torch.zeros_like(state["cache"])
            elif beam_src is not None:  # _reorder_cache (:1091-1094): the filled positions of the source rows into the other buffer
                hip.kv_cache_reorder(state["cache"], state["spare"], beam_src, state["t"])
                state["cache"], state["spare"] = state["spare"], state["cache"]
            ws = enc._encoder_weights()
            y = None
            while state["t"] < ids.shape[1]:  # the prompt on the first call, one token afterwards
                t = state["t"]
                x, _ = hip.bert_embed(ids[:, t:t + 1].contiguous(), emb.word_embeddings.weight, emb.position_embeddings.weight[t:],
                                      emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps, lp=None)
                # (an item's num_beams rows share its encoder K/V block: one cross-attention sequence of num_beams queries per item)
                y = hip.bert_decode_step(ws, x.view(rows, D), state["cache"], t, sel.kv, item_index, 0, sel.Nk, group=num_beams)
                state["t"] = t + 1
            _, padded = self.prediction_scores(y.view(rows, 1, D))
            return padded[:, 0, :]
        step = step_cached if use_cache else step_full
        prompt = input_ids.to(dev).to(torch.int64).repeat_interleave(num_beams, dim=0)
        if do_sample:
            with torch.no_grad():
                return generation.sample(step, prompt, max_length, min_length, eos_token_id, pad_token_id, V, top_p, top_k=top_k,
                                         repetition_penalty=repetition_penalty, generator=generator)
        with torch.no_grad():
            return generation.beam_search(step, prompt, num_beams, max_length, min_length, eos_token_id, pad_token_id, V,
                                          repetition_penalty=repetition_penalty, length_penalty=length_penalty,
                                          early_stopping=early_stopping)


class NlvrBertModel(_BertModelBase):
    """models/nlvr_encoder.py BertModel :775-1015."""
    encoder_cls = NlvrBertEncoder

    def forward(self, input_ids=None, attention_mask=None, space_dict=None, temperature=0, position_ids=None,
                head_mask=None, inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_values=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, is_decoder=False, mode='multimodal'):
        if position_ids is not None or head_mask is not None or past_key_values is not None or is_decoder:
            raise NotImplementedError("position_ids / head_mask / past_key_values / is_decoder are off the pruned encoder path")
        emb, ext, enc_ext = self._run(input_ids, attention_mask, space_dict, temperature, encoder_embeds,
                                      encoder_hidden_states, encoder_attention_mask, mode, inputs_embeds)
        out, sd_txt_ft = self.encoder(emb, attention_mask=ext, space_dict=space_dict, temperature=temperature,
                                      encoder_hidden_states=encoder_hidden_states, encoder_attention_mask=enc_ext,
                                      mode=mode)
        return out, sd_txt_ft
