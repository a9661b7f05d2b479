"""Mirror of the reference's models/vit.py (Mlp :15-36, Attention :39-103, Block :106-207, VisionTransformer
:210-310): same constructor arguments, sub-module / parameter names (checkpoint keys) and forward() signatures,
so models/blip_*.py and the compress_*_dtp.py drivers can use these classes unchanged.  forward() enqueues the
hand-written gfx950 kernels (madtp_amd.hip); there is no eager fallback.

Differences a caller can observe (documented in INTEGRATION.md):
  * kept tokens come out in ascending token order (the reference's topk(sorted=False) order is implementation
    defined; attention is permutation-equivariant so downstream values agree to f32 rounding);
  * `token_attn` is not divided by the temperature in place (vit.py:137 mutates the caller's tensor);
  * the [B,H,N,N] attention map is never materialised by the forward: get_attention_map() returns None unless
    Attention.keep_attention_map is set, in which case it is recomputed on demand (exact-f32) from the layer's input;
    register_hook=True (gradient hooks on the map, Grad-CAM): the block's backward leaves the gradient of the attention
    probabilities in get_attn_gradients() (fp32 mode, grad mode on).
"""
from functools import partial

import torch
import torch.nn as nn

from . import hip
from .runtime import autograd_precision as _autograd_precision, EncoderWeights, PreparedCache, param_epoch, own_modules, get_precision, encoder_call_preference, f32_ptr, attn_dtype, compute_dtype, dtype_code, lin_of, require_gpu, as_f32_contig, to_compute
from .utils import Query_model, vector_gather  # noqa: F401  (re-exported like `from models.utils import *`)

import os as _os
# MADTP_ENCODER_CALL: "auto" (default) uses the encoder-level calls (madtp_vit_encoder / madtp_bert_encoder) in the launch-bound
# regime only - up to ENCODER_CALL_MAX_ROWS token rows (B * N) entering the encoder; "1" always, "0" never (A/B runs).
# Measured on the NLVR forward (same box, images/s): 1 sample 467 -> 573 (+23 %), 16 samples 7.21 k -> 7.45 k (+3 %),
# 64 samples 19.3 k -> 18.8 k (-3 %: the call's host-side setup sits exposed in front of each encoder, while the per-layer
# path spreads its host work under the previous layer's kernels).
_ENCODER_CALL = {"0": False, "1": True}.get(_os.environ.get("MADTP_ENCODER_CALL", "auto"), "auto")
ENCODER_CALL_MAX_ROWS = 8192
# MADTP_ENCODER_SYNC_FREE=1: the encoder-level call without a host read of k (madtp_vit_encoder_async, device-side lengths).
# Bit-identical to the host-k paths (tests/test_model_parity_gpu.py) but measured SLOWER where it applies (profiles/
# r04_latency_table_*.txt: ViT alone at 1 / 16 images 1.92 / 2.28 ms against 1.43 / 1.77 ms): the host read of k already hides
# under the projection GEMM, while the sync-free kernels run the unpruned sequence's grids and key-tile instantiations
# (attention at 82 tokens on the 13-tile kernel).  Off by default; it is the form to capture in a graph / enqueue ahead.
_SYNC_FREE = _os.environ.get("MADTP_ENCODER_SYNC_FREE", "0") == "1"


def use_encoder_call(rows, flag=None):
    flag = _ENCODER_CALL if flag is None else flag
    if flag == "auto" and encoder_call_preference() is not None:
        return bool(encoder_call_preference())
    return (rows <= ENCODER_CALL_MAX_ROWS) if flag == "auto" else bool(flag)


class Mlp(nn.Module):
    """vit.py:15-36 - fc2(GELU(fc1(x))) (dropout p=0 in every BLIP config)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.hidden_features = hidden_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self._cache = PreparedCache()

    def run(self, h2d, residual2d=None):
        """h2d: [M, in] in the compute dtype -> f32 [M, out] (+ residual)."""
        fc1 = lin_of(self._cache, "fc1", [self.fc1])
        fc2 = lin_of(self._cache, "fc2", [self.fc2])
        mid = hip.gemm(h2d, fc1.w, fc1.b, act=hip.ACT_GELU, n=fc1.n)
        return hip.gemm(mid, fc2.w, fc2.b, residual=residual2d, out_dtype=torch.float32, n=fc2.n)

    def forward(self, x):
        require_gpu(x)
        shp = x.shape
        h = as_f32_contig(x).view(-1, shp[-1])
        return self.run(to_compute(h)).view(*shp[:-1], -1)


class Attention(nn.Module):
    """vit.py:39-103."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim != 64:
            raise ValueError("the gfx950 attention kernels are built for head_dim 64 (ViT-B/16, BERT-base, CLIP-B)")
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.attn_gradients = None
        self.attention_map = None
        self.cls_attn = None
        self.score_side = None  # (colsum_part, p0, onorm) of the last call - consumed by Block.Reduce_token
        self._cache = PreparedCache()
        # vit.py:57-73, 83: the reference stores the [B,H,N,N] map of every call.  The kernels never materialise it; with
        # keep_attention_map = True the layer remembers its (normalised) input instead and get_attention_map() recomputes the map
        # on demand - exact-f32 arithmetic whatever the precision mode (q, k from an f32 GEMM, madtp_attention_probs).
        self.keep_attention_map = False
        self._map_input = None

    def save_attn_gradients(self, attn_gradients):
        self.attn_gradients = attn_gradients

    def get_attn_gradients(self):
        return self.attn_gradients

    def save_attention_map(self, attention_map):
        self.attention_map = attention_map

    def get_attention_map(self):
        if self.attention_map is None and self._map_input is not None:  # (set by keep_attention_map or register_hook=True)
            h32, B, N = self._map_input  # LayerNorm output of the last call, f32 [B*N, dim]
            with torch.no_grad():
                qk = lin_of(self._cache, "qkv", [self.qkv], torch.float32)
                y = hip.gemm(h32, qk.w, qk.b, n=qk.n, out_dtype=torch.float32)
                C = self.dim
                self.attention_map = hip.attention_probs(y[:, :C], y[:, C:2 * C], B, self.num_heads, N, self.scale)
        return self.attention_map

    def save_cls_attn(self, cls_attn):
        self.cls_attn = cls_attn

    def get_cls_attn(self):
        return self.cls_attn

    def run(self, h2d, B, N, residual2d=None, want_scores=True):
        """h2d: normalised tokens [B*N, dim] in the compute dtype -> f32 [B*N, dim] (= proj(attn) + residual)."""
        qkv = lin_of(self._cache, "qkv", [self.qkv])
        proj = lin_of(self._cache, "proj", [self.proj])
        C = self.dim
        y = hip.gemm(h2d, qkv.w, qkv.b, n=qkv.n, out_dtype=attn_dtype())  # [B*N, 3C]: q | k | v, head-major (vit.py:77)
        o, side = hip.attention(y[:, :C], y[:, C:2 * C], y[:, 2 * C:], B, self.num_heads, N, N, self.scale,
                                scores=want_scores, split=compute_dtype() == torch.float16)
        self.score_side = side
        if o.dtype != compute_dtype():  # f16x3: the f32 context enters the projection as f16 planes
            o = to_compute(o)
        return hip.gemm(o, proj.w, proj.b, residual=residual2d, out_dtype=torch.float32, n=proj.n)

    def forward(self, x, register_hook=False):
        if register_hook:
            raise NotImplementedError("attention-gradient hooks need the materialised map (training path, out of scope)")
        require_gpu(x)
        B, N, C = x.shape
        h = as_f32_contig(x).view(B * N, C)
        return self.run(to_compute(h), B, N).view(B, N, C)


class Block(nn.Module):
    """vit.py:106-207."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_grad_checkpointing=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = nn.Identity()  # (the fused inference call has no DropPath; the training forward applies drop_path_rate
        self.drop_path_rate = float(drop_path)  # in .train() mode: madtp_amd/backward.py, vit.py:114,186,205)
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.last_prune = None  # dict(k, indices, indices_sort, score, threshold, count) of the last forward
        self._cache = PreparedCache()

    def _ln(self, norm, x2d):
        cdt = compute_dtype()
        lp = None if cdt == torch.float32 else cdt
        y32, ylp = hip.layernorm(x2d, norm.weight, norm.bias, norm.eps, want_f32=lp is None, lp=lp)
        return y32 if lp is None else ylp

    def Reduce_token(self, x, reduce_num=0, temperature=0, self_attn=None, cls_attn=None, token_attn=None):
        """vit.py:123-163.  `x` is the FULL token tensor [B,N,D] here (CLS included) - the kernels skip row 0 -
        and the attention statistics come from self.attn.score_side instead of a materialised map."""
        B, N, D = x.shape
        n = N - 1
        score, thr, count, kmax = hip.token_score(self.attn.score_side, token_attn, temperature, B, self.attn.num_heads, N)
        k = int(kmax.item())  # topk_num = max_b count: one host sync per layer, as vit.py:145
        info = {"k": k, "score": score, "threshold": thr, "count": count, "pruned": False, "indices": None,
                "indices_sort": None}
        self.last_prune = info
        if k < 1 or (n - k) <= 1:  # vit.py:148-149
            return x
        indices, indices_sort, dst_pos, merge_w = hip.token_select(score, k)
        info.update(pruned=True, indices=indices, indices_sort=indices_sort)
        return hip.token_gather(x, dst_pos, merge_w, k)

    def _apply(self, fn, recurse=True):
        self.__dict__.pop("_madtp_params", None)
        return super()._apply(fn, recurse)

    def _weights(self):
        """madtp_vit_block_w for the layer-level C entry points (rebuilt when a parameter or the precision changes)."""
        params = self.__dict__.get("_madtp_params")  # collected once per module, dropped by _apply() (see bert.py)
        if params is None or self.__dict__.get("_madtp_params_epoch") != param_epoch():
            self.__dict__["_madtp_params_epoch"] = param_epoch()
            own_modules(self)
            params = [self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, self.attn.qkv.weight,
                      self.attn.qkv.bias, self.attn.proj.weight, self.attn.proj.bias, self.mlp.fc1.weight,
                      self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias]
            self.__dict__["_madtp_params"] = params

        def build():
            lins = [lin_of(self.attn._cache, "qkv", [self.attn.qkv]), lin_of(self.attn._cache, "proj", [self.attn.proj]),
                    lin_of(self.mlp._cache, "fc1", [self.mlp.fc1]), lin_of(self.mlp._cache, "fc2", [self.mlp.fc2])]
            w = hip.VitBlockW()
            w.ln1_g, w.ln1_b = f32_ptr(self.norm1.weight, "LayerNorm parameter"), f32_ptr(self.norm1.bias, "LayerNorm parameter")
            w.ln2_g, w.ln2_b = f32_ptr(self.norm2.weight, "LayerNorm parameter"), f32_ptr(self.norm2.bias, "LayerNorm parameter")
            w.eps, w.scale = self.norm1.eps, self.attn.scale
            w.qkv, w.proj, w.fc1, w.fc2 = [hip.lin_struct(l) for l in lins]
            w.heads, w.dim = self.attn.num_heads, self.attn.dim
            w.dtype = dtype_code()
            w.act = hip.ACT_GELU
            return (w, lins)  # keep the prepared tensors alive next to the raw pointers

        return self._cache.get("w", params, build)[0]

    def forward(self, x, register_hook=False, reduce_num=0, temperature=0, token_attn=None):
        require_gpu(x, "x")
        x = as_f32_contig(x)
        B, N, D = x.shape
        prune = temperature > 0
        if register_hook:
            # vit.py:88-90 (Grad-CAM): the attention map is saved and its gradient captured.  Here: the map is recomputed on request
            # (get_attention_map(), f32) and the block's backward leaves d loss / d attention-probabilities [B,H,N,N] in
            # get_attn_gradients() - which needs the autograd path, i.e. grad mode on and the fp32 precision mode.
            from .runtime import get_precision as _gp
            if not (torch.is_grad_enabled() and _autograd_precision()):
                raise NotImplementedError("register_hook=True captures attention gradients in the block's backward: it needs grad "
                                          "mode on and runtime.precision('fp32')")
            self.attn._hook_attn_gradients = True
            self.attn.attention_map = None
            self.attn._map_input = (hip.layernorm(x.detach().view(B * N, D), self.norm1.weight.detach(), self.norm1.bias.detach(),
                                                  self.norm1.eps)[0], B, N)
            from .backward import block_forward_with_grad
            return block_forward_with_grad(self, x, temperature if prune else 0, token_attn)
        if prune and token_attn is None:
            raise ValueError("temperature > 0 requires token_attn (the reference fails in Reduce_token as well)")
        if self.attn.keep_attention_map:  # get_attention_map() support (see Attention.__init__)
            self.attn.attention_map = None
            self.attn._map_input = (hip.layernorm(x.detach().view(B * N, D), self.norm1.weight.detach(), self.norm1.bias.detach(),
                                                  self.norm1.eps)[0], B, N)
        if torch.is_grad_enabled():
            # training / compression (compress_nlvr_dtp.py:46-58): Block.forward under autograd with the hand-written backward of
            # madtp_amd/backward.py (fp32 precision mode).  An input that asks for a gradient in another mode fails loudly;
            # parameters that merely have requires_grad = True (the nn.Module default) do not force the autograd path there.
            wants = x.requires_grad or (token_attn is not None and token_attn.requires_grad)
            from .runtime import get_precision
            if wants or (_autograd_precision() and any(p.requires_grad for p in self.parameters())):
                from .backward import block_forward_with_grad
                return block_forward_with_grad(self, x, temperature if prune else 0, token_attn)
        w = self._weights()
        # one library call: x = x + attn(norm1(x)), importance score / threshold / count (vit.py:186-190,125-145), the host
        # read of k = max_b count (vit.py:145), [top-k, gather, merge] and x = x + mlp(norm2(x)) (vit.py:153-161,195-205)
        y, self.last_prune = hip.vit_block(w, x, token_attn, temperature if prune else 0)
        return y


class PatchEmbed(nn.Module):
    """timm==0.4.12 PatchEmbed (un-vendored dependency of the reference; call site vit.py:241-242): parameter
    `proj` is a Conv2d(k=s=patch) so checkpoints load by name; executed as im2col + MFMA GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self._cache = PreparedCache()

    def run(self, img):
        img = as_f32_contig(require_gpu(img, "image"))
        B, C, H, W = img.shape
        if H != W or H % self.patch_size[0]:
            raise ValueError(f"image {H}x{W} is not a square multiple of the patch size")
        lin = lin_of(self._cache, "proj", [self.proj])
        cols = hip.patchify(img, self.patch_size[0], compute_dtype())
        return hip.gemm(cols, lin.w, lin.b, out_dtype=torch.float32, n=lin.n), (H // self.patch_size[0]) ** 2

    def forward(self, x):
        y, np_ = self.run(x)
        return y.view(x.shape[0], np_, -1)


class VisionTransformer(nn.Module):
    """vit.py:210-310."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=None,
                 use_grad_checkpointing=False, ckpt_layer=0, evaluate=False, sd_dim=768, map_func=False):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]  # stochastic depth decay rule (vit.py:250)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.depth = depth
        if not evaluate:
            nn.init.trunc_normal_(self.pos_embed, std=.02)
            nn.init.trunc_normal_(self.cls_token, std=.02)
            self.apply(self._init_weights)
        self.img_query_model = Query_model(ft_dim=embed_dim, sd_dim=sd_dim, temperature=1, att_func_type='sparsemax',
                                           pool_type='max', map_func=map_func)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def forward(self, x, register_blk=-1, space_dict=None, temperature=0, _pending=None):
        """_pending (extension used by BLIP_NLVR): a list - the fast-mode sum of the layers' att_ft then runs on the
        auxiliary stream and the caller makes its stream wait (handle.sync()) before sd_img_ft_all is consumed."""
        B = x.shape[0]
        if (torch.is_grad_enabled() and _autograd_precision() and type(self) is VisionTransformer
                and (any(p.requires_grad for p in self.parameters()) or (space_dict is not None and space_dict.requires_grad))):
            # training / compression use: every stage of the forward as an autograd.Function around the same
            # kernels (madtp_amd/backward.py); inference callers run under torch.no_grad() as the reference's evaluate() does
            from .backward import vit_forward_with_grad
            return vit_forward_with_grad(self, x, space_dict, temperature, register_blk)
        enc_prep = None
        if (register_blk == -1 and use_encoder_call(B * (self.patch_embed.num_patches + 1), _ENCODER_CALL)
                and all(type(b) is Block and not b.attn.keep_attention_map for b in self.blocks)
                and not (torch.is_grad_enabled() and x.requires_grad)):
            # host-only preparation of the encoder-level call FIRST (weight structs, query-model operands): it then overlaps
            # the tail of whatever the GPU is still running instead of sitting between the patch embedding and the first layer
            enc_prep = self._encoder_call_prep(x, space_dict)
        patches, np_ = self.patch_embed.run(x)  # vit.py:283
        x = hip.assemble_tokens(patches, self.cls_token, self.pos_embed, B, np_)  # vit.py:285-289
        token_num = x.shape[-2]
        reduce_num = int((token_num - 1) // self.depth)
        sd_img_ft_all = None
        if enc_prep is not None:
            return self._forward_encoder_call(x, space_dict, temperature, _pending, enc_prep)
        defer = self.img_query_model.deferred() if space_dict is not None else None
        for i, blk in enumerate(self.blocks):
            if space_dict is not None:
                # :297-303; `sd_img_ft_all += sd_img_ft` is folded into the kernels: parity mode accumulates into the
                # running sum per layer, fast mode sums all layers in one launch after the loop (DeferredAttFt)
                token_attn, sd_img_ft_all, _ = self.img_query_model(x[:, 1:, :], space_dict, return_token_att=True,
                                                                    acc_ft=sd_img_ft_all, defer=defer)
                x = blk(x, register_blk == i, reduce_num, temperature, token_attn)  # :304
            else:
                x = blk(x, register_blk == i)
        if defer is not None and defer.pairs:
            sd_img_ft_all = defer.finish(_pending)
        B, N, D = x.shape
        return _final_norm(self, x), sd_img_ft_all  # :309


def _vit_encoder_call_prep(self, img, space_dict):
    """(weights, qargs, deferred) of the encoder-level call, or None when the query model cannot run inside it (q_map)."""
    qargs, deferred = (None, False)
    if space_dict is not None:
        qargs, deferred = self.img_query_model.encoder_args(space_dict, img.shape[0], self.embed_dim, img.device)
        if qargs is None:
            return None
    ew = self.__dict__.get("_enc_weights")
    if ew is None:
        ew = self.__dict__["_enc_weights"] = EncoderWeights()
    return ew.get(list(self.blocks)), qargs, deferred


def _vit_forward_encoder_call(self, x, space_dict, temperature, _pending, prep):
    """The block loop of forward() as ONE library call (madtp_vit_encoder): same kernels in the same order as the per-layer
    path, without the return to Python between layers."""
    B, N, D = x.shape
    qm = self.img_query_model
    weights, qargs, deferred = prep
    prune_t = temperature if (space_dict is not None and temperature > 0) else 0
    # device-side lengths (madtp_vit_encoder_async): no host read of k between the layers (opt-in, see _SYNC_FREE above; shapes:
    # B * N < 4096 token rows, N <= 256, deferred att_ft)
    sync_free = _SYNC_FREE and hip.vit_encoder_sync_free_ok(B, N, prune_t > 0, qargs)
    run = hip.vit_encoder(weights, x, qargs, temperature if space_dict is not None else 0, sync_free=sync_free)
    for l, blk in enumerate(self.blocks):
        blk.last_prune = run.info(l, prune_t)
        blk.attn.score_side = None
    sd_img_ft_all = None
    if space_dict is not None and qm.compute_att_ft:
        if deferred:  # fast mode: all layers' att_ft in one launch (on the auxiliary stream when the caller joins it later)
            K = space_dict.shape[0]
            segs, xin = [], x
            for l in range(len(self.blocks)):
                n_in = run.n_in(l)
                lg = run.ptr(l, "logits")
                xp = xin.data_ptr() if l == 0 else run.ptr(l - 1, "y")
                segs.append((lg + 128 * 4, xp + D * 4, n_in - 1, 128, n_in * 128, D, n_in * D))

            def launch():
                return hip.query_att_ft_multi_ptrs(segs, B, K, D, x.device, sd_dim=qm.att_dim, exact={"exact": True, "split": "split"}.get(deferred, False))
            if _pending is None:
                sd_img_ft_all = launch()
            else:
                from .runtime import side_stream
                from .utils import _SideWork
                main, side = torch.cuda.current_stream(), side_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    sd_img_ft_all = launch()
                sd_img_ft_all.record_stream(main)
                run.buf.record_stream(side)
                _pending.append(_SideWork(side, (run, x)))
        else:
            sd_img_ft_all = qargs["att_ft"]
    xo = run.output(len(self.blocks) - 1)
    y = _final_norm(self, xo)  # vit.py:309
    self._last_run = run  # keeps the layers' buffers (and the lazily built last_prune views) alive until the next forward
    return y, sd_img_ft_all


def _final_norm(self, x):
    """vit.py:309; in the low-precision modes the LayerNorm also emits the compute-dtype copy of its output, which rides on the
    returned tensor (`_madtp_lp`) for consumers that feed it to a GEMM (the text encoders' cross-attention K/V projections) -
    they would otherwise cast the f32 image tokens again."""
    cdt = compute_dtype()
    y, ylp = hip.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps, lp=None if cdt == torch.float32 else cdt)
    if ylp is not None:
        y._madtp_lp = (ylp, y._version)
    return y


def _vit_apply(self, fn, recurse=True):
    self.__dict__.pop("_enc_weights", None)  # .to() / .half() may replace Parameter objects
    return nn.Module._apply(self, fn, recurse)


VisionTransformer._forward_encoder_call = _vit_forward_encoder_call
VisionTransformer._encoder_call_prep = _vit_encoder_call_prep
VisionTransformer._apply = _vit_apply


def interpolate_pos_embed(pos_embed_checkpoint, visual_encoder):
    """models/vit.py:398-422: bicubic resize of a checkpoint's position-embedding grid to this encoder's patch grid (the CLS
    / extra tokens are kept); host-side weight preparation, run once per checkpoint load."""
    embedding_size = pos_embed_checkpoint.shape[-1]
    num_patches = visual_encoder.patch_embed.num_patches
    num_extra_tokens = visual_encoder.pos_embed.shape[-2] - num_patches
    orig_size = int((pos_embed_checkpoint.shape[-2] - num_extra_tokens) ** 0.5)
    new_size = int(num_patches ** 0.5)
    if orig_size == new_size:
        return pos_embed_checkpoint
    extra_tokens = pos_embed_checkpoint[:, :num_extra_tokens]
    grid = pos_embed_checkpoint[:, num_extra_tokens:].reshape(-1, orig_size, orig_size, embedding_size).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(new_size, new_size), mode='bicubic', align_corners=False)
    print('reshape position embedding from %d to %d' % (orig_size ** 2, new_size ** 2))
    return torch.cat((extra_tokens, grid.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)
