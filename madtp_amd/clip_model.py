"""Mirror of the reference's CLIP vision path with token pruning (clip/model.py: LayerNorm :158, QuickGELU :167,
ResidualAttentionBlock :174-261, Transformer :264-272, VisionTransformer :275-313, with clip/mock.py's patched
nn.MultiheadAttention): same constructor arguments, parameter names and forward() contracts - a block takes and returns the
5-tuple (x[L,B,C], space_dict, temperature, sd_ft_all, max_keep).

Scope: the VISION tower.  A block built with an attn_mask (the causal text tower)
raises NotImplementedError - the text tower's positional causal mask / EOT read-out depend on the implementation-defined
order of topk(sorted=False) and are tolerance-only in the survey.
"""
from collections import OrderedDict

import torch
from torch import nn

from . import hip
from .runtime import PreparedCache, f32_ptr, compute_dtype, dtype_code, lin_of, prepare_linear, require_gpu, as_f32_contig, to_compute
from .utils import Query_model


class LayerNorm(nn.LayerNorm):
    """clip/model.py:158-165 (fp16-safe LayerNorm; parameters only - the kernels always normalise in f32)."""


class QuickGELU(nn.Module):
    """clip/model.py:167-169: x * sigmoid(1.702 x); fused into the c_fc GEMM epilogue (MADTP_ACT_QUICK_GELU)."""

    def forward(self, x):
        raise NotImplementedError("fused into ResidualAttentionBlock.forward")


class _InProj:
    """adapter: nn.MultiheadAttention's packed in_proj_{weight,bias} viewed as a Linear for lin_of()."""

    def __init__(self, mha):
        self.weight, self.bias = mha.in_proj_weight, mha.in_proj_bias


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, attn_mask: torch.Tensor = None, sd_dim=768):
        super().__init__()
        if d_model // n_head != 64:
            raise ValueError("the gfx950 attention kernels are built for head_dim 64 (CLIP ViT-B/16, ViT-L/14)")
        # parameter container with the reference's names: attn.in_proj_weight/in_proj_bias/out_proj.{weight,bias}
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask
        self.query_model = Query_model(ft_dim=d_model, sd_dim=sd_dim, temperature=1, att_func_type='sparsemax',
                                       pool_type='max', map_func=True)
        self.n_head, self.d_model = n_head, d_model
        self.last_prune = None
        self._cache = PreparedCache()

    def _weights(self):
        params = [self.ln_1.weight, self.ln_1.bias, self.ln_2.weight, self.ln_2.bias, self.attn.in_proj_weight,
                  self.attn.in_proj_bias, self.attn.out_proj.weight, self.attn.out_proj.bias, self.mlp.c_fc.weight,
                  self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias]

        def build():
            lins = [lin_of(self._cache, "qkv", [_InProj(self.attn)]), lin_of(self._cache, "proj", [self.attn.out_proj]),
                    lin_of(self._cache, "fc1", [self.mlp.c_fc]), lin_of(self._cache, "fc2", [self.mlp.c_proj])]
            w = hip.VitBlockW()
            w.ln1_g, w.ln1_b = f32_ptr(self.ln_1.weight, "LayerNorm parameter"), f32_ptr(self.ln_1.bias, "LayerNorm parameter")
            w.ln2_g, w.ln2_b = f32_ptr(self.ln_2.weight, "LayerNorm parameter"), f32_ptr(self.ln_2.bias, "LayerNorm parameter")
            w.eps, w.scale = self.ln_1.eps, (self.d_model // self.n_head) ** -0.5
            w.qkv, w.proj, w.fc1, w.fc2 = [hip.lin_struct(l) for l in lins]
            w.heads, w.dim = self.n_head, self.d_model
            w.dtype = dtype_code()
            w.act = hip.ACT_QUICK_GELU
            return (w, lins)

        return self._cache.get("w", params, build)[0]

    def forward(self, inputs):
        x, space_dict, temperature, sd_ft_all, max_keep = inputs  # x: (N, B, C)   clip/model.py:238
        require_gpu(x, "x")
        if self.attn_mask is not None:
            raise NotImplementedError("causal (text-tower) blocks are out of scope: vision tower only")
        xb = as_f32_contig(x.permute(1, 0, 2))  # (B, N, C); a no-op view when x came from the previous block
        B, N, C = xb.shape
        token_attn = None
        if space_dict is not None:  # :239-245
            token_attn, sd_ft_all, _ = self.query_model(xb[:, 1:, :], space_dict, return_token_att=True, acc_ft=sd_ft_all)
        prune = space_dict is not None and temperature > 0
        w = self._weights()
        x_attn, po = hip.vit_block_attn(w, xb, token_attn, temperature if prune else 0)  # :247 + Reduce_token :196-218
        self.last_prune = None
        k_use, score = 0, None
        if prune:
            score, thr, count, kmax = po
            k = hip.batch_max_count(count)
            self.last_prune = {"k": k, "score": score, "threshold": thr, "count": count, "pruned": False,
                               "indices": None, "indices_sort": None}
            if not (k <= max_keep or (N - 1 - k) <= 1):  # :220-221
                k_use = k
        y, indices, indices_sort = hip.vit_block_mlp(w, x_attn, k_use, score)  # :222-234, :260
        if k_use:
            self.last_prune.update(pruned=True, indices=indices, indices_sort=indices_sort)
        return y.permute(1, 0, 2), space_dict, temperature, sd_ft_all, max_keep


class Transformer(nn.Module):
    """clip/model.py:264-272."""

    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None, sd_dim=768):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask, sd_dim=sd_dim) for _ in range(layers)])

    def forward(self, x: torch.Tensor, space_dict=None, temperature=0, sd_ft_all=None, max_keep=1):
        return self.resblocks((x, space_dict, temperature, sd_ft_all, max_keep))


class VisionTransformer(nn.Module):
    """clip/model.py:275-313."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int, sd_dim=768):
        super().__init__()
        self.input_resolution = input_resolution
        self.output_dim = output_dim
        self.patch_size = patch_size
        self.conv1 = nn.Conv2d(in_channels=3, out_channels=width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, sd_dim=sd_dim)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._cache = PreparedCache()

    def forward(self, x: torch.Tensor, space_dict=None, temperature=0, max_keep=1):
        img = as_f32_contig(require_gpu(x, "image"))
        B = img.shape[0]
        cdt = compute_dtype()
        conv = self._cache.get(("conv", cdt), [self.conv1.weight], lambda: prepare_linear([self.conv1.weight], None, cdt))
        cols = hip.patchify(img, self.patch_size, cdt)
        patches = hip.gemm(cols, conv.w, None, out_dtype=torch.float32, n=conv.n)  # :293-295
        np_ = patches.shape[0] // B
        tok = hip.assemble_tokens(patches, self.class_embedding, self.positional_embedding, B, np_)  # :296-297
        tok, _ = hip.layernorm(tok, self.ln_pre.weight, self.ln_pre.bias, self.ln_pre.eps)  # :298
        xs = tok.permute(1, 0, 2)  # NLD -> LND (a view; the blocks undo it without a copy)
        sd_img_ft_all = None
        if space_dict is not None:
            xs, _, _, sd_img_ft_all, _ = self.transformer(xs, space_dict, temperature, sd_img_ft_all, max_keep)
        else:
            xs = self.transformer(xs)[0]
        cls = xs.permute(1, 0, 2)[:, 0, :].contiguous()  # :307-309
        cls, _ = hip.layernorm(cls, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)
        if self.proj is not None:  # x @ proj  :311-312
            pj = self._cache.get(("proj", cdt), [self.proj], lambda: prepare_linear([self.proj.t()], None, cdt))
            cls = hip.gemm(to_compute(cls), pj.w, None, out_dtype=torch.float32, n=pj.n)
        return cls, sd_img_ft_all
