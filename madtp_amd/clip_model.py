"""Mirror of the reference's CLIP with token pruning (clip/model.py: LayerNorm :158, QuickGELU :167, ResidualAttentionBlock
:174-261, Transformer :264-272, VisionTransformer :275-313, CLIP :316-503 evaluation side, build_model :678-716, with
clip/mock.py's patched nn.MultiheadAttention): same constructor arguments, parameter names and forward() contracts - a block
takes and returns the 5-tuple (x[L,B,C], space_dict, temperature, sd_ft_all, max_keep).

Both towers run (BASELINE config 4).  The text tower's blocks carry the causal attn_mask, applied as
attn_mask[:L,:L] to the CURRENT (possibly pruned) sequence exactly as clip/mock.py:309-310 does.  One documented difference:
the reference keeps pruned tokens in torch.topk(sorted=False)'s implementation-defined order, this path in ascending token
order ("replicate a canonical order and document it").  In the vision tower the order is immaterial; in
the text tower the positional mask and the read-out x[b, argmax(text[b])] (:501) see the order, so text features of PRUNED
sequences are pinned by the oracle run with order="ascending" (oracle.clip_encode_text), and the oracle with
order="reference" is pinned to the reference fixture.
"""
from collections import OrderedDict

import torch
from torch import nn

from . import hip
from .runtime import autograd_precision as _autograd_precision, PreparedCache, get_precision, f32_ptr, compute_dtype, dtype_code, lin_of, prepare_linear, require_gpu, as_f32_contig, to_compute
from .utils import DeferredAttFt, Query_model


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
        self._mask_dev = None  # f32 GPU copy of attn_mask handed to the attention kernel
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
            keep = [lins]
            if self.attn_mask is not None:  # :191-192 attention(): the mask follows x's device
                dev = self.ln_1.weight.device
                if self._mask_dev is None or self._mask_dev.device != dev:
                    self._mask_dev = self.attn_mask.detach().to(device=dev, dtype=torch.float32).contiguous()
                w.attn_mask, w.ld_attn_mask = self._mask_dev.data_ptr(), self._mask_dev.stride(0)
                keep.append(self._mask_dev)
            return (w, keep)

        return self._cache.get("w", params, build)[0]

    def forward(self, inputs):
        x, space_dict, temperature, sd_ft_all, max_keep = inputs  # x: (N, B, C)   clip/model.py:238
        require_gpu(x, "x")
        if self.attn_mask is not None and x.shape[0] > self.attn_mask.shape[0]:
            raise ValueError(f"sequence of {x.shape[0]} tokens exceeds the {self.attn_mask.shape[0]}-token attention mask")
        xb = as_f32_contig(x.permute(1, 0, 2))  # (B, N, C); a no-op view when x came from the previous block
        B, N, C = xb.shape
        if torch.is_grad_enabled() and _autograd_precision() and (
                xb.requires_grad or (space_dict is not None and space_dict.requires_grad) or any(p.requires_grad for p in self.parameters())):
            # training use: the block as autograd Functions around the same kernels (madtp_amd/backward.py);
            # sd_ft_all (the running att_ft sum, :241-245) carries a graph as well
            from .backward import QueryModelFunction, block_forward_with_grad
            token_attn = None
            if space_dict is not None:
                if isinstance(sd_ft_all, DeferredAttFt):
                    sd_ft_all = None
                qmap = self.query_model.q_map[0]
                token_attn, sd_ft = QueryModelFunction.apply(self.query_model, xb, space_dict, qmap.weight, qmap.bias)
                sd_ft_all = sd_ft if sd_ft_all is None else sd_ft_all + sd_ft
            prune = space_dict is not None and temperature > 0
            if self.attn_mask is not None:
                self._weights()  # (makes the f32 device copy of the mask the backward reads)
            y = block_forward_with_grad(self, xb, temperature if prune else 0, token_attn, max_keep=int(max_keep))
            return y.permute(1, 0, 2), space_dict, temperature, sd_ft_all, max_keep
        token_attn = None
        if space_dict is not None:  # :239-245
            if isinstance(sd_ft_all, DeferredAttFt):  # Transformer.forward sums the blocks' att_ft in one launch at the end
                token_attn, _, _ = self.query_model(xb[:, 1:, :], space_dict, return_token_att=True, defer=sd_ft_all)
            else:
                token_attn, sd_ft_all, _ = self.query_model(xb[:, 1:, :], space_dict, return_token_att=True, acc_ft=sd_ft_all)
        prune = space_dict is not None and temperature > 0
        w = self._weights()
        # :247 + Reduce_token :196-218 + :222-234, :260 in one library call: the host read of k overlaps the projection GEMM
        # (max_keep is a 0-dim tensor on the text side, :492 - Transformer.forward converts it once)
        y, self.last_prune = hip.vit_block(w, xb, token_attn, temperature if prune else 0, max_keep=int(max_keep))
        return y.permute(1, 0, 2), space_dict, temperature, sd_ft_all, max_keep


class Transformer(nn.Module):
    """clip/model.py:264-272."""

    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None, sd_dim=768):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask, sd_dim=sd_dim) for _ in range(layers)])

    def forward(self, x: torch.Tensor, space_dict=None, temperature=0, sd_ft_all=None, max_keep=1):
        # `sd_ft_all = sd_ft_all + sd_ft` of every block (clip/model.py:243-245) as ONE launch behind the last block when the
        # sum starts here (sd_ft_all is None): the blocks record their (logits, mapped q) pairs in a DeferredAttFt that rides in
        # the tuple's sd_ft_all slot (same values: fast mode one bf16-MFMA launch, parity modes the exact kernel in layer order)
        defer = None
        if space_dict is not None and sd_ft_all is None and len(self.resblocks) > 0 \
                and all(isinstance(b, ResidualAttentionBlock) for b in self.resblocks):
            defer = self.resblocks[0].query_model.deferred()
        if torch.is_tensor(max_keep):
            max_keep = int(max_keep)  # one host read per tower instead of one per block
        if defer is None:
            return self.resblocks((x, space_dict, temperature, sd_ft_all, max_keep))
        x, space_dict, temperature, d, max_keep = self.resblocks((x, space_dict, temperature, defer, max_keep))
        if not isinstance(d, DeferredAttFt):  # the blocks ran under autograd and summed their att_ft themselves
            return x, space_dict, temperature, d, max_keep
        return x, space_dict, temperature, (d.finish() if d.pairs else None), max_keep


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
        if torch.is_grad_enabled() and _autograd_precision() and (
                (space_dict is not None and space_dict.requires_grad) or any(p.requires_grad for p in self.parameters())):
            from .backward import clip_vision_forward_with_grad  # (the tower under autograd)
            return clip_vision_forward_with_grad(self, img, space_dict, temperature, max_keep)
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


class CLIP(nn.Module):
    """clip/model.py:316-503, evaluation side: the towers, embeddings and projections with the reference's parameter names
    (checkpoints load by key; the momentum copies `*_m` and the queues of :395-436 are training state and are not created -
    load_state_dict(strict=False), as build_model :715 does, skips them), encode_image :482-483 and encode_text :485-503."""

    def __init__(self, embed_dim: int, image_resolution: int, vision_layers, vision_width: int, vision_patch_size: int,
                 context_length: int, vocab_size: int, transformer_width: int, transformer_heads: int,
                 transformer_layers: int, evaluate: bool = True, config=None):
        super().__init__()
        self.context_length = context_length
        self.sd_num = 100 if config is None else config['sd_num']
        self.sd_dim = 768 if config is None else config['sd_dim']
        self.space_dict = nn.Parameter(torch.randn(self.sd_num, self.sd_dim))
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet vision towers (clip/model.py:94-155) are off the pruned ViT path")
        self.visual = VisionTransformer(input_resolution=image_resolution, patch_size=vision_patch_size, width=vision_width,
                                        layers=vision_layers, heads=vision_width // 64, output_dim=embed_dim, sd_dim=self.sd_dim)
        self.transformer = Transformer(width=transformer_width, layers=transformer_layers, heads=transformer_heads,
                                       attn_mask=self.build_attention_mask(), sd_dim=self.sd_dim)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592600369327779)  # log(1 / 0.07)
        self.embed_dim = embed_dim
        self.tokenize = None
        self.vision_layers, self.transformer_layers = vision_layers, transformer_layers
        self._cache = PreparedCache()
        if not evaluate:
            self.initialize_parameters()

    def initialize_parameters(self):
        """clip/model.py:438-464."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std = (self.transformer.width ** -0.5) * ((2 * self.transformer.layers) ** -0.5)
        attn_std = self.transformer.width ** -0.5
        fc_std = (2 * self.transformer.width) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)

    def build_attention_mask(self):
        """clip/model.py:466-472: additive causal mask, -inf above the diagonal."""
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image, space_dict=None, temperature=0):
        return self.visual(image, space_dict=space_dict, temperature=temperature)  # :482-483

    def encode_text(self, text, space_dict=None, temperature=0):
        """:485-503.  -> (features [B, embed_dim], sd_txt_ft_all)"""
        require_gpu(text, "text")
        B, L = text.shape
        # token + positional embeddings: an embedding-row gather and an add (the BERT kernel with its LayerNorm does not apply)
        x = (self.token_embedding.weight[text] + self.positional_embedding[:L]).float().contiguous()  # :486-488
        eot = text.argmax(dim=-1)
        max_keep = eot.max() + 2  # :492
        xs = x.permute(1, 0, 2)  # NLD -> LND (a view)
        xs, _, _, sd_txt_ft_all, _ = self.transformer(xs, space_dict, temperature, None, max_keep)  # :493
        xb = as_f32_contig(xs.permute(1, 0, 2))
        rows = xb[torch.arange(B, device=xb.device), eot].contiguous()  # :501 (LayerNorm is row-wise: gather first)
        rows, _ = hip.layernorm(rows, self.ln_final.weight, self.ln_final.bias, self.ln_final.eps)  # :497
        cdt = compute_dtype()
        pj = self._cache.get(("text_projection", cdt), [self.text_projection],
                             lambda: prepare_linear([self.text_projection.t()], None, cdt))
        return hip.gemm(to_compute(rows), pj.w, None, out_dtype=torch.float32, n=pj.n), sd_txt_ft_all


def build_model(state_dict: dict, evaluate: bool = False, config=None):
    """clip/model.py:678-716: geometry inferred from the checkpoint's tensor shapes (ViT towers), strict=False load.  The
    reference casts weights to fp16 (convert_weights :653-675) and back to fp32 (clip/clip.py:148); here they stay fp32."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("ResNet CLIP checkpoints are off the pruned ViT path")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict.keys() if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length, vocab_size,
                 transformer_width, transformer_heads, transformer_layers, evaluate, config)
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    model.load_state_dict(sd, strict=False)
    return model
