"""Import-time shims that let the (read-only) MADTP reference at /root/reference be imported
in THIS container for golden-vector generation.  Nothing here ships to the GPU box as product
code and nothing here edits or copies reference files: the shims only provide the third-party
symbols the reference imports that are absent from this image (timm 0.4.12, fairscale, tkinter,
a few transformers 4.15 helpers removed in transformers 5.x, torchvision, ftfy).

Shim list follows SURVEY.md section 8(c).  Call ``install()`` BEFORE importing ``models.*``.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package so sub-modules can hang off it
    sys.modules[name] = m
    return m


class _PatchEmbed(nn.Module):
    """timm==0.4.12 PatchEmbed restated: Conv2d(kernel=stride=patch) -> flatten(2) -> transpose(1,2).
    (third-party dependency absent from /root/reference; call site models/vit.py:241-242,283)"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):  # eval-only harness: identity
        return x


class FakeTokenizer:
    """Stands in for BertTokenizer('./pretrained/bert-base-uncased') (models/blip.py:219-225): the
    harness feeds synthetic id tensors straight through.  ``text`` is a dict-like with input_ids /
    attention_mask tensors."""
    enc_token_id = 30523
    bos_token_id = 30522
    pad_token_id = 0
    sep_token_id = 102
    cls_token_id = 101
    additional_special_tokens_ids = [30523]

    class _Batch(dict):
        def __getattr__(self, k):
            return self[k]

        def to(self, device):
            return FakeTokenizer._Batch({k: v.to(device) for k, v in self.items()})

    # the one string the reference tokenises itself: BLIP_Decoder's prompt (models/blip.py:109,170); bert-base-uncased ids
    STRINGS = {'a picture of ': [101, 1037, 3861, 1997, 102]}

    def __call__(self, text, **kw):
        if isinstance(text, dict):
            return FakeTokenizer._Batch({k: v.clone() for k, v in text.items()})
        if isinstance(text, str) and text in self.STRINGS:
            return FakeTokenizer._Batch({"input_ids": list(self.STRINGS[text])})
        if isinstance(text, (list, tuple)) and all(t in self.STRINGS for t in text):
            ids = torch.tensor([self.STRINGS[t] for t in text], dtype=torch.long)
            return FakeTokenizer._Batch({"input_ids": ids, "attention_mask": torch.ones_like(ids)})
        raise TypeError("FakeTokenizer expects {'input_ids','attention_mask'} tensors")


_installed = False


def install(chdir=True, import_models=True):
    """import_models=False: only the third-party stand-ins (used with madtp_amd.dropin, which must be installed BEFORE the
    first `import models.*`)."""
    global _installed
    if _installed:
        return
    _installed = True
    import transformers  # noqa: F401  (must precede the fakes below)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    # (4) helpers that moved / were removed after transformers 4.15
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", lambda *a, **k: (set(), None))

    # (5) PreTrainedModel conveniences whose semantics changed in 5.x
    PT = mu.PreTrainedModel
    PT.get_head_mask = lambda self, head_mask, num_hidden_layers, is_attention_chunked=False: [None] * num_hidden_layers
    PT.init_weights = lambda self: self.apply(self._init_weights)
    PT.tie_weights = lambda self, *a, **k: None
    if not hasattr(PT, "invert_attention_mask_orig"):
        def invert_attention_mask(self, encoder_attention_mask):
            # transformers 4.15 ModuleUtilsMixin.invert_attention_mask (fp32 branch): (1-m) * -1e4
            m = encoder_attention_mask
            if m.dim() == 3:
                ext = m[:, None, :, :]
            else:
                ext = m[:, None, None, :]
            ext = ext.to(dtype=torch.float32)
            return (1.0 - ext) * -10000.0
        PT.invert_attention_mask = invert_attention_mask

    # (1) timm 0.4.12 surface used by models/vit.py:7-10 and models/blip*.py
    timm = _mod("timm")
    tm = _mod("timm.models")
    vt = _mod("timm.models.vision_transformer")
    vt.PatchEmbed = _PatchEmbed
    vt._cfg = lambda **kw: dict(kw)
    reg = _mod("timm.models.registry")
    reg.register_model = lambda f: f
    lay = _mod("timm.models.layers")
    lay.trunc_normal_ = lambda t, std=1.0, **kw: nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std)
    lay.DropPath = _DropPath
    hlp = _mod("timm.models.helpers")
    hlp.named_apply = lambda *a, **k: None
    hlp.adapt_input_conv = lambda *a, **k: None
    hub = _mod("timm.models.hub")
    hub.download_cached_file = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("offline"))
    timm.models = tm

    # (2) fairscale checkpoint_wrapper -> identity
    _mod("fairscale")
    _mod("fairscale.nn")
    _mod("fairscale.nn.checkpoint")
    ca = _mod("fairscale.nn.checkpoint.checkpoint_activations")
    ca.checkpoint_wrapper = lambda m, *a, **k: m

    # (3) tkinter.messagebox.NO (models/blip_retrieval.py:1)
    if "tkinter" not in sys.modules:
        tk = _mod("tkinter")
        mb = _mod("tkinter.messagebox")
        mb.NO = "no"
        tk.messagebox = mb

    # telnetlib was removed in python 3.13; present in 3.10 - nothing to do.

    # (7) CLIP-only fakes: torchvision.transforms, ftfy
    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        tvt = _mod("torchvision.transforms")
        for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "InterpolationMode"):
            setattr(tvt, n, type(n, (), {"BICUBIC": 3, "__init__": lambda self, *a, **k: None}))
        tv.transforms = tvt
    if "ftfy" not in sys.modules:
        ft = _mod("ftfy")
        ft.fix_text = lambda s: s

    # (7) CLIP only: clip/mock.py imports torch-1.11 internals of nn.functional / nn.modules.activation
    import math
    import typing
    import warnings as _warnings
    import torch.nn.functional as F
    import torch.nn.modules.activation as act
    if not hasattr(F, "_scaled_dot_product_attention"):
        def _scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0):
            """torch==1.11 nn.functional._scaled_dot_product_attention restated (clip/mock.py:220 call site)."""
            B, Nt, E = q.shape
            q = q / math.sqrt(E)
            if attn_mask is not None:
                attn = torch.baddbmm(attn_mask, q, k.transpose(-2, -1))
            else:
                attn = torch.bmm(q, k.transpose(-2, -1))
            attn = F.softmax(attn, dim=-1)
            if dropout_p > 0.0:
                attn = F.dropout(attn, p=dropout_p)
            return torch.bmm(attn, v), attn
        F._scaled_dot_product_attention = _scaled_dot_product_attention
    for name, val in (("math", math), ("warnings", _warnings)):
        if not hasattr(F, name):
            setattr(F, name, val)
    for name, val in (("torch", torch), ("Optional", typing.Optional), ("Tuple", typing.Tuple), ("Tensor", torch.Tensor),
                      ("Parameter", torch.nn.Parameter)):
        if not hasattr(act, name):
            setattr(act, name, val)

    # (9) compress_retrieval_dtp.py imports: ruamel_yaml, the dataset package, fvcore's FLOP counter
    if "ruamel_yaml" not in sys.modules:
        import yaml as _yaml
        ry = _mod("ruamel_yaml")
        ry.load = lambda f, Loader=None: _yaml.safe_load(f)
        ry.Loader = None
        ry.dump = _yaml.safe_dump
    if "data" not in sys.modules:
        dm = _mod("data")
        dm.create_dataset = dm.create_sampler = dm.create_loader = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no datasets offline"))
    if "fvcore" not in sys.modules:
        _mod("fvcore")
        fn = _mod("fvcore.nn")

        class _Flops:  # the evaluation only averages .total(); the traced forward itself is the training branch
            def __init__(self, *a, **k): pass
            def total(self): return 0.0
            def unsupported_ops_warnings(self, *a): pass
            def uncalled_modules_warnings(self, *a): pass
            def tracer_warnings(self, *a): pass
        fn.FlopCountAnalysis = _Flops
        fn.flop_count_str = fn.flop_count_table = lambda *a, **k: ""

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if chdir:
        os.chdir(REFERENCE_ROOT)  # (8) relative 'configs/med_config.json'

    if not import_models:
        return
    # (6) tokenizer: replace by-name copies after the modules are imported
    import models.blip as blip_mod
    blip_mod.init_tokenizer = lambda: FakeTokenizer()


def patch_tokenizer(module):
    """models.blip_nlvr & co. did `from models.blip import init_tokenizer` - rebind the local name."""
    module.init_tokenizer = lambda: FakeTokenizer()


def fast_init():
    """Construct-only tests: make the random initialisers no-ops (nn.Linear / nn.Embedding / nn.LayerNorm defaults, the
    reference's `_init_weights`: `normal_`, `trunc_normal_`, `kaiming_uniform_` ...).  Building BLIP / CLIP at full size on
    8 CPU cores spends minutes in those; the tests that use this only compare state-dict keys and shapes."""
    same = lambda t, *a, **k: t  # noqa: E731
    for name in ("normal_", "uniform_", "trunc_normal_", "kaiming_uniform_", "kaiming_normal_", "xavier_uniform_",
                 "xavier_normal_", "constant_", "zeros_", "ones_", "orthogonal_"):
        if hasattr(torch.nn.init, name):
            setattr(torch.nn.init, name, same)
    for name in ("normal_", "uniform_", "trunc_normal_"):
        if hasattr(torch.Tensor, name):
            setattr(torch.Tensor, name, same)
    torch.randn = lambda *size, **k: torch.zeros(*size, **{kk: v for kk, v in k.items() if kk in ("dtype", "device")})
    import timm.models.layers as tl  # the stand-in module installed by install()
    if hasattr(tl, "trunc_normal_"):
        tl.trunc_normal_ = same


def enable_generate():
    """Make the reference's `text_decoder.generate(...)` call sites (models/blip_vqa.py:134, models/blip.py:177,189) runnable
    under the installed transformers (5.x): (a) since 4.50 `PreTrainedModel` no longer inherits `GenerationMixin`, so it is
    appended to the bases of the reference's own `BertLMHeadModel`; (b) 5.x repeats EVERY tensor in model_kwargs num_beams times,
    4.15's `_expand_inputs_for_generation` only input_ids / attention_mask / token_type_ids (the reference pre-expands
    encoder_hidden_states itself, blip_vqa.py:128) - restated here; (c) FakeTokenizer.decode returns the ids.
    The SEARCH that then runs is the installed library's re-implementation, not 4.15's (see oracle/madtp_oracle.py)."""
    from transformers.generation import GenerationMixin
    import models.med as med
    if GenerationMixin not in med.BertLMHeadModel.__mro__:
        med.BertLMHeadModel.__bases__ = med.BertLMHeadModel.__bases__ + (GenerationMixin,)

    def _expand_415(expand_size=1, is_encoder_decoder=False, input_ids=None, **model_kwargs):
        if expand_size > 1:
            if input_ids is not None:
                input_ids = input_ids.repeat_interleave(expand_size, dim=0)
            for k in ("attention_mask", "token_type_ids"):
                if model_kwargs.get(k) is not None:
                    model_kwargs[k] = model_kwargs[k].repeat_interleave(expand_size, dim=0)
        return input_ids, model_kwargs
    GenerationMixin._expand_inputs_for_generation = staticmethod(_expand_415)
    FakeTokenizer.decode = lambda self, ids, skip_special_tokens=True: [int(t) for t in ids]
