"""Drop-in installation ("drops into the compress_*_dtp.py scripts unchanged").

    import madtp_amd.dropin as dropin
    dropin.install("/path/to/MADTP")        # before the first `import models.*`
    from models.blip_nlvr import blip_nlvr   # the REFERENCE's own, unmodified file ...
    model = blip_nlvr(...)                   # ... now built from the MI355X mirrors of vit / med / nlvr_encoder / utils

`install()` registers the mirrors under the reference's module names in sys.modules - `models.vit`, `models.med`,
`models.nlvr_encoder`, `models.utils` - and leaves every other `models.*` module (blip.py, blip_nlvr.py, blip_retrieval.py,
blip_vqa.py: pure glue) to be imported from the reference tree itself.  The CLIP driver (compress_retrieval_clip_dtp.py:21,
262) goes `from clip import clip; clip.load(...)`: `clip.model` is registered as the mirror (madtp_amd.clip_model: `build_model`,
`CLIP`) and `clip.mock` - the reference's import-time monkey-patch of torch.nn.MultiheadAttention
(clip/mock.py:354-359), which the mirror's blocks do not use - as an empty module, while `clip.clip` (load / tokenize /
_transform: host glue) and `clip.simple_tokenizer` are imported from the reference tree itself.  Names those glue files import that are off the
pruned forward path (the contrastive-loss helpers of models/utils.py) resolve to stubs that raise on use, so a script that needs
them (retrieval / CLIP training: momentum encoders, queues) fails loudly instead of silently running something else;
`models.med.BertLMHeadModel` is the decoder mirror (teacher-forced with or without labels, rank_answer, beam-search `generate`;
nucleus sampling raises).  Since round 4 the mirrors build an autograd graph in the fp32 precision mode when grad mode is on
(madtp_amd/backward.py), so the NLVR / VQA / caption glue files' `forward(train=True)` - whose loss heads are plain torch code in
the reference's own files - can be back-propagated through; the mirrors have no dropout / DropPath."""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.modules.loss import _Loss

_MIRRORS = ("vit", "med", "nlvr_encoder", "utils")


def _off_path(name, what):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError(f"models.{what}.{name} is off the pruned forward path (training / decoding); "
                                      "madtp_amd provides the evaluation forward only")
    _Stub.__name__ = _Stub.__qualname__ = name
    return _Stub


def accuracy(output, target, topk=(1,)):
    """models/utils.py:321-335 (metric helper used by the drivers)."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


def _module(name, base, extra):
    m = types.ModuleType(name)
    m.__dict__.update({k: v for k, v in vars(base).items() if not k.startswith("__")})
    m.__dict__.update(extra)
    m.__doc__ = f"madtp_amd drop-in for the reference's {name} (mirror: {base.__name__})"
    m.__madtp_mirror__ = base
    return m


def install(reference_root=None):
    """Registers the mirrors as models.{vit,med,nlvr_encoder,utils}.  reference_root: directory that contains the reference's
    `models/` package (needed unless `models` is already importable); it is put on sys.path like the drivers expect."""
    if reference_root is not None:
        reference_root = os.path.abspath(reference_root)
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)
    for n in _MIRRORS:
        full = f"models.{n}"
        if full in sys.modules and not hasattr(sys.modules[full], "__madtp_mirror__"):
            raise RuntimeError(f"{full} is already imported from {getattr(sys.modules[full], '__file__', '?')}: "
                               "call madtp_amd.dropin.install() before the first `import models.*`")
    pkg = sys.modules.get("models")
    if pkg is None:
        pkg = types.ModuleType("models")
        roots = [os.path.join(p, "models") for p in ([reference_root] if reference_root else sys.path)
                 if p and os.path.isdir(os.path.join(p, "models"))]
        if not roots:
            raise RuntimeError("cannot find the reference's `models/` package: pass reference_root")
        pkg.__path__ = roots[:1]  # blip.py, blip_nlvr.py, ... are still imported from the reference tree
        pkg.__package__ = "models"
        sys.modules["models"] = pkg
    extras = {
        "vit": {},
        "med": {},  # (BertLMHeadModel: the teacher-forced decoder mirror of madtp_amd.bert)
        "nlvr_encoder": {},
        # `from models.utils import *` in the reference also leaks that module's own imports; keep the same names available
        "utils": {"torch": torch, "nn": nn, "np": np, "math": math, "F": F, "_Loss": _Loss,
                  "device": torch.device("cuda" if torch.cuda.is_available() else "cpu"), "accuracy": accuracy,
                  **{n: _off_path(n, "utils") for n in ("Sparsemax", "AllGather", "ClipInfoCELoss", "NT_Xent", "NT_Xent_gather")}},
    }
    try:
        import einops
        extras["utils"]["einops"] = einops
    except ImportError:
        pass
    for n in _MIRRORS:
        mod = _module(f"models.{n}", importlib.import_module(f"madtp_amd.{n}"), extras[n])
        sys.modules[f"models.{n}"] = mod
        setattr(pkg, n, mod)
    _install_clip(reference_root)
    return pkg


def _install_clip(reference_root):
    """`clip` as a package whose `model` / `mock` sub-modules are ours and whose other files come from the reference tree
    (clip/__init__.py itself is `from .clip import *; from . import mock`: the first half is done lazily by __getattr__ so
    that installing does not import the tokenizer / torchvision glue).  Skipped when the tree has no clip/ directory."""
    for full in ("clip", "clip.model", "clip.mock"):
        if full in sys.modules and not hasattr(sys.modules[full], "__madtp_mirror__"):
            raise RuntimeError(f"{full} is already imported from {getattr(sys.modules[full], '__file__', '?')}: "
                               "call madtp_amd.dropin.install() before the first `import clip`")
    roots = [os.path.join(p, "clip") for p in ([reference_root] if reference_root else sys.path)
             if p and os.path.isfile(os.path.join(p, "clip", "clip.py"))]
    if not roots:
        return None
    from . import clip_model
    pkg = types.ModuleType("clip")
    pkg.__path__ = roots[:1]
    pkg.__package__ = "clip"
    pkg.__madtp_mirror__ = clip_model
    model = _module("clip.model", clip_model, {})
    mock = types.ModuleType("clip.mock")
    mock.__doc__ = "madtp_amd drop-in: the reference's MultiheadAttention monkey-patch is not needed by the mirror (no-op)"
    mock.__madtp_mirror__ = clip_model

    def _lazy(name):  # clip.load / clip.tokenize / clip.available_models re-exported as clip/__init__.py does
        if name.startswith("__"):
            raise AttributeError(name)
        sub = importlib.import_module("clip.clip")
        if name == "clip":
            return sub
        if name in getattr(sub, "__all__", ()):
            return getattr(sub, name)
        raise AttributeError(f"module 'clip' has no attribute {name!r}")
    pkg.__getattr__ = _lazy
    pkg.model, pkg.mock = model, mock
    sys.modules["clip"], sys.modules["clip.model"], sys.modules["clip.mock"] = pkg, model, mock
    return pkg


def installed():
    return all(hasattr(sys.modules.get(f"models.{n}"), "__madtp_mirror__") for n in _MIRRORS)
