"""Checkpoint loading with the reference's semantics (host-side, once per model):
  * models/blip.py:254-278 `load_checkpoint`: checkpoint['model'], position embeddings of the vision tower(s) interpolated to
    the model's patch grid, keys whose shape differs from the model's dropped, load_state_dict(strict=False);
  * models/blip_nlvr.py:130-159: additionally every `crossattention.self.*` key is duplicated to `self0` / `self1` and every
    `crossattention.output.dense.*` key to `dense0` / `dense1` (the NLVR twin branches start from the single-branch weights).
There is no network in the deployment image: URLs are rejected (the reference downloads them with timm's hub helper)."""
import os
from urllib.parse import urlparse

import torch

from .vit import interpolate_pos_embed


def is_url(url_or_filename):
    return urlparse(str(url_or_filename)).scheme in ("http", "https")


def _read(url_or_filename):
    if isinstance(url_or_filename, dict):
        return url_or_filename
    if is_url(url_or_filename):
        raise RuntimeError("checkpoint URLs need network access; download the file and pass its path")
    if os.path.isfile(url_or_filename):
        return torch.load(url_or_filename, map_location="cpu")
    raise RuntimeError("checkpoint url or path is invalid")


def load_checkpoint(model, url_or_filename):
    """models/blip.py:254-278.  `url_or_filename` may also be an already-loaded checkpoint dict.  -> (model, msg)"""
    checkpoint = _read(url_or_filename)
    state_dict = dict(checkpoint["model"])
    state_dict["visual_encoder.pos_embed"] = interpolate_pos_embed(state_dict["visual_encoder.pos_embed"], model.visual_encoder)
    own = model.state_dict()
    if "visual_encoder_m.pos_embed" in own and "visual_encoder_m.pos_embed" in state_dict:
        state_dict["visual_encoder_m.pos_embed"] = interpolate_pos_embed(state_dict["visual_encoder_m.pos_embed"],
                                                                         model.visual_encoder_m)
    for key in own.keys():
        if key in state_dict and state_dict[key].shape != own[key].shape:
            del state_dict[key]
    msg = model.load_state_dict(state_dict, strict=False)
    return model, msg


def load_checkpoint_nlvr(model, url_or_filename):
    """models/blip_nlvr.py:130-159 (twin cross-attention key duplication; no shape filtering there).  -> (model, msg)"""
    checkpoint = _read(url_or_filename)
    state_dict = dict(checkpoint["model"])
    state_dict["visual_encoder.pos_embed"] = interpolate_pos_embed(state_dict["visual_encoder.pos_embed"], model.visual_encoder)
    for key in list(state_dict.keys()):
        if "crossattention.self." in key:
            state_dict[key.replace("self", "self0")] = state_dict[key]
            state_dict[key.replace("self", "self1")] = state_dict[key]
        elif "crossattention.output.dense." in key:
            state_dict[key.replace("dense", "dense0")] = state_dict[key]
            state_dict[key.replace("dense", "dense1")] = state_dict[key]
    msg = model.load_state_dict(state_dict, strict=False)
    return model, msg
