"""TEST INFRASTRUCTURE (like the rest of oracle/): the CPU-oracle counterparts of madtp_amd/workloads.py - one forward of each
BASELINE config through oracle/madtp_oracle.py on the same synthetic inputs, returning the outputs and the per-encoder token
counts.  Used by tools/calibrate_temperature.py (p -> temperature bisection), bench.py's cpu_baseline leg and the tests."""
import torch
import torch.nn.functional as F

from madtp_amd import harness, specs, synth
from oracle import madtp_oracle as O


def _lens(trace, n0):
    return harness.token_lengths(trace, n0)


def weights(name, size):
    shapes = {"nlvr": specs.blip_nlvr_shapes, "retrieval": specs.blip_retrieval_shapes, "vqa": specs.blip_vqa_shapes,
              "clip": specs.clip_shapes}[name](size)
    return specs.synth_weights(shapes, 0)


@torch.no_grad()
def forward(name, W, B, T, seed=0, size=None):
    """-> (outputs, lens dict as madtp_amd.workloads.<W>.lens returns)"""
    if name == "nlvr":
        size = size or 224
        images, ids = synth.synth_images(2 * B, size, seed), synth.synth_token_ids(B, 20, seed)
        tr = {}
        out = O.blip_nlvr_forward(W, images, ids, torch.ones_like(ids), T, trace=tr)
        n0 = (size // 16) ** 2 + 1
        return out, {"vit": _lens(tr["vit"], n0), "text": _lens(tr["text"], 20)}
    if name == "retrieval":
        size = size or 224
        batches, ids, att = harness.retrieval_inputs(B, B, B, size, 35, seed)
        sd = W["space_dict"]
        vt, tt, mt = [], [], []
        img, _ = O.vit_forward(W, "visual_encoder.", batches[0], sd, T, trace=vt)
        img_emb = F.normalize(F.linear(img[:, 0, :], W["vision_proj.weight"], W["vision_proj.bias"]), dim=-1)
        hid, _, _ = O.bert_model(W, "text_encoder.", ids, att, sd, T, mode="text", variant="med", trace=tt)
        txt_emb = F.normalize(F.linear(hid[:, 0, :], W["text_proj.weight"], W["text_proj.bias"]))
        ids_mm = ids.clone()
        ids_mm[:, 0] = O.ENC_TOKEN_ID
        atts = torch.ones(img.shape[:-1], dtype=torch.long)
        mm, _, _ = O.bert_model(W, "text_encoder.", ids_mm, att, sd, T, enc=img, enc_atts=atts, mode="multimodal", variant="med",
                                trace=mt)
        itm = F.linear(mm[:, 0, :], W["itm_head.weight"], W["itm_head.bias"])[:, 1]
        n0 = (size // 16) ** 2 + 1
        return (itm, (img_emb * txt_emb).sum(-1)), {"vit": _lens(vt, n0), "text": _lens(tt, 35), "mm": _lens(mt, 35)}
    if name == "vqa":
        size = size or 480
        images, ids = synth.synth_images(B, size, seed), synth.synth_token_ids(B, 20, seed)
        tr = {}
        out = O.blip_vqa_encoder_forward(W, images, ids, torch.ones_like(ids), T, trace=tr)
        return out, {"vit": _lens(tr["vit"], (size // 16) ** 2 + 1), "mm": _lens(tr["text"], 20)}
    if name == "clip":
        size = size or 224
        images, text = synth.synth_images(B, size, seed), synth.synth_clip_tokens(B, 77, seed)
        vt, tt = [], []
        fi, _ = O.clip_encode_image(W, images, W["space_dict"], T, trace=vt)
        ft, _ = O.clip_encode_text(W, text, W["space_dict"], T, order="ascending", trace=tt)
        sims = F.normalize(fi, dim=-1) @ F.normalize(ft, dim=-1).t()
        return sims, {"vit": _lens(vt, (size // 16) ** 2 + 1), "text": _lens(tt, 77)}
    raise KeyError(name)
