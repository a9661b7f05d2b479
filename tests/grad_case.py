"""Shared by the CPU oracle test and the GPU test of the block backward: rebuilds the inputs of a blockgrad_* fixture
(tools/make_golden.py::vit_block_grad_case) from the deterministic generators and the CPU oracle."""
import numpy as np
import torch

from madtp_amd import specs, synth
from oracle import madtp_oracle as O


def grad_sample_index(numel, n=1024, stride=7919):
    return (np.arange(min(n, numel), dtype=np.int64) * stride) % numel


def build(g):
    """g: loaded fixture -> dict(W, prefix, x, token_attn, T, G, layer)."""
    B, size, T, seed, layer = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"]), int(g["layer"])
    W = specs.synth_weights(specs.vit_shapes("", size), seed)
    images = synth.synth_images(B, size, seed)
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed)
    layer_inputs = []
    with torch.no_grad():
        O.vit_forward(W, "", images, space_dict, T, depth=layer + 1, layer_inputs=layer_inputs)
        x = layer_inputs[layer]
        token_attn, _ = O.query_model(x[:, 1:, :], space_dict)
    out_shape = tuple(int(v) for v in g["out_shape"])
    G = torch.from_numpy(synth.uniform_pm1("grad_out", int(np.prod(out_shape)), seed).reshape(out_shape))
    return {"W": W, "prefix": f"blocks.{layer}.", "x": x, "token_attn": token_attn.contiguous(), "T": T, "G": G, "layer": layer}


def check_against_fixture(g, grads, rtol, what):
    """grads: {name: tensor}; compares norm, sum and the sampled entries of every recorded gradient."""
    names = [k[2:-7] for k in g.files if k.startswith("g_") and k.endswith("_sample")]
    assert names, "fixture holds no gradients"
    for name in names:
        t = grads[name].detach().float().cpu().reshape(-1)
        ref = torch.from_numpy(g[f"g_{name}_sample"])
        got = t[torch.from_numpy(grad_sample_index(t.numel(), int(g["nsample"]) if "nsample" in g.files else 1024))]
        scale = max(float(ref.abs().max()), 1e-12)
        if name.endswith("key.bias"):
            # softmax_j(q.(k_j + b)) does not depend on b: the true gradient of a key bias is 0 and both sides hold rounding noise
            # only - measured against the query bias' gradient instead of against itself
            scale = max(scale, float(np.abs(g[f"g_{name[:-8]}query.bias_sample"]).max()))
        err = float((got - ref).abs().max()) / scale
        assert err < rtol, f"{what}: grad {name}: sampled entries differ by {err:.3e} of their maximum (tolerance {rtol})"
        if f"g_{name}_bigsample" in g.files:  # the largest parameters of a model-level fixture: 4096 more entries each
            refb = torch.from_numpy(g[f"g_{name}_bigsample"])
            gotb = t[torch.from_numpy(grad_sample_index(t.numel(), int(g["nsample_big"]), stride=104729))]
            errb = float((gotb - refb).abs().max()) / max(float(refb.abs().max()), scale)
            assert errb < rtol, f"{what}: grad {name}: the 4096-entry sample differs by {errb:.3e} of its maximum (tolerance {rtol})"
        nrm = float(t.double().norm())
        nref = float(g[f"g_{name}_norm"])
        if name.endswith("key.bias"):
            nref_scale = float(g[f"g_{name[:-8]}query.bias_norm"])
            assert abs(nrm - nref) <= rtol * nref_scale, f"{what}: |grad {name}| (noise-level gradient)"
            continue
        assert abs(nrm - nref) <= rtol * max(nref, 1e-12), f"{what}: |grad {name}|"


def permute_G(G, ref_indices, own_indices):
    """The reference keeps tokens in torch.topk(sorted=False)'s order, the HIP path in ascending token order:
    row 1 + p of the block output is token own_indices[b, p] here and token ref_indices[b, p] there.  Returns G re-ordered so that
    every TOKEN receives the upstream gradient it gets in the reference (CLS row and merged-token row stay in place)."""
    out = G.clone()
    B, k = ref_indices.shape
    for b in range(B):
        pos = {int(t): p for p, t in enumerate(ref_indices[b])}
        for p in range(k):
            out[b, 1 + p] = G[b, 1 + pos[int(own_indices[b, p])]]
    return out


def vit_loss_vectors(g):
    """(g, h) of oracle.vit_loss for a encgrad_* fixture (tools/make_golden.py::vit_grad_case)."""
    from madtp_amd import synth as S
    B, seed = int(g["B"]), int(g["seed"])
    a = torch.from_numpy(S.uniform_pm1("vitgrad_a", B * 100 * 768, seed).reshape(B, 100, 768)) if "loss_has_sd_all" in g.files else None
    return (torch.from_numpy(S.uniform_pm1("vitgrad_g", B * 768, seed).reshape(B, 768)),
            torch.from_numpy(S.uniform_pm1("vitgrad_h", B * 768, seed).reshape(B, 768)), a)


def build_med(g):
    """Inputs of a medgrad_* fixture (tools/make_golden.py::med_layer_grad_case) rebuilt from the deterministic generators and the
    CPU oracle: -> dict(W, prefix, hidden, add_mask [B,1,1,L], token_attn, T, g, h, layer)."""
    from madtp_amd import harness
    B, L, T, seed, layer, pad_tail = (int(g["B"]), int(g["L"]), float(g["temperature"]), int(g["seed"]), int(g["layer"]),
                                      int(g["pad_tail"]))
    W = specs.synth_weights(specs.bert_shapes("", "med"), seed)
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed)
    ids = synth.synth_token_ids(B, L, seed + 1)
    att = torch.ones_like(ids)
    if pad_tail:
        for b in range(B):
            att[b, L - (b % (pad_tail + 1)):] = 0
    add_mask = O.extended_mask(att)
    mode = str(g["mode"]) if "mode" in g.files else "text"
    enc = synth.synth_tensor("image_embeds", (B, int(g["Nimg"]), 768), seed).mul(25.0) if mode == "multimodal" else None
    with torch.no_grad():
        hidden = O.bert_embeddings(W, "embeddings.", ids)
        for l in range(layer):
            ta, _ = O.query_model(hidden[:, 1:, :], space_dict)
            hidden, add_mask, _ = O.bert_layer(W, f"encoder.layer.{l}.", hidden, add_mask, T, ta, enc, None, mode, l, "med")
        token_attn, _ = O.query_model(hidden[:, 1:, :], space_dict)
    gv = torch.from_numpy(synth.uniform_pm1("vitgrad_g", B * 768, seed).reshape(B, 768))
    hv = torch.from_numpy(synth.uniform_pm1("vitgrad_h", B * 768, seed).reshape(B, 768))
    return {"W": W, "prefix": f"encoder.layer.{layer}.", "hidden": hidden, "add_mask": add_mask, "token_attn": token_attn.contiguous(),
            "T": T, "g": gv, "h": hv, "layer": layer, "enc": enc, "mode": mode}


def build_med_layer(c):
    """The mirror MED BertLayer `layer` of a build_med() case with the case's weights, on the GPU, parameters requiring grad."""
    from madtp_amd.med import BertConfig, BertModel
    model = BertModel(BertConfig.med_default(), add_pooling_layer=False)
    model.load_state_dict(c["W"], strict=False)
    layer = model.encoder.layer[c["layer"]].cuda().eval()
    for p in layer.parameters():
        p.requires_grad_(True)
        p.grad = None
    return layer


def build_nlvr(g):
    """Inputs of an nlvrgrad_* fixture (tools/make_golden.py::nlvr_layer_grad_case) rebuilt from the deterministic generators and
    the CPU oracle: -> dict(W, prefix, hidden, add_mask, token_attn, T, g, h, layer, enc [2 tensors], enc_mask [2 x [B,1,1,Nk]])."""
    B, L, T, seed, layer, pad_tail, Nimg = (int(g["B"]), int(g["L"]), float(g["temperature"]), int(g["seed"]), int(g["layer"]),
                                            int(g["pad_tail"]), int(g["Nimg"]))
    W = specs.synth_weights(specs.bert_shapes("", "nlvr"), seed)
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed)
    ids = synth.synth_token_ids(B, L, seed + 1)
    att = torch.ones_like(ids)
    if pad_tail:
        for b in range(B):
            att[b, L - (b % (pad_tail + 1)):] = 0
    add_mask = O.extended_mask(att)
    enc = [synth.synth_tensor(f"image_embeds{i}", (B, Nimg, 768), seed).mul(25.0) for i in range(2)]
    enc_mask = [torch.zeros(B, 1, 1, Nimg) for _ in range(2)]
    with torch.no_grad():
        hidden = O.bert_embeddings(W, "embeddings.", ids)
        for l in range(layer):
            ta, _ = O.query_model(hidden[:, 1:, :], space_dict)
            hidden, add_mask, _ = O.bert_layer(W, f"encoder.layer.{l}.", hidden, add_mask, T, ta, enc, enc_mask, "multimodal", l, "nlvr")
        token_attn, _ = O.query_model(hidden[:, 1:, :], space_dict)
    gv = torch.from_numpy(synth.uniform_pm1("vitgrad_g", B * 768, seed).reshape(B, 768))
    hv = torch.from_numpy(synth.uniform_pm1("vitgrad_h", B * 768, seed).reshape(B, 768))
    return {"W": W, "prefix": f"encoder.layer.{layer}.", "hidden": hidden, "add_mask": add_mask, "token_attn": token_attn.contiguous(),
            "T": T, "g": gv, "h": hv, "layer": layer, "enc": enc, "enc_mask": enc_mask, "mode": "multimodal"}


def build_nlvr_layer(c):
    """The mirror NLVR BertLayer of a build_nlvr() case with the case's weights, on the GPU, parameters requiring grad."""
    from madtp_amd.nlvr_encoder import BertConfig, BertModel
    model = BertModel(BertConfig.med_default(), add_pooling_layer=False)
    model.load_state_dict(c["W"], strict=False)
    layer = model.encoder.layer[c["layer"]].cuda().eval()
    for p in layer.parameters():
        p.requires_grad_(True)
        p.grad = None
    return layer


def build_decoder(g):
    """Inputs of a decgrad_* fixture (tools/make_golden.py::decoder_grad_case): -> dict(W, ids, att, labels, enc, enc_att, w)."""
    B, L, Nq, seed, pad_tail = int(g["B"]), int(g["L"]), int(g["Nq"]), int(g["seed"]), int(g["pad_tail"])
    W = specs.tie_keys(specs.synth_weights(specs.lm_head_shapes(""), seed))
    ids = synth.synth_token_ids(B, L, seed + 3)
    att = torch.ones_like(ids)
    for b in range(B):
        att[b, L - (b % (pad_tail + 1)):] = 0
    return {"W": W, "ids": ids, "att": att, "labels": ids.masked_fill(att == 0, -100),
            "enc": synth.synth_tensor("question_states", (B, Nq, 768), seed), "enc_att": torch.ones(B, Nq, dtype=torch.long),
            "w": torch.from_numpy(g["weights"])}


def build_vqa_train(g):
    """Inputs of a trainstep_vqa_* fixture (tools/make_golden.py::vqa_train_case)."""
    from madtp_amd import harness
    B, size, L, seed = int(g["B"]), int(g["size"]), int(g["L"]), int(g["seed"])
    n_list = [int(v) for v in g["n_list"]]
    a_ids, a_att = synth.synth_answer_ids(sum(n_list), int(g["answer_len"]), seed)
    return {"W": specs.tie_keys(specs.synth_weights(specs.blip_vqa_shapes(size, decoder=True), seed)),
            "images": synth.synth_images(B, size, seed), "ids": synth.synth_token_ids(B, L, seed),
            "att": harness.padded_mask(B, L, int(g["pad_tail"])), "a_ids": a_ids, "a_att": a_att, "n_list": n_list,
            "weights": torch.from_numpy(g["weights"]), "T": float(g["temperature"])}


def build_clip_block(g):
    """Inputs of a clipgrad_* fixture (tools/make_golden.py::clip_block_grad_case): the block input x [B,N,C] (batch first) from the
    CPU oracle's forward of the preceding blocks."""
    B, size, T, seed, layer = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"]), int(g["layer"])
    W = specs.synth_weights(specs.clip_vit_shapes("", size), seed)
    images = synth.synth_images(B, size, seed)
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed)
    import torch.nn.functional as F
    with torch.no_grad():  # clip/model.py:292-303 up to block `layer`
        x = F.conv2d(images, W["conv1.weight"], None, stride=16)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        x = torch.cat([W["class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1]), x], dim=1) + W["positional_embedding"]
        x = O.layer_norm(x, W["ln_pre.weight"], W["ln_pre.bias"], 1e-5)
        for i in range(layer):
            x, _, _ = O.clip_block(W, f"transformer.resblocks.{i}.", x, space_dict, T, int(g["max_keep"]))
    return {"W": W, "images": images, "space_dict": space_dict, "T": T, "layer": layer, "x": x, "max_keep": int(g["max_keep"]),
            "prefix": f"transformer.resblocks.{layer}.",
            "g": torch.from_numpy(synth.uniform_pm1("vitgrad_g", B * 768, seed).reshape(B, 768)),
            "h": torch.from_numpy(synth.uniform_pm1("vitgrad_h", B * 768, seed).reshape(B, 768)),
            "a": torch.from_numpy(synth.uniform_pm1("vitgrad_a", B * 100 * 768, seed).reshape(B, 100, 768))}
