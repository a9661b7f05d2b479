"""Pins the CPU oracle (oracle/madtp_oracle.py) to fixtures recorded from the REFERENCE itself
(tools/make_golden.py imported /root/reference in the build container).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from madtp_amd import specs, synth
from oracle import madtp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nlvr_*.npz")))


def test_synth_generator_is_bit_stable():
    # known-answer: the generator must give identical bits on every box (weights are regenerated, never shipped)
    h = synth.hash_u32("visual_encoder.blocks.0.attn.qkv.weight", 4, 0)
    assert h.dtype == np.uint32
    t = synth.synth_tensor("visual_encoder.blocks.0.attn.qkv.weight", (2, 3), 0)
    t2 = synth.synth_tensor("visual_encoder.blocks.0.attn.qkv.weight", (2, 3), 0)
    assert torch.equal(t, t2)
    assert abs(float(synth.synth_tensor("x.weight", (512, 512), 1).std()) - 0.02) < 5e-4
    ids = synth.synth_token_ids(3, 20, 0)
    assert ids.min() >= 1000 and ids.max() < 30000


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_oracle_matches_reference_fixture(path):
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    shapes = specs.blip_nlvr_shapes(size)
    assert sorted(shapes.keys()) == [str(k) for k in g["state_dict_keys"]], "state_dict key names are an API"
    W = specs.synth_weights(shapes, seed)
    images = synth.synth_images(2 * B, size, seed)
    ids = synth.synth_token_ids(B, L, seed)
    from madtp_amd import harness
    att = harness.padded_mask(B, L, int(g["pad_tail"]) if "pad_tail" in g.files else 0)  # padded captions: mask compaction
    trace = {}
    with torch.no_grad():
        logits = O.blip_nlvr_forward(W, images, ids, att, T, trace=trace)
    # same torch ops on the same box => exact; 1e-5 leaves room for a different CPU/BLAS on the GPU box's host
    assert np.abs(logits.numpy() - g["logits"]).max() < 1e-5
    assert np.abs(trace["image_embeds"][:, 0, :16].numpy() - g["img_embeds_cls"]).max() < 1e-4
    for side, key in (("vit", "vit"), ("text", "txt")):
        lens = g[f"{key}_lens"]
        for l, info in enumerate(trace[side]):
            if f"{key}{l}_idx" not in g.files:
                assert not info["pruned"]
                continue
            assert info["pruned"] and info["k"] + 2 == lens[l]
            # bit-exact kept-token index SETS (order of topk(sorted=False) is implementation-defined)
            assert (np.sort(info["indices"].numpy(), 1) == np.sort(g[f"{key}{l}_idx"], 1)).all()
            assert (info["indices_sort"].numpy() == g[f"{key}{l}_sort"]).all()


MED_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "med_*.npz")))


def med_inputs(g):
    """inputs of tools/make_golden.py::med_case, rebuilt from the fixture's scalars."""
    B, L, Nimg, seed, pad_tail = int(g["B"]), int(g["L"]), int(g["Nimg"]), int(g["seed"]), int(g["pad_tail"])
    mode = str(g["mode"])
    ids = synth.synth_token_ids(B, L, seed + 1)
    att = torch.ones_like(ids)
    if pad_tail:
        for b in range(B):
            att[b, L - (b % (pad_tail + 1)):] = 0
    enc = synth.synth_tensor("image_embeds", (B, Nimg, 768), seed).mul(25.0) if mode == "multimodal" else None
    enc_att = torch.ones(B, Nimg, dtype=torch.long) if enc is not None else None
    sd = synth.synth_tensor("space_dict", (100, 768), seed)
    return ids, att, enc, enc_att, sd, mode, float(g["temperature"])


@pytest.mark.parametrize("path", MED_CASES, ids=[os.path.basename(c)[:-4] for c in MED_CASES])
def test_oracle_med_matches_reference_fixture(path):
    """models/med.py BertModel (text / multimodal mode, padded attention masks) - SURVEY.md 8 rows a9-a11."""
    g = np.load(path)
    ids, att, enc, enc_att, sd, mode, T = med_inputs(g)
    W = specs.synth_weights(specs.bert_shapes("", "med"), int(g["seed"]))
    trace = []
    with torch.no_grad():
        hidden, _, _ = O.bert_model(W, "", ids, att, sd, T, enc=enc, enc_atts=enc_att, mode=mode, variant="med", trace=trace)
    assert list(hidden.shape) == g["hidden_shape"].tolist()
    assert np.abs(hidden[:, 0, :32].numpy() - g["hidden_cls"]).max() < 1e-4
    lens = g["txt_lens"]
    for l, info in enumerate(trace):
        if f"txt{l}_idx" not in g.files:
            assert not info["pruned"]
            continue
        assert info["pruned"] and info["k"] + 2 == lens[l]
        # the reference gathers topk(k+1) and keeps the first k (med.py:377-378)
        assert (np.sort(info["indices"].numpy(), 1) == np.sort(g[f"txt{l}_idx"][:, : info["k"]], 1)).all()
        assert (info["indices_sort"].numpy() == g[f"txt{l}_sort"]).all()


CLIP_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clip_vit_*.npz")))


@pytest.mark.parametrize("path", CLIP_CASES, ids=[os.path.basename(c)[:-4] for c in CLIP_CASES])
def test_oracle_clip_vision_matches_reference_fixture(path):
    """clip/model.py VisionTransformer + ResidualAttentionBlock (vision tower)."""
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    shapes = specs.clip_vit_shapes("", size)
    assert sorted(shapes.keys()) == [str(k) for k in g["state_dict_keys"]]
    W = specs.synth_weights(shapes, seed)
    trace = []
    with torch.no_grad():
        feat, sd_ft = O.clip_vision_forward(W, "", synth.synth_images(B, size, seed),
                                            synth.synth_tensor("space_dict", (100, 768), seed), T, 1, trace=trace)
    assert np.abs(feat.numpy() - g["features"]).max() < 1e-4
    assert np.abs(sd_ft[:, :4, :16].numpy() - g["sd_ft_head"]).max() < 1e-3
    lens = g["vit_lens"]
    for l, info in enumerate(trace):
        if f"vit{l}_idx" not in g.files:
            assert not info["pruned"]
            continue
        assert info["pruned"] and info["k"] + 2 == lens[l]
        assert (np.sort(info["indices"].numpy(), 1) == np.sort(g[f"vit{l}_idx"], 1)).all()


VIT_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vit*_b*.npz")))


@pytest.mark.parametrize("path", VIT_CASES, ids=[os.path.basename(c)[:-4] for c in VIT_CASES])
def test_oracle_vit_large_image_matches_reference_fixture(path):
    """models/vit.py VisionTransformer at 384^2 (577 tokens) / 480^2 (901 tokens, the VQA configuration)."""
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    W = specs.synth_weights(specs.vit_shapes("", size), seed)
    trace = []
    with torch.no_grad():
        out, sd_ft = O.vit_forward(W, "", synth.synth_images(B, size, seed), synth.synth_tensor("space_dict", (100, 768), seed),
                                   T, trace=trace)
    assert list(out.shape) == g["out_shape"].tolist()
    assert np.abs(out[:, 0, :32].numpy() - g["cls"]).max() < 1e-4
    for l, info in enumerate(trace):
        if f"vit{l}_idx" in g.files:
            assert (np.sort(info["indices"].numpy(), 1) == np.sort(g[f"vit{l}_idx"], 1)).all()
        else:
            assert not info["pruned"]


RETR_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "retr_*.npz")))


@pytest.mark.parametrize("path", RETR_CASES, ids=[os.path.basename(c)[:-4] for c in RETR_CASES])
def test_oracle_retrieval_matches_reference_fixture(path):
    """ITM re-ranking evaluation: the fixture was produced by the reference's OWN evaluate()
    (compress_retrieval_dtp.py:84-207) on the synthetic set; the oracle restatement must give the same score matrices
    (same candidates re-ranked, same scores) including the cross-batch CLS-repeat padding of the image tokens."""
    from madtp_amd import harness
    g = np.load(path)
    n_img, img_bs, n_txt, size = int(g["n_img"]), int(g["img_bs"]), int(g["n_txt"]), int(g["size"])
    T, k_test, seed = float(g["temperature"]), int(g["k_test"]), int(g["seed"])
    shapes = specs.blip_retrieval_shapes(size)
    assert sorted(shapes.keys()) == g["state_dict_keys"].tolist()
    W = specs.synth_weights(shapes, seed)
    batches, ids, att = harness.retrieval_inputs(n_img, img_bs, n_txt, size, 35, seed)
    ex = {}
    with torch.no_grad():
        i2t, t2i = O.retrieval_evaluate(W, batches, ids, att, T, k_test, extras=ex)
    assert ex["image_feats"].shape[1] == int(g["vit_out_lens"].max())
    for ours, ref in ((i2t.numpy(), g["score_i2t"]), (t2i.numpy(), g["score_t2i"])):
        assert ((ours == -100.0) == (ref == -100.0)).all()          # the same candidates were re-ranked
        assert np.abs(ours - ref).max() < 1e-5


def test_retrieval_mirror_state_dict_keys():
    """The BLIP_Retrieval mirror exposes every evaluation-path key of the reference's state dict (checked against the key
    list the fixture recorded from the reference model), so reference checkpoints load by name (strict=False drops the
    momentum / queue entries)."""
    from madtp_amd.blip_retrieval import BLIP_Retrieval
    ref_keys = set(np.load(RETR_CASES[0])["state_dict_keys"].tolist())
    mine = set(BLIP_Retrieval(image_size=224, evaluate=True).state_dict().keys())
    assert ref_keys <= mine, sorted(ref_keys - mine)[:5]
    assert all("query_model" in k for k in mine - ref_keys), sorted(mine - ref_keys)[:5]


def test_controller_matches_reference_rule_and_constant():
    """temperature step rule (compress_nlvr_dtp.py:175-200) and the analytic GFLOPs counter pinned on
    the reference's own unpruned figure Ori_Gflops = 132.54 (:162, fvcore)."""
    from madtp_amd import controller as C
    for cur, step in ((200.0, 1.0), (80.0, 0.5), (73.0, 0.25), (68.0, 0.1), (66.5, 0.01)):
        assert abs(C.step_temperature(3.0, cur, 66.27) - (3.0 + step)) < 1e-12
        assert abs(C.step_temperature(3.0, 2 * 66.27 - cur, 66.27) - (3.0 - step)) < 1e-12
    assert abs(C.step_temperature(3.0, 66.27, 66.27) - 2.99) < 1e-12          # not greater -> the `else` branch, smallest step
    full = C.nlvr_gflops([577] * 12, [20] * 12, 384, 20)
    assert abs(full - C.ORI_GFLOPS["nlvr"]) / C.ORI_GFLOPS["nlvr"] < 5e-3, full
    # a toy plant: GFLOPs fall with T; the controller settles within one small step of the target
    log = C.run_controller(lambda T: 132.54 / (1.0 + 0.2 * T), 0.0, 0.5, 132.54, 60)
    assert abs(log[-1][2] - 66.27) < 1.5


def test_controller_ladders_of_every_driver():
    """Each compress_*_dtp.py driver has its own step ladder (thresholds on |Cur - Target| -> step) and its own
    calculate_temperature() search; spot values below are read off the reference sources (file:line in controller.py)."""
    from madtp_amd import controller as C
    tgt = 100.0
    cases = {  # task -> [(Cur - Target, expected step)]
        "retrieval": [(60, 0.5), (40, 0.3), (25, 0.2), (15, 0.1), (7, 0.05), (3, 0.02), (1, 0.01)],      # compress_retrieval_dtp.py:402-434
        "retrieval_clip": [(60, 0.5), (40, 0.3), (25, 0.2), (15, 0.1), (7, 0.05), (3, 0.02), (1, 0.01)],  # compress_retrieval_clip_dtp.py:300-332
        "caption": [(60, 0.5), (40, 0.3), (25, 0.2), (15, 0.1), (7, 0.05), (3, 0.02), (1, 0.01)],         # compress_caption_dtp.py:235-267
        "vqa": [(60, 0.25), (40, 0.15), (15, 0.1), (7, 0.05), (3, 0.01), (1, 0.0)],                       # compress_vqa_dtp.py:245-268 (no else)
        "nlvr": [(40, 1.0), (15, 0.5), (7, 0.25), (2, 0.1), (0.5, 0.01)],                                 # compress_nlvr_dtp.py:175-200
    }
    for task, rows in cases.items():
        for d, step in rows:
            assert abs(C.step_temperature(2.0, tgt + d, tgt, task) - (2.0 + step)) < 1e-12, (task, d)
            assert abs(C.step_temperature(2.0, tgt - d, tgt, task) - (2.0 - step)) < 1e-12, (task, d)
    assert C.ORI_GFLOPS["retrieval"] == 153.2 and C.ORI_GFLOPS["vqa"] == 186.1 and C.ORI_GFLOPS["retrieval_clip"] == 395.7
    # calculate_temperature on a toy plant (GFLOPs fall with T): start value, first step and the stopping tolerance per driver
    plant = lambda ori: (lambda T: ori / (1.0 + 0.15 * max(T, 0.0)))
    for task, start, first, tol in (("retrieval", 0.0, 0.5, 10), ("caption", 1.0, 0.3, 10), ("retrieval_clip", 1.0, 0.5, 5),
                                    ("vqa", 0.0, 0.5, 10)):
        ori = C.ORI_GFLOPS[task]
        seen = []
        m = plant(ori)
        cur, T = C.calculate_temperature(lambda t: (seen.append(t), m(t))[1], ori, ori * 0.5, task)
        assert abs(seen[0] - (start + first)) < 1e-12, (task, seen[0])     # Cur - Target = ori / 2 picks the driver's first rung
        assert abs(cur - ori * 0.5) <= tol + 1e-9 and cur == m(T), (task, cur, T)
    # retrieval: |Cur - Target| = 76.6 -> +0.5; caption: 32.85 -> +0.3; clip: 197.85 -> +0.5; vqa: 93.05 -> +0.5


CLIP_FULL_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clip_full_*.npz")))


@pytest.mark.parametrize("path", CLIP_FULL_CASES, ids=[os.path.basename(c)[:-4] for c in CLIP_FULL_CASES])
def test_oracle_clip_full_matches_reference_fixture(path):
    """clip/model.py CLIP.encode_image / encode_text (both towers; BASELINE config 4) recorded from the
    reference's own CLIP: the oracle in order="reference" mode reproduces features, per-layer lengths and the kept indices IN
    the reference's order (torch.topk(sorted=False) on the same machine); order="ascending" - the HIP path's canonical order -
    keeps the same SETS at the first pruned layer and differs afterwards only through the positional mask / read-out."""
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    shapes = specs.clip_shapes(size)
    W = specs.synth_weights(shapes, seed)
    ref_keys = {str(k) for k in g["state_dict_keys"]}
    assert set(shapes.keys()) <= ref_keys
    assert all(k.endswith("_m") or "_m." in k or "queue" in k for k in ref_keys - set(shapes.keys()))  # training state only
    images = synth.synth_images(B, size, seed)
    text = synth.synth_clip_tokens(B, 77, seed, int(g["min_len"]), int(g["max_len"]))
    assert text.argmax(-1).tolist() == g["eot_pos"].tolist()
    vtr, ttr = [], []
    with torch.no_grad():
        fi, sdi = O.clip_encode_image(W, images, W["space_dict"], T, trace=vtr)
        ft, sdt = O.clip_encode_text(W, text, W["space_dict"], T, order="reference", trace=ttr)
    assert np.abs(fi.numpy() - g["image_features"]).max() < 1e-4
    assert np.abs(ft.numpy() - g["text_features"]).max() < 1e-4
    assert np.abs(sdt[:, :4, :16].numpy() - g["sd_txt_head"]).max() < 1e-3
    for key, tr, lens in (("vit", vtr, g["vit_lens"]), ("txt", ttr, g["txt_lens"])):
        for l, info in enumerate(tr):
            if f"{key}{l}_idx" not in g.files:
                assert not info["pruned"]
                continue
            assert info["pruned"] and info["k"] + 2 == lens[l]
            if key == "txt":
                assert (info["indices"].numpy() == g[f"txt{l}_idx"]).all()  # same order as the reference on this machine
            else:
                assert (np.sort(info["indices"].numpy(), 1) == np.sort(g[f"vit{l}_idx"], 1)).all()
    atr = []
    with torch.no_grad():
        fa, _ = O.clip_encode_text(W, text, W["space_dict"], T, order="ascending", trace=atr)
    first = next(l for l, info in enumerate(ttr) if info["pruned"])
    assert (np.sort(atr[first]["indices"].numpy(), 1) == np.sort(g[f"txt{first}_idx"], 1)).all()
    assert [i["k"] if i["pruned"] else None for i in atr][: first + 1] == [i["k"] if i["pruned"] else None for i in ttr][: first + 1]
    assert torch.isfinite(fa).all()


VQA_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vqa[0-9]*.npz")))


def vqa_inputs(g):
    from madtp_amd import harness
    B, size, L, seed = int(g["B"]), int(g["size"]), int(g["L"]), int(g["seed"])
    images = synth.synth_images(B, size, seed)
    ids = synth.synth_token_ids(B, L, seed, first_id=None)
    att = harness.padded_mask(B, L, int(g["pad_tail"]))
    return images, ids, att


@pytest.mark.parametrize("path", VQA_CASES, ids=[os.path.basename(c)[:-4] for c in VQA_CASES])
def test_oracle_vqa_encoder_matches_reference_fixture(path):
    """models/blip_vqa.py BLIP_VQA encoder leg (BASELINE config 5: 480^2 = 901 visual tokens): ViT + MED multimodal encoder."""
    g = np.load(path)
    T = float(g["temperature"])
    shapes = specs.blip_vqa_shapes(int(g["size"]))
    keys = {str(k) for k in g["state_dict_keys"]}
    assert set(shapes.keys()) <= keys and all("position_ids" in k for k in keys - set(shapes.keys()))
    W = specs.synth_weights(shapes, int(g["seed"]))
    images, ids, att = vqa_inputs(g)
    tr = {}
    with torch.no_grad():
        hid = O.blip_vqa_encoder_forward(W, images, ids, att, T, trace=tr)
    assert list(hid.shape) == g["hidden_shape"].tolist()
    assert np.abs(hid[:, 0, :32].numpy() - g["hidden_cls"]).max() < 1e-4
    assert np.abs(tr["image_embeds"][:, 0, :16].numpy() - g["img_embeds_cls"]).max() < 1e-4
    for key, trace, lens, n0 in (("vit", tr["vit"], g["vit_lens"], 901), ("txt", tr["text"], g["txt_lens"], int(g["L"]))):
        for l, info in enumerate(trace):
            if f"{key}{l}_idx" not in g.files:
                assert info is None or not info["pruned"]
                continue
            assert info["pruned"] and info["k"] + 2 == lens[l]
            ref = g[f"{key}{l}_idx"][:, : info["k"]]
            assert (np.sort(info["indices"].numpy(), 1) == np.sort(ref, 1)).all()


VQA_RANK_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vqa_rank_*.npz")))


def vqa_rank_inputs(g):
    images, ids, att = vqa_inputs(g)
    a_ids, a_att = synth.synth_answer_ids(int(g["n_answers"]), int(g["answer_len"]), int(g["seed"]))
    return images, ids, att, a_ids, a_att


@pytest.mark.parametrize("path", VQA_RANK_CASES, ids=[os.path.basename(c)[:-4] for c in VQA_RANK_CASES])
def test_oracle_vqa_rank_answer_matches_reference_fixture(path):
    """SURVEY.md 8(f) rank 4, inference half: BLIP_VQA.forward(train=False, inference='rank') of the reference - encoder leg,
    teacher-forced answer decoder (models/med.py BertLMHeadModel) and rank_answer (blip_vqa.py:156-203) - recorded from the
    reference's own modules: first-token probabilities, the k candidates per question, their sequence log-likelihoods and the
    chosen answers."""
    g = np.load(path)
    T, k = float(g["temperature"]), int(g["k_test"])
    shapes = specs.blip_vqa_shapes(int(g["size"]), decoder=True)
    dec_keys = {str(x) for x in g["decoder_state_dict_keys"]}
    mine = {x for x in shapes if x.startswith("text_decoder.")}
    assert mine == dec_keys, (sorted(mine - dec_keys)[:5], sorted(dec_keys - mine)[:5])
    W = vqa_decoder_weights(int(g["size"]), int(g["seed"]))
    assert torch.equal(W["text_decoder.cls.predictions.decoder.weight"], W["text_decoder.bert.embeddings.word_embeddings.weight"])
    images, ids, att, a_ids, a_att = vqa_rank_inputs(g)
    tr, det = {}, {}
    with torch.no_grad():
        max_ids = O.blip_vqa_rank_forward(W, images, ids, att, a_ids, a_att, T, k, trace=tr, detail=det)
    n0 = (int(g["size"]) // 16) ** 2 + 1
    from madtp_amd import harness
    assert harness.token_lengths(tr["vit"], n0) == g["vit_lens"].tolist()
    assert harness.token_lengths(tr["text"], int(g["L"])) == g["txt_lens"].tolist()
    assert np.abs(det["first_logits"][:, :64].numpy() - g["first_logits_sample"]).max() < 1e-4
    assert np.abs(det["prob_first_token"].numpy() - g["prob_first_token"]).max() < 1e-6
    assert det["topk_ids"].tolist() == g["topk_ids"].tolist()
    assert np.abs(det["log_probs_sum"].numpy() - g["log_probs_sum"]).max() < 1e-3
    assert max_ids.tolist() == g["max_ids"].tolist()


NLVR_PAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nlvrpad_*.npz")))


def nlvr_pad_layers(g):
    """Text layers of a ragged-caption fixture at which some sample's top-(k+1) (nlvr_encoder.py:452: indices_sort[:, :k+1])
    contains a PADDED position - from the recorded `indices_sort` / lengths alone, following the mask compaction rule."""
    from madtp_amd import harness
    B, L = int(g["B"]), int(g["L"])
    mask = harness.padded_mask(B, L, g["pad_list"].tolist()).numpy()
    layers = []
    for l in range(12):
        if f"txt{l}_sort" not in g.files:
            continue
        k = int(g["txt_lens"][l]) - 2
        srt = g[f"txt{l}_sort"][:, :k + 1]
        picked = np.take_along_axis(mask[:, 1:], srt, axis=1)
        if (picked == 0).any():
            layers.append(l)
        mask = np.concatenate([mask[:, :1], picked], axis=1)
    return layers


@pytest.mark.parametrize("path", NLVR_PAD_CASES, ids=[os.path.basename(c)[:-4] for c in NLVR_PAD_CASES])
def test_oracle_nlvr_pad_inside_topk_fixture(path):
    """nlvr_encoder.py:440-452 with ragged captions: k = max_b count comes from the long caption, the short ones keep PADDED
    tokens inside their top-(k+1), and the reference pairs tokens in topk(sorted=False) order with mask entries in SORTED order.
    The oracle in text_order="reference" reproduces the recording (same torch build: kept sets of every layer, logits); in
    text_order="ascending" - the HIP path's canonical pairing - the kept SETS still agree up to and including the first layer
    with a pad inside the top-(k+1) (the pairing only acts on the layers after it)."""
    from madtp_amd import harness
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    W = specs.synth_weights(specs.blip_nlvr_shapes(size), seed)
    images, ids = synth.synth_images(2 * B, size, seed), synth.synth_token_ids(B, L, seed)
    att = harness.padded_mask(B, L, g["pad_list"].tolist())
    pad_layers = nlvr_pad_layers(g)
    assert pad_layers, "the fixture must exercise the pad-inside-top-(k+1) case"
    first = pad_layers[0]
    tr_ref, tr_asc = {}, {}
    with torch.no_grad():
        logits = O.blip_nlvr_forward(W, images, ids, att, T, trace=tr_ref, text_order="reference")
        logits_asc = O.blip_nlvr_forward(W, images, ids, att, T, trace=tr_asc, text_order="ascending")
    assert np.abs(logits.numpy() - g["logits"]).max() < 1e-5
    for l, info in enumerate(tr_ref["text"]):
        if f"txt{l}_idx" not in g.files:
            assert not info["pruned"]
            continue
        assert info["pruned"] and info["k"] + 2 == g["txt_lens"][l]
        assert (info["indices"].numpy() == g[f"txt{l}_idx"]).all()       # this build's topk(sorted=False) order
        assert (info["indices_sort"].numpy() == g[f"txt{l}_sort"]).all()
    ref_sets, asc_sets = harness.compose_ids(tr_ref["text"], L - 1), harness.compose_ids(tr_asc["text"], L - 1)
    for l in range(first + 1):
        assert asc_sets[l] == ref_sets[l], l
    differ = [l for l in range(12) if asc_sets[l] != ref_sets[l]]
    print(f"pads inside top-(k+1) at text layers {pad_layers}; ascending-order pairing differs from this build's at layers "
          f"{differ}; |dlogit| {np.abs(logits_asc.numpy() - logits.numpy()).max():.4f}")
    assert all(l > first for l in differ)
    # What exactly differs AT the pad layer (stated slot by slot): both pairings take the compacted mask by RANK (indices_sort[:, :k+1],
    # nlvr_encoder.py:452), so the mask vector handed to the next layer is the SAME; the kept token SET is the same; only the
    # permutation of the kept tokens under that mask differs (torch.topk(sorted=False)'s order there, ascending token order here),
    # i.e. WHICH kept token sits under a -10000 entry.
    ref_i, asc_i = tr_ref["text"][first], tr_asc["text"][first]
    k = ref_i["k"]
    in_mask = harness.padded_mask(B, L, g["pad_list"].tolist())
    for l in range(first):  # layers before the first pad layer that pruned: follow the mask compaction
        if tr_ref["text"][l]["pruned"]:
            srt = tr_ref["text"][l]["indices_sort"][:, : tr_ref["text"][l]["k"] + 1]
            in_mask = torch.cat([in_mask[:, :1], torch.gather(in_mask[:, 1:], 1, srt)], 1)
    pad_in = in_mask[:, 1:] == 0                                      # padded positions (attention mask 0) of the layer input, per token
    by_rank = torch.gather(pad_in, 1, ref_i["indices_sort"][:, :k])   # the -10000 pattern of slots 0..k-1 (both pairings)
    assert torch.equal(ref_i["indices_sort"], asc_i["indices_sort"])
    under_ref = torch.gather(pad_in, 1, ref_i["indices"])             # is the token in slot p a padded one? (reference order)
    under_asc = torch.gather(pad_in, 1, asc_i["indices"])
    slots_ref = [(by_rank[b] != under_ref[b]).nonzero().flatten().tolist() for b in range(B)]
    slots_asc = [(by_rank[b] != under_asc[b]).nonzero().flatten().tolist() for b in range(B)]
    print(f"layer {first}: slots whose mask entry does not belong to the token in them - reference order {slots_ref}, ascending "
          f"order {slots_asc}")
    for b in range(B):
        if int(g["pad_list"][b]) == 0:      # the unpadded caption has no -10000 entry at all: nothing to mis-pair
            assert slots_ref[b] == [] and slots_asc[b] == []
        assert len(slots_ref[b]) % 2 == 0 and len(slots_asc[b]) % 2 == 0  # mis-pairings come in (real under -10000, pad under 0) pairs
    if len(pad_layers) == 1:
        # single-layer fixture: the number of kept padded tokens is the same in both pairings (same SET); bounded by the pads present
        for b in range(B):
            assert int(under_ref[b].sum()) == int(under_asc[b].sum()) <= int(g["pad_list"][b])


# ---- beam-search generation (inference half; transformers 4.15 restated in oracle.beam_search) --------
VQA_GEN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vqa_gen_*.npz")))
# items of a fixture whose winner ran to max_length on both sides (no finished hypothesis is involved in the choice): the
# recording was made under transformers 5.15, whose re-implemented search scores FINISHED hypotheses differently from 4.15
VQA_GEN_COMPARABLE = {"vqa_gen_b2": [0, 1], "vqa_gen_b3_T30_eos": [1], "vqa_gen_b4_eos26": []}


import functools


@functools.lru_cache(maxsize=3)
def vqa_decoder_weights(size, seed):
    """(generated once per session and shared by the rank / generate tests: ~360 M values, 15-20 s)"""
    return specs.synth_weights(specs.blip_vqa_shapes(size, decoder=True), seed)


def vqa_gen_weights(g):
    W = dict(vqa_decoder_weights(int(g["size"]), int(g["seed"])))  # shallow copy: only the two bias entries are replaced
    b = W["text_decoder.cls.predictions.bias"].clone()
    b[102] += float(g["eos_bias"])
    W["text_decoder.cls.predictions.bias"] = b
    return specs.tie_keys(W)


@pytest.mark.parametrize("path", VQA_GEN_CASES, ids=[os.path.basename(c)[:-4] for c in VQA_GEN_CASES])
def test_oracle_vqa_generate_matches_reference_fixture(path):
    """BLIP_VQA.forward(train=False, inference='generate') of the reference (blip_vqa.py:127-147) recorded with its own modules:
    the first step's six best accumulated log-probabilities of every item (log-softmax + top-2k) and - where no finished
    hypothesis decides - the generated sequences."""
    g = np.load(path)
    name = os.path.basename(path)[:-4]
    images, ids, att = vqa_inputs(g)
    tr = []
    with torch.no_grad():
        seq = O.blip_vqa_generate_forward(vqa_gen_weights(g), images, ids, att, float(g["temperature"]), num_beams=int(g["num_beams"]),
                                          max_length=int(g["max_length"]), min_length=int(g["min_length"]), beam_trace=tr)
    nb = int(g["num_beams"])
    assert np.abs(tr[0]["next_scores"].numpy() - g["first_log_probs_top"][::nb]).max() < 1e-4
    ref = g["sequences"]
    for b in VQA_GEN_COMPARABLE[name]:
        assert seq[b].tolist() == ref[b].tolist()[:seq.shape[1]], (b, seq[b].tolist(), ref[b].tolist())
    # the restatement of the search that RAN (transformers 5.15, oracle.beam_search_hf5) reproduces every recorded sequence, the
    # early-finished ones included: what differs between the two library versions is confined to the lines its docstring names
    with torch.no_grad():
        seq5 = O.blip_vqa_generate_forward(vqa_gen_weights(g), images, ids, att, float(g["temperature"]), num_beams=nb,
                                           max_length=int(g["max_length"]), min_length=int(g["min_length"]), library="5.15")
    assert seq5.tolist() == ref.tolist()
    # the decoder inputs of the reference's second step = the beams the first step kept (same rule in both library versions)
    step1 = g["step_input_ids"][1][:, :2]
    mine = torch.cat([torch.full((step1.shape[0], 1), O.BOS_TOKEN_ID), tr[0]["beam_tokens"].view(-1, 1)], 1).numpy()
    live = [r for r in range(step1.shape[0]) if (r // nb) in VQA_GEN_COMPARABLE[name]]
    assert np.array_equal(mine[live], step1[live])


def toy_lm(table):
    """step_fn of a first-order toy language model: next-token LOG-probabilities depend on the last token only."""
    lt = torch.log(torch.tensor(table, dtype=torch.float32))
    return lambda ids: lt[ids[:, -1]]


TOY = [[0.2, 0.2, 0.2, 0.2, 0.2],            # after [PAD] (rows of finished items)
       [0.2, 0.2, 0.2, 0.2, 0.2],            # after EOS (never expanded)
       [0.01, 0.05, 0.04, 0.5, 0.4],         # after 2
       [0.001, 0.9, 0.05, 0.03, 0.019],      # after 3
       [0.001, 0.8, 0.1, 0.06, 0.039]]       # after 4


def test_beam_search_hypothesis_bookkeeping_hand_worked():
    """transformers 4.15 BeamSearchScorer / BeamHypotheses on a 5-token toy model (0 = PAD, 1 = EOS), worked by hand:
    item 0 (prompt [2]): step 1 finishes [2,3] (-0.798 / 2 = -0.399) and [2,4] (-1.139 / 2 = -0.570); not done, because the best
      candidate could still reach -0.798 / 2 > -0.570; step 2's best open candidate is -3.912 / 3 = -1.304 < -0.570 -> done;
      answer [2,3,EOS].
    item 1 (prompt [4]): EOS is the best first candidate -> hypothesis [4] (-0.223 / 1); step 1 finishes [4,3] (-2.918 / 2) and
      the heuristic closes the item (worst finished == best possible); answer [4,EOS] padded to the batch length."""
    prompt = torch.tensor([[2], [2], [4], [4]])
    out = O.beam_search(toy_lm(TOY), prompt, num_beams=2, max_length=6, min_length=0, eos_token_id=1, pad_token_id=0)
    assert out.tolist() == [[2, 3, 1], [4, 1, 0]]
    # min_length = 2 forbids the EOS directly after the one-token prompt: item 1 must open with its best non-EOS token
    out = O.beam_search(toy_lm(TOY), prompt, num_beams=2, max_length=6, min_length=2, eos_token_id=1, pad_token_id=0)
    assert out[1].tolist()[:2] == [4, 2] and out[0].tolist()[:3] == [2, 3, 1]


def test_beam_search_length_normalisation_and_eos_rank_rule():
    """max_length = 3: [2,3]+EOS finishes with -1.204 / 2 = -0.602, the open beam [2,4,2] ends with -1.309 / 3 = -0.436 and wins
    (sum_logprobs / len ** 1.0); an EOS that ranks below the top num_beams candidates is not a hypothesis."""
    table = [[0.2] * 5, [0.2] * 5,
             [0.0001, 0.15, 0.0499, 0.5, 0.3],
             [0.0001, 0.6, 0.2999, 0.06, 0.04],
             [0.0001, 0.05, 0.9, 0.03, 0.0199]]
    tr = []
    out = O.beam_search(toy_lm(table), torch.tensor([[2], [2]]), num_beams=2, max_length=3, min_length=0, eos_token_id=1,
                        pad_token_id=0, trace=tr)
    assert out.tolist() == [[2, 4, 2]]
    assert tr[0]["next_tokens"][0].tolist()[:3] == [3, 4, 1] and tr[0]["beam_tokens"].tolist() == [3, 4]  # EOS third: skipped


CAP_GEN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cap_gen_*.npz")))


def cap_gen_weights(g):
    W = specs.synth_weights(specs.blip_decoder_shapes(int(g["size"])), int(g["seed"]))
    W["text_decoder.cls.predictions.bias"][102] += float(g["eos_bias"])
    return specs.tie_keys(W)


@pytest.mark.parametrize("path", CAP_GEN_CASES, ids=[os.path.basename(c)[:-4] for c in CAP_GEN_CASES])
def test_oracle_caption_generate_matches_reference_fixture(path):
    """models/blip.py BLIP_Decoder.generate(sample=False, num_beams=3) (:161-196; compress_caption_dtp.py:86) recorded from the
    reference's own modules: state-dict keys, the pruned ViT's token counts, the prompt ids, the beams kept after the first step
    and the generated sequences (both fixtures: the winner is the same under the 4.15 and the 5.15 scoring of finished
    hypotheses - in cap_gen_b3_T30_eos every item closes with [SEP] right at min_length)."""
    g = np.load(path)
    shapes = specs.blip_decoder_shapes(int(g["size"]))
    assert set(shapes) == {str(k) for k in g["state_dict_keys"]}
    images = synth.synth_images(int(g["B"]), int(g["size"]), int(g["seed"]))
    tr, vt = [], []
    with torch.no_grad():
        seq = O.blip_decoder_generate_forward(cap_gen_weights(g), images, float(g["temperature"]), num_beams=int(g["num_beams"]),
                                              max_length=int(g["max_length"]), min_length=int(g["min_length"]), beam_trace=tr,
                                              trace=vt)
    from madtp_amd import harness
    assert harness.token_lengths(vt, (int(g["size"]) // 16) ** 2 + 1) == g["vit_lens"].tolist()
    assert g["prompt_input_ids"][0].tolist() == [O.BOS_TOKEN_ID] + list(O.CAPTION_PROMPT_IDS[1:-1])
    if float(g["eos_bias"]) == 0.0:
        assert np.abs(tr[0]["next_scores"].numpy() - g["first_log_probs_top"][::int(g["num_beams"])]).max() < 1e-4
    assert np.array_equal(g["second_step_input_ids"][:, -1], tr[0]["beam_tokens"].numpy())
    assert seq.tolist() == g["sequences"].tolist()
    with torch.no_grad():
        seq5 = O.blip_decoder_generate_forward(cap_gen_weights(g), images, float(g["temperature"]), num_beams=int(g["num_beams"]),
                                               max_length=int(g["max_length"]), min_length=int(g["min_length"]), library="5.15")
    assert seq5.tolist() == g["sequences"].tolist()


GRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "blockgrad_*.npz")))


@pytest.mark.parametrize("path", GRAD_CASES, ids=[os.path.basename(c)[:-4] for c in GRAD_CASES])
def test_oracle_block_backward_matches_reference_grads(path):
    """autograd through oracle.vit_block == the reference's own .grad of models/vit.py Block.forward (x,
    token_attn and all 12 parameters), recorded by tools/make_golden.py::vit_block_grad_case.  Pins the checker of the HIP
    backward (tests/test_backward_gpu.py)."""
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build(g)
    # (layer 0's input is the patch embedding; deeper inputs went through `layer` blocks of CPU matmuls whose summation order
    #  depends on the thread count of the moment: float noise, not bits)
    assert np.allclose(c["x"][:, :2, :8].numpy(), g["x_head"], rtol=2e-5, atol=2e-6), "block input differs from the recording"
    assert np.allclose(c["token_attn"][:, :2, :8].numpy(), g["ta_head"], rtol=2e-5, atol=1e-4)
    grads, y, info = O.vit_block_grads(c["W"], c["prefix"], c["x"], c["token_attn"], c["T"], c["G"])
    assert tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
    assert {int(v) for v in info["indices"][0]} == {int(v) for v in g["blk_idx"][0]}
    assert np.allclose(y[:, :3, :16].numpy(), g["y_head"], rtol=1e-5, atol=1e-6)
    grad_case.check_against_fixture(g, grads, 2e-5, "oracle autograd vs reference")


VITGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "encgrad_*.npz")))


@pytest.mark.parametrize("path", VITGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in VITGRAD_CASES])
def test_oracle_vit_backward_matches_reference_grads(path):
    """SURVEY 8(f) rank 4, the whole encoder: autograd through oracle.vit_forward == the reference's own .grad of
    models/vit.py VisionTransformer.forward (all 150 parameters + space_dict; loss oracle.vit_loss on the image tokens), recorded
    by tools/make_golden.py::vit_grad_case.  Pins the checker of the HIP path's ViT backward (tests/test_backward_gpu.py)."""
    from madtp_amd import specs, synth
    from tests import grad_case
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    W = specs.synth_weights(specs.vit_shapes("", size), seed)
    images = synth.synth_images(B, size, seed)
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed)
    gv, hv, av = grad_case.vit_loss_vectors(g)
    grads, y, trace = O.vit_grads(W, "", images, space_dict, T, gv, hv, a=av)
    assert tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
    assert abs(float(y.double().norm()) - float(g["y_norm"])) < 1e-5 * float(g["y_norm"])
    for i, t in enumerate(trace):
        if f"vit{i}_idx" in g.files:
            assert [set(r.tolist()) for r in t["indices"]] == [set(r.tolist()) for r in g[f"vit{i}_idx"]], f"layer {i} kept sets"
    grad_case.check_against_fixture(g, grads, 5e-5, "oracle autograd vs reference (whole ViT)")


MEDGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "medgrad_*.npz")))


@pytest.mark.parametrize("path", MEDGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in MEDGRAD_CASES])
def test_oracle_med_layer_backward_matches_reference_grads(path):
    """SURVEY 8(f) rank 4, text side: autograd through oracle.bert_layer (mode 'text', pruned) == the reference's own .grad of
    models/med.py BertLayer.forward (hidden, token_attn, 16 parameters), recorded by tools/make_golden.py::med_layer_grad_case."""
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_med(g)
    assert np.allclose(c["hidden"][:, :2, :8].numpy(), g["h_head"], rtol=2e-5, atol=2e-6), "layer input differs from the recording"
    assert np.allclose(c["token_attn"][:, :2, :8].numpy(), g["ta_head"], rtol=2e-5, atol=1e-4)
    grads, y, mask_out, info = O.bert_layer_grads(c["W"], c["prefix"], c["hidden"], c["add_mask"], c["T"], c["token_attn"],
                                                  c["g"], c["h"], layer_num=c["layer"], enc=c["enc"])
    assert tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
    assert abs(float(y.double().norm()) - float(g["y_norm"])) < 1e-5 * float(g["y_norm"])
    assert np.array_equal(mask_out[:, 0, 0, :].numpy(), g["mask_out"])
    grad_case.check_against_fixture(g, grads, 5e-5, "oracle autograd vs reference (MED text layer)")


NLVRGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nlvrgrad_*.npz")))


@pytest.mark.parametrize("path", NLVRGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in NLVRGRAD_CASES])
def test_oracle_nlvr_layer_backward_matches_reference_grads(path):
    """SURVEY 8(f) rank 4, the headline's text layer: autograd through oracle.bert_layer (variant 'nlvr', twin cross-attention,
    average below layer 6 / merge_layer from 6 on) == the reference's own .grad of models/nlvr_encoder.py BertLayer.forward (hidden,
    token_attn, both image sequences, 36 / 38 parameters), recorded by tools/make_golden.py::nlvr_layer_grad_case."""
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_nlvr(g)
    assert np.allclose(c["hidden"][:, :2, :8].numpy(), g["h_head"], rtol=2e-5, atol=2e-6), "layer input differs from the recording"
    assert np.allclose(c["token_attn"][:, :2, :8].numpy(), g["ta_head"], rtol=2e-5, atol=1e-4)
    grads, y, mask_out, info = O.bert_layer_grads(c["W"], c["prefix"], c["hidden"], c["add_mask"], c["T"], c["token_attn"],
                                                  c["g"], c["h"], layer_num=c["layer"], variant="nlvr", enc=c["enc"],
                                                  enc_mask=c["enc_mask"])
    assert tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
    assert abs(float(y.double().norm()) - float(g["y_norm"])) < 1e-5 * float(g["y_norm"])
    grad_case.check_against_fixture(g, grads, 5e-5, "oracle autograd vs reference (NLVR layer)")


MODELGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "modelgrad_nlvr_*.npz")))


@pytest.mark.parametrize("path", MODELGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in MODELGRAD_CASES])
def test_oracle_nlvr_model_backward_matches_reference_grads(path):
    """SURVEY 8(f) rank 4, the headline model end to end: autograd through oracle.blip_nlvr_forward == the reference's own .grad
    of models/blip_nlvr.py BLIP_NLVR.forward(train=False) for all 579 parameters (space_dict included), loss = sum(logits * c);
    recorded by tools/make_golden.py::nlvr_model_grad_case.  Pins the checker of the HIP path's model-level backward."""
    from madtp_amd import specs, synth, harness
    from tests import grad_case
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    W = specs.synth_weights(specs.blip_nlvr_shapes(size), seed)
    images = synth.synth_images(2 * B, size, seed)
    ids = synth.synth_token_ids(B, L, seed)
    att = harness.padded_mask(B, L, int(g["pad_tail"]))
    c = torch.from_numpy(synth.uniform_pm1("nlvrgrad_c", B * 2, seed).reshape(B, 2))
    grads, logits, trace = O.blip_nlvr_grads(W, images, ids, att, T, c)
    assert np.abs(logits.numpy() - g["logits"]).max() < 1e-5
    assert harness.token_lengths(trace["vit"], (size // 16) ** 2 + 1) == g["vit_lens"].tolist()
    assert harness.token_lengths(trace["text"], L) == g["txt_lens"].tolist()
    grad_case.check_against_fixture(g, grads, 1e-4, "oracle autograd vs reference (BLIP_NLVR)")


TRAINSTEP_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trainstep_nlvr_*.npz")))


@pytest.mark.parametrize("path", TRAINSTEP_CASES, ids=[os.path.basename(c)[:-4] for c in TRAINSTEP_CASES])
def test_oracle_nlvr_training_step_matches_reference(path):
    """The reference's compression training step (compress_nlvr_dtp.py:52-56: loss_ori + 0.1 loss_fdt from
    BLIP_NLVR.forward(train=True), blip_nlvr.py:84-98; model.eval(), i.e. without dropout): both losses and the gradients of all
    579 parameters, oracle autograd vs the recording of tools/make_golden.py::nlvr_model_grad_case(train=True)."""
    from madtp_amd import specs, synth, harness
    from tests import grad_case
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    W = specs.synth_weights(specs.blip_nlvr_shapes(size), seed)
    images = synth.synth_images(2 * B, size, seed)
    ids = synth.synth_token_ids(B, L, seed)
    att = harness.padded_mask(B, L, int(g["pad_tail"]))
    grads, lo, lf = O.blip_nlvr_train_grads(W, images, ids, att, torch.arange(B) % 2, T)
    assert abs(float(lo) - float(g["loss_ori"])) < 1e-5 and abs(float(lf) - float(g["loss_fdt"])) < 1e-5
    grad_case.check_against_fixture(g, grads, 1e-4, "oracle training step vs reference")


DECGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "decgrad_*.npz")))


@pytest.mark.parametrize("path", DECGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in DECGRAD_CASES])
def test_oracle_decoder_backward_matches_reference_grads(path):
    """SURVEY 8(f) rank 4, the answer decoder's training forward (blip_vqa.py:101-113 -> med.py BertLMHeadModel.forward with
    labels, reduction='none'): per-sequence losses and the gradients of all 321 decoder parameters + the question states, oracle
    autograd vs the recording of tools/make_golden.py::decoder_grad_case (tied output embedding: one leaf under two names)."""
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_decoder(g)
    grads, loss = O.bert_lm_grads(c["W"], "", c["ids"], c["att"], c["enc"], c["enc_att"], c["labels"], c["w"])
    assert np.abs(loss.numpy() - g["loss"]).max() < 1e-4 * np.abs(g["loss"]).max()
    grad_case.check_against_fixture(g, grads, 1e-4, "oracle autograd vs reference (decoder)")


VQATRAIN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trainstep_vqa_*.npz")))


@pytest.mark.parametrize("path", VQATRAIN_CASES, ids=[os.path.basename(c)[:-4] for c in VQATRAIN_CASES])
def test_oracle_vqa_training_step_matches_reference(path):
    """The reference's BLIP_VQA training step (blip_vqa.py:57-115, loss_vqa + 0.1 loss_fdt; model.eval()): both losses and the
    gradients of all 788 parameters, oracle autograd vs the recording of tools/make_golden.py::vqa_train_case."""
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_vqa_train(g)
    grads, lv, lf = O.blip_vqa_train_grads(c["W"], c["images"], c["ids"], c["att"], c["a_ids"], c["a_att"], c["n_list"],
                                           c["weights"], c["T"])
    assert abs(float(lv) - float(g["loss_vqa"])) < 1e-4 * float(g["loss_vqa"]) and abs(float(lf) - float(g["loss_fdt"])) < 1e-5
    grad_case.check_against_fixture(g, grads, 1e-4, "oracle VQA training step vs reference")


CAPTRAIN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trainstep_cap_*.npz")))


@pytest.mark.parametrize("path", CAPTRAIN_CASES, ids=[os.path.basename(c)[:-4] for c in CAPTRAIN_CASES])
def test_oracle_caption_training_step_matches_reference(path):
    """The reference's BLIP_Decoder training step (models/blip.py:111-158; model.eval()): loss_lm and the gradients of all 472
    parameters, oracle autograd vs the recording of tools/make_golden.py::cap_train_case."""
    from madtp_amd import specs, synth
    from tests import grad_case
    g = np.load(path)
    B, size, seed = int(g["B"]), int(g["size"]), int(g["seed"])
    W = specs.tie_keys(specs.synth_weights(specs.blip_decoder_shapes(size), seed))
    grads, loss = O.blip_decoder_train_grads(W, synth.synth_images(B, size, seed), torch.from_numpy(g["ids"]), torch.from_numpy(g["att"]),
                                             float(g["temperature"]), int(g["prompt_length"]))
    assert abs(float(loss) - float(g["loss_lm"])) < 1e-4 * float(g["loss_lm"])
    grad_case.check_against_fixture(g, grads, 1e-4, "oracle caption training step vs reference")


CLIPGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clipgrad_*.npz")))


@pytest.mark.parametrize("path", CLIPGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in CLIPGRAD_CASES])
def test_oracle_clip_block_backward_matches_reference_grads(path):
    """SURVEY 8(f) rank 4, CLIP: autograd through oracle.clip_block == the reference's own .grad of clip/model.py
    ResidualAttentionBlock.forward (x, space_dict, the block's 12 parameters and its query model's q_map), recorded by
    tools/make_golden.py::clip_block_grad_case."""
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_clip_block(g)
    assert np.allclose(c["x"][:, :2, :8].numpy(), g["x_head"], rtol=2e-5, atol=2e-6)
    grads, y, sd_ft, info = O.clip_block_grads(c["W"], c["prefix"], c["x"], c["space_dict"], c["T"], c["max_keep"], c["g"], c["h"], c["a"])
    assert tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
    assert abs(float(y.double().norm()) - float(g["y_norm"])) < 1e-5 * float(g["y_norm"])
    grad_case.check_against_fixture(g, grads, 5e-5, "oracle autograd vs reference (CLIP block)")


CLIPTEXTGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cliptextgrad_*.npz")))


def _clip_text_block_case(g):
    from madtp_amd import synth
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    names = ["ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "attn.in_proj_weight", "attn.in_proj_bias",
             "attn.out_proj.weight", "attn.out_proj.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias",
             "query_model.q_map.0.weight", "query_model.q_map.0.bias"]
    shapes = {"ln_1.weight": (512,), "ln_1.bias": (512,), "ln_2.weight": (512,), "ln_2.bias": (512,),
              "attn.in_proj_weight": (1536, 512), "attn.in_proj_bias": (1536,), "attn.out_proj.weight": (512, 512),
              "attn.out_proj.bias": (512,), "mlp.c_fc.weight": (2048, 512), "mlp.c_fc.bias": (2048,), "mlp.c_proj.weight": (512, 2048),
              "mlp.c_proj.bias": (512,), "query_model.q_map.0.weight": (768, 512), "query_model.q_map.0.bias": (768,)}
    W = {"b." + k: synth.synth_tensor("clip_text_block." + k, shapes[k], seed) for k in names}
    mask = torch.empty(77, 77).fill_(float("-inf")).triu_(1)
    return {"W": W, "mask": mask, "x": synth.synth_tensor("clip_text_x", (N, B, 512), seed).permute(1, 0, 2).contiguous(),
            "space_dict": synth.synth_tensor("space_dict", (100, 768), seed), "T": float(g["temperature"]), "max_keep": int(g["max_keep"]),
            "g": torch.from_numpy(synth.uniform_pm1("vitgrad_g", B * 512, seed).reshape(B, 512)),
            "h": torch.from_numpy(synth.uniform_pm1("vitgrad_h", B * 512, seed).reshape(B, 512)),
            "a": torch.from_numpy(synth.uniform_pm1("vitgrad_a", B * 100 * 768, seed).reshape(B, 100, 768))}


@pytest.mark.parametrize("path", CLIPTEXTGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in CLIPTEXTGRAD_CASES])
def test_oracle_clip_text_block_backward_matches_reference_grads(path):
    """CLIP text-tower block (width 512, 8 heads, causal mask): oracle autograd == the reference's own .grad
    (tools/make_golden.py::clip_text_block_grad_case)."""
    from tests import grad_case
    g = np.load(path)
    c = _clip_text_block_case(g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in c["W"].items()}
    xl, sl = c["x"].clone().requires_grad_(True), c["space_dict"].clone().requires_grad_(True)
    y, sd_ft, info = O.clip_block(leaves, "b.", xl, sl, c["T"], c["max_keep"], num_heads=8, attn_mask=c["mask"])
    (O.vit_loss(y, c["g"], c["h"]) + (sd_ft * c["a"]).sum()).backward()
    assert tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
    grads = {"x": xl.grad, "space_dict": sl.grad}
    grads.update({k[2:]: v.grad for k, v in leaves.items()})
    grad_case.check_against_fixture(g, grads, 5e-5, "oracle autograd vs reference (CLIP text block)")


CLIPVITGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clipvitgrad_*.npz")))


def _clip_vit_case(g):
    from madtp_amd import specs, synth
    B, size, seed = int(g["B"]), int(g["size"]), int(g["seed"])
    return {"W": specs.synth_weights(specs.clip_vit_shapes("", size), seed), "images": synth.synth_images(B, size, seed),
            "space_dict": synth.synth_tensor("space_dict", (100, 768), seed), "T": float(g["temperature"]),
            "c": torch.from_numpy(synth.uniform_pm1("clipgrad_c", B * 512, seed).reshape(B, 512)),
            "a": torch.from_numpy(synth.uniform_pm1("vitgrad_a", B * 100 * 768, seed).reshape(B, 100, 768))}


@pytest.mark.parametrize("path", CLIPVITGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in CLIPVITGRAD_CASES])
def test_oracle_clip_vision_backward_matches_reference_grads(path):
    """CLIP's vision tower end to end: oracle autograd == the reference's own .grad of clip/model.py VisionTransformer.forward
    (176 parameters + space_dict; tools/make_golden.py::clip_vit_grad_case)."""
    from tests import grad_case
    g = np.load(path)
    c = _clip_vit_case(g)
    grads, feat, trace = O.clip_vision_grads(c["W"], "", c["images"], c["space_dict"], c["T"], c["c"], c["a"])
    assert np.abs(feat.numpy() - g["features"]).max() < 1e-5
    grad_case.check_against_fixture(g, grads, 1e-4, "oracle autograd vs reference (CLIP vision tower)")


def test_oracle_nucleus_filter_matches_transformers_warpers():
    """oracle.nucleus_filter / repetition_penalty_scores (transformers 4.15 restated: the warpers and the processor behind
    models/blip.py:175-186 generate(do_sample=True, top_p=0.9, repetition_penalty=1.1)) against the warpers of the transformers
    version installed in this container, which kept these semantics: the same surviving token sets on random scores (no ties),
    top_k = 50 (config.top_k) in front of top_p, and the same penalised scores."""
    from oracle import madtp_oracle as O
    lp = pytest.importorskip("transformers.generation.logits_process")
    g = torch.Generator().manual_seed(1)
    for V, scale, p in ((1000, 3.0, 0.9), (30524, 1.5, 0.9), (300, 0.2, 0.5), (40, 5.0, 0.95)):
        x = torch.randn(5, V, generator=g) * scale
        mine = O.nucleus_filter(x, p, 50)
        ref = lp.TopPLogitsWarper(p)(None, lp.TopKLogitsWarper(50)(None, x.clone()))
        assert torch.equal(torch.isfinite(mine), torch.isfinite(ref))
        assert torch.equal(mine[torch.isfinite(mine)], ref[torch.isfinite(ref)])
        ids = torch.randint(0, V, (5, 7), generator=g)
        ids[:, 3] = ids[:, 1]  # a token that occurs twice is penalised once
        assert torch.allclose(O.repetition_penalty_scores(x, ids, 1.1), lp.RepetitionPenaltyLogitsProcessor(1.1)(ids, x.clone()))
    # the inverse-CDF draw: u sweeps the unit interval -> every survivor is drawn, in proportion to its probability
    x = torch.randn(1, 64, generator=g) * 2
    w = O.nucleus_filter(x, 0.9, 50).softmax(-1)[0]
    us = (torch.arange(20000, dtype=torch.float64) + 0.5) / 20000
    draws = torch.stack([O.sample_step(x, torch.zeros(1, 1, dtype=torch.long), torch.tensor([float(u)]), 0.9) for u in us[::40]]).flatten()
    freq = torch.bincount(draws, minlength=64).double() / draws.numel()
    assert (freq - w.double()).abs().max().item() < 5e-3 and set(torch.nonzero(freq).flatten().tolist()) <= set(torch.nonzero(w).flatten().tolist())


def test_oracle_philox_matches_random123_known_answers():
    """oracle.philox_uniform (the restatement of csrc/backward.hip's dropout mask generator) against the published Philox4x32-10
    known-answer vectors of Random123 (kat_vectors: counter / key all zero, all ones, and the digits of pi)."""
    import numpy as np

    def words(ctr, key):
        # the restatement's counter is (index lo, index hi, site lo, site hi), its key the seed
        idx, site, seed = ctr[0] | (ctr[1] << 32), ctr[2] | (ctr[3] << 32), key[0] | (key[1] << 32)
        n = 4 * (idx + 1) if idx < 1000 else None
        if n is not None:
            u = O.philox_uniform(seed, site, n)[-4:]
            return [int(round(float(v) * 16777216.0)) for v in u]
        return None
    assert words((0, 0, 0, 0), (0, 0)) == [x >> 8 for x in (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)]
    # counters with a large index: evaluate the generator's words directly through the same arithmetic on one index
    import oracle.madtp_oracle as M
    def raw(ctr, key):
        m = np.uint64(0xFFFFFFFF)
        c0, c1, c2, c3 = (np.array([v], dtype=np.uint64) for v in ctr)
        k0, k1 = np.uint64(key[0]), np.uint64(key[1])
        for _ in range(10):
            p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
            c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & m, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & m
            k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m, (k1 + np.uint64(0xBB67AE85)) & m
        return [int(c0[0]), int(c1[0]), int(c2[0]), int(c3[0])]
    assert raw((0xffffffff,) * 4, (0xffffffff,) * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert raw((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # and the restatement agrees with that arithmetic at an arbitrary (seed, site, index)
    seed, site, i = 0x299f31d0a4093822, 0x0370734413198a2e, 12345
    u = O.philox_uniform(seed, site, 4 * (i + 1))[-4:]
    w = raw((i & 0xffffffff, i >> 32, site & 0xffffffff, site >> 32), (seed & 0xffffffff, seed >> 32))
    assert [int(round(float(v) * 16777216.0)) for v in u] == [x >> 8 for x in w]
    m = O.dropout_mask(5, 3, (4, 1000), 0.1)
    assert len(set(m.reshape(-1).tolist())) == 2 and abs(float(m.max()) - 1.0 / 0.9) < 1e-6 and abs(float((m > 0).float().mean()) - 0.9) < 0.02


def check_rows(g, key, t, tol, what=""):
    """a hidden-state / cache tensor against its fixture record (medopts_*: shape, the first 16 columns and the L2 norm of every row)"""
    t = t.detach().float().cpu()
    assert tuple(t.shape) == tuple(int(v) for v in g[key + "_shape"]), (what, key, tuple(t.shape), g[key + "_shape"])
    ref_sl, ref_nrm = torch.from_numpy(g[key + "_sl"]), torch.from_numpy(g[key + "_rownorm"])
    err = (t[..., :16] - ref_sl).abs().max().item()
    nerr = ((t.double().norm(dim=-1) - ref_nrm).abs() / ref_nrm.clamp_min(1e-6)).max().item()
    assert err < tol and nerr < tol, (what, key, err, nerr)


def options_inputs(g):
    """(case of tests/grad_case.py::build_med, head_mask [1,H,1,1]) of a medopts_* fixture"""
    from tests import grad_case
    c = grad_case.build_med(g)
    assert np.abs(c["hidden"][:, :2, :8].numpy() - g["h_head"]).max() < 1e-5  # the layer input the reference saw
    return c, torch.from_numpy(g["head_mask"]).view(1, -1, 1, 1)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "medopts_*.npz"))))
def test_oracle_bert_layer_options_match_reference_fixture(path):
    """med.py:393-407's head_mask / past_key_value / output_attentions on ONE BertLayer call, recorded from the reference
    (tools/make_golden.py::med_layer_options_case): the oracle returns the same layer outputs, attention probabilities, caches,
    pruning decisions and masks."""
    g = np.load(path)
    c, hm = options_inputs(g)
    W, p, hid, mask, ta, T, enc = c["W"], c["prefix"], c["hidden"], c["add_mask"], c["token_attn"], c["T"], c["enc"]
    Lp = int(g["Lp"])
    with torch.no_grad():
        for tag, t in (("oa0", 0.0), ("oaT", T)):
            ex = {}
            y, m, info = O.bert_layer(W, p, hid, mask, t, ta if t > 0 else None, enc, None, "multimodal", 0, "med", extras=ex)
            check_rows(g, tag + "_out", y, 2e-5, tag)
            assert (ex["self_probs"] - torch.from_numpy(g[tag + "_self_probs"])).abs().max() < 1e-6
            assert (ex["cross_probs"] - torch.from_numpy(g[tag + "_cross_probs"])).abs().max() < 1e-6
            check_rows(g, tag + "_present_k", ex["present"][0], 2e-5, tag)
            check_rows(g, tag + "_present_v", ex["present"][1], 2e-5, tag)
            assert np.array_equal(m[:, 0, 0, :].numpy(), g[tag + "_mask_out"])
            if t > 0:
                assert np.array_equal(info["indices"].numpy(), g["oaT_idx"][:, : info["k"]])  # (med.py:377: topk(k+1), the first k are kept)
        y, _, _ = O.bert_layer(W, p, hid, mask, 0.0, None, enc, None, "multimodal", 0, "med", head_mask=hm)
        check_rows(g, "hm0_out", y, 2e-5, "hm0")
        y, m, info = O.bert_layer(W, p, hid, mask, T, ta, None, None, "text", 0, "med", head_mask=hm)
        check_rows(g, "hmT_out", y, 2e-5, "hmT")
        assert np.array_equal(info["indices"].numpy(), g["hmT_idx"][:, : info["k"]]) and np.array_equal(m[:, 0, 0, :].numpy(), g["hmT_mask_out"])
        ex = {}
        y0, _, _ = O.bert_layer(W, p, hid[:, :Lp], mask[:, :, :, :Lp], 0.0, None, enc, None, "multimodal", 0, "med", extras=ex)
        check_rows(g, "pk0_out", y0, 2e-5, "pk0")
        past = ex["present"]
        for tag, n_new in (("pk1", 1), ("pk2", 2)):
            ex = {}
            y, _, _ = O.bert_layer(W, p, hid[:, Lp:Lp + n_new], mask[:, :, :, :Lp + n_new], 0.0, None, enc, None, "multimodal", 0, "med",
                                   past_key_value=past, extras=ex)
            assert (y - torch.from_numpy(g[tag + "_out"])).abs().max() < 2e-5
            check_rows(g, tag + "_present_k", ex["present"][0], 2e-5, tag)
            check_rows(g, tag + "_present_v", ex["present"][1], 2e-5, tag)
