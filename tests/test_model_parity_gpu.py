"""End-to-end parity of the HIP path on a real MI355X:
   * fp32 parity mode vs (a) the fixtures recorded from the REFERENCE and (b) the CPU oracle run on the same
     seeded inputs: bit-exact kept-token id sets per layer, logits within 1e-3 (north_star tolerance);
   * bf16 fast mode: logits within 5e-2 of the fp32 oracle and the kept-set match rate is reported/bounded.
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nlvr_*.npz")))
# the two modes that carry the parity claim: exact-f32 MFMA, and the fp32-accurate f16-split GEMMs on the f16 MFMA
EXACT_MODES = ["fp32", "f16x3"]


@pytest.fixture(scope="module")
def env():
    from madtp_amd import build, hip, harness, runtime
    build.build(verbose=False)
    hip.load()
    model = harness.build_nlvr(224, 0, "cuda")
    return harness, runtime, model


def _golden_sets(g, key, B2, n0):
    """reference `indices` per layer -> per-layer id sets (same bookkeeping as harness.compose_ids)."""
    from madtp_amd import harness
    trace = []
    for l in range(12):
        if f"{key}{l}_idx" in g.files:
            trace.append({"pruned": True, "indices": g[f"{key}{l}_idx"]})
        else:
            trace.append(None)
    return harness.compose_ids(trace, n0)


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_fp32_mode_matches_reference_fixture(env, path, mode):
    harness, runtime, model = env
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    images, text, targets = harness.nlvr_inputs(B, size, L, seed, pad_tail=int(g["pad_tail"]) if "pad_tail" in g.files else 0)
    with runtime.precision(mode):
        logits, trace = harness.run_nlvr(model, images, text, targets, T)
    assert harness.token_lengths(trace["vit"], 197) == g["vit_lens"].tolist()
    assert harness.token_lengths(trace["text"], L) == g["txt_lens"].tolist()
    for side, key, n0 in (("vit", "vit", 196), ("text", "txt", L - 1)):
        mine = harness.compose_ids(trace[side], n0)
        ref = _golden_sets(g, key, 2 * B, n0)
        for l in range(12):
            assert (mine[l] is None) == (ref[l] is None), (side, l)
            if mine[l] is not None:
                assert mine[l] == ref[l], f"{side} layer {l}: kept-token sets differ from the reference"
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-3


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("B,T,seed", [(3, 2.0, 1), (2, 8.0, 2)])
def test_fp32_mode_matches_oracle(env, B, T, seed, mode):
    """fresh seeds (new weights are NOT regenerated - inputs only), oracle computed on this box's CPU."""
    from madtp_amd import specs, synth
    from oracle import madtp_oracle as O
    harness, runtime, model = env
    images, text, targets = harness.nlvr_inputs(B, 224, 20, seed)
    W = specs.synth_weights(specs.blip_nlvr_shapes(224), 0)
    tr = {}
    with torch.no_grad():
        ref_logits = O.blip_nlvr_forward(W, images.cpu(), text["input_ids"].cpu(), text["attention_mask"].cpu(), T, trace=tr)
    with runtime.precision(mode):
        logits, trace = harness.run_nlvr(model, images, text, targets, T)
    for side, n0 in (("vit", 196), ("text", 19)):
        mine = harness.compose_ids(trace[side], n0)
        ref = O.compose_ids(tr[side], n0)
        assert mine == ref, f"{side}: kept-token sets differ from the oracle"
    assert (logits.cpu() - ref_logits).abs().max().item() < 1e-3


@pytest.mark.parametrize("B,T", [(4, 2.0), (8, 8.612223847001898)])
def test_bf16_mode_index_match_and_logits(env, B, T):
    """bf16 fast mode vs the oracle: the per-layer decisions are sound (teacher-forced: every block fed the oracle's input),
    the free-running sets stay close although one flipped token shifts k = max_b count for the whole batch, logits within
    bf16 error.  Thresholds: about half the slack measured on MI355X (printed), so a regression that halves the match fails."""
    from tests.parity_util import nlvr_index_match
    harness, runtime, model = env
    rep = nlvr_index_match(model, T, ["bf16"], B=B, seed=3)["bf16"]
    print(f"bf16 B={B} T={T:.2f}: {rep}")
    assert rep["vit_layerwise_jaccard"] >= 0.998        # measured 0.9998 / 0.9994
    assert rep["vit_layerwise_exact_match"] >= (0.975 if B <= 4 else 0.95)   # measured 0.989 / 0.974
    assert rep["mean_jaccard"] >= 0.95                  # measured 0.984 / 0.979 (free running)
    assert rep["max_abs_dlogit"] < 1.5e-2               # measured 0.005 / 0.006


@pytest.mark.parametrize("B,T", [(4, 2.0), (8, 8.612223847001898)])
def test_f16_mode_index_match_and_logits(env, B, T):
    """The "f16" fast mode (IEEE f16 operands on the f16 MFMA, round 4) against the oracle, next to the bf16 mode on the same
    inputs: same kernels and speed, three more significand bits - the logit error must come out clearly smaller and the kept
    sets at least as close."""
    from tests.parity_util import nlvr_index_match
    harness, runtime, model = env
    rep = nlvr_index_match(model, T, ["bf16", "f16"], B=B, seed=3)
    print(f"B={B} T={T:.2f}: bf16 {rep['bf16']}\n             f16  {rep['f16']}")
    f, b = rep["f16"], rep["bf16"]
    # per-layer decisions on the oracle's inputs (teacher-forced): fewer flipped tokens than bf16 (measured MI355X: exact 0.989 /
    # 0.995 against bf16's 0.989 / 0.974).  Free running, ONE flipped token still changes k = max_b count for the whole batch
    # (B = 4: every set identical, |dlogit| 1.5e-4; B = 8: an early flip cascades, Jaccard 0.96) - so only the logits of a run
    # whose sets all match are bounded tightly.
    assert f["vit_layerwise_jaccard"] >= 0.9995 and f["vit_layerwise_exact_match"] >= b["vit_layerwise_exact_match"]
    # |dlogit|: about 1.5 x the measured slack (MI355X, round 5: 1.5e-4 at B = 4 with every set identical; 1.7e-2 at B = 8, where ONE
    # early flip changes k = max_b count for the whole batch and cascades - exact sets 0.10, Jaccard 0.96; the headline batch of 64
    # measures 6e-3, bench.py index_match)
    assert f["mean_jaccard"] >= 0.9 and f["max_abs_dlogit"] < (1.2e-2 if B <= 4 else 2.5e-2)
    if f["kept_set_exact_match"] == 1.0:
        assert f["max_abs_dlogit"] < 1e-3


def test_headline_batch_index_match(env):
    """The claims of bench.py's index_match leg at the HEADLINE batch (64 samples = 128 images, calibrated T), asserted:
    (a) the parity modes (fp32, f16x3) reproduce every kept set of the oracle and its logits within 1e-3;
    (b) bf16: per-layer decisions sound (teacher-forced), free-running sets / logits within about half the measured slack
        (MI355X, round 2: exact 0.17, Jaccard 0.68, teacher-forced exact 0.868 / Jaccard 0.998, |dlogit| 0.02), and the
        count-flip report names the layer where k = max_b count first leaves the oracle's (the cascade's start);
    (c) the batch-dependent GEMM dispatch (256x256 / 32x32x16 kernels vs the 16x16x32 wave-specialised one) adds no error:
        the teacher-forced figures under madtp_gemm_set_config(7) are the same within noise."""
    from madtp_amd import configs, hip
    from oracle.index_match import nlvr_index_match
    harness, runtime, model = env
    T = configs.temperature_for("nlvr", 64, 0.5)[0]
    rep = nlvr_index_match(model, T, ["fp32", "f16x3", "bf16", "f16"], B=64, seed=11, count_flips=True)
    # (d) the f16 fast mode at the batch the claim is made on (round 6, the review's item 7c; the B = 8 bounds of
    #     test_f16_mode_index_match_and_logits are cascade bounds and would pass almost any regression): measured on MI355X, rounds 4-6,
    #     driver runs included: exact sets 0.926, Jaccard 0.998, per-layer decisions 0.998, |dlogit| 6.0e-3, no layer where one side
    #     pruned and the other did not
    f = rep["f16"]
    print(f"f16 B=64: {({k: v for k, v in f.items() if k != 'vit_count_flips'})}")
    assert f["kept_set_exact_match"] >= 0.90 and f["mean_jaccard"] >= 0.995, f
    assert f["max_abs_dlogit"] <= 8e-3 and f["vit_layerwise_exact_match"] >= 0.995 and f["pruned_vs_unpruned_layers"] == [], f
    for mode in ("fp32", "f16x3"):
        r = rep[mode]
        print(f"{mode} B=64: {({k: v for k, v in r.items() if k != 'vit_count_flips'})}")
        assert r["kept_set_exact_match"] == 1.0 and r["pruned_vs_unpruned_layers"] == [], r
        assert r["max_abs_dlogit"] <= 1e-3, r
        assert r["vit_layerwise_exact_match"] == 1.0
        assert r["vit_count_flips"]["first_layer_k_differs"] is None
    b = rep["bf16"]
    print(f"bf16 B=64: {({k: v for k, v in b.items() if k != 'vit_count_flips'})}")
    print("bf16 count flips:", b["vit_count_flips"])
    assert b["vit_layerwise_jaccard"] >= 0.997 and b["vit_layerwise_exact_match"] >= 0.80
    assert b["mean_jaccard"] >= 0.5 and b["max_abs_dlogit"] < 4e-2
    # the free-running drop from B = 8 (Jaccard 0.96) is a cascade: it starts where k first differs
    fl = b["vit_count_flips"]
    if b["kept_set_exact_match"] < 0.5:
        assert fl["first_layer_k_differs"] is not None
    with hip.gemm_config(7):
        b7 = nlvr_index_match(model, T, ["bf16"], B=64, seed=11)["bf16"]
    print(f"bf16 B=64, 16x16x32 wave-specialised GEMM only: {b7}")
    assert abs(b7["vit_layerwise_jaccard"] - b["vit_layerwise_jaccard"]) < 2e-3
    assert abs(b7["vit_layerwise_exact_match"] - b["vit_layerwise_exact_match"]) < 0.05


def test_module_error_behaviour(env):
    harness, runtime, model = env
    blk = model.visual_encoder.blocks[0]
    with pytest.raises(RuntimeError):
        blk(torch.zeros(1, 197, 768))  # CPU tensor: loud failure, no eager fallback
    with pytest.raises(ValueError):
        blk(torch.zeros(1, 197, 768, device="cuda"), False, 0, 1.0, None)  # temperature>0 without token_attn


MED_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "med_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", MED_CASES, ids=[os.path.basename(c)[:-4] for c in MED_CASES])
def test_med_bert_fp32_matches_reference_fixture(path, mode):
    """models/med.py BertModel mirror (text mode and multimodal mode, padded masks) vs the reference fixture."""
    from madtp_amd import build, hip, harness, runtime, specs
    from madtp_amd.med import BertConfig, BertModel
    from tests.test_oracle_golden import med_inputs
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    ids, att, enc, enc_att, sd, bert_mode, T = med_inputs(g)
    cfg = BertConfig.med_default()
    model = BertModel(cfg, add_pooling_layer=False)
    model.load_state_dict(specs.synth_weights(specs.bert_shapes("", "med"), int(g["seed"])), strict=False)
    model = model.eval().cuda()
    with runtime.precision(mode), torch.no_grad():
        out, _ = model(ids.cuda(), attention_mask=att.cuda(), encoder_hidden_states=None if enc is None else enc.cuda(),
                       encoder_attention_mask=None if enc_att is None else enc_att.cuda(), mode=bert_mode,
                       space_dict=sd.cuda(), temperature=T)
    hid = out.last_hidden_state
    assert list(hid.shape) == g["hidden_shape"].tolist()
    trace = [None if l.last_prune is None else {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in l.last_prune.items()}
             for l in model.encoder.layer]
    assert harness.token_lengths(trace, int(g["L"])) == g["txt_lens"].tolist()
    mine = harness.compose_ids(trace, int(g["L"]) - 1)
    ref_trace = [{"pruned": True, "indices": g[f"txt{l}_idx"][:, : trace[l]["k"]]} if f"txt{l}_idx" in g.files else None
                 for l in range(12)]
    ref = harness.compose_ids(ref_trace, int(g["L"]) - 1)
    assert mine == ref
    assert np.abs(hid[:, 0, :32].cpu().numpy() - g["hidden_cls"]).max() < 1e-3
    with runtime.precision("bf16"), torch.no_grad():
        outb, _ = model(ids.cuda(), attention_mask=att.cuda(), encoder_hidden_states=None if enc is None else enc.cuda(),
                        encoder_attention_mask=None if enc_att is None else enc_att.cuda(), mode=bert_mode,
                        space_dict=sd.cuda(), temperature=T)
    hb = outb.last_hidden_state
    assert torch.isfinite(hb).all()
    # the [ENC]/CLS row is the one every caller reads (itm_head, cls_head): bf16 GEMM error plus the occasional different kept
    # token leave it close to the fp32-mode row (the other rows hold whichever tokens survived, in a mode-dependent order)
    cb, cf = hb[:, 0, :].float(), hid[:, 0, :].float()
    cos = torch.nn.functional.cosine_similarity(cb, cf, dim=-1).min().item()
    errb = (cb - cf).abs().max().item()
    print(f"MED bf16 vs fp32-mode CLS row: min cosine {cos:.5f}, max abs {errb:.4f} (|cls| max {cf.abs().max().item():.2f})")
    assert cos > 0.99 and errb < 0.5   # measured 0.9969 / 0.29 on the 35 -> 7 token fixtures


CLIP_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clip_vit_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", CLIP_CASES, ids=[os.path.basename(c)[:-4] for c in CLIP_CASES])
def test_clip_vision_fp32_matches_reference_fixture(path, mode):
    """clip/model.py VisionTransformer/ResidualAttentionBlock mirror vs the reference fixture (row a14, config 4)."""
    from madtp_amd import build, hip, harness, runtime, specs, synth
    from madtp_amd.clip_model import VisionTransformer
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    model = VisionTransformer(input_resolution=size, patch_size=16, width=768, layers=12, heads=12, output_dim=512, sd_dim=768)
    sd = specs.synth_weights(specs.clip_vit_shapes("", size), seed)
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd, strict=True)
    model = model.eval().cuda()
    images = synth.synth_images(B, size, seed).cuda()
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed).cuda()
    with runtime.precision(mode), torch.no_grad():
        feat, sd_ft = model(images, space_dict, T, 1)
    trace = [None if b.last_prune is None else {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.last_prune.items()}
             for b in model.transformer.resblocks]
    assert harness.token_lengths(trace, 197) == g["vit_lens"].tolist()
    ref_trace = [{"pruned": True, "indices": g[f"vit{l}_idx"]} if f"vit{l}_idx" in g.files else None for l in range(12)]
    assert harness.compose_ids(trace, 196) == harness.compose_ids(ref_trace, 196)
    assert np.abs(feat.cpu().numpy() - g["features"]).max() < 1e-3
    assert np.abs(sd_ft[:, :4, :16].cpu().numpy() - g["sd_ft_head"]).max() < 1e-2
    with runtime.precision("bf16"), torch.no_grad():
        fb, _ = model(images, space_dict, T, 1)
    errb = (fb.cpu() - torch.from_numpy(g["features"])).abs().max().item()
    print(f"CLIP bf16 features: max abs err {errb:.4f} (|feat| max {np.abs(g['features']).max():.2f})")
    assert torch.isfinite(fb).all() and errb < 0.1
    blk = model.transformer.resblocks[0]
    blk.attn_mask = torch.zeros(4, 4)
    with pytest.raises(ValueError):  # a sequence longer than the block's attention mask
        blk((torch.zeros(5, 1, 768, device="cuda"), None, 0, None, 1))
    blk.attn_mask = None


VIT_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vit*_b*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", VIT_CASES, ids=[os.path.basename(c)[:-4] for c in VIT_CASES])
def test_vit_large_image_fp32_matches_reference_fixture(path, mode):
    """VisionTransformer mirror at 384^2 (577 tokens) and 480^2 (901 tokens - BASELINE config 5, heaviest ragged
    compaction): long-sequence attention kernel + the same pruning kernels, vs the reference fixture."""
    from madtp_amd import build, hip, harness, runtime, specs, synth
    from madtp_amd.vit import VisionTransformer
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    model = VisionTransformer(img_size=size, patch_size=16, embed_dim=768, depth=12, num_heads=12, evaluate=True, sd_dim=768)
    model.load_state_dict(specs.synth_weights(specs.vit_shapes("", size), seed), strict=True)
    model = model.eval().cuda()
    images = synth.synth_images(B, size, seed).cuda()
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed).cuda()
    n0 = (size // 16) ** 2
    with runtime.precision(mode), torch.no_grad():
        out, sd_ft = model(images, space_dict=space_dict, temperature=T)
    trace = [None if b.last_prune is None else {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.last_prune.items()}
             for b in model.blocks]
    assert harness.token_lengths(trace, n0 + 1) == g["vit_lens"].tolist()
    ref_trace = [{"pruned": True, "indices": g[f"vit{l}_idx"]} if f"vit{l}_idx" in g.files else None for l in range(12)]
    assert harness.compose_ids(trace, n0) == harness.compose_ids(ref_trace, n0)
    assert list(out.shape) == g["out_shape"].tolist()
    assert np.abs(out[:, 0, :32].cpu().numpy() - g["cls"]).max() < 1e-3
    with runtime.precision("bf16"), torch.no_grad():
        ob, sd_b = model(images, space_dict=space_dict, temperature=T)
    assert torch.isfinite(ob).all()
    # all modes sum the 12 layers' att_ft in one deferred launch after the last layer; forcing the per-layer path with the
    # per-layer accumulate chain (`sd_ft_all += sd_ft`, vit.py:297-303) gives the same result: bit for bit in the fp32 mode (the
    # exact kernel keeps the layer-by-layer summation order), to f32 rounding in the f16x3 mode (its deferred launch runs f16-split
    # products, the per-layer chain the exact-f32 kernel), to bf16-MFMA rounding in the fast mode
    import madtp_amd.vit as vit_mod
    saved = vit_mod._ENCODER_CALL
    vit_mod._ENCODER_CALL = False
    model.img_query_model.deferred = lambda: None
    try:
        with runtime.precision("bf16"), torch.no_grad():
            ob2, sd_b2 = model(images, space_dict=space_dict, temperature=T)
        with runtime.precision(mode), torch.no_grad():
            out2, sd_ft2 = model(images, space_dict=space_dict, temperature=T)
    finally:
        del model.img_query_model.deferred
        vit_mod._ENCODER_CALL = saved
    assert torch.equal(ob, ob2) and sd_b.shape == sd_ft.shape
    assert (sd_b - sd_b2).abs().max().item() < 1e-4 * sd_b2.abs().max().item()
    assert torch.equal(out, out2)
    if mode == "fp32":
        assert torch.equal(sd_ft, sd_ft2)
    else:
        assert (sd_ft - sd_ft2).abs().max().item() < 2e-6 * max(1.0, sd_ft2.abs().max().item())


@pytest.mark.parametrize("mode", ["fp32", "f16x3", "bf16"])
def test_fused_layer_calls_equal_two_step(mode):
    """madtp_vit_block / madtp_bert_layer (one library call, k through pinned host memory, score launched before the
    projection) give bit-identical results to the half-layer entry points with the host decision made in Python."""
    from madtp_amd import build, hip, runtime, synth
    from madtp_amd.vit import Block
    from madtp_amd.utils import Query_model
    build.build(verbose=False)
    hip.load()
    torch.manual_seed(0)
    B, N, D = 6, 101, 768
    blk = Block(dim=D, num_heads=12, mlp_ratio=4, qkv_bias=True).cuda().eval()
    qm = Query_model(ft_dim=D, sd_dim=D, temperature=1, att_func_type='sparsemax', pool_type='max').cuda()
    x = synth.synth_tensor("x", (B, N, D), 3).cuda()
    sd = synth.synth_tensor("space_dict", (100, D), 3).cuda()
    with runtime.precision(mode), torch.no_grad():
        ta, _, _ = qm(x[:, 1:, :], sd, return_token_att=True)
        w = blk._weights()
        for T in (0.0, 3.0, 50.0):
            y, info = hip.vit_block(w, x, ta if T > 0 else None, T)
            xa, po = hip.vit_block_attn(w, x, ta if T > 0 else None, T)
            k_use = 0
            if T > 0:
                k = hip.batch_max_count(po[2])
                assert info["k"] == k and torch.equal(info["score"], po[0]) and torch.equal(info["count"], po[2])
                if not (k < 1 or (N - 1 - k) <= 1):
                    k_use = k
            y2, idx, idx_sort = hip.vit_block_mlp(w, xa, k_use, po[0] if po else None)
            assert y.shape == y2.shape and torch.equal(y, y2)
            if k_use:
                assert info["pruned"] and torch.equal(info["indices"], idx) and torch.equal(info["indices_sort"], idx_sort)


@pytest.mark.parametrize("mode", ["fp32", "f16x3", "bf16"])
def test_encoder_level_calls_equal_per_layer_path(mode, monkeypatch):
    """madtp_vit_encoder / madtp_bert_encoder (one library call per encoder: the layer loop in C) give bit-identical results
    to the per-layer path (one call per Block / BertLayer, Python in between; `_ENCODER_CALL` = True / False forces either): NLVR logits, every layer's pruning record, the
    summed att_ft of both encoders; padded captions so that the text side prunes and compacts its mask as well."""
    from madtp_amd import bert, build, harness, hip, runtime, vit
    build.build(verbose=False)
    hip.load()
    model = harness.build_nlvr(224, 0, "cuda")
    images, text, targets = harness.nlvr_inputs(3, 224, 35, seed=7, pad_tail=9)
    results = []
    for enc_call in (True, False):
        monkeypatch.setattr(vit, "_ENCODER_CALL", enc_call)
        monkeypatch.setattr(bert, "_ENCODER_CALL", enc_call)
        with runtime.precision(mode):
            logits, trace = harness.run_nlvr(model, images, text, targets, 30.0)
        sd_img, sd_txt = model.last_sd_ft
        torch.cuda.synchronize()
        results.append((logits.clone(), trace, sd_img.clone(), sd_txt.clone()))
    (la, ta, ia, xa), (lb, tb, ib, xb) = results
    assert torch.equal(la, lb)
    assert torch.equal(ia, ib) and torch.equal(xa, xb)
    pruned_layers = 0
    for side in ("vit", "text"):
        for a, b in zip(ta[side], tb[side]):
            assert (a is None) == (b is None)
            if a is None:
                continue
            assert a["k"] == b["k"] and a["pruned"] == b["pruned"]
            assert torch.equal(a["score"], b["score"]) and torch.equal(a["threshold"], b["threshold"]) and torch.equal(a["count"], b["count"])
            if a["pruned"]:
                pruned_layers += 1
                assert torch.equal(a["indices"], b["indices"]) and torch.equal(a["indices_sort"], b["indices_sort"])
    assert pruned_layers >= 6


@pytest.mark.parametrize("mode", ["fp32", "f16x3", "bf16", "f16"])
@pytest.mark.parametrize("B,T,size", [(3, 30.0, 224), (1, 5.0, 224), (8, 8.6, 224), (2, 0.02, 48), (2, 3.0, 64)])
def test_sync_free_vit_encoder_equals_host_k_paths(mode, B, T, size, monkeypatch):
    """SURVEY 8(f) rank 2, device-side lengths: madtp_vit_encoder_async (the whole ViT enqueued without a host read of k - every
    kernel takes its token count from the device-side record token_score leaves) against madtp_vit_encoder (k handed to the host
    per layer) and the per-layer path: bit-identical encoder output, att_ft sum and pruning records (scores, thresholds, counts,
    kept indices, full sort order) in all precision modes; the 48 x 48 and 64 x 64 images (10 / 17 tokens) drive layers into the
    not-pruned branch of vit.py:148-149 (k >= n - 1: the sync-free path then runs its gather as a copy)."""
    from madtp_amd import build, harness, hip, runtime, vit
    build.build(verbose=False)
    hip.load()
    model = harness.build_nlvr(224, 0, "cuda")
    images, _, _ = harness.nlvr_inputs(B, 224, 20, seed=5)
    images = images[:, :, :size, :size].contiguous()
    venc = model.visual_encoder
    outs = []
    for enc_call, sync_free in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(vit, "_ENCODER_CALL", enc_call)
        monkeypatch.setattr(vit, "_SYNC_FREE", sync_free)
        with runtime.precision(mode), torch.no_grad():
            y, sd = venc(images[:B], space_dict=model.space_dict, temperature=T)
            recs = [dict(blk.last_prune.items()) if blk.last_prune is not None else None for blk in venc.blocks]
        torch.cuda.synchronize()
        outs.append((y.clone(), sd.clone(), [None if r is None else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in r.items()} for r in recs]))
    ref = outs[0]
    pruned = unpruned = 0
    for y, sd, recs in outs[1:]:
        assert y.shape == ref[0].shape and torch.equal(y, ref[0]) and torch.equal(sd, ref[1])
        for a, b in zip(recs, ref[2]):
            assert a["k"] == b["k"] and a["pruned"] == b["pruned"]
            assert torch.equal(a["score"], b["score"]) and torch.equal(a["threshold"], b["threshold"]) and torch.equal(a["count"], b["count"])
            if a["pruned"]:
                pruned += 1
                assert torch.equal(a["indices"], b["indices"]) and torch.equal(a["indices_sort"], b["indices_sort"])
            else:
                unpruned += 1
    assert (unpruned > 0) if size < 224 else (pruned >= 12)


def _force_sync_free(monkeypatch):
    """Both encoders on the encoder-level call with device-side lengths (MADTP_ENCODER_CALL=1 MADTP_ENCODER_SYNC_FREE=1)."""
    from madtp_amd import bert, vit
    monkeypatch.setattr(vit, "_ENCODER_CALL", True)
    monkeypatch.setattr(bert, "_ENCODER_CALL", True)
    monkeypatch.setattr(vit, "_SYNC_FREE", True)


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_sync_free_encoders_match_reference_fixture(env, path, mode, monkeypatch):
    """SURVEY 8(f) rank 2 pinned by the REFERENCE's recordings, not by the host-k path: BLIP_NLVR end to end with
    madtp_vit_encoder_async AND madtp_bert_encoder_async (no host read of k inside either encoder: one read of the device-side
    records per encoder, at its end) reproduces every fixture's per-layer token counts, kept-token sets of both encoders (the
    padded-caption fixture prunes the text 35 -> 8 with mask compaction on the device) and the logits within 1e-3."""
    from madtp_amd import hip
    harness, runtime, model = env
    _force_sync_free(monkeypatch)
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    images, text, targets = harness.nlvr_inputs(B, size, L, seed, pad_tail=int(g["pad_tail"]) if "pad_tail" in g.files else 0)
    calls = {"vit": 0, "bert": 0}
    real_v, real_b = hip.vit_encoder, hip.bert_encoder
    def spy_v(*a, **kw):
        calls["vit"] += bool(kw.get("sync_free"))
        return real_v(*a, **kw)
    def spy_b(*a, **kw):
        calls["bert"] += bool(kw.get("sync_free"))
        return real_b(*a, **kw)
    monkeypatch.setattr(hip, "vit_encoder", spy_v)
    monkeypatch.setattr(hip, "bert_encoder", spy_b)
    with runtime.precision(mode):
        logits, trace = harness.run_nlvr(model, images, text, targets, T)
    assert calls["vit"] >= 1 and calls["bert"] >= 1, calls  # both encoders really took the device-side-length entry points
    assert harness.token_lengths(trace["vit"], 197) == g["vit_lens"].tolist()
    assert harness.token_lengths(trace["text"], L) == g["txt_lens"].tolist()
    for side, key, n0 in (("vit", "vit", 196), ("text", "txt", L - 1)):
        mine = harness.compose_ids(trace[side], n0)
        ref = _golden_sets(g, key, 2 * B, n0)
        for l in range(12):
            assert (mine[l] is None) == (ref[l] is None), (side, l)
            if mine[l] is not None:
                assert mine[l] == ref[l], f"{side} layer {l}: kept-token sets differ from the reference"
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-3


@pytest.mark.parametrize("mode", EXACT_MODES + ["f16"])
@pytest.mark.parametrize("path", MED_CASES, ids=[os.path.basename(c)[:-4] for c in MED_CASES])
def test_sync_free_med_encoder_matches_reference_fixture(path, mode, monkeypatch):
    """madtp_bert_encoder_async on models/med.py's BertModel (text mode and multimodal mode with single cross-attention, ragged
    padding masks, text pruned 35 -> 7 with the MED mask rule - kept indices, then the (k+1)-th ranked token) against the
    reference fixture: token counts, kept sets, CLS row within 1e-3 (parity modes); finite and close in the f16 fast mode."""
    from madtp_amd import build, hip, harness, runtime, specs
    from madtp_amd.med import BertConfig, BertModel
    from tests.test_oracle_golden import med_inputs
    build.build(verbose=False)
    hip.load()
    _force_sync_free(monkeypatch)
    g = np.load(path)
    ids, att, enc, enc_att, sd, bert_mode, T = med_inputs(g)
    model = BertModel(BertConfig.med_default(), add_pooling_layer=False)
    model.load_state_dict(specs.synth_weights(specs.bert_shapes("", "med"), int(g["seed"])), strict=False)
    model = model.eval().cuda()
    used = []
    real_b = hip.bert_encoder
    monkeypatch.setattr(hip, "bert_encoder", lambda *a, **kw: (used.append(bool(kw.get("sync_free"))), real_b(*a, **kw))[1])
    with runtime.precision(mode), torch.no_grad():
        out, _ = model(ids.cuda(), attention_mask=att.cuda(), encoder_hidden_states=None if enc is None else enc.cuda(),
                       encoder_attention_mask=None if enc_att is None else enc_att.cuda(), mode=bert_mode,
                       space_dict=sd.cuda(), temperature=T)
    assert used == [True]
    hid = out.last_hidden_state
    if mode == "f16":
        assert torch.isfinite(hid).all() and list(hid.shape[::2]) == g["hidden_shape"].tolist()[::2]
        return
    assert list(hid.shape) == g["hidden_shape"].tolist()
    trace = [None if l.last_prune is None else {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in l.last_prune.items()}
             for l in model.encoder.layer]
    assert harness.token_lengths(trace, int(g["L"])) == g["txt_lens"].tolist()
    mine = harness.compose_ids(trace, int(g["L"]) - 1)
    ref_trace = [{"pruned": True, "indices": g[f"txt{l}_idx"][:, : trace[l]["k"]]} if f"txt{l}_idx" in g.files else None
                 for l in range(12)]
    assert mine == harness.compose_ids(ref_trace, int(g["L"]) - 1)
    assert np.abs(hid[:, 0, :32].cpu().numpy() - g["hidden_cls"]).max() < 1e-3


RETR_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "retr_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", RETR_CASES, ids=[os.path.basename(c)[:-4] for c in RETR_CASES])
def test_retrieval_itm_reranking_matches_reference_fixture(path, mode):
    """blip_retrieval.evaluate() mirror vs the score matrices the reference's own evaluate() produced:
    fp32 mode re-ranks the same candidates with the same scores (incl. the cross-batch CLS-repeat padding and the text-side
    pruning inside the k_test-pair multimodal batches); bf16 mode stays close; rank slicing (2 ranks) sums to the whole."""
    from madtp_amd import build, hip, harness, runtime
    from madtp_amd import blip_retrieval as br
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    n_img, img_bs, n_txt, size = int(g["n_img"]), int(g["img_bs"]), int(g["n_txt"]), int(g["size"])
    T, k_test, seed = float(g["temperature"]), int(g["k_test"]), int(g["seed"])
    model = harness.build_retrieval(size, seed)
    batches, ids, att = harness.retrieval_inputs(n_img, img_bs, n_txt, size, 35, seed, device="cuda")
    loader = harness.RetrievalLoader(batches, ids, att)
    cfg = {"k_test": k_test}
    with runtime.precision(mode):
        i2t, t2i, _ = br.evaluate(model, loader, torch.device("cuda"), cfg, T)
        parts = [br.evaluate(model, loader, torch.device("cuda"), cfg, T, rank=r, world_size=2) for r in range(2)]
    for ours, ref in ((i2t, g["score_i2t"]), (t2i, g["score_t2i"])):
        assert ((ours == -100.0) == (ref == -100.0)).all()
        assert np.abs(ours - ref).max() < 1e-3
    # the reference all-reduces (SUM) matrices initialised to -100 on every rank: rows owned by a rank carry its scores
    for k, full in ((0, i2t), (1, t2i)):
        merged = np.where(parts[0][k] != -100.0, parts[0][k], parts[1][k])
        assert np.array_equal(merged, full)
        assert not ((parts[0][k] != -100.0) & (parts[1][k] != -100.0)).any()
    # the K/V cache (images projected once per layer, attention indexes the cache) changes no bit of the scores
    with runtime.precision(mode):
        n_i2t, n_t2i, _ = br.evaluate(model, loader, torch.device("cuda"), cfg, T, kv_cache=False)
    assert np.array_equal(n_i2t, i2t) and np.array_equal(n_t2i, t2i)
    with runtime.precision("bf16"):
        b_i2t, b_t2i, _ = br.evaluate(model, loader, torch.device("cuda"), cfg, T)
        c_i2t, c_t2i, _ = br.evaluate(model, loader, torch.device("cuda"), cfg, T, kv_cache=False)
    assert np.isfinite(b_i2t).all() and np.isfinite(b_t2i).all()
    # bf16 re-ranks (almost) the same candidate sets and scores them within bf16 error of the reference
    same = (b_i2t != -100.0) == (g["score_i2t"] != -100.0)
    both = (b_i2t != -100.0) & (g["score_i2t"] != -100.0)
    errb = np.abs(b_i2t - g["score_i2t"])[both].max()
    print(f"retrieval bf16: candidate agreement {same.mean():.4f}, max |dscore| {errb:.4f}")
    assert same.mean() > 0.95 and errb < 0.08   # measured 0.972 / 0.040
    assert np.array_equal(b_i2t, c_i2t) and np.array_equal(b_t2i, c_t2i)


CLIP_FULL_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clip_full_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", CLIP_FULL_CASES, ids=[os.path.basename(c)[:-4] for c in CLIP_FULL_CASES])
def test_clip_both_towers(path, mode):
    """CLIP mirror (clip/model.py CLIP.encode_image / encode_text, BASELINE config 4) built by build_model() from a state dict:
    vision tower vs the reference fixture (kept sets identical, features within 1e-3); causal text tower vs the oracle run in
    the HIP path's canonical (ascending) token order - kept sets per layer identical, features within 1e-3 - and vs the
    reference fixture where the order cannot matter yet (lengths, first pruned layer's kept set)."""
    from madtp_amd import build, hip, harness, runtime, specs, synth
    from madtp_amd import clip_model as cm
    from oracle import madtp_oracle as O
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    W = specs.synth_weights(specs.clip_shapes(size), seed)
    model = cm.build_model(dict(W), evaluate=True).eval().cuda()
    assert model.transformer.width == 512 and model.context_length == 77 and model.visual.input_resolution == size
    images = synth.synth_images(B, size, seed).cuda()
    text = synth.synth_clip_tokens(B, 77, seed, int(g["min_len"]), int(g["max_len"]))
    otr = []
    with torch.no_grad():
        ref_ft, ref_sd = O.clip_encode_text(W, text, W["space_dict"], T, order="ascending", trace=otr)

    def traces(blocks):
        return [None if b.last_prune is None else {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.last_prune.items()}
                for b in blocks]

    with runtime.precision(mode), torch.no_grad():
        fi, sd_i = model.encode_image(images, model.space_dict, T)
        vtr = traces(model.visual.transformer.resblocks)
        ft, sd_t = model.encode_text(text.cuda(), model.space_dict, T)
        ttr = traces(model.transformer.resblocks)
    # vision tower == reference
    assert harness.token_lengths(vtr, 197) == g["vit_lens"].tolist()
    ref_v = [{"pruned": True, "indices": g[f"vit{l}_idx"]} if f"vit{l}_idx" in g.files else None for l in range(12)]
    assert harness.compose_ids(vtr, 196) == harness.compose_ids(ref_v, 196)
    assert np.abs(fi.cpu().numpy() - g["image_features"]).max() < 1e-3
    # text tower == oracle in canonical order
    assert harness.token_lengths(ttr, 77) == harness.token_lengths(otr, 77)
    assert harness.compose_ids(ttr, 76) == O.compose_ids(otr, 76)
    for l, info in enumerate(ttr):  # canonical order: kept ids ascend
        if info is not None and info["pruned"]:
            idx = info["indices"].numpy()
            assert (np.diff(idx, axis=1) > 0).all()
    assert (ft.cpu() - ref_ft).abs().max().item() < 1e-3
    assert (sd_t.cpu() - ref_sd).abs().max().item() < 1e-3 * max(1.0, ref_sd.abs().max().item())
    # ... and == reference up to the first pruned layer (same input there, so the same kept SET and the same k)
    first = next(l for l in range(12) if f"txt{l}_idx" in g.files)
    assert harness.token_lengths(ttr, 77)[: first + 1] == g["txt_lens"].tolist()[: first + 1]
    assert (np.sort(ttr[first]["indices"].numpy(), 1) == np.sort(g[f"txt{first}_idx"], 1)).all()
    with runtime.precision("bf16"), torch.no_grad():
        fb, _ = model.encode_text(text.cuda(), model.space_dict, T)
        ib, _ = model.encode_image(images, model.space_dict, T)
    cos_t = torch.nn.functional.cosine_similarity(fb.float(), ft.float(), dim=-1).min().item()
    cos_i = torch.nn.functional.cosine_similarity(ib.float(), fi.float(), dim=-1).min().item()
    print(f"CLIP bf16 vs {mode}: min cosine text {cos_t:.4f} image {cos_i:.4f}")
    assert cos_t > 0.98 and cos_i > 0.98


VQA_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vqa[0-9]*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", VQA_CASES, ids=[os.path.basename(c)[:-4] for c in VQA_CASES])
def test_vqa_encoder_leg_matches_reference_fixture(path, mode):
    """BLIP_VQA mirror, encoder leg (models/blip_vqa.py:59-64,118-125; BASELINE config 5: 480^2 images = 901 visual tokens,
    MED multimodal encoder over the pruned image tokens) vs the fixture recorded from the reference's modules: kept sets of both
    encoders identical, question states within 1e-3."""
    from madtp_amd import build, hip, harness, runtime, specs
    from madtp_amd.blip_vqa import BLIP_VQA
    from tests.test_oracle_golden import vqa_inputs
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    T, L = float(g["temperature"]), int(g["L"])
    model = BLIP_VQA(image_size=int(g["size"]), evaluate=True, decoder=False)
    msg = model.load_state_dict(specs.synth_weights(specs.blip_vqa_shapes(int(g["size"])), int(g["seed"])), strict=False)
    assert not msg.unexpected_keys and all("query_model" in k or "position_ids" in k for k in msg.missing_keys), msg
    model = model.eval().cuda()
    images, ids, att = vqa_inputs(g)
    with runtime.precision(mode), torch.no_grad():
        hid = model(images.cuda(), {"input_ids": ids.cuda(), "attention_mask": att.cuda()}, temperature=T, train=False)

    def traces(layers):
        return [None if l.last_prune is None else {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in l.last_prune.items()}
                for l in layers]
    vtr, ttr = traces(model.visual_encoder.blocks), traces(model.text_encoder.encoder.layer)
    assert list(hid.shape) == g["hidden_shape"].tolist()
    assert harness.token_lengths(vtr, 901) == g["vit_lens"].tolist()
    assert harness.token_lengths(ttr, L) == g["txt_lens"].tolist()
    for key, tr, n0 in (("vit", vtr, 900), ("txt", ttr, L - 1)):
        ref = [{"pruned": True, "indices": g[f"{key}{l}_idx"][:, : tr[l]["k"]]} if f"{key}{l}_idx" in g.files else None
               for l in range(12)]
        assert harness.compose_ids(tr, n0) == harness.compose_ids(ref, n0), key
    assert np.abs(hid[:, 0, :32].cpu().numpy() - g["hidden_cls"]).max() < 1e-3
    with runtime.precision("bf16"), torch.no_grad():
        hb = model(images.cuda(), {"input_ids": ids.cuda(), "attention_mask": att.cuda()}, temperature=T, train=False)
    cos = torch.nn.functional.cosine_similarity(hb[:, 0, :].float(), hid[:, 0, :].float(), dim=-1).min().item()
    print(f"VQA bf16 vs {mode} CLS row: min cosine {cos:.4f}")
    assert torch.isfinite(hb).all() and cos > 0.97  # measured 0.9996 (T=6) / 0.981 (T=30: 901 -> 11 image tokens)


VQA_RANK_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vqa_rank_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES + ["bf16"])
@pytest.mark.parametrize("path", VQA_RANK_CASES, ids=[os.path.basename(c)[:-4] for c in VQA_RANK_CASES])
def test_vqa_rank_answer_matches_reference_fixture(path, mode):
    """SURVEY.md 8(f) rank 4, inference half: BLIP_VQA.forward(train=False, inference='rank') on the HIP path - encoder leg,
    teacher-forced answer decoder (BertLMHeadModel: causal self-attention through madtp_bert_layer_w.self_mask_qk, LM head,
    madtp_lm_loss) and rank_answer (madtp_token_prob, cached cross-attention K/V of the question states) - vs the fixture
    recorded from the reference's own models/blip_vqa.py.  Parity modes: the same candidate sets, the same answers, first-token
    probabilities within 1e-5 and sequence log-likelihoods within 1e-3 RELATIVE of the reference's; bf16: finite and close."""
    from madtp_amd import build, hip, runtime, specs
    from madtp_amd.blip_vqa import BLIP_VQA
    from tests.test_oracle_golden import vqa_rank_inputs
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    T, k = float(g["temperature"]), int(g["k_test"])
    model = BLIP_VQA(image_size=int(g["size"]), evaluate=True)
    sd = specs.synth_weights(specs.blip_vqa_shapes(int(g["size"]), decoder=True), int(g["seed"]))
    msg = model.load_state_dict(sd, strict=False)
    assert not msg.unexpected_keys and all("query_model" in x or "position_ids" in x for x in msg.missing_keys), msg
    dec = model.text_decoder
    assert dec.cls.predictions.decoder.weight is dec.bert.embeddings.word_embeddings.weight   # tied (transformers tie_weights)
    assert sorted(x for x in model.state_dict() if x.startswith("text_decoder.") and "position_ids" not in x) == \
        sorted(str(x) for x in g["decoder_state_dict_keys"] if "position_ids" not in str(x))
    model = model.eval().cuda()
    images, ids, att, a_ids, a_att = vqa_rank_inputs(g)
    det = {}
    with runtime.precision(mode), torch.no_grad():
        q = {"input_ids": ids.cuda(), "attention_mask": att.cuda()}
        qs, _, _ = model.encode_question(images.cuda(), q, T)
        max_ids = model.rank_answer(qs, att.cuda(), a_ids.cuda(), a_att.cuda(), k, detail=det)
        again = model(images.cuda(), q, {"input_ids": a_ids.cuda(), "attention_mask": a_att.cuda()}, temperature=T, train=False,
                      inference='rank', k_test=k)
    assert torch.equal(max_ids, again)
    if mode == "fp32":
        # grad mode ON (a caller that forgot no_grad; parameters carry requires_grad=True by default): rank_answer runs against the
        # cached K/V of the question states - an inference call, not the autograd route (it raised AssertionError)
        with runtime.precision(mode):
            with torch.no_grad():
                qs2, _, _ = model.encode_question(images.cuda(), q, T)
            assert torch.equal(model.rank_answer(qs2, att.cuda(), a_ids.cuda(), a_att.cuda(), k), max_ids)
    lp, ref_lp = det["log_probs_sum"].cpu().numpy(), g["log_probs_sum"]
    pf, ref_pf = det["prob_first_token"].cpu().numpy(), g["prob_first_token"]
    if mode == "bf16":
        assert np.isfinite(lp).all() and np.abs(pf - ref_pf).max() < 2e-4
        print(f"bf16 rank_answer: max |d log-lik| {np.abs(np.sort(lp, 1) - np.sort(ref_lp, 1)).max():.3f}, answers {max_ids.tolist()} "
              f"(reference {g['max_ids'].tolist()})")
        return
    assert np.abs(det["first_logits"][:, :64].cpu().numpy() - g["first_logits_sample"]).max() < 1e-3
    assert np.abs(pf - ref_pf).max() < 1e-5
    assert [sorted(r) for r in det["topk_ids"].tolist()] == [sorted(r) for r in g["topk_ids"].tolist()]
    # same candidates possibly in another order: compare per (question, answer id)
    mine = {(qi, int(a)): lp[qi, j] for qi in range(lp.shape[0]) for j, a in enumerate(det["topk_ids"][qi].tolist())}
    ref = {(qi, int(a)): ref_lp[qi, j] for qi in range(ref_lp.shape[0]) for j, a in enumerate(g["topk_ids"][qi].tolist())}
    assert max(abs(mine[x] - ref[x]) / max(1.0, abs(ref[x])) for x in ref) < 1e-3
    assert max_ids.tolist() == g["max_ids"].tolist()


@pytest.mark.parametrize("mode", ["fp32", "f16x3", "f16"])
def test_incremental_decoding_equals_full_prefix_generate(mode, monkeypatch):
    """BertLMHeadModel.generate through madtp_bert_decode_step (one new token per beam and step against the layers' self-attention
    K/V cache, beams re-ordered by a gather over the cache rows: models/med.py:1071-1094) against the full-prefix path
    (MADTP_DECODE_CACHE=0: the decoder re-run over the whole prefix with the causal mask every step): the same generated ids, and
    the per-step scores of the two paths agree to rounding (the projections' split-K factors differ with the row count)."""
    from madtp_amd import build, hip, runtime, specs
    from madtp_amd.med import BertConfig, BertLMHeadModel
    build.build(verbose=False)
    hip.load()
    model = BertLMHeadModel(BertConfig.med_default())
    sd = specs.synth_weights(specs.bert_shapes("bert.", "med"), 3)
    model.load_state_dict(sd, strict=False)
    model.tie_weights()
    model = model.cuda().eval()
    model.tie_weights()
    g = torch.Generator().manual_seed(5)
    B, nb, Nq = 5, 3, 23
    enc = torch.randn(B, Nq, 768, generator=g).cuda()
    prompt = torch.tensor([[30522, 1037, 3861, 1997]] * B)  # [BOS] "a picture of" (models/blip.py:84-86)
    outs, scores = {}, {}
    for cached in ("1", "0"):
        monkeypatch.setenv("MADTP_DECODE_CACHE", cached)
        rec = []
        real = model.prediction_scores
        monkeypatch.setattr(model, "prediction_scores", lambda h, _r=real, _rec=rec: (lambda o: (_rec.append(o[1][:, 0, :64].float().cpu()), o)[1])(_r(h)))
        with runtime.precision(mode), torch.no_grad():
            outs[cached] = model.generate(input_ids=prompt, max_length=18, min_length=6, num_beams=nb, eos_token_id=102, pad_token_id=0,
                                          repetition_penalty=1.0, encoder_hidden_states=enc.repeat_interleave(nb, dim=0),
                                          encoder_attention_mask=torch.ones(B * nb, Nq, dtype=torch.long).cuda()).cpu()
        scores[cached] = rec
        monkeypatch.setattr(model, "prediction_scores", real)
    assert torch.equal(outs["1"], outs["0"]), (outs["1"], outs["0"])
    assert len(scores["1"]) == len(scores["0"]) >= 6
    tol = 2e-3 if mode != "f16" else 6e-2
    for a, b in zip(scores["1"], scores["0"]):
        assert (a - b).abs().max().item() < tol * max(1.0, b.abs().max().item())


def test_nucleus_sampling_step_equals_oracle_and_generate_samples():
    """Missing item of the round-4 review: `generate(do_sample=True, top_p=0.9)` (models/blip.py:175-186).  (1) madtp_sample_top_p
    against oracle.sample_step (transformers 4.15's processors and warpers restated, checked on CPU against the installed library's
    warpers) on random score rows at the BLIP vocabulary and on a peaked toy row: the SAME token for the same uniform numbers, with
    and without repetition penalty / EOS suppression; (2) the draw frequencies of one row follow the nucleus distribution; (3)
    BertLMHeadModel.generate(do_sample=True) and BLIP_Decoder.generate(sample=True) run end to end: same ids for the same generator
    seed, different ids for another, finished rows are padded, every sampled token lies inside its step's nucleus."""
    from madtp_amd import build, hip, runtime, specs
    from madtp_amd.med import BertConfig, BertLMHeadModel
    from oracle import madtp_oracle as O
    build.build(verbose=False)
    hip.load()
    g = torch.Generator().manual_seed(2)
    for V, scale in ((30524, 2.0), (30524, 0.3), (512, 4.0)):
        Vp = (V + 7) // 8 * 8
        logits = torch.randn(6, Vp, generator=g) * scale
        ids = torch.randint(0, V, (6, 9), generator=g)
        ids[:, 5] = ids[:, 2]
        u = torch.rand(6, generator=g)
        for pen, sup in ((1.0, -1), (1.1, -1), (1.1, 102)):
            ref = O.sample_step(logits[:, :V], ids, u, 0.9, 50, pen, sup)
            got, prob = hip.sample_top_p(logits.cuda(), u.cuda(), V, 0.9, top_k=50, suppress_token=sup,
                                         prev_ids=ids.cuda() if pen != 1.0 else None, repetition_penalty=pen, want_prob=True)
            assert got.cpu().tolist() == ref.tolist(), (V, scale, pen, sup)
            assert ((prob > 0) & (prob <= 1.0 + 1e-6)).all()
    # ties AT the k-th best score (quantised scores: dozens of equal values): TopKLogitsWarper keeps every tie (it removes `scores <
    # kth`), and which ones survive must not depend on scheduling - repeated launches give the oracle's tokens every time, also when
    # the ties outnumber the kernel's 128 survivor slots (the strictly better scores and the lowest-index ties fill them)
    for V, levels, top_p in ((30524, 40, 0.9), (2048, 40, 0.999), (2048, 40, 0.9), (1024, 16, 0.95)):
        Vp = (V + 7) // 8 * 8
        logits = torch.randint(0, levels, (8, Vp), generator=g).float() * 0.25
        logits[:, :20] += torch.rand(8, 20, generator=g) + levels * 0.25  # twenty distinct leaders, then plateaus of equal scores
        u = torch.rand(8, generator=g) * 0.999
        ref = O.sample_step(logits[:, :V], torch.zeros(8, 1, dtype=torch.long), u, top_p, 50, 1.0, -1)
        runs = [hip.sample_top_p(logits.cuda(), u.cuda(), V, top_p, top_k=50).cpu().tolist() for _ in range(5)]
        assert all(r == runs[0] for r in runs), "tie handling depends on arrival order"
        kth = torch.topk(logits[:, :V], 50)[0][:, -1]
        n_ge = (logits[:, :V] >= kth[:, None]).sum(1)
        if int(n_ge.max()) <= 128:  # every survivor fits the kernel's slots: the library's result exactly
            assert runs[0] == ref.tolist(), (V, levels, top_p, n_ge.tolist())
        else:  # more ties than slots: deterministic, and every draw is a survivor of the library's filter
            w = O.nucleus_filter(logits[:, :V], top_p, 50)
            assert all(w[b, t] > -float("inf") for b, t in enumerate(runs[0]))
    # (2) frequencies on one peaked row
    row = (torch.randn(1, 64, generator=g) * 2)
    w = O.nucleus_filter(row, 0.9, 50).softmax(-1)[0]
    n = 20000
    u = (torch.arange(n, dtype=torch.float32) + 0.5) / n
    draws = hip.sample_top_p(row.cuda().expand(n, 64).contiguous(), u.cuda(), 64, 0.9).cpu()
    freq = torch.bincount(draws, minlength=64).double() / n
    assert (freq - w.double()).abs().max().item() < 2e-3
    # (3) end to end
    model = BertLMHeadModel(BertConfig.med_default())
    model.load_state_dict(specs.synth_weights(specs.bert_shapes("bert.", "med"), 3), strict=False)
    model.tie_weights()
    model = model.cuda().eval()
    model.tie_weights()
    B, Nq = 6, 23
    enc = torch.randn(B, Nq, 768, generator=g).cuda()
    prompt = torch.tensor([[30522, 1037, 3861, 1997]] * B)
    outs = []
    for seed in (7, 7, 8):
        gen = torch.Generator(device="cuda").manual_seed(seed)
        with runtime.precision("f16"), torch.no_grad():
            outs.append(model.generate(input_ids=prompt, max_length=16, min_length=6, do_sample=True, top_p=0.9, num_return_sequences=1,
                                       eos_token_id=102, pad_token_id=0, repetition_penalty=1.1, encoder_hidden_states=enc,
                                       encoder_attention_mask=torch.ones(B, Nq, dtype=torch.long).cuda(), generator=gen).cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    o = outs[0]
    assert o.shape[0] == B and 6 <= o.shape[1] <= 16 and torch.equal(o[:, :4], prompt)
    for b in range(B):  # after EOS only padding
        pos = (o[b] == 102).nonzero().flatten()
        if len(pos):
            assert pos[0] >= 6 - 1 and (o[b, pos[0] + 1:] == 0).all()


def test_generation_beam_search_equals_oracle_on_toy_models():
    """madtp_amd.generation.beam_search (madtp_beam_topk + the hypothesis book-keeping, device- and host-side) against oracle.beam_search
    (transformers 4.15 restated) on first-order toy language models with random tables: EOS frequent enough that hypotheses
    finish at every length, min_length in play, items finishing at different steps (padding), the hand-worked cases of
    tests/test_oracle_golden.py included."""
    from madtp_amd import build, generation, hip
    from oracle import madtp_oracle as O
    from tests.test_oracle_golden import TOY, toy_lm
    build.build(verbose=False)
    hip.load()

    def gpu_lm(table, ld):
        lt = torch.full((len(table), ld), 1e9)
        lt[:, :len(table[0])] = torch.log(torch.tensor(table, dtype=torch.float32))
        lt = lt.cuda()
        return lambda ids: lt[ids[:, -1]]
    out = generation.beam_search(gpu_lm(TOY, 8), torch.tensor([[2], [2], [4], [4]]).cuda(), 2, 6, 0, 1, 0, 5)
    assert out.tolist() == [[2, 3, 1], [4, 1, 0]]
    gen = torch.Generator().manual_seed(5)
    n_eos = 0
    for trial in range(24):
        V = [11, 37, 200][trial % 3]
        nb = 2 + trial % 3
        B = 1 + trial % 4
        table = torch.rand(V, V, generator=gen) ** 4 + 1e-4
        table[:, 1] *= (2.0 + 6.0 * torch.rand(V, generator=gen)) * (V / 11.0)   # EOS competitive
        table = (table / table.sum(1, keepdim=True)).tolist()
        prompt = torch.randint(2, V, (B, 1 + trial % 2), generator=gen).repeat_interleave(nb, dim=0)
        kw = dict(num_beams=nb, max_length=4 + trial % 5, min_length=trial % 3, eos_token_id=1, pad_token_id=0)
        # every other trial with the library's repetition penalty (generate(repetition_penalty=...), models/blip.py:161,195):
        # tokens already in a beam's sequence have their log-probability scaled (madtp_beam_topk_penalty)
        rp = [1.0, 1.3, 0.8, 2.0][trial % 4]
        # hypothesis scoring / stopping rules beyond the call sites' defaults (BeamHypotheses.add, is_done)
        lp, es = [1.0, 0.7, 1.6][trial % 3], trial % 5 == 4
        ref = O.beam_search(toy_lm(table), prompt, repetition_penalty=rp, length_penalty=lp, early_stopping=es, **kw)
        # the book-keeping on the device (madtp_beam_update, the default) and on the host (MADTP_BEAM_DEVICE=0)
        for dev_side in ("1", "0"):
            os.environ["MADTP_BEAM_DEVICE"] = dev_side
            try:
                mine = generation.beam_search(gpu_lm(table, (V + 3) // 4 * 4), prompt.cuda(), kw["num_beams"], kw["max_length"],
                                              kw["min_length"], 1, 0, V, repetition_penalty=rp, length_penalty=lp, early_stopping=es)
            finally:
                del os.environ["MADTP_BEAM_DEVICE"]
            assert mine.cpu().tolist() == ref.tolist(), (trial, dev_side, rp, lp, es, mine.tolist(), ref.tolist())
        n_eos += int((ref == 1).any())
    assert n_eos >= 8   # the finishing rules were exercised


VQA_GEN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "vqa_gen_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES + ["bf16"])
@pytest.mark.parametrize("path", VQA_GEN_CASES, ids=[os.path.basename(c)[:-4] for c in VQA_GEN_CASES])
def test_vqa_generate_matches_oracle_and_reference_fixture(path, mode):
    """BLIP_VQA.forward(train=False, inference='generate') on the HIP path (blip_vqa.py:127-147: encoder leg, question states
    cached as cross-attention K/V per item, three beams, max_length 10, min_length 1, decoder over the whole prefix per step,
    LM head at the last position, madtp_beam_topk) vs the oracle's restatement on the same inputs - identical token sequences in
    the parity modes, also where [SEP] is biased into the candidate set (hypotheses finish early) - and vs the sequences the
    reference itself produced where no finished hypothesis decides (see tests/test_oracle_golden.py)."""
    from madtp_amd import build, hip, runtime, specs
    from madtp_amd.blip_vqa import BLIP_VQA
    from oracle import madtp_oracle as O
    from tests.test_oracle_golden import VQA_GEN_COMPARABLE, vqa_gen_weights, vqa_inputs
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    name = os.path.basename(path)[:-4]
    T = float(g["temperature"])
    W = vqa_gen_weights(g)
    model = BLIP_VQA(image_size=int(g["size"]), evaluate=True)
    msg = model.load_state_dict(W, strict=False)
    assert not msg.unexpected_keys, msg
    model = model.eval().cuda()
    images, ids, att = vqa_inputs(g)
    with runtime.precision(mode), torch.no_grad():
        out = model(images.cuda(), {"input_ids": ids.cuda(), "attention_mask": att.cuda()}, None, temperature=T, train=False,
                    inference='generate')
    out = out.cpu()
    assert out.shape[0] == int(g["B"]) and out.shape[1] <= 10 and (out[:, 0] == 30522).all()
    if mode == "bf16":
        print(f"bf16 generate {name}: {out.tolist()}")
        return
    with torch.no_grad():
        ref = O.blip_vqa_generate_forward(W, images, ids, att, T)
    assert out.tolist() == ref.tolist(), (out.tolist(), ref.tolist())
    for b in VQA_GEN_COMPARABLE[name]:
        assert out[b].tolist() == g["sequences"][b].tolist()[:out.shape[1]]


CAP_GEN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cap_gen_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES + ["bf16"])
@pytest.mark.parametrize("path", CAP_GEN_CASES, ids=[os.path.basename(c)[:-4] for c in CAP_GEN_CASES])
def test_caption_generate_matches_reference_fixture(path, mode):
    """BLIP_Decoder.generate(sample=False, num_beams=3) on the HIP path (models/blip.py:161-196, the evaluation call of
    compress_caption_dtp.py:86): pruned ViT, image tokens cached as cross-attention K/V per image, beam search over the MED
    decoder - the token sequences the reference itself generated (parity modes), the pruned ViT's token counts included."""
    from madtp_amd import build, hip, runtime, synth
    from madtp_amd.blip import BLIP_Decoder
    from tests.test_oracle_golden import cap_gen_weights
    build.build(verbose=False)
    hip.load()
    g = np.load(path)
    model = BLIP_Decoder(image_size=int(g["size"]), evaluate=True)
    msg = model.load_state_dict(cap_gen_weights(g), strict=False)
    assert not msg.unexpected_keys and all("query_model" in x or "position_ids" in x for x in msg.missing_keys), msg
    assert sorted(k for k in model.state_dict() if "position_ids" not in k) == \
        sorted(str(k) for k in g["state_dict_keys"] if "position_ids" not in str(k))
    model = model.eval().cuda()
    images = synth.synth_images(int(g["B"]), int(g["size"]), int(g["seed"]))
    with runtime.precision(mode), torch.no_grad():
        out = model.generate(images.cuda(), sample=False, num_beams=int(g["num_beams"]), max_length=int(g["max_length"]),
                             min_length=int(g["min_length"]), temperature=float(g["temperature"]))
    out = out.cpu()
    if mode == "bf16":
        assert out.shape[0] == int(g["B"]) and out[:, :4].tolist() == g["sequences"][:, :4].tolist()
        print(f"bf16 caption ids: {out.tolist()}")
        return
    from madtp_amd import harness
    vtr = [None if b.last_prune is None else {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.last_prune.items()}
           for b in model.visual_encoder.blocks]
    assert harness.token_lengths(vtr, (int(g["size"]) // 16) ** 2 + 1) == g["vit_lens"].tolist()
    assert out.tolist() == g["sequences"].tolist()


NLVR_PAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nlvrpad_*.npz")))


@pytest.mark.parametrize("mode", EXACT_MODES)
@pytest.mark.parametrize("path", NLVR_PAD_CASES, ids=[os.path.basename(c)[:-4] for c in NLVR_PAD_CASES])
def test_nlvr_pad_inside_topk_pairing(env, path, mode):
    """Ragged captions: padded tokens rank inside the short captions' top-(k+1) (nlvr_encoder.py:440-452), where the reference's
    token / mask pairing depends on torch.topk(sorted=False)'s implementation-defined order.  The HIP path implements the
    documented canonical pairing (tokens ascending, mask entries in indices_sort order) = the oracle with
    text_order="ascending": identical kept sets on EVERY layer and logits within 1e-3; against the recording of the reference
    itself the kept sets agree up to and including the first layer with a pad inside the top-(k+1)."""
    from madtp_amd import specs, synth
    from oracle import madtp_oracle as O
    from tests.test_oracle_golden import nlvr_pad_layers
    harness, runtime, model = env
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    pads = g["pad_list"].tolist()
    images, text, targets = harness.nlvr_inputs(B, size, L, seed, pad_tail=pads)
    W = specs.synth_weights(specs.blip_nlvr_shapes(size), 0)
    tr = {}
    with torch.no_grad():
        ref = O.blip_nlvr_forward(W, images.cpu(), text["input_ids"].cpu(), text["attention_mask"].cpu(), T, trace=tr,
                                  text_order="ascending")
    with runtime.precision(mode):
        logits, trace = harness.run_nlvr(model, images, text, targets, T)
    for side, n0 in (("vit", 196), ("text", L - 1)):
        assert harness.compose_ids(trace[side], n0) == O.compose_ids(tr[side], n0), f"{side}: kept sets differ from the oracle"
    assert (logits.cpu() - ref).abs().max().item() < 1e-3
    if seed == 0:  # the env model carries the seed-0 weights the fixture was recorded with
        first = nlvr_pad_layers(g)[0]
        mine, rec = harness.compose_ids(trace["text"], L - 1), _golden_sets(g, "txt", B, L - 1)
        for l in range(first + 1):
            assert mine[l] == rec[l], l
        assert harness.token_lengths(trace["vit"], 197) == g["vit_lens"].tolist()
        if len(nlvr_pad_layers(g)) == 1:
            # pads inside the top-(k+1) at ONE layer only (nlvrpad_b3_T20_one): there the ascending pairing mis-pairs nothing and
            # neither does the recording's order (tests/test_oracle_golden.py prints the slots) - the reference's OWN result is
            # reproduced: every layer's kept sets and the recorded logits
            assert mine == rec
            assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-3


def test_block_sees_reassigned_parameters_and_replaced_submodules():
    """The collected parameter lists behind Block._weights() follow a re-assigned Parameter and a replaced sub-module (the
    registration hooks of madtp_amd.runtime, filtered to modules that have collected their lists)."""
    from madtp_amd import build, hip, runtime, vit
    build.build(verbose=False)
    hip.load()
    torch.manual_seed(0)
    blk = vit.Block(768, 12, qkv_bias=True).cuda().eval()
    x = torch.randn(2, 50, 768, device="cuda")
    with runtime.precision("fp32"), torch.no_grad():
        y0 = blk(x).clone()
        blk.mlp.fc1.weight = torch.nn.Parameter(torch.zeros_like(blk.mlp.fc1.weight))   # re-assignment, not an in-place change
        y1 = blk(x).clone()
        fresh = torch.nn.Linear(3072, 768).cuda()
        blk.mlp.fc2 = fresh                                                               # replaced sub-module
        y2 = blk(x).clone()
        torch.nn.Linear(8, 8)                                                             # unrelated module: no effect
        y3 = blk(x).clone()
    assert not torch.equal(y0, y1) and not torch.equal(y1, y2) and torch.equal(y2, y3)
    # fc1.weight = 0: mlp(x) = fc2(gelu(fc1.bias)) for every token
    with torch.no_grad():
        hb = torch.nn.functional.gelu(blk.mlp.fc1.bias)
        delta = fresh(hb)
    with runtime.precision("fp32"), torch.no_grad():
        blk.mlp.fc2 = torch.nn.Linear(3072, 768).cuda()
        torch.nn.init.zeros_(blk.mlp.fc2.weight); torch.nn.init.zeros_(blk.mlp.fc2.bias)
        y_attn = blk(x).clone()                                                           # mlp contributes exactly 0
    assert (y2 - (y_attn + delta)).abs().max().item() < 1e-4


def _rows_check(g, key, t, tol, what, perm=None):
    """t against a medopts_* record (shape, first 16 columns, row norms); perm [B, rows]: row r of the RECORD is row perm[b, r] of t"""
    import numpy as np
    t = t.detach().float().cpu()
    assert tuple(t.shape) == tuple(int(v) for v in g[key + "_shape"]), (what, key, tuple(t.shape), g[key + "_shape"])
    if perm is not None:
        t = torch.stack([t[b][perm[b]] for b in range(t.shape[0])])
    ref_sl, ref_nrm = torch.from_numpy(g[key + "_sl"]), torch.from_numpy(g[key + "_rownorm"]).float()
    err = (t[..., :16] - ref_sl).abs().max().item()
    nerr = ((t.norm(dim=-1) - ref_nrm).abs() / ref_nrm.clamp_min(1e-6)).max().item()
    assert err < tol and nerr < tol, (what, key, err, nerr)


@pytest.mark.parametrize("mode,tol,ptol", [("fp32", 2e-4, 2e-5), ("f16x3", 2e-4, 2e-5), ("f16", 6e-2, 2e-3)])
def test_bert_layer_signature_options_match_reference_fixture(mode, tol, ptol):
    """med.py:393-407 beyond the pruned-encoder call - `output_attentions=True`, `head_mask`, `past_key_value` on BertLayer.forward -
    against the reference's own recording of one layer call each (tests/golden/medopts_b2.npz, tools/make_golden.py::
    med_layer_options_case): layer outputs, self- and cross-attention probabilities, the returned cache, pruning decisions and the
    compacted mask.  Kept tokens come out in ascending order here and in topk order there: rows are compared token by token."""
    import glob
    import os
    import numpy as np
    from madtp_amd import build, hip, runtime
    from tests import grad_case
    build.build(verbose=False)
    hip.load()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "medopts_b2.npz"))
    c = grad_case.build_med(g)
    layer = grad_case.build_med_layer(c)
    for p in layer.parameters():
        p.requires_grad_(False)
    hid, mask, ta, T = c["hidden"].cuda(), c["add_mask"].cuda(), c["token_attn"].cuda(), c["T"]
    enc = c["enc"].cuda()
    B, L, D = hid.shape
    Lp, H = int(g["Lp"]), 12
    hm = torch.from_numpy(g["head_mask"]).view(1, H, 1, 1).cuda()
    exact = mode in ("fp32", "f16x3")

    def perm_for(info, ref_idx):
        """record row r (CLS, the reference's kept order, merged) -> own row (CLS, ascending kept, merged)"""
        k = int(info["indices"].shape[1])
        own = info["indices"].cpu().numpy()
        out = []
        for b in range(B):
            pos = {int(t): p for p, t in enumerate(own[b])}
            out.append([0] + [1 + pos[int(t)] for t in ref_idx[b][:k]] + [k + 1])
        return out

    with runtime.precision(mode), torch.no_grad():
        # --- output_attentions, unpruned and pruned
        for tag, t in (("oa0", 0.0), ("oaT", T)):
            out = layer(hid, mask, None, enc, None, None, True, mode="multimodal", token_attn=ta.clone() if t > 0 else None, temperature=t)
            assert len(out) == 5  # (layer_output, self probs, cross probs, present_key_value, attention_mask): med.py:456-460
            y, sp, cp, present, m = out
            perm = None
            if t > 0:
                info = layer.last_prune
                assert info["pruned"]
                if exact:
                    assert sorted(map(tuple, np.sort(info["indices"].cpu().numpy(), 1).tolist())) == \
                        sorted(map(tuple, np.sort(g["oaT_idx"][:, : info["indices"].shape[1]], 1).tolist()))
                elif not np.array_equal(np.sort(info["indices"].cpu().numpy(), 1), np.sort(g["oaT_idx"][:, : info["indices"].shape[1]], 1)):
                    continue  # (fast mode: a flipped decision changes the sequence; the exact modes carry the claim)
                perm = perm_for(info, g["oaT_idx"])
            _rows_check(g, tag + "_out", y, tol, (mode, tag), perm)
            assert (sp.cpu() - torch.from_numpy(g[tag + "_self_probs"])).abs().max().item() < ptol, (mode, tag, "self probs")
            cpr = cp.cpu() if perm is None else torch.stack([cp[b].cpu()[:, perm[b], :] for b in range(B)])
            assert (cpr - torch.from_numpy(g[tag + "_cross_probs"])).abs().max().item() < ptol, (mode, tag, "cross probs")
            _rows_check(g, tag + "_present_k", present[0], tol, (mode, tag))
            _rows_check(g, tag + "_present_v", present[1], tol, (mode, tag))
            mo = m[:, 0, 0, :].cpu()
            mo = mo if perm is None else torch.stack([mo[b][perm[b]] for b in range(B)])
            assert torch.equal(mo, torch.from_numpy(g[tag + "_mask_out"]))
        # --- head_mask on the fused layer call (value projections scaled per head)
        out = layer(hid, mask, hm, enc, None, None, False, mode="multimodal", temperature=0)
        _rows_check(g, "hm0_out", out[0], tol, (mode, "hm0"))
        out = layer(hid, mask, hm, None, None, None, False, mode="text", token_attn=ta.clone(), temperature=T)
        info = layer.last_prune
        same = np.array_equal(np.sort(info["indices"].cpu().numpy(), 1), np.sort(g["hmT_idx"][:, : info["indices"].shape[1]], 1))
        assert same or not exact, "head-masked pruning decision differs from the reference"
        if same:
            perm = perm_for(info, g["hmT_idx"])
            _rows_check(g, "hmT_out", out[0], tol, (mode, "hmT"), perm)
            mo = out[-1][:, 0, 0, :].cpu()
            assert torch.equal(torch.stack([mo[b][perm[b]] for b in range(B)]), torch.from_numpy(g["hmT_mask_out"]))
        # head_mask AND output_attentions (composed path): the same output, probabilities unmasked (med.py:203-217)
        o2 = layer(hid, mask, hm, enc, None, None, True, mode="multimodal", temperature=0)
        _rows_check(g, "hm0_out", o2[0], tol, (mode, "hm0+oa"))
        assert (o2[1].cpu() - torch.from_numpy(g["oa0_self_probs"])).abs().max().item() < ptol
        _rows_check(g, "oa0_present_v", o2[3][1], tol, (mode, "hm0+oa: the cache holds unscaled values"))
        # --- past_key_value: the cache of the first Lp tokens, then one / two new tokens against it
        first = layer(hid[:, :Lp].contiguous(), mask[:, :, :, :Lp].contiguous(), None, enc, None, None, True, mode="multimodal", temperature=0)
        _rows_check(g, "pk0_out", first[0], tol, (mode, "pk0"))
        past = first[-2]
        assert past[0].shape == (B, H, Lp, 64)
        for tag, n_new in (("pk1", 1), ("pk2", 2)):
            out = layer(hid[:, Lp:Lp + n_new].contiguous(), mask[:, :, :, :Lp + n_new].contiguous(), None, enc, None, past, False,
                        mode="multimodal", temperature=0)
            assert len(out) == 3 and out[0].shape == (B, n_new, D)
            assert (out[0].cpu() - torch.from_numpy(g[tag + "_out"])).abs().max().item() < tol, (mode, tag)
            _rows_check(g, tag + "_present_k", out[1][0], tol, (mode, tag))
            _rows_check(g, tag + "_present_v", out[1][1], tol, (mode, tag))
        with pytest.raises(NotImplementedError):
            layer(hid[:, Lp:Lp + 1].contiguous(), mask[:, :, :, :Lp + 1].contiguous(), None, enc, None, past, False, mode="multimodal",
                  token_attn=ta[:, :1].clone(), temperature=T)


def test_incremental_decoding_through_the_reference_cache_protocol():
    """BertLMHeadModel.forward(past_key_values=..., use_cache=True) - the protocol transformers' generate drives through
    prepare_inputs_for_generation / _reorder_cache (med.py:1071-1094): feeding a sequence one token at a time against the returned
    caches gives the logits of the full-prefix decoder forward at every position; output_hidden_states / output_attentions ride along."""
    from madtp_amd import build, hip, runtime, specs
    from madtp_amd.med import BertConfig, BertLMHeadModel
    build.build(verbose=False)
    hip.load()
    model = BertLMHeadModel(BertConfig.med_default())
    model.load_state_dict(specs.synth_weights(specs.bert_shapes("bert.", "med"), 5), strict=False)
    model.tie_weights()
    model = model.cuda().eval()
    model.tie_weights()
    g = torch.Generator().manual_seed(4)
    B, Lt, Nq = 3, 7, 11
    ids = torch.randint(1000, 30000, (B, Lt), generator=g).cuda()
    ids[:, 0] = 30522
    enc = torch.randn(B, Nq, 768, generator=g).cuda()
    enc_att = torch.ones(B, Nq, dtype=torch.long).cuda()
    for mode, tol in (("fp32", 2e-3), ("f16x3", 2e-3), ("f16", 0.15)):
        with runtime.precision(mode), torch.no_grad():
            full = model(ids, attention_mask=torch.ones_like(ids), encoder_hidden_states=enc, encoder_attention_mask=enc_att,
                         return_dict=True, is_decoder=True)
            past, steps = None, []
            for t in range(Lt):
                kw = model.prepare_inputs_for_generation(ids[:, :t + 1], past=past, attention_mask=torch.ones_like(ids[:, :t + 1]),
                                                         encoder_hidden_states=enc, encoder_attention_mask=enc_att)
                out = model(**kw, use_cache=True, return_dict=True)
                past = out.past_key_values
                assert len(past) == 12 and past[0][0].shape == (B, 12, t + 1, 64)
                steps.append(out.logits[:, -1, :])
            inc = torch.stack(steps, 1)
            scale = full.logits.abs().max().item()
            assert (inc - full.logits).abs().max().item() < tol * max(1.0, scale), (mode, (inc - full.logits).abs().max().item(), scale)
            # beam re-ordering of the cache (med.py:1091-1094) and the optional outputs
            ro = model._reorder_cache(past, torch.tensor([2, 0, 1]).cuda())
            assert torch.equal(ro[3][1][0], past[3][1][2])
            out = model(ids, attention_mask=torch.ones_like(ids), encoder_hidden_states=enc, encoder_attention_mask=enc_att,
                        return_dict=True, is_decoder=True, output_attentions=True, output_hidden_states=True)
            assert len(out.hidden_states) == 13 and len(out.attentions) == 12 and len(out.cross_attentions) == 12
            assert out.attentions[0].shape == (B, 12, Lt, Lt) and out.cross_attentions[0].shape == (B, 12, Lt, Nq)
            assert (out.attentions[5].sum(-1) - 1).abs().max().item() < 1e-4
            assert out.attentions[5][0, 0, 2, 3:].abs().max().item() < 1e-6  # causal: token 2 does not see tokens 3..
            assert (out.logits - full.logits).abs().max().item() < tol * max(1.0, scale)


def test_enqueue_only_vit_encoder_and_graph_replay_equal_the_sync_free_call():
    """madtp_vit_encoder_async with dims_host = NULL (round 6): the twelve blocks are enqueued without a copy or a wait; the device
    record then holds the sync-free call's per-layer decisions and the buffers its outputs - also when the same enqueue is captured
    into a hipGraph once and replayed (the measurement that says a replay does not pay is tools/graph_replay_probe.py)."""
    from madtp_amd import build, configs, harness, hip, runtime
    build.build(verbose=False)
    hip.load()
    T = configs.temperature_for("nlvr", 64, 0.5)[0]
    model = harness.build_nlvr(224, 0, "cuda")
    venc = model.visual_encoder
    images, _, _ = harness.nlvr_inputs(2, 224, 20, seed=3)
    img = images[:3].contiguous()
    for mode in ("f16x3", "f16"):
        with runtime.precision(mode), torch.no_grad():
            patches, np_ = venc.patch_embed.run(img)
            x = hip.assemble_tokens(patches, venc.cls_token, venc.pos_embed, 3, np_)
            weights, qargs, _ = venc._encoder_call_prep(img, model.space_dict)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                ref = hip.vit_encoder(weights, x, qargs, T, sync_free=True)
                k_out, k_used, n_out = ref.results()
                run = hip.vit_encoder(weights, x, qargs, T, sync_free=True, enqueue_only=True)
                s.synchronize()
                d = run.dims_dev[: 13 * 4].view(13, 4).cpu()
                assert d[:12, 1].tolist() == list(k_out) and d[:12, 2].tolist() == list(k_used) and d[:12, 3].tolist() == list(n_out)
                nb = 3 * int(n_out[-1]) * x.shape[-1] * 4
                o_ref = ref.ptr(11, "y") - ref.buf.data_ptr()
                o_run = run.ptr(11, "y") - run.buf.data_ptr()
                assert torch.equal(run.buf[o_run:o_run + nb], ref.buf[o_ref:o_ref + nb])
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    cap = hip.vit_encoder(weights, x, qargs, T, sync_free=True, enqueue_only=True)
                for _ in range(3):
                    g.replay()
                s.synchronize()
                assert torch.equal(cap.dims_dev[: 13 * 4], run.dims_dev[: 13 * 4])
                o_cap = cap.ptr(11, "y") - cap.buf.data_ptr()
                assert torch.equal(cap.buf[o_cap:o_cap + nb], ref.buf[o_ref:o_ref + nb])
