"""Backward of the pruned ViT block on a real MI355X (first half): the hand-written HIP backward
(madtp_amd/backward.py + csrc/backward.hip, fp32 precision mode) against
  * the reference's own .grad (tests/golden/blockgrad_*.npz, recorded from models/vit.py by tools/make_golden.py),
  * autograd through the CPU oracle on the same inputs (full tensors),
with the tolerance SURVEY/VERDICT name: 1e-3 relative to each gradient's largest entry; and the single kernels against torch."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "blockgrad_*.npz")))


@pytest.fixture(scope="module")
def hip():
    from madtp_amd import build, hip as h
    build.build(verbose=False)
    h.load()
    assert torch.cuda.is_available()
    return h


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


TRAIN_MODES = ["fp32", "f16x3"]


def _train_mode(mode):
    """precision context of a gradient test: the exact-f32 mode, or - round 5 - the f16x3 mode with its autograd route switched on
    (every GEMM of the forward, the recomputed forward, dgrad and wgrad as three f16 MFMA products; same fixtures, same tolerance)"""
    import contextlib
    from madtp_amd import runtime
    st = contextlib.ExitStack()
    st.enter_context(runtime.precision(mode))
    if mode == "f16x3":
        st.enter_context(runtime.training_f16x3())
    return st


def test_transpose_colsum_act(hip):
    from madtp_amd import backward as bw
    x = _rand(197, 100, seed=1).cuda()
    t = bw.transpose_pad(x, 224, 128)
    assert t.shape == (128, 224) and torch.equal(t[:100, :197], x.t()) and float(t[100:].abs().max()) == 0 and float(t[:, 197:].abs().max()) == 0
    y = _rand(5000, 776, seed=2).cuda()
    assert _rel(bw.colsum(y).cpu(), y.double().sum(0).float().cpu()) < 1e-5
    assert torch.equal(bw.colsum(y), bw.colsum(y))  # fixed-order reduction
    u, dg = _rand(300, 3072, seed=3).cuda(), _rand(300, 3072, seed=4).cuda()
    ur = u.clone().requires_grad_(True)
    g = F.gelu(ur)
    g.backward(dg)
    assert _rel(bw.act_fwd(u, hip.ACT_GELU), g.detach()) < 1e-6 and _rel(bw.act_bwd(u, dg, hip.ACT_GELU), ur.grad) < 1e-5


def test_dgrad_wgrad(hip):
    from madtp_amd import backward as bw
    M, N, K = 394, 2304, 768
    dy, x, w = _rand(M, N, seed=1).cuda(), _rand(M, K, seed=2).cuda(), _rand(N, K, seed=3, scale=0.05).cuda()
    assert _rel(bw.dgrad(dy, w), (dy.double() @ w.double()).float()) < 1e-5
    assert _rel(bw.wgrad(dy, x), (dy.double().t() @ x.double()).float()) < 1e-5


def test_transpose_split_equals_transpose_then_split(hip):
    """madtp_transpose_split (one pass) writes the bits of madtp_transpose_pad + madtp_split_f16 / madtp_split_f16_weight(scale 1),
    zero padding included - the operands of the f16x3 weight gradient."""
    from madtp_amd import backward as bw
    for (R, C, Rp, Cp) in [(394, 2304, 512, 2304), (197, 100, 256, 128), (1001, 768, 1024, 768), (70, 3, 128, 64)]:
        x = (_rand(R, C, seed=R) * torch.logspace(-6, 1, C)[None, :]).cuda()  # tiny and O(1) columns: both planes carry digits
        t = bw.transpose_pad(x, Rp, Cp)
        act, wt = bw.transpose_split(x, Rp, Cp, False), bw.transpose_split(x, Rp, Cp, True)
        assert act.shape == (Cp, 2 * Rp) and torch.equal(act, hip.split_f16(t))
        ref_w = hip.split_f16_weight(t, log2_scale=0)
        assert torch.equal(wt, ref_w) and wt._madtp_log2_scale == 0 and hip.w_scale_of(wt) == 1.0
        act2, cs = bw.transpose_split(x, Rp, Cp, False, colsum=True)  # the bias gradient's column sums from the same pass
        assert torch.equal(act2, act) and _rel(cs.cpu(), x.double().sum(0).float().cpu()) < 1e-5
        assert torch.equal(cs, bw.transpose_split(x, Rp, Cp, False, colsum=True)[1])


@pytest.mark.parametrize("M,N,K", [(6304, 2304, 768), (25216, 768, 768), (4096, 768, 3072), (2500, 3072, 768), (2100, 264, 776),
                                   (25088, 100, 768)])
def test_wgrad_splitk(hip, M, N, K):
    """the f16x3 weight gradient on split-K partials of the 256x256 ping-pong kernel (madtp_gemm_splitk_pp + madtp_splitk_sum) vs
    float64, next to the plain dispatch it replaces; gradients of 1e-4 keep their digits (activation-format dY^T: ~2^-36 absolute)."""
    from madtp_amd import backward as bw, runtime
    dy = (_rand(M, N, seed=1) * 1e-2).cuda()
    dy[:, : N // 2] *= 1e-2
    x = _rand(M, K, seed=2).cuda()
    ref = dy.double().t() @ x.double()
    with runtime.precision("f16x3"):
        assert bw._wgrad_splits(M, N, K) >= 1
        got = bw.wgrad(dy, x)
        os.environ["MADTP_WGRAD_SPLITK"] = "0"
        try:
            plain = bw.wgrad(dy, x)
        finally:
            del os.environ["MADTP_WGRAD_SPLITK"]
        assert torch.equal(got, bw.wgrad(dy, x))  # fixed summation order
        got2, db = bw.wgrad(dy, x, bias=True)
        assert torch.equal(got2, got) and _rel(db.cpu(), dy.double().sum(0).float().cpu()) < 1e-5
    col = ref.abs().amax(1, keepdim=True)  # per output row (a dY column): small-gradient rows are held to their own scale
    assert float(((got.double() - ref).abs() / col).max()) < 2e-6, float(((got.double() - ref).abs() / col).max())
    assert float(((plain.double() - ref).abs() / col).max()) < 1e-5  # (one f32 accumulation chain over all of K)


def test_backward_refuses_parameters_modified_after_forward(hip):
    """the hand-written backward pairs the forward's activations with the module's current parameters: an in-place update between
    forward and backward raises, as native autograd's version check would"""
    from madtp_amd import runtime, vit
    blk = vit.Block(768, 12, 4.0, qkv_bias=True).cuda()
    x = _rand(2, 50, 768, seed=1).cuda().requires_grad_(True)
    with runtime.precision("fp32"):
        y = blk(x, temperature=0)
        with torch.no_grad():
            blk.mlp.fc1.weight.mul_(1.0)
        with pytest.raises(RuntimeError, match="modified in place"):
            y.sum().backward()
        y = blk(x, temperature=0)
        y.sum().backward()
    assert x.grad is not None and blk.mlp.fc1.weight.grad is not None


def test_update_epochs_are_scoped_to_the_stepped_parameters(hip):
    """ADVICE r5 (round 6): an optimizer step bumps a step count ON ITS OWN parameters (runtime._on_optimizer_step), not a process-wide
    epoch.  (1) a step of ANOTHER model's optimizer between this block's forward and backward no longer raises 'modified in place';
    (2) a frozen block's prepared weights survive it (same prepared tensor object); (3) a step of the block's own optimizer still
    invalidates both; (4) the f16-split scale of a re-prepared weight is reused across an optimizer step but NOT across a copy_ /
    load_state_dict over the same storage (a checkpoint 64x larger gets a fresh scale instead of inf planes), and a reused scale that
    IS outgrown raises the range flag."""
    from madtp_amd import hip as H, runtime, vit
    from madtp_amd.runtime import lin_of
    torch.manual_seed(1)
    blk = vit.Block(768, 12, qkv_bias=True).cuda()
    other = torch.nn.Linear(16, 16).cuda()
    opt_other = torch.optim.SGD(other.parameters(), lr=0.1)
    opt_own = torch.optim.SGD(blk.parameters(), lr=1e-3)
    x = _rand(2, 50, 768, seed=1).cuda().requires_grad_(True)
    with runtime.precision("f16x3"), runtime.training_f16x3():
        with torch.no_grad():
            blk(x.detach())
        prepared = lin_of(blk.mlp._cache, "fc1", [blk.mlp.fc1])
        y = blk(x, temperature=0)
        other(torch.randn(4, 16, device="cuda")).sum().backward()
        opt_other.step()                                            # (1) somebody else's step
        y.sum().backward()
        assert x.grad is not None
        assert lin_of(blk.mlp._cache, "fc1", [blk.mlp.fc1]) is prepared   # (2)
        y = blk(x, temperature=0)
        opt_own.step()                                              # (3) the block's own step
        with pytest.raises(RuntimeError, match="modified in place"):
            y.sum().backward()
        again = lin_of(blk.mlp._cache, "fc1", [blk.mlp.fc1])
        assert again is not prepared and again.log2_scale == prepared.log2_scale and again.age == prepared.age + 1   # (4) reused across a step
        with torch.no_grad():
            blk.mlp.fc1.weight.copy_(blk.mlp.fc1.weight * 64.0)     # a "checkpoint" far from the old values, no optimizer involved
        fresh = lin_of(blk.mlp._cache, "fc1", [blk.mlp.fc1])
        assert fresh.age == 0 and fresh.log2_scale == again.log2_scale - 6
        assert torch.isfinite(fresh.w.float()).all()
        # an outgrown REUSED scale is flagged, not silently turned into inf / NaN planes
        H.range_status(True)
        w = blk.mlp.fc1.weight.detach().float().contiguous()
        H.split_f16_weight((w * 4096.0).contiguous(), log2_scale=fresh.log2_scale + 6)
        torch.cuda.synchronize()
        assert H.range_status(True) == 1


@pytest.mark.parametrize("mode", ["fp32", "f16x3", "bf16"])
def test_prepared_weights_follow_fused_optimizers_and_data_edits(hip, mode):
    """torch.optim's fused=True steps (and edits through .data) bump no Parameter._version: the prepared-weight caches follow them
    through the process-wide update epoch (runtime.py: optimizer step post-hook / parameters_updated()).  Two copies of a block, one
    stepped by the foreach AdamW, one by the fused one, give the same inference output after the steps - and a different one than
    before; a .data edit + parameters_updated() is seen as well."""
    import copy
    from madtp_amd import runtime, vit
    torch.manual_seed(0)
    blk_a = vit.Block(768, 12, qkv_bias=True).cuda().eval()
    blk_b = copy.deepcopy(blk_a)
    x = _rand(2, 50, 768, seed=1).cuda()
    opts = [torch.optim.AdamW(blk_a.parameters(), lr=1e-2), torch.optim.AdamW(blk_b.parameters(), lr=1e-2, fused=True)]
    with runtime.precision(mode), torch.no_grad():
        y0 = [blk_a(x), blk_b(x)]
    assert torch.equal(y0[0], y0[1])
    for _ in range(2):
        for blk, opt in zip((blk_a, blk_b), opts):
            opt.zero_grad(set_to_none=True)
            for i, p_ in enumerate(blk.parameters()):  # (the same synthetic gradient for both copies)
                p_.grad = _rand(*p_.shape, seed=100 + i).cuda()
            opt.step()
    with runtime.precision(mode), torch.no_grad():
        y1 = [blk_a(x), blk_b(x)]
    tol = 1e-4 if mode != "bf16" else 5e-2
    assert _rel(y1[1], y1[0]) < tol, "the fused optimizer's update did not reach the prepared weights"
    assert _rel(y1[0], y0[0]) > 1e-2, "the steps changed the block"
    blk_a.mlp.fc2.weight.data.mul_(0.0)
    runtime.parameters_updated()
    with runtime.precision(mode), torch.no_grad():
        y2 = blk_a(x)
    assert _rel(y2, y1[0]) > 1e-3


def test_layernorm_bwd(hip):
    from madtp_amd import backward as bw
    rows, dim = 1001, 768
    x, dy, add = _rand(rows, dim, seed=1).cuda(), _rand(rows, dim, seed=2).cuda(), _rand(rows, dim, seed=3).cuda()
    gamma, beta = (1 + 0.1 * _rand(dim, seed=4)).cuda(), _rand(dim, seed=5).cuda()
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    F.layer_norm(xr, (dim,), gr, br, 1e-6).backward(dy)
    dx, dgamma, dbeta = bw.layernorm_bwd(x, gamma, dy, 1e-6, add=add)
    assert _rel(dx, xr.grad + add) < 1e-5 and _rel(dgamma, gr.grad) < 1e-5 and _rel(dbeta, br.grad) < 1e-5


@pytest.mark.parametrize("B,N", [(2, 197), (3, 50), (1, 300)])
def test_attention_bwd_plain(hip, B, N):
    """softmax attention backward with recomputed probabilities vs torch autograd (no score terms)."""
    from madtp_amd import backward as bw
    H, D = 12, 768
    qkv = _rand(B * N, 3 * D, seed=1, scale=0.5).cuda()
    dout = _rand(B * N, D, seed=2).cuda()
    r = qkv.clone().requires_grad_(True)
    q, k, v = [t.reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in (r[:, :D], r[:, D:2 * D], r[:, 2 * D:])]
    p = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B * N, D)
    o.backward(dout)
    dqkv = bw.attention_bwd(qkv, dout, o.detach().contiguous(), B, H, N, 0.125)
    assert _rel(dqkv, r.grad) < 2e-5
    assert torch.equal(dqkv, bw.attention_bwd(qkv, dout, o.detach().contiguous(), B, H, N, 0.125))


@pytest.mark.parametrize("B,Nq,Nk", [(2, 35, 197), (3, 20, 50), (1, 7, 577)])
def test_attention_bwd_cross(hip, B, Nq, Nk):
    """cross-attention backward (Nq queries against Nk keys of another sequence, fused [k|v] projection, optional key mask) vs
    torch autograd."""
    from madtp_amd import backward as bw
    H, D = 12, 768
    q, kv, dout = _rand(B * Nq, D, seed=1, scale=0.5).cuda(), _rand(B * Nk, 2 * D, seed=2, scale=0.5).cuda(), _rand(B * Nq, D, seed=3).cuda()
    mask = torch.zeros(B, Nk)
    mask[:, Nk - 3:] = -10000.0
    for km in (None, mask.cuda()):
        qr, kvr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        qh = qr.reshape(B, Nq, H, 64).permute(0, 2, 1, 3)
        kh = kvr[:, :D].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
        vh = kvr[:, D:].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
        sc = (qh @ kh.transpose(-2, -1)) * 0.125
        if km is not None:
            sc = sc + km[:, None, None, :]
        o = (sc.softmax(-1) @ vh).transpose(1, 2).reshape(B * Nq, D)
        o.backward(dout)
        dq, dkv = bw.attention_bwd_cross(q, kv, dout, B, H, Nq, Nk, 0.125, key_mask=km)
        assert _rel(dq, qr.grad) < 2e-5 and _rel(dkv, kvr.grad) < 2e-5


@pytest.mark.parametrize("B,n,D", [(3, 50, 768), (2, 196, 768), (1, 300, 768), (2, 197, 512), (2, 40, 72), (64, 197, 768)])
def test_att_ft_bwd(hip, B, n, D):
    """madtp_att_ft_bwd vs torch autograd through models/utils.py:174-178 (softmax over tokens of inner / sqrt(d), W q); the kernel
    ADDS to the gradients it is given."""
    import math
    from madtp_amd import backward as bw
    K = 100  # (D % 16 == 0: q dA^T as a batched exact-f32 MFMA product; else the per-column dot-product loop)
    q, sd, dA = _rand(B, n, D, seed=1).cuda(), _rand(K, D, seed=2, scale=0.3).cuda(), _rand(B, K, D, seed=3).cuda()
    qr, ir = q.clone().requires_grad_(True), (q @ sd.t()).clone().requires_grad_(True)
    w = torch.softmax((ir / math.sqrt(D)).permute(0, 2, 1), dim=-1)
    (torch.bmm(w, qr) * dA).sum().backward()
    dinner0, dq0 = _rand(B, n, K, seed=4).cuda(), _rand(B, n, D, seed=5).cuda()
    dinner, dq = dinner0.clone(), dq0.clone()
    bw.att_ft_bwd(ir.detach().contiguous(), q, dA, D, dinner, dq)
    assert _rel(dinner - dinner0, ir.grad) < 2e-5 and _rel(dq - dq0, qr.grad) < 2e-5
    d2, q2 = dinner0.clone(), dq0.clone()
    bw.att_ft_bwd(ir.detach().contiguous(), q, dA, D, d2, q2)
    assert torch.equal(d2, dinner) and torch.equal(q2, dq)  # fixed summation order


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", GRAD_CASES, ids=[os.path.basename(c)[:-4] for c in GRAD_CASES])
def test_block_backward_matches_reference_grads(hip, path, mode):
    from madtp_amd import runtime, vit
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build(g)
    blk = vit.Block(768, 12, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    blk.load_state_dict({k[len(c["prefix"]):]: v for k, v in c["W"].items() if k.startswith(c["prefix"])}, strict=True)
    blk = blk.cuda()
    x = c["x"].cuda().requires_grad_(True)
    ta = c["token_attn"].cuda().requires_grad_(True)
    with _train_mode(mode):
        y = blk(x, False, 0, c["T"], ta)
        assert tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
        info = blk.last_prune
        for b in range(x.shape[0]):
            assert {int(v) for v in info["indices"][b]} == {int(v) for v in g["blk_idx"][b]}, "kept set differs from the reference"
        # same upstream gradient per TOKEN as in the recording (the two paths order the kept tokens differently)
        G = grad_case.permute_G(c["G"], g["blk_idx"], info["indices"].cpu().numpy())
        (y * G.cuda()).sum().backward()
    grads = {"x": x.grad, "token_attn": ta.grad}
    grads.update({k: p.grad for k, p in blk.named_parameters()})
    # (1) the reference's own gradients (sampled entries, norms)
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP backward vs reference")
    # (2) autograd through the CPU oracle, every entry
    ref, _, oinfo = O.vit_block_grads(c["W"], c["prefix"], c["x"], c["token_attn"], c["T"], c["G"])
    assert np.array_equal(oinfo["indices"].numpy(), g["blk_idx"]), "the oracle keeps the recording's token order on this box"
    for name, r in ref.items():
        e = _rel(grads[name].cpu(), r)
        assert e < 1e-3, f"grad {name}: {e:.3e} of its maximum"
    # the alignment-logit gradient has one entry per token row, at the row maximum
    nz = (ta.grad != 0).sum(-1)
    assert int(nz.max()) <= 1
    # not pruned (temperature 0): the plain residual-block backward
    x2 = c["x"].cuda().requires_grad_(True)
    for p in blk.parameters():
        p.grad = None
    G2 = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, size=tuple(c["x"].shape)).astype(np.float32))
    with runtime.precision("fp32"):
        y2 = blk(x2, False, 0, 0, None)
        (y2 * G2.cuda()).sum().backward()
    Wl = {k: v for k, v in c["W"].items() if k.startswith(c["prefix"])}
    leaves = {k: v.clone().requires_grad_(True) for k, v in Wl.items()}
    xl = c["x"].clone().requires_grad_(True)
    yo, _ = O.vit_block(leaves, c["prefix"], xl, 0, None)
    (yo * G2).sum().backward()
    assert _rel(x2.grad.cpu(), xl.grad) < 1e-3
    for k, p in blk.named_parameters():
        assert _rel(p.grad.cpu(), leaves[c["prefix"] + k].grad) < 1e-3, k


def test_block_backward_needs_fp32_mode(hip):
    from madtp_amd import runtime, vit
    blk = vit.Block(768, 12, qkv_bias=True).cuda()
    x = torch.randn(1, 20, 768, device="cuda", requires_grad=True)
    with runtime.precision("bf16"), pytest.raises(NotImplementedError):
        blk(x)


VITGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "encgrad_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", VITGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in VITGRAD_CASES])
def test_vit_backward_matches_reference_grads(hip, path, mode):
    """The whole pruned ViT under autograd (madtp_amd/backward.py::vit_forward_with_grad: patch embedding + CLS / position, the
    query model's logits, twelve VitBlockFunctions, final LayerNorm; fp32 mode) against the reference's own .grad of
    models/vit.py VisionTransformer.forward for all 150 parameters and space_dict (tests/golden/encgrad_*.npz), loss =
    oracle.vit_loss on the image tokens (token-order invariant).  Tolerance 1e-3 of each gradient's largest entry; the per-layer
    kept sets must equal the recording (otherwise the gradients are not comparable)."""
    from madtp_amd import runtime, synth
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(path)
    B, size, T, seed = int(g["B"]), int(g["size"]), float(g["temperature"]), int(g["seed"])
    from madtp_amd import specs, vit as mvit
    venc = mvit.VisionTransformer(img_size=size, patch_size=16, embed_dim=768, depth=12, num_heads=12, evaluate=True, sd_dim=768)
    venc.load_state_dict(specs.synth_weights(specs.vit_shapes("", size), seed), strict=True)
    venc = venc.cuda().eval()
    for p_ in venc.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    images = synth.synth_images(B, size, seed).cuda()
    space_dict = synth.synth_tensor("space_dict", (100, 768), seed).cuda().requires_grad_(True)
    gv, hv, av = [t.cuda() for t in grad_case.vit_loss_vectors(g)]
    with _train_mode(mode):
        y, sd_all = venc(images, space_dict=space_dict, temperature=T)
        assert y.requires_grad and tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
        # kept sets as ORIGINAL patch ids (the two sides order a layer's tokens differently)
        from madtp_amd import harness
        n0 = (size // 16) ** 2
        ref_trace = [{"pruned": True, "indices": g[f"vit{i}_idx"]} if f"vit{i}_idx" in g.files else None for i in range(12)]
        own_trace = [None if blk.last_prune is None or not blk.last_prune.get("pruned") else
                     {"pruned": True, "indices": blk.last_prune["indices"].cpu()} for blk in venc.blocks]
        assert harness.compose_ids(own_trace, n0) == harness.compose_ids(ref_trace, n0), "kept sets differ from the recording"
        assert abs(float(y.detach().double().norm()) - float(g["y_norm"])) < 1e-4 * float(g["y_norm"])
        assert sd_all.requires_grad and abs(float(sd_all.detach().double().norm()) - float(g["sd_all_norm"])) < 1e-4 * float(g["sd_all_norm"])
        (O.vit_loss(y, gv, hv) + (sd_all * av).sum()).backward()   # both outputs of the forward enter the loss
    grads = {k: v.grad for k, v in venc.named_parameters() if v.grad is not None}
    grads["space_dict"] = space_dict.grad
    missing = [k[2:-7] for k in g.files if k.startswith("g_") and k.endswith("_sample") and k[2:-7] not in grads]
    assert not missing, f"no gradient produced for {missing[:5]}"
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP ViT backward vs reference")


MEDGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "medgrad_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", MEDGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in MEDGRAD_CASES])
def test_med_layer_backward_matches_reference_grads(hip, path, mode):
    """The MED BertLayer in mode 'text' under autograd (madtp_amd/backward.py::MedTextLayerFunction: masked self-attention, output
    LayerNorm, Reduce_token on the post-LN tokens, FFN; fp32 mode) against the reference's own .grad of models/med.py
    BertLayer.forward (hidden, token_attn, the layer's 16 parameters; tests/golden/medgrad_*.npz, ragged padding masks) and against
    autograd through the CPU oracle on every entry.  Loss = oracle.vit_loss on the layer output (token-order invariant)."""
    from madtp_amd import med, runtime
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_med(g)
    layer = grad_case.build_med_layer(c)
    hidden = c["hidden"].cuda().requires_grad_(True)
    ta = c["token_attn"].cuda().requires_grad_(True)
    mask = c["add_mask"].cuda()
    gv, hv = c["g"].cuda(), c["h"].cuda()
    enc = c["enc"].cuda().requires_grad_(True) if c["enc"] is not None else None
    enc_mask = torch.zeros(enc.shape[0], 1, 1, enc.shape[1], device="cuda") if enc is not None else None
    with _train_mode(mode):
        out = layer(hidden, mask, None, enc, enc_mask, None, False, mode=c["mode"], token_attn=ta, reduce_num=0, temperature=c["T"])
        y, mask_out = out[0], out[-1]
        assert y.requires_grad and tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
        assert abs(float(y.detach().double().norm()) - float(g["y_norm"])) < 1e-4 * float(g["y_norm"])
        assert sorted(mask_out[:, 0, 0, :].cpu().reshape(-1).tolist()) == sorted(g["mask_out"].reshape(-1).tolist())
        O.vit_loss(y, gv, hv).backward()
    grads = {"hidden": hidden.grad, "token_attn": ta.grad}
    if enc is not None:
        grads["enc"] = enc.grad
    grads.update({k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP MED layer backward vs reference")
    ref, yo, _, _ = O.bert_layer_grads(c["W"], c["prefix"], c["hidden"], c["add_mask"], c["T"], c["token_attn"], c["g"], c["h"],
                                       layer_num=c["layer"], enc=c["enc"])
    for name, r in ref.items():
        scale = ref[name[:-8] + "query.bias"] if name.endswith("key.bias") else r   # (a key bias has no true gradient: noise)
        e = float((grads[name].cpu() - r).abs().max()) / max(float(scale.abs().max()), 1e-12)
        assert e < 1e-3, f"grad {name}: {e:.3e} of its maximum"


NLVRGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nlvrgrad_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", NLVRGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in NLVRGRAD_CASES])
def test_nlvr_layer_backward_matches_reference_grads(hip, path, mode):
    """The NLVR BertLayer (the headline model's text layer: twin cross-attention to the two images' tokens, averaged below layer 6,
    merged by merge_layer from layer 6 on) under autograd against the reference's own .grad of models/nlvr_encoder.py
    BertLayer.forward (hidden, token_attn, both image sequences, all parameters; tests/golden/nlvrgrad_*.npz) and against autograd
    through the CPU oracle on every entry."""
    from madtp_amd import runtime
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_nlvr(g)
    layer = grad_case.build_nlvr_layer(c)
    hidden = c["hidden"].cuda().requires_grad_(True)
    ta = c["token_attn"].cuda().requires_grad_(True)
    mask = c["add_mask"].cuda()
    enc = [e.cuda().requires_grad_(True) for e in c["enc"]]
    enc_mask = [m.cuda() for m in c["enc_mask"]]
    gv, hv = c["g"].cuda(), c["h"].cuda()
    with _train_mode(mode):
        out = layer(hidden, mask, None, None, enc, enc_mask, None, False, mode="multimodal", token_attn=ta, reduce_num=0,
                    temperature=c["T"])
        y = out[0]
        assert y.requires_grad and tuple(y.shape) == tuple(int(v) for v in g["out_shape"])
        assert abs(float(y.detach().double().norm()) - float(g["y_norm"])) < 1e-4 * float(g["y_norm"])
        O.vit_loss(y, gv, hv).backward()
    grads = {"hidden": hidden.grad, "token_attn": ta.grad, "enc0": enc[0].grad, "enc1": enc[1].grad}
    grads.update({k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP NLVR layer backward vs reference")
    ref, _, _, _ = O.bert_layer_grads(c["W"], c["prefix"], c["hidden"], c["add_mask"], c["T"], c["token_attn"], c["g"], c["h"],
                                      layer_num=c["layer"], variant="nlvr", enc=c["enc"], enc_mask=c["enc_mask"])
    for name, r in ref.items():
        scale = ref[name[:-8] + "query.bias"] if name.endswith("key.bias") else r   # (a key bias has no true gradient: noise)
        e = float((grads[name].cpu() - r).abs().max()) / max(float(scale.abs().max()), 1e-12)
        assert e < 1e-3, f"grad {name}: {e:.3e} of its maximum"


MODELGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "modelgrad_nlvr_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", MODELGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in MODELGRAD_CASES])
def test_nlvr_model_backward_matches_reference_grads(hip, path, mode):
    """loss.backward() through the headline model on the HIP path (BLIP_NLVR.forward(train=False) in the fp32 mode with grad mode
    on: pruned ViT on both images, BERT embeddings, twelve NLVR layers with twin cross-attention, cls_head - every stage an
    autograd.Function of madtp_amd/backward.py) against the reference's own .grad of models/blip_nlvr.py for all 579 parameters
    (space_dict included), loss = sum(logits * c).  The per-layer token counts must equal the recording."""
    from madtp_amd import harness, runtime, synth
    from tests import grad_case
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    model = harness.build_nlvr(size, seed, "cuda")
    for p_ in model.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    images, text, targets = harness.nlvr_inputs(B, size, L, seed, "cuda", pad_tail=int(g["pad_tail"]))
    c = torch.from_numpy(synth.uniform_pm1("nlvrgrad_c", B * 2, seed).reshape(B, 2)).cuda()
    with _train_mode(mode):
        logits = model(images, text, targets, temperature=T, train=False)
        assert logits.requires_grad and (logits.detach().cpu() - torch.from_numpy(g["logits"])).abs().max().item() < 1e-4
        trace = {"vit": [harness._cpu_info(b.last_prune) for b in model.visual_encoder.blocks],
                 "text": [harness._cpu_info(l.last_prune) for l in model.text_encoder.encoder.layer]}
        assert harness.token_lengths(trace["vit"], (size // 16) ** 2 + 1) == g["vit_lens"].tolist()
        assert harness.token_lengths(trace["text"], L) == g["txt_lens"].tolist()
        (logits * c).sum().backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    missing = [k[2:-7] for k in g.files if k.startswith("g_") and k.endswith("_sample") and k[2:-7] not in grads]
    assert not missing, f"no gradient produced for {len(missing)} tensors, e.g. {missing[:5]}"
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP BLIP_NLVR backward vs reference")


TRAINSTEP_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trainstep_nlvr_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", TRAINSTEP_CASES, ids=[os.path.basename(c)[:-4] for c in TRAINSTEP_CASES])
def test_nlvr_training_step_matches_reference(hip, path, mode):
    """One compression training step of the headline model on the HIP path (compress_nlvr_dtp.py:52-56): BLIP_NLVR.forward(
    train=True) -> (loss_ori, loss_fdt), loss = loss_ori + 0.1 loss_fdt, backward - both losses and the gradients of all 579
    parameters against the reference's own (model.eval(): the mirror has no dropout); then three AdamW steps on the same batch
    lower the loss (the optimizer is torch's, on the parameters' .grad)."""
    from madtp_amd import harness, runtime
    from tests import grad_case
    g = np.load(path)
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    model = harness.build_nlvr(size, seed, "cuda")
    for p_ in model.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    images, text, _ = harness.nlvr_inputs(B, size, L, seed, "cuda", pad_tail=int(g["pad_tail"]))
    targets = (torch.arange(B) % 2).cuda()
    with _train_mode(mode):
        lo, lf = model(images, text, targets, temperature=T, train=True)
        assert abs(float(lo.detach()) - float(g["loss_ori"])) < 1e-4 and abs(float(lf.detach()) - float(g["loss_fdt"])) < 1e-4
        (lo + 0.1 * lf).backward()
        grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        grad_case.check_against_fixture(g, grads, 1e-3, "HIP training step vs reference")
        opt = torch.optim.AdamW(model.parameters(), lr=2e-5, weight_decay=0.05)
        losses = [float((lo + 0.1 * lf).detach())]
        opt.step()
        for _ in range(3):
            opt.zero_grad()
            lo, lf = model(images, text, targets, temperature=T, train=True)
            loss = lo + 0.1 * lf
            losses.append(float(loss.detach()))
            loss.backward()
            opt.step()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


DECGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "decgrad_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", DECGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in DECGRAD_CASES])
def test_decoder_backward_matches_reference_grads(hip, path, mode):
    """The answer / caption decoder's training forward on the HIP path (BertLMHeadModel.forward with labels, reduction='none':
    embeddings, twelve MED layers with the CAUSAL self-attention mask and cross-attention to the question states, LM head with the
    tied output embedding, label-smoothed next-token cross-entropy) under autograd in the fp32 mode: per-sequence losses and the
    gradients of every decoder parameter + the question states against the reference's own (tests/golden/decgrad_*.npz)."""
    from madtp_amd import runtime
    from madtp_amd.med import BertConfig, BertLMHeadModel
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_decoder(g)
    model = BertLMHeadModel(BertConfig.med_default())
    model.load_state_dict(c["W"], strict=False)
    model.tie_weights()
    model = model.cuda().eval()
    model.tie_weights()
    for p_ in model.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    enc = c["enc"].cuda().requires_grad_(True)
    B = c["ids"].shape[0]
    with _train_mode(mode):
        out = model(c["ids"].cuda(), attention_mask=c["att"].cuda(), encoder_hidden_states=enc,
                    encoder_attention_mask=c["enc_att"].cuda(), labels=c["labels"].cuda(), return_dict=True, reduction='none')
        assert (out.loss.detach().cpu() - torch.from_numpy(g["loss"])).abs().max().item() < 1e-3 * float(np.abs(g["loss"]).max())
        ((c["w"].cuda() * out.loss).sum() / B).backward()
    grads = {"enc": enc.grad}
    seen = set()
    for k, p_ in model.named_parameters():
        if p_.grad is not None and id(p_) not in seen:
            seen.add(id(p_))
            grads[k] = p_.grad
    missing = [k[2:-7] for k in g.files if k.startswith("g_") and k.endswith("_sample") and k[2:-7] not in grads]
    assert not missing, f"no gradient produced for {len(missing)} tensors, e.g. {missing[:5]}"
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP decoder backward vs reference")


VQATRAIN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trainstep_vqa_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", VQATRAIN_CASES, ids=[os.path.basename(c)[:-4] for c in VQATRAIN_CASES])
def test_vqa_training_step_matches_reference(hip, path, mode):
    """One training step of BLIP_VQA on the HIP path (blip_vqa.py:57-115: pruned ViT, question encoder = MED in mode 'multimodal'
    with text pruning, answer decoder teacher-forced on the question states repeated n[b] times; loss_vqa + 0.1 loss_fdt) in the
    fp32 mode: both losses and the gradients of all 788 parameters against the reference's own."""
    from madtp_amd import runtime
    from madtp_amd.blip_vqa import BLIP_VQA
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_vqa_train(g)
    model = BLIP_VQA(image_size=int(g["size"]), evaluate=True, decoder=True)
    msg = model.load_state_dict(c["W"], strict=False)
    assert not msg.unexpected_keys, msg.unexpected_keys[:5]
    model.text_decoder.tie_weights()
    model = model.cuda().eval()
    model.text_decoder.tie_weights()
    for p_ in model.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    with _train_mode(mode):
        lv, lf = model(c["images"].cuda(), {"input_ids": c["ids"].cuda(), "attention_mask": c["att"].cuda()},
                       {"input_ids": c["a_ids"].cuda(), "attention_mask": c["a_att"].cuda()}, temperature=c["T"], train=True,
                       n=c["n_list"], weights=c["weights"])
        assert abs(float(lv.detach()) - float(g["loss_vqa"])) < 1e-3 * float(g["loss_vqa"])
        assert abs(float(lf.detach()) - float(g["loss_fdt"])) < 1e-4
        (lv + 0.1 * lf).backward()
    grads, seen = {}, set()
    for k, p_ in model.named_parameters():
        if p_.grad is not None and id(p_) not in seen:
            seen.add(id(p_))
            grads[k] = p_.grad
    missing = [k[2:-7] for k in g.files if k.startswith("g_") and k.endswith("_sample") and k[2:-7] not in grads]
    assert not missing, f"no gradient produced for {len(missing)} tensors, e.g. {missing[:5]}"
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP VQA training step vs reference")
    # model.train(): embeddings, the encoder's and the decoder's layers (causal self-attention, cross-attention to the question
    # states) drop as the reference's loop does - finite, repeatable per seed, different across seeds and from model.eval()
    model.train()
    vals = []
    for seed in (11, 11, 12):
        model.zero_grad(set_to_none=True)
        runtime.set_dropout_seed(seed)
        with _train_mode(mode):
            lv2, lf2 = model(c["images"].cuda(), {"input_ids": c["ids"].cuda(), "attention_mask": c["att"].cuda()},
                             {"input_ids": c["a_ids"].cuda(), "attention_mask": c["a_att"].cuda()}, temperature=c["T"], train=True,
                             n=c["n_list"], weights=c["weights"])
            (lv2 + 0.1 * lf2).backward()
        vals.append(float(lv2.detach()))
        assert np.isfinite(vals[-1]) and all(bool(torch.isfinite(p_.grad).all()) for p_ in model.parameters() if p_.grad is not None)
    assert vals[0] == vals[1] and vals[0] != vals[2] and vals[0] != float(lv.detach())


CAPTRAIN_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trainstep_cap_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", CAPTRAIN_CASES, ids=[os.path.basename(c)[:-4] for c in CAPTRAIN_CASES])
def test_caption_training_step_matches_reference(hip, path, mode):
    """One training step of the caption model on the HIP path (BLIP_Decoder.forward(train=True), models/blip.py:111-158: pruned ViT
    + decoder teacher-forced on the caption, prompt and padding masked out of the targets) in the fp32 mode: loss_lm and the
    gradients of all 472 parameters against the reference's own."""
    from madtp_amd import runtime, specs, synth
    from madtp_amd.blip import BLIP_Decoder
    from tests import grad_case
    g = np.load(path)
    B, size, seed = int(g["B"]), int(g["size"]), int(g["seed"])
    model = BLIP_Decoder(image_size=size, evaluate=True)
    msg = model.load_state_dict(specs.tie_keys(specs.synth_weights(specs.blip_decoder_shapes(size), seed)), strict=False)
    assert not msg.unexpected_keys, msg.unexpected_keys[:5]
    model = model.cuda().eval()
    model.text_decoder.tie_weights()
    assert model.prompt_length == int(g["prompt_length"])
    for p_ in model.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    with _train_mode(mode):
        lm, lf = model(synth.synth_images(B, size, seed).cuda(), {"input_ids": torch.from_numpy(g["ids"]).cuda(),
                                                                  "attention_mask": torch.from_numpy(g["att"]).cuda()},
                       temperature=float(g["temperature"]), train=True)
        assert abs(float(lm.detach()) - float(g["loss_lm"])) < 1e-3 * float(g["loss_lm"])
        (lm + 0.1 * lf).backward()
    grads, seen = {}, set()
    for k, p_ in model.named_parameters():
        if p_.grad is not None and id(p_) not in seen:
            seen.add(id(p_))
            grads[k] = p_.grad
    missing = [k[2:-7] for k in g.files if k.startswith("g_") and k.endswith("_sample") and k[2:-7] not in grads]
    assert not missing, f"no gradient produced for {len(missing)} tensors, e.g. {missing[:5]}"
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP caption training step vs reference")
    model.train()  # dropout in the decoder (causal self-attention + cross-attention to the image tokens), DropPath in the ViT
    vals = []
    for seed in (21, 21, 22):
        model.zero_grad(set_to_none=True)
        runtime.set_dropout_seed(seed)
        with _train_mode(mode):
            lm2, lf2 = model(synth.synth_images(B, size, seed=int(g["seed"])).cuda(), {"input_ids": torch.from_numpy(g["ids"]).cuda(),
                                                                                   "attention_mask": torch.from_numpy(g["att"]).cuda()},
                             temperature=float(g["temperature"]), train=True)
            (lm2 + 0.1 * lf2).backward()
        vals.append(float(lm2.detach()))
        assert np.isfinite(vals[-1]) and all(bool(torch.isfinite(p_.grad).all()) for p_ in model.parameters() if p_.grad is not None)
    assert vals[0] == vals[1] and vals[0] != vals[2] and vals[0] != float(lm.detach())


CLIPGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clipgrad_*.npz")))


@pytest.mark.parametrize("path", CLIPGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in CLIPGRAD_CASES])
def test_clip_block_backward_matches_reference_grads(hip, path):
    """CLIP's ResidualAttentionBlock (clip/model.py:174-261: fused in_proj, QuickGELU, the block's own query model with q_map, the
    k <= max_keep rule) under autograd in the fp32 mode - the block backward of the BLIP ViT behind a parameter adapter,
    QueryModelFunction with the mapped tokens - against the reference's own .grad (x, space_dict, 14 parameters; sd_ft_all enters
    the loss too)."""
    from madtp_amd import clip_model, runtime
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_clip_block(g)
    blk = clip_model.ResidualAttentionBlock(768, 12, None, sd_dim=768)
    pre = c["prefix"]
    blk.load_state_dict({k[len(pre):]: v for k, v in c["W"].items() if k.startswith(pre)}, strict=True)
    blk = blk.cuda().eval()
    for p_ in blk.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    x = c["x"].permute(1, 0, 2).contiguous().cuda().requires_grad_(True)   # (N, B, C) as the reference passes it
    sd = c["space_dict"].cuda().requires_grad_(True)
    with runtime.precision("fp32"):
        y, _, _, sd_ft, _ = blk((x, sd, c["T"], None, c["max_keep"]))
        yb = y.permute(1, 0, 2)
        assert tuple(yb.shape) == tuple(int(v) for v in g["out_shape"])
        assert abs(float(yb.detach().double().norm()) - float(g["y_norm"])) < 1e-4 * float(g["y_norm"])
        assert abs(float(sd_ft.detach().double().norm()) - float(g["sd_ft_norm"])) < 1e-4 * float(g["sd_ft_norm"])
        (O.vit_loss(yb, c["g"].cuda(), c["h"].cuda()) + (sd_ft * c["a"].cuda()).sum()).backward()
    grads = {"x": x.grad.permute(1, 0, 2).contiguous(), "space_dict": sd.grad}
    grads.update({k: p_.grad for k, p_ in blk.named_parameters() if p_.grad is not None})
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP CLIP block backward vs reference")


CLIPTEXTGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cliptextgrad_*.npz")))


@pytest.mark.parametrize("path", CLIPTEXTGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in CLIPTEXTGRAD_CASES])
def test_clip_text_block_backward_matches_reference_grads(hip, path):
    """A block of CLIP's text tower (width 512, 8 heads, causal attn_mask applied as mask[:N,:N]) under autograd: the causal mask
    together with the pruning score terms in the attention backward, against the reference's own .grad."""
    from madtp_amd import clip_model, runtime
    from oracle import madtp_oracle as O
    from tests import grad_case
    from tests.test_oracle_golden import _clip_text_block_case
    g = np.load(path)
    c = _clip_text_block_case(g)
    blk = clip_model.ResidualAttentionBlock(512, 8, c["mask"], sd_dim=768)
    blk.load_state_dict({k[2:]: v for k, v in c["W"].items()}, strict=True)
    blk = blk.cuda().eval()
    for p_ in blk.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    x = c["x"].permute(1, 0, 2).contiguous().cuda().requires_grad_(True)
    sd = c["space_dict"].cuda().requires_grad_(True)
    with runtime.precision("fp32"):
        y, _, _, sd_ft, _ = blk((x, sd, c["T"], None, c["max_keep"]))
        yb = y.permute(1, 0, 2)
        assert tuple(yb.shape) == tuple(int(v) for v in g["out_shape"])
        assert abs(float(yb.detach().double().norm()) - float(g["y_norm"])) < 1e-4 * float(g["y_norm"])
        (O.vit_loss(yb, c["g"].cuda(), c["h"].cuda()) + (sd_ft * c["a"].cuda()).sum()).backward()
    grads = {"x": x.grad.permute(1, 0, 2).contiguous(), "space_dict": sd.grad}
    grads.update({k: p_.grad for k, p_ in blk.named_parameters() if p_.grad is not None})
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP CLIP text block backward vs reference")


CLIPVITGRAD_CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "clipvitgrad_*.npz")))


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", CLIPVITGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in CLIPVITGRAD_CASES])
def test_clip_vision_backward_matches_reference_grads(hip, path, mode):
    """CLIP's vision tower (encode_image) under autograd on the HIP path: conv1 + class / positional embedding, ln_pre, twelve
    pruned blocks with their own query models, ln_post, proj - features, per-layer lengths and the gradients of all 176 parameters
    + space_dict against the reference's own."""
    from madtp_amd import clip_model, runtime
    from tests import grad_case
    from tests.test_oracle_golden import _clip_vit_case
    g = np.load(path)
    c = _clip_vit_case(g)
    vt = clip_model.VisionTransformer(input_resolution=int(g["size"]), patch_size=16, width=768, layers=12, heads=12, output_dim=512,
                                      sd_dim=768)
    vt.load_state_dict(c["W"], strict=True)
    vt = vt.cuda().eval()
    for p_ in vt.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    sd = c["space_dict"].cuda().requires_grad_(True)
    with _train_mode(mode):
        feat, sd_all = vt(c["images"].cuda(), sd, c["T"], 1)
        assert feat.requires_grad and (feat.detach().cpu() - torch.from_numpy(g["features"])).abs().max().item() < 1e-4
        lens = [int(b.last_prune["indices"].shape[1]) + 2 if (b.last_prune and b.last_prune.get("pruned")) else None
                for b in vt.transformer.resblocks]
        n = (int(g["size"]) // 16) ** 2 + 1
        got = []
        for v in lens:
            n = v if v is not None else n
            got.append(n)
        assert got == g["vit_lens"].tolist()
        ((feat * c["c"].cuda()).sum() + (sd_all * c["a"].cuda()).sum()).backward()
    grads = {k: p_.grad for k, p_ in vt.named_parameters() if p_.grad is not None}
    grads["space_dict"] = sd.grad
    missing = [k[2:-7] for k in g.files if k.startswith("g_") and k.endswith("_sample") and k[2:-7] not in grads]
    assert not missing, f"no gradient produced for {missing[:5]}"
    grad_case.check_against_fixture(g, grads, 1e-3, "HIP CLIP vision tower backward vs reference")


def test_register_hook_captures_attention_gradients(hip):
    """Block.forward(register_hook=True) (vit.py:88-90, 189: the Grad-CAM hook on the attention map): after loss.backward() the
    block's get_attn_gradients() holds d loss / d attention-probabilities [B,H,N,N] and get_attention_map() the probabilities -
    against torch autograd through the CPU oracle's block with the probabilities retained (pruned block: the score terms of the
    merge weights are part of that gradient)."""
    from madtp_amd import runtime, vit
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(GRAD_CASES[0])
    c = grad_case.build(g)
    blk = vit.Block(768, 12, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    blk.load_state_dict({k[len(c["prefix"]):]: v for k, v in c["W"].items() if k.startswith(c["prefix"])}, strict=True)
    blk = blk.cuda()
    x = c["x"].cuda().requires_grad_(True)
    ta = c["token_attn"].cuda()
    with runtime.precision("fp32"):
        y = blk(x, True, 0, c["T"], ta)
        G = grad_case.permute_G(c["G"], g["blk_idx"], blk.last_prune["indices"].cpu().numpy())
        (y * G.cuda()).sum().backward()
    dP, P = blk.attn.get_attn_gradients(), blk.attn.get_attention_map()
    B, N = x.shape[0], x.shape[1]
    assert tuple(dP.shape) == (B, 12, N, N) and tuple(P.shape) == (B, 12, N, N)
    # reference: autograd through the oracle's block with the softmax output retained
    kept = {}
    orig = O.vit_attention

    def tapped(W, p, xx, num_heads=12):
        out = orig(W, p, xx, num_heads)
        probs = [t for t in out if torch.is_tensor(t) and t.dim() == 4][0]
        probs.retain_grad()
        kept["P"] = probs
        return out
    O.vit_attention = tapped
    try:
        Wl = {k: v.clone().requires_grad_(True) for k, v in c["W"].items() if k.startswith(c["prefix"])}
        xl = c["x"].clone().requires_grad_(True)
        yo, _ = O.vit_block(Wl, c["prefix"], xl, c["T"], c["token_attn"].clone())
        (yo * c["G"]).sum().backward()
    finally:
        O.vit_attention = orig
    assert _rel(P.cpu(), kept["P"].detach()) < 1e-5
    assert _rel(dP.cpu(), kept["P"].grad) < 1e-3
    # outside the autograd path the hook cannot work: loud error
    with runtime.precision("bf16"), pytest.raises(NotImplementedError):
        blk(x.detach(), True, 0, c["T"], ta)


def test_vit_register_blk_routes_the_hook(hip):
    """VisionTransformer.forward(register_blk=i) (vit.py:281, 304): block i - and only block i - captures its attention gradients."""
    from madtp_amd import runtime, specs, synth, vit as mvit
    venc = mvit.VisionTransformer(img_size=96, patch_size=16, embed_dim=768, depth=12, num_heads=12, evaluate=True, sd_dim=768)
    venc.load_state_dict(specs.synth_weights(specs.vit_shapes("", 96), 0), strict=True)
    venc = venc.cuda().eval()
    images = synth.synth_images(2, 96, 0).cuda()
    sd = synth.synth_tensor("space_dict", (100, 768), 0).cuda().requires_grad_(True)
    with runtime.precision("fp32"):
        y, _ = venc(images, register_blk=2, space_dict=sd, temperature=5.0)
        y.square().sum().backward()
    for i, blk in enumerate(venc.blocks):
        gA = blk.attn.get_attn_gradients()
        if i == 2:
            n_in = 37 if i == 0 else int(venc.blocks[i - 1].last_prune["indices"].shape[1]) + 2
            assert gA is not None and tuple(gA.shape) == (2, 12, n_in, n_in) and torch.isfinite(gA).all() and float(gA.abs().max()) > 0
            assert tuple(blk.attn.get_attention_map().shape) == tuple(gA.shape)
        else:
            assert gA is None


# ---- dropout / DropPath of the training forward (round 5) -----------------------------------------------------------------------------
# The reference's compress_*_dtp loops run model.train(): BERT dropout 0.1 on hidden states and attention probabilities
# (med_config.json:5,7), DropPath in the ViT (vit.py:114,186,205).  The HIP path draws its masks from a counter-based generator
# (Philox4x32-10, csrc/backward.hip); oracle.dropout_mask restates it (pinned to the published Random123 vectors in
# tests/test_oracle_golden.py), so the oracle can run the reference's graph WITH the very masks the kernels used and autograd gives
# the gradients to compare - same fixtures' inputs, same 1e-3 tolerance as the eval-mode tests.

def test_dropout_kernel_equals_philox_restatement(hip):
    from madtp_amd import backward as bw
    from oracle import madtp_oracle as O
    x = _rand(37, 768, seed=1).cuda()
    res = _rand(37, 768, seed=2).cuda()
    for p, seed, site in [(0.1, 1234567890123, 5), (0.5, 2 ** 63 + 11, 2 ** 40 + 3), (0.0, 7, 0)]:
        y = bw.dropout(x, p, seed, site, residual=res)
        m = O.dropout_mask(seed, site, tuple(x.shape), p)
        assert torch.equal(y.cpu(), (x.cpu() * m + res.cpu())), (p, seed, site)
        assert abs(float((m > 0).float().mean()) - (1 - p)) < 0.02
    xs = _rand(6, 50, 768, seed=3).cuda()
    y = bw.dropout(xs, 0.4, 99, 17, per_sample=50 * 768)  # DropPath: one draw per sample
    m = O.dropout_mask(99, 17, (6, 50, 768), 0.4, per_sample=True)
    assert torch.equal(y.cpu(), xs.cpu() * m) and len(set(m.reshape(-1).tolist())) == 2 and abs(float(m.max()) - 1.0 / 0.6) < 1e-6


@pytest.mark.parametrize("mode", TRAIN_MODES)
def test_block_droppath_matches_oracle_with_the_same_masks(hip, mode):
    """Block.forward in .train() mode with drop_path = 0.35 (vit.py:186,205) - pruned and unpruned - against autograd through the
    oracle with the HIP path's masks injected; eval mode and rate 0 are the deterministic block."""
    from madtp_amd import runtime, vit
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(GRAD_CASES[0])
    c = grad_case.build(g)
    p_drop, seed = 0.35, 71  # (a seed under which each of the four DropPath sites below drops exactly one of the two samples)
    blk = vit.Block(768, 12, qkv_bias=True, drop_path=p_drop, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    blk.load_state_dict({k[len(c["prefix"]):]: v for k, v in c["W"].items() if k.startswith(c["prefix"])}, strict=True)
    blk = blk.cuda().train()
    B = c["x"].shape[0]
    for T, base in ((c["T"], 0), (0, 1)):
        x = c["x"].cuda().requires_grad_(True)
        ta = c["token_attn"].cuda().requires_grad_(True) if T > 0 else None
        for p in blk.parameters():
            p.grad = None
        if base == 0:
            runtime.set_dropout_seed(seed)  # (the counter restarts: this call draws base 0, the next base 1)
        drop = lambda code, t, base=base: t * O.dropout_mask(seed, base * 32 + code, tuple(t.shape), p_drop, per_sample=True)
        with _train_mode(mode):
            y = blk(x, False, 0, T, ta)
            Gref = torch.from_numpy(np.random.RandomState(3).uniform(-1, 1, size=tuple(y.shape)).astype(np.float32))
            if T > 0:
                # the oracle keeps tokens in torch.topk's order, the HIP path in ascending order: every TOKEN gets the same upstream
                # gradient in both (grad_case.permute_G), outputs are compared token by token
                ref, yo, oinfo = O.vit_block_grads(c["W"], c["prefix"], c["x"], c["token_attn"], T, Gref, drop=drop)
                own = blk.last_prune["indices"].cpu().numpy()
                assert np.array_equal(np.sort(oinfo["indices"].numpy(), 1), np.sort(own, 1)), "kept sets differ"
                Gm = grad_case.permute_G(Gref, oinfo["indices"].numpy(), own)
                yo = grad_case.permute_G(yo, oinfo["indices"].numpy(), own)
            else:
                Gm = Gref
            (y * Gm.cuda()).sum().backward()
        if T > 0:
            grads = {"x": x.grad, "token_attn": ta.grad}
        else:
            Wl = {k: v.clone().requires_grad_(True) for k, v in c["W"].items() if k.startswith(c["prefix"])}
            xl = c["x"].clone().requires_grad_(True)
            yo, _ = O.vit_block(Wl, c["prefix"], xl, 0, None, drop=drop)
            (yo * Gm).sum().backward()
            ref = {"x": xl.grad}
            ref.update({k[len(c["prefix"]):]: v.grad for k, v in Wl.items()})
            grads = {"x": x.grad}
        grads.update({k: p.grad for k, p in blk.named_parameters()})
        assert _rel(y.detach().cpu(), yo.detach()) < 1e-4
        for name, r in ref.items():
            assert _rel(grads[name].cpu(), r) < 1e-3, (T, name)
        for code in (0, 1):
            assert int((O.dropout_mask(seed, base * 32 + code, (B, 1, 1), p_drop, per_sample=True) == 0).sum()) == 1, "one sample's branch dropped"
    blk.eval()
    with _train_mode(mode), torch.no_grad():
        y_eval = blk(c["x"].cuda(), False, 0, 0, None)
    blk.train()
    blk.drop_path_rate = 0.0
    with _train_mode(mode):
        y_p0 = blk(c["x"].cuda().requires_grad_(True), False, 0, 0, None)
    assert _rel(y_p0.detach().cpu(), y_eval.cpu()) < 1e-5


class _LayerDropHook:
    """oracle.bert_layer's `drop` callable carrying the HIP path's masks: site code -> mask of (seed, base * 32 + code).  The masks are
    drawn over the tensors' elements in memory order; behind the pruning step the HIP path keeps the tokens in ascending order, the
    oracle in torch.topk's - there the mask rows are re-ordered so that every TOKEN is dropped as on the HIP path."""

    def __init__(self, O, seed, base, ph, pa, own_indices):
        self.O, self.seed, self.base, self.ph, self.pa, self.own, self.ref = O, seed, base, ph, pa, own_indices, None

    def on_prune(self, info):
        self.ref = info["indices"].numpy() if (info is not None and info.get("indices") is not None) else None

    def __call__(self, code, t):
        from tests import grad_case
        p = self.pa if code in (0, 2, 3) else self.ph
        if p <= 0:
            return t
        m = self.O.dropout_mask(self.seed, self.base * 32 + code, tuple(t.shape), p)
        if code >= 2 and self.ref is not None and self.own is not None:
            if m.dim() == 4:  # attention probabilities [B,H,Lq,Nk]: the query rows
                m = grad_case.permute_G(m.permute(0, 2, 1, 3).contiguous(), self.own, self.ref).permute(0, 2, 1, 3)
            else:
                m = grad_case.permute_G(m, self.own, self.ref)
        return t * m


def _layer_drop_hook(O, seed, base, ph, pa, layer=None):
    info = getattr(layer, "last_prune", None) if layer is not None else None
    own = info["indices"].cpu().numpy() if (info is not None and info.get("pruned")) else None
    return _LayerDropHook(O, seed, base, ph, pa, own)


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", MEDGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in MEDGRAD_CASES])
def test_med_layer_dropout_matches_oracle_with_the_same_masks(hip, path, mode):
    """The MED BertLayer in .train() mode - hidden dropout 0.1 on the self-output, cross-output and FFN output, attention_probs
    dropout 0.1 in self- and cross-attention (med.py:212,248,327) - against autograd through the oracle with the same masks."""
    from madtp_amd import runtime
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_med(g)
    layer = grad_case.build_med_layer(c).train()
    ph, pa, seed = float(layer.output.dropout.p), float(layer.attention.self.dropout.p), 77123
    assert ph == 0.1 and pa == 0.1
    hidden = c["hidden"].cuda().requires_grad_(True)
    ta = c["token_attn"].cuda().requires_grad_(True)
    mask = c["add_mask"].cuda()
    gv, hv = c["g"].cuda(), c["h"].cuda()
    enc = c["enc"].cuda().requires_grad_(True) if c["enc"] is not None else None
    enc_mask = torch.zeros(enc.shape[0], 1, 1, enc.shape[1], device="cuda") if enc is not None else None
    runtime.set_dropout_seed(seed)
    with _train_mode(mode):
        out = layer(hidden, mask, None, enc, enc_mask, None, False, mode=c["mode"], token_attn=ta, reduce_num=0, temperature=c["T"])
        y = out[0]
        O.vit_loss(y, gv, hv).backward()
    grads = {"hidden": hidden.grad, "token_attn": ta.grad}
    if enc is not None:
        grads["enc"] = enc.grad
    grads.update({k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
    ref, yo, _, oinfo = O.bert_layer_grads(c["W"], c["prefix"], c["hidden"], c["add_mask"], c["T"], c["token_attn"], c["g"], c["h"],
                                           layer_num=c["layer"], enc=c["enc"], drop=_layer_drop_hook(O, seed, 0, ph, pa, layer))
    assert tuple(y.shape) == tuple(yo.shape)
    assert abs(float(y.detach().double().norm()) - float(yo.double().norm())) < 1e-4 * float(yo.double().norm())
    with torch.no_grad():
        y_eval = O.bert_layer(c["W"], c["prefix"], c["hidden"], c["add_mask"], c["T"], c["token_attn"], c["enc"], None,
                              "multimodal" if c["enc"] is not None else "text", c["layer"], "med")[0]
    assert tuple(y_eval.shape) != tuple(yo.shape) or float((y_eval - yo).abs().max()) > 1e-2, "dropout changed the output"
    assert _rel(y.detach().cpu().sum(1), yo.sum(1)) < 1e-4  # (the two paths order the kept tokens differently: compare over tokens)
    for name, r in ref.items():
        scale = ref[name[:-8] + "query.bias"] if name.endswith("key.bias") else r
        e = float((grads[name].cpu() - r).abs().max()) / max(float(scale.abs().max()), 1e-12)
        assert e < 1e-3, f"grad {name}: {e:.3e} of its maximum"
    # a second forward draws new masks; the same seed reproduces the first
    with _train_mode(mode), torch.no_grad():
        torch.set_grad_enabled(True)
        y2 = layer(hidden.detach().requires_grad_(True), mask, None, enc, enc_mask, None, False, mode=c["mode"], token_attn=ta.detach(), reduce_num=0, temperature=c["T"])[0]
        runtime.set_dropout_seed(seed)
        y3 = layer(hidden.detach().requires_grad_(True), mask, None, enc, enc_mask, None, False, mode=c["mode"], token_attn=ta.detach(), reduce_num=0, temperature=c["T"])[0]
    assert not torch.equal(y2.detach(), y.detach()) and torch.equal(y3.detach(), y.detach())


@pytest.mark.parametrize("mode", TRAIN_MODES)
@pytest.mark.parametrize("path", NLVRGRAD_CASES, ids=[os.path.basename(c)[:-4] for c in NLVRGRAD_CASES])
def test_nlvr_layer_dropout_matches_oracle_with_the_same_masks(hip, path, mode):
    """The NLVR BertLayer (twin cross-attention, averaged or merged; nlvr_encoder.py:211,259-271,380) in .train() mode against
    autograd through the oracle with the same masks."""
    from madtp_amd import runtime
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(path)
    c = grad_case.build_nlvr(g)
    layer = grad_case.build_nlvr_layer(c).train()
    ph, pa, seed = float(layer.output.dropout.p), float(layer.attention.self.dropout.p), 4242
    hidden = c["hidden"].cuda().requires_grad_(True)
    ta = c["token_attn"].cuda().requires_grad_(True)
    mask = c["add_mask"].cuda()
    enc = [e.cuda().requires_grad_(True) for e in c["enc"]]
    enc_mask = [m.cuda() for m in c["enc_mask"]]
    gv, hv = c["g"].cuda(), c["h"].cuda()
    runtime.set_dropout_seed(seed)
    with _train_mode(mode):
        y = layer(hidden, mask, None, None, enc, enc_mask, None, False, mode="multimodal", token_attn=ta, reduce_num=0,
                  temperature=c["T"])[0]
        O.vit_loss(y, gv, hv).backward()
    grads = {"hidden": hidden.grad, "token_attn": ta.grad, "enc0": enc[0].grad, "enc1": enc[1].grad}
    grads.update({k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
    ref, yo, _, _ = O.bert_layer_grads(c["W"], c["prefix"], c["hidden"], c["add_mask"], c["T"], c["token_attn"], c["g"], c["h"],
                                       layer_num=c["layer"], variant="nlvr", enc=c["enc"], enc_mask=c["enc_mask"],
                                       drop=_layer_drop_hook(O, seed, 0, ph, pa, layer))
    assert abs(float(y.detach().double().norm()) - float(yo.double().norm())) < 1e-4 * float(yo.double().norm())
    for name, r in ref.items():
        scale = ref[name[:-8] + "query.bias"] if name.endswith("key.bias") else r
        e = float((grads[name].cpu() - r).abs().max()) / max(float(scale.abs().max()), 1e-12)
        assert e < 1e-3, f"grad {name}: {e:.3e} of its maximum"


def test_nlvr_model_train_mode_runs_with_dropout(hip):
    """BLIP_NLVR.forward(train=True) in model.train(): the losses are finite, depend on the dropout seed, repeat for the same seed,
    every parameter gets a gradient; model.eval() is the deterministic forward of the fixtures."""
    from madtp_amd import harness, runtime
    model = harness.build_nlvr(224, 0, "cuda")
    images, text, _ = harness.nlvr_inputs(2, 224, 20, 0, "cuda")
    targets = torch.tensor([0, 1]).cuda()
    T = 8.612223847001898
    losses = {}
    for tag, train, seed in (("eval", False, 1), ("a", True, 1), ("a2", True, 1), ("b", True, 2)):
        model.train(train)
        model.zero_grad(set_to_none=True)
        runtime.set_dropout_seed(seed)
        with runtime.precision("fp32"):
            lo, lf = model(images, text, targets, temperature=T, train=True)
            (lo + 0.1 * lf).backward()
        losses[tag] = (float(lo.detach()), float(lf.detach()))
        assert all(np.isfinite(v) for v in losses[tag])
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.requires_grad)
    assert losses["a"] == losses["a2"] and losses["a"] != losses["b"] and losses["a"] != losses["eval"]


@pytest.mark.parametrize("mode,tol", [("f16", 5e-3), ("bf16", 5e-2)])  # measured on MI355X: 6.1e-4 / 6.2e-3
def test_amp_block_backward_tracks_reference_grads(hip, mode, tol):
    """`--amp` (compress_nlvr_dtp.py:46-53: autocast forward + GradScaler backward), round 6: with runtime.training_amp() the FAST modes
    take the autograd route - every GEMM of the training forward, dgrad and wgrad on the mode's 2-byte operands (f32 accumulation; f32
    parameters, gradients, LayerNorm, softmax and pruning scores, as under autocast).  Gradients are approximate by construction: on the
    reference-recorded block fixture they follow the reference's to the operand format's rounding (when the block keeps the reference's
    token set, which a 2-byte forward need not), a scaled loss (GradScaler) scales them exactly, and without the opt-in the fast modes
    still refuse."""
    from madtp_amd import runtime, vit
    from oracle import madtp_oracle as O
    from tests import grad_case
    g = np.load(GRAD_CASES[0])
    c = grad_case.build(g)
    blk = vit.Block(768, 12, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    blk.load_state_dict({k[len(c["prefix"]):]: v for k, v in c["W"].items() if k.startswith(c["prefix"])}, strict=True)
    blk = blk.cuda()

    def run(scale):
        for p in blk.parameters():
            p.grad = None
        x = c["x"].cuda().requires_grad_(True)
        ta = c["token_attn"].cuda().requires_grad_(True)
        with runtime.precision(mode), runtime.training_amp():
            y = blk(x, False, 0, c["T"], ta)
            info = blk.last_prune
            same = all({int(v) for v in info["indices"][b]} == {int(v) for v in g["blk_idx"][b]} for b in range(x.shape[0]))
            G = grad_case.permute_G(c["G"], g["blk_idx"], info["indices"].cpu().numpy()) if same else c["G"][:, : y.shape[1]]
            ((y * G.cuda()).sum() * scale).backward()
        grads = {"x": x.grad, "token_attn": ta.grad}
        grads.update({k: p.grad.clone() for k, p in blk.named_parameters()})
        return grads, same, y

    grads, same, y = run(1.0)
    assert all(torch.isfinite(v).all() for v in grads.values()) and y.dtype == torch.float32
    assert all(v.dtype == torch.float32 for v in grads.values())
    if same:
        ref, _, _ = O.vit_block_grads(c["W"], c["prefix"], c["x"], c["token_attn"], c["T"], c["G"])
        worst = max((_rel(grads[n].cpu(), r), n) for n, r in ref.items() if not n.endswith("token_attn"))
        print(f"amp {mode}: kept set == reference; worst gradient error {worst[0]:.3e} of its maximum ({worst[1]})")
        assert worst[0] < tol, worst
    else:
        print(f"amp {mode}: the 2-byte forward kept another token set than the fp32 reference (gradients not comparable entry by entry)")
        assert mode == "bf16"  # (f16 reproduces this fixture's sets)
    # GradScaler: a loss scaled by 2^10 gives gradients scaled by exactly 2^10 (powers of two commute with every rounding here)
    g2, same2, _ = run(1024.0)
    assert same2 == same
    for n in ("x", "attn.qkv.weight", "mlp.fc2.weight", "norm1.weight"):
        assert _rel(g2[n] / 1024.0, grads[n]) < (1e-6 if mode == "bf16" else 2e-3), n  # (f16: tiny entries leave the subnormals when scaled)
    # without the opt-in the fast modes refuse, as before
    x = c["x"].cuda().requires_grad_(True)
    with runtime.precision(mode), pytest.raises(NotImplementedError):
        blk(x, False, 0, c["T"], c["token_attn"].cuda())


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["f16", "bf16"])
def test_amp_nlvr_training_step_learns(hip, mode):
    """The headline model's compression training step under runtime.training_amp() (the reference's --amp): losses next to the
    reference's recorded fp32 losses, finite f32 gradients for every parameter that has one in the fixture, norms of the largest
    gradients within the operand format's error, and AdamW steps on the same batch lower the loss."""
    from madtp_amd import harness, runtime
    g = np.load(TRAINSTEP_CASES[0])
    B, size, L, T, seed = int(g["B"]), int(g["size"]), int(g["L"]), float(g["temperature"]), int(g["seed"])
    model = harness.build_nlvr(size, seed, "cuda")
    for p_ in model.parameters():
        p_.requires_grad_(True)
        p_.grad = None
    images, text, _ = harness.nlvr_inputs(B, size, L, seed, "cuda", pad_tail=int(g["pad_tail"]))
    targets = (torch.arange(B) % 2).cuda()
    with runtime.precision(mode), runtime.training_amp():
        lo, lf = model(images, text, targets, temperature=T, train=True)
        assert abs(float(lo.detach()) - float(g["loss_ori"])) < 2e-2 and abs(float(lf.detach()) - float(g["loss_fdt"])) < 5e-2 * max(1.0, float(g["loss_fdt"]))
        (lo + 0.1 * lf).backward()
        grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        names = [k[2:-5] for k in g.files if k.startswith("g_") and k.endswith("_norm")]
        assert names and all(n in grads and torch.isfinite(grads[n]).all() and grads[n].dtype == torch.float32 for n in names)
        big = sorted(names, key=lambda n: -float(g[f"g_{n}_norm"]))[:12]
        for n in big:
            ref = float(g[f"g_{n}_norm"])
            got = float(grads[n].double().norm())
            assert abs(got - ref) < (0.1 if mode == "f16" else 0.35) * ref, (n, got, ref)
        opt = torch.optim.AdamW(model.parameters(), lr=2e-5, weight_decay=0.05)
        losses = [float((lo + 0.1 * lf).detach())]
        opt.step()
        for _ in range(3):
            opt.zero_grad()
            lo, lf = model(images, text, targets, temperature=T, train=True)
            loss = lo + 0.1 * lf
            losses.append(float(loss.detach()))
            loss.backward()
            opt.step()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
