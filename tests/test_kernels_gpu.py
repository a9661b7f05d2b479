"""Per-kernel parity on a real MI355X: each C-ABI entry point vs a plain PyTorch fp32 (CPU) restatement of the
same op on seeded inputs, including ragged / edge shapes.  Integer outputs (indices, counts, compaction maps)
must be bit-exact; f32 kernels within 2e-5 relative; bf16-input GEMMs within bf16 rounding of the inputs."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from madtp_amd import build, hip as h
    build.build(verbose=False)
    h.load()
    assert torch.cuda.is_available()
    h.set_score_fast(0)  # kernel tests check the reference arithmetic; the fast modes' form has its own test below
    return h


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _pad128(w):
    n = w.shape[0]
    npad = (n + 127) // 128 * 128
    if npad == n:
        return w.contiguous()
    return torch.cat([w, torch.zeros(npad - n, w.shape[1], dtype=w.dtype)], 0).contiguous()


@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (197, 768, 768), (300, 100, 768), (1000, 2304, 768), (130, 768, 3072),
                                   (257, 2, 768)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm(hip, M, N, K, dtype):
    a = _rand(M, K, seed=1)
    w = _rand(N, K, seed=2, scale=0.05)
    bias = _rand(N, seed=3)
    res = _rand(M, N, seed=4)
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    a_d, w_d = a.to(td), _pad128(w).to(td)
    ref_a, ref_w = a_d.float(), w.to(td).float()  # reference sees the same (rounded) operands
    for act, fn in ((hip.ACT_NONE, lambda t: t), (hip.ACT_GELU, F.gelu), (hip.ACT_RELU, F.relu),
                    (hip.ACT_QUICK_GELU, lambda t: t * torch.sigmoid(1.702 * t))):
        out = hip.gemm(a_d.cuda(), w_d.cuda(), bias.cuda(), res.cuda(), out_dtype=torch.float32, act=act, n=N)
        ref = fn(ref_a.double() @ ref_w.double().t() + bias.double()).float() + res
        err = (out.cpu() - ref).abs().max().item()
        tol = 2e-5 * max(1.0, ref.abs().max().item()) * (1 if dtype == "f32" else 4)
        assert err < tol, (act, err, tol)
    # no bias / no residual / bf16 out / scaling
    out = hip.gemm(a_d.cuda(), w_d.cuda(), out_dtype=torch.bfloat16, n=N, out_scale=0.5)
    ref = ((ref_a.double() @ ref_w.double().t()) * 0.5).float()
    assert (out.float().cpu() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cfg", [7, 8], ids=["mfma16x16x32", "mfma32x32x16"])
@pytest.mark.parametrize("M,N,K", [(10533, 768, 768), (10533, 776, 768), (10400, 2304, 768), (10533, 768, 3072)])
def test_gemm_wave_specialised(hip, M, N, K, cfg):
    """Both consumer loops of gemm_ws_kernel (madtp_gemm_set_config 7: 16x16x32 fragments - what the dispatch uses; 8: the
    32x32x16 fragments, kept opt-in: measured slower).
    Shapes that dispatch to gemm_ws_kernel (bf16, M >= 4096, >= 200 tiles of 256x128): ragged last row tile (M % 256 = 37),
    a last column tile with 8 valid columns (N = 776), every epilogue (bias / GELU / f32 residual stream / bf16 out / scale)
    and an output that is a column slice of a wider buffer (ldc > N: the descriptor-bounded stores must not touch the rest).
    Reference: float64 matmul of the same bf16-rounded operands on the GPU."""
    td = torch.bfloat16
    a = _rand(M, K, seed=1).to(td).cuda()
    w = _rand(N, K, seed=2, scale=0.05).to(td)
    wp = _pad128(w.float()).to(td).cuda()
    bias, res = _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    core = a.double() @ w.cuda().double().t()
    scale = max(1.0, core.abs().max().item())
    with hip.gemm_config(cfg):
        # f32 residual stream (proj / fc2)
        out = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
        ref = (core + bias.double()).float() + res
        assert (out - ref).abs().max().item() < 1e-4 * scale
        # f32 without residual, scaled
        out = hip.gemm(a, wp, bias, out_dtype=torch.float32, n=N, out_scale=0.5)
        assert (out - ((core + bias.double()) * 0.5).float()).abs().max().item() < 1e-4 * scale
        # bf16 out (qkv) and GELU (fc1)
        out = hip.gemm(a, wp, bias, out_dtype=td, n=N)
        assert (out.float() - (core + bias.double()).float()).abs().max().item() < 1e-2 * scale
        out = hip.gemm(a, wp, bias, out_dtype=td, act=hip.ACT_GELU, n=N)
        assert (out.float() - F.gelu(core + bias.double()).float()).abs().max().item() < 1e-2 * scale
        # strided output: columns [8, 8+N) of a wider buffer, the rest must stay untouched
        for odt in (torch.float32, td):
            wide = torch.full((M, N + 24), 7.0, device="cuda", dtype=odt)
            hip.gemm(a, wp, bias, out_dtype=odt, n=N, out=wide[:, 8:8 + N])
            assert (wide[:, 8:8 + N].float() - (core + bias.double()).float()).abs().max().item() < 1e-2 * scale
            assert torch.all(wide[:, :8] == 7.0) and torch.all(wide[:, 8 + N:] == 7.0)


@pytest.mark.parametrize("M,N,K", [(10533, 768, 768), (10533, 776, 768), (6000, 2304, 768), (10533, 768, 3072), (300, 100, 768),
                                   (25216, 3072, 768)])
@pytest.mark.parametrize("cfg", [6, 9, 10], ids=["lockstep", "pingpong", "pingpong192"])
def test_gemm_256x256(hip, M, N, K, cfg):
    """(cfg 10, round 5: the ping-pong kernel on 192 x 256 tiles - gemm_pp.hip FA = 3 - same checks, ragged last 192-row tile.)
    gemm_sq_kernel (256x256 tiles, forced with madtp_gemm_set_config(6)) and gemm_pp_kernel (the same tile with the
    two-wave-row ping-pong main loop, config 9; repeated launches must give identical bits - a race between the run-ahead
    LDS-DMA stream and the fragment reads would not): ragged last row tile, a last column tile with 8 / 4
    valid columns, N below one tile (W rows past the 128-row padding are dropped by the buffer descriptor), every epilogue and
    a strided output.  Reference: float64 matmul of the same bf16-rounded operands on the GPU."""
    td = torch.bfloat16
    a = _rand(M, K, seed=1).to(td).cuda()
    w = _rand(N, K, seed=2, scale=0.05).to(td)
    wp = _pad128(w.float()).to(td).cuda()
    bias, res = _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    core = a.double() @ w.cuda().double().t()
    scale = max(1.0, core.abs().max().item())
    with hip.gemm_config(cfg):
        out = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
        assert (out - ((core + bias.double()).float() + res)).abs().max().item() < 1e-4 * scale
        out = hip.gemm(a, wp, bias, out_dtype=torch.float32, n=N, out_scale=0.5)
        assert (out - ((core + bias.double()) * 0.5).float()).abs().max().item() < 1e-4 * scale
        out = hip.gemm(a, wp, None, out_dtype=td, n=N)
        assert (out.float() - core.float()).abs().max().item() < 1e-2 * scale
        out = hip.gemm(a, wp, bias, out_dtype=td, act=hip.ACT_GELU, n=N)
        assert (out.float() - F.gelu(core + bias.double()).float()).abs().max().item() < 1e-2 * scale
        for odt in (torch.float32, td):
            wide = torch.full((M, N + 24), 7.0, device="cuda", dtype=odt)
            hip.gemm(a, wp, bias, out_dtype=odt, n=N, out=wide[:, 8:8 + N])
            assert (wide[:, 8:8 + N].float() - (core + bias.double()).float()).abs().max().item() < 1e-2 * scale
            assert torch.all(wide[:, :8] == 7.0) and torch.all(wide[:, 8 + N:] == 7.0)
        # A as a column slice of a wider matrix (lda > K)
        big = _rand(M, K + 128, seed=9).to(td).cuda()
        out = hip.gemm(big[:, 64:64 + K], wp, bias, out_dtype=torch.float32, n=N)
        ref = (big[:, 64:64 + K].double() @ w.cuda().double().t() + bias.double()).float()
        assert (out - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    # the automatic dispatch gives the same values whichever kernel it picks (same k order per accumulator)
    auto = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
    with hip.gemm_config(cfg):
        sq = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
    assert (auto - sq).abs().max().item() < 1e-4 * scale
    if cfg in (9, 10):
        with hip.gemm_config(6):
            lock = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
        assert torch.equal(lock, sq)  # same k order per accumulator as the lockstep kernel
        with hip.gemm_config(cfg):
            for _ in range(20):
                assert torch.equal(hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N), sq)


@pytest.mark.parametrize("M,N,K", [(12288, 768, 3072),   # 288 tiles: one round + 4 tail tiles per XCD cut into 8 pieces
                                   (14208, 768, 768),    # 336 tiles: 10 tail tiles per XCD, 3 pieces of 4 slabs
                                   (25216, 768, 768),    # 594 tiles: two rounds + a tail of 2 pieces; XCDs with 10 and 11 tail tiles
                                   (11000, 2304, 768),   # ragged last row tile inside a split tail tile
                                   (8200, 1100, 1024)])   # uneven pieces (16 slabs in 5 pieces), N % 8 != 0: the scalar epilogue
def test_gemm_stream_k_tail(hip, M, N, K):
    """Stream-K tail of the wave-specialised kernel (madtp_gemm_set_config(5)) against the same kernel without it (7) and a
    float64 product of the same bf16 operands: every epilogue, and repeated launches on one workspace (the tickets reset
    themselves).  The pieces are summed in piece order, so two launches give identical bits."""
    td = torch.bfloat16
    a = _rand(M, K, seed=1).to(td).cuda()
    w = _rand(N, K, seed=2, scale=0.05).to(td)
    wp = _pad128(w.float()).to(td).cuda()
    bias, res = _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    core = a.double() @ w.cuda().double().t()
    scale = max(1.0, core.abs().max().item())
    with hip.gemm_config(7):
        plain = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
    with hip.gemm_config(5):
        sk = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
        assert (sk - ((core + bias.double()).float() + res)).abs().max().item() < 1e-4 * scale
        assert (sk - plain).abs().max().item() < 2e-5 * scale
        for _ in range(3):
            again = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
            assert torch.equal(again, sk)
        out = hip.gemm(a, wp, bias, out_dtype=td, act=hip.ACT_GELU, n=N)
        assert (out.float() - F.gelu(core + bias.double()).float()).abs().max().item() < 1e-2 * scale
        out = hip.gemm(a, wp, None, out_dtype=td, n=N)
        assert (out.float() - core.float()).abs().max().item() < 1e-2 * scale
        # a second stream has its own workspace
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            other = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(other, sk)


def test_gemm_wave_specialised_scalar_epilogue(hip):
    """N = 100 (not a multiple of 8) on 60000 rows: the wave-specialised kernel with its scalar fallback epilogue (the shape of a
    `x @ space_dict^T` product done as a plain GEMM), f32 and bf16 outputs, bias and residual."""
    M, N, K = 60000, 100, 768
    td = torch.bfloat16
    a = _rand(M, K, seed=1).to(td).cuda()
    w = _rand(N, K, seed=2, scale=0.05).to(td)
    wp = _pad128(w.float()).to(td).cuda()
    bias, res = _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    core = a.double() @ w.cuda().double().t() + bias.double()
    out = hip.gemm(a, wp, bias, res, out_dtype=torch.float32, n=N)
    assert (out - (core.float() + res)).abs().max().item() < 1e-4 * max(1.0, core.abs().max().item())
    out = hip.gemm(a, wp, bias, out_dtype=td, n=N)
    assert (out.float() - core.float()).abs().max().item() < 1e-2 * max(1.0, core.abs().max().item())


@pytest.mark.parametrize("dtype,Nq,Nk", [("bf16", 20, 84), ("bf16", 35, 190), ("bf16", 20, 577), ("f32", 20, 84)])
def test_attention_pair(hip, dtype, Nq, Nk):
    """madtp_attention_pair (one launch for bf16 / <= 256 keys, two launches otherwise) == two madtp_attention calls, with
    different masks per branch and q / k|v given as column slices of fused projections."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    B, H, D = 5, 12, 768
    q2 = _rand(B * Nq, 2 * D, seed=1).to(td).cuda()                      # [q0|q1]
    kv0, kv1 = _rand(B * Nk, 2 * D, seed=2).to(td).cuda(), _rand(B * Nk, 2 * D, seed=3).to(td).cuda()  # [k|v] per branch
    m0 = torch.zeros(B, Nk); m0[1, Nk // 2:] = -10000.0
    m1 = torch.zeros(B, Nk); m1[3, 5:9] = -10000.0
    m0, m1 = m0.cuda(), m1.cuda()
    o0, o1 = hip.attention_pair(q2[:, :D], q2[:, D:], kv0[:, :D], kv1[:, :D], kv0[:, D:], kv1[:, D:], B, H, Nq, Nk, 0.125, m0, m1)
    r0, _ = hip.attention(q2[:, :D], kv0[:, :D], kv0[:, D:], B, H, Nq, Nk, 0.125, add_mask=m0)
    r1, _ = hip.attention(q2[:, D:], kv1[:, :D], kv1[:, D:], B, H, Nq, Nk, 0.125, add_mask=m1)
    assert torch.equal(o0, r0) and torch.equal(o1, r1)


@pytest.mark.parametrize("M,N,K", [(5120, 1536, 768), (5043, 1536, 768), (300, 256, 768)])
def test_gemm_pair(hip, M, N, K):
    """madtp_gemm_pair: two problems in one wave-specialised launch (the first two shapes: 2 x 240 tiles, the second with a
    ragged last row tile) or two plain launches (small shape) - either way bit-identical to two madtp_gemm calls."""
    td = torch.bfloat16
    a0, a1 = _rand(M, K, seed=1).to(td).cuda(), _rand(M, K, seed=2).to(td).cuda()
    w0, w1 = _pad128(_rand(N, K, seed=3, scale=0.05)).to(td).cuda(), _pad128(_rand(N, K, seed=4, scale=0.05)).to(td).cuda()
    b0, b1 = _rand(N, seed=5).cuda(), _rand(N, seed=6).cuda()
    c0, c1 = hip.gemm_pair(a0, a1, w0, w1, b0, b1, N)
    assert torch.equal(c0, hip.gemm(a0, w0, b0, n=N)) and torch.equal(c1, hip.gemm(a1, w1, b1, n=N))
    ref = a1.double() @ w1[:N].double().t() + b1.double()
    assert (c1.double() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K,S", [(1280, 768, 3072, 4), (1280, 768, 1536, 4), (300, 768, 768, 2), (77, 768, 768, 12)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_splitk_ln(hip, M, N, K, S, dtype):
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    a = _rand(M, K, seed=1).to(td)
    w = _rand(N, K, seed=2, scale=0.05).to(td)
    bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
    g, b = _rand(N, seed=5), _rand(N, seed=6)
    y32, ybf = hip.gemm_splitk_ln(a.cuda(), _pad128(w.float()).to(td).cuda(), bias.cuda(), res.cuda(), g.cuda(), b.cuda(),
                                  1e-12, S, N, scale=0.5, want_bf16=True)
    t = (0.5 * (a.double() @ w.double().t() + bias.double()) + res.double()).float()
    ref = F.layer_norm(t, (N,), g, b, 1e-12)
    tol = 5e-5 if dtype == "f32" else 5e-5
    assert (y32.cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    assert (ybf.float().cpu() - ref).abs().max().item() < 0.05


def test_gemm_strided_a_and_transpose_detect(hip):
    # A given as a column slice of a wider matrix (lda > K); asymmetric operands catch a swapped C layout
    big = _rand(70, 2304, seed=5)
    a = big[:, 768:1536]
    w = _rand(256, 768, seed=6, scale=0.05)
    out = hip.gemm(big.cuda()[:, 768:1536], w.cuda(), n=256)
    ref = a.double() @ w.double().t()
    assert (out.cpu().double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("rows,dim,eps", [(1, 768, 1e-6), (197 * 3, 768, 1e-12), (5, 512, 1e-5), (7, 1024, 1e-5)])
def test_layernorm(hip, rows, dim, eps):
    x = _rand(rows, dim, seed=7) * 3 + 0.5
    g, b = _rand(dim, seed=8), _rand(dim, seed=9)
    y32, ybf = hip.layernorm(x.cuda(), g.cuda(), b.cuda(), eps, want_f32=True, want_bf16=True)
    ref = F.layer_norm(x, (dim,), g, b, eps)
    assert (y32.cpu() - ref).abs().max().item() < 2e-5
    assert (ybf.float().cpu() - ref).abs().max().item() < 0.04


def test_patchify_assemble(hip):
    B, S, P, D = 3, 64, 16, 768
    img = _rand(B, 3, S, S, seed=10)
    w = _rand(D, 3, P, P, seed=11, scale=0.02)
    bias = _rand(D, seed=12, scale=0.02)
    cls, pos = _rand(1, 1, D, seed=13), _rand(1, (S // P) ** 2 + 1, D, seed=14)
    cols = hip.patchify(img.cuda(), P, torch.float32)
    unf = F.unfold(img, P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P)
    assert torch.equal(cols.cpu(), unf)
    pe = hip.gemm(cols, w.reshape(D, -1).contiguous().cuda(), bias.cuda(), n=D)
    x = hip.assemble_tokens(pe, cls.cuda(), pos.cuda(), B, (S // P) ** 2)
    ref = F.conv2d(img, w, bias, stride=P).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.expand(B, -1, -1), ref], 1) + pos
    assert (x.cpu() - ref).abs().max().item() < 2e-5
    colsb = hip.patchify(img.cuda(), P, torch.bfloat16)
    assert torch.equal(colsb.cpu(), unf.to(torch.bfloat16))


def test_bert_embed(hip):
    B, L, D, V = 3, 20, 768, 1000
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(1))
    we, pe = _rand(V, D, seed=15, scale=0.02), _rand(512, D, seed=16, scale=0.02)
    g, b = _rand(D, seed=17), _rand(D, seed=18)
    y32, ybf = hip.bert_embed(ids.cuda(), we.cuda(), pe.cuda(), g.cuda(), b.cuda(), 1e-12, want_bf16=True)
    ref = F.layer_norm(we[ids] + pe[:L][None], (D,), g, b, 1e-12)
    assert (y32.cpu() - ref).abs().max().item() < 2e-5
    assert (ybf.float().cpu() - ref).abs().max().item() < 0.04


def _ref_attention(qkv, B, N, H, scale, mask=None):
    q, k, v = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask[:, None, None, :]
    p = s.softmax(-1)
    o = p @ v
    colsum = p[:, :, 1:, :].max(1)[0].sum(1)          # [B,N] column mass of head-max P over rows i>=1
    return o.transpose(1, 2).reshape(B * N, H * 64), p, colsum, p[:, :, 0, :], o.norm(dim=-1)


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (3, 20, 12), (1, 1 + 16 * 4, 2), (2, 130, 12), (2, 180, 12), (1, 256, 3), (2, 17, 1),
                                   (1, 577, 12), (2, 901, 3), (1, 257, 2), (1, 1024, 1),
                                   # 641..1024 keys, all heads: the three-sweep order of attn_bf16_large_kernel (round 5; 6 chunks = a
                                   # second key half of two chunks, 901 = VQA's token count)
                                   (2, 700, 12), (1, 901, 12),
                                   # head split (two workgroups per row block, halves merged through the ticket) on many row
                                   # blocks at once: 16 x 5 / 24 x 2 blocks, and an odd head count (no split)
                                   (16, 320, 12), (24, 96, 12), (4, 300, 3)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_self_attention_with_scores(hip, B, N, H, dtype):
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    qkv = (_rand(B * N, 3 * H * 64, seed=20)).to(td)
    mask = (torch.rand(B, N, generator=torch.Generator().manual_seed(3)) > 0.8).float() * -10000.0
    mask[:, 0] = 0
    for m in (None, mask):
        qd = qkv.cuda()
        out, (cs, p0, on) = hip.attention(qd[:, : H * 64], qd[:, H * 64: 2 * H * 64], qd[:, 2 * H * 64:], B, H, N, N,
                                          0.125, add_mask=None if m is None else m.cuda(), scores=True)
        ro, rp, rcol, rp0, rn = _ref_attention(qkv.float(), B, N, H, 0.125, m)
        tol = 3e-5 if dtype == "f32" else 2e-2
        assert (out.float().cpu() - ro).abs().max().item() < tol * max(1, ro.abs().max().item())
        # f32 storage -> exact-f32 MFMA kernel; bf16 storage -> bf16-MFMA fast kernel (P rounded to bf16 for P.V)
        # (bf16 kernels: the head-max is held as f16 pairs, 2^-11 relative per value; the bound scales with the column mass)
        assert (cs.sum(1).cpu() - rcol).abs().max().item() < (1e-4 if dtype == "f32" else 2e-4 * max(1.0, 0.5 * rcol.max().item()))
        assert (p0.cpu() - rp0).abs().max().item() < (1e-5 if dtype == "f32" else 2e-5)
        assert (on.cpu() - rn).abs().max().item() < (1e-4 if dtype == "f32" else 1e-2) * max(1, rn.max().item())


@pytest.mark.parametrize("B,L,Nk", [(2, 20, 143), (3, 35, 197), (1, 5, 9), (64, 20, 130), (2, 20, 577), (1, 35, 901)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_cross_attention(hip, B, L, Nk, dtype):
    H = 12
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    q = _rand(B * L, H * 64, seed=21).to(td)
    kv = _rand(B * Nk, 2 * H * 64, seed=22).to(td)
    kvd = kv.cuda()
    out, side = hip.attention(q.cuda(), kvd[:, : H * 64], kvd[:, H * 64:], B, H, L, Nk, 0.125)
    assert side is None
    qq = q.float().reshape(B, L, H, 64).transpose(1, 2)
    kk, vv = kv.float().reshape(B, Nk, 2, H, 64).permute(2, 0, 3, 1, 4)
    ref = ((qq @ kk.transpose(-1, -2)) * 0.125).softmax(-1) @ vv
    ref = ref.transpose(1, 2).reshape(B * L, H * 64)
    tol = 3e-5 if dtype == "f32" else 2e-2
    assert (out.float().cpu() - ref).abs().max().item() < tol * max(1, ref.abs().max().item())


def _ref_reduce(x, probs, cls_attn, token_attn, T):
    """the reference rule, plain torch (vit.py:123-163)."""
    a = probs[:, :, 1:, 1:].max(1)[0].sum(1)
    a = a / (a.sum(1, keepdim=True) + 1e-8)
    t = token_attn.max(2)[0]
    t = t / (t.sum(1, keepdim=True) + 1e-8)
    score = (a + t + cls_attn) / 3.0
    w = torch.softmax(token_attn / T, dim=1).permute(0, 2, 1)
    thr = torch.bmm(w, score.unsqueeze(-1)).min(1)[0]
    cnt = (score > thr).sum(1)
    return score, thr.squeeze(-1), cnt


@pytest.mark.parametrize("B,N,T", [(4, 197, 1.0), (3, 197, 5.0), (2, 20, 3.0), (5, 131, 10.0), (2, 250, 2.0)])
def test_token_score_select_gather(hip, B, N, T):
    H, K, D = 12, 100, 768
    qkv = _rand(B * N, 3 * H * 64, seed=30)
    x = _rand(B, N, D, seed=31)
    sd = _rand(K, D, seed=32)
    qd = qkv.cuda()
    _, side = hip.attention(qd[:, :768], qd[:, 768:1536], qd[:, 1536:], B, H, N, N, 0.125, scores=True)
    ta_full = hip.gemm(x.reshape(B * N, D).cuda(), _pad128(sd).cuda(), n=128)     # rows incl. CLS, ld 128
    ta_view = ta_full.view(B, N, 128)[:, 1:, :K]
    score, thr, count, kmax = hip.token_score(side, ta_view, T, B, H, N)
    ta_c = ta_view.contiguous()                                                   # dense [B,n,K] layout must agree
    score2, thr2, count2, _ = hip.token_score(side, ta_c, T, B, H, N)
    assert torch.equal(score, score2) and torch.equal(thr, thr2) and torch.equal(count, count2)
    _, rp, _, _, _ = _ref_attention(qkv, B, N, H, 0.125)
    q, k, v = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    o = rp @ v
    hi = o[..., 1:, :].norm(dim=-1)
    cls_attn = (rp[:, :, 0, 1:] * (hi / (hi.sum(1, keepdim=True) + 1e-8))).sum(1)
    token_attn = x[:, 1:, :] @ sd.t()
    rs, rthr, rcnt = _ref_reduce(x[:, 1:], rp, cls_attn, token_attn, T)
    assert (score.cpu() - rs).abs().max().item() < 2e-6 * max(1.0, rs.abs().max().item() * 100)
    assert (thr.cpu() - rthr).abs().max().item() < 1e-6
    # counts are integers: exact unless a score sits within float noise of the threshold
    margin = (rs - rthr[:, None]).abs().min(1)[0]
    safe = margin > 2e-7
    assert torch.equal(count.cpu()[safe].long(), rcnt[safe])
    assert int(kmax.item()) == int(count.max().item())

    # selection on the DEVICE scores must be exactly the stable descending order of those same scores
    n = N - 1
    for k in sorted({1, max(1, n // 2), n - 2, int(kmax.item())} - {0, n}):
        idx, idx_sort, dst, mw = hip.token_select(score, k)
        s = score.cpu()
        order = torch.argsort(s, dim=1, descending=True, stable=True)
        assert torch.equal(idx_sort.cpu(), order)
        keep = torch.zeros(B, n, dtype=torch.bool).scatter_(1, order[:, :k], True)
        ref_idx = torch.stack([torch.nonzero(keep[b]).squeeze(1) for b in range(B)])
        assert torch.equal(idx.cpu(), ref_idx)
        ref_dst = torch.full((B, n), -1, dtype=torch.int32)
        for b in range(B):
            ref_dst[b, ref_idx[b]] = torch.arange(k, dtype=torch.int32)
        assert torch.equal(dst.cpu(), ref_dst)
        w = torch.where(keep, torch.zeros_like(s), s)
        w = w / (w.sum(1, keepdim=True) + 1e-8)
        assert (mw.cpu() - w).abs().max().item() < 1e-6
        y = hip.token_gather(x.cuda(), dst, mw, k)
        ref_y = torch.cat([x[:, :1], torch.gather(x[:, 1:], 1, ref_idx[..., None].expand(-1, -1, D)),
                           torch.bmm(w.unsqueeze(1), x[:, 1:])], 1)
        assert y.shape == (B, k + 2, D)
        assert torch.equal(y[:, : k + 1].cpu(), ref_y[:, : k + 1])           # pure copies: bit-exact
        assert (y[:, k + 1].cpu() - ref_y[:, k + 1]).abs().max().item() < 1e-5
        m2 = _rand(B, N, seed=33)
        mg = hip.mask_gather(m2.cuda(), idx_sort, k)
        ref_m = torch.cat([m2[:, :1], torch.gather(m2[:, 1:], 1, order[:, : k + 1])], 1)
        assert torch.equal(mg.cpu(), ref_m)


@pytest.mark.parametrize("B,N,T", [(4, 197, 1.0), (5, 131, 10.0), (2, 20, 3.0)])
def test_token_score_fast_arithmetic(hip, B, N, T):
    """madtp_set_score_fast: the fast modes' softmax over tokens (log2 units, hardware exp2, one reciprocal per column) against the
    reference arithmetic of the same kernel: same scores (phase A is shared), thresholds within float noise, counts equal wherever
    no score sits within that noise of the threshold."""
    H, K = 12, 100
    g = torch.Generator().manual_seed(77)
    nrt = (N + 15) // 16
    side = tuple(torch.rand(*shp, generator=g).cuda() for shp in ((B, nrt, N), (B, H, N), (B, H, N)))
    ta = torch.randn(B, N - 1, K, generator=g).cuda() * 3.0
    exact = hip.token_score(side, ta, T, B, H, N)
    assert hip.set_score_fast(1) == 0
    try:
        fast = hip.token_score(side, ta, T, B, H, N)
    finally:
        assert hip.set_score_fast(0) == 1
    assert torch.equal(fast[0], exact[0])
    assert (fast[1] - exact[1]).abs().max().item() < 2e-6 * exact[1].abs().max().item() + 1e-9
    margin = (exact[0] - exact[1][:, None]).abs().min(1)[0]
    safe = margin > 1e-6 * exact[1].abs()
    assert safe.any() and torch.equal(fast[2][safe], exact[2][safe])


@pytest.mark.parametrize("B,N", [(1, 20), (7, 131), (130, 81)])
def test_token_score_host_visible_k(hip, B, N):
    """k = max_b count delivered through pinned host memory (last workgroup publishes, host spins) equals the device
    results, repeatedly (the ticket counter re-arms itself) and with other work queued behind it."""
    H, K, D = 12, 100, 768
    qkv = _rand(B * N, 3 * H * 64, seed=50).cuda()
    x = _rand(B, N, D, seed=51).cuda()
    sd = _pad128(_rand(K, D, seed=52)).cuda()
    _, side = hip.attention(qkv[:, :768], qkv[:, 768:1536], qkv[:, 1536:], B, H, N, N, 0.125, scores=True)
    ta = hip.gemm(x.view(B * N, D), sd, n=128).view(B, N, 128)[:, 1:, :K]
    ref = hip.token_score(side, ta, 4.0, B, H, N)
    for _ in range(5):
        score, thr, count, k = hip.token_score_sync(side, ta, 4.0, B, H, N)
        junk = hip.gemm(x.view(B * N, D), sd, n=128)  # queued right behind: must not disturb the protocol
        assert k == int(ref[2].max().item()) == int(count.max().item())
        assert torch.equal(score, ref[0]) and torch.equal(thr, ref[1]) and torch.equal(count, ref[2])
    del junk


@pytest.mark.parametrize("B,N,T", [(2, 901, 2.8), (5, 577, 40.0), (40, 577, 3.0), (1, 1024, 1.0)])
def test_token_score_long_sequence_split(hip, B, N, T):
    """Long sequences at small batches run token_score_split_kernel on the host-visible path (G workgroups per sample, split
    by dictionary columns, partial minima combined by the last arriver): same scores bit for bit as the one-workgroup kernel
    (identical phase A), threshold within float noise of it and of the torch reference, counts exact outside the noise band,
    k = max count, and the self-resetting tickets survive repeated launches."""
    H, K, D = 3, 100, 768
    qkv = _rand(B * N, 3 * H * 64, seed=70).cuda()
    x = _rand(B, N, D, seed=71)
    sd = _rand(K, D, seed=72)
    _, side = hip.attention(qkv[:, :H * 64], qkv[:, H * 64:2 * H * 64], qkv[:, 2 * H * 64:], B, H, N, N, 0.125, scores=True)
    ta = hip.gemm(x.view(B * N, D).cuda(), _pad128(sd).cuda(), n=128).view(B, N, 128)[:, 1:, :K]
    one = hip.token_score(side, ta, T, B, H, N)            # one workgroup per sample (no hand-over slot)
    token_attn = ta.cpu()
    for it in range(3):
        score, thr, count, k = hip.token_score_sync(side, ta, T, B, H, N)
        assert torch.equal(score, one[0])
        assert (thr - one[1]).abs().max().item() < 1e-7
        margin = (score - thr[:, None]).abs().min(1)[0]
        safe = (margin > 2e-7).cpu()
        assert torch.equal(count.cpu()[safe], one[2].cpu()[safe])
        assert k == int(count.max().item())
    w = torch.softmax(token_attn.double() / T, dim=1).permute(0, 2, 1)
    rthr = torch.bmm(w, score.cpu().double().unsqueeze(-1)).min(1)[0].squeeze(-1)
    assert (thr.cpu().double() - rthr).abs().max().item() < 1e-7
    rcnt = (score.cpu().double() > rthr[:, None]).sum(1)
    assert torch.equal(count.cpu()[safe].long(), rcnt[safe])


@pytest.mark.parametrize("B,N,k,D", [(3, 197, 120, 768), (2, 20, 7, 768), (2, 131, 129, 512)])
def test_token_gather_fused_layernorm(hip, B, N, k, D):
    """gather + merge with the following LayerNorm fused in == gather, then layernorm (bit for bit)."""
    x = _rand(B, N, D, seed=60).cuda()
    score = _rand(B, N - 1, seed=61).abs().cuda().contiguous()
    gamma, beta = _rand(D, seed=62).cuda(), _rand(D, seed=63).cuda()
    _, _, dst, mw = hip.token_select(score, k)
    y = hip.token_gather(x, dst, mw, k)
    h32, hlp = hip.layernorm(y, gamma, beta, 1e-6, want_f32=True, want_bf16=True)
    y2, g32, glp = hip.token_gather_ln(x, dst, mw, k, gamma, beta, 1e-6)
    assert torch.equal(y, y2) and torch.equal(h32, g32) and torch.equal(hlp, glp)


def test_select_ties_and_extremes(hip):
    # ties resolve to the lower index (stable), all-equal scores, k = n
    s = torch.tensor([[0.5, 0.5, 0.1, 0.9, 0.5, 0.1, 0.9, 0.0]] * 2)
    idx, idx_sort, dst, mw = hip.token_select(s.cuda(), 3)
    assert idx_sort.cpu()[0].tolist() == [3, 6, 0, 1, 4, 2, 5, 7]
    assert idx.cpu()[0].tolist() == [0, 3, 6]
    idx, idx_sort, dst, mw = hip.token_select(torch.ones(1, 8).cuda(), 8)
    assert idx.cpu()[0].tolist() == list(range(8)) and (mw.cpu() == 0).all()


def test_query_att_ft(hip):
    B, N, D, K = 3, 131, 768, 100
    x = _rand(B, N, D, seed=40)
    sd = _rand(K, D, seed=41, scale=0.2)
    ta = hip.gemm(x.reshape(B * N, D).cuda(), _pad128(sd).cuda(), n=128)
    tav = ta.view(B, N, 128)[:, 1:, :K]
    xd = x.cuda()
    out = hip.query_att_ft(tav, xd[:, 1:, :])
    inner = x[:, 1:] @ sd.t()
    ref = torch.bmm(torch.softmax((inner / math.sqrt(D)).permute(0, 2, 1), -1), x[:, 1:])
    assert (out.cpu() - ref).abs().max().item() < 2e-5 * max(1, ref.abs().max().item())
    out2 = hip.query_att_ft(tav, xd[:, 1:, :].contiguous(), out=out.clone())
    assert (out2.cpu() - 2 * ref).abs().max().item() < 4e-5 * max(1, ref.abs().max().item())


def test_query_att_ft_multi_segment(hip):
    """The one-launch sum over an encoder's layers equals the per-layer accumulate chain (ragged token counts, >16 segments)."""
    B, D, K = 3, 768, 100
    sd = _rand(K, D, seed=81)
    sdp = _pad128(sd).cuda()
    hi = hip.cast_bf16(sdp)
    lo = hip.cast_bf16((sdp - hi.float()).contiguous())
    pairs, ref, chain = [], 0, None
    for li, N in enumerate([150, 131, 64, 65, 33, 2, 20] + [17] * 12):
        x = _rand(B, N, D, seed=90 + li)
        xd = x.cuda()
        tav = hip.align_logits(xd.view(B * N, D), hi, lo).view(B, N, 128)[:, 1:, :K]
        pairs.append((tav, xd[:, 1:, :]))
        inner = x[:, 1:] @ sd.t()
        ref = ref + torch.bmm(torch.softmax((inner / math.sqrt(D)).permute(0, 2, 1), -1), x[:, 1:])
        chain = hip.query_att_ft(tav, xd[:, 1:, :], out=chain, fast=True)
    out = hip.query_att_ft_multi(pairs)
    scale = max(1, ref.abs().max().item())
    assert (out.cpu() - ref).abs().max().item() < 2e-2 * scale
    assert (out - chain).abs().max().item() < 1e-4 * scale  # same products, different f32 summation order
    out2 = hip.query_att_ft_multi(pairs[:3], out=out.clone())
    part = hip.query_att_ft_multi(pairs[:3])
    assert (out2 - out - part).abs().max().item() < 1e-4 * scale


def test_query_att_ft_multi_f16_split(hip):
    """madtp_query_att_ft_multi_split (the f16x3 mode's att_ft: f16-split token rows and softmax weights, three f16 MFMA products)
    against float64: within 3x the error of the exact-f32 kernel on the same segments (ragged token counts, > 16 segments),
    accumulate form included."""
    B, D, K = 3, 768, 100
    sd = _rand(K, D, seed=81)
    sdp = _pad128(sd).cuda()
    pairs, ref = [], 0
    for li, N in enumerate([150, 131, 64, 65, 33, 2, 20] + [17] * 12):
        x = _rand(B, N, D, seed=90 + li)
        xd = x.cuda()
        tav = hip.gemm(xd.view(B * N, D), sdp, n=128).view(B, N, 128)[:, 1:, :K]     # exact-f32 logits, row pitch 128
        pairs.append((tav, xd[:, 1:, :]))
        inner = tav.double().cpu()
        ref = ref + torch.bmm(torch.softmax((inner / math.sqrt(D)).permute(0, 2, 1), -1), x[:, 1:].double())
    exact = hip.query_att_ft_multi(pairs, exact=True)
    split = hip.query_att_ft_multi(pairs, exact="split")
    e_ex = (exact.double().cpu() - ref).abs().max().item()
    e_sp = (split.double().cpu() - ref).abs().max().item()
    print(f"att_ft vs float64: exact-f32 kernel {e_ex:.3e}, f16-split kernel {e_sp:.3e}")
    assert e_sp <= 3.0 * e_ex + 2e-7 * max(1.0, ref.abs().max().item()), (e_sp, e_ex)
    out2 = hip.query_att_ft_multi(pairs[:3], out=split.clone(), exact="split")
    part = hip.query_att_ft_multi(pairs[:3], exact="split")
    assert (out2 - split - part).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(hip.query_att_ft_multi(pairs, exact="split"), split)   # deterministic


def test_fast_mode_alignment_and_att_ft(hip):
    """bf16x3 split-precision logits (~2^-16 relative) and the bf16-MFMA att_ft."""
    B, N, D, K = 3, 150, 768, 100
    x = _rand(B, N, D, seed=70)
    sd = _rand(K, D, seed=71)
    sdp = _pad128(sd).cuda()
    hi = hip.cast_bf16(sdp)
    lo = hip.cast_bf16((sdp - hi.float()).contiguous())
    xd = x.cuda()
    out = hip.align_logits(xd.view(B * N, D), hi, lo)
    ref = (x.reshape(B * N, D).double() @ sd.double().t()).float()
    err = (out.cpu()[:, :K] - ref).abs().max().item()
    assert err < 3e-5 * ref.abs().max().item() * 4, err
    assert (out.cpu()[:, K:] == 0).all()
    tav = out.view(B, N, 128)[:, 1:, :K]
    af = hip.query_att_ft(tav, xd[:, 1:, :], fast=True)
    inner = x[:, 1:] @ sd.t()
    refa = torch.bmm(torch.softmax((inner / math.sqrt(D)).permute(0, 2, 1), -1), x[:, 1:])
    assert (af.cpu() - refa).abs().max().item() < 2e-2 * max(1, refa.abs().max().item())
    af2 = hip.query_att_ft(tav, xd[:, 1:, :], out=af.clone(), fast=True)
    assert (af2.cpu() - 2 * refa).abs().max().item() < 4e-2 * max(1, refa.abs().max().item())
    # layer-level call, both modes
    ta, ft = hip.query_model(xd, sdp, K, sd_split=(hi, lo))
    assert torch.equal(ta, tav) and (ft.cpu() - refa).abs().max().item() < 2e-2 * max(1, refa.abs().max().item())
    ta32, ft32 = hip.query_model(xd, sdp, K)
    assert (ta32.cpu() - ref.view(B, N, K)[:, 1:]).abs().max().item() < 2e-5 * ref.abs().max().item()
    assert (ft32.cpu() - refa).abs().max().item() < 2e-5 * max(1, refa.abs().max().item())


def test_vector_gather(hip):
    v = _rand(3, 50, 768, seed=60)
    idx = torch.randint(0, 50, (3, 17), generator=torch.Generator().manual_seed(2))
    out = hip.vector_gather(v.cuda(), idx.cuda())
    assert torch.equal(out.cpu(), torch.gather(v, 1, idx[..., None].expand(-1, -1, 768)))


def test_small_elementwise(hip):
    a, b = _rand(8, 768, seed=50), _rand(8, 768, seed=51)
    assert torch.equal(hip.add_scale(a.cuda(), b.cuda(), 0.5).cpu(), (a + b) / 2)
    assert torch.equal(hip.cast_bf16(a.cuda()).cpu(), a.to(torch.bfloat16))


# ---- f16-split ("f16x3") operands: fp32-accurate products on the f16 MFMA ------------------------------------------------
def _unsplit(p, K):
    """f16 planes [..., 2K] -> f32 value P0 + 2^-11 P1 (exact in f64)."""
    return (p[..., :K].double() + p[..., K:].double() / 2048.0)


def test_split_f16_roundtrip(hip):
    """madtp_split_f16: P0 + 2^-11 P1 reproduces an f32 activation to ~2^-23 relative over 12 orders of magnitude;
    madtp_split_f16_weight: (Q0 + Q1) * 2^-s reproduces the weight."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(257, 768, generator=g) * torch.exp(torch.randn(257, 768, generator=g) * 4.0)
    x = x.clamp(-6.0e4, 6.0e4)
    p = hip.split_f16(x.cuda())
    assert p.dtype == torch.float16 and p.shape == (257, 1536)
    back = _unsplit(p.cpu(), 768)
    # error <= max(2^-23 |x|, 2^-36): relative while P1 is a normal f16 number, absolute (far below f32 resolution of O(1)
    # activations) for entries under ~2^-13
    err = (back - x.double()).abs()
    assert (err <= torch.maximum(x.double().abs() * 2.0 ** -22, torch.tensor(2.0 ** -35, dtype=torch.float64))).all()
    w = torch.randn(300, 768, generator=g) * 0.02 * torch.exp(torch.randn(300, 768, generator=g))
    q = hip.split_f16_weight(w.cuda())
    s = q._madtp_w_scale
    assert math.log2(s) == round(math.log2(s)) and q.shape == (300, 2 * 768)
    qc = q.cpu()
    wb = (qc[:, :768].double() + qc[:, 768:].double()) * s
    # relative to the tensor's scale: every weight above 2^-17 of the maximum keeps ~2^-22 relative accuracy
    big = w.abs() > w.abs().max() * 2.0 ** -10
    assert (((wb - w.double()).abs() / w.double().abs())[big]).max().item() < 2.0 ** -21


@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (197, 768, 768), (300, 100, 768), (1000, 2304, 768), (130, 768, 3072),
                                   (257, 2, 768), (10533, 768, 768), (10400, 2304, 768), (10533, 776, 3072)])
def test_gemm_f16x3(hip, M, N, K):
    """F16S GEMM vs a float64 matmul of the ORIGINAL f32 operands: the error must be in the rounding class of an f32 dot product
    (the exact-f32 MFMA kernel is run on the same data as the yardstick), for the small-tile kernels and the wave-specialised
    one (M >= 4096), with bias / activation / residual and with the split epilogue (F16S output feeding the next GEMM)."""
    a = _rand(M, K, seed=1)
    w = _rand(N, K, seed=2, scale=0.05)
    bias = _rand(N, seed=3)
    res = _rand(M, N, seed=4)
    ag, bg, rg = a.cuda(), bias.cuda(), res.cuda()
    ws = hip.split_f16_weight(_pad128(w).cuda())
    asp = hip.split_f16(ag)
    core = ag.double() @ w.cuda().double().t()
    mag = (ag.double().abs() @ w.cuda().double().abs().t())  # sum |a||w|: the scale of the rounding error
    out32 = hip.gemm(ag, _pad128(w).cuda(), bg, rg, out_dtype=torch.float32, n=N)
    ref = (core + bg.double()).float() + rg
    e32 = ((out32 - ref).abs().double() / mag).max().item()
    out = hip.gemm(asp, ws, bg, rg, out_dtype=torch.float32, n=N)
    e16 = ((out - ref).abs().double() / mag).max().item()
    # f32 dot products of K terms err by ~1e-7 * sum|a||w| (cdna_hip_programming.md section 3); allow 3x the exact kernel + ulps
    assert e16 < max(3.0 * e32, 3e-7), (e16, e32)
    for act, fn in ((hip.ACT_GELU, F.gelu), (hip.ACT_RELU, F.relu), (hip.ACT_QUICK_GELU, lambda t: t * torch.sigmoid(1.702 * t))):
        o = hip.gemm(asp, ws, bg, None, out_dtype=torch.float32, act=act, n=N, out_scale=0.5)
        r = (fn(core + bg.double()) * 0.5).float()
        assert ((o - r).abs().double() / mag.clamp_min(1.0)).max().item() < 1e-6, act
    if N % 8 == 0:
        # split output (the fused epilogue of fc1 / LayerNorm-free producers): planes of act(a w^T + b)
        o = hip.gemm(asp, ws, bg, None, out_dtype=torch.float16, act=hip.ACT_GELU, n=N)
        assert o.shape == (M, 2 * N)
        r = F.gelu(core + bg.double())
        assert ((_unsplit(o, N) - r).abs() / mag.clamp_min(1.0)).max().item() < 1e-6
        # strided split output into a wider buffer: nothing outside the two planes is touched
    # split-K partials + fused LayerNorm (the BERT projections)
    if K % 256 == 0 and N % 4 == 0 and N <= 1024 and M <= 2000:
        gamma, beta = 1.0 + 0.1 * _rand(N, seed=7).cuda(), 0.1 * _rand(N, seed=8).cuda()
        y32, ylp = hip.gemm_splitk_ln(asp, ws, bg, rg, gamma, beta, 1e-12, 2, N, scale=0.5, lp=torch.float16)
        r = F.layer_norm(((core + bg.double()) * 0.5 + rg.double()), (N,), gamma.double(), beta.double(), 1e-12)
        assert (y32.double() - r).abs().max().item() < 2e-5
        assert (_unsplit(ylp, N) - y32.double()).abs().max().item() < 1e-6


@pytest.mark.parametrize("cfg", [9, 10], ids=["pingpong", "pingpong192"])
@pytest.mark.parametrize("M,N,K", [(10533, 768, 768), (10400, 2304, 768), (10533, 776, 3072), (12288, 3072, 768)])
def test_gemm_f16x3_pingpong(hip, M, N, K, cfg):
    """Round 5: f16-split operands on gemm_pp_kernel<.., MODE 2, ..> (a k-slab as three slabs of the ping-pong stream: (P0,Q1),
    (P0,Q0), (P1,Q0 2^-11)) on both tile heights, against a float64 product of the ORIGINAL f32 operands with the exact-f32 MFMA
    kernel as the yardstick (the criterion of test_gemm_f16x3), every epilogue incl. the split output, ragged last row / column
    tiles, repeated launches bit-identical (a race of the run-ahead DMA stream would not be), and within f32 rounding of the
    wave-specialised f16-split kernel."""
    a = _rand(M, K, seed=1)
    w = _rand(N, K, seed=2, scale=0.05)
    ag, bg, rg = a.cuda(), _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    ws = hip.split_f16_weight(_pad128(w).cuda())
    asp = hip.split_f16(ag)
    core = ag.double() @ w.cuda().double().t()
    mag = (ag.double().abs() @ w.cuda().double().abs().t())
    out32 = hip.gemm(ag, _pad128(w).cuda(), bg, rg, out_dtype=torch.float32, n=N)
    ref = (core + bg.double()).float() + rg
    e32 = ((out32 - ref).abs().double() / mag).max().item()
    with hip.gemm_config(7):
        ws_out = hip.gemm(asp, ws, bg, rg, out_dtype=torch.float32, n=N)
    with hip.gemm_config(cfg):
        out = hip.gemm(asp, ws, bg, rg, out_dtype=torch.float32, n=N)
        e16 = ((out - ref).abs().double() / mag).max().item()
        assert e16 < max(3.0 * e32, 3e-7), (e16, e32)
        assert ((out - ws_out).abs().double() / mag).max().item() < 3e-7
        for _ in range(10):
            assert torch.equal(hip.gemm(asp, ws, bg, rg, out_dtype=torch.float32, n=N), out)
        o = hip.gemm(asp, ws, bg, None, out_dtype=torch.float32, act=hip.ACT_GELU, n=N, out_scale=0.5)
        r = (F.gelu(core + bg.double()) * 0.5).float()
        assert ((o - r).abs().double() / mag.clamp_min(1.0)).max().item() < 1e-6
        if N % 8 == 0:
            o = hip.gemm(asp, ws, bg, None, out_dtype=torch.float16, act=hip.ACT_GELU, n=N)
            assert o.shape == (M, 2 * N)
            r = F.gelu(core + bg.double())
            assert ((_unsplit(o, N) - r).abs() / mag.clamp_min(1.0)).max().item() < 1e-6
            with hip.gemm_config(7):
                assert ((_unsplit(hip.gemm(asp, ws, bg, None, out_dtype=torch.float16, act=hip.ACT_GELU, n=N), N) - _unsplit(o, N)).abs()
                        / mag.clamp_min(1.0)).max().item() < 3e-7


def test_gemm_pair_f16x3(hip):
    M, N, K = 5043, 1536, 768
    a0, a1 = hip.split_f16(_rand(M, K, seed=1).cuda()), hip.split_f16(_rand(M, K, seed=2).cuda())
    w0 = hip.split_f16_weight(_pad128(_rand(N, K, seed=3, scale=0.05)).cuda())
    w1 = hip.split_f16_weight(_pad128(_rand(N, K, seed=4, scale=0.3)).cuda())  # a different power-of-two scale per problem
    assert w0._madtp_w_scale != w1._madtp_w_scale
    b0, b1 = _rand(N, seed=5).cuda(), _rand(N, seed=6).cuda()
    c0, c1 = hip.gemm_pair(a0, a1, w0, w1, b0, b1, N, out_dtype=torch.float32)
    assert torch.equal(c0, hip.gemm(a0, w0, b0, n=N, out_dtype=torch.float32))
    assert torch.equal(c1, hip.gemm(a1, w1, b1, n=N, out_dtype=torch.float32))


def test_f16s_producers(hip):
    """LayerNorm / gather+LayerNorm / patchify / BERT embeddings emitting f16 planes == split of their f32 outputs."""
    x = _rand(333, 768, seed=1).cuda()
    gamma, beta = 1.0 + 0.1 * _rand(768, seed=2).cuda(), 0.1 * _rand(768, seed=3).cuda()
    y32, ylp = hip.layernorm(x, gamma, beta, 1e-6, want_f32=True, lp=torch.float16)
    assert torch.equal(ylp, hip.split_f16(y32))
    img = _rand(2, 3, 64, 64, seed=4).cuda()
    assert torch.equal(hip.patchify(img, 16, torch.float16), hip.split_f16(hip.patchify(img, 16, torch.float32)))
    ids = torch.randint(0, 1000, (3, 20), generator=torch.Generator().manual_seed(5)).cuda()
    wemb, pemb = _rand(1000, 768, seed=6).cuda(), _rand(64, 768, seed=7).cuda()
    e32, elp = hip.bert_embed(ids, wemb, pemb, gamma, beta, 1e-12, lp=torch.float16)
    assert torch.equal(elp, hip.split_f16(e32))
    B, N, k = 3, 50, 30
    xt = _rand(B, N, 768, seed=8).cuda()
    score = torch.rand(B, N - 1, generator=torch.Generator().manual_seed(9)).cuda()
    _, _, dst_pos, merge_w = hip.token_select(score, k)
    y, h32, hlp = hip.token_gather_ln(xt, dst_pos, merge_w, k, gamma, beta, 1e-6, want_f32=True, lp=torch.float16)
    assert torch.equal(hlp, hip.split_f16(h32))


def test_align_logits_f16x3(hip):
    """f16-split alignment logits (x split in registers, three f16 MFMA products): error in the rounding class of an f32 dot
    product - compared with the exact-f32 MFMA GEMM on the same data - and the layer-level query_model call in this mode."""
    B, N, D, K = 5, 197, 768, 100
    x = _rand(B, N, D, seed=70)
    sd = _rand(K, D, seed=71)
    sdp = _pad128(sd).cuda()
    q = hip.split_f16_weight(sdp)
    q0, q1, sc = q[:, :D].contiguous(), q[:, D:].contiguous(), q._madtp_w_scale
    xd = x.cuda()
    out = hip.align_logits(xd.view(B * N, D), q0, q1, sc)
    ref = xd.view(B * N, D).double() @ sd.cuda().double().t()
    mag = xd.view(B * N, D).double().abs() @ sd.cuda().double().abs().t()
    exact = hip.gemm(xd.view(B * N, D), sdp, n=128)
    e16 = ((out[:, :K].double() - ref).abs() / mag).max().item()
    e32 = ((exact[:, :K].double() - ref).abs() / mag).max().item()
    assert e16 < max(3.0 * e32, 3e-7), (e16, e32)
    assert (out[:, K:] == 0).all()
    ta, ft = hip.query_model(xd, sdp, K, sd_split=(q0, q1, sc))
    assert torch.equal(ta, out.view(B, N, 128)[:, 1:, :K])
    inner = x[:, 1:] @ sd.t()
    refa = torch.bmm(torch.softmax((inner / math.sqrt(D)).permute(0, 2, 1), -1), x[:, 1:])
    assert (ft.cpu() - refa).abs().max().item() < 2e-5 * max(1, refa.abs().max().item())  # att_ft stays on the exact-f32 kernel


def test_token_select_nan_inf_scores_stay_memory_safe(hip):
    """Scores containing NaN / +-Inf / -0 still give fully written permutations (a NaN tied with every token and
    left index slots unwritten): NaNs rank last, every rank is produced exactly once, kept ids ascend."""
    B, n, k = 4, 37, 20
    g = torch.Generator().manual_seed(3)
    score = torch.rand(B, n, generator=g)
    score[0, 5] = float("nan"); score[0, 17] = float("nan")
    score[1, :] = float("nan")
    score[2, 3] = float("inf"); score[2, 4] = -float("inf"); score[2, 9] = -0.0; score[2, 10] = 0.0
    idx, idx_sort, dst, mw = hip.token_select(score.cuda(), k)
    idx, idx_sort, dst = idx.cpu(), idx_sort.cpu(), dst.cpu()
    for b in range(B):
        assert sorted(idx_sort[b].tolist()) == list(range(n)), b          # a permutation: nothing unwritten
        kept = idx[b].tolist()
        assert kept == sorted(kept) and len(set(kept)) == k and set(kept) == set(idx_sort[b, :k].tolist())
        assert sorted(d for d in dst[b].tolist() if d >= 0) == list(range(k))
    assert set(idx_sort[0, -2:].tolist()) == {5, 17}                      # NaNs rank below every number
    assert idx_sort[1].tolist() == list(range(n))                         # all-NaN row: index order
    assert idx_sort[2, 0].item() == 3 and idx_sort[2, -1].item() == 4     # +Inf first, -Inf last
    p9, p10 = idx_sort[2].tolist().index(9), idx_sort[2].tolist().index(10)
    assert p9 + 1 == p10                                                  # -0 == +0: a tie, lower index first
    ref = torch.argsort(score[3], descending=True, stable=True)
    assert idx_sort[3].tolist() == ref.tolist()


def test_k_handover_slots(hip):
    """madtp_token_score_publish / _wait: sequence numbers name per-device slots, a publish that is never waited for leaks one
    slot and exhausting the ring returns MADTP_E_BUSY (-5) instead of blocking; waits may come in any order and from other
    threads; a stale or repeated wait is rejected."""
    import ctypes
    import threading
    lib = hip.load()
    B, H, N, K = 3, 12, 50, 100
    nrt = (N + 15) // 16
    cs = torch.rand(B, nrt, N).cuda(); p0 = torch.rand(B, H, N).cuda(); on = torch.rand(B, H, N).cuda() + 0.1
    ta = torch.randn(B, N - 1, K).cuda()
    score = torch.empty(B, N - 1).cuda(); thr = torch.empty(B).cuda(); cnt = torch.empty(B, dtype=torch.int32).cuda()
    st = torch.cuda.current_stream().cuda_stream

    def publish():
        seq = ctypes.c_int(0)
        rc = lib.madtp_token_score_publish(cs.data_ptr(), nrt, p0.data_ptr(), on.data_ptr(), ta.data_ptr(), K, (N - 1) * K, K, 2.0,
                                           score.data_ptr(), thr.data_ptr(), cnt.data_ptr(), B, H, N, ctypes.byref(seq), st)
        return rc, seq.value

    def wait(seq):
        k = ctypes.c_int32(-1)
        rc = lib.madtp_token_score_wait(seq, cnt.data_ptr(), B, ctypes.byref(k), st)
        return rc, k.value

    _, _, _, k_ref = hip.token_score_sync((cs, p0, on), ta, 2.0, B, H, N)
    seqs = []
    for _ in range(16):
        rc, seq = publish()
        assert rc == 0 and seq != 0
        seqs.append(seq)
    assert len(set(seqs)) == 16
    assert publish()[0] == -5                                   # ring exhausted: an error, not a deadlock
    out = {}
    ths = [threading.Thread(target=lambda s=s: out.__setitem__(s, wait(s))) for s in reversed(seqs)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert all(out[s] == (0, k_ref) for s in seqs)
    assert wait(seqs[0])[0] == -1                               # already consumed
    rc, seq = publish()
    assert rc == 0 and wait(seq) == (0, k_ref)
    assert int(cnt.max().item()) == k_ref


@pytest.mark.parametrize("B,L,V,ld", [(5, 7, 30524, 30528), (3, 2, 1000, 1000), (2, 9, 131, 132)])
def test_lm_loss_and_token_prob(hip, B, L, V, ld):
    """madtp_lm_loss vs F.cross_entropy(label_smoothing=0.1, reduction='none') on the shifted scores summed per sequence
    (med.py:1036-1042) incl. ignored (-100) targets and a fully ignored sequence; madtp_token_prob vs softmax + index_select
    (blip_vqa.py:170-171) on strided rows.  Padding columns [V, ld) hold garbage that must not be read."""
    g = torch.Generator().manual_seed(B * 100 + L)
    buf = torch.randn(B, L, ld, generator=g) * 3.0
    buf[..., V:] = 1e9
    labels = torch.randint(0, V, (B, L), generator=g)
    labels[0, 2:] = -100
    labels[-1, :] = -100
    ref = F.cross_entropy(buf[:, :-1, :V].reshape(-1, V).double(), labels[:, 1:].reshape(-1), reduction="none", label_smoothing=0.1)
    ref = ref.view(B, -1).sum(1).float()
    out = hip.lm_loss(buf.cuda(), labels.cuda(), V, 0.1).cpu()
    assert (out - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (out, ref)
    assert out[-1].item() == 0.0
    tok = torch.randint(0, V, (11,), generator=g)
    rows = buf[:, 0, :].cuda()          # row stride L * ld
    p = hip.token_prob(rows, tok.cuda(), V).cpu()
    refp = torch.softmax(buf[:, 0, :V].double(), 1).index_select(1, tok).float()
    assert (p - refp).abs().max().item() < 1e-6


@pytest.mark.parametrize("B,nb,V,ld", [(3, 3, 30524, 30528), (2, 2, 37, 40), (5, 3, 1000, 1000), (1, 8, 131, 132)])
def test_beam_topk(hip, B, nb, V, ld):
    """madtp_beam_topk vs torch: topk(2 * num_beams) of (log_softmax(logits) + beam_scores) over the num_beams * V candidates of
    every item (transformers 4.15 beam_search), with the EOS suppression of MinLengthLogitsProcessor, on strided rows whose
    padding columns hold garbage, incl. the -1e9 start scores of beams 1.. and a row of all-equal logits (ties -> lower index)."""
    g = torch.Generator().manual_seed(B * 10 + nb)
    buf = torch.randn(B * nb, 2, ld, generator=g) * 2.0
    buf[..., V:] = 1e9
    buf[-1, 0, :V] = 0.25                                   # ties
    bs = torch.randn(B * nb, generator=g)
    bs.view(B, nb)[0, 1:] = -1e9                            # first step of an item
    rows = buf[:, 0, :].cuda()                              # row stride 2 * ld
    for suppress in (-1, 5):
        sc, ix = hip.beam_topk(rows, bs.cuda(), nb, V, suppress_token=suppress)
        lp = torch.log_softmax(buf[:, 0, :V].double(), -1)
        if suppress >= 0:
            lp[:, suppress] = -float("inf")
        tot = (lp + bs.double()[:, None]).view(B, nb * V)
        rs, ri = tot.topk(2 * nb, dim=1)
        assert (sc.cpu().double() - rs).abs().max().item() < 1e-4 * max(1.0, rs[rs > -1e8].abs().max().item())
        mine = ix.cpu().long()
        for b in range(B):   # same candidates; order may differ only between numerically tied scores
            assert sorted(mine[b].tolist()) == sorted(ri[b].tolist()) or \
                (tot[b, mine[b]] - rs[b]).abs().max().item() < 1e-5
        assert (sc[:, :-1] >= sc[:, 1:]).all()
    # all-equal row: the winners of the last item's last beam come in ascending index order
    sc, ix = hip.beam_topk(rows[-nb:], torch.zeros(nb).cuda() - torch.arange(nb).cuda() * 100.0, nb, V)
    if nb == 1:
        assert ix[0].tolist() == list(range(2))


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (3, 20, 12), (1, 65, 2), (2, 130, 12), (2, 82, 12), (1, 256, 3), (2, 17, 1),
                                   (16, 96, 12), (24, 96, 12), (4, 241, 3),
                                   (1, 577, 12), (2, 901, 3), (1, 257, 2), (1, 1024, 1)])   # attn_large_f16s_kernel
def test_attention_f16_split_products(hip, B, N, H):
    """attn_f16s_kernel / attn_large_f16s_kernel (io_dtype MADTP_F16S: f32 storage, QK^T and P.V as three f16 MFMA products of f16-split operands - the
    f16x3 precision mode) against a float64 reference: its error stays within 3x the exact-f32 MFMA kernel's on the same
    inputs (the rounding class of an f32 dot product - the criterion of test_gemm_f16x3), for the context and every score
    side output, with and without a padding mask, incl. the head-split launches (<= 384 row blocks) and a [N,N] causal mask."""
    qkv = _rand(B * N, 3 * H * 64, seed=20)
    mask = (torch.rand(B, N, generator=torch.Generator().manual_seed(3)) > 0.8).float() * -10000.0
    mask[:, 0] = 0
    causal = torch.where(torch.arange(N)[None, :] <= torch.arange(N)[:, None], 0.0, -10000.0).contiguous()
    qd = qkv.cuda()
    q, k, v = qd[:, : H * 64], qd[:, H * 64: 2 * H * 64], qd[:, 2 * H * 64:]
    for m, mqk in ((None, None), (mask, None), (None, causal)):
        if mqk is not None and N > 256:
            continue  # the [N,N] mask operand belongs to the <= 256-key kernels
        kw = dict(add_mask=None if m is None else m.cuda(), scores=True, mask_qk=None if mqk is None else mqk.cuda())
        ex_o, (ex_cs, ex_p0, ex_on) = hip.attention(q, k, v, B, H, N, N, 0.125, **kw)
        sp_o, (sp_cs, sp_p0, sp_on) = hip.attention(q, k, v, B, H, N, N, 0.125, split=True, **kw)
        if mqk is None:
            ro, rp, rcol, rp0, rn = _ref_attention(qkv.double(), B, N, H, 0.125, m)
        else:
            ro, rp, rcol, rp0, rn = _ref_attention_causal(qkv.double(), B, N, H, 0.125, causal)
        for name, ex, sp, ref in (("out", ex_o, sp_o, ro), ("colsum", ex_cs.sum(1), sp_cs.sum(1), rcol), ("p0", ex_p0, sp_p0, rp0),
                                  ("onorm", ex_on, sp_on, rn)):
            e_ex = (ex.double().cpu() - ref).abs().max().item()
            e_sp = (sp.double().cpu() - ref).abs().max().item()
            assert e_sp <= 3.0 * e_ex + 2e-7 * max(1.0, ref.abs().max().item()), (name, e_sp, e_ex)
    # the split kernel is deterministic
    again, _ = hip.attention(q, k, v, B, H, N, N, 0.125, split=True, scores=True)
    first, _ = hip.attention(q, k, v, B, H, N, N, 0.125, split=True, scores=True)
    assert torch.equal(again, first)


def _ref_attention_causal(qkv, B, N, H, scale, causal):
    q, k, v = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale + causal.double()[None, None]
    p = s.softmax(-1)
    o = p @ v
    colsum = p[:, :, 1:, :].max(1)[0].sum(1)
    return o.transpose(1, 2).reshape(B * N, H * 64), p, colsum, p[:, :, 0, :], o.norm(dim=-1)


# ---- the "f16" fast precision mode (plain IEEE f16 operands on the f16 MFMA, MADTP_F16) ---------------------------------------
@pytest.mark.parametrize("M,N,K", [(197, 768, 768), (1000, 2304, 768), (10533, 768, 768), (10400, 2304, 768), (10533, 768, 3072),
                                   (257, 2, 768)])
def test_gemm_f16_operands(hip, M, N, K):
    """MADTP_F16 operands through every GEMM kernel family (small tiles, wave-specialised 256x128, ping-pong 256x256) and every
    epilogue: against a float64 product of the same f16-rounded operands (f32 output: accumulation error only; f16 output: one
    f16 rounding, 2^-11 relative), with the weight pre-scale 2^s of the prepared weights undone by the accumulator scale."""
    from madtp_amd import runtime
    a = _rand(M, K, seed=1).cuda()
    w = _rand(N, K, seed=2, scale=0.05)
    bias, res = _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    with runtime.precision("f16"):
        ad = hip.cast_bf16(a)
        wd = hip.cast_lp_weight(_pad128(w).cuda())
        assert hip.w_scale_of(wd) < 1.0  # |w| <= 0.25: stored as w * 2^s
        ar = hip.lp_to_f32(ad).double()
        wr = (hip.lp_to_f32(wd)[:N].double() * hip.w_scale_of(wd))
        core = ar @ wr.t()
        scale = max(1.0, core.abs().max().item())
        out = hip.gemm(ad, wd, bias, res, out_dtype=torch.float32, n=N)
        assert (out - ((core + bias.double()).float() + res)).abs().max().item() < 1e-4 * scale
        out = hip.gemm(ad, wd, bias, out_dtype=torch.bfloat16, act=hip.ACT_GELU, n=N)  # (a 2-byte container: f16 elements here)
        ref = F.gelu(core + bias.double()).float()
        assert (hip.lp_to_f32(out) - ref).abs().max().item() < 1.5e-3 * scale
        out = hip.gemm(ad, wd, None, out_dtype=torch.bfloat16, n=N)
        assert (hip.lp_to_f32(out) - core.float()).abs().max().item() < 1e-3 * scale
        if M >= 4096:  # every big-tile kernel family on f16 operands: wave-specialised, ping-pong 256x256, ping-pong 192x256
            auto = hip.gemm(ad, wd, bias, res, out_dtype=torch.float32, n=N)
            for cfg in (7, 9, 10):
                with hip.gemm_config(cfg):
                    o = hip.gemm(ad, wd, bias, res, out_dtype=torch.float32, n=N)
                    assert (o - ((core + bias.double()).float() + res)).abs().max().item() < 1e-4 * scale, cfg
                    assert (o - auto).abs().max().item() < 1e-4 * scale
                    o = hip.gemm(ad, wd, bias, out_dtype=torch.bfloat16, act=hip.ACT_GELU, n=N)
                    assert (hip.lp_to_f32(o) - ref).abs().max().item() < 1.5e-3 * scale, cfg
    # the same problem in the bf16 mode is 8x coarser: the f16 operands carry three more significand bits
    with runtime.precision("bf16"):
        ob = hip.gemm(hip.cast_bf16(a), hip.cast_lp_weight(_pad128(w).cuda()), None, out_dtype=torch.float32, n=N)
    exact = a.double() @ w.cuda().double().t()
    with runtime.precision("f16"):
        of = hip.gemm(ad, wd, None, out_dtype=torch.float32, n=N)
    if M * N > 4096:
        assert (of - exact.float()).abs().mean().item() < 0.25 * (ob - exact.float()).abs().mean().item()


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (3, 20, 12), (2, 130, 12), (1, 577, 12), (16, 320, 12), (1, 901, 12)])
def test_self_attention_f16_operands(hip, B, N, H):
    from madtp_amd import runtime
    qkv = _rand(B * N, 3 * H * 64, seed=20)
    with runtime.precision("f16"):
        qd = hip.cast_bf16(qkv.cuda())
        out, (cs, p0, on) = hip.attention(qd[:, : H * 64], qd[:, H * 64: 2 * H * 64], qd[:, 2 * H * 64:], B, H, N, N, 0.125, scores=True)
        qr = hip.lp_to_f32(qd).cpu()
        ro, rp, rcol, rp0, rn = _ref_attention(qr, B, N, H, 0.125, None)
        assert (hip.lp_to_f32(out).cpu() - ro).abs().max().item() < 3e-3 * max(1, ro.abs().max().item())   # bf16 kernels: 2e-2
        assert (p0.cpu() - rp0).abs().max().item() < 2e-5
        assert (on.cpu() - rn).abs().max().item() < 2e-3 * max(1, rn.max().item())


def test_f16_range_flag(hip):
    """A value outside the f16 range must surface as an error, not as NaNs: the producers of f16 planes /
    f16 operands raise the library's range flag; the layer-level calls turn it into MADTP_E_RANGE at their host read of k."""
    from madtp_amd import runtime
    assert hip.range_status() == 0
    x = _rand(64, 768, seed=1).cuda()
    hip.split_f16(x)
    assert hip.range_status() == 0
    x[:, 5] *= 1e5  # one channel far outside the f16 range
    hip.split_f16(x)
    assert hip.range_status() == 1 and hip.range_status() == 0  # reported once, then cleared
    with runtime.precision("f16"):
        hip.cast_bf16(x)
        assert hip.range_status() == 1
        big = hip.gemm(hip.cast_bf16(_rand(300, 768, seed=2).cuda() * 40), hip.cast_lp_weight(_pad128(_rand(768, 768, seed=3)).cuda() * 90),
                       None, out_dtype=torch.bfloat16, n=768)   # outputs ~ 40 * 90 * sqrt(768) = 1e5: infinities in the f16 output
        assert not torch.isfinite(hip.lp_to_f32(big)).all()
        # the vector epilogue of an f16-output GEMM does not test its values (cost); the overflow is caught by the LayerNorm behind
        # the next f32-output GEMM, i.e. still inside the layer
        nxt = hip.gemm(big, hip.cast_lp_weight(_pad128(_rand(768, 768, seed=4, scale=0.05)).cuda()), None, out_dtype=torch.float32, n=768)
        hip.layernorm(nxt, torch.ones(768, device="cuda"), torch.zeros(768, device="cuda"), 1e-6, want_f32=False, lp=torch.bfloat16)
        assert hip.range_status() == 1
    g, b = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
    hip.layernorm(x, g * 1e5, b, 1e-6, want_f32=False, lp=torch.float16)
    assert hip.range_status() == 1


def test_f16_modes_range_error_in_a_block(hip):
    """model level: one LayerNorm channel scaled to 1e5 makes norm1's output leave the f16 range - the block call fails with the
    range error in both f16 modes instead of returning NaNs, and the fp32 mode still runs."""
    from madtp_amd import runtime, vit
    blk = vit.Block(768, 12, qkv_bias=True).cuda()
    x = _rand(2, 50, 768, seed=1).cuda()
    ta = _rand(2, 49, 100, seed=2).cuda()
    with torch.no_grad():
        blk.norm1.weight[7] = 1e5
        for mode in ("f16x3", "f16"):
            with runtime.precision(mode), pytest.raises(RuntimeError, match="f16 range"):
                blk(x, False, 0, 1.0, ta.clone())
            assert hip.range_status() == 0
        with runtime.precision("fp32"):
            assert torch.isfinite(blk(x, False, 0, 1.0, ta.clone())).all()


def test_attention_map_accessor(hip):
    """Block.attn.get_attention_map() (vit.py:57-73, 83): None by default (the kernels never materialise P); with
    keep_attention_map = True the [B,H,N,N] map of the last call, recomputed on demand in exact f32 - against torch on the same
    weights, in a fast precision mode as well."""
    from madtp_amd import runtime, vit
    blk = vit.Block(768, 12, qkv_bias=True).cuda()
    x = _rand(2, 50, 768, seed=1).cuda()
    with torch.no_grad(), runtime.precision("f16"):
        blk(x)
        assert blk.attn.get_attention_map() is None
        blk.attn.keep_attention_map = True
        blk(x)
        P = blk.attn.get_attention_map()
    h = F.layer_norm(x, (768,), blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
    qkv = F.linear(h, blk.attn.qkv.weight, blk.attn.qkv.bias).reshape(2, 50, 3, 12, 64).permute(2, 0, 3, 1, 4)
    ref = ((qkv[0] @ qkv[1].transpose(-2, -1)) * blk.attn.scale).softmax(-1)
    assert P.shape == (2, 12, 50, 50) and (P - ref).abs().max().item() < 2e-6


def test_align_logits_128_row_tile_gives_the_64_row_kernels_bits():
    """align_ws2_kernel (MADTP_ALIGN_ROWS=128; round 6 - the "taller row tile" experiment: measured slower, shipped off,
    profiles/r06_align_rows_ab.txt) accumulates every logit in align_ws_kernel's product order: identical bits for both operand
    flavours (bf16 x 3, f16 x 3) and ragged row counts.  The tile height is a process-wide environment switch, hence two processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fps = {}
    for rows in ("64", "128"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "align_bench.py")], env=dict(os.environ, MADTP_ALIGN_ROWS=rows),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        fps[rows] = [(l.split()[2], l.split()[3], l.split()[-1]) for l in r.stdout.splitlines() if l.startswith("rows=")]
    assert len(fps["64"]) == 18 and fps["64"] == fps["128"]
