// The 256x256 "ping-pong" GEMM kernel (its own translation unit: gemm.hip takes minutes to compile).
#include "gemm_device.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------------------------
// 256x256 "ping-pong" kernel (round 4, "pp"): the tile, wave grid (2 x 4 waves, 128x64 outputs per wave) and epilogue of
// gemm_sq_kernel with the main loop re-cut after the MI355X guide's 8-phase schedule.  gemm_sq_kernel runs its 8 waves in
// lockstep (one barrier per slab): both waves of a SIMD read fragments at the same time and then compete for the matrix pipe
// at the same time, and its ablations put the fragment reads + barriers at 16 of 82 us on top of DMA + MFMA.  Here
//   * a 64-deep K slab is FOUR phases of 16 MFMAs per wave (one 64x32 quadrant of the wave's 128x64 block, both k-steps):
//     Q(a0,b0), Q(a1,b0), Q(a1,b1), Q(a0,b1), fed by fragment reads of 8, 8, 4, 4 ds_read_b128 - a0 / a1 = the two 64-row
//     halves of the wave's A rows, b0 / b1 = the two 32-row halves of its W rows; b0 of the NEXT slab is read in phase 3;
//   * every phase is  [fragment reads + LDS-DMA issue + counted vmcnt]  s_barrier  [16 MFMAs at s_setprio 1]  s_barrier, and
//     the two wave rows (wr = 0 / 1: one wave of each on every SIMD) run ONE barrier apart: while one row's waves issue MFMAs
//     the other row's waves read fragments and issue DMA, so the matrix pipe of a SIMD always has exactly one client;
//   * operands travel as HALF-TILES of 128 rows x 128 B (16 KiB = 2 buffer_load_dwordx4 ... lds per wave), cut by the phase
//     that reads them: B0 (W rows 64 wc + 0..31), A0 (A rows 128 wr + 0..63), A1, B1.  Half-tile n = 4 slab + {B0, A0, A1, B1}
//     is READ in phase n - 1 and lives in ring slot n & 7 (8 x 16 KiB = 128 KiB); one half-tile is issued per phase, SEVEN
//     phases ahead (n issued in phase n - 7), and every phase ends its read part with s_waitcnt vmcnt(10): all but the five
//     youngest half-tiles have landed, i.e. half-tile <= phase + 2, which is read from the NEXT phase on (a landed DMA is
//     visible to the other waves only behind a barrier, and the lagging wave row's wait sits one barrier later).  Re-use:
//     slot n & 7 was last read in phase n - 9 by the lagging row, whose lgkmcnt(0) precedes barrier 2 (n - 9) + 2; the
//     leading row issues half-tile n behind barrier 2 (n - 7) - 1.  The stream simply runs on across tile boundaries (the
//     next tile's first 7 half-tiles fly during the epilogue) and past the workgroup's last tile (rows beyond M read as
//     zeros through the descriptor, nobody reads those slots).
// K must be a multiple of 128 (an even slab count: ring slots are compile-time constants in a body of two slabs).
//
// Round 5, two generalisations of the same loop (the ring, the phase order and every hazard argument above are unchanged):
//   * FA = A fragments per half of a wave's rows: 4 = the 256-row tile, 3 = a 192 x 256 tile (wave block 96 x 64, phases of 12
//     MFMAs).  The half-tile IMAGE keeps its 128 row slots of 128 B - the wave rows' halves sit at slots 0 and 64 - and the
//     16 (4 - FA) unused slots of each half are fetched from beyond the descriptor (zeros, no traffic): fragment addresses, ring
//     slots, DMA instruction counts (so the counted vmcnt) are those of FA = 4.  What it buys is the round count of the
//     N = 768 launches at 11-16 k rows (M = 12288: 144 tiles of 256 x 256 on 256 CUs -> 192 tiles of 192 x 256).
//   * MODE 2 = f16-split operands (common.h: x = P0 + 2^-11 P1, w 2^s = Q0 + Q1): a k-slab is THREE slabs of the stream,
//     (P0, Q1), (P0, Q0), (P1, Q0 2^-11) - plane offsets on the scalar side of the descriptor loads, Q0 2^-11 formed in registers
//     (v_pk_mul_f16 on the 8 W fragment registers per half, behind the phase's lgkmcnt wait) - accumulated in that order into
//     the same accumulators.  P0 and Q0 are staged twice (L2 -> LDS bytes per MFMA = the plain kernel's).
// ABL (timing experiments only, results are wrong unless 0 or 8/16): bit 0 no steady-state LDS-DMA, bit 1 no MFMAs, bit 2 no
// steady-state fragment reads, bit 3 the DMA issue moves from the read part to the head of the MFMA part, bit 4 no s_setprio.
template <int OM, int MODE, int FA = 4, int ABL = 0, bool SPLITK = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs g) {
    static_assert(FA == 4 || FA == 3, "256- or 192-row tiles");
    constexpr bool X3 = MODE == 2, F16 = MODE != 0;
    constexpr bool LP_OUT = OM != OM_F32;
    constexpr int BM = 64 * FA, BN = 256, HT = 128 * ROWB;  // half-tile bytes (128 row slots, 32 FA of them used by A halves)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, gl = gridDim.x >> 3;
    // split-K (round 5, the backward's weight gradients: few tiles, K = every token row of the batch): a unit of work is
    // (tile, K range s of S), s fastest; unit (t, s) writes the f32 partial product of its range to C + s M ldc (no bias /
    // activation / residual - the caller reduces the S slabs in order).  S = 1: units are tiles.  SPLITK instantiation: f16-split
    // operands, f32 output, 256-row tile.
    const int S = SPLITK ? g.splitk : 1;  // (compile-time 1 in the forward's instantiations)
    int t0, nslots;
    xcd_tiles(g.ntm * g.ntn * S, xcd, t0, nslots);
    if (lb >= nslots) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nkp = g.K / 64;  // k-slabs per operand plane
    const int nks = __builtin_amdgcn_readfirstlane(nkp / S);  // k-slabs of a unit

    // ---- fragment read addresses inside a half-tile image (128 rows of 128 B, 16-byte chunk index ^= key(row)) ----
    // A fragment i: row wr 64 + l16 + 16 i, key l16 & 7; W fragment jj: row rw0 + (wfrag_row(1,0) - wfrag_row(0,0)) jj.
    // (Recomputed from an opaque copy of the lane id at the head of every tile: values that live across the epilogue compete
    //  with its 200+ registers, and the allocator spilled these bases INTO the main loop - reloads behind s_waitcnt vmcnt(0).)
    constexpr int WJ1 = (wfrag_row<LP_OUT>(1, 0) - wfrag_row<LP_OUT>(0, 0)) * ROWB;
    const char* ab[2][2];  // [slab parity][k-step]
    const char* wb[2][2];
    auto frag_bases = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int l16 = ln & 15, grp4 = ln >> 4;
        const int ra0 = wr * 64 + l16, rw0 = wc * 32 + wfrag_row<LP_OUT>(0, l16);
        const int a_rd = ra0 * ROWB + ((grp4 ^ (l16 & 7)) << 4);
        const int w_rd = rw0 * ROWB + ((grp4 ^ swz_key<LP_OUT>(rw0)) << 4);
        ab[0][0] = smem + a_rd; ab[0][1] = smem + (a_rd ^ 64); ab[1][0] = smem + 65536 + a_rd; ab[1][1] = smem + 65536 + (a_rd ^ 64);
        wb[0][0] = smem + w_rd; wb[0][1] = smem + (w_rd ^ 64); wb[1][0] = smem + 65536 + w_rd; wb[1][1] = smem + 65536 + (w_rd ^ 64);
    };
    frag_bases();

    // ---- LDS-DMA stream: instruction q of wave w fills rows 8 (2 w + q) .. + 7 of a half-tile ----
    const int npad = ((g.N + 127) / 128) * 128;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)((size_t)g.M * g.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)((size_t)npad * g.ldw * 2), 0x00020000);
    unsigned va[2][2], vw[2][2];  // [half][q]: per-lane source byte offsets at k = 0 of the tile the stream is in
    int is_slot = lb, is_koff = 0, is_kend = nkp * ROWB;  // the stream's unit and its byte range [is_koff, is_kend) of a row's plane
    const int plane_b = g.K * 2;                  // bytes from a row's first plane to its second (f16-split rows: [P0 | P1], [Q0 | Q1])
    int is_s = 0;                                 // f16-split: which of the k-slab's three products the stream is in
    int is_offa = 0, is_offw = X3 ? plane_b : 0;  // scalar byte offsets of the stream's current slab: (P0, Q1) first
    auto tile_offsets = [&]() {
        int tm, tn;
        tile_mn(g, S > 1 ? __builtin_amdgcn_readfirstlane((t0 + is_slot) / S) : t0 + is_slot, tm, tn);
        const unsigned lda2 = (unsigned)g.lda * 2u, ldw2 = (unsigned)g.ldw * 2u;
        int ln = lane;  // opaque copy: the per-lane constants below are recomputed per tile instead of living through the main loop
        asm volatile("" : "+v"(ln));
        const int sub8 = ln >> 3, pos = ln & 7;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int R = (2 * wave + q) * 8 + sub8;  // row of the half-tile image
            const unsigned ca = (unsigned)((pos ^ (R & 7)) << 4), cw = (unsigned)((pos ^ swz_key<LP_OUT>(R)) << 4);
            const unsigned rowa = (unsigned)(tm * BM + (R >> 6) * (32 * FA) + (R & 63)), roww = (unsigned)(tn * BN + (R >> 5) * 64 + (R & 31));
            const bool a_ok = FA == 4 || (R & 63) < 16 * FA;  // unused row slots of a 192-row tile: beyond the descriptor -> zeros
            va[0][q] = a_ok ? rowa * lda2 + ca : 0x80000000u;
            va[1][q] = a_ok ? (rowa + 16u * FA) * lda2 + ca : 0x80000000u;
            vw[0][q] = roww * ldw2 + cw;
            vw[1][q] = (roww + 32u) * ldw2 + cw;
        }
    };
    tile_offsets();
    auto unit_krange = [&]() {  // K range of the stream's unit
        if (S > 1) {
            const int u = t0 + is_slot, sp = __builtin_amdgcn_readfirstlane(u - (u / S) * S);
            is_koff = sp * nks * ROWB;
            is_kend = is_koff + nks * ROWB;
        } else {
            is_koff = 0;
        }
    };
    unit_krange();
    is_offa = is_koff; is_offw = is_koff + (X3 ? plane_b : 0);  // first slab of the first unit: (P0, Q1) at its K range
    bool steady = false;
#define PP_ISSUE(RS, V, SLOT, OFF)                                                                                \
    if (!(ABL & 1) || !steady) {                                                                                  \
        char* d__ = smem + (SLOT) * HT + wave * 2048;                                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, LDS_PTR(d__), 16, V[0], OFF, 0, 0);                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, LDS_PTR(d__ + 1024), 16, V[1], OFF, 0, 0);                    \
    }
#define PP_ISSUE_A(V, SLOT) PP_ISSUE(rs_a, V, SLOT, is_offa)
#define PP_ISSUE_W(V, SLOT) PP_ISSUE(rs_w, V, SLOT, is_offw)
    // the stream moves on to its next slab: the next k-slab - or, f16-split, the next product of this k-slab - or the next tile
#define PP_NEXT_SLAB()                                                                         \
    do {                                                                                       \
        if (!X3 || ++is_s == 3) {                                                              \
            is_s = 0;                                                                          \
            is_koff += ROWB;                                                                   \
            if (is_koff == is_kend) { is_slot += gl; tile_offsets(); unit_krange(); }          \
        }                                                                                      \
        is_offa = is_koff + ((X3 && is_s == 2) ? plane_b : 0);                                 \
        is_offw = is_koff + ((X3 && is_s == 0) ? plane_b : 0);                                 \
    } while (0)

    f32x4 acc[2 * FA][4];
    bf16x8 a0[FA][2], a1[FA][2], b0[2][2], b1[2][2];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 res[1][1];
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_READ_A(R, P, IDX)                                                                   \
    if (!(ABL & 4) || !steady)                                                                 \
    _Pragma("unroll") for (int i = 0; i < FA; ++i) {                                           \
        R[i][0] = *(const bf16x8*)(ab[P][0] + (IDX) * HT + i * 16 * ROWB);                     \
        R[i][1] = *(const bf16x8*)(ab[P][1] + (IDX) * HT + i * 16 * ROWB);                     \
    }
#define PP_READ_W(R, P, IDX)                                                                   \
    if (!(ABL & 4) || !steady)                                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
        R[j][0] = *(const bf16x8*)(wb[P][0] + (IDX) * HT + j * WJ1);                           \
        R[j][1] = *(const bf16x8*)(wb[P][1] + (IDX) * HT + j * WJ1);                           \
    }
#define PP_MFMA(RA, RW, SA, SB)                                                                \
    if (!(ABL & 2))                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
        _Pragma("unroll") for (int i = 0; i < FA; ++i)                                         \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
                acc[(SA) * FA + i][(SB) * 2 + j] = mfma_16x16x32<F16>(RW[j][ks], RA[i][ks], acc[(SA) * FA + i][(SB) * 2 + j]);
    // f16-split, third product of a k-slab: W fragments Q0 -> Q0 2^-11 (behind the lgkmcnt wait that completes their reads)
#define PP_SCALE_W(R, ON)                                                                      \
    if constexpr (X3 && (ON))                                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) { R[j][0] = x3_scale_lo(R[j][0]); R[j][1] = x3_scale_lo(R[j][1]); }
    // one phase: READS, ISSUE run beside the other wave row's MFMAs; the MFMAs (behind PRE, register-only work) beside its reads
#define PP_PHASE(READS, ISSUE, PRE, MFMAS)                                                     \
    {                                                                                          \
        READS                                                                                  \
        PP_FENCE();                                                                            \
        if constexpr (!(ABL & 8)) { ISSUE }                                                    \
        PP_FENCE();                                                                            \
        wait_vmcnt<(ABL & 8) ? 8 : 10>();                                                      \
        PP_FENCE();                                                                            \
        __builtin_amdgcn_s_barrier();                                                          \
        PP_FENCE();                                                                            \
        if constexpr ((ABL & 8) != 0) { ISSUE }                                                \
        PP_FENCE();                                                                            \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                                   \
        PP_FENCE();                                                                            \
        PRE                                                                                    \
        PP_FENCE();                                                                            \
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);                              \
        MFMAS                                                                                  \
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);                              \
        PP_FENCE();                                                                            \
        __builtin_amdgcn_s_barrier();                                                          \
        PP_FENCE();                                                                            \
    }
    // slab of parity P: ring slots 4 P + {0: B0, 1: A0, 2: A1, 3: B1}; the stream issues B1 of the next slab, then B0, A0, A1 of
    // the slab after it (slot parity P again)
    // (SC: this slab is the third product of an f16-split k-slab - its W fragments are scaled by 2^-11 where they are first used)
#define PP_SLAB(P, SC)                                                                                                              \
    PP_PHASE(PP_READ_A(a0, P, 1), PP_ISSUE_W(vw[1], 4 * ((P) ^ 1) + 3); PP_NEXT_SLAB();, PP_SCALE_W(b0, SC), PP_MFMA(a0, b0, 0, 0))   \
    PP_PHASE(PP_READ_A(a1, P, 2), PP_ISSUE_W(vw[0], 4 * (P) + 0);, , PP_MFMA(a1, b0, 1, 0))                                          \
    PP_PHASE(PP_READ_W(b1, P, 3), PP_ISSUE_A(va[0], 4 * (P) + 1);, PP_SCALE_W(b1, SC), PP_MFMA(a1, b1, 1, 1))                        \
    PP_PHASE(PP_READ_W(b0, (P) ^ 1, 0), PP_ISSUE_A(va[1], 4 * (P) + 2);, , PP_MFMA(a0, b1, 0, 1))

    // ---- prologue: half-tiles 0..6 (slab 0 whole, slab 1 without its B1), then b0 of slab 0 ----
    PP_ISSUE_W(vw[0], 0); PP_ISSUE_A(va[0], 1); PP_ISSUE_A(va[1], 2); PP_ISSUE_W(vw[1], 3);
    PP_NEXT_SLAB();
    PP_ISSUE_W(vw[0], 4); PP_ISSUE_A(va[0], 5); PP_ISSUE_A(va[1], 6);
    PP_FENCE();
    wait_vmcnt<10>();  // half-tiles 0 and 1 of this wave
    __builtin_amdgcn_s_barrier();
    PP_FENCE();

    for (int slot = lb; slot < nslots; slot += gl) {
        // the second wave row runs one barrier behind the first inside a tile; the rows re-join for the epilogue (run one after
        // the other - the leading row's next tile waits for the lagging row's stores - two epilogues of four waves each cost a
        // K = 768 tile a third of its time)
        if (wr == 1) __builtin_amdgcn_s_barrier();
        PP_FENCE();
        // Nothing but the accumulators lives across the epilogue (it needs 200+ registers itself; whatever else is live there gets
        // spilled, and the reloads land in the main loop behind s_waitcnt vmcnt(0)): fragment bases, the stream's source offsets
        // and b0 of the first slab (landed and visible since the previous tile's last phase) are set up again per tile.
        if (slot != lb) { frag_bases(); tile_offsets(); }
        PP_READ_W(b0, 0, 0)
        PP_FENCE();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): slot 0 is re-staged from phase 1 on
        PP_FENCE();
#pragma unroll
        for (int i = 0; i < 2 * FA; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = zero4;
        for (int kt = 0; kt < nks; kt += 2) {
            if constexpr (X3) {  // two k-slabs = six slabs of the stream: (P0,Q1) (P0,Q0) (P1,Q0') (P0,Q1) (P0,Q0) (P1,Q0')
                PP_SLAB(0, false)
                PP_SLAB(1, false)
                PP_SLAB(0, true)
                PP_SLAB(1, false)
                PP_SLAB(0, false)
                PP_SLAB(1, true)
            } else {
                PP_SLAB(0, false)
                PP_SLAB(1, false)
            }
            steady = true;
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the other row's last barrier
        PP_FENCE();
        int ctm, ctn;
        const int cu = t0 + slot, ct = S > 1 ? __builtin_amdgcn_readfirstlane(cu / S) : cu;
        tile_mn(g, ct, ctm, ctn);
        const size_t c_off = S > 1 ? (size_t)(cu - ct * S) * g.M * (size_t)(g.ldc < 0 ? -g.ldc : g.ldc) : 0;
        const int m0 = ctm * BM, n0 = ctn * BN + (wc >> 1) * 128;
        // the epilogue's per-lane addresses are loop invariants of the tile loop: left alone the compiler hoists them above the
        // main loop, where every register is taken (it then spilled the fragment bases and reloaded them with s_waitcnt vmcnt(0)
        // at the head of every slab pair - draining the DMA stream); an opaque copy of the lane coordinates keeps them down here
        int lne = lane;
        asm volatile("" : "+v"(lne));
        const int l16e = lne & 15, grp4e = lne >> 4;
#define EPI(ACT)                                                                                                  \
    if constexpr (LP_OUT) {                                                                                       \
        epilogue<OM, ACT, false, 2 * FA, 4, BM, 128>(g, acc, res, m0, n0, wr, wc & 1, l16e, grp4e, c_off);             \
    } else {                                                                                                      \
        if (g.residual) epilogue<OM, ACT, true, 2 * FA, 4, BM, 128>(g, acc, res, m0, n0, wr, wc & 1, l16e, grp4e, c_off); \
        else epilogue<OM, ACT, false, 2 * FA, 4, BM, 128>(g, acc, res, m0, n0, wr, wc & 1, l16e, grp4e, c_off);        \
    }
        switch (g.act) {
            case MADTP_ACT_GELU_ERF: EPI(MADTP_ACT_GELU_ERF) break;
            case MADTP_ACT_QUICK_GELU: EPI(MADTP_ACT_QUICK_GELU) break;
            case MADTP_ACT_RELU: EPI(MADTP_ACT_RELU) break;
            default: EPI(MADTP_ACT_NONE) break;
        }
#undef EPI
    }
    wait_vmcnt<0>();  // the run-ahead half-tiles past the last tile: land before the LDS is released
#undef PP_SLAB
#undef PP_PHASE
#undef PP_SCALE_W
#undef PP_ISSUE_A
#undef PP_ISSUE_W
#undef PP_MFMA
#undef PP_READ_W
#undef PP_READ_A
#undef PP_FENCE
#undef PP_NEXT_SLAB
#undef PP_ISSUE
}

// launcher called by gemm.hip's dispatch (args points at its GemmArgs, same definition from gemm_device.h)
//   om: OM_F32 / OM_BF16 / OM_F16 / OM_F16S;  mode: 0 bf16, 1 f16, 2 f16-split operands;  rows: 256 or 192 (tile height)
__attribute__((visibility("hidden"))) int madtp_gemm_pp_launch(const void* args, int om, int mode, int rows, int grid, void* stream) {
    const GemmArgs& g = *(const GemmArgs*)args;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)8 * 128 * ROWB;
    static int abl = -1;  // MADTP_PP_ABLATE: timing experiments (bf16 operands and output, 256-row tile only; see ABL above)
    if (abl < 0) { const char* e = getenv("MADTP_PP_ABLATE"); abl = e ? atoi(e) : 0; }
#define PP_LAUNCH(OM_, MODE_, FA_, ABL_)                                                                        \
    do {                                                                                                        \
        MADTP_ENSURE_MAX_LDS((gemm_pp_kernel<OM_, MODE_, FA_, ABL_>), lds);                                      \
        hipLaunchKernelGGL((gemm_pp_kernel<OM_, MODE_, FA_, ABL_>), dim3(grid), dim3(512), lds, s, g);           \
    } while (0)
#define PP_ROWS(OM_, MODE_)                                                                                     \
    do { if (rows == 192) PP_LAUNCH(OM_, MODE_, 3, 0); else PP_LAUNCH(OM_, MODE_, 4, 0); } while (0)
    if (rows != 256 && rows != 192) return MADTP_E_BADARG;
    if (g.splitk > 1) {  // split-K partials (the backward's weight gradients)
        if (mode != 2 || om != OM_F32 || rows != 256 || g.bias || g.residual || g.act != MADTP_ACT_NONE) return MADTP_E_DTYPE;
        MADTP_ENSURE_MAX_LDS((gemm_pp_kernel<OM_F32, 2, 4, 0, true>), lds);
        hipLaunchKernelGGL((gemm_pp_kernel<OM_F32, 2, 4, 0, true>), dim3(grid), dim3(512), lds, s, g);
        return 0;
    }
    if (mode == 2) {
        if (om == OM_F16S) PP_ROWS(OM_F16S, 2); else if (om == OM_F32) PP_ROWS(OM_F32, 2); else return MADTP_E_DTYPE;
    } else if (mode == 1) {
        if (om == OM_F16) PP_ROWS(OM_F16, 1); else if (om == OM_F32) PP_ROWS(OM_F32, 1); else return MADTP_E_DTYPE;
    } else if (om == OM_BF16) {
        switch (rows == 256 ? abl : 0) {
#ifdef MADTP_PP_ABLATIONS
            case 1: PP_LAUNCH(OM_BF16, 0, 4, 1); break;
            case 2: PP_LAUNCH(OM_BF16, 0, 4, 2); break;
            case 3: PP_LAUNCH(OM_BF16, 0, 4, 3); break;
            case 4: PP_LAUNCH(OM_BF16, 0, 4, 4); break;
            case 5: PP_LAUNCH(OM_BF16, 0, 4, 5); break;
            case 6: PP_LAUNCH(OM_BF16, 0, 4, 6); break;
            case 8: PP_LAUNCH(OM_BF16, 0, 4, 8); break;
            case 16: PP_LAUNCH(OM_BF16, 0, 4, 16); break;
#endif
            default: PP_ROWS(OM_BF16, 0); break;
        }
    } else if (om == OM_F32) {
        PP_ROWS(OM_F32, 0);
    } else {
        return MADTP_E_DTYPE;
    }
#undef PP_ROWS
#undef PP_LAUNCH
    return 0;
}
