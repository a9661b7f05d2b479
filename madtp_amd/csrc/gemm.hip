// C[M,N] = act(A[M,K] @ W[N,K]^T + bias) * out_scale (+ residual)   -- every nn.Linear on the path.
//
// gfx950 design (MFMA-bound kernel; roofline = dense MFMA peak of the operand dtype):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16 fragments,
//     64 f32 accumulators per lane).
//   * K is walked in 128-BYTE row slabs (64 bf16 / 32 f32): both operand tiles are 128 rows x 128 B = 16 KiB and
//     are fetched HBM -> LDS by direct LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), double
//     buffered: the DMA of slab t+1 is in flight while slab t feeds the matrix cores; one barrier per slab.
//   * LDS image is XOR-swizzled (16-B chunk index ^= row&7).  The DMA destination is lane-linear, so the swizzle
//     is applied to the per-lane SOURCE address and again on the fragment read (both-sides rule).
//   * bf16: v_mfma_f32_16x16x32_bf16, f32 accumulate.  f32 ("parity" mode): v_mfma_f32_16x16x4_f32, which is an
//     exact k-ordered f32 fma chain.  For f32 each lane reads 4 consecutive k with one ds_read_b128 and feeds them
//     to 4 MFMAs: the k-slot <-> k mapping is a permutation shared by A and W, so the dot product is unchanged.
//   * blockIdx -> tile map is XCD-aware: block b runs on XCD b%8; the 8 XCDs take interleaved row panels and each
//     walks all column tiles of a panel back-to-back, so an A panel is fetched from HBM once and then hits in that
//     XCD's private L2 while W (<= 4.7 MB) streams from L2/Infinity Cache.
//   * M edge: source rows are clamped to M-1 (reads stay in bounds, results discarded); W is padded by the caller
//     to a multiple of 128 rows; stores are guarded by row<M, col<N.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

#include "gemm_device.h"
#include "internal.h"

namespace {

template <typename T, int OM, int BM, int BN, int STAGES>
__global__ __launch_bounds__(NTHREADS, (BM + BN) * ROWB * STAGES <= 65536 ? 2 : 1) void gemm_kernel(GemmArgs g) {
    constexpr bool X3 = std::is_same<T, F16S>::value;  // f16-split operands: 3 plane products walked as one slab stream
    constexpr bool LP_OUT = OM != OM_F32;
    constexpr int ESZ = X3 ? 2 : (int)sizeof(T);
    constexpr int FM = BM / 32, FN = BN / 32;          // 16x16 fragments per wave (wave tile = BM/2 x BN/2)
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int PER = (BM + BN) / 32;                // LDS-DMA instructions per wave per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (g.m_dev) {  // device-side row count: the grid was sized for the host's (worst-case) M
        g.M = g.m_mul * *g.m_dev;
        g.ntm = (g.M + BM - 1) / BM;
    }

    // Persistent, XCD-aware tile schedule: block b runs on XCD b%8 (observed dispatch rule; only speed depends on
    // it); XCD x owns the tile range of xcd_tiles() and its workgroups walk it round-robin.
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, gl = gridDim.x >> 3;
    int t0, ntile_x;
    xcd_tiles(g.ntm * g.ntn, xcd, t0, ntile_x);
    // split-K (small-M problems): every output tile is cut into g.splitk K ranges, each its own slot; a slot stores its
    // raw f32 partial tile to C + split*M*ldc and a row kernel (madtp_splitk_ln) reduces them in a fixed order.
    const int S = g.splitk;
    const int nslots = ntile_x * S;
    int slot = lb;
    if (slot >= nslots) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l16 = lane & 15, grp4 = lane >> 4;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nkp = g.K * ESZ / ROWB;             // slabs per operand plane
    const int nk = (X3 ? 2 : 1) * nkp / S;        // staged slab steps per slot (f16-split: [P0|Q1], [P1|Q0] per k-slab; nkp % S == 0)
    const int n_pad_max = g.ntn * BN - 1;

    // LDS byte offsets of this lane's fragment rows (row*128) and their swizzle keys (row&7)
    int a_off[FM], a_key[FM], b_off[FN], b_key[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ra = wr * (BM / 2) + i * 16 + l16;
        a_off[i] = ra * ROWB; a_key[i] = ra & 7;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int rb = wc * (BN / 2) + wfrag_row<LP_OUT>(j, l16);
        b_off[j] = rb * ROWB; b_key[j] = swz_key<LP_OUT>(rb);
    }

    // The operand slabs of ALL tiles of this workgroup form one continuous stream through a ring of STAGES LDS
    // stages: slab q lives in stage q % STAGES and STAGES-1 slabs are always in flight (LDS-DMA), across tile
    // boundaries too, so the pipeline never drains and a tile's epilogue overlaps the next tile's loads.  The ring
    // is synchronised with a COUNTED s_waitcnt vmcnt + raw s_barrier (a __syncthreads() would drain every DMA).
    const int my_slots = (nslots - lb + gl - 1) / gl;
    const long total_slabs = (long)my_slots * nk;
    auto decode = [&](int sl, int& tm0, int& tn0, int& kb) {
        const int ts = sl / S, sp = sl - ts * S, t = t0 + ts;
        tm0 = (t / g.ntn) * BM; tn0 = (t % g.ntn) * BN; kb = sp * nk;
    };
    int is_slot = slot, is_kt = 0, is_stage = 0;  // coordinates / ring stage of the next slab to issue
    int im0, in0, ikb;
    decode(is_slot, im0, in0, ikb);
    long issued = 0;
    // Descriptor-based LDS-DMA (g.desc: both operands are below 4 GiB): the per-lane part of a DMA instruction's source - row
    // within the tile, swizzled 16-byte slot - is a constant VGPR offset, tile origin and K position a scalar offset, and rows
    // past the matrix fall outside the descriptor (they read as zeros; their outputs are never stored).  stage_tile() rebuilt
    // the clamped 64-bit address of every instruction of every slab on the VALU: ~50 of the ~180 instructions of a 64x64 slab
    // step, which at one wave per SIMD (the text encoder's small problems) is what bounds the K loop.
    unsigned roff_a[BM / 32], roff_w[BN / 32];
    {
        const int sub = lane >> 3;
#pragma unroll
        for (int q = 0; q < BM / 32; ++q) {
            const int r = (wave * (BM / 32) + q) * 8 + sub;
            roff_a[q] = (unsigned)r * (unsigned)g.lda * ESZ + (unsigned)(((lane & 7) ^ swz_key<false>(r)) << 4);
        }
#pragma unroll
        for (int q = 0; q < BN / 32; ++q) {
            const int r = (wave * (BN / 32) + q) * 8 + sub;
            roff_w[q] = (unsigned)r * (unsigned)g.ldw * ESZ + (unsigned)(((lane & 7) ^ swz_key<LP_OUT>(r)) << 4);
        }
    }
    auto issue_next = [&]() {
        if (issued < total_slabs && !(g.dbg & 2)) {
            char* st = smem + is_stage * STAGE_BYTES;
            int kba = (ikb + is_kt) * ROWB, kbw = kba;
            if (X3) {  // (a plain `if` on the constant: an `if constexpr` makes this lambda's body a dependent context that the
                       //  compiler's HOST pass substitutes too - the descriptor builtins below then fail there, silently, and the
                       //  kernel template gets no launch stub)
                const int sl = ikb + is_kt, odd = sl & 1, kt = sl >> 1;
                kba = (odd * g.K + kt * 64) * 2;
                kbw = ((1 - odd) * g.K + kt * 64) * 2;
            }
            if (g.desc) {
                stage_tile_desc<BM / 32>(g.A, (unsigned)((size_t)g.M * g.lda * ESZ), roff_a, im0 * g.lda * ESZ + kba,
                                         st + wave * (BM / 32) * 1024);
                stage_tile_desc<BN / 32>(g.W, (unsigned)((size_t)(n_pad_max + 1) * g.ldw * ESZ), roff_w, in0 * g.ldw * ESZ + kbw,
                                         st + A_BYTES + wave * (BN / 32) * 1024);
            } else {
                stage_tile<ESZ, BM>(g.A, im0, g.M - 1, g.lda, kba, st, wave, lane);
                stage_tile<ESZ, BN, LP_OUT>(g.W, in0, n_pad_max, g.ldw, kbw, st + A_BYTES, wave, lane);
            }
        }
        ++issued;  // phantom slabs past the end keep the wait counts uniform (they issue nothing: see tail wait)
        if (++is_stage == STAGES) is_stage = 0;
        if (++is_kt == nk) {
            is_kt = 0;
            is_slot += gl;
            decode(is_slot, im0, in0, ikb);
        }
    };
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p) issue_next();

    int m0, n0, kb_unused;
    decode(slot, m0, n0, kb_unused);
    size_t c_off = S > 1 ? (size_t)(slot % S) * g.M * (g.ldc < 0 ? -g.ldc : g.ldc) : 0;
    long s = 0;
    int cur_stage = 0;
    // f32-output tiles carry the residual stream: its float4s are fetched two slabs before the epilogue so the HBM
    // latency hides under the last MFMAs instead of stalling the store phase.
    f32x4 res[LP_OUT ? 1 : FM][LP_OUT ? 1 : FN];
    bf16x8 keep_p0[X3 ? 2 : 1][X3 ? FM : 1];  // f16-split: the P0 fragments of the even step, used again by the odd step
    const bool res_pref = !LP_OUT && g.residual && g.fast_epi;
    const int kt_pref = nk >= 2 ? nk - 2 : 0;
    while (true) {
        for (int kt = 0; kt < nk; ++kt, ++s) {
            if constexpr (!LP_OUT) {
                if (res_pref && kt == kt_pref) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        int row = m0 + wr * (BM / 2) + i * 16 + l16;
                        row = row < g.M ? row : g.M - 1;
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            int col = n0 + wc * (BN / 2) + 16 * j + 4 * grp4;
                            col = col < g.N ? col : 0;
                            res[i][j] = *(const f32x4*)(g.residual + (size_t)row * g.ldr + col);
                        }
                    }
                }
            }
            // slab s must have landed: this wave's DMAs newer than slab s are those of slabs s+1 .. s+STAGES-2
            if (s + STAGES - 1 <= total_slabs) wait_vmcnt<(STAGES - 2) * PER>();
            else wait_vmcnt<0>();  // tail of the stream: fewer real slabs behind slab s
            __builtin_amdgcn_s_barrier();  // all waves' parts of slab s landed; stage (s-1)%STAGES is free again
            issue_next();                   // slab s+STAGES-1 -> stage (s-1)%STAGES
            const char* sa = smem + cur_stage * STAGE_BYTES;
            const char* sw = sa + A_BYTES;
            if (++cur_stage == STAGES) cur_stage = 0;
            if (g.dbg & 4) continue;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int chunk = kk * 4 + grp4;
                if constexpr (ESZ == 2) {
                    bf16x8 a[FM], b[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i) a[i] = *(const bf16x8*)(sa + a_off[i] + ((chunk ^ a_key[i]) << 4));
#pragma unroll
                    for (int j = 0; j < FN; ++j) b[j] = *(const bf16x8*)(sw + b_off[j] + ((chunk ^ b_key[j]) << 4));
                    if constexpr (X3) {
                        if ((kt & 1) == 0) {  // [P0 | Q1] (a slot starts on an even step): P0 Q1, keep P0
#pragma unroll
                            for (int i = 0; i < FM; ++i) {
                                keep_p0[kk][i] = a[i];
#pragma unroll
                                for (int j = 0; j < FN; ++j) acc[i][j] = mfma_16x16x32<true>(b[j], a[i], acc[i][j]);
                            }
                        } else {              // [P1 | Q0]: P0 Q0 + P1 (Q0 2^-11)
#pragma unroll
                            for (int i = 0; i < FM; ++i)
#pragma unroll
                                for (int j = 0; j < FN; ++j) acc[i][j] = mfma_16x16x32<true>(b[j], keep_p0[kk][i], acc[i][j]);
#pragma unroll
                            for (int j = 0; j < FN; ++j) b[j] = x3_scale_lo(b[j]);
#pragma unroll
                            for (int i = 0; i < FM; ++i)
#pragma unroll
                                for (int j = 0; j < FN; ++j) acc[i][j] = mfma_16x16x32<true>(b[j], a[i], acc[i][j]);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < FM; ++i)
#pragma unroll
                            for (int j = 0; j < FN; ++j)
                                acc[i][j] = mfma_16x16x32<std::is_same<T, F16P>::value>(b[j], a[i], acc[i][j]);
                    }
                } else {
                    f32x4 a[FM], b[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i) a[i] = *(const f32x4*)(sa + a_off[i] + ((chunk ^ a_key[i]) << 4));
#pragma unroll
                    for (int j = 0; j < FN; ++j) b[j] = *(const f32x4*)(sw + b_off[j] + ((chunk ^ b_key[j]) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < FM; ++i)
#pragma unroll
                            for (int j = 0; j < FN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][e], a[i][e], acc[i][j], 0, 0, 0);
                }
            }
        }

        // ---- epilogue of tile (m0,n0): vectors straight from the accumulators ----
#define EPI(ACT)                                                                                                  \
    if constexpr (LP_OUT) {                                                                                       \
        epilogue<OM, ACT, false, FM, FN, BM, BN>(g, acc, res, m0, n0, wr, wc, l16, grp4, c_off);                  \
    } else {                                                                                                      \
        if (g.residual) epilogue<OM, ACT, true, FM, FN, BM, BN>(g, acc, res, m0, n0, wr, wc, l16, grp4, c_off);   \
        else epilogue<OM, ACT, false, FM, FN, BM, BN>(g, acc, res, m0, n0, wr, wc, l16, grp4, c_off);             \
    }
        if (!((g.dbg & 1) && acc[0][0][0] != 12345.678f)) {
            switch (g.act) {
                case MADTP_ACT_GELU_ERF: EPI(MADTP_ACT_GELU_ERF) break;
                case MADTP_ACT_QUICK_GELU: EPI(MADTP_ACT_QUICK_GELU) break;
                case MADTP_ACT_RELU: EPI(MADTP_ACT_RELU) break;
                default: EPI(MADTP_ACT_NONE) break;
            }
        }
#undef EPI
        slot += gl;
        if (slot >= nslots) break;
        decode(slot, m0, n0, kb_unused);
        c_off = S > 1 ? (size_t)(slot % S) * g.M * (g.ldc < 0 ? -g.ldc : g.ldc) : 0;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}


}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Wave-specialised bf16 kernel for the big GEMMs: 256x128 output tile, one 768-thread workgroup per CU =
//   8 CONSUMER waves (4x2, 64x64 each: ds_read_b128 fragments -> MFMA -> epilogue stores, never touch vmcnt) and
//   4 LOADER waves (address generation + LDS-DMA of the operand slabs + counted vmcnt waits, no MFMA).
// Why: (a) the 256-row tile shares one W slab between two 128-row halves, so the L2->LDS volume per flop drops by a
// quarter versus two independent 128x128 workgroups (the 128x128 kernel's DMA stream alone runs at ~20 TB/s of L2
// bandwidth, as long as its MFMA phase); (b) a 3-stage ring (144 KiB) keeps TWO slabs in flight; (c) gfx950 counts
// loads and stores in ONE vmcnt, so in the unspecialised kernel the first slab wait after an epilogue also drains the
// tile's output stores - here the stores belong to waves that never wait on vmcnt; (d) the consumers' issue slots are
// not spent on DMA address arithmetic.  One s_barrier per slab synchronises all twelve waves:
//   loader  : wait own DMAs of slab s (vmcnt(PER): only slab s+1 may still fly) -> barrier s -> issue slab s+2
//             into the stage slab s-1 used (every consumer finished reading it before it arrived at barrier s)
//   consumer: barrier s -> 16 x ds_read_b128 + 32 x MFMA on slab s
// The slab stream is persistent across the workgroup's tiles exactly as in gemm_kernel.
// MADTP_WS_ABLATE (tools/build_ablate.py, timing experiments only - results are wrong): bit 0 drops the steady-state
// fragment reads, bit 1 the per-slab barriers, bit 2 the MFMAs.
#ifndef MADTP_WS_ABLATE
#define MADTP_WS_ABLATE 0
#endif
// -DMADTP_WS_TIMING (ABLATE=wstime tools/build_ablate.py): consumer wave 0 of workgroup 0 accumulates wall_clock64 (100 MHz)
// phase times over its tiles: [0] main loops, [1] epilogues, [2] tiles, [3] kernel total.  tools/gemm_ws_phases.py reads them.
#ifdef MADTP_WS_TIMING
__device__ long long g_ws_dbg[8];
__device__ __forceinline__ long long ws_now() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const long long t = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define WS_NOW() ws_now()
#endif
template <bool X3, int OM, bool M32 = false, bool F16 = false>
__global__ __launch_bounds__(768, 1) void gemm_ws_kernel(GemmArgs g) {
    static_assert(!(X3 && M32), "the 32x32x16 consumer loop exists for the bf16 kernel only");
    static_assert(!(F16 && (X3 || M32)), "plain f16 operands run the 16x16x32 consumer loop");
    constexpr bool MF16 = X3 || F16;  // which 16x16x32 MFMA: f16 (split planes or plain f16 operands) or bf16
    constexpr bool LP_OUT = OM != OM_F32;
    constexpr int ESZ = 2, BM = 256, BN = 128, STAGES = 3, NCW = 8, NLW = 4;
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int PER = (BM + BN) / 8 / NLW;  // 1 KiB DMA instructions per loader wave per slab (12)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, gl = gridDim.x >> 3;
    int t0, nslots;
    const int tiles1 = g.ntm * g.ntn;
    xcd_tiles(g.pair ? 2 * tiles1 : tiles1, xcd, t0, nslots);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkp = g.K * ESZ / ROWB;     // slabs per operand plane
    const int nk = (X3 ? 2 : 1) * nkp;    // staged slab steps per tile (f16-split: [P0|Q1], [P1|Q0] per k-slab)
    // work units of this workgroup: n_full whole tiles (slots lb, lb + gl, ...) and, with a stream-K tail, one K-piece
    // [sk_k0, sk_k1) of tail tile sk_slot (sk_plan above; never on the f16-split kernel - its summation order is part of parity)
    // (integer division runs on the VALU: readfirstlane puts the - uniform - results back into SGPRs)
    int sk_p = 0, sk_rounds = 0;
    if constexpr (!X3) {
        if (g.sk) {
            sk_rounds = __builtin_amdgcn_readfirstlane(nslots / gl);
            sk_p = __builtin_amdgcn_readfirstlane(sk_parts(nslots - sk_rounds * gl, gl, nk));
        }
    }
    const int n_full = __builtin_amdgcn_readfirstlane(sk_p ? sk_rounds : (lb < nslots ? (nslots - lb + gl - 1) / gl : 0));
    const int sk_tile = __builtin_amdgcn_readfirstlane(sk_p ? lb / sk_p : 0);  // tail tile of this workgroup's piece
    const bool has_tail = sk_p && lb < (nslots - sk_rounds * gl) * sk_p;
    const int sk_slot = sk_rounds * gl + sk_tile, sk_part = lb - sk_tile * sk_p;
    const int sk_k0 = __builtin_amdgcn_readfirstlane(has_tail ? sk_part * nk / sk_p : 0);
    const int sk_k1 = __builtin_amdgcn_readfirstlane(has_tail ? (sk_part + 1) * nk / sk_p : 0);
    const int n_units = n_full + (has_tail ? 1 : 0);
    if (n_units == 0) return;
    const int total_slabs = n_full * nk + (sk_k1 - sk_k0);

    if (wave >= NCW) {
        // ------------------------------------------ loader ------------------------------------------
        const int lw = wave - NCW, sub = lane >> 3;
        const int n_pad_max = g.ntn * BN - 1;
        int is_unit = 0, is_stage = 0;
        int is_slot = n_full ? lb : sk_slot, is_kt = n_full ? 0 : sk_k0, is_kend = n_full ? nk : sk_k1;
        int is_odd = 0, is_pk = 0;  // f16-split: parity of the next step ([P0|Q1] or [P1|Q0]) and its k-slab
        int issued = 0;
        const char* srcp[PER];  // per-lane source of each DMA instruction at k = 0 of the current tile
        auto tile_ptrs = [&]() {
            int t = t0 + is_slot;
            const bool second = g.pair && t >= tiles1;
            if (second) t -= tiles1;
            const char* pA = second ? g.A2 : g.A;
            const char* pW = second ? g.W2 : g.W;
            int itm, itn;
            tile_mn(g, t, itm, itn);
            const int im0 = itm * BM, in0 = itn * BN;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int idx = lw * PER + q;                     // 8-row group of the stage image: A groups, then W groups
                const bool is_a = idx < BM / 8;
                const int r = (is_a ? idx : idx - BM / 8) * 8 + sub;  // row within the A / W tile
                const int key = M32 ? swz_key32(r) : (is_a ? (r & 7) : swz_key<LP_OUT>(r));
                int row = (is_a ? im0 : in0) + r;
                const int lim = is_a ? g.M - 1 : n_pad_max;
                row = row < lim ? row : lim;
                srcp[q] = (is_a ? pA : pW) + (size_t)row * (is_a ? g.lda : g.ldw) * ESZ + (((lane & 7) ^ key) << 4);
            }
        };
        tile_ptrs();
        auto issue_next = [&]() {
            if (issued < total_slabs && !(g.dbg & 2)) {
                char* st = smem + is_stage * STAGE_BYTES + lw * PER * 1024;
                if constexpr (X3) {
                    const int offa = (is_odd * g.K + is_pk * 64) * 2, offw = ((1 - is_odd) * g.K + is_pk * 64) * 2;
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(srcp[q] + (lw * PER + q < BM / 8 ? offa : offw)),
                                                         LDS_PTR(st + q * 1024), 16, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(srcp[q] + is_kt * ROWB), LDS_PTR(st + q * 1024), 16, 0, 0);
                }
            }
            ++issued;
            if (++is_stage == STAGES) is_stage = 0;
            if constexpr (X3) {
                if (is_odd) { if (++is_pk == nkp) is_pk = 0; }
                is_odd ^= 1;
            }
            if (++is_kt == is_kend) {
                if (++is_unit < n_full) { is_kt = 0; is_slot += gl; }
                else { is_kt = sk_k0; is_kend = sk_k1; is_slot = sk_slot; }
                if (issued < total_slabs) tile_ptrs();
            }
        };
        issue_next();
        issue_next();
        for (int s = 0; s < total_slabs; ++s) {
            if (s + STAGES - 1 <= total_slabs) wait_vmcnt<(STAGES - 2) * PER>();
            else wait_vmcnt<0>();
            if (!(MADTP_WS_ABLATE & 2)) __builtin_amdgcn_s_barrier();
            issue_next();
        }
        return;
    }

    // ------------------------------------------ consumer ------------------------------------------
    const int grp = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    if constexpr (M32) {
        // 32x32x16 consumer loop: the wave's 64x64 block is 2x2 fragments of 32x32 (4 x 16 accumulators); a 64-deep slab is
        // four 16-deep k-steps, read as two halves X (k-steps 0, 1) and Y (2, 3) of 4 A + 4 W fragments each - the same 16
        // ds_read_b128 per slab as the 16x16x32 loop, half a slab ahead of the MFMAs that use them, but 16 MFMA issues per
        // slab instead of 32.  No stream-K tail on this variant (g.sk == 0).
        const int l32 = lane & 31, h = lane >> 5;
        const int a_rd = (grp * 128 + wr * 64 + l32) * ROWB, a_k = swz_key32(l32);
        const int rw = wc * 64 + wfrag_row32<LP_OUT>(0, l32);
        const int w_rd = A_BYTES + rw * ROWB, w_k = swz_key32(rw);
        f32x16 acc[2][2];
        bf16x8 xa[4], xb[4], ya[4], yb[4];
        int cur_stage = 0;
#define WS32_READ(XA, XB, CH)                                                                      \
    if (!(MADTP_WS_ABLATE & 1) || first)                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
            XA[ks * 2 + i] = *(const bf16x8*)(st + a_rd + i * 32 * ROWB + ((((CH) + 2 * ks + h) ^ a_k) << 4)); \
            XB[ks * 2 + i] = *(const bf16x8*)(st + w_rd + i * 32 * ROWB + ((((CH) + 2 * ks + h) ^ w_k) << 4)); \
        }
#define WS32_MFMA(XA, XB)                                                                          \
    if (!(MADTP_WS_ABLATE & 4))                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(XB[ks * 2 + j], XA[ks * 2 + i], acc[i][j], 0, 0, 0);
        for (int unit = 0; unit < n_units; ++unit) {
            const int slot = lb + unit * gl;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
            {
                constexpr bool first = true;
                const char* st = smem + cur_stage * STAGE_BYTES;
                if (!(MADTP_WS_ABLATE & 2)) __builtin_amdgcn_s_barrier();
                WS32_READ(xa, xb, 0)
                WS32_READ(ya, yb, 4)
                __builtin_amdgcn_sched_barrier(0);
                WS32_MFMA(xa, xb)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
                if (++cur_stage == STAGES) cur_stage = 0;
            }
            for (int kt = 1; kt < nk; ++kt) {
                constexpr bool first = false;
                const char* st = smem + cur_stage * STAGE_BYTES;
                if (!(MADTP_WS_ABLATE & 2)) __builtin_amdgcn_s_barrier();
                WS32_READ(xa, xb, 0)
                __builtin_amdgcn_sched_barrier(0);
                WS32_MFMA(ya, yb)
                __builtin_amdgcn_sched_barrier(0);
                WS32_READ(ya, yb, 4)
                __builtin_amdgcn_sched_barrier(0);
                WS32_MFMA(xa, xb)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
                if (++cur_stage == STAGES) cur_stage = 0;
            }
            { WS32_MFMA(ya, yb) }
            int t = t0 + slot;
            const bool second = g.pair && t >= tiles1;
            if (second) t -= tiles1;
            int ctm, ctn;
            tile_mn(g, t, ctm, ctn);
            const int row_t = ctm * BM + grp * 128 + wr * 64, col_w = ctn * BN + wc * 64;
            if (!((g.dbg & 1) && acc[0][0][0] != 12345.678f)) {
                GemmArgs ge = g;
                if (second) { ge.bias = g.bias2; ge.C = g.C2; ge.acc_scale = g.acc_scale2; }
#define EPI(ACT)                                                                                   \
    if constexpr (LP_OUT) {                                                                        \
        epilogue32<OM, ACT, false>(ge, acc, row_t, col_w, l32, h);                                  \
    } else {                                                                                       \
        if (g.residual) epilogue32<OM, ACT, true>(ge, acc, row_t, col_w, l32, h);                   \
        else epilogue32<OM, ACT, false>(ge, acc, row_t, col_w, l32, h);                             \
    }
                switch (g.act) {
                    case MADTP_ACT_GELU_ERF: EPI(MADTP_ACT_GELU_ERF) break;
                    case MADTP_ACT_QUICK_GELU: EPI(MADTP_ACT_QUICK_GELU) break;
                    case MADTP_ACT_RELU: EPI(MADTP_ACT_RELU) break;
                    default: EPI(MADTP_ACT_NONE) break;
                }
#undef EPI
            }
        }
#undef WS32_READ
#undef WS32_MFMA
        return;
    }
    const int l16 = lane & 15, grp4 = lane >> 4;
    f32x4 acc[4][4];
    int a_off[4], a_key[4], b_off[4], b_key[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = grp * 128 + wr * 64 + i * 16 + l16;
        a_off[i] = ra * ROWB; a_key[i] = ra & 7;
        const int rb = wc * 64 + wfrag_row<LP_OUT>(i, l16);
        b_off[i] = A_BYTES + rb * ROWB; b_key[i] = swz_key<LP_OUT>(rb);
    }
    f32x4 res[1][1];  // the epilogue reads the residual itself (RES_LOAD)
    // Fragment reads run HALF A SLAB ahead of the MFMAs that use them: X = the kk=0 fragments of slab s are read right
    // after barrier s and land while the 16 MFMAs on Y = the kk=1 fragments of slab s-1 execute; Y(s) is read while
    // the MFMAs on X(s) execute.  (Holding Y(s-1) in registers across barrier s is fine: its ds_reads completed
    // before the barrier, which is what frees the stage for the loaders.)  The loop bodies are straight-line code -
    // the compiler's s_waitcnt insertion turns conservative at control-flow merges - so a tile's first slab is peeled.
    bf16x8 xa[4], xb[4], ya[4], yb[4];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cur_stage = 0;
#define MADTP_WS_READ(XA, XB, CH)                                                         \
    if (!(MADTP_WS_ABLATE & 1) || first)                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
        XA[i] = *(const bf16x8*)(st + a_off[i] + ((((CH) + grp4) ^ a_key[i]) << 4));      \
        XB[i] = *(const bf16x8*)(st + b_off[i] + ((((CH) + grp4) ^ b_key[i]) << 4));      \
    }
#define MADTP_WS_MFMA(XA, XB)                                                             \
    if (!(MADTP_WS_ABLATE & 4))                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
            acc[i][j] = mfma_16x16x32<MF16>(XB[j], XA[i], acc[i][j]);
#ifdef MADTP_WS_TIMING
    long long ws_t_main = 0, ws_t_epi = 0, ws_tiles = 0;
    const long long ws_t_begin = WS_NOW();
    long long ws_t0 = ws_t_begin;
#endif
    for (int unit = 0; unit < n_units; ++unit) {
      const bool tail = unit == n_full;  // the K-piece of a stream-K tail tile
      const int slot = tail ? sk_slot : lb + unit * gl;
      const int nku = tail ? sk_k1 - sk_k0 : nk;
      if constexpr (X3) {
        // f16-split: per k-slab two staged steps SA = [P0|Q1], SB = [P1|Q0] and six groups of 16 MFMAs,
        //   G1 P0a Q1a   G2 P0b Q1b   G3 P0a Q0a   G4 P1a Q0a'   G5 P0b Q0b   G6 P1b Q0b'      (a / b: the two 32-deep halves, ' : * 2^-11)
        // with every fragment read issued one group ahead of its first use, so at most four 4-fragment sets are live (as in the
        // bf16 loop).  barrier(SB) sits between G1 and G2, barrier(next SA) between G5 and G6: all reads of a stage are complete
        // (lgkmcnt(0)) before the barrier that lets the loaders overwrite it.
#define X3_READ_A(R, ST, CH) _Pragma("unroll") for (int i = 0; i < 4; ++i) R[i] = *(const bf16x8*)((ST) + a_off[i] + ((((CH) + grp4) ^ a_key[i]) << 4));
#define X3_READ_W(R, ST, CH) _Pragma("unroll") for (int i = 0; i < 4; ++i) R[i] = *(const bf16x8*)((ST) + b_off[i] + ((((CH) + grp4) ^ b_key[i]) << 4));
#define X3_MFMA(RA, RW)                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
            acc[i][j] = mfma_16x16x32<true>(RW[j], RA[i], acc[i][j]);
#define X3_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X3_LGKM0() do { X3_FENCE(); __builtin_amdgcn_s_waitcnt(0xC07F); X3_FENCE(); } while (0)
        bf16x8 p0a[4], p0b[4], q1a[4], q1b[4], q0[4], p1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = zero4;
        {
            const char* sA = smem + cur_stage * STAGE_BYTES;
            __builtin_amdgcn_s_barrier();  // SA of the tile's first k-slab landed
            X3_READ_A(p0a, sA, 0) X3_READ_W(q1a, sA, 0)
        }
        // (the last k-slab of a tile is peeled: a conditional re-read of p0a / q1a inside the loop would keep their old values
        //  live across the whole body - two more fragment sets than the register budget of a 12-wave workgroup holds)
#define X3_KSLAB(LAST)                                                                                                    \
        {                                                                                                                 \
            const char* sA = smem + cur_stage * STAGE_BYTES;                                                              \
            if (++cur_stage == STAGES) cur_stage = 0;                                                                     \
            const char* sB = smem + cur_stage * STAGE_BYTES;                                                              \
            if (++cur_stage == STAGES) cur_stage = 0;                                                                     \
            const char* sN = smem + cur_stage * STAGE_BYTES; /* SA of the next k-slab */                                  \
            X3_READ_A(p0b, sA, 4) X3_READ_W(q1b, sA, 4)                                                                   \
            X3_FENCE();                                                                                                   \
            X3_MFMA(p0a, q1a) /* G1 */                                                                                    \
            X3_LGKM0();                                                                                                   \
            __builtin_amdgcn_s_barrier(); /* SB landed; every wave is done reading SA */                                  \
            X3_READ_W(q0, sB, 0)                                                                                          \
            X3_FENCE();                                                                                                   \
            X3_MFMA(p0b, q1b) /* G2 */                                                                                    \
            X3_FENCE();                                                                                                   \
            X3_READ_A(p1, sB, 0)                                                                                          \
            X3_FENCE();                                                                                                   \
            X3_MFMA(p0a, q0) /* G3 */                                                                                     \
            X3_FENCE();                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) q1a[i] = x3_scale_lo(q0[i]);                                    \
            X3_READ_W(q0, sB, 4)                                                                                          \
            X3_FENCE();                                                                                                   \
            X3_MFMA(p1, q1a) /* G4 */                                                                                     \
            X3_FENCE();                                                                                                   \
            X3_READ_A(p1, sB, 4)                                                                                          \
            X3_FENCE();                                                                                                   \
            X3_MFMA(p0b, q0) /* G5 */                                                                                     \
            X3_FENCE();                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) q1b[i] = x3_scale_lo(q0[i]);                                    \
            X3_LGKM0();                                                                                                   \
            if constexpr (!(LAST)) {                                                                                      \
                __builtin_amdgcn_s_barrier(); /* next SA landed; every wave is done reading SB */                         \
                X3_READ_A(p0a, sN, 0) X3_READ_W(q1a, sN, 0)                                                               \
            }                                                                                                             \
            X3_FENCE();                                                                                                   \
            X3_MFMA(p1, q1b) /* G6 */                                                                                     \
            X3_FENCE();                                                                                                   \
        }
        for (int kt = 0; kt + 1 < nkp; ++kt) X3_KSLAB(false)
        X3_KSLAB(true)
#undef X3_KSLAB
#undef X3_READ_A
#undef X3_READ_W
#undef X3_MFMA
#undef X3_FENCE
#undef X3_LGKM0
      } else {
        {   // first slab of the tile: accumulators start from zero, no Y pending
            constexpr bool first = true;
            const char* st = smem + cur_stage * STAGE_BYTES;
            if (!(MADTP_WS_ABLATE & 2)) __builtin_amdgcn_s_barrier();  // slab landed (the loaders waited before arriving)
            MADTP_WS_READ(xa, xb, 0)
            MADTP_WS_READ(ya, yb, 4)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma_16x16x32<MF16>(xb[j], xa[i], zero4);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): Y is in registers, this wave is done with the stage
            __builtin_amdgcn_sched_barrier(0);
            if (++cur_stage == STAGES) cur_stage = 0;
        }
        for (int kt = 1; kt < nku; ++kt) {
            constexpr bool first = false;
            const char* st = smem + cur_stage * STAGE_BYTES;
            if (!(MADTP_WS_ABLATE & 2)) __builtin_amdgcn_s_barrier();
            MADTP_WS_READ(xa, xb, 0)
            __builtin_amdgcn_sched_barrier(0);
            MADTP_WS_MFMA(ya, yb)
            __builtin_amdgcn_sched_barrier(0);
            MADTP_WS_READ(ya, yb, 4)
            __builtin_amdgcn_sched_barrier(0);
            MADTP_WS_MFMA(xa, xb)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            if (++cur_stage == STAGES) cur_stage = 0;
        }
        { MADTP_WS_MFMA(ya, yb) }
        if (tail) {
            // park this wave's 64x64 partial ([fragment][lane] f32x4: every store is one contiguous KiB), take the ticket of
            // (tail tile, wave position); the last of the sk_p arrivals sums all pieces in piece order and goes on to the epilogue
            f32x4* ws4 = (f32x4*)g.sk_ws;  // (uniform base + 32-bit element offsets: nothing 64-bit per lane stays live)
            constexpr unsigned WAVE_STRIDE = 16 * 64, UNIT_STRIDE = NCW * WAVE_STRIDE;
            unsigned in_unit = (unsigned)wave * WAVE_STRIDE + (unsigned)lane;
            asm volatile("" : "+v"(in_unit));  // the addresses are built HERE: hoisted out of the tile loop they would be spilled
            {
                const unsigned mine = (unsigned)(xcd * 32 + lb) * UNIT_STRIDE + in_unit;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ws4[mine + (i * 4 + j) * 64] = acc[i][j];
            }
            // All pieces of a tile run on ONE XCD (same L2), so the hand-over needs no agent-scope fence - __threadfence() would
            // write back and invalidate the whole L2 under the other workgroups' feet (measured: +60-90 us per launch).  It is the
            // protocol the compiler emits for workgroup scope in threadgroup-split mode: the stores are complete in L2 once vmcnt
            // reaches 0 (the vector L1 is write-through), the ticket is an L2 atomic, and the reader drops its CU's L1.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int* tk = g.sk_tick + (xcd * 16 + sk_tile) * NCW + wave;
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != sk_p - 1) continue;
            asm volatile("buffer_inv sc0" ::: "memory");
            if (lane == 0) *tk = 0;  // clean for the next launch on this workspace
            unsigned piece = (unsigned)(xcd * 32 + sk_tile * sk_p) * UNIT_STRIDE + in_unit;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = ws4[piece + (i * 4 + j) * 64];
            for (int p = 1; p < sk_p; ++p) {
                piece += UNIT_STRIDE;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += ws4[piece + (i * 4 + j) * 64];
            }
        }
      }
        int t = t0 + slot;
        const bool second = g.pair && t >= tiles1;
        if (second) t -= tiles1;
        int ctm, ctn;
        tile_mn(g, t, ctm, ctn);
        const int m0 = ctm * BM + grp * 128, n0 = ctn * BN;
#ifdef MADTP_WS_TIMING
        { const long long now = WS_NOW(); ws_t_main += now - ws_t0; ws_t0 = now; }
#endif
        if (!((g.dbg & 1) && acc[0][0][0] != 12345.678f)) {
            GemmArgs ge = g;  // (kernel arguments live in SGPRs: this is two scalar selects)
            if (second) { ge.bias = g.bias2; ge.C = g.C2; ge.acc_scale = g.acc_scale2; }
#define EPI(ACT)                                                                                              \
    if constexpr (LP_OUT) {                                                                                   \
        epilogue<OM, ACT, false, 4, 4, 128, 128>(ge, acc, res, m0, n0, wr, wc, l16, grp4, 0);                  \
    } else {                                                                                                  \
        if (g.residual) epilogue<OM, ACT, true, 4, 4, 128, 128>(ge, acc, res, m0, n0, wr, wc, l16, grp4, 0);   \
        else epilogue<OM, ACT, false, 4, 4, 128, 128>(ge, acc, res, m0, n0, wr, wc, l16, grp4, 0);             \
    }
            switch (g.act) {
                case MADTP_ACT_GELU_ERF: EPI(MADTP_ACT_GELU_ERF) break;
                case MADTP_ACT_QUICK_GELU: EPI(MADTP_ACT_QUICK_GELU) break;
                case MADTP_ACT_RELU: EPI(MADTP_ACT_RELU) break;
                default: EPI(MADTP_ACT_NONE) break;
            }
#undef EPI
        }
#ifdef MADTP_WS_TIMING
        { const long long now = WS_NOW(); ws_t_epi += now - ws_t0; ws_t0 = now; ++ws_tiles; }
#endif
    }
#ifdef MADTP_WS_TIMING
    if (blockIdx.x == 0 && tid == 0) {
        g_ws_dbg[0] = ws_t_main; g_ws_dbg[1] = ws_t_epi; g_ws_dbg[2] = ws_tiles; g_ws_dbg[3] = ws_t0 - ws_t_begin;
    }
#endif
#undef MADTP_WS_READ
#undef MADTP_WS_MFMA
}

// ------------------------------------------------------------------------------------------------------------------
// 256x256 bf16 kernel ("sq"): one 512-thread workgroup per CU = 8 waves as 2 (M) x 4 (N), 128x64 outputs per wave
// (8 x 4 MFMA 16x16x32 fragments = 128 f32 accumulators per lane), K walked in 64-deep slabs of 256 A rows + 256 W rows
// (64 KiB) through a TWO-stage LDS ring.  Why next to the 256x128 wave-specialised kernel:
//   * L2 -> LDS bytes per flop drop by a quarter (the 256x128 kernel's LDS-DMA stream alone needs as long as its MFMA
//     stream: ~23 TB/s of L2 bandwidth chip-wide), LDS fragment reads per flop by a quarter as well (wave tile 128x64
//     instead of 64x64);
//   * the ViT shapes of the forward (M = 10.5k-25k rows, N = 768 / 2304 / 3072) quantise far better: N = 768 is 3 column
//     tiles, so everything up to 21.7k rows is ONE round on 256 CUs where the 256x128 tiling needs two.
// Every wave both loads and computes: per slab each wave issues 8 LDS-DMA instructions (buffer_load_dwordx4 ... lds: the
// per-lane row offset in a VGPR, the K offset in an SGPR, rows past M dropped by the descriptor) for slab s+1 right after
// barrier s, then runs 4 sub-phases of 16 MFMAs (k-step x 64-row half) whose fragment reads are issued one sub-phase
// ahead; the last sub-phase of slab s executes after barrier s+1 from registers, under the first fragment reads of slab
// s+1, so no ds_read latency is exposed at the barrier.  One s_barrier per slab; the only vmcnt wait sits at the END of a
// slab (the DMA of the next slab was issued a whole slab earlier; a tile's output stores are a whole slab old as well).
// MADTP_SQ_ABLATE (timing experiments only, results are wrong): bit 0 drops the steady-state LDS-DMA, bit 1 the MFMAs, bit 2 the
// steady-state fragment reads.
#ifndef MADTP_SQ_ABLATE
#define MADTP_SQ_ABLATE 0
#endif
template <int OM>
__global__ __launch_bounds__(512, 2) void gemm_sq_kernel(GemmArgs g) {
    constexpr bool LP_OUT = OM != OM_F32;
    constexpr int BM = 256, BN = 256, A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, gl = gridDim.x >> 3;
    int t0, nslots;
    xcd_tiles(g.ntm * g.ntn, xcd, t0, nslots);
    if (lb >= nslots) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l16 = lane & 15, grp4 = lane >> 4;
    const int nk = g.K / 64;
    const int my_slots = (nslots - lb + gl - 1) / gl;
    const int total_slabs = my_slots * nk;  // (32-bit on purpose: a 64-bit counter is compared on the VALU and spilled)

    // ---- fragment addresses: one base per operand, the fragment index is an immediate offset (the swizzle key of a lane's
    //      rows does not depend on the fragment: A rows 128 wr + 16 i + l16 have key l16 & 7, W rows see wfrag_row / swz_key)
    const int key_a = l16 & 7;
    const int rw0 = wc * 64 + wfrag_row<LP_OUT>(0, l16);
    const int key_w = swz_key<LP_OUT>(rw0);
    const int a_rd = (wr * 128 + l16) * ROWB + ((grp4 ^ key_a) << 4);            // k-step 0; k-step 1 = ^ 64
    const int w_rd = A_BYTES + rw0 * ROWB + ((grp4 ^ key_w) << 4);
    constexpr int WJ1 = (wfrag_row<LP_OUT>(1, 0) - wfrag_row<LP_OUT>(0, 0)) * ROWB;
    constexpr int WJ2 = (wfrag_row<LP_OUT>(2, 0) - wfrag_row<LP_OUT>(0, 0)) * ROWB;
    constexpr int WJ3 = (wfrag_row<LP_OUT>(3, 0) - wfrag_row<LP_OUT>(0, 0)) * ROWB;

    // ---- LDS-DMA: stage image = A rows 0..255, then W rows 0..255 (128-byte rows, chunk ^= key(row)); instruction idx covers
    //      rows 8 (idx & 31) .. +7 of A (idx < 32) or W; wave w issues idx = 8 w .. 8 w + 7, i.e. waves 0-3 load A, 4-7 load W
    const bool ld_a = wave < 4;
    const int sub = lane >> 3, pos = lane & 7;
    const int npad = ((g.N + 127) / 128) * 128;  // W is padded to a multiple of 128 rows by the caller
    const __amdgpu_buffer_rsrc_t rsrc = ld_a
        ? __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)((size_t)g.M * g.lda * 2), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)((size_t)npad * g.ldw * 2), 0x00020000);
    unsigned roff[8];  // per-lane source byte offset of each DMA instruction at k = 0 of the tile being staged
    int is_slot = lb, is_kt = 0, is_stage = 0;
    int issued = 0;
    auto tile_offsets = [&]() {
        int itm, itn;
        tile_mn(g, t0 + is_slot, itm, itn);
        const int r0 = ld_a ? itm * BM : itn * BN;
        const unsigned ld2 = (unsigned)(ld_a ? g.lda : g.ldw) * 2u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = ((wave & 3) * 8 + q) * 8 + sub;                          // row within the A / W tile
            const int key = ld_a ? (r & 7) : swz_key<LP_OUT>(r);
            roff[q] = (unsigned)(r0 + r) * ld2 + (unsigned)((pos ^ key) << 4);     // rows past the matrix fall outside the descriptor
        }
    };
    tile_offsets();
    auto issue_next = [&]() {
        if (issued < total_slabs && !((MADTP_SQ_ABLATE & 1) && issued > 0)) {
            char* st = smem + is_stage * STAGE_BYTES + wave * 8 * 1024;
            const int koff = is_kt * ROWB;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(st + q * 1024), 16, roff[q], koff, 0, 0);
        }
        ++issued;
        is_stage ^= 1;
        if (++is_kt == nk) {
            is_kt = 0;
            is_slot += gl;
            if (issued < total_slabs) tile_offsets();
        }
    };

    f32x4 acc[8][4];
    bf16x8 alo0[4], ahi0[4], alo1[4], ahi1[4], b0[4], b1[4];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 res[1][1];
#define SQ_FENCE() __builtin_amdgcn_sched_barrier(0)
#define SQ_READ_A(R, KX, HALF)                                                              \
    if (!(MADTP_SQ_ABLATE & 4) || slot == lb)                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
        R[i] = *(const bf16x8*)(st + ((a_rd + ((HALF) * 4 + i) * 16 * ROWB) ^ (KX)));
#define SQ_READ_W(R, KX)                                                                    \
    if (!(MADTP_SQ_ABLATE & 4) || slot == lb) {                                             \
    R[0] = *(const bf16x8*)(st + ((w_rd) ^ (KX)));                                          \
    R[1] = *(const bf16x8*)(st + ((w_rd + WJ1) ^ (KX)));                                    \
    R[2] = *(const bf16x8*)(st + ((w_rd + WJ2) ^ (KX)));                                    \
    R[3] = *(const bf16x8*)(st + ((w_rd + WJ3) ^ (KX))); }
#define SQ_MFMA(RA, RW, HALF)                                                               \
    if (!(MADTP_SQ_ABLATE & 2))                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
            acc[(HALF) * 4 + i][j] = mfma_16x16x32<false>(RW[j], RA[i], acc[(HALF) * 4 + i][j]);

    issue_next();  // slab 0 -> stage 0
    wait_vmcnt<0>();
    int cur_stage = 0;
    for (int slot = lb; slot < nslots; slot += gl) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = zero4;
        // (the first slab of a tile is peeled: it has no pending sub-phase, and a conditional inside the loop body would be a
        //  control-flow merge where the compiler's s_waitcnt insertion turns conservative)
#define SQ_SLAB(FIRST)                                                                                                \
        {                                                                                                             \
            const char* st = smem + cur_stage * STAGE_BYTES;                                                          \
            cur_stage ^= 1;                                                                                           \
            __builtin_amdgcn_s_barrier(); /* slab landed for every wave; every wave finished reading the other stage */ \
            issue_next();                                                                                             \
            SQ_FENCE();                                                                                               \
            SQ_READ_A(alo0, 0, 0) SQ_READ_W(b0, 0)                                                                    \
            SQ_FENCE();                                                                                               \
            if (!(FIRST)) { SQ_MFMA(ahi1, b1, 1) } /* last sub-phase of the previous slab, from registers */          \
            SQ_FENCE();                                                                                               \
            SQ_READ_A(ahi0, 0, 1)                                                                                     \
            SQ_FENCE();                                                                                               \
            SQ_MFMA(alo0, b0, 0)                                                                                      \
            SQ_FENCE();                                                                                               \
            SQ_READ_A(alo1, 64, 0) SQ_READ_W(b1, 64)                                                                  \
            SQ_FENCE();                                                                                               \
            SQ_MFMA(ahi0, b0, 1)                                                                                      \
            SQ_FENCE();                                                                                               \
            SQ_READ_A(ahi1, 64, 1)                                                                                    \
            SQ_FENCE();                                                                                               \
            SQ_MFMA(alo1, b1, 0)                                                                                      \
            SQ_FENCE();                                                                                               \
            __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): this wave is done reading the stage */                 \
            wait_vmcnt<0>(); /* own DMA of the next slab landed (issued a slab ago); a tile's stores are a slab old */ \
            SQ_FENCE();                                                                                               \
        }
        SQ_SLAB(true)
        for (int kt = 1; kt < nk; ++kt) SQ_SLAB(false)
#undef SQ_SLAB
        { SQ_MFMA(ahi1, b1, 1) }
        int ctm, ctn;
        tile_mn(g, t0 + slot, ctm, ctn);
        const int m0 = ctm * BM, n0 = ctn * BN + (wc >> 1) * 128;
#define EPI(ACT)                                                                                                  \
    if constexpr (LP_OUT) {                                                                                       \
        epilogue<OM, ACT, false, 8, 4, 256, 128>(g, acc, res, m0, n0, wr, wc & 1, l16, grp4, 0);                   \
    } else {                                                                                                      \
        if (g.residual) epilogue<OM, ACT, true, 8, 4, 256, 128>(g, acc, res, m0, n0, wr, wc & 1, l16, grp4, 0);    \
        else epilogue<OM, ACT, false, 8, 4, 256, 128>(g, acc, res, m0, n0, wr, wc & 1, l16, grp4, 0);              \
    }
        switch (g.act) {
            case MADTP_ACT_GELU_ERF: EPI(MADTP_ACT_GELU_ERF) break;
            case MADTP_ACT_QUICK_GELU: EPI(MADTP_ACT_QUICK_GELU) break;
            case MADTP_ACT_RELU: EPI(MADTP_ACT_RELU) break;
            default: EPI(MADTP_ACT_NONE) break;
        }
#undef EPI
    }
#undef SQ_FENCE
#undef SQ_READ_A
#undef SQ_READ_W
#undef SQ_MFMA
}


#ifdef MADTP_WS_TIMING
extern "C" int madtp_debug_read_ws_ts(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_dbg), sizeof(long long) * 8);
}
#endif

// gemm_pp.hip: the ping-pong 256x256 kernel lives in its own translation unit (args = const GemmArgs*)
__attribute__((visibility("hidden"))) int madtp_gemm_pp_launch(const void* args, int om, int mode, int rows, int grid, void* stream);
#include "gemm_table.h"  // per-shape kernel choice of the big problems, measured (tools/gemm_autotune.py)

// ---- optional HIP-event profiling of every GEMM launch (bench.py's roofline leg) -----------------------------------
namespace {
struct GemmRecord { hipEvent_t e0, e1; double flops, bytes; int dt, M, N, K; };
bool g_prof_on = false;
std::vector<GemmRecord> g_prof;
}  // namespace

extern "C" int madtp_profile_begin(void) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = true;
    return 0;
}

// Stops recording, waits for the recorded events and writes one text line per (dtype, M, N, K):
// "dtype M N K launches total_ms flops algorithmic_bytes".  Returns the number of bytes written (0 if nothing was recorded).
extern "C" int madtp_profile_end(char* buf, int cap) {
    g_prof_on = false;
    std::map<std::tuple<int, int, int, int>, std::tuple<int, double, double, double>> agg;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto& a = agg[std::make_tuple(r.dt, r.M, r.N, r.K)];
            std::get<0>(a) += 1; std::get<1>(a) += ms; std::get<2>(a) += r.flops; std::get<3>(a) += r.bytes;
        }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    int off = 0;
    for (auto& kv : agg) {
        const int n = snprintf(buf + off, cap > off ? cap - off : 0, "%d %d %d %d %d %.6f %.0f %.0f\n", std::get<0>(kv.first),
                               std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), std::get<0>(kv.second),
                               std::get<1>(kv.second), std::get<2>(kv.second), std::get<3>(kv.second));
        if (n < 0 || off + n >= cap) break;
        off += n;
    }
    return off;
}

// second problem of a madtp_gemm_pair launch (same shape, leading dimensions and dtypes as the first)
struct GemmPair { const void* A; const void* W; const float* bias; void* C; float acc_scale; };
constexpr int PAIR_UNSUPPORTED = 1000;  // internal: this shape does not run on the wave-specialised kernel

static int gemm_launch(const void* A, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K,
                       int lda, int ldw, int ldc, int ldr, int ab_dtype, int c_dtype, int act, float acc_scale, float out_scale,
                       int splitk, void* stream, const GemmPair* pair = nullptr, DevN m_dev = DevN{nullptr, 0, 0});

// Workgroups per XCD of the two big-GEMM kernels (default 32 = one persistent workgroup per CU walking its share of the tiles).
// A larger cap gives every workgroup fewer tiles (>= tiles / 8: one tile each) - the launch then frees CUs tile by tile, which
// lets the small kernels of ANOTHER stream in between (madtp_amd/pipeline.py) at the price of the cross-tile pipelining.
static std::atomic<int> g_wg_per_xcd{-1};
static thread_local int t_wg_cap = 0;  // per-thread override for the launches of one library call (madtp_internal_gemm_wg_cap)
int madtp_internal_gemm_wg_cap(int cap) { const int prev = t_wg_cap; t_wg_cap = cap > 0 ? cap : 0; return prev; }

// Per-stream scheduling attributes (madtp_stream_set_sched, include/madtp_hip.h): what slice of the chip a stream owns (a CU-masked
// stream of a caller that partitions the GPU between forwards in flight) and that caller's dispatch hints.  Readers are the launch
// paths (lock-free scan of a small table: a slot's key is published last), writers take the mutex.
struct StreamSched { int cus_per_xcd; float sq_cost; int small_tile; };
constexpr int SCHED_SLOTS = 64;
static std::atomic<void*> g_sched_key[SCHED_SLOTS];
static std::atomic<int> g_sched_cus[SCHED_SLOTS];
static std::atomic<float> g_sched_cost[SCHED_SLOTS];
static std::atomic<int> g_sched_small[SCHED_SLOTS];
static std::atomic<int> g_sched_used{0};  // number of slots ever handed out (the readers' scan bound; 0 = nobody uses the table)
static std::mutex g_sched_mu;
static StreamSched stream_sched(void* stream) {
    StreamSched r{32, -1.f, -2};
    const int n = g_sched_used.load(std::memory_order_acquire);
    for (int i = 0; i < n; i++)
        if (g_sched_key[i].load(std::memory_order_acquire) == stream && stream) {
            r.cus_per_xcd = g_sched_cus[i].load(std::memory_order_relaxed);
            r.sq_cost = g_sched_cost[i].load(std::memory_order_relaxed);
            r.small_tile = g_sched_small[i].load(std::memory_order_relaxed);
            break;
        }
    return r;
}
extern "C" int madtp_stream_set_sched(void* stream, int cus_per_xcd, float sq_cost, int small_tile) {
    if (!stream || cus_per_xcd < 0 || cus_per_xcd > 32 || small_tile < -2 || small_tile > 3) return MADTP_E_BADARG;
    std::lock_guard<std::mutex> lk(g_sched_mu);
    const int n = g_sched_used.load(std::memory_order_relaxed);
    int slot = -1;
    for (int i = 0; i < n; i++) if (g_sched_key[i].load(std::memory_order_relaxed) == stream) { slot = i; break; }
    if (slot < 0)
        for (int i = 0; i < n; i++) if (g_sched_key[i].load(std::memory_order_relaxed) == nullptr) { slot = i; break; }
    const bool fresh = slot < 0 || g_sched_key[slot].load(std::memory_order_relaxed) != stream;
    if (slot < 0) {
        if (n >= SCHED_SLOTS) return MADTP_E_BADARG;
        slot = n;
    }
    if (fresh) { g_sched_cus[slot].store(32, std::memory_order_relaxed); g_sched_cost[slot].store(-1.f, std::memory_order_relaxed); g_sched_small[slot].store(-2, std::memory_order_relaxed); }
    if (cus_per_xcd > 0) g_sched_cus[slot].store(cus_per_xcd, std::memory_order_relaxed);
    g_sched_cost[slot].store(sq_cost > 0.f ? sq_cost : -1.f, std::memory_order_relaxed);
    g_sched_small[slot].store(small_tile, std::memory_order_relaxed);
    g_sched_key[slot].store(stream, std::memory_order_release);
    if (slot >= n) g_sched_used.store(slot + 1, std::memory_order_release);
    return 0;
}
extern "C" int madtp_stream_get_sched(void* stream, int* cus_per_xcd, float* sq_cost, int* small_tile) {
    const StreamSched r = stream_sched(stream);
    if (cus_per_xcd) *cus_per_xcd = r.cus_per_xcd;
    if (sq_cost) *sq_cost = r.sq_cost;
    if (small_tile) *small_tile = r.small_tile;
    return 0;
}
static void stream_sched_forget(void* stream) {
    std::lock_guard<std::mutex> lk(g_sched_mu);
    const int n = g_sched_used.load(std::memory_order_relaxed);
    for (int i = 0; i < n; i++) if (g_sched_key[i].load(std::memory_order_relaxed) == stream) g_sched_key[i].store(nullptr, std::memory_order_release);
}
extern "C" int madtp_stream_create_cumask(void** stream_out, const uint32_t* mask, int words) {
    if (!stream_out || !mask || words < 1 || words > 32) return MADTP_E_BADARG;
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    *stream_out = (void*)s;
    return 0;
}
extern "C" int madtp_stream_destroy(void* stream) {
    if (!stream) return MADTP_E_BADARG;
    stream_sched_forget(stream);
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return 0;
}

// workgroups per XCD of a persistent big-GEMM launch on `stream`: the process / thread setting scaled to the CUs the stream owns
static int gemm_wg_per_xcd(const StreamSched& ss) {
    int v = t_wg_cap;
    if (v <= 0) {
        v = g_wg_per_xcd.load(std::memory_order_relaxed);
        if (v < 0) {
            const char* e = getenv("MADTP_GEMM_WG_PER_XCD");
            v = e ? atoi(e) : 32;
            if (v < 1) v = 32;
            g_wg_per_xcd.store(v, std::memory_order_relaxed);
        }
    }
    if (ss.cus_per_xcd < 32) { v = v * ss.cus_per_xcd / 32; if (v < 1) v = 1; }
    return v;
}

// forced tile configuration (madtp_gemm_set_config / MADTP_GEMM_CFG): A/B measurements and the per-kernel tests
static std::atomic<int> g_force_cfg{-1};
static int gemm_force_cfg() {
    int v = g_force_cfg.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MADTP_GEMM_CFG");
        v = e ? atoi(e) : 0;
        if (v < 0 || v > 10) v = 0;
        g_force_cfg.store(v, std::memory_order_relaxed);
    }
    return v;
}
// Per-round cost of a 256x256 tile relative to a 256x128 tile in the dispatch rule below.  1.7 is what an isolated launch
// measures (the rule then counts rounds).  With several forwards in flight on one GPU the tail of a sparse last round is filled
// by the other streams' kernels, so the round count matters less than the per-flop efficiency of the tile (the 256x256 tile
// reads half the LDS bytes per MFMA): madtp_amd/pipeline.py lowers the cost to 0.9 while its workers run - measured NLVR
// 25.2 -> 25.7 k images/s with four in flight, but 20.1 -> 19.2 k on the serial loop, which keeps 1.7.
static std::atomic<float> g_sq_cost{-1.f};
static std::atomic<int> g_sq_cost_hinted{0};  // a caller's hint is in force (madtp_gemm_set_sq_cost): the measured table steps aside
static float gemm_sq_cost() {
    float v = g_sq_cost.load(std::memory_order_relaxed);
    if (v <= 0.f) {
        const char* e = getenv("MADTP_GEMM_SQ_COST");
        v = e ? (float)atof(e) : 1.7f;
        if (!(v > 0.f)) v = 1.7f;
        g_sq_cost.store(v, std::memory_order_relaxed);
    }
    return v;
}
// Tile configuration of the SMALL problems (the 1280-row GEMMs of the text encoders), -1 = automatic: the automatic rule takes
// the smallest tile that still fits one round, which is what a lone latency-bound launch wants; with several forwards in flight
// the CU time of a launch counts, not its latency, and the 128x128 tile (a quarter of the operand re-reads of 64x64) wins:
// NLVR 25.5 -> 26.2 k images/s, retrieval 28.5 -> 30.4 k with four in flight, -3.5 % on the serial loop (which keeps -1).
static std::atomic<int> g_small_tile{-2};
static int gemm_small_tile() {
    int v = g_small_tile.load(std::memory_order_relaxed);
    if (v == -2) {
        const char* e = getenv("MADTP_GEMM_SMALL_CFG");
        v = e ? atoi(e) : -1;
        if (v < -1 || v > 3) v = -1;
        g_small_tile.store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" int madtp_gemm_set_small_tile(int cfg) {
    const int prev = gemm_small_tile();
    g_small_tile.store((cfg < 0 || cfg > 3) ? -1 : cfg, std::memory_order_relaxed);
    return prev;
}
extern "C" float madtp_gemm_set_sq_cost(float cost) {
    const float prev = gemm_sq_cost();
    g_sq_cost.store(cost > 0.f ? cost : -1.f, std::memory_order_relaxed);
    g_sq_cost_hinted.store(cost > 0.f && cost != 1.7f ? 1 : 0, std::memory_order_relaxed);
    return prev;
}
extern "C" int madtp_gemm_set_config(int cfg) {
    const int prev = gemm_force_cfg();
    g_force_cfg.store((cfg < 0 || cfg > 10) ? 0 : cfg, std::memory_order_relaxed);
    return prev;
}

// Two independent GEMMs of identical shape (C_i = A_i @ W_i^T + bias_i) in ONE launch of the wave-specialised kernel: the
// twin cross-attention branches of the NLVR text layers project their image tokens to [k|v] with two 240-tile problems, each
// a single round on 256 CUs - as one 480-tile launch they share the launch, ramp-up and drain (~6 us of a 17.8 us kernel).
// Shapes the wave-specialised kernel does not take run as two madtp_gemm launches.
extern "C" int madtp_gemm_pair(const void* A0, const void* A1, const void* W0, const void* W1, const float* bias0,
                               const float* bias1, void* C0, void* C1, int M, int N, int K, int lda, int ldw, int ldc,
                               int ab_dtype, int c_dtype, float acc_scale0, float acc_scale1, void* stream) {
    if (!A1 || !W1 || !C1 || (!bias0) != (!bias1)) return MADTP_E_BADARG;
    const GemmPair p{A1, W1, bias1, C1, acc_scale1};
    int rc = gemm_launch(A0, W0, bias0, nullptr, C0, M, N, K, lda, ldw, ldc, 0, ab_dtype, c_dtype, MADTP_ACT_NONE, acc_scale0, 1.f, 1,
                         stream, &p);
    if (rc != PAIR_UNSUPPORTED) return rc;
    rc = gemm_launch(A0, W0, bias0, nullptr, C0, M, N, K, lda, ldw, ldc, 0, ab_dtype, c_dtype, MADTP_ACT_NONE, acc_scale0, 1.f, 1, stream);
    if (rc) return rc;
    return gemm_launch(A1, W1, bias1, nullptr, C1, M, N, K, lda, ldw, ldc, 0, ab_dtype, c_dtype, MADTP_ACT_NONE, acc_scale1, 1.f, 1,
                       stream);
}

extern "C" int madtp_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C,
                          int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                          int ab_dtype, int c_dtype, int act, float acc_scale, float out_scale, void* stream) {
    return gemm_launch(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, ab_dtype, c_dtype, act, acc_scale, out_scale, 1, stream);
}

// Split-K variant for small-M problems: part[s, M, N] (f32, contiguous) = A[:, Ks] @ W[:, Ks]^T for K range s of
// `splits`; no bias/activation/residual (madtp_splitk_ln applies them after the fixed-order reduction).
extern "C" int madtp_gemm_splitk(const void* A, const void* W, float* part, int M, int N, int K, int lda, int ldw, int splits,
                                 int ab_dtype, void* stream) {
    if (splits < 1) return MADTP_E_BADARG;
    const int esz = ab_dtype == MADTP_F32 ? 4 : 2;
    if ((K * esz / ROWB) % splits) return MADTP_E_SHAPE;  // (f16-split: every K range holds whole [P0|Q1],[P1|Q0] step pairs)
    return gemm_launch(A, W, nullptr, nullptr, part, M, N, K, lda, ldw, N, 0, ab_dtype, MADTP_F32, MADTP_ACT_NONE, 1.f, 1.f, splits,
                       stream);
}

// Split-K partials of a LONG-K product on the 256x256 ping-pong kernel (round 5; f16-split operands only): part[s, M, N] (f32) =
// A[:, Ks] W[:, Ks]^T with the accumulator scale applied.  The backward's weight gradient dW = dY^T X is this shape - a few
// hundred to a few thousand output rows and columns, K = every token row of the batch (10-25 k) - and ran on 64x64 tiles at about
// half the rate: 9-36 tiles of 256x256 times `splits` K ranges fill the chip instead.  K % (128 splits) == 0, N % 8 == 0;
// madtp_splitk_sum reduces the slabs in order.
extern "C" int madtp_gemm_splitk_pp(const void* A, const void* W, float* part, int M, int N, int K, int lda, int ldw, int splits,
                                    float acc_scale, void* stream) {
    if (!A || !W || !part || M <= 0 || N <= 0 || K <= 0 || splits < 1) return MADTP_E_BADARG;
    if (K % (128 * splits) || N % 8 || lda < 2 * K || ldw < 2 * K) return MADTP_E_SHAPE;
    if (!aligned16(A) || !aligned16(W) || !aligned16(part) || (lda * 2) % 16 || (ldw * 2) % 16) return MADTP_E_ALIGN;
    if (((size_t)M + 255) * (size_t)lda * 2 >= ((size_t)1 << 32) || ((size_t)N + 255) * (size_t)ldw * 2 >= ((size_t)1 << 32) ||
        (size_t)N * 256 * 4 >= ((size_t)1 << 31)) return MADTP_E_SHAPE;
    GemmArgs g{};
    g.A = (const char*)A; g.W = (const char*)W; g.C = part;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = N; g.act = MADTP_ACT_NONE;
    g.out_scale = 1.f; g.acc_scale = g.acc_scale2 = acc_scale;
    g.splitk = splits; g.fast_epi = 1;
    g.ntm = (M + 255) / 256; g.ntn = (N + 255) / 256;
    const int slots_max = (g.ntm * g.ntn * splits + 7) / 8, cap = gemm_wg_per_xcd(stream_sched(stream));
    const int grid = 8 * (slots_max < cap ? slots_max : cap);
    const int rc = madtp_gemm_pp_launch(&g, OM_F32, 2, 256, grid, stream);
    if (rc) return rc;
    MADTP_LAUNCH_CHECK();
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ part, int splits, size_t count4, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    float4 a = ((const float4*)part)[i];
    for (int s = 1; s < splits; ++s) {
        const float4 b = ((const float4*)part)[(size_t)s * count4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    ((float4*)out)[i] = a;
}
}  // namespace
// out[i] = part[0][i] + part[1][i] + ... (in this order), count % 4 == 0
extern "C" int madtp_splitk_sum(const float* part, int splits, size_t count, float* out, void* stream) {
    if (!part || !out || splits < 1 || count == 0) return MADTP_E_BADARG;
    if (count % 4) return MADTP_E_SHAPE;
    if (!aligned16(part) || !aligned16(out)) return MADTP_E_ALIGN;
    hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, splits, count / 4, out);
    MADTP_LAUNCH_CHECK();
    return 0;
}

// Stream-K workspace of the wave-specialised kernel: one per (device, stream) - launches on one stream are ordered, launches
// on different streams must not share partials - allocated on first use (32 MiB of partials + the tickets, zeroed once; every
// ticket resets itself).  OFF by default (MADTP_GEMM_SK=1 turns it on; madtp_gemm_set_config(5) always uses it): on isolated
// launches it wins 8-16 % on the K = 3072, N = 768 problems at 11-14 k rows, inside the forward - where that GEMM reads its
// 75 MB operand and the f32 residual from HBM rather than from the Infinity Cache - the same launches take the same ~92 us with
// and without it and the next GEMM loses ~3 us to the evicted lines (profiles/r02_gemm_sk_ab.txt).
struct SkWorkspace { float* ws; int* tick; };
static bool sk_enabled() {
    static int sk_env = -1;
    if (sk_env < 0) { const char* e = getenv("MADTP_GEMM_SK"); sk_env = e ? atoi(e) : 0; }
    return sk_env != 0;
}
static bool sk_workspace(hipStream_t s, SkWorkspace& out) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SkWorkspace> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    auto it = pool.find({dev, s});
    if (it == pool.end()) {
        constexpr size_t WS_BYTES = (size_t)8 * 32 * 8 * 16 * 64 * 16, TICK_BYTES = (size_t)8 * 16 * 8 * sizeof(int);
        char* base = nullptr;
        if (hipMalloc((void**)&base, WS_BYTES + TICK_BYTES) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipMemsetAsync(base + WS_BYTES, 0, TICK_BYTES, s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(base); return false; }
        it = pool.emplace(std::make_pair(dev, s), SkWorkspace{(float*)base, (int*)(base + WS_BYTES)}).first;
    }
    out = it->second;
    return true;
}
// Cost (in rounds of 256x128 tiles) of the wave-specialised kernel on t256 tiles.  The stream-K tail pays for K >= 2048 only
// (measured, tools/gemm_bench.py ab / profiles/r02_gemm_sk_ab.txt): parking and re-reading the partials costs ~16 us per
// launch, while the few tiles of a plain last round run ~25 % faster than in a full round (no contention), so with K = 768
// (12 slabs, ~13 us per lone tile) the split loses 4-10 us and with K = 3072 it wins 6-14 us (M = 11-14 k rows, N = 768).
constexpr int SK_MIN_SLABS = 32;
static float ws_cost(int t256, int nk, bool sk, int cpx = 32) {
    const int nsl = (t256 + 7) / 8, rounds = nsl / cpx, rem = nsl - rounds * cpx;
    if (rem == 0) return (float)rounds;
    const int parts = (sk && cpx == 32 && nk >= SK_MIN_SLABS) ? sk_parts(rem, 32, nk) : 0;
    return (float)rounds + (parts ? 0.65f : 1.0f);
}

int madtp_i_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K, int lda, int ldw,
                 int ldc, int ldr, int ab_dtype, int c_dtype, int act, float acc_scale, float out_scale, DevN m_dev, void* stream) {
    return gemm_launch(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, ab_dtype, c_dtype, act, acc_scale, out_scale, 1, stream, nullptr,
                       m_dev);
}

static int gemm_launch(const void* A, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K,
                       int lda, int ldw, int ldc, int ldr, int ab_dtype, int c_dtype, int act, float acc_scale, float out_scale,
                       int splitk, void* stream, const GemmPair* pair, DevN m_dev) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return MADTP_E_BADARG;
    if (ab_dtype != MADTP_F32 && ab_dtype != MADTP_BF16 && ab_dtype != MADTP_F16S && ab_dtype != MADTP_F16) return MADTP_E_DTYPE;
    if (c_dtype != MADTP_F32 && c_dtype != MADTP_BF16 && c_dtype != MADTP_F16S && c_dtype != MADTP_F16) return MADTP_E_DTYPE;
    const bool x3 = ab_dtype == MADTP_F16S;
    const bool f16 = ab_dtype == MADTP_F16;  // plain f16 operands: the bf16 kernels' instantiations on the f16 MFMA
    if (c_dtype == MADTP_F16S && !x3) return MADTP_E_DTYPE;  // the split epilogue exists on the f16-split kernels only
    if (c_dtype == MADTP_BF16 && (x3 || f16)) return MADTP_E_DTYPE;  // a 2-byte output of 2-byte operands has their element format
    if (c_dtype == MADTP_F16 && !f16) return MADTP_E_DTYPE;
    const int esz = ab_dtype == MADTP_F32 ? 4 : 2;
    if ((K * esz) % ROWB != 0) return MADTP_E_SHAPE;
    if (!aligned16(A) || !aligned16(W) || (lda * esz) % 16 || (ldw * esz) % 16) return MADTP_E_ALIGN;
    // f16-split operands: leading dimensions count f16 elements (2 planes of K per activation row and per weight row)
    if (lda < (x3 ? 2 : 1) * K || ldw < (x3 ? 2 : 1) * K || ldc < (c_dtype == MADTP_F16S ? 2 : 1) * N || (residual && ldr < N))
        return MADTP_E_SHAPE;
    GemmArgs g;
    g.A = (const char*)A; g.W = (const char*)W; g.bias = bias; g.residual = residual; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.act = act; g.out_scale = out_scale;
    g.acc_scale = acc_scale; g.acc_scale2 = pair ? pair->acc_scale : acc_scale;
    g.ldc = (c_dtype == MADTP_BF16 || c_dtype == MADTP_F16) ? -ldc : ldc;  // negative: a 2-byte output (scalar fallback epilogue)
    g.range_flag = (c_dtype == MADTP_F16S || c_dtype == MADTP_F16) ? madtp_internal_range_flag() : nullptr;
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("MADTP_GEMM_DEBUG"); dbg = e ? atoi(e) : 0; }
    const int force_cfg = gemm_force_cfg();
    g.dbg = dbg;
    g.splitk = splitk;
    g.ngrp = 0;
    g.pair = 0; g.A2 = g.W2 = nullptr; g.bias2 = nullptr; g.C2 = nullptr;
    g.sk = 0; g.sk_ws = nullptr; g.sk_tick = nullptr;
    g.m_dev = m_dev.p; g.m_mul = m_dev.mul;
    if (m_dev.p && (M >= 4096 || pair || splitk != 1)) return MADTP_E_SHAPE;  // device-side M: the small-tile kernels only
    {
        static int desc_env = -1;  // MADTP_GEMM_DESC=0: gemm_kernel builds its LDS-DMA addresses per instruction (A/B runs)
        if (desc_env < 0) { const char* e = getenv("MADTP_GEMM_DESC"); desc_env = e ? atoi(e) : 1; }
        const size_t a_bytes = ((size_t)M + 127) * (size_t)lda * esz, w_bytes = ((size_t)N + 255) * (size_t)ldw * esz;
        g.desc = desc_env && a_bytes < ((size_t)1 << 31) && w_bytes < ((size_t)1 << 31);
    }
    // vector epilogue needs 16-byte aligned rows on every epilogue operand
    // (and, for the descriptor-bounded stores, a 256-row block of C below 2 GiB; bf16 output with an f32 residual has no
    // caller on the path and takes the scalar epilogue)
    g.fast_epi = (N % 8 == 0) && (ldc % 8 == 0) && aligned16(C) && (!bias || aligned16(bias)) &&
                 (!residual || (aligned16(residual) && ldr % 4 == 0 && c_dtype != MADTP_BF16 && c_dtype != MADTP_F16)) &&
                 (size_t)ldc * 256 * 4 < ((size_t)1 << 31);
    // tile configuration (MADTP_GEMM_CFG=1..4 forces one of the gemm_kernel variants for A/B measurements):
    //   0: 128x128, 2-stage ring, 2 workgroups/CU  - default, and the f32 path
    //   1: 64x128, 2 stages, 3 WG/CU   2: 64x128, 3 stages, 2 WG/CU   3: 64x64, 3 stages, 3 WG/CU
    // Small bf16 problems (the 1280-row GEMMs of the text encoder) are bound by the LDS-DMA rate of a CU (~40 GB/s with one
    // resident workgroup): what counts is spreading the operand bytes over ALL CUs in one round, so they take the
    // smallest tile whose grid still fits one round of 3 workgroups per CU (measured: 64x64 beats 128x128 by 20-45 % on
    // M=1280, N<=2304; 64x128 wins for N=3072).
    // The wave-specialised 256x128 kernel takes a problem once its tiles fill most of the chip (>= 200 of 256 CUs) - e.g. not
    // the 4480 x 768 GEMMs of a 128-pair re-ranking batch (108 tiles), which run better on 420 64x128 tiles.
    int cfg = 0;
    const StreamSched ss = stream_sched(stream);  // the CUs this stream owns (32 per XCD unless the caller said otherwise) + its hints
    const int cpx = ss.cus_per_xcd, ncu = 8 * cpx;
    const int t256 = ((M + 255) / 256) * ((N + 127) / 128);
    // Thresholds of the big-tile kernels (persistent 256-row tiles): M >= 4096 and most of the chip covered - what a lone launch
    // wants (latency).  MADTP_GEMM_BIG_MIN_M / _TILES lower them (experiments with several forwards in flight, where a launch's
    // CU time counts and a 1280-row problem on 45 efficient tiles costs a third of the CU time of 720 small ones).
    static int big_min_m = -1, big_min_t = -1;
    if (big_min_m < 0) { const char* e = getenv("MADTP_GEMM_BIG_MIN_M"); big_min_m = e ? atoi(e) : 4096; if (big_min_m < 256) big_min_m = 256; }
    if (big_min_t < 0) { const char* e = getenv("MADTP_GEMM_BIG_MIN_TILES"); big_min_t = e ? atoi(e) : 200; if (big_min_t < 1) big_min_t = 1; }
    const int big_min_tiles = cpx == 32 ? big_min_t : (big_min_t * ncu + 255) / 256;  // "most of the chip" = most of the stream's CUs
    const bool big = !m_dev.p && M >= big_min_m && t256 >= big_min_tiles;
    const bool lp16 = ab_dtype != MADTP_F32;  // 2-byte operand planes: bf16, or f16-split (three times the slab stream)
    if (lp16 && !big) {
        const int t64 = ((M + 63) / 64) * ((N + 63) / 64) * splitk, t64x128 = ((M + 63) / 64) * ((N + 127) / 128) * splitk;
        if (t64 <= 3 * ncu) cfg = 3;
        else if (t64x128 <= 3 * ncu) cfg = 1;
    }
    const int auto_cfg = cfg;  // the kernel choice below follows the AUTOMATIC tile rule; the hint only picks among the small tiles
    if (lp16 && !big) {
        const int small = ss.small_tile > -2 ? ss.small_tile : gemm_small_tile();  // scheduling hint (per stream, else madtp_gemm_set_small_tile), -1 = the rule above
        if (small >= 0 && small <= 3) cfg = small;
    }
    // MADTP_GEMM_CFG=5 forces the wave-specialised kernel, 1..4 force a gemm_kernel variant (A/B measurements)
    bool ws_ok = lp16 && splitk == 1 &&
                 (force_cfg == 5 || force_cfg == 7 || force_cfg == 8 || (force_cfg == 0 && !m_dev.p && M >= big_min_m && (big || (auto_cfg == 0 && M >= 4096))));
    if (pair) {
        static int pair_env = -1;  // MADTP_GEMM_PAIR=0: always two launches (A/B runs)
        if (pair_env < 0) { const char* e = getenv("MADTP_GEMM_PAIR"); pair_env = e ? atoi(e) : 1; }
        ws_ok = pair_env && lp16 && force_cfg == 0 && M >= big_min_m && 2 * t256 >= big_min_tiles && g.fast_epi &&
                aligned16(pair->A) && aligned16(pair->W) && aligned16(pair->C) && (!pair->bias || aligned16(pair->bias));
        if (!ws_ok) return PAIR_UNSUPPORTED;
        g.pair = 1; g.A2 = (const char*)pair->A; g.W2 = (const char*)pair->W; g.bias2 = pair->bias; g.C2 = pair->C;
    }
    if (force_cfg > 0 && force_cfg <= 4) cfg = force_cfg - 1;
    // (round 5, measured and dropped: "deep ring" variants of the small tiles - 64x64 x 6 stages / 64x128 x 5, one workgroup per CU,
    //  five / four slabs in flight - are no faster on the text encoder's 1280-row problems (768x768: 11.7 vs 11.8 us, and 24 vs 13.5 us
    //  where the tiles need three rounds): their ~10 us are launch ramp, first-touch latency and drain, not the K loop)
    if (x3 && cfg == 0) cfg = 1;  // f16-split: 64x128 tiles (the 128x128 variant would spill the kept P0 fragments)
    hipStream_t s = (hipStream_t)stream;
    GemmRecord rec;
    if (g_prof_on) {
        // timing-only events: no system-scope fence when they complete (hipEventDisableSystemFence: "can improve the accuracy of timing
        // measurements by avoiding the cost of cache writeback and invalidation, and the performance impact of those actions on the
        // execution of following work") - the un-instrumented forward has no such fences between its kernels either
        (void)hipEventCreateWithFlags(&rec.e0, hipEventDisableSystemFence); (void)hipEventCreateWithFlags(&rec.e1, hipEventDisableSystemFence);
        rec.flops = 2.0 * M * N * K; rec.dt = ab_dtype; rec.M = M; rec.N = N; rec.K = K;
        // algorithmic HBM bytes: A and W once, C once (x splits), bias, residual once
        rec.bytes = (double)esz * (x3 ? 2.0 : 1.0) * ((double)M * K + (double)N * K) + (double)M * N * ((c_dtype == MADTP_BF16 || c_dtype == MADTP_F16) ? 2 : 4) * splitk +
                    (bias ? 4.0 * N : 0.0) + (residual ? 4.0 * M * N : 0.0);
        if (pair) { rec.flops *= 2.0; rec.bytes *= 2.0; }  // two problems in this launch
        (void)hipEventRecord(rec.e0, s);
    }
#define MADTP_LAUNCH_GEMM(TT, LP, BM_, BN_, ST_, WGCU)                                                                   \
    do {                                                                                                               \
        g.ntm = (M + BM_ - 1) / BM_;                                                                                   \
        g.ntn = (N + BN_ - 1) / BN_;                                                                                   \
        const int slots_max = ((g.ntm * g.ntn + 7) / 8) * g.splitk;                                                    \
        const int per_xcd = cpx * WGCU;                                                                                \
        const int grid = 8 * (slots_max < per_xcd ? slots_max : per_xcd);                                              \
        const size_t lds = (size_t)(BM_ + BN_) * ROWB * ST_;                                                           \
        MADTP_ENSURE_MAX_LDS((gemm_kernel<TT, LP, BM_, BN_, ST_>), lds);                                               \
        hipLaunchKernelGGL((gemm_kernel<TT, LP, BM_, BN_, ST_>), dim3(grid), dim3(NTHREADS), lds, s, g);               \
    } while (0)
#define MADTP_DISPATCH_CFG(TT, LP)                                             \
    do {                                                                       \
        if (cfg == 0) MADTP_LAUNCH_GEMM(TT, LP, 128, 128, 2, 2);               \
        else if (cfg == 1) MADTP_LAUNCH_GEMM(TT, LP, 64, 128, 2, 3);           \
        else if (cfg == 2) MADTP_LAUNCH_GEMM(TT, LP, 64, 128, 3, 2);           \
        else MADTP_LAUNCH_GEMM(TT, LP, 64, 64, 3, 3);                          \
    } while (0)

    // 256x256 kernel: bf16 operands, no split-K / pair.  Chosen when its round count times its per-tile cost (measured ~1.7x a
    // 256x128 tile) beats the wave-specialised kernel's; MADTP_GEMM_CFG=6 forces it, MADTP_GEMM_SQ=0 turns it off (A/B runs).
    bool sq_ok = false, pp_ok = false;
    int pp_rows = 256;  // tile height of the ping-pong kernel: 256, or 192 (gemm_pp.hip FA = 3)
    SkWorkspace skw{nullptr, nullptr};
    const bool sk_on = ws_ok && !x3 && ab_dtype == MADTP_BF16 && (force_cfg == 5 || (force_cfg == 0 && sk_enabled() && K / 64 >= SK_MIN_SLABS)) &&
                       sk_workspace(s, skw);
    if (lp16 && splitk == 1 && !pair && (K % 64) == 0 &&
        ((size_t)M + 255) * (size_t)lda * 2 < ((size_t)1 << 32) && ((size_t)N + 255) * (size_t)ldw * 2 < ((size_t)1 << 32)) {
        static int sq_env = -1;
        if (sq_env < 0) { const char* e = getenv("MADTP_GEMM_SQ"); sq_env = e ? atoi(e) : 1; }
        const int t_sq = ((M + 255) / 256) * ((N + 255) / 256), t_192 = ((M + 191) / 192) * ((N + 255) / 256);
        // the ping-pong main loop (gemm_pp_kernel, gemm_pp.hip) needs an even slab count; MADTP_GEMM_PP=0 keeps the lockstep kernel
        // (A/B runs), cfg 9 forces its 256-row tile, cfg 10 its 192-row tile, cfg 6 forces the lockstep kernel.  Its tile costs
        // ~1.5 tiles of 256x128 (lockstep: 1.7) - profiles/r04_gemm_pp_ab.txt; a caller's sq_cost hint (several forwards in flight)
        // applies to both.  Plain f16 and f16-split operands: the 256-column tile exists as the ping-pong kernel only.
        static int pp_env = -1;
        if (pp_env < 0) { const char* e = getenv("MADTP_GEMM_PP"); pp_env = e ? atoi(e) : 1; }
        const bool pp_can = (K % 128) == 0;
        pp_ok = pp_can && (force_cfg == 9 || force_cfg == 10 || (force_cfg == 0 && pp_env));
        const bool sq_allowed = sq_env && ((!f16 && !x3) || pp_ok);
        float unit = ss.sq_cost > 0.f ? ss.sq_cost : gemm_sq_cost();
        if (pp_ok && unit > 1.5f) unit = 1.5f;
        const bool hinted = ss.sq_cost > 0.f || cpx != 32 || g_sq_cost_hinted.load(std::memory_order_relaxed);  // (the table was measured on the whole idle chip)
        // Choice for an automatic launch: (1) the measured table (gemm_table.h: per (operand class, N, K, output) and 64-row bucket
        // of M the fastest of {wave-specialised 256x128, ping-pong 256x256, ping-pong 192x256} on an idle MI355X; MADTP_GEMM_TABLE=0
        // turns it off; it steps aside while a caller's in-flight hint is in force), else (2) the round-count cost model.
        static int tab_env = -1;
        if (tab_env < 0) { const char* e = getenv("MADTP_GEMM_TABLE"); tab_env = e ? atoi(e) : 1; }
        int choice = -1;  // 0 wave-specialised, 1 ping-pong / lockstep 256x256, 2 ping-pong 192x256
        if (force_cfg == 6 && !f16 && !x3) choice = 1;
        else if (force_cfg == 9 && pp_can) choice = 1;
        else if (force_cfg == 10 && pp_can) choice = 2;
        else if (force_cfg == 0 && sq_allowed && ws_ok) {
            if (pp_ok && tab_env && cpx == 32 && (tab_env == 2 || !hinted))
                choice = gemm_table_lookup(x3, M, N, K, c_dtype == MADTP_F32);
            if (choice < 0) {
                const float cost_ws = ws_cost(t256, K / 64, sk_on, cpx);
                const float cost_sq = 2 * t_sq >= big_min_tiles ? unit * (float)((t_sq + ncu - 1) / ncu) : 1e9f;
                // a 192-row tile: 3/4 of the MFMAs of a 256-row one behind the same barriers and 7/8 of its DMA stream (measured ~0.8)
                const float cost_192 = (pp_ok && 2 * t_192 >= big_min_tiles) ? 0.8f * unit * (float)((t_192 + ncu - 1) / ncu) : 1e9f;
                choice = (cost_192 < cost_sq && cost_192 < cost_ws) ? 2 : (cost_sq < cost_ws ? 1 : 0);
            }
        }
        sq_ok = choice >= 1;
        pp_ok = pp_ok && sq_ok;
        if (choice == 2) pp_rows = 192;
        if (sq_ok && !pp_ok && (f16 || x3)) sq_ok = false;  // (no lockstep instantiation for these operand formats)
    }
    if (sq_ok) {
        g.ntm = (M + pp_rows - 1) / pp_rows;
        g.ntn = (N + 255) / 256;
        {
            // MADTP_GEMM_NGRP: column-group width of the tile order (0 = row-panel major; unset = row-panel major up to 15 column
            // tiles - every shape of the forward - and groups of 8 beyond: with 32 column tiles (8192^3) an XCD's 32 concurrent
            // tiles then share 4 A panels and 8 W panels instead of 1 + 32: 1.32 -> 1.53-1.55 PF, profiles/r04_gemm_pp_ab.txt)
            static int grp_env = -2;
            if (grp_env == -2) { const char* e = getenv("MADTP_GEMM_NGRP"); grp_env = e ? atoi(e) : -1; }
            const int grp = grp_env >= 0 ? grp_env : (g.ntn >= 16 ? 8 : 0);
            g.ngrp = (grp > 0 && grp < g.ntn) ? grp : 0;
        }
        const int slots_max = (g.ntm * g.ntn + 7) / 8;
        const int cap = gemm_wg_per_xcd(ss);
        const int grid = 8 * (slots_max < cap ? slots_max : cap);
        const size_t lds = (size_t)2 * (256 + 256) * ROWB;
        if (pp_ok) {
            const int om = c_dtype == MADTP_F32 ? OM_F32 : (c_dtype == MADTP_F16S ? OM_F16S : (c_dtype == MADTP_F16 ? OM_F16 : OM_BF16));
            const int rc = madtp_gemm_pp_launch(&g, om, x3 ? 2 : (f16 ? 1 : 0), pp_rows, grid, s);
            if (rc) return rc;
        } else if (c_dtype == MADTP_BF16) {
            MADTP_ENSURE_MAX_LDS((gemm_sq_kernel<OM_BF16>), lds);
            hipLaunchKernelGGL((gemm_sq_kernel<OM_BF16>), dim3(grid), dim3(512), lds, s, g);
        } else {
            MADTP_ENSURE_MAX_LDS((gemm_sq_kernel<OM_F32>), lds);
            hipLaunchKernelGGL((gemm_sq_kernel<OM_F32>), dim3(grid), dim3(512), lds, s, g);
        }
    } else if (ws_ok) {
        // wave-specialised 256x128 kernel (one 12-wave workgroup per CU, 144 KiB LDS ring)
        g.ntm = (M + 255) / 256;
        g.ntn = (N + 127) / 128;
        // column groups (tile_mn): keep one group's W rows (~2.4 MB) L2-resident when W as a whole is far larger than L2
        {
            static int grp_env = -2;  // MADTP_GEMM_NGRP: unset = automatic, 0 = off, n > 0 = forced group width (A/B runs)
            if (grp_env == -2) { const char* e = getenv("MADTP_GEMM_NGRP"); grp_env = e ? atoi(e) : -1; }
            int G = grp_env > 0 ? grp_env : (12 * 768) / K;
            if (G < 1) G = 1;
            const bool on = grp_env > 0 || (grp_env == -1 && g.ntn >= 4 * G);
            g.ngrp = (on && G < g.ntn) ? G : 0;
        }
        const int slots_max = (g.ntm * g.ntn * (g.pair ? 2 : 1) + 7) / 8;
        const int cap = gemm_wg_per_xcd(ss);
        const int grid = 8 * (slots_max < cap ? slots_max : cap);
        if (sk_on && grid == 256) { g.sk = 1; g.sk_ws = skw.ws; g.sk_tick = skw.tick; }
        const size_t lds = (size_t)3 * (256 + 128) * ROWB;
#define MADTP_LAUNCH_WS(X3_, OM_, M32_, ...)                                                                  \
    do {                                                                                                     \
        MADTP_ENSURE_MAX_LDS((gemm_ws_kernel<X3_, OM_, M32_ __VA_OPT__(,) __VA_ARGS__>), lds);                \
        hipLaunchKernelGGL((gemm_ws_kernel<X3_, OM_, M32_ __VA_OPT__(,) __VA_ARGS__>), dim3(grid), dim3(768), lds, s, g); \
    } while (0)
        // 32x32x16 consumer loop (bf16 operands, vector epilogue, no stream-K tail).  OFF by default - measured SLOWER than the
        // 16x16x32 loop on every ViT shape of the forward (profiles/r03_gemm_m32_ab.txt: MFMA-only stream 1.57 vs 1.66 PF, whole
        // kernel 623-837 vs 851-981 TF): MADTP_GEMM_M32=1 turns it on in the automatic dispatch, cfg 8 forces it (tests, A/B runs)
        static int m32_env = -1;
        if (m32_env < 0) { const char* e = getenv("MADTP_GEMM_M32"); m32_env = e ? atoi(e) : 0; }
        const bool m32 = !x3 && !f16 && g.fast_epi && !g.sk && (force_cfg == 8 || (force_cfg == 0 && m32_env));
        if (f16) {
            if (c_dtype == MADTP_F16) MADTP_LAUNCH_WS(false, OM_F16, false, true); else MADTP_LAUNCH_WS(false, OM_F32, false, true);
        } else if (x3) {
            if (c_dtype == MADTP_F16S) MADTP_LAUNCH_WS(true, OM_F16S, false); else MADTP_LAUNCH_WS(true, OM_F32, false);
        } else if (m32) {
            if (c_dtype == MADTP_BF16) MADTP_LAUNCH_WS(false, OM_BF16, true); else MADTP_LAUNCH_WS(false, OM_F32, true);
        } else {
            if (c_dtype == MADTP_BF16) MADTP_LAUNCH_WS(false, OM_BF16, false); else MADTP_LAUNCH_WS(false, OM_F32, false);
        }
#undef MADTP_LAUNCH_WS
    } else if (ab_dtype == MADTP_BF16) {
        if (c_dtype == MADTP_BF16) MADTP_DISPATCH_CFG(bf16_t, OM_BF16); else MADTP_DISPATCH_CFG(bf16_t, OM_F32);
    } else if (f16) {
        if (c_dtype == MADTP_F16) MADTP_DISPATCH_CFG(F16P, OM_F16); else MADTP_DISPATCH_CFG(F16P, OM_F32);
    } else if (x3) {
        if (c_dtype == MADTP_F16S) MADTP_DISPATCH_CFG(F16S, OM_F16S); else MADTP_DISPATCH_CFG(F16S, OM_F32);
    } else {
        if (c_dtype == MADTP_BF16) MADTP_DISPATCH_CFG(float, OM_BF16); else MADTP_DISPATCH_CFG(float, OM_F32);
    }
    if (g_prof_on) {
        (void)hipEventRecord(rec.e1, s);
        g_prof.push_back(rec);
    }
    MADTP_LAUNCH_CHECK();
    return 0;
}
