// Device-side pieces shared by the GEMM translation units (gemm.hip, gemm_pp.hip): argument block, tile maps, LDS-DMA
// staging helpers, activations and the register-to-global epilogues.  See gemm.hip for the design notes.
#pragma once
#include "common.h"
#include <type_traits>

namespace {


constexpr int ROWB = 128;      // bytes per operand row slab (64 bf16 / 32 f32 along K)
constexpr int NTHREADS = 256;

struct GemmArgs {
    const char* A; const char* W; const float* bias; const float* residual; void* C;
    int M, N, K, lda, ldw, ldc, ldr, act, ntm, ntn, dbg, fast_epi, splitk, ngrp;
    // madtp_gemm_pair (wave-specialised kernel only): a second problem of the same shape, tiles [ntm*ntn, 2*ntm*ntn)
    const char* A2; const char* W2; const float* bias2; void* C2; int pair;
    float out_scale;
    float acc_scale;  // multiplies the raw accumulator before the bias: 2^-s of a pre-scaled f16-split weight, 1 otherwise
    float acc_scale2; // the same for the second problem of a pair launch
    // stream-K tail of the wave-specialised kernel (sk_plan): partial accumulators [8 XCDs][32 units][8 waves][16][64 lanes] f32x4
    // and the per-(tail tile, wave) tickets [8][16][8]; sk == 0: off
    float* sk_ws; int* sk_tick; int sk;
    int desc;  // gemm_kernel: both operands fit a 32-bit buffer descriptor -> descriptor-based LDS-DMA (no per-slab address arithmetic)
    int* range_flag;  // f16 outputs (OM_F16S / OM_F16): the pinned range flag (common.h), else unused
    const int32_t* m_dev; int m_mul;  // gemm_kernel: M = m_mul * *m_dev, read by the kernel (sync-free encoder path, common.h DevN)
};
struct F16P { unsigned short bits; };  // operand tag of gemm_kernel<>: plain f16 elements (MADTP_F16) on the f16 MFMA

// f16-split operands (common.h).  The kernels walk K as a stream of 2 * K/64 slab steps: step 2t stages [P0 | Q1] of k-slab t,
// step 2t+1 stages [P1 | Q0]; the three products of a k-slab are P0 Q1 (first step), P0 Q0 and P1 (Q0 2^-11) (second step, with
// the P0 fragments kept in registers and Q0 scaled in registers): 96 MFMAs per 2 staged slab pairs instead of 96 per 3.
struct F16S {};  // operand tag of gemm_kernel<>
__device__ __forceinline__ bf16x8 x3_scale_lo(bf16x8 q) {  // Q0 * 2^-11 (v_pk_mul_f16; f16 denormals are kept)
    const f16x8 v = __builtin_bit_cast(f16x8, q) * (_Float16)(1.0f / F16S_LO_SCALE);
    return __builtin_bit_cast(bf16x8, v);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// output modes of the epilogue
constexpr int OM_F32 = 0, OM_BF16 = 1, OM_F16S = 2, OM_F16 = 3;  // OM_F16: plain f16 output (operands MADTP_F16)

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case MADTP_ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case MADTP_ACT_QUICK_GELU: return v / (1.0f + expf(-1.702f * v));
        case MADTP_ACT_RELU: return fmaxf(v, 0.0f);
        default: return v;
    }
}

// LDS swizzle key of a tile row: the 16-byte chunk index of row r is XORed with key(r).  Consecutive fragment rows use
// r&7.  The bf16-output kernels read W fragments through the permuted row map of wfrag_row<true> (rows 8a+b+const), for
// which r&7 repeats every 8 rows and costs a 2-way bank conflict; PERM keys on bits 1,3,4 instead (conflict-free,
// checked with the ds_read_b128 lane-group model of MI355X_MICROARCH.md).
template <bool PERM>
__device__ __forceinline__ int swz_key(int r) { return PERM ? (((r >> 1) & 1) | (((r >> 3) & 3) << 1)) : (r & 7); }

// Balanced XCD-aware tile partition: the ntm*ntn output tiles are numbered row-panel major (t = panel*ntn + column tile)
// and XCD x (workgroups with blockIdx%8 == x) owns the contiguous range [x*T/8, (x+1)*T/8): every XCD gets T/8 tiles
// +-1 (no 8-panel quantisation), the column tiles of an A panel are walked back-to-back by neighbouring workgroups of
// one XCD (the panel is fetched from HBM once and then hits that XCD's L2), and W streams from L2 / Infinity Cache.
__device__ __forceinline__ void xcd_tiles(int T, int xcd, int& t0, int& nt) {
    t0 = (int)(((long)xcd * T) >> 3);
    nt = (int)(((long)(xcd + 1) * T) >> 3) - t0;
}

// issue the LDS-DMA of one ROWS x 128 B operand tile: ROWS/8 wave-instructions of 1 KiB, ROWS/32 per wave
// (tile_row0: row of the tile this piece starts at - the swizzle key is a function of the row WITHIN the tile)
// Tile t of the launch order -> (row tile, column tile).  Plain row-major when ngrp == 0.  For very wide outputs (the
// stacked cross-attention K/V projection of all text layers, N = 18432: W is 28 MB) the column tiles are walked in GROUPS
// of ngrp: all row tiles of one group before the next group, so the group's W rows (<= ~2.4 MB) stay in every XCD's L2
// instead of the whole W streaming through it once per row panel.
__device__ __forceinline__ void tile_mn(const GemmArgs& g, int t, int& tm, int& tn) {
    if (g.ngrp == 0) { tm = t / g.ntn; tn = t % g.ntn; return; }
    const int per = g.ntm * g.ngrp;
    const int grp = t / per, r = t - grp * per;
    const int gw = min(g.ngrp, g.ntn - grp * g.ngrp);
    tm = r / gw; tn = grp * g.ngrp + (r - tm * gw);
}

// Stream-K tail of a persistent kernel (the wave-specialised 256x128 one): an XCD's `nslots` tiles are `rounds` full rounds of
// its `gl` workgroups plus `rem` tiles.  With rem <= gl/2 the last round would leave most CUs idle for a whole tile time (the
// N = 768 GEMMs of the ViT layers at 11-14 k rows are 1.03-1.3 rounds), so each of the rem tiles is cut along K into `parts`
// pieces run by `parts` workgroups; every consumer wave parks its 64x64 partial in the workspace and the LAST wave to arrive
// (ticket per tile and wave position) sums the pieces in piece order - deterministic - and runs the epilogue.
constexpr int SK_MAX_PARTS = 8;
__host__ __device__ __forceinline__ int sk_parts(int rem, int gl, int nk) {
    if (rem <= 0) return 0;
    int p = gl / rem;
    if (p > SK_MAX_PARTS) p = SK_MAX_PARTS;
    if (p > nk / 2) p = nk / 2;  // at least two slabs per piece
    return p >= 2 ? p : 0;
}

// The same with a buffer descriptor: NQ instructions of this wave, per-lane source offsets roff[] (row within the tile, swizzled
// slot - constants of the kernel), scalar offset soff (tile origin + K position); rows past `bytes` read as zeros.
// (A __device__ function on purpose: with these builtins inside the kernel's issue lambda the host pass of this compiler emits
//  no launch stub for the kernel template.)
template <int NQ>
__device__ __forceinline__ void stage_tile_desc(const char* base, unsigned bytes, const unsigned* roff, int soff, char* lds) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < NQ; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + q * 1024), 16, roff[q], soff, 0, 0);
}

template <int ESZ, int ROWS, bool PERM = false>
__device__ __forceinline__ void stage_tile(const char* base, int row0, int max_row, int ld_elems, int kbyte0,
                                           char* lds_tile, int wave, int lane, int tile_row0 = 0) {
    const int sub = lane >> 3;                       // row within the 8-row group
#pragma unroll
    for (int q = 0; q < ROWS / 32; ++q) {
        const int grp = wave * (ROWS / 32) + q;
        const int chunk = (lane & 7) ^ swz_key<PERM>(tile_row0 + grp * 8 + sub);  // inverse swizzle on the SOURCE
        int row = row0 + grp * 8 + sub;
        row = row < max_row ? row : max_row;
        const char* src = base + ((size_t)row * ld_elems) * ESZ + kbyte0 + chunk * 16;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds_tile + grp * 1024), 16, 0, 0);
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// GELU for outputs that are rounded to bf16 anyway (fast mode only; the f32 path keeps erff):
//   gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) ~= x * sigmoid(x (a + b x^2 + c x^4)),   x^2 clamped to 25,
// a, b, c fitted (minimax on [-8, 8]) to a maximum ABSOLUTE error of 2.6e-5 - a tenth of a bf16 ulp at |y| = 0.06 -
// for 7 VALU + 2 transcendental instructions per element (the erf form costs twice that, and with one
// 64-accumulator epilogue per 12 K slabs the activation is a visible part of the fc1 GEMM).
__device__ __forceinline__ float gelu_fast(float v) {
    constexpr float L2E = 1.4426950408889634f;
    const float x2 = fminf(v * v, 25.0f);
    float p = fmaf(x2, 7.03033581e-04f * L2E, -7.40112921e-02f * L2E);
    p = fmaf(x2, p, -1.59501577f * L2E);
    const float e = __builtin_amdgcn_exp2f(v * p);  // exp(-u); +inf for very negative v -> rcp = 0 -> -0
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// compile-time activation: the epilogue is instantiated per activation code so it stays straight-line code
// (LP_OUT here = the output is rounded to bf16 anyway: the cheap forms; f32 and f16-split outputs keep erff / expf)
template <bool LP_OUT, int ACT>
__device__ __forceinline__ float epi_act(float v) {
    if constexpr (ACT == MADTP_ACT_GELU_ERF) return LP_OUT ? gelu_fast(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    else if constexpr (ACT == MADTP_ACT_QUICK_GELU)
        return LP_OUT ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v))
                      : v / (1.0f + expf(-1.702f * v));
    else if constexpr (ACT == MADTP_ACT_RELU) return fmaxf(v, 0.0f);
    else return v;
}

// Output-fragment geometry.  The MFMA is issued with the operands SWAPPED (D = Wfrag . Afrag^T), so lane
// (l16 = lane&15, g = lane>>4) holds, for fragment (i,j), output row m = 16i + l16 and the FOUR CONSECUTIVE columns
// owned by W-fragment rows rho = 4g..4g+3: the epilogue stores vectors straight from registers, no LDS transpose.
//   f32 out : W-fragment row rho of fragment j is tile column 16j + rho       -> float4 per (i,j), 64 B per row/instr
//   bf16 out: W-fragment row rho of fragment j is tile column 32(j>>1) + 8(rho>>2) + 4(j&1) + (rho&3)
//             -> fragments (2jp, 2jp+1) give 8 consecutive columns = one 16-byte store, 64 B per row/instr
// (which W row feeds which fragment row is only the LDS row a lane reads - free to choose.)
template <bool LP_OUT>
__host__ __device__ __forceinline__ constexpr int wfrag_row(int j, int rho) {
    return LP_OUT ? 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3) : 16 * j + rho;
}

// ---- 32x32x16 fragments (gemm_ws_kernel<.., M32 = true>) ---------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 does the work of two 16x16x32 instructions in one issue slot and runs the matrix pipe at its
// full rate (32 cycles per instruction against 2 x ~17).  Operand lane map: lane l supplies fragment row l & 31 and the 8
// consecutive k of 16-byte chunk (l >> 5) of the 16-deep k-step; C/D: column l & 31, rows (t & 3) + 8 (t >> 2) + 4 (l >> 5)
// for register t.  With the operands swapped (D = Wfrag . Afrag^T) lane (l32 = l & 31, h = l >> 5) holds output row
// 32 i + l32 and, per fragment (i, j), sixteen columns in four runs of four consecutive W-fragment rows 8 q + 4 h + e.
//   f32 out : W-fragment row rho of fragment j is tile column 32 j + rho                 -> float4 per (i, j, q)
//   bf16 out: W-fragment row rho is tile column 32 j + swap(bit 2, bit 3)(rho)           -> registers 8p..8p+7 are the 8
//             consecutive columns 32 j + 16 p + 8 h .. +7 = one 16-byte store per (i, j, p)
// LDS swizzle of the M32 stage image: a ds_read_b128 is serviced in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}
// (+32): the 16 rows of a group (same chunk) must fall on 16 different 16-byte slots of the 256-byte bank row, i.e. the
// key must take all 8 values over the 8 even and over the 8 odd rows of a group - (r >> 1) & 7 does (the bf16-output row
// permutation maps each group onto itself).  One key function for both operands.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__host__ __device__ __forceinline__ constexpr int swz_key32(int r) { return (r >> 1) & 7; }
template <bool LP_OUT>
__host__ __device__ __forceinline__ constexpr int wfrag_row32(int j, int rho) {
    return 32 * j + (LP_OUT ? ((rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1)) : rho);
}

// epilogue of one consumer wave's 64x64 block held as 2x2 fragments of 32x32 (fast path only: the launcher keeps shapes
// that need the scalar fallback on the 16x16 kernel).  Same structure as epilogue(): every load first, one wait, then
// descriptor-bounded 16-byte stores; the f32 residual is fetched per 32-row fragment row.
template <int OM, int ACT, bool HAS_RES>
__device__ __forceinline__ void epilogue32(const GemmArgs& g, const f32x16 (&acc)[2][2], int row_t, int col_w, int l32, int h) {
    constexpr bool LP_OUT = OM != OM_F32;
    constexpr bool FAST_ACT = OM == OM_BF16 || OM == OM_F16;
    constexpr int CSZ = LP_OUT ? 2 : 4;
    constexpr int NJ = LP_OUT ? 2 : 4;   // column vectors per lane, fragment row and fragment column: p (8 columns) or q (4)
    constexpr int NV = LP_OUT ? 2 : 1;   // float4s per column vector
    const int ldc = g.ldc < 0 ? -g.ldc : g.ldc;
    const int rows_valid = max(min(g.M - row_t, 64), 0);
    const unsigned long long cb = (unsigned long long)((char*)g.C + (size_t)row_t * ldc * CSZ);
    const unsigned cb_lo = __builtin_amdgcn_readfirstlane((unsigned)cb);
    const unsigned cb_hi = __builtin_amdgcn_readfirstlane((unsigned)(cb >> 32));
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)cb_hi << 32) | cb_lo), 0, __builtin_amdgcn_readfirstlane(rows_valid * ldc * CSZ), 0x00020000);
    // one pass per fragment column j (32 output columns): its bias and - f32 residual stream - the residual of both fragment
    // rows are fetched first, ONE wait, then the 2 x NJ stores: two exposed round trips per 64x64 block, 48 live load registers
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int colv[NJ];
        bool cok[NJ];
        f32x4 bv[NJ][NV];
#pragma unroll
        for (int jv = 0; jv < NJ; ++jv) {
            colv[jv] = col_w + 32 * j + (LP_OUT ? 16 * jv + 8 * h : 8 * jv + 4 * h);
            cok[jv] = colv[jv] < g.N;
#pragma unroll
            for (int u = 0; u < NV; ++u) bv[jv][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (g.bias) {  // uniform; the unconditional wait below closes this diamond
#pragma unroll
            for (int jv = 0; jv < NJ; ++jv)
#pragma unroll
                for (int u = 0; u < NV; ++u) bv[jv][u] = *(const f32x4*)(g.bias + (cok[jv] ? colv[jv] : 0) + 4 * u);
        }
        f32x4 rv[HAS_RES ? 2 : 1][HAS_RES ? NJ : 1];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = min(row_t + 32 * i + l32, g.M - 1);
#pragma unroll
                for (int jv = 0; jv < NJ; ++jv)
                    rv[i][jv] = *(const f32x4*)(g.residual + (size_t)row * g.ldr + (cok[jv] ? colv[jv] : 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) (the first one also covers the previous tile's stores, a main loop old)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jv = 0; jv < NJ; ++jv) {
                const int t0 = LP_OUT ? 8 * jv : 4 * jv;  // first accumulator register of this column vector
                f32x4 v[NV];
#pragma unroll
                for (int u = 0; u < NV; ++u) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[u][e] = epi_act<FAST_ACT, ACT>(fmaf(acc[i][j][t0 + 4 * u + e], g.acc_scale, bv[jv][u][e])) * g.out_scale;
                    if constexpr (HAS_RES) v[u] += rv[i][jv];
                }
                const unsigned off = cok[jv] ? (unsigned)((32 * i + l32) * ldc + colv[jv]) * CSZ : 0x80000000u;
                u32x4 bits;
                if constexpr (OM == OM_BF16) bits = __builtin_bit_cast(u32x4, pack_bf16x8(v[0], v[1]));
                else if constexpr (OM == OM_F16) bits = __builtin_bit_cast(u32x4, pack_f16x8(v[0], v[1]));
                else bits = __builtin_bit_cast(u32x4, v[0]);
                __builtin_amdgcn_raw_buffer_store_b128(bits, crsrc, off, 0, 0);
            }
    }
}

// HAS_RES: the residual is read by the epilogue itself (kernels that do not prefetch it into `res`); compile-time so
// that the fast path below is straight-line code.
template <int OM, int ACT, bool HAS_RES, int FM, int FN, int BM, int BN, bool FAST_ONLY = false, int RM = 1, int RN = 1>
__device__ __forceinline__ void epilogue(const GemmArgs& g, const f32x4 (&acc)[FM][FN], const f32x4 (&res)[RM][RN],
                                         int m0, int n0, int wr, int wc, int l16, int grp4, size_t c_off) {
    constexpr bool LP_OUT = OM != OM_F32;    // 8-consecutive-column fragment layout (wfrag_row<true>)
    constexpr bool FAST_ACT = OM == OM_BF16 || OM == OM_F16;
    const bool c_bf16 = g.ldc < 0;
    const int ldc = c_bf16 ? -g.ldc : g.ldc;
            const int col_w = n0 + wc * (BN / 2);
            if (FAST_ONLY || g.fast_epi) {
                // The fast path is STRAIGHT-LINE code with every load (bias, residual) issued before the first store.
                // gfx950 counts loads and stores in one in-order vmcnt and the compiler's s_waitcnt insertion is
                // conservative at control-flow merges: with the old per-vector `if (row < M) { load residual; store }`
                // every store was followed by an s_waitcnt vmcnt(0), i.e. one L2 write round trip per output vector
                // (~3 us per 256x128 tile, a fifth of the big GEMMs).  So: loads first, one wait, then fire-and-forget
                // stores, and the row / column bounds are enforced by the buffer descriptor (out-of-range offsets are
                // dropped by the hardware) instead of by branches.
                constexpr int NJ = LP_OUT ? FN / 2 : FN;  // column vectors per lane: 8 (bf16 out) or 4 (f32 out) columns each
                constexpr int NV = LP_OUT ? 2 : 1;        // float4s per column vector
                constexpr int CSZ = LP_OUT ? 2 : 4;
                constexpr bool RES_PREF = !LP_OUT && RM == FM;
                float range_mx = 0.f;  // f16 outputs: a value outside the f16 range raises the range flag (f16_range_acc)
                const int row_t = m0 + wr * (BM / 2);  // first row of this wave's 64-row (BM/2) block: wave-uniform
                const int rows_valid = max(min(g.M - row_t, BM / 2), 0);
                const unsigned long long cb = (unsigned long long)((char*)g.C + (c_off + (size_t)((g.dbg & 8) ? (row_t & 127) : row_t) * ldc) * CSZ);  // dbg 8: timing experiment, all tiles store to the first 128 rows (L2-resident)
                const unsigned cb_lo = __builtin_amdgcn_readfirstlane((unsigned)cb);  // pin the descriptor in SGPRs
                const unsigned cb_hi = __builtin_amdgcn_readfirstlane((unsigned)(cb >> 32));
                const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(((unsigned long long)cb_hi << 32) | cb_lo), 0,
                    __builtin_amdgcn_readfirstlane(rows_valid * ldc * CSZ), 0x00020000);
                int colv[NJ];
                bool cok[NJ];
                f32x4 bv[NJ][NV];
#pragma unroll
                for (int jv = 0; jv < NJ; ++jv) {
                    colv[jv] = col_w + (LP_OUT ? 32 * jv + 8 * grp4 : 16 * jv + 4 * grp4);
                    cok[jv] = colv[jv] < g.N;
#pragma unroll
                    for (int u = 0; u < NV; ++u) bv[jv][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (g.bias) {  // uniform; the unconditional wait below closes this diamond
#pragma unroll
                    for (int jv = 0; jv < NJ; ++jv)
#pragma unroll
                        for (int u = 0; u < NV; ++u) bv[jv][u] = *(const f32x4*)(g.bias + (cok[jv] ? colv[jv] : 0) + 4 * u);
                }
                // Residual of the f32-output tiles (HAS_RES):
                //   RES_PREF: the kernel prefetched all FM fragment rows into `res` two slabs before the epilogue;
                //   RES_LOAD: (12-wave kernel, 168 VGPRs) the epilogue fetches IC = FM/2 fragment rows at a time: loads, ONE
                //             full wait, then that chunk's stores - two exposed round trips per tile instead of one per vector.
                constexpr bool RES_LOAD = HAS_RES && !RES_PREF;
                constexpr int IC = RES_LOAD ? (FM > 2 ? FM / 2 : FM) : FM;
                f32x4 rv[RES_LOAD ? IC : 1][RES_LOAD ? NJ : 1][NV];
#pragma unroll
                for (int i0 = 0; i0 < FM; i0 += IC) {
                    if constexpr (RES_LOAD) {
#pragma unroll
                        for (int ii = 0; ii < IC; ++ii) {
                            const int row = min(row_t + l16 + 16 * (i0 + ii), g.M - 1);
#pragma unroll
                            for (int jv = 0; jv < NJ; ++jv) {
                                const float* rp = g.residual + (size_t)row * g.ldr + (cok[jv] ? colv[jv] : 0);
#pragma unroll
                                for (int u = 0; u < NV; ++u) rv[ii][jv][u] = *(const f32x4*)(rp + 4 * u);
                            }
                        }
                    }
                    if (RES_LOAD || i0 == 0) {
                        // (the first wait also covers the previous tile's stores, issued a whole main loop ago)
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int ii = 0; ii < IC; ++ii) {
                        const int i = i0 + ii;
#pragma unroll
                        for (int jv = 0; jv < NJ; ++jv) {
                            f32x4 v[NV];
#pragma unroll
                            for (int u = 0; u < NV; ++u) {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    v[u][e] = epi_act<FAST_ACT, ACT>(fmaf(acc[i][NV * jv + u][e], g.acc_scale, bv[jv][u][e])) * g.out_scale;
                                if constexpr (RES_PREF && HAS_RES) v[u] += res[i][jv];
                                if constexpr (RES_LOAD) v[u] += rv[ii][jv][u];
                            }
                            // rows past M fall outside the descriptor; columns past N are pushed outside it
                            const unsigned off = cok[jv] ? (unsigned)((16 * i + l16) * ldc + colv[jv]) * CSZ : 0x80000000u;
                            u32x4 bits;
                            if constexpr (OM == OM_F16S) {
                                // the two planes of the split output: P0 at column c, P1 at column N + c of the same row
                                u32x4 lo_bits;
                                split_f16x8(v[0], v[1], bits, lo_bits);
                                __builtin_amdgcn_raw_buffer_store_b128(lo_bits, crsrc, cok[jv] ? off + (unsigned)g.N * 2u : off, 0, 0);
                                range_mx = f16_range_acc(f16_range_acc(range_mx, v[0]), v[1]);
                            } else if constexpr (OM == OM_BF16) bits = __builtin_bit_cast(u32x4, pack_bf16x8(v[0], v[1]));
                            else if constexpr (OM == OM_F16) {
                                // (no range accumulation here: an f16 output that overflows turns the next f32-output GEMM's
                                //  result into inf / NaN, which the LayerNorm behind it flags - f16_range_bad is true for NaN -
                                //  still inside the forward; the 64 extra VALU per lane cost the 60 us launches 2-3 %)
                                bits = __builtin_bit_cast(u32x4, pack_f16x8(v[0], v[1]));
                            }
                            else bits = __builtin_bit_cast(u32x4, v[0]);
                            __builtin_amdgcn_raw_buffer_store_b128(bits, crsrc, off, 0, 0);
                        }
                    }
                }
                if constexpr (OM == OM_F16S) f16_range_raise(g.range_flag, !(range_mx < 65520.0f));
            } else if constexpr (!FAST_ONLY) {
                // generic fallback (N or a leading dimension not a multiple of 8 elements, e.g. the 2-logit head)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = col_w + wfrag_row<LP_OUT>(j, 4 * grp4 + r);
                        if (col >= g.N) continue;
                        const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            const int row = m0 + wr * (BM / 2) + i * 16 + l16;
                            if (row >= g.M) continue;
                            float v = epi_act<FAST_ACT, ACT>(fmaf(acc[i][j][r], g.acc_scale, bv)) * g.out_scale;
                            if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                            const size_t ci = c_off + (size_t)row * ldc + col;  // c_off: the split-K partial slab of this slot
                            if constexpr (OM == OM_F16S) {
                                const _Float16 h = (_Float16)v;
                                ((_Float16*)g.C)[ci] = h;
                                ((_Float16*)g.C)[ci + g.N] = (_Float16)((v - (float)h) * F16S_LO_SCALE);
                                f16_range_raise(g.range_flag, f16_range_bad(v));
                            } else if constexpr (OM == OM_F16) {
                                ((_Float16*)g.C)[ci] = (_Float16)v;
                                f16_range_raise(g.range_flag, f16_range_bad(v));
                            } else if (c_bf16) ((bf16_t*)g.C)[ci] = f32_to_bf16(v);
                            else ((float*)g.C)[ci] = v;
                        }
                    }
            }
}

}  // namespace
