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
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;      // tile rows, bytes per row slab
constexpr int TILE_BYTES = BM * ROWB;              // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;        // A + W
constexpr int NTHREADS = 256;

struct GemmArgs {
    const char* A; const char* W; const float* bias; const float* residual; void* C;
    int M, N, K, lda, ldw, ldc, ldr, act, ntm, ntn, dbg, fast_epi;
    float out_scale;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case MADTP_ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case MADTP_ACT_QUICK_GELU: return v / (1.0f + expf(-1.702f * v));
        case MADTP_ACT_RELU: return fmaxf(v, 0.0f);
        default: return v;
    }
}

// issue the LDS-DMA of one 128x128B operand tile: 16 wave-instructions of 1 KiB, 4 per wave
template <int ESZ>
__device__ __forceinline__ void stage_tile(const char* base, int row0, int max_row, int ld_elems, int kbyte0,
                                           char* lds_tile, int wave, int lane) {
    const int sub = lane >> 3;                       // row within the 8-row group
    const int chunk = (lane & 7) ^ sub;              // inverse swizzle on the SOURCE
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int grp = wave * 4 + q;
        int row = row0 + grp * 8 + sub;
        row = row < max_row ? row : max_row;
        const char* src = base + ((size_t)row * ld_elems) * ESZ + kbyte0 + chunk * 16;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds_tile + grp * 1024), 16, 0, 0);
    }
}

// fast erf for outputs that are rounded to bf16 anyway: Abramowitz-Stegun 7.1.26, |err| < 2e-7 (+ fast exp/rcp)
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __frcp_rn(1.0f + 0.3275911f * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-x * x);
    return 0.5f * v * (1.0f + copysignf(erf_abs, v));
}

template <bool LP_OUT>
__device__ __forceinline__ float epi_act(float v, int act) {
    if (LP_OUT && act == MADTP_ACT_GELU_ERF) return gelu_fast(v);
    return apply_act(v, act);
}

// Output-fragment geometry.  The MFMA is issued with the operands SWAPPED (D = Wfrag . Afrag^T), so lane
// (l16 = lane&15, g = lane>>4) holds, for fragment (i,j), output row m = 16i + l16 and the FOUR CONSECUTIVE columns
// owned by W-fragment rows rho = 4g..4g+3: the epilogue stores vectors straight from registers, no LDS transpose.
//   f32 out : W-fragment row rho of fragment j is tile column 16j + rho       -> float4 per (i,j), 64 B per row/instr
//   bf16 out: W-fragment row rho of fragment j is tile column 32(j>>1) + 8(rho>>2) + 4(j&1) + (rho&3)
//             -> fragments (2jp, 2jp+1) give 8 consecutive columns = one 16-byte store, 64 B per row/instr
// (which W row feeds which fragment row is only the LDS row a lane reads - free to choose.)
template <bool LP_OUT>
__device__ __forceinline__ int wfrag_row(int j, int rho) {
    return LP_OUT ? 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3) : 16 * j + rho;
}

template <typename T, bool LP_OUT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(GemmArgs g) {
    constexpr int ESZ = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // Persistent, XCD-aware tile schedule: block b runs on XCD b%8 (observed dispatch rule; only speed depends on
    // it).  XCD x owns row panels x, x+8, ... and walks (panel, column tile) slots in order, so an A panel is
    // fetched from HBM once and then served by that XCD's L2 to the workgroups computing its other column tiles.
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, gl = gridDim.x >> 3;
    const int npanel = (g.ntm - xcd + 7) >> 3;
    const int nslots = npanel * g.ntn;
    int slot = lb;
    if (slot >= nslots) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l16 = lane & 15, grp4 = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K * ESZ / ROWB;
    const int n_pad_max = g.ntn * BN - 1;
    const bool c_bf16 = g.ldc < 0;  // sign bit of ldc carries the output dtype (see launcher)
    const int ldc = c_bf16 ? -g.ldc : g.ldc;

    // LDS byte offsets of this lane's fragment rows (row*128) and their swizzle keys (row&7)
    int a_off[4], a_key[4], b_off[4], b_key[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + l16;
        a_off[i] = ra * ROWB; a_key[i] = ra & 7;
        const int rb = wc * 64 + wfrag_row<LP_OUT>(i, l16);
        b_off[i] = rb * ROWB; b_key[i] = rb & 7;
    }

    int m0 = ((slot / g.ntn) * 8 + xcd) * BM, n0 = (slot % g.ntn) * BN;
    // the operand slabs of ALL tiles of this workgroup form one continuous double-buffered stream: slab s lives in
    // stage s&1 and the DMA of slab s+1 (same tile, or the first slab of the NEXT tile) is issued right after the
    // barrier of slab s, so the pipeline never drains and the epilogue stores overlap the next tile's loads.
    unsigned s = 0;
    if (!(g.dbg & 2)) {
        stage_tile<ESZ>(g.A, m0, g.M - 1, g.lda, 0, smem, wave, lane);
        stage_tile<ESZ>(g.W, n0, n_pad_max, g.ldw, 0, smem + TILE_BYTES, wave, lane);
    }
    while (true) {
        const int next_slot = slot + gl;
        for (int kt = 0; kt < nk; ++kt, ++s) {
            __syncthreads();  // slab s landed (the barrier drains the LDS-DMA); stage (s+1)&1 is free again
            if (!(g.dbg & 2)) {
                char* nxt = smem + ((s + 1) & 1) * STAGE_BYTES;
                if (kt + 1 < nk) {
                    stage_tile<ESZ>(g.A, m0, g.M - 1, g.lda, (kt + 1) * ROWB, nxt, wave, lane);
                    stage_tile<ESZ>(g.W, n0, n_pad_max, g.ldw, (kt + 1) * ROWB, nxt + TILE_BYTES, wave, lane);
                } else if (next_slot < nslots) {
                    const int nm0 = ((next_slot / g.ntn) * 8 + xcd) * BM, nn0 = (next_slot % g.ntn) * BN;
                    stage_tile<ESZ>(g.A, nm0, g.M - 1, g.lda, 0, nxt, wave, lane);
                    stage_tile<ESZ>(g.W, nn0, n_pad_max, g.ldw, 0, nxt + TILE_BYTES, wave, lane);
                }
            }
            const char* sa = smem + (s & 1) * STAGE_BYTES;
            const char* sw = sa + TILE_BYTES;
            if (g.dbg & 4) continue;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int chunk = kk * 4 + grp4;
                if constexpr (ESZ == 2) {
                    bf16x8 a[4], b[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = *(const bf16x8*)(sa + a_off[i] + ((chunk ^ a_key[i]) << 4));
                        b[i] = *(const bf16x8*)(sw + b_off[i] + ((chunk ^ b_key[i]) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                } else {
                    f32x4 a[4], b[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = *(const f32x4*)(sa + a_off[i] + ((chunk ^ a_key[i]) << 4));
                        b[i] = *(const f32x4*)(sw + b_off[i] + ((chunk ^ b_key[i]) << 4));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][e], a[i][e], acc[i][j], 0, 0, 0);
                }
            }
        }

        // ---- epilogue of tile (m0,n0): vectors straight from the accumulators ----
        if (!((g.dbg & 1) && acc[0][0][0] != 12345.678f)) {
            const int col_w = n0 + wc * 64;
            if (g.fast_epi) {
                if constexpr (LP_OUT) {
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const int col = col_w + 32 * jp + 8 * grp4;
                        if (col >= g.N) continue;
                        f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f}, b1 = b0;
                        if (g.bias) { b0 = *(const f32x4*)(g.bias + col); b1 = *(const f32x4*)(g.bias + col + 4); }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = m0 + wr * 64 + i * 16 + l16;
                            if (row >= g.M) continue;
                            f32x4 v0 = acc[i][2 * jp] + b0, v1 = acc[i][2 * jp + 1] + b1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v0[e] = epi_act<true>(v0[e], g.act) * g.out_scale;
                                v1[e] = epi_act<true>(v1[e], g.act) * g.out_scale;
                            }
                            if (g.residual) {
                                const float* rp = g.residual + (size_t)row * g.ldr + col;
                                v0 += *(const f32x4*)rp; v1 += *(const f32x4*)(rp + 4);
                            }
                            *(bf16x8*)((bf16_t*)g.C + (size_t)row * ldc + col) = pack_bf16x8(v0, v1);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = col_w + 16 * j + 4 * grp4;
                        if (col >= g.N) continue;
                        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (g.bias) bv = *(const f32x4*)(g.bias + col);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = m0 + wr * 64 + i * 16 + l16;
                            if (row >= g.M) continue;
                            f32x4 v = acc[i][j] + bv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], g.act) * g.out_scale;
                            if (g.residual) v += *(const f32x4*)(g.residual + (size_t)row * g.ldr + col);
                            *(f32x4*)((float*)g.C + (size_t)row * ldc + col) = v;
                        }
                    }
                }
            } else {
                // generic fallback (N or a leading dimension not a multiple of 8 elements, e.g. the 2-logit head)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = col_w + wfrag_row<LP_OUT>(j, 4 * grp4 + r);
                        if (col >= g.N) continue;
                        const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = m0 + wr * 64 + i * 16 + l16;
                            if (row >= g.M) continue;
                            float v = apply_act(acc[i][j][r] + bv, g.act) * g.out_scale;
                            if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                            if (c_bf16) ((bf16_t*)g.C)[(size_t)row * ldc + col] = f32_to_bf16(v);
                            else ((float*)g.C)[(size_t)row * ldc + col] = v;
                        }
                    }
            }
        }
        slot = next_slot;
        if (slot >= nslots) break;
        m0 = ((slot / g.ntn) * 8 + xcd) * BM; n0 = (slot % g.ntn) * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace

extern "C" int madtp_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C,
                          int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                          int ab_dtype, int c_dtype, int act, float out_scale, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return MADTP_E_BADARG;
    if (ab_dtype != MADTP_F32 && ab_dtype != MADTP_BF16) return MADTP_E_DTYPE;
    if (c_dtype != MADTP_F32 && c_dtype != MADTP_BF16) return MADTP_E_DTYPE;
    const int esz = ab_dtype == MADTP_BF16 ? 2 : 4;
    if ((K * esz) % ROWB != 0) return MADTP_E_SHAPE;
    if (!aligned16(A) || !aligned16(W) || (lda * esz) % 16 || (ldw * esz) % 16) return MADTP_E_ALIGN;
    if (lda < K || ldw < K || ldc < N || (residual && ldr < N)) return MADTP_E_SHAPE;
    GemmArgs g;
    g.A = (const char*)A; g.W = (const char*)W; g.bias = bias; g.residual = residual; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.act = act; g.out_scale = out_scale;
    g.ldc = c_dtype == MADTP_BF16 ? -ldc : ldc;
    g.ntm = (M + BM - 1) / BM;
    g.ntn = (N + BN - 1) / BN;
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("MADTP_GEMM_DEBUG"); dbg = e ? atoi(e) : 0; }
    g.dbg = dbg;
    // vector epilogue needs 16-byte aligned rows on every epilogue operand
    g.fast_epi = (N % 8 == 0) && (ldc % 8 == 0) && aligned16(C) && (!bias || aligned16(bias)) &&
                 (!residual || (aligned16(residual) && ldr % 4 == 0));
    // persistent grid: at most 2 workgroups per CU (64 KiB LDS each), 64 per XCD
    const int slots_max = ((g.ntm + 7) / 8) * g.ntn;
    const int grid = 8 * (slots_max < 64 ? slots_max : 64);
    const size_t lds = 2 * STAGE_BYTES;
    hipStream_t s = (hipStream_t)stream;
    const bool lp = c_dtype == MADTP_BF16;
    if (ab_dtype == MADTP_BF16) {
        if (lp) hipLaunchKernelGGL((gemm_kernel<bf16_t, true>), dim3(grid), dim3(NTHREADS), lds, s, g);
        else hipLaunchKernelGGL((gemm_kernel<bf16_t, false>), dim3(grid), dim3(NTHREADS), lds, s, g);
    } else {
        if (lp) hipLaunchKernelGGL((gemm_kernel<float, true>), dim3(grid), dim3(NTHREADS), lds, s, g);
        else hipLaunchKernelGGL((gemm_kernel<float, false>), dim3(grid), dim3(NTHREADS), lds, s, g);
    }
    MADTP_LAUNCH_CHECK();
    return 0;
}
