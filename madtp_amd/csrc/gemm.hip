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

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;      // tile rows, bytes per row slab
constexpr int TILE_BYTES = BM * ROWB;              // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;        // A + W
constexpr int NTHREADS = 256;

struct GemmArgs {
    const char* A; const char* W; const float* bias; const float* residual; void* C;
    int M, N, K, lda, ldw, ldc, ldr, act, ntm, ntn;
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

template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs g) {
    constexpr int ESZ = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile map (see header comment)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tm = (slot / g.ntn) * 8 + xcd;
    const int tn = slot % g.ntn;
    if (tm >= g.ntm) return;
    const int m0 = tm * BM, n0 = tn * BN;

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

    // prologue: slab 0 -> stage 0
    stage_tile<ESZ>(g.A, m0, g.M - 1, g.lda, 0, smem, wave, lane);
    stage_tile<ESZ>(g.W, n0, n_pad_max, g.ldw, 0, smem + TILE_BYTES, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // slab kt landed (the barrier drains the LDS-DMA) and stage (kt+1)&1 is free again
        if (kt + 1 < nk) {
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            stage_tile<ESZ>(g.A, m0, g.M - 1, g.lda, (kt + 1) * ROWB, nxt, wave, lane);
            stage_tile<ESZ>(g.W, n0, n_pad_max, g.ldw, (kt + 1) * ROWB, nxt + TILE_BYTES, wave, lane);
        }
        const char* sa = smem + (kt & 1) * STAGE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + grp4;
            if constexpr (ESZ == 2) {
                bf16x8 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wr * 64 + i * 16 + l16;
                    a[i] = *(const bf16x8*)(sa + ra * ROWB + ((chunk ^ (ra & 7)) << 4));
                    const int rb = wc * 64 + i * 16 + l16;
                    b[i] = *(const bf16x8*)(sw + rb * ROWB + ((chunk ^ (rb & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
                f32x4 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wr * 64 + i * 16 + l16;
                    a[i] = *(const f32x4*)(sa + ra * ROWB + ((chunk ^ (ra & 7)) << 4));
                    const int rb = wc * 64 + i * 16 + l16;
                    b[i] = *(const f32x4*)(sw + rb * ROWB + ((chunk ^ (rb & 7)) << 4));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        }
    }

    // epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
    const bool c_bf16 = g.ldc < 0;  // sign bit of ldc carries the output dtype (see launcher)
    const int ldc = c_bf16 ? -g.ldc : g.ldc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wc * 64 + j * 16 + l16;
        if (col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 64 + i * 16 + grp4 * 4 + r;
                if (row >= g.M) continue;
                float v = apply_act(acc[i][j][r] + bv, g.act) * g.out_scale;
                if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                if (c_bf16) ((bf16_t*)g.C)[(size_t)row * ldc + col] = f32_to_bf16(v);
                else ((float*)g.C)[(size_t)row * ldc + col] = v;
            }
        }
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
    const int grid = ((g.ntm + 7) / 8) * 8 * g.ntn;
    const size_t lds = 2 * STAGE_BYTES;
    hipStream_t s = (hipStream_t)stream;
    if (ab_dtype == MADTP_BF16) {
        hipLaunchKernelGGL(gemm_kernel<bf16_t>, dim3(grid), dim3(NTHREADS), lds, s, g);
    } else {
        hipLaunchKernelGGL(gemm_kernel<float>, dim3(grid), dim3(NTHREADS), lds, s, g);
    }
    MADTP_LAUNCH_CHECK();
    return 0;
}
