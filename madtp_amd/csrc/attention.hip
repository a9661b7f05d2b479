// Multi-head attention core fused with the pruning-score reductions (k4 + k5, and k11 for
// cross-attention).  The reference materialises P[B,H,N,N] in HBM (vit.py:81-83, 238 MB/layer at B=128,N=197)
// and re-reads it for the head-max column mass (vit.py:126-127) and the CLS row (vit.py:96); here P never leaves
// registers.
//
// Work decomposition (gfx950, 64-lane waves):
//   * workgroup = (batch element b, 64 query rows); wave w owns the 16-row query tile 4*blockIdx.x + w and
//     loops over ALL heads, so the running head-max of P for its rows stays in registers.
//   * per head the workgroup stages K_h and V_h (Nk x 64) in LDS once (row pitch padded by 16 B), shared by the
//     four waves; Q rows are read straight from global/L2 into registers.
//   * S^T = K Q^T is computed "swapped" with v_mfma_f32_16x16x4_f32 so that lane (i = lane&15, g = lane>>4) ends
//     up holding key columns j = 16t + 4g + r of query row i: the softmax row reduction is in-lane plus two
//     cross-lane steps, and the same registers are exactly the A operand (row i, k-slot g) of the P.V MFMA.
//   * arithmetic is f32 throughout (exact-f32 MFMA); T is only the storage type of q/k/v/out (f32 or bf16).
//   * side outputs: colsum_part[b, row tile, j] (column mass of head-max P over this wave's rows, row 0 excluded),
//     p0[b,h,j] (CLS row), onorm[b,h,i] (||P V||_2 per row).  The tiny cross-tile / cross-head combination is
//     done by madtp_token_score in a fixed order (deterministic, no atomics).
// Roofline: MFMA f32 (157 TF) - 4*Nq*Nk*64 flops per (b,h); HBM traffic is q,k,v,out once (K/V re-reads by the
// other query tiles of the same b hit L2).
#include "common.h"
#include "internal.h"
#include <map>
#include <mutex>
#include <stdlib.h>

namespace {

struct AttnArgs {
    const char* q; const char* k; const char* v; char* out; const float* mask;
    float* colsum; float* p0; float* onorm;
    int B, H, Nq, Nk, ldq, ldk, ldv, ldo, nrt;
    float scale;
    const int* kvidx;  // optional: sample b reads K/V block kvidx[b] (cross-attention against a cache of encoder K/V)
    // madtp_attention_pair (attn_bf16_kernel without scores only): a second problem of identical shape, blockIdx.y in [B, 2B)
    const char* q2; const char* k2; const char* v2; char* out2; const float* mask2; int pair;
    // optional additive [Nq, Nk] mask shared by all samples and heads (CLIP's causal text mask, clip/mock.py:309-310): element
    // (i, j) at mask_qk[i * ld_mqk + j]; folded into the per-lane key mask (a lane owns one query row)
    const float* mask_qk; int ld_mqk;
    // attn_bf16_large_kernel with scores and gridDim.z == 2 (two head halves per row block): exchange area of the head-max
    // [B * row blocks][2][4 waves][2 NT dwords][64 lanes] and one ticket per (row block, wave)
    unsigned* hm_ws; int* hm_tick;
    // sync-free encoder path (self-attention): Nq = Nk = *n_dev tokens per sample, read by the kernel; the launch geometry and the
    // key-tile instantiation are the host's worst case (the unpruned sequence)
    int kvb;        // rows from one sample's K/V block to the next (= Nk for dense [B * Nk, ld] operands; the decoder's self-attention
                    // cache [rows, Lmax, 2 dim] passes Lmax: a step attends to the first Nk positions of every row's block)
    float* o_part;  // attn_bf16_large_kernel<.., HV = 2>: f32 scratch [B, Nq, H, 64] (the first key half's P.V sums)
    const int32_t* n_dev;
    int dev_q_only;  // cross-attention on the sync-free path: *n_dev is the number of QUERY tokens per sample, Nk stays the host's
    // f16x3 mode (attn_f16s_kernel / attn_large_f16s_kernel), round 6: split_dim > 0 = the context leaves the kernel AS the f16-split
    // operand planes of the GEMM that consumes it - out is then _Float16 [rows, ldo] (ldo in f16 elements), P0 = f16(o) at column c,
    // P1 = f16((o - P0) 2^11) at column split_dim + c (common.h) - instead of f32 that a separate split_f16 launch re-reads (49
    // launches per forward); range_flag: the f16 range flag the split raises (madtp_range_status)
    int split_dim; int* range_flag;
};
#define ATTN_DEV_DIMS(a)                                   \
    if ((a).n_dev) {                                       \
        (a).Nq = *(a).n_dev;                               \
        if (!(a).dev_q_only) { (a).Nk = (a).Nq; (a).kvb = (a).Nq; } \
        (a).nrt = ((a).Nq + 15) / 16;                      \
    }

// Exchange area of the head split (one per device and stream, allocated on first use; the tickets reset themselves).
struct HmWorkspace { unsigned* ws; int* tick; };
constexpr int HM_MAX_WGS = 384, HM_MAX_NT = 40, HM_MAX_GZ = 4;  // (attn_bf16_large_kernel splits the heads over up to 4 workgroups)
static bool hm_workspace(hipStream_t s, HmWorkspace& out) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("MADTP_ATTN_HEAD_SPLIT"); env = e ? atoi(e) : 1; }
    if (!env) return false;
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, HmWorkspace> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    auto it = pool.find({dev, s});
    if (it == pool.end()) {
        constexpr size_t WS_BYTES = (size_t)HM_MAX_WGS * HM_MAX_GZ * 4 * (2 * HM_MAX_NT) * 64 * 4, TICK_BYTES = (size_t)HM_MAX_WGS * 4 * sizeof(int);
        char* base = nullptr;
        if (hipMalloc((void**)&base, WS_BYTES + TICK_BYTES) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipMemsetAsync(base + WS_BYTES, 0, TICK_BYTES, s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(base); return false; }
        it = pool.emplace(std::make_pair(dev, s), HmWorkspace{(unsigned*)base, (int*)(base + WS_BYTES)}).first;
    }
    out = it->second;
    return true;
}

// 16x16x32 MFMA of the fast modes' 2-byte operands: bf16 (MADTP_BF16) or IEEE f16 (MADTP_F16), f32 accumulate
template <bool F16>
__device__ __forceinline__ f32x4 mfma_lp(bf16x8 a, bf16x8 b, f32x4 c, int, int, int) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ f32x4 load4(const char* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const char* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const char* p) {
    const uint2 u = *(const uint2*)p;
    f32x4 r;
    r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
template <typename T> __device__ __forceinline__ float load1(const char* p);
template <> __device__ __forceinline__ float load1<float>(const char* p) { return *(const float*)p; }
template <> __device__ __forceinline__ float load1<bf16_t>(const char* p) { return bf16_to_f32(*(const bf16_t*)p); }

template <typename T, int NT, bool SCORES>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
    ATTN_DEV_DIMS(a)
    constexpr int ESZ = sizeof(T);
    constexpr int RB = 64 * ESZ + 16;        // LDS row pitch in bytes (pad one 16-B slot)
    constexpr int CPR = 64 * ESZ / 16;       // 16-byte chunks per row
    constexpr int NKP = NT * 16;             // padded key count
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NKP * RB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int bkv = a.kvidx ? a.kvidx[b] : b;  // K/V block of this sample
    const int rt = blockIdx.x * 4 + wave;    // 16-row query tile of this wave
    const int i0 = rt * 16;
    const bool active = i0 < a.Nq;
    const int irow = min(i0 + l16, a.Nq - 1);  // clamped query row for loads

    f32x4 pmax[NT];
    if constexpr (SCORES) {
#pragma unroll
        for (int t = 0; t < NT; ++t) pmax[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // additive key mask of this lane's columns, loaded ONCE (it does not depend on the head): 0 without a mask, -inf for the
    // padding columns j >= Nk.  (Inside the head loop these were 4*NT predicated scalar loads, each with its own wait.)
    f32x4 mk[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r;
            mk[t][r] = j < a.Nk ? (a.mask ? a.mask[(size_t)b * a.Nk + j] : 0.f) : -INFINITY;
            if (a.mask_qk && j < a.Nk) mk[t][r] += a.mask_qk[(size_t)irow * a.ld_mqk + j];
        }

    // K_h / V_h go global -> registers -> LDS.  Up to 8 x 16 B per operand and thread (NT <= 8 in f32) the registers of head
    // h + 1 are fetched right after head h's image is complete, so the loads fly under head h's MFMAs instead of sitting,
    // with their full latency, between two barriers of every head (the arithmetic is untouched: bit-identical results).
    constexpr int NLD = (NKP * CPR + 255) / 256;
    constexpr bool PREFETCH = NLD <= 8;
    uint4 kreg[PREFETCH ? NLD : 1], vreg[PREFETCH ? NLD : 1];
    auto fetch = [&](int h) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPR, c = idx % CPR;
            kreg[i] = make_uint4(0, 0, 0, 0);
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (idx < NKP * CPR && row < a.Nk) {
                const size_t grow = (size_t)bkv * a.kvb + row;
                kreg[i] = *(const uint4*)(a.k + (grow * a.ldk + h * 64) * ESZ + c * 16);
                vreg[i] = *(const uint4*)(a.v + (grow * a.ldv + h * 64) * ESZ + c * 16);
            }
        }
    };
    if constexpr (PREFETCH) {
        if ((int)blockIdx.z < a.H) fetch(blockIdx.z);
    }
    for (int h = blockIdx.z; h < a.H; h += gridDim.z) {
        __syncthreads();
        // ---- stage K_h, V_h (zero rows beyond Nk: 0 * finite stays 0 in P.V) ----
        if constexpr (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = tid + 256 * i;
                const int row = idx / CPR, c = idx % CPR;
                if (idx < NKP * CPR) {
                    *(uint4*)(Ks + row * RB + c * 16) = kreg[i];
                    *(uint4*)(Vs + row * RB + c * 16) = vreg[i];
                }
            }
        } else {
            for (int idx = tid; idx < NKP * CPR; idx += 256) {
                const int row = idx / CPR, c = idx % CPR;
                uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
                if (row < a.Nk) {
                    const size_t grow = (size_t)bkv * a.kvb + row;
                    kv = *(const uint4*)(a.k + (grow * a.ldk + h * 64) * ESZ + c * 16);
                    vv = *(const uint4*)(a.v + (grow * a.ldv + h * 64) * ESZ + c * 16);
                }
                *(uint4*)(Ks + row * RB + c * 16) = kv;
                *(uint4*)(Vs + row * RB + c * 16) = vv;
            }
        }
        __syncthreads();
        if constexpr (PREFETCH) {
            if (h + (int)gridDim.z < a.H) fetch(h + gridDim.z);
        }
        if (!active) continue;

        // ---- Q fragment: row i, d = 4*(4s+g)+e ----
        f32x4 q[4];
        {
            const char* qp = a.q + (((size_t)b * a.Nq + irow) * a.ldq + h * 64) * ESZ;
#pragma unroll
            for (int s = 0; s < 4; ++s) q[s] = load4<T>(qp + (4 * s + g) * 4 * ESZ);
        }

        // ---- S^T = K Q^T ----
        f32x4 sc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 kf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) kf[t] = load4<T>(Ks + (16 * t + l16) * RB + (4 * s + g) * 4 * ESZ);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t][e], q[s][e], sc[t], 0, 0, 0);
        }

        // ---- softmax over keys (lane holds j = 16t+4g+r of row i) ----
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // parity mode: multiply, then add, each rounded (the reference scales, then adds the mask)
                const float v = __fadd_rn(__fmul_rn(sc[t][r], a.scale), mk[t][r]);
                sc[t][r] = v;
                m = fmaxf(m, v);
            }
        m = rows4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = expf(sc[t][r] - m);
                sc[t][r] = p;
                sum += p;
            }
        sum = rows4_sum(sum);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = sc[t][r] / sum;
                sc[t][r] = p;
                if constexpr (SCORES) pmax[t][r] = fmaxf(pmax[t][r], p);
            }
        if constexpr (SCORES) {
            if (i0 + l16 == 0) {  // CLS row of this batch element
                float* dst = a.p0 + ((size_t)b * a.H + h) * a.Nk;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * t + 4 * g + r;
                        if (j < a.Nk) dst[j] = sc[t][r];
                    }
            }
        }

        // ---- O = P V : A = P (row i, k-slot g <-> j = 16t+4g+r), B = V[j][d] ----
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const char* vrow = Vs + (16 * t + 4 * g + r) * RB + l16 * ESZ;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[t][r], load1<T>(vrow + dt * 16 * ESZ), o[dt], 0, 0, 0);
            }

        // ---- write O (row = i0+4g+r, col = h*64+dt*16+l16), row norms ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * g + r;
            float n2 = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) n2 += o[dt][r] * o[dt][r];
            if constexpr (SCORES) n2 = row16_sum(n2);
            if (i < a.Nq) {
                T* orow = (T*)(a.out + (((size_t)b * a.Nq + i) * a.ldo + h * 64) * ESZ);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) orow[dt * 16 + l16] = from_f32<T>(o[dt][r]);
                if constexpr (SCORES)
                    if (l16 == 0) a.onorm[((size_t)b * a.H + h) * a.Nq + i] = sqrtf(n2);
            }
        }
    }

    if constexpr (SCORES) {
        if (gridDim.z == 2) {
            // two workgroups per row block, half of the heads each (launch_attn): the head-max halves are merged by the wave that
            // arrives second - a max of the same f32 values, so the column sums are bit-identical to the one-workgroup launch
            // (protocol: attn_bf16_large_kernel)
            const size_t slot = ((size_t)b * gridDim.x + blockIdx.x) * 4 + wave;
            unsigned* mine = a.hm_ws + ((slot * 2 + blockIdx.z) * (2 * HM_MAX_NT)) * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(mine + (4 * t + r) * 64, __float_as_uint(pmax[t][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(a.hm_tick + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == 0) return;
            if (lane == 0) __hip_atomic_store(a.hm_tick + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned* theirs = a.hm_ws + ((slot * 2 + (1 - blockIdx.z)) * (2 * HM_MAX_NT)) * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pmax[t][r] = fmaxf(pmax[t][r], __uint_as_float(__hip_atomic_load(theirs + (4 * t + r) * 64, __ATOMIC_RELAXED,
                                                                                      __HIP_MEMORY_SCOPE_AGENT)));
        }
        if (active) {
            const int i = i0 + l16;
            const bool valid = i >= 1 && i < a.Nq;
            float* dst = a.colsum + ((size_t)b * a.nrt + rt) * a.Nk;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(valid ? pmax[t][r] : 0.f);
                    const int j = 16 * t + 4 * g + r;
                    if (l16 == 0 && j < a.Nk) dst[j] = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// f16-split ("f16x3") flavour of the kernel above for the fp32-ACCURATE precision mode: same decomposition, f32 storage of
// q / k / v / out, the same softmax arithmetic (separately rounded scale and mask, expf, true division) and the same score
// side outputs - but both products run on the f16 MFMA as THREE products of f16-split operands (common.h):
//     x = P0 + 2^-11 P1  (P0 = f16(x), P1 = f16((x - P0) 2^11))     x y = P0 Q0 + 2^-11 (P0 Q1 + P1 Q0)   (+ 2^-22 P1 Q1, dropped)
// with exact partial products and f32 accumulation in two accumulators (hi: P0 Q0, lo: the two cross terms): the rounding
// class of an f32 dot product at 3/16 of the exact-f32 MFMA's cost (v_mfma_f32_16x16x4_f32: 32 cycles per 16x16x4 step; the
// f16 instruction does a 16x16x32 step in 16).  K_h and V_h are split WHILE they are staged (global f32 -> registers ->
// two f16 planes in LDS, row pitch 160 B: conflict-free ds_read_b128 of the K fragments), Q and P are split in registers.
// Register / operand layouts are those of attn_bf16_kernel: S^T = K Q^T (lane (i = l16, g) holds keys 16t + 4g + r of query
// row i), P V with the k-slot <-> key permutation key(chunk c, g, e) = 32c + 16(e >> 2) + 4g + (e & 3) so that the A operand is
// the lane's own eight probabilities and the B operand two transpose reads (ds_read_b64_tr_b16) of row-major V.
template <int NT, bool SCORES>
__global__ __launch_bounds__(256) void attn_f16s_kernel(AttnArgs a) {
    ATTN_DEV_DIMS(a)
    constexpr int PITCH = NT <= 15 ? 160 : 144;  // bytes per 64-element f16 row (144: 2-way conflicts, what fits at 256 keys)
    constexpr int NKP = NT * 16;                 // padded key count
    constexpr int NC = (NT + 1) / 2;             // 32-key chunks of the P.V product
    constexpr int VR = NC * 32;                  // V rows (whole chunks; rows >= NKP stay zero)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* K0 = smem;
    char* K1 = K0 + NKP * PITCH;
    char* V0 = K1 + NKP * PITCH;
    char* V1 = V0 + VR * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int bkv = a.kvidx ? a.kvidx[b] : b;
    const int rt = blockIdx.x * 4 + wave;
    const int i0 = rt * 16;
    const bool active = i0 < a.Nq;
    const int irow = min(i0 + l16, a.Nq - 1);

    f32x4 pmax[NT];
    if constexpr (SCORES) {
#pragma unroll
        for (int t = 0; t < NT; ++t) pmax[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 mk[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r;
            mk[t][r] = j < a.Nk ? (a.mask ? a.mask[(size_t)b * a.Nk + j] : 0.f) : -INFINITY;
            if (a.mask_qk && j < a.Nk) mk[t][r] += a.mask_qk[(size_t)irow * a.ld_mqk + j];
        }
    if constexpr (VR > NKP) {  // the last chunk's missing key tile: zero V rows (their probabilities are zero as well), once
        for (int idx = tid; idx < (VR - NKP) * (PITCH / 16); idx += 256) {
            *(uint4*)(V0 + NKP * PITCH + idx * 16) = make_uint4(0, 0, 0, 0);
            *(uint4*)(V1 + NKP * PITCH + idx * 16) = make_uint4(0, 0, 0, 0);
        }
    }

    // staging: thread chunk idx = (key row, 4 consecutive d) of K_h and V_h as f32, split into the two planes on the way to LDS;
    // up to 8 chunks per operand the registers of head h + 1 are fetched under head h's MFMAs (as in attn_kernel)
    constexpr int NLD = NT;  // NKP * 16 chunks / 256 threads
    constexpr bool PREFETCH = NLD <= 8;
    uint4 kreg[PREFETCH ? NLD : 1], vreg[PREFETCH ? NLD : 1];
    auto fetch = [&](int h) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c = idx & 15;
            kreg[i] = make_uint4(0, 0, 0, 0);
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (row < a.Nk) {
                const size_t grow = (size_t)bkv * a.kvb + row;
                kreg[i] = *(const uint4*)(a.k + (grow * a.ldk + h * 64) * 4 + c * 16);
                vreg[i] = *(const uint4*)(a.v + (grow * a.ldv + h * 64) * 4 + c * 16);
            }
        }
    };
    auto put = [&](int idx, uint4 kv, uint4 vv) {
        const int row = idx >> 4, c = idx & 15;
        f16x4 h4, l4;
        split_f16x4(__builtin_bit_cast(f32x4, kv), h4, l4);
        *(f16x4*)(K0 + row * PITCH + c * 8) = h4;
        *(f16x4*)(K1 + row * PITCH + c * 8) = l4;
        split_f16x4(__builtin_bit_cast(f32x4, vv), h4, l4);
        *(f16x4*)(V0 + row * PITCH + c * 8) = h4;
        *(f16x4*)(V1 + row * PITCH + c * 8) = l4;
    };
    if constexpr (PREFETCH) {
        if ((int)blockIdx.z < a.H) fetch(blockIdx.z);
    }
    constexpr float LO = 1.0f / F16S_LO_SCALE;
    for (int h = blockIdx.z; h < a.H; h += gridDim.z) {
        __syncthreads();
        if constexpr (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) put(tid + 256 * i, kreg[i], vreg[i]);
        } else {
            for (int idx = tid; idx < NKP * 16; idx += 256) {
                const int row = idx >> 4, c = idx & 15;
                uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
                if (row < a.Nk) {
                    const size_t grow = (size_t)bkv * a.kvb + row;
                    kv = *(const uint4*)(a.k + (grow * a.ldk + h * 64) * 4 + c * 16);
                    vv = *(const uint4*)(a.v + (grow * a.ldv + h * 64) * 4 + c * 16);
                }
                put(idx, kv, vv);
            }
        }
        __syncthreads();
        if constexpr (PREFETCH) {
            if (h + (int)gridDim.z < a.H) fetch(h + gridDim.z);
        }
        if (!active) continue;

        // ---- Q fragment (B operand of S^T = K Q^T): row i, k-slots of group g <-> d = 32 kk + 8g .. +7, split in registers ----
        f16x8 qh[2], ql[2];
        {
            const char* qp = a.q + (((size_t)b * a.Nq + irow) * a.ldq + h * 64) * 4 + g * 32;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 p0, p1;
                split_f16x8(*(const f32x4*)(qp + kk * 128), *(const f32x4*)(qp + kk * 128 + 16), p0, p1);
                qh[kk] = __builtin_bit_cast(f16x8, p0);
                ql[kk] = __builtin_bit_cast(f16x8, p1);
            }
        }

        // ---- S^T = K Q^T: hi = K0 Q0, lo = K0 Q1 + K1 Q0, S = hi + 2^-11 lo ----
        f32x4 sc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int off = (16 * t + l16) * PITCH + g * 16;
            const f16x8 kh0 = *(const f16x8*)(K0 + off), kh1 = *(const f16x8*)(K0 + off + 64);
            const f16x8 kl0 = *(const f16x8*)(K1 + off), kl1 = *(const f16x8*)(K1 + off + 64);
            const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, qh[0], z, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, qh[1], hi, 0, 0, 0);
            f32x4 lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, ql[0], z, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, ql[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl0, qh[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl1, qh[1], lo, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t][r] = fmaf(lo[r], LO, hi[r]);
        }

        // ---- softmax over keys: the parity arithmetic of attn_kernel ----
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = __fadd_rn(__fmul_rn(sc[t][r], a.scale), mk[t][r]);
                sc[t][r] = v;
                m = fmaxf(m, v);
            }
        m = rows4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = expf(sc[t][r] - m);
                sc[t][r] = p;
                sum += p;
            }
        sum = rows4_sum(sum);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = sc[t][r] / sum;
                sc[t][r] = p;
                if constexpr (SCORES) pmax[t][r] = fmaxf(pmax[t][r], p);
            }
        if constexpr (SCORES) {
            if (i0 == 0 && l16 == 0) {  // CLS row of this batch element
                float* dst = a.p0 + ((size_t)b * a.H + h) * a.Nk;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * t + 4 * g + r;
                        if (j < a.Nk) dst[j] = sc[t][r];
                    }
            }
        }

        // ---- O^T = V^T P^T: hi = V0 P0, lo = V0 P1 + V1 P0 ----
        f32x4 oh[4], ol[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { oh[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ol[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 t1 = (2 * c + 1 < NT) ? sc[2 * c + 1 < NT ? 2 * c + 1 : 0] : (f32x4){0.f, 0.f, 0.f, 0.f};
            u32x4 pp0, pp1;
            split_f16x8(sc[2 * c], t1, pp0, pp1);
            const f16x8 ph = __builtin_bit_cast(f16x8, pp0), pl = __builtin_bit_cast(f16x8, pp1);
            // transpose read: lane 4r + q of a 16-lane group addresses (key row r, columns 4q .. 4q+3) and receives column l16
            const int voff = (32 * c + 4 * g + (l16 >> 2)) * PITCH + 8 * (l16 & 3);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vh = cat_bf16x4(lds_read_tr16(V0 + voff + dt * 32), lds_read_tr16(V0 + voff + 16 * PITCH + dt * 32));
                const bf16x8 vl = cat_bf16x4(lds_read_tr16(V1 + voff + dt * 32), lds_read_tr16(V1 + voff + 16 * PITCH + dt * 32));
                oh[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), ph, oh[dt], 0, 0, 0);
                ol[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), pl, ol[dt], 0, 0, 0);
                ol[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vl), ph, ol[dt], 0, 0, 0);
            }
        }

        // ---- write O: lane (i = l16, g) holds columns h*64 + 16 dt + 4g .. +3 of row i (16-byte stores), row norms ----
        {
            const int i = i0 + l16;
            f32x4 o[4];
            float n2 = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[dt][r] = fmaf(ol[dt][r], LO, oh[dt][r]);
                    n2 += o[dt][r] * o[dt][r];
                }
            if constexpr (SCORES) n2 = rows4_sum(n2);
            if (i < a.Nq) {
                if (a.split_dim) {  // straight into the consumer GEMM's operand planes (8-byte stores per plane)
                    _Float16* prow = (_Float16*)a.out + ((size_t)b * a.Nq + i) * a.ldo + h * 64 + 4 * g;
                    bool bad = false;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        f16x4 ph_, pl_;
                        split_f16x4(o[dt], ph_, pl_);
                        *(f16x4*)(prow + dt * 16) = ph_;
                        *(f16x4*)(prow + a.split_dim + dt * 16) = pl_;
                        bad |= f16_range_bad(o[dt]);
                    }
                    f16_range_raise(a.range_flag, bad);
                } else {
                    float* orow = (float*)(a.out + (((size_t)b * a.Nq + i) * a.ldo + h * 64) * 4) + 4 * g;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) *(f32x4*)(orow + dt * 16) = o[dt];
                }
                if constexpr (SCORES)
                    if (g == 0) a.onorm[((size_t)b * a.H + h) * a.Nq + i] = sqrtf(n2);
            }
        }
    }

    if constexpr (SCORES) {
        if (gridDim.z == 2) {  // two workgroups per row block, half of the heads each: exact head-max merge (attn_kernel)
            const size_t slot = ((size_t)b * gridDim.x + blockIdx.x) * 4 + wave;
            unsigned* mine = a.hm_ws + ((slot * 2 + blockIdx.z) * (2 * HM_MAX_NT)) * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(mine + (4 * t + r) * 64, __float_as_uint(pmax[t][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(a.hm_tick + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == 0) return;
            if (lane == 0) __hip_atomic_store(a.hm_tick + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned* theirs = a.hm_ws + ((slot * 2 + (1 - blockIdx.z)) * (2 * HM_MAX_NT)) * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pmax[t][r] = fmaxf(pmax[t][r], __uint_as_float(__hip_atomic_load(theirs + (4 * t + r) * 64, __ATOMIC_RELAXED,
                                                                                      __HIP_MEMORY_SCOPE_AGENT)));
        }
        if (active) {
            const int i = i0 + l16;
            const bool valid = i >= 1 && i < a.Nq;
            float* dst = a.colsum + ((size_t)b * a.nrt + rt) * a.Nk;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(valid ? pmax[t][r] : 0.f);
                    const int j = 16 * t + 4 * g + r;
                    if (l16 == 0 && j < a.Nk) dst[j] = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 fast-mode kernel: same decomposition and register layouts, but both products run on the bf16 MFMA
// (v_mfma_f32_16x16x32_bf16, f32 accumulate); softmax and all score reductions stay f32.
//   * K_h is DMA'd into LDS as 128-byte rows with the 16-B chunk index XOR (row&7) (swizzle on the source address),
//     so the K fragment reads (ds_read_b128) are conflict-free.
//   * The P.V product needs V with the KEY index running along the MFMA k slots.  V_h is staged ROW-major (128-byte
//     rows, LDS-DMA like K) and the B operand is fetched with the hardware transpose read ds_read_b64_tr_b16
//     (4 keys x 16 columns per 16 lanes).
//     The k-slot <-> key mapping of the MFMA is a free permutation; with key(chunk c, group g, slot e) =
//     32c + 16(e>>2) + 4g + (e&3) the A operand is exactly the lane's own eight probabilities (tiles 2c, 2c+1) -
//     no cross-lane traffic - and the B operand is two transpose reads (keys 32c+4g.. and 32c+16+4g..).
//   * exp via v_exp_f32 (__expf) and one reciprocal per row: this is the fast mode; the f32 kernel above keeps
//     expf/division for the parity mode.
//   * HS = 2 (scores variant, <= 128 keys): the workgroup is 8 waves = 4 query-row tiles x 2 HEAD PARITIES.  With 64 rows
//     per workgroup a 128-image batch of ~90-token sequences is only 256 workgroups - one wave per SIMD, every LDS /
//     exp / cross-lane latency of the per-head chain exposed.  The two parity groups walk heads h = 0,2,4.. and 1,3,5..
//     with their own K/V rings, two independent instruction streams per SIMD, and merge their head-max once at the end.
//   * RB = 2 (scores variant, > 128 keys, HS = 1): the workgroup is 8 waves = TWO 64-row blocks on ONE K/V ring.  At 134 /
//     197 keys a ring is 82 / 110 KiB, so only one workgroup fits a CU: with 64-row workgroups that is 4 waves per CU, 3-4
//     workgroups per sample each re-reading the sample's K and V (310 MB at 197 tokens).  128 rows per workgroup halve
//     the K/V traffic and put 8 waves on the CU.
#ifdef MADTP_TS_TIMING
__device__ long long g_attn_dbg[8];
#define AT_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && it == 2) { __builtin_amdgcn_s_waitcnt(0); g_attn_dbg[i] = wall_clock64(); } } while (0)
#else
#define AT_MARK(i)
#endif
template <int NT, bool SCORES, int HS, int RB = 1, bool F16 = false>
__global__ __launch_bounds__(256 * HS * RB, (HS == 1 && RB == 1) ? 2 : 1) void attn_bf16_kernel(AttnArgs a) {
    ATTN_DEV_DIMS(a)
    static_assert(HS == 1 || RB == 1, "head-parity split and two row blocks are alternatives");
    constexpr int NKP = NT * 16;
    constexpr int NC = (NT + 1) / 2;          // 32-key chunks
    constexpr int VR = NC * 32;               // staged V rows (whole chunks)
    constexpr int STAGE = (NKP + VR) * 128;   // bytes per head: K image then V image, 128-byte rows
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;                    // query-row tile within its 64-row block
    const int hp = HS == 2 ? wave8 >> 2 : 0;       // head parity group (HS = 2)
    const int rb = RB == 2 ? wave8 >> 2 : 0;       // 64-row block within the workgroup (RB = 2)
    constexpr int NSW = 4 * RB;                    // waves that stage one ring
    const int swave = RB == 2 ? wave8 : wave;      // this wave's index among them
    char* const ring = smem + hp * 2 * STAGE;
    const int l16 = lane & 15, g = lane >> 4;
    int b = blockIdx.y;
    if (a.pair && b >= a.B) {  // second problem of a pair launch (the twin cross-attention branches of an NLVR text layer)
        b -= a.B;
        a.q = a.q2; a.k = a.k2; a.v = a.v2; a.out = a.out2; a.mask = a.mask2;
    }
    const int bkv = a.kvidx ? a.kvidx[b] : b;  // K/V block of this sample
    const int rt = (blockIdx.x * RB + rb) * 4 + wave;
    const int i0 = rt * 16;
    const bool active = i0 < a.Nq;
    const int irow = min(i0 + l16, a.Nq - 1);

    f32x4 pmax[NT];
    if constexpr (SCORES) {
#pragma unroll
        for (int t = 0; t < NT; ++t) pmax[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 mk[NT];  // additive key mask of this lane's columns (0 without a mask, -inf for j >= Nk), loaded once
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r;
            mk[t][r] = j < a.Nk ? (a.mask ? a.mask[(size_t)b * a.Nk + j] : 0.f) : -INFINITY;
            if (a.mask_qk && j < a.Nk) mk[t][r] += a.mask_qk[(size_t)irow * a.ld_mqk + j];
            mk[t][r] *= 1.44269504088896341f;  // the softmax runs in log2 units: exp2(c2 s + mk - max), no multiply inside expf
        }
    const float c2 = a.scale * 1.44269504088896341f;

    // K_h and V_h are LDS-DMA'd (8 rows = 1 KiB per wave-instruction) into a 2-stage ring over the heads: the DMA of
    // head h+1 is in flight while head h is computed.  Swizzles live on the SOURCE address (the DMA destination is
    // lane-linear): K chunk ^= row&7 (conflict-free ds_read_b128 of 16 rows); V chunk ^= 2*((row>>1)&3), which puts
    // the 8 key rows of a transpose read (ds_read_b64_tr_b16) on 8 disjoint bank octets.  Rows >= Nk are clamped
    // to a valid row: their probabilities are exactly 0, so any finite data is fine.
    const int sub = lane >> 3, pos = lane & 7;
    // per-lane source of every DMA instruction of this wave for head 0, computed ONCE (the 64-bit row arithmetic used to be
    // redone for each of the 6-14 instructions of every head); head h adds 128 bytes
    constexpr int NDMA = ((NKP + VR) / 8 + NSW - 1) / NSW;
    const char* srcb[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int grp = swave + NSW * i;
        const bool is_v = grp >= NKP / 8;
        int row = (is_v ? grp - NKP / 8 : grp) * 8 + sub;
        const int chunk = is_v ? (pos ^ (((row >> 1) & 3) << 1)) : (pos ^ (row & 7));
        row = row < a.Nk ? row : a.Nk - 1;
        srcb[i] = (is_v ? a.v + ((size_t)bkv * a.kvb + row) * a.ldv * 2 : a.k + ((size_t)bkv * a.kvb + row) * a.ldk * 2) + chunk * 16;
    }
    auto stage_head = [&](int h, int st) {
        char* base = ring + st * STAGE;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int grp = swave + NSW * i;
            if (grp < (NKP + VR) / 8)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(srcb[i] + h * 128), LDS_PTR(base + grp * 1024), 16, 0, 0);
        }
    };
    // Q fragment (B operand of S^T = K Q^T): row i, k-slot group g <-> d = 32kk + 8g .. +7; fetched one head ahead
    const char* qbase = a.q + ((size_t)b * a.Nq + irow) * a.ldq * 2 + g * 16;
    auto load_q = [&](int h, bf16x8 (&qq)[2]) {
        const char* qp = qbase + h * 128;
        qq[0] = *(const bf16x8*)qp;
        qq[1] = *(const bf16x8*)(qp + 64);
    };
    int st = 0;
    const int hstep = gridDim.z * HS;
    const int hfirst = blockIdx.z * HS + hp;
    const int niter = (a.H - (int)blockIdx.z * HS + hstep - 1) / hstep;  // the same for both parity groups (barriers)
    bf16x8 qn[2];
    if (hfirst < a.H) {
        stage_head(hfirst, 0);
        load_q(hfirst, qn);
    }
    for (int it = 0; it < niter; ++it, st ^= 1) {
        const int h = hfirst + it * hstep;
        bf16x8 q[2] = {qn[0], qn[1]};
        AT_MARK(0);
        __syncthreads();  // head h landed (the barrier drains the DMA); the other stage is free again
        AT_MARK(1);
        if (h + hstep < a.H) {
            stage_head(h + hstep, st ^ 1);
            load_q(h + hstep, qn);
        }
        if (!active || h >= a.H) continue;
        AT_MARK(2);
        const char* Ks = ring + st * STAGE;
        const char* Vs = Ks + NKP * 128;

        // ---- S^T = K Q^T ----
        f32x4 sc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = 16 * t + l16;
            const bf16x8 k0 = *(const bf16x8*)(Ks + row * 128 + (((0 + g) ^ (row & 7)) << 4));
            const bf16x8 k1 = *(const bf16x8*)(Ks + row * 128 + (((4 + g) ^ (row & 7)) << 4));
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = mfma_lp<F16>(k0, q[0], acc, 0, 0, 0);
            sc[t] = mfma_lp<F16>(k1, q[1], acc, 0, 0, 0);
        }
        AT_MARK(3);
        // ---- softmax over keys (lane holds j = 16t+4g+r of row i) ----
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fmaf(sc[t][r], c2, mk[t][r]);
                sc[t][r] = v;
                m = fmaxf(m, v);
            }
        m = rows4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc[t][r] - m);
                sc[t][r] = p;
                sum += p;
            }
        sum = rows4_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sc[t] *= inv;
            if constexpr (SCORES) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pmax[t][r] = fmaxf(pmax[t][r], sc[t][r]);
            }
        }
        if constexpr (SCORES) {
            if (i0 == 0) {  // wave-uniform: only the wave that owns query row 0 enters (the other waves skip 4*NT branches)
                if (l16 == 0) {
                    float* dst = a.p0 + ((size_t)b * a.H + h) * a.Nk;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = 16 * t + 4 * g + r;
                            if (j < a.Nk) dst[j] = sc[t][r];
                        }
                }
            }
        }
        AT_MARK(4);
        // ---- O = P V on the bf16 MFMA ----
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 hi = (2 * c + 1 < NT) ? sc[2 * c + 1 < NT ? 2 * c + 1 : 0] : (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16x8 pa = pack_lp8<F16>(sc[2 * c], hi);
            // transpose read: lane 4r+q of the group addresses (key row r, columns 4q..4q+3), receives column l16.
            // key rows of this lane group: 32c + 4g + r (first read) and +16 (second); both have the same (row>>1)&3.
            // (The compiler puts s_waitcnt vmcnt(0) in front of the first of these builtin reads - it cannot tell them from the
            //  LDS-DMA targets - so P.V of head h waits for the K/V of head h+1 issued at the top of the iteration.  Inline-asm
            //  reads as in attn_bf16_large_kernel remove the wait; measured neutral here (29.7 vs 28.7 us, the DMA has landed
            //  by then) and the NT = 16 instantiation spills with the extra buffer, so the builtin stays.)
            const int vrow = 32 * c + 4 * g + (l16 >> 2);
            const int vkey = ((vrow >> 1) & 3) << 1;
            const char* vr = Vs + vrow * 128 + 8 * (l16 & 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int p = ((2 * dt + ((l16 & 3) >> 1)) ^ vkey) << 4;
                const bf16x8 vb = cat_bf16x4(lds_read_tr16(vr + p), lds_read_tr16(vr + 16 * 128 + p));
                o[dt] = mfma_lp<F16>(vb, pa, o[dt], 0, 0, 0);  // O^T: row d = 4g+r, col i
            }
        }
        AT_MARK(5);
        // ---- write O: with the operands swapped lane (i = l16, g) holds columns h*64 + 16dt + 4g .. +3 of row i ----
        {
            const int i = i0 + l16;
            float n2 = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) n2 += o[dt][r] * o[dt][r];
            if constexpr (SCORES) {
                n2 = rows4_sum(n2);
            }
            if (i < a.Nq) {
                bf16_t* orow = (bf16_t*)(a.out + (((size_t)b * a.Nq + i) * a.ldo + h * 64) * 2) + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(orow + dt * 16) = pack_lp4<F16>(o[dt]);
                if constexpr (SCORES)
                    if (g == 0) a.onorm[((size_t)b * a.H + h) * a.Nq + i] = sqrtf(n2);
            }
        }
        AT_MARK(6);
    }

    if constexpr (SCORES && HS == 2) {  // merge the two parity groups' head-max through the (now idle) ring memory
        f32x4* xch = (f32x4*)smem;
        __syncthreads();
        if (hp == 1)
#pragma unroll
            for (int t = 0; t < NT; ++t) xch[(wave * NT + t) * 64 + lane] = pmax[t];
        __syncthreads();
        if (hp == 1) return;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 o = xch[(wave * NT + t) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) pmax[t][r] = fmaxf(pmax[t][r], o[r]);
        }
    }
    if constexpr (SCORES) {
        if (active) {
            const int i = i0 + l16;
            const bool valid = i >= 1 && i < a.Nq;
            float* dst = a.colsum + ((size_t)b * a.nrt + rt) * a.Nk;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(valid ? pmax[t][r] : 0.f);
                    const int j = 16 * t + 4 * g + r;
                    if (l16 == 0 && j < a.Nk) dst[j] = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Short-sequence self-attention with scores (Nq = Nk <= 32: the 20-token text side of NLVR).  With one or two 16-row
// query tiles the general kernel leaves half its waves idle and serialises 12 heads behind block barriers.  Here a
// workgroup is one sample and every wave is independent: wave w takes heads w, w+SMALL_NW, .. for BOTH query tiles, reads K
// and Q fragments straight from global memory, stages V_h (<= 32 x 64 bf16 = 4 KiB) in its private LDS slice for the
// transpose reads, and keeps its partial head-max in registers; the four partial maxima are combined through LDS once.
// SMALL_NW waves per workgroup: with 12 heads every wave owns exactly one head (a head is a ~3 us dependent chain; four
// waves walking three heads each made the kernel three chains long).
constexpr int SMALL_NW = 12;
template <bool F16>
__global__ __launch_bounds__(64 * SMALL_NW, 1) void attn_bf16_small_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char vbuf[SMALL_NW][32 * 128];
    __shared__ float pm[SMALL_NW][2][2][64][4];  // [wave][row tile][key tile][lane][r]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int N = a.Nk;
    char* Vs = vbuf[wave];
    f32x4 pmax[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t) pmax[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nrt = (N + 15) / 16;
    f32x4 mk[2];  // additive key mask of this lane's columns (0 without a mask, -inf for j >= N), loaded once
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r;
            mk[t][r] = (j < N ? (a.mask ? a.mask[(size_t)b * N + j] : 0.f) : -INFINITY) * 1.44269504088896341f;
        }
    const float c2 = a.scale * 1.44269504088896341f;  // log2 units, the same arithmetic as attn_bf16_kernel (bit-identical results:
                                                      // the sync-free encoder runs that kernel where this one serves the host-k path)

    for (int h = wave; h < a.H; h += SMALL_NW) {
        // V_h rows -> private LDS (row-major 128-byte rows, chunk ^= 2*((row>>1)&3) as in the general kernel)
        {
            const int sub = lane >> 3, pos = lane & 7;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                int row = grp * 8 + sub;
                const int chunk = pos ^ (((row >> 1) & 3) << 1);
                row = row < N ? row : N - 1;
                *(bf16x8*)(Vs + grp * 1024 + lane * 16) =
                    *(const bf16x8*)(a.v + (((size_t)b * N + row) * a.ldv + h * 64) * 2 + chunk * 16);
            }
        }
        // K fragments (A operand): key row j = 16t + l16, d = 32kk + 8g .. +7, straight from global
        bf16x8 kf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = min(16 * t + l16, N - 1);
            const char* kp = a.k + (((size_t)b * N + j) * a.ldk + h * 64) * 2 + g * 16;
            kf[t][0] = *(const bf16x8*)kp;
            kf[t][1] = *(const bf16x8*)(kp + 64);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt >= nrt) break;
            const int i0 = rt * 16;
            const int irow = min(i0 + l16, N - 1);
            const char* qp = a.q + (((size_t)b * N + irow) * a.ldq + h * 64) * 2 + g * 16;
            const bf16x8 q0 = *(const bf16x8*)qp, q1 = *(const bf16x8*)(qp + 64);
            f32x4 sc[2];
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc = mfma_lp<F16>(kf[t][0], q0, acc, 0, 0, 0);
                acc = mfma_lp<F16>(kf[t][1], q1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaf(acc[r], c2, mk[t][r]);
                    acc[r] = v;
                    m = fmaxf(m, v);
                }
                sc[t] = acc;
            }
            m = rows4_max(m);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[t][r] = __builtin_amdgcn_exp2f(sc[t][r] - m); sum += sc[t][r]; }
            sum = rows4_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sc[t] *= inv;
#pragma unroll
                for (int r = 0; r < 4; ++r) pmax[rt][t][r] = fmaxf(pmax[rt][t][r], sc[t][r]);
            }
            if (i0 == 0) if (l16 == 0) {
                float* dst = a.p0 + ((size_t)b * a.H + h) * N;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * t + 4 * g + r;
                        if (j < N) dst[j] = sc[t][r];
                    }
            }
            // O^T = V^T P^T (operands swapped): lane (i = l16, g) gets columns 16dt + 4g .. +3 of row i
            const bf16x8 pa = pack_lp8<F16>(sc[0], sc[1]);
            const int vrow = 4 * g + (l16 >> 2);
            const int vkey = ((vrow >> 1) & 3) << 1;
            const char* vr = Vs + vrow * 128 + 8 * (l16 & 1);
            f32x4 o[4];
            float n2 = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int p = ((2 * dt + ((l16 & 3) >> 1)) ^ vkey) << 4;
                const bf16x8 vb = cat_bf16x4(lds_read_tr16(vr + p), lds_read_tr16(vr + 16 * 128 + p));
                o[dt] = mfma_lp<F16>(vb, pa, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) n2 += o[dt][r] * o[dt][r];
            }
            n2 = rows4_sum(n2);
            const int i = i0 + l16;
            if (i < N) {
                bf16_t* orow = (bf16_t*)(a.out + (((size_t)b * N + i) * a.ldo + h * 64) * 2) + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(orow + dt * 16) = pack_lp4<F16>(o[dt]);
                if (g == 0) a.onorm[((size_t)b * a.H + h) * N + i] = sqrtf(n2);
            }
        }
    }
    // combine the per-wave partial head-max, then the column mass per 16-row tile
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) pm[wave][rt][t][lane][r] = pmax[rt][t][r];
    __syncthreads();
    if (wave < nrt) {  // wave w finishes row tile w
        const int rt = wave, i = rt * 16 + l16;
        const bool valid = i >= 1 && i < N;
        float* dst = a.colsum + ((size_t)b * a.nrt + rt) * N;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = pm[0][rt][t][lane][r];
#pragma unroll
                for (int w = 1; w < SMALL_NW; ++w) v = fmaxf(v, pm[w][rt][t][lane][r]);
                v = row16_sum(valid ? v : 0.f);
                const int j = 16 * t + 4 * g + r;
                if (l16 == 0 && j < N) dst[j] = v;
            }
    }
}

template <int NT, bool SCORES, bool F16>
int launch_attn_bf16(const AttnArgs& a, hipStream_t s) {
    constexpr int NC = (NT + 1) / 2;
    constexpr int HS = (SCORES && NT <= 8) ? 2 : 1;  // head-parity split: 4 rings must fit the 160 KiB of LDS
    constexpr int RB = (SCORES && NT > 8) ? 2 : 1;   // two 64-row blocks per workgroup where only one ring fits a CU
    const size_t lds = (size_t)2 * HS * (NT * 16 + NC * 32) * 128;
    MADTP_ENSURE_MAX_LDS((attn_bf16_kernel<NT, SCORES, HS, RB, F16>), lds);
    int gz = 1;
    size_t lds_used = lds;
    if (!SCORES) {
        // cross-attention (few query rows per sample): below 512 workgroups every head gets its own workgroup - no head loop,
        // so ONE ring stage is allocated and 2-6 workgroups share a CU (the two-stage ring of a 190-key problem is 98 KiB,
        // one 4-wave workgroup per CU)
        const int wgs = ((a.Nq + 63) / 64) * a.B;
        gz = wgs >= 512 ? 1 : a.H;
        if (gz == a.H) lds_used = lds / 2;
    }
    hipLaunchKernelGGL((attn_bf16_kernel<NT, SCORES, HS, RB, F16>), dim3((a.Nq + 64 * RB - 1) / (64 * RB), (!SCORES && a.pair) ? 2 * a.B : a.B, gz),
                       dim3(256 * HS * RB), lds_used, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

template <bool SCORES, bool F16 = false>
int dispatch_nt_bf16(const AttnArgs& a, hipStream_t s) {
    const int nt = (a.Nk + 15) / 16;
    if (nt <= 4) return launch_attn_bf16<4, SCORES, F16>(a, s);
    if (nt <= 6) return launch_attn_bf16<6, SCORES, F16>(a, s);
    if (nt <= 8) return launch_attn_bf16<8, SCORES, F16>(a, s);
    if (nt <= 10) return launch_attn_bf16<10, SCORES, F16>(a, s);
    if (nt <= 12) return launch_attn_bf16<12, SCORES, F16>(a, s);
    if (nt <= 13) return launch_attn_bf16<13, SCORES, F16>(a, s);
    if (nt <= 16) return launch_attn_bf16<16, SCORES, F16>(a, s);
    return MADTP_E_SHAPE;
}

// ------------------------------------------------------------------------------------------------------------------
// Long-sequence variant (256 < Nk <= 1024: 384x384 / 480x480 images give 577 / 901 tokens).  The score row no longer
// fits in registers next to the head-max, so every head makes two passes over the keys in 128-key chunks staged in LDS:
//   pass A: S = K Q^T chunk by chunk -> running row maximum m and sum l (online softmax statistics, exact at the end);
//   pass B: S recomputed, P = exp(S - m) / l (the same normalised probabilities as the short kernel), head-max of P
//           kept in registers for ALL key tiles (NCH*8 x 4 floats per lane), P.V accumulated, CLS row written.
// Arithmetic is the exact-f32 MFMA path in both precision modes (T is only the storage type): this kernel exists for
// the parity cases at large images, not for the headline benchmark.
template <typename T, int NCH, bool SCORES>
__global__ __launch_bounds__(256, 1) void attn_large_kernel(AttnArgs a) {
    constexpr int ESZ = sizeof(T);
    constexpr int RB = 64 * ESZ + 16;
    constexpr int CPR = 64 * ESZ / 16;
    constexpr int CK = 128;                  // keys per chunk
    constexpr int NT = NCH * 8;              // 16-key tiles in total
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + CK * RB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int bkv = a.kvidx ? a.kvidx[b] : b;  // K/V block of this sample
    const int rt = blockIdx.x * 4 + wave;
    const int i0 = rt * 16;
    const bool active = i0 < a.Nq;
    const int irow = min(i0 + l16, a.Nq - 1);

    f32x4 pmax[SCORES ? NT : 1];
    if constexpr (SCORES) {
#pragma unroll
        for (int t = 0; t < NT; ++t) pmax[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    auto stage = [&](int h, int c, bool with_v) {
        for (int idx = tid; idx < CK * CPR; idx += 256) {
            const int row = idx / CPR, ch = idx % CPR, j = c * CK + row;
            uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
            if (j < a.Nk) {
                const size_t grow = (size_t)bkv * a.kvb + j;
                kv = *(const uint4*)(a.k + (grow * a.ldk + h * 64) * ESZ + ch * 16);
                if (with_v) vv = *(const uint4*)(a.v + (grow * a.ldv + h * 64) * ESZ + ch * 16);
            }
            *(uint4*)(Ks + row * RB + ch * 16) = kv;
            if (with_v) *(uint4*)(Vs + row * RB + ch * 16) = vv;
        }
    };
    // S^T for the 8 tiles of one chunk (scale and mask applied, keys beyond Nk -> -inf)
    auto scores = [&](int c, const f32x4 (&q)[4], f32x4 (&sc)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 kf[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) kf[t] = load4<T>(Ks + (16 * t + l16) * RB + (4 * s + g) * 4 * ESZ);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 8; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t][e], q[s][e], sc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = c * CK + 16 * t + 4 * g + r;
                float v = sc[t][r] * a.scale;
                if (a.mask && j < a.Nk) v += a.mask[(size_t)b * a.Nk + j];
                sc[t][r] = j < a.Nk ? v : -INFINITY;
            }
    };

    for (int h = blockIdx.z; h < a.H; h += gridDim.z) {
        f32x4 q[4];
        {
            const char* qp = a.q + (((size_t)b * a.Nq + irow) * a.ldq + h * 64) * ESZ;
#pragma unroll
            for (int s = 0; s < 4; ++s) q[s] = load4<T>(qp + (4 * s + g) * 4 * ESZ);
        }
        // ---- pass A: row statistics ----
        float m = -INFINITY, l = 0.f;
        for (int c = 0; c < NCH; ++c) {
            if (c * CK >= a.Nk) break;
            __syncthreads();
            stage(h, c, false);
            __syncthreads();
            if (!active) continue;
            f32x4 sc[8];
            scores(c, q, sc);
            float cm = -INFINITY;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cm = fmaxf(cm, sc[t][r]);
            cm = rows4_max(cm);
            const float mn = fmaxf(m, cm);  // chunk 0 always holds key 0, so mn is finite
            float cs = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs += expf(sc[t][r] - mn);
            cs = rows4_sum(cs);
            l = l * expf(m - mn) + cs;
            m = mn;
        }
        // ---- pass B: probabilities, head-max, P.V ----
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c * CK < a.Nk) {  // block-uniform
                __syncthreads();
                stage(h, c, true);
                __syncthreads();
                if (active) {
                    f32x4 sc[8];
                    scores(c, q, sc);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = expf(sc[t][r] - m) / l;
                            sc[t][r] = p;
                            if constexpr (SCORES) pmax[c * 8 + t][r] = fmaxf(pmax[c * 8 + t][r], p);
                        }
                    }
                    if constexpr (SCORES) {
                        if (i0 + l16 == 0) {
                            float* dst = a.p0 + ((size_t)b * a.H + h) * a.Nk;
#pragma unroll
                            for (int t = 0; t < 8; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int j = c * CK + 16 * t + 4 * g + r;
                                    if (j < a.Nk) dst[j] = sc[t][r];
                                }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const char* vrow = Vs + (16 * t + 4 * g + r) * RB + l16 * ESZ;
#pragma unroll
                            for (int dt = 0; dt < 4; ++dt)
                                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[t][r], load1<T>(vrow + dt * 16 * ESZ), o[dt], 0, 0, 0);
                        }
                }
            }
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 4 * g + r;
                float n2 = 0.f;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) n2 += o[dt][r] * o[dt][r];
                if constexpr (SCORES) n2 = row16_sum(n2);
                if (i < a.Nq) {
                    T* orow = (T*)(a.out + (((size_t)b * a.Nq + i) * a.ldo + h * 64) * ESZ);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) orow[dt * 16 + l16] = from_f32<T>(o[dt][r]);
                    if constexpr (SCORES)
                        if (l16 == 0) a.onorm[((size_t)b * a.H + h) * a.Nq + i] = sqrtf(n2);
                }
            }
        }
    }
    if constexpr (SCORES) {
        if (active) {
            const int i = i0 + l16;
            const bool valid = i >= 1 && i < a.Nq;
            float* dst = a.colsum + ((size_t)b * a.nrt + rt) * a.Nk;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(valid ? pmax[t][r] : 0.f);
                    const int j = 16 * t + 4 * g + r;
                    if (l16 == 0 && j < a.Nk) dst[j] = v;
                }
        }
    }
}

// f16-split flavour of attn_large_kernel for the f16x3 precision mode (see attn_f16s_kernel): the same two passes and the same
// softmax arithmetic, S^T = K Q^T and O^T = V^T P^T as three f16 MFMA products of f16-split operands.  A 128-key chunk is staged
// as two f16 planes per operand (160-byte rows, 80 KiB for K and V); all eight 16-byte pieces of a thread are loaded before the
// first is split and stored, so their latencies overlap.
template <int NCH, bool SCORES>
__global__ __launch_bounds__(256, 1) void attn_large_f16s_kernel(AttnArgs a) {  // (two per CU would fit the LDS, not the head-max registers)
    constexpr int PITCH = 160;
    constexpr int CK = 128;
    constexpr int NT = NCH * 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* K0 = smem;
    char* K1 = K0 + CK * PITCH;
    char* V0 = K1 + CK * PITCH;
    char* V1 = V0 + CK * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int bkv = a.kvidx ? a.kvidx[b] : b;
    const int rt = blockIdx.x * 4 + wave;
    const int i0 = rt * 16;
    const bool active = i0 < a.Nq;
    const int irow = min(i0 + l16, a.Nq - 1);
    constexpr float LO = 1.0f / F16S_LO_SCALE;

    f32x4 pmax[SCORES ? NT : 1];
    if constexpr (SCORES) {
#pragma unroll
        for (int t = 0; t < NT; ++t) pmax[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    auto stage = [&](int h, int c, bool with_v) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // four pieces per operand in flight at a time (register budget)
            uint4 kr[4], vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * (4 * half + i), row = idx >> 4, ch = idx & 15, j = c * CK + row;
                kr[i] = make_uint4(0, 0, 0, 0);
                vr[i] = make_uint4(0, 0, 0, 0);
                if (j < a.Nk) {
                    const size_t grow = (size_t)bkv * a.kvb + j;
                    kr[i] = *(const uint4*)(a.k + (grow * a.ldk + h * 64) * 4 + ch * 16);
                    if (with_v) vr[i] = *(const uint4*)(a.v + (grow * a.ldv + h * 64) * 4 + ch * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * (4 * half + i), row = idx >> 4, ch = idx & 15;
                f16x4 h4, l4;
                split_f16x4(__builtin_bit_cast(f32x4, kr[i]), h4, l4);
                *(f16x4*)(K0 + row * PITCH + ch * 8) = h4;
                *(f16x4*)(K1 + row * PITCH + ch * 8) = l4;
                if (with_v) {
                    split_f16x4(__builtin_bit_cast(f32x4, vr[i]), h4, l4);
                    *(f16x4*)(V0 + row * PITCH + ch * 8) = h4;
                    *(f16x4*)(V1 + row * PITCH + ch * 8) = l4;
                }
            }
        }
    };
    // S^T for the 8 tiles of one chunk (scale and mask applied as in attn_large_kernel, keys beyond Nk -> -inf)
    auto scores = [&](int c, const f16x8 (&qh)[2], const f16x8 (&ql)[2], f32x4 (&sc)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int off = (16 * t + l16) * PITCH + g * 16;
            const f16x8 kh0 = *(const f16x8*)(K0 + off), kh1 = *(const f16x8*)(K0 + off + 64);
            const f16x8 kl0 = *(const f16x8*)(K1 + off), kl1 = *(const f16x8*)(K1 + off + 64);
            const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, qh[0], z, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, qh[1], hi, 0, 0, 0);
            f32x4 lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, ql[0], z, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, ql[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl0, qh[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl1, qh[1], lo, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = c * CK + 16 * t + 4 * g + r;
                float v = fmaf(lo[r], LO, hi[r]) * a.scale;
                if (a.mask && j < a.Nk) v += a.mask[(size_t)b * a.Nk + j];
                sc[t][r] = j < a.Nk ? v : -INFINITY;
            }
        }
    };

    for (int h = blockIdx.z; h < a.H; h += gridDim.z) {
        f16x8 qh[2], ql[2];
        {
            const char* qp = a.q + (((size_t)b * a.Nq + irow) * a.ldq + h * 64) * 4 + g * 32;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 p0, p1;
                split_f16x8(*(const f32x4*)(qp + kk * 128), *(const f32x4*)(qp + kk * 128 + 16), p0, p1);
                qh[kk] = __builtin_bit_cast(f16x8, p0);
                ql[kk] = __builtin_bit_cast(f16x8, p1);
            }
        }
        // ---- pass A: row statistics ----
        float m = -INFINITY, l = 0.f;
        for (int c = 0; c < NCH; ++c) {
            if (c * CK >= a.Nk) break;
            __syncthreads();
            stage(h, c, false);
            __syncthreads();
            if (!active) continue;
            f32x4 sc[8];
            scores(c, qh, ql, sc);
            float cm = -INFINITY;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cm = fmaxf(cm, sc[t][r]);
            cm = rows4_max(cm);
            const float mn = fmaxf(m, cm);
            float cs = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs += expf(sc[t][r] - mn);
            cs = rows4_sum(cs);
            l = l * expf(m - mn) + cs;
            m = mn;
        }
        // ---- pass B: probabilities, head-max, O^T = V^T P^T ----
        f32x4 oh[4], ol[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { oh[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ol[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c * CK < a.Nk) {  // block-uniform
                __syncthreads();
                stage(h, c, true);
                __syncthreads();
                if (active) {
                    f32x4 sc[8];
                    scores(c, qh, ql, sc);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = expf(sc[t][r] - m) / l;
                            sc[t][r] = p;
                            if constexpr (SCORES) pmax[c * 8 + t][r] = fmaxf(pmax[c * 8 + t][r], p);
                        }
                    }
                    if constexpr (SCORES) {
                        if (i0 == 0 && l16 == 0) {
                            float* dst = a.p0 + ((size_t)b * a.H + h) * a.Nk;
#pragma unroll
                            for (int t = 0; t < 8; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int j = c * CK + 16 * t + 4 * g + r;
                                    if (j < a.Nk) dst[j] = sc[t][r];
                                }
                        }
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {  // 32-key sub-chunks of the staged chunk
                        u32x4 pp0, pp1;
                        split_f16x8(sc[2 * cc], sc[2 * cc + 1], pp0, pp1);
                        const f16x8 ph = __builtin_bit_cast(f16x8, pp0), pl = __builtin_bit_cast(f16x8, pp1);
                        const int voff = (32 * cc + 4 * g + (l16 >> 2)) * PITCH + 8 * (l16 & 3);
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            const bf16x8 vh = cat_bf16x4(lds_read_tr16(V0 + voff + dt * 32), lds_read_tr16(V0 + voff + 16 * PITCH + dt * 32));
                            const bf16x8 vl = cat_bf16x4(lds_read_tr16(V1 + voff + dt * 32), lds_read_tr16(V1 + voff + 16 * PITCH + dt * 32));
                            oh[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), ph, oh[dt], 0, 0, 0);
                            ol[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), pl, ol[dt], 0, 0, 0);
                            ol[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vl), ph, ol[dt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (active) {
            const int i = i0 + l16;
            f32x4 o[4];
            float n2 = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[dt][r] = fmaf(ol[dt][r], LO, oh[dt][r]);
                    n2 += o[dt][r] * o[dt][r];
                }
            if constexpr (SCORES) n2 = rows4_sum(n2);
            if (i < a.Nq) {
                if (a.split_dim) {  // straight into the consumer GEMM's operand planes (8-byte stores per plane)
                    _Float16* prow = (_Float16*)a.out + ((size_t)b * a.Nq + i) * a.ldo + h * 64 + 4 * g;
                    bool bad = false;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        f16x4 ph_, pl_;
                        split_f16x4(o[dt], ph_, pl_);
                        *(f16x4*)(prow + dt * 16) = ph_;
                        *(f16x4*)(prow + a.split_dim + dt * 16) = pl_;
                        bad |= f16_range_bad(o[dt]);
                    }
                    f16_range_raise(a.range_flag, bad);
                } else {
                    float* orow = (float*)(a.out + (((size_t)b * a.Nq + i) * a.ldo + h * 64) * 4) + 4 * g;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) *(f32x4*)(orow + dt * 16) = o[dt];
                }
                if constexpr (SCORES)
                    if (g == 0) a.onorm[((size_t)b * a.H + h) * a.Nq + i] = sqrtf(n2);
            }
        }
    }
    if constexpr (SCORES) {
        if (active) {
            const int i = i0 + l16;
            const bool valid = i >= 1 && i < a.Nq;
            float* dst = a.colsum + ((size_t)b * a.nrt + rt) * a.Nk;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(valid ? pmax[t][r] : 0.f);
                    const int j = 16 * t + 4 * g + r;
                    if (l16 == 0 && j < a.Nk) dst[j] = v;
                }
        }
    }
}

template <int NCH, bool SCORES>
int launch_attn_large_f16s(const AttnArgs& a, hipStream_t s) {
    const size_t lds = (size_t)4 * 128 * 160;
    MADTP_ENSURE_MAX_LDS((attn_large_f16s_kernel<NCH, SCORES>), lds);
    int gz = 1;
    if (!SCORES) {
        const int wgs = ((a.Nq + 63) / 64) * a.B;
        gz = wgs >= 512 ? 1 : (wgs >= 128 ? 4 : a.H);
        if (gz > a.H) gz = a.H;
    }
    hipLaunchKernelGGL((attn_large_f16s_kernel<NCH, SCORES>), dim3((a.Nq + 63) / 64, a.B, gz), dim3(256), lds, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

template <bool SCORES>
int dispatch_large_f16s(const AttnArgs& a, hipStream_t s) {
    const int nch = (a.Nk + 127) / 128;
    if (nch <= 5) return launch_attn_large_f16s<5, SCORES>(a, s);
    if (nch <= 8) return launch_attn_large_f16s<8, SCORES>(a, s);
    return MADTP_E_SHAPE;
}

template <typename T, int NCH, bool SCORES>
int launch_attn_large(const AttnArgs& a, hipStream_t s) {
    constexpr int RB = 64 * (int)sizeof(T) + 16;
    const size_t lds = (size_t)2 * 128 * RB;
    MADTP_ENSURE_MAX_LDS((attn_large_kernel<T, NCH, SCORES>), lds);
    int gz = 1;
    if (!SCORES) {
        const int wgs = ((a.Nq + 63) / 64) * a.B;
        gz = wgs >= 512 ? 1 : (wgs >= 128 ? 4 : a.H);
        if (gz > a.H) gz = a.H;
    }
    hipLaunchKernelGGL((attn_large_kernel<T, NCH, SCORES>), dim3((a.Nq + 63) / 64, a.B, gz), dim3(256), lds, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 fast-mode kernel for long sequences (256 < Nk <= 1024).  Same two passes per head as attn_large_kernel (the head-max
// of the NORMALISED probabilities needs a row's final max / sum before any probability can be compared across heads), but
//   * S^T = K Q^T and O^T = V^T P^T run on the bf16 MFMA with the register layouts of attn_bf16_kernel (swapped products:
//     a lane owns one query row, the softmax reductions are in-lane + two permlane steps, P feeds P.V without leaving the lane,
//     V is consumed with ds_read_b64_tr_b16);
//   * the 128-key chunks of K (pass A) and K|V (pass B) stream through a 2-stage LDS ring by LDS-DMA: the chunk after the
//     current one - across pass and head boundaries - is in flight while the current one is computed, one barrier per chunk;
//   * exp via v_exp_f32, one reciprocal per row.
// The head-max stays in registers for all key tiles as packed f16 pairs (NT x 2 registers per lane: 80 at 577 keys, 114 at 901),
// two 4-wave workgroups per CU up to 640 keys (the 1024-key instantiation would spill at 256 registers and keeps one).
// Work per (b, h): 6 Nq Nk 64 flop (Q K^T twice) on the bf16 MFMA; the bound is the VALU softmax work (2 exp per score, each
// one fma + v_exp_f32 on the unscaled score, see scores()).
// STG: stages of the chunk ring (2 is what launch_attn_bf16_large uses; 3 keeps two chunks in flight behind counted vmcnt waits
// and was measured slower, see there).
// HV = 2 (round 5, the 641..1024-key instantiation with scores: VQA's 901 tokens): the head-max of all 64 key tiles is 128
// registers and pinned the kernel at ONE wave per SIMD (318 registers; a chunk step of a lone wave is bound by its own dependent
// instruction stream, ~1.85 us against ~1.2 us with two waves per SIMD).  Here the work is re-ordered into three sweeps over the
// heads - (1) pass A of every head, the row statistics (max, normalising offset) parked in LDS; (2) pass B on the FIRST half of the
// key chunks of every head: probabilities, head-max of those 32 key tiles (64 registers), CLS row, partial P.V, parked as f32 in a
// scratch buffer; (3) pass B on the second half, its head-max, the partial sums completed and stored - so that two workgroups fit
// a CU.  Same arithmetic per element; the P.V sum of a row is split at the half boundary (f32 partial written and re-read
// exactly), i.e. the same association as the one-sweep order.  gridDim.z == 1.
template <int NCH, bool SCORES, int STG, bool F16 = false, int HV = 1>
__global__ __launch_bounds__(256, (STG == 2 && (NCH <= 5 || !SCORES || HV == 2)) ? 2 : 1) void attn_bf16_large_kernel(AttnArgs a) {
    static_assert(HV == 1 || (HV == 2 && SCORES && STG == 2 && NCH % 2 == 0), "the three-sweep order exists with scores on the two-stage ring");
    constexpr int CK = 128;                   // keys per chunk
    constexpr int STAGE = 2 * CK * 128;       // K image (128-byte rows), then V image
    constexpr int NT = NCH * 8 / HV;          // key tiles whose head-max is held at a time
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    // (A 1-D grid that puts all row blocks / head groups of a sample on ONE XCD - so that its K and V are fetched by one L2 only -
    //  was measured SLOWER: 140 -> 164 us at 605 keys, 650 -> 785 us at 901: the 64 resident workgroups of an XCD then walk the
    //  same K/V lines at the same time.  The round-robin 3-D grid interleaves the samples on every XCD.)
    const int nrb = gridDim.x, ngz = gridDim.z;
    const int blk_x = blockIdx.x, blk_z = blockIdx.z;
    const int b = blockIdx.y;
    const int bkv = a.kvidx ? a.kvidx[b] : b;
    const int rt = blk_x * 4 + wave;
    const int i0 = rt * 16;
    const bool active = i0 < a.Nq;
    const int irow = min(i0 + l16, a.Nq - 1);
    const int nch = (a.Nk + CK - 1) / CK;

    // head-max of P for all key tiles, held as PACKED f16 pairs (probabilities <= 1: 2^-11 relative rounding, the column sums
    // over ~N rows average it out; halves the registers of this array - 114 instead of 228 at 901 keys - so that two
    // workgroups fit a CU, and the update is one v_pk_max_f16 per two values)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 pmax[SCORES ? NT : 1][2];
    if constexpr (SCORES) {
#pragma unroll
        for (int t = 0; t < NT; ++t) { pmax[t][0] = (h2){0, 0}; pmax[t][1] = (h2){0, 0}; }
    }

    // LDS-DMA of one chunk: 16 K and 16 V instructions of 8 rows (1 KiB) each, 4 + 4 per wave.  Swizzles on the SOURCE address
    // as in attn_bf16_kernel: K chunk ^= row & 7, V chunk ^= 2 * ((row >> 1) & 3).  buffer_load ... lds: the per-lane part of
    // the address (row within the chunk, swizzled 16-byte slot) is a constant VGPR offset, the step-dependent part (chunk, head)
    // a scalar offset - no per-step address arithmetic on the VALU (it was ~150 of the ~450 instructions of a chunk step) - and
    // rows >= Nk fall outside the sample's descriptor and read as zeros (their P is 0 / masked).
    const int sub = lane >> 3, pos = lane & 7;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)(a.k + (size_t)bkv * a.kvb * a.ldk * 2), 0,
                                                                          (unsigned)((size_t)a.Nk * a.ldk * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)(a.v + (size_t)bkv * a.kvb * a.ldv * 2), 0,
                                                                          (unsigned)((size_t)a.Nk * a.ldv * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)(a.q + (size_t)b * a.Nq * a.ldq * 2), 0,
                                                                          (unsigned)((size_t)a.Nq * a.ldq * 2), 0x00020000);
    unsigned kro[4], vro[4], qro[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + sub;
        kro[i] = (unsigned)r * (unsigned)a.ldk * 2u + (unsigned)((pos ^ (r & 7)) << 4);
        vro[i] = (unsigned)r * (unsigned)a.ldv * 2u + (unsigned)((pos ^ (((r >> 1) & 3) << 1)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) qro[i] = (unsigned)(i0 + i * 8 + sub) * (unsigned)a.ldq * 2u + (unsigned)((pos ^ sub) << 4);
    // Q rows of this wave (16 x 128 B per head) travel by LDS-DMA as well, into a wave-private 2 KiB image behind the ring (same
    // swizzle as K), issued with the first K chunk of their head: a plain global load inside the chunk loops makes the compiler
    // wait for vmcnt(0) at its first use - i.e. for every chunk DMA in flight - on each iteration.
    char* const qimg = smem + STG * STAGE + wave * 2048;
    auto stage_q = [&](int h) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, LDS_PTR(qimg + i * 1024), 16, qro[i], h * 128, 0, 0);
    };
    auto stage = [&](int h, int c, bool with_v, int st) {
        char* base = smem + st * STAGE + wave * 4 * 1024;
        const int sk = (c * CK * a.ldk + h * 64) * 2, sv = (c * CK * a.ldv + h * 64) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, LDS_PTR(base + i * 1024), 16, kro[i], sk, 0, 0);
            if (with_v) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, LDS_PTR(base + CK * 128 + i * 1024), 16, vro[i], sv, 0, 0);
        }
    };
    const float c2 = a.scale * 1.44269504088896341f, inv_scale = 1.f / a.scale;
    // S^T of one chunk (8 key tiles), unscaled (see below), masked; keys >= Nk -> -inf
    auto scores = [&](const char* Ks, int c, const bf16x8 (&q)[2], f32x4 (&sc)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 16 * t + l16;
            const bf16x8 k0 = *(const bf16x8*)(Ks + row * 128 + (((0 + g) ^ (row & 7)) << 4));
            const bf16x8 k1 = *(const bf16x8*)(Ks + row * 128 + (((4 + g) ^ (row & 7)) << 4));
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = mfma_lp<F16>(k0, q[0], acc, 0, 0, 0);
            sc[t] = mfma_lp<F16>(k1, q[1], acc, 0, 0, 0);
        }
        // The scores stay UNSCALED: exp(scale s - max) is evaluated as exp2(fma(s, c2, off)) with c2 = scale log2(e) and a per-row
        // offset (pass A: -c2 max; pass B: -(c2 max + log2 sum), which normalises as well) - one fma + one v_exp_f32 per score
        // instead of scale, subtract, the log2(e) multiply of expf, v_exp_f32 and the 1 / sum multiply: the VALU softmax work
        // is this kernel's bound.  An additive mask joins in units of 1 / scale.
        const int jb = c * CK + 4 * g;
        if (a.mask) {
            const float* mrow = a.mask + (size_t)b * a.Nk;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jb + 16 * t + r;
                    sc[t][r] = j < a.Nk ? fmaf(mrow[j], inv_scale, sc[t][r]) : -INFINITY;
                }
        } else if ((c + 1) * CK > a.Nk) {  // (a full chunk needs no bounds)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][r] = (jb + 16 * t + r) < a.Nk ? sc[t][r] : -INFINITY;
        }
    };
    const int hstep = ngz;
    // The chunk steps of this workgroup form one stream (head, pass A / B, chunk); the DMA of step s + STG - 1 is issued at step s.
    // step_sync(): wait for the DMA of the current step (the STG - 2 younger ones may still fly: vmcnt counts 4 K instructions
    // of a pass-A step, 8 K|V instructions of a pass-B step per wave; output stores issued in between only make the wait a
    // little stricter), barrier (every wave is done with the stage that is overwritten next), issue.
    int st = 0, is_st = 0, is_h = blk_z, is_p = 0, is_c = 0;
    auto issue = [&]() {
        if (is_h < a.H) {
            if (is_p == 0 && is_c == 0) stage_q(is_h);
            stage(is_h, is_c, is_p == 1, is_st);
        }
        if (++is_st == STG) is_st = 0;
        if (++is_c == nch) { is_c = 0; if (++is_p == 2) { is_p = 0; is_h += hstep; } }
    };
    auto step_sync = [&](int h, int p, int c) {
        if constexpr (STG == 2) {
            __syncthreads();  // chunk landed (the barrier drains the DMA); the other stage is free again
        } else {
            int nh = h, np = p, nc = c + 1;  // the step after the current one: its DMA stays in flight
            if (nc == nch) { nc = 0; if (++np == 2) { np = 0; nh += hstep; } }
            const int fly = nh < a.H ? (np ? 8 : (nc == 0 ? 6 : 4)) : 0;  // (+2: the Q rows ride with a head's first K chunk)
            if (fly == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (fly == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (fly == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        issue();
    };
    if constexpr (HV == 2) {
        constexpr int CH0 = NCH / 2;  // chunks of the first key half
        // stream of this workgroup's chunk steps: sweep 0 = (h, all chunks, K only), sweep 1 = (h, chunks [0, CH0), K|V),
        // sweep 2 = (h, chunks [CH0, nch), K|V); the DMA of the step after the current one is issued behind the current barrier
        int sw_i = 0, h_i = 0, c_i = 0, st_i = 0;
        auto issue2 = [&]() {
            if (sw_i < 3) {
                const int c_lo = sw_i == 2 ? CH0 : 0, c_hi = sw_i == 1 ? CH0 : nch;
                if (c_i == c_lo) stage_q(h_i);
                stage(h_i, c_i, sw_i != 0, st_i);
                st_i ^= 1;
                if (++c_i == c_hi) {
                    if (++h_i == a.H) { h_i = 0; ++sw_i; }
                    c_i = sw_i == 2 ? CH0 : 0;
                }
            }
        };
        bf16x8 q[2];
        // barrier (the current step's chunk landed - the barrier drains the DMA - and the other stage is free), the head's Q rows
        // on its first step (read BEFORE the next step's DMA is issued: that one may carry the next head's Q rows), then the issue
        auto sync2 = [&](bool first) {
            __syncthreads();
            if (first) {
                q[0] = *(const bf16x8*)(qimg + l16 * 128 + (((0 + g) ^ (l16 & 7)) << 4));
                q[1] = *(const bf16x8*)(qimg + l16 * 128 + (((4 + g) ^ (l16 & 7)) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]) :: "memory");
            }
            issue2();
        };
        float* const stats = (float*)(smem + STG * STAGE + 4 * 2048) + (wave * 16 + l16) * 2;  // [head][wave][row]{max, offset}
        float* const opart = a.o_part;  // f32 [B, Nq, H, 64]: the first half's P.V sums
        issue2();
        // ---- sweep 0: row maximum and sum over all keys of every head ----
        for (int h = 0; h < a.H; ++h) {
            float m = -INFINITY, l = 0.f;
#pragma nounroll
            for (int c = 0; c < nch; ++c, st ^= 1) {
                sync2(c == 0);
                if (!active) continue;
                f32x4 sc[8];
                scores(smem + st * STAGE, c, q, sc);
                float cm = -INFINITY;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cm = fmaxf(cm, sc[t][r]);
                cm = rows4_max(cm);
                const float mn = fmaxf(m, cm);
                const float off = -mn * c2;
                float cs = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cs += __builtin_amdgcn_exp2f(fmaf(sc[t][r], c2, off));
                cs = rows4_sum(cs);
                l = l * __builtin_amdgcn_exp2f((m - mn) * c2) + cs;
                m = mn;
            }
            if (g == 0) stats[h * 128 + 1] = -(m * c2 + __builtin_amdgcn_logf(l));  // P = exp2(c2 s + offset): normalised
        }
        // ---- sweeps 1, 2: probabilities, head-max, CLS row, P.V on one half of the key chunks ----
        for (int half = 0; half < 2; ++half) {
            const int c_lo = half ? CH0 : 0, c_hi = half ? nch : CH0;
#pragma unroll
            for (int t = 0; t < NT; ++t) { pmax[t][0] = (h2){0, 0}; pmax[t][1] = (h2){0, 0}; }
            for (int h = 0; h < a.H; ++h) {
                const float poff = stats[h * 128 + 1];
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma nounroll
                for (int c = c_lo; c < c_hi; ++c, st ^= 1) {
                    sync2(c == c_lo);
                    if (!active) continue;
                    const char* Ks = smem + st * STAGE;
                    const char* Vs = Ks + CK * 128;
                    f32x4 sc[8];
                    scores(Ks, c, q, sc);
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[t][r], c2, poff));
#define PM2_CASE(C)                                                                                     \
    case C:                                                                                             \
        if constexpr (C < CH0) {                                                                        \
            _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                             \
                constexpr int TT = (C < CH0 ? C : 0) * 8;                                               \
                const h2 lo = (h2){(_Float16)sc[t][0], (_Float16)sc[t][1]};                           \
                const h2 hi = (h2){(_Float16)sc[t][2], (_Float16)sc[t][3]};                           \
                pmax[TT + t][0] = __builtin_elementwise_max(pmax[TT + t][0], lo);                       \
                pmax[TT + t][1] = __builtin_elementwise_max(pmax[TT + t][1], hi);                       \
            }                                                                                           \
        }                                                                                               \
        break;
                    switch (c - c_lo) { PM2_CASE(0) PM2_CASE(1) PM2_CASE(2) PM2_CASE(3) default: break; }
#undef PM2_CASE
                    if (i0 == 0 && l16 == 0) {  // the wave that owns query row 0: the CLS row of P
                        float* dst = a.p0 + ((size_t)b * a.H + h) * a.Nk;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int j = c * CK + 16 * t + 4 * g + r;
                                if (j < a.Nk) dst[j] = sc[t][r];
                            }
                    }
                    bf16x4 vt[2][8];
                    auto v_reads = [&](int cc, bf16x4 (&d)[8]) {
                        const int vrow = 32 * cc + 4 * g + (l16 >> 2);
                        const int vkey = ((vrow >> 1) & 3) << 1;
                        const char* vr = Vs + vrow * 128 + 8 * (l16 & 1);
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            const unsigned ad = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(vr + (((2 * dt + ((l16 & 3) >> 1)) ^ vkey) << 4));
                            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d[2 * dt]) : "v"(ad) : "memory");
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(d[2 * dt + 1]) : "v"(ad) : "memory");
                        }
                    };
                    v_reads(0, vt[0]);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const bf16x8 pa = pack_lp8<F16>(sc[2 * cc], sc[2 * cc + 1]);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[cc & 1][0]), "+v"(vt[cc & 1][1]), "+v"(vt[cc & 1][2]), "+v"(vt[cc & 1][3]),
                                     "+v"(vt[cc & 1][4]), "+v"(vt[cc & 1][5]), "+v"(vt[cc & 1][6]), "+v"(vt[cc & 1][7]) :: "memory");
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            o[dt] = mfma_lp<F16>(cat_bf16x4(vt[cc & 1][2 * dt], vt[cc & 1][2 * dt + 1]), pa, o[dt], 0, 0, 0);
                        if (cc + 1 < 4) v_reads(cc + 1, vt[(cc + 1) & 1]);
                    }
                }
                if (active) {  // lane (i = l16, g) holds columns h*64 + 16dt + 4g .. +3 of row i
                    const int i = i0 + l16;
                    float* pr = opart + (((size_t)b * a.Nq + min(i, a.Nq - 1)) * a.H + h) * 64 + 4 * g;
                    if (half == 0) {
                        if (i < a.Nq) {
#pragma unroll
                            for (int dt = 0; dt < 4; ++dt) *(f32x4*)(pr + dt * 16) = o[dt];
                        }
                    } else {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) o[dt] += *(const f32x4*)(pr + dt * 16);
                        float n2 = 0.f;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) n2 += o[dt][r] * o[dt][r];
                        n2 = rows4_sum(n2);
                        if (i < a.Nq) {
                            bf16_t* orow = (bf16_t*)(a.out + (((size_t)b * a.Nq + i) * a.ldo + h * 64) * 2) + 4 * g;
#pragma unroll
                            for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(orow + dt * 16) = pack_lp4<F16>(o[dt]);
                            if (g == 0) a.onorm[((size_t)b * a.H + h) * a.Nq + i] = sqrtf(n2);
                        }
                    }
                }
            }
            if (active) {  // column sums of this half's head-max (vit.py:126-127)
                const int i = i0 + l16;
                const bool valid = i >= 1 && i < a.Nq;
                float* dst = a.colsum + ((size_t)b * a.nrt + rt) * a.Nk;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = row16_sum(valid ? (float)pmax[t][r >> 1][r & 1] : 0.f);
                        const int j = c_lo * CK + 16 * t + 4 * g + r;
                        if (l16 == 0 && j < a.Nk) dst[j] = v;
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < STG - 1; ++i) issue();
    for (int h = blk_z; h < a.H; h += hstep) {
        bf16x8 q[2];
        // ---- pass A: row maximum and sum over all keys (online, exact at the end) ----
        float m = -INFINITY, l = 0.f;
#pragma nounroll
        for (int c = 0; c < nch; ++c, st = (st + 1 == STG ? 0 : st + 1)) {
            step_sync(h, 0, c);
            if (c == 0) {  // the head's Q rows landed with this chunk
                q[0] = *(const bf16x8*)(qimg + l16 * 128 + (((0 + g) ^ (l16 & 7)) << 4));
                q[1] = *(const bf16x8*)(qimg + l16 * 128 + (((4 + g) ^ (l16 & 7)) << 4));
            }
            if (!active) continue;
            f32x4 sc[8];
            scores(smem + st * STAGE, c, q, sc);
            float cm = -INFINITY;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cm = fmaxf(cm, sc[t][r]);
            cm = rows4_max(cm);
            const float mn = fmaxf(m, cm);  // chunk 0 always holds a valid key, so mn is finite (m, mn: unscaled)
            const float off = -mn * c2;
            float cs = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs += __builtin_amdgcn_exp2f(fmaf(sc[t][r], c2, off));
            cs = rows4_sum(cs);
            l = l * __builtin_amdgcn_exp2f((m - mn) * c2) + cs;
            m = mn;
        }
        const float poff = -(m * c2 + __builtin_amdgcn_logf(l));  // P = exp2(c2 s + poff): normalised
        // ---- pass B: probabilities, head-max, CLS row, P.V ----
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma nounroll
        for (int c = 0; c < nch; ++c, st = (st + 1 == STG ? 0 : st + 1)) {  // (a runtime loop: unrolled, the per-chunk DMA addresses get hoisted and spilled)
            step_sync(h, 1, c);
            if (!active) continue;
            const char* Ks = smem + st * STAGE;
            const char* Vs = Ks + CK * 128;
            f32x4 sc[8];
            scores(Ks, c, q, sc);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[t][r], c2, poff));
            if constexpr (SCORES) {
                // head-max of this chunk's 8 key tiles: pmax is indexed statically (registers), so the chunk selects a case
#define PM_CASE(C)                                                                                      \
    case C:                                                                                             \
        if constexpr (C < NCH) {                                                                        \
            _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                             \
                constexpr int TT = (C < NCH ? C : 0) * 8;                                               \
                const h2 lo = (h2){(_Float16)sc[t][0], (_Float16)sc[t][1]}; /* round-to-nearest: rtz would bias the column sums */                           \
                const h2 hi = (h2){(_Float16)sc[t][2], (_Float16)sc[t][3]};                           \
                pmax[TT + t][0] = __builtin_elementwise_max(pmax[TT + t][0], lo);                       \
                pmax[TT + t][1] = __builtin_elementwise_max(pmax[TT + t][1], hi);                       \
            }                                                                                           \
        }                                                                                               \
        break;
                switch (c) { PM_CASE(0) PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) default: break; }
#undef PM_CASE
                if (i0 == 0) {  // wave-uniform: the wave that owns query row 0
                    if (l16 == 0) {
                        float* dst = a.p0 + ((size_t)b * a.H + h) * a.Nk;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int j = c * CK + 16 * t + 4 * g + r;
                                if (j < a.Nk) dst[j] = sc[t][r];
                            }
                    }
                }
            }
            // 32-key sub-chunks: key(k-slot group g, e) = 32cc + 16(e>>2) + 4g + (e&3).  The transposing V reads are inline asm
            // with their own lgkmcnt wait: through the builtin the compiler cannot tell them from the LDS-DMA targets and puts
            // s_waitcnt vmcnt(0) in front of the first one - which would drain the chunks in flight on every step.
            bf16x4 vt[2][8];
            auto v_reads = [&](int cc, bf16x4 (&d)[8]) {
                const int vrow = 32 * cc + 4 * g + (l16 >> 2);
                const int vkey = ((vrow >> 1) & 3) << 1;
                const char* vr = Vs + vrow * 128 + 8 * (l16 & 1);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const unsigned ad = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(vr + (((2 * dt + ((l16 & 3) >> 1)) ^ vkey) << 4));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d[2 * dt]) : "v"(ad) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(d[2 * dt + 1]) : "v"(ad) : "memory");
                }
            };
            v_reads(0, vt[0]);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const bf16x8 pa = pack_lp8<F16>(sc[2 * cc], sc[2 * cc + 1]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[cc & 1][0]), "+v"(vt[cc & 1][1]), "+v"(vt[cc & 1][2]), "+v"(vt[cc & 1][3]),
                             "+v"(vt[cc & 1][4]), "+v"(vt[cc & 1][5]), "+v"(vt[cc & 1][6]), "+v"(vt[cc & 1][7]) :: "memory");
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = mfma_lp<F16>(cat_bf16x4(vt[cc & 1][2 * dt], vt[cc & 1][2 * dt + 1]), pa, o[dt], 0, 0, 0);  // O^T: row d = 4g+r, col i
                if (cc + 1 < 4) v_reads(cc + 1, vt[(cc + 1) & 1]);
            }
        }
        if (active) {  // lane (i = l16, g) holds columns h*64 + 16dt + 4g .. +3 of row i
            const int i = i0 + l16;
            float n2 = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) n2 += o[dt][r] * o[dt][r];
            if constexpr (SCORES) n2 = rows4_sum(n2);
            if (i < a.Nq) {
                bf16_t* orow = (bf16_t*)(a.out + (((size_t)b * a.Nq + i) * a.ldo + h * 64) * 2) + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(orow + dt * 16) = pack_lp4<F16>(o[dt]);
                if constexpr (SCORES)
                    if (g == 0) a.onorm[((size_t)b * a.H + h) * a.Nq + i] = sqrtf(n2);
            }
        }
    }
    if constexpr (SCORES) {
        if (ngz >= 2) {
            // gridDim.z workgroups (2..4) share this row block, each with a share of the heads (launch_attn_bf16_large: launches
            // that would leave most SIMDs with one wave or none, or whose round count a finer split lowers).  The head-max is a max
            // - exact and order-free - so the shares are merged by whichever wave arrives LAST: agent-scope stores / loads (written
            // through and read past the L2s: the workgroups may sit on different XCDs) around one agent-scope ticket per (row
            // block, wave).
            const int G = ngz;
            const size_t slot = ((size_t)b * nrb + blk_x) * 4 + wave;
            unsigned* mine = a.hm_ws + ((slot * HM_MAX_GZ + blk_z) * (2 * HM_MAX_NT)) * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                __hip_atomic_store(mine + (2 * t) * 64, __builtin_bit_cast(unsigned, pmax[t][0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + (2 * t + 1) * 64, __builtin_bit_cast(unsigned, pmax[t][1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores are complete (at memory scope) before the ticket
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(a.hm_tick + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != G - 1) return;  // the last arriver writes the column sums
            if (lane == 0) __hip_atomic_store(a.hm_tick + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int z = 0; z < G; ++z) {
                if (z == blk_z) continue;
                const unsigned* theirs = a.hm_ws + ((slot * HM_MAX_GZ + z) * (2 * HM_MAX_NT)) * 64 + lane;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const unsigned u0 = __hip_atomic_load(theirs + (2 * t) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned u1 = __hip_atomic_load(theirs + (2 * t + 1) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pmax[t][0] = __builtin_elementwise_max(pmax[t][0], __builtin_bit_cast(h2, u0));
                    pmax[t][1] = __builtin_elementwise_max(pmax[t][1], __builtin_bit_cast(h2, u1));
                }
            }
        }
        if (active) {
            const int i = i0 + l16;
            const bool valid = i >= 1 && i < a.Nq;
            float* dst = a.colsum + ((size_t)b * a.nrt + rt) * a.Nk;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(valid ? (float)pmax[t][r >> 1][r & 1] : 0.f);
                    const int j = 16 * t + 4 * g + r;
                    if (l16 == 0 && j < a.Nk) dst[j] = v;
                }
        }
    }
}

// f32 scratch of the three-sweep order (one per device and stream, grow-only)
static float* opart_workspace(hipStream_t s, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<char*, size_t>> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto& e = pool[{dev, s}];
    if (e.second < bytes) {
        if (e.first) { (void)hipStreamSynchronize(s); (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
        if (hipMalloc((void**)&e.first, bytes) != hipSuccess) { (void)hipGetLastError(); e.first = nullptr; return nullptr; }
        e.second = bytes;
    }
    return (float*)e.first;
}

template <int NCH, bool SCORES, bool F16>
int launch_attn_bf16_large(const AttnArgs& a_in, hipStream_t s) {
    AttnArgs a = a_in;
    int gz = 1;
    if constexpr (SCORES && NCH == 8) {
        // 641..1024 keys with scores: the three-sweep order (HV = 2), two workgroups per CU (MADTP_ATTN_HV=0: the one-sweep order)
        static int hv_env = -1;
        if (hv_env < 0) { const char* e = getenv("MADTP_ATTN_HV"); hv_env = e ? atoi(e) : 1; }
        if (hv_env && a.H <= 16 && (a.Nk + 127) / 128 > NCH / 2) {
            float* op = opart_workspace(s, (size_t)a.B * a.Nq * a.H * 64 * sizeof(float));
            if (op) {
                a.o_part = op;
                const size_t lds2 = (size_t)2 * 2 * 128 * 128 + 4 * 2048 + (size_t)a.H * 128 * sizeof(float);
                MADTP_ENSURE_MAX_LDS((attn_bf16_large_kernel<NCH, true, 2, F16, 2>), lds2);
                hipLaunchKernelGGL((attn_bf16_large_kernel<NCH, true, 2, F16, 2>), dim3((a.Nq + 63) / 64, a.B, 1), dim3(256), lds2, s, a);
                MADTP_LAUNCH_CHECK();
                return 0;
            }
        }
    }
    const int wgs = ((a.Nq + 63) / 64) * a.B;
    if (!SCORES) {  // cross-attention against a long image sequence: few query rows, spread the heads over workgroups
        gz = wgs >= 512 ? 1 : (wgs >= 128 ? 4 : a.H);
        if (gz > a.H) gz = a.H;
    } else if (NCH * 8 <= HM_MAX_NT && wgs <= HM_MAX_WGS && a.H % 2 == 0) {
        // With scores a row block walks all heads (the head-max).  B x ceil(N/64) <= 256 workgroups of four waves leave the
        // 1024 SIMDs with one wave or none (VQA: 32 x 7), each bound by its own dependent instruction stream: 2..4 workgroups
        // per row block take a share of the heads each and merge their head-max at the end (MADTP_ATTN_HEAD_SPLIT=0: off).
        // The split that minimises rounds x heads per workgroup (512 workgroup slots at two per CU, 256 at one): e.g. 32 samples
        // x 605 keys = 320 row blocks: 2 groups = 640 workgroups = 2 rounds x 6 heads, 3 groups = 960 = 2 rounds x 4 heads.
        HmWorkspace hw;
        if (hm_workspace(s, hw)) {
            const int cap = 256 * ((NCH <= 5) ? 2 : 1);
            int best = 2, best_cost = 1 << 30;
            for (int z = 2; z <= HM_MAX_GZ; ++z) {
                if (a.H % z) continue;
                const int cost = ((wgs * z + cap - 1) / cap) * (a.H / z);
                if (cost < best_cost) { best_cost = cost; best = z; }
            }
            static int gz_env = -1;  // MADTP_ATTN_HEAD_GROUPS=2..4 forces the split (A/B runs)
            if (gz_env < 0) { const char* e = getenv("MADTP_ATTN_HEAD_GROUPS"); gz_env = e ? atoi(e) : 0; }
            if (gz_env >= 2 && gz_env <= HM_MAX_GZ && a.H % gz_env == 0) best = gz_env;
            gz = best; a.hm_ws = hw.ws; a.hm_tick = hw.tick;
        }
    }
    const dim3 grid((a.Nq + 63) / 64, a.B, gz);
    // Two stages.  The three-stage ring (STG = 3: two chunks in flight behind counted vmcnt waits, 104 KiB, one workgroup per
    // CU) was measured 4-8 % SLOWER at one workgroup per CU and 40 % slower where two would fit (tools/attn_large_bench.py): a
    // chunk step of a lone wave is bound by its own dependent instruction stream (~1.85 us against ~1.2 us with two waves per
    // SIMD), not by the DMA round trip.
    const size_t lds = (size_t)2 * 2 * 128 * 128 + 4 * 2048;  // two stages of K|V chunk images + the waves' Q rows
    MADTP_ENSURE_MAX_LDS((attn_bf16_large_kernel<NCH, SCORES, 2, F16>), lds);
    hipLaunchKernelGGL((attn_bf16_large_kernel<NCH, SCORES, 2, F16>), grid, dim3(256), lds, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

template <bool SCORES, bool F16 = false>
int dispatch_bf16_large(const AttnArgs& a, hipStream_t s) {
    const int nch = (a.Nk + 127) / 128;
    if (nch <= 3) return launch_attn_bf16_large<3, SCORES, F16>(a, s);
    if (nch <= 5) return launch_attn_bf16_large<5, SCORES, F16>(a, s);
    if (nch <= 8) return launch_attn_bf16_large<8, SCORES, F16>(a, s);
    return MADTP_E_SHAPE;
}

template <typename T, bool SCORES>
int dispatch_large(const AttnArgs& a, hipStream_t s) {
    const int nch = (a.Nk + 127) / 128;
    if (nch <= 5) return launch_attn_large<T, 5, SCORES>(a, s);
    if (nch <= 8) return launch_attn_large<T, 8, SCORES>(a, s);
    return MADTP_E_SHAPE;
}

template <typename T, int NT, bool SCORES>
int launch_attn(const AttnArgs& a_in, hipStream_t s) {
    constexpr int RB = 64 * (int)sizeof(T) + 16;
    const size_t lds = (size_t)2 * NT * 16 * RB;
    MADTP_ENSURE_MAX_LDS((attn_kernel<T, NT, SCORES>), lds);
    int gz = 1;
    AttnArgs a = a_in;
    const int wgs = ((a.Nq + 63) / 64) * a.B;
    if (!SCORES) {  // cross-attention has few query rows: spread heads over workgroups
        gz = wgs >= 512 ? 1 : (wgs >= 128 ? 4 : a.H);
        if (gz > a.H) gz = a.H;
    } else if (4 * NT <= 2 * HM_MAX_NT && wgs <= HM_MAX_WGS && a.H % 2 == 0) {
        // with scores a row block walks all heads: a launch that leaves the SIMDs with one wave or none (128 images x 2 row
        // blocks in the parity modes, 64 text samples) runs two workgroups per row block, half of the heads each
        HmWorkspace hw;
        if (hm_workspace(s, hw)) { gz = 2; a.hm_ws = hw.ws; a.hm_tick = hw.tick; }
    }
    hipLaunchKernelGGL((attn_kernel<T, NT, SCORES>), dim3((a.Nq + 63) / 64, a.B, gz), dim3(256), lds, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

template <int NT, bool SCORES>
int launch_attn_f16s(const AttnArgs& a_in, hipStream_t s) {
    constexpr int PITCH = NT <= 15 ? 160 : 144;
    const size_t lds = (size_t)(2 * NT * 16 + 2 * ((NT + 1) / 2) * 32) * PITCH;
    MADTP_ENSURE_MAX_LDS((attn_f16s_kernel<NT, SCORES>), lds);
    int gz = 1;
    AttnArgs a = a_in;
    const int wgs = ((a.Nq + 63) / 64) * a.B;
    if (!SCORES) {
        gz = wgs >= 512 ? 1 : (wgs >= 128 ? 4 : a.H);
        if (gz > a.H) gz = a.H;
    } else if (4 * NT <= 2 * HM_MAX_NT && wgs <= HM_MAX_WGS && a.H % 2 == 0) {
        HmWorkspace hw;
        if (hm_workspace(s, hw)) { gz = 2; a.hm_ws = hw.ws; a.hm_tick = hw.tick; }
    }
    hipLaunchKernelGGL((attn_f16s_kernel<NT, SCORES>), dim3((a.Nq + 63) / 64, a.B, gz), dim3(256), lds, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

template <bool SCORES>
int dispatch_nt_f16s(const AttnArgs& a, hipStream_t s) {
    const int nt = (a.Nk + 15) / 16;
    if (nt <= 2) return launch_attn_f16s<2, SCORES>(a, s);
    if (nt <= 4) return launch_attn_f16s<4, SCORES>(a, s);
    if (nt <= 6) return launch_attn_f16s<6, SCORES>(a, s);
    if (nt <= 8) return launch_attn_f16s<8, SCORES>(a, s);
    if (nt <= 10) return launch_attn_f16s<10, SCORES>(a, s);
    if (nt <= 13) return launch_attn_f16s<13, SCORES>(a, s);
    if (nt <= 16) return launch_attn_f16s<16, SCORES>(a, s);
    return MADTP_E_SHAPE;
}

template <typename T, bool SCORES>
int dispatch_nt(const AttnArgs& a, hipStream_t s) {
    const int nt = (a.Nk + 15) / 16;
    if (nt <= 2) return launch_attn<T, 2, SCORES>(a, s);  // <= 32 keys: the text encoder's self-attention (20-35 tokens)
    if (nt <= 4) return launch_attn<T, 4, SCORES>(a, s);
    if (nt <= 6) return launch_attn<T, 6, SCORES>(a, s);  // 81-96 keys: nine of the twelve ViT layers of the headline workload
    if (nt <= 8) return launch_attn<T, 8, SCORES>(a, s);
    if (nt <= 10) return launch_attn<T, 10, SCORES>(a, s);
    if (nt <= 12) return launch_attn<T, 12, SCORES>(a, s);
    if (nt <= 13) return launch_attn<T, 13, SCORES>(a, s);
    if (nt <= 16) return launch_attn<T, 16, SCORES>(a, s);
    return MADTP_E_SHAPE;
}

}  // namespace

extern "C" int madtp_attention(const void* q, const void* k, const void* v, void* out, const float* add_mask,
                               float* colsum_part, float* p0, float* onorm, int B, int H, int Nq, int Nk, int ldq,
                               int ldk, int ldv, int ldo, float scale, int io_dtype, void* stream) {
    return madtp_attention_indexed(q, k, v, nullptr, out, add_mask, colsum_part, p0, onorm, B, H, Nq, Nk, ldq, ldk, ldv, ldo,
                                   scale, io_dtype, stream);
}

static int attention_launch(const void* q, const void* k, const void* v, const int32_t* kv_batch_index, void* out,
                            const float* add_mask, const float* mask_qk, int ld_mask_qk, float* colsum_part, float* p0,
                            float* onorm, int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, float scale,
                            int io_dtype, void* stream, const int32_t* n_dev = nullptr, int dev_q_only = 0, int kv_block_rows = 0);

int madtp_i_attention(const void* q, const void* k, const void* v, void* out, float* colsum_part, float* p0, float* onorm, int B,
                      int H, int N, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype, const int32_t* n_dev,
                      void* stream) {
    if (N > 256) return MADTP_E_SHAPE;  // the two-pass long-sequence kernels keep host-side lengths
    return attention_launch(q, k, v, nullptr, out, nullptr, nullptr, 0, colsum_part, p0, onorm, B, H, N, N, ldq, ldk, ldv, ldo, scale,
                            io_dtype, stream, n_dev);
}
// the same with an additive key mask [B, n] (rows of the DEVICE length: the text encoders' compacted padding mask)
int madtp_i_attention_mask(const void* q, const void* k, const void* v, void* out, const float* add_mask, float* colsum_part, float* p0,
                           float* onorm, int B, int H, int N, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype,
                           const int32_t* n_dev, void* stream) {
    if (N > 256) return MADTP_E_SHAPE;
    return attention_launch(q, k, v, nullptr, out, add_mask, nullptr, 0, colsum_part, p0, onorm, B, H, N, N, ldq, ldk, ldv, ldo, scale,
                            io_dtype, stream, n_dev);
}
// attention of Nq queries per sample against the first Nk rows of that sample's K/V BLOCK of kv_block_rows rows (the decoder's
// self-attention cache [rows, Lmax, 2 dim]: incremental decoding, models/med.py:1071-1094); no scores, <= 256 keys
int madtp_i_attention_cached(const void* q, const void* k, const void* v, int kv_block_rows, void* out, int B, int H, int Nq, int Nk,
                             int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype, void* stream) {
    return attention_launch(q, k, v, nullptr, out, nullptr, nullptr, 0, nullptr, nullptr, nullptr, B, H, Nq, Nk, ldq, ldk, ldv, ldo, scale,
                            io_dtype, stream, nullptr, 0, kv_block_rows);
}
// cross-attention with *nq_dev query tokens per sample against Nk (host-side) keys of another sequence, optional K/V batch index
int madtp_i_attention_cross(const void* q, const void* k, const void* v, const int32_t* kv_batch_index, void* out, const float* add_mask,
                            int B, int H, int Nq_max, int Nk, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype,
                            const int32_t* nq_dev, void* stream) {
    if (Nk > 256 || Nq_max > 256) return MADTP_E_SHAPE;
    return attention_launch(q, k, v, kv_batch_index, out, add_mask, nullptr, 0, nullptr, nullptr, nullptr, B, H, Nq_max, Nk, ldq, ldk,
                            ldv, ldo, scale, io_dtype, stream, nq_dev, 1);
}

extern "C" int madtp_attention_indexed(const void* q, const void* k, const void* v, const int32_t* kv_batch_index, void* out,
                                       const float* add_mask, float* colsum_part, float* p0, float* onorm, int B, int H,
                                       int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype,
                                       void* stream) {
    return attention_launch(q, k, v, kv_batch_index, out, add_mask, nullptr, 0, colsum_part, p0, onorm, B, H, Nq, Nk, ldq, ldk,
                            ldv, ldo, scale, io_dtype, stream);
}

extern "C" int madtp_attention_qk_mask(const void* q, const void* k, const void* v, void* out, const float* add_mask,
                                       const float* mask_qk, int ld_mask_qk, float* colsum_part, float* p0, float* onorm, int B,
                                       int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype,
                                       void* stream) {
    if (mask_qk && (ld_mask_qk < Nk || Nk > 256)) return MADTP_E_SHAPE;  // the <= 256-key kernels carry the [Nq,Nk] mask
    return attention_launch(q, k, v, nullptr, out, add_mask, mask_qk, ld_mask_qk, colsum_part, p0, onorm, B, H, Nq, Nk, ldq, ldk,
                            ldv, ldo, scale, io_dtype, stream);
}

// Split-plane output request of the f16x3 layer calls (layers.hip): the NEXT attention launches of this thread with io_dtype
// MADTP_F16S write their context as f16-split planes (AttnArgs.split_dim; `out` = f16 planes, ldo in f16 elements) when they run on
// an f16s kernel, and say so through madtp_internal_attn_split_done() - a launch that falls back to another kernel writes f32 as
// before is impossible with a planes buffer, so the request is only made where the caller checked the same conditions.
static thread_local int t_split_dim = 0, t_split_done = 0;
int madtp_internal_attn_split_out(int split_dim) { t_split_dim = split_dim > 0 ? split_dim : 0; t_split_done = 0; return 0; }
int madtp_internal_attn_split_done() { return t_split_done; }
bool madtp_internal_attn_f16s_enabled() {
    static int f16s_env = -1;  // MADTP_ATTN_F16S=0: the f16x3 mode keeps its attention on the exact-f32 MFMA kernels (A/B runs)
    if (f16s_env < 0) { const char* e = getenv("MADTP_ATTN_F16S"); f16s_env = e ? atoi(e) : 1; }
    return f16s_env != 0;
}

static int attention_launch(const void* q, const void* k, const void* v, const int32_t* kv_batch_index, void* out,
                            const float* add_mask, const float* mask_qk, int ld_mask_qk, float* colsum_part, float* p0,
                            float* onorm, int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, float scale,
                            int io_dtype, void* stream, const int32_t* n_dev, int dev_q_only, int kv_block_rows) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return MADTP_E_BADARG;
    if (kv_block_rows && (kv_block_rows < Nk || Nk > 256 || colsum_part)) return MADTP_E_SHAPE;  // (the <= 256-key kernels without scores)
    if (kv_batch_index && colsum_part) return MADTP_E_BADARG;  // indexed K/V is a cross-attention feature
    // io_dtype MADTP_F16S: f32 storage, products as three f16 MFMA products of f16-split operands (the f16x3 precision mode)
    if (io_dtype != MADTP_F32 && io_dtype != MADTP_BF16 && io_dtype != MADTP_F16S && io_dtype != MADTP_F16) return MADTP_E_DTYPE;
    bool f16s = io_dtype == MADTP_F16S;
    const bool f16 = io_dtype == MADTP_F16;  // plain f16 operands: the bf16 kernels on the f16 MFMA
    int split_dim = 0;
    if (f16s) {
        f16s = madtp_internal_attn_f16s_enabled();
        io_dtype = MADTP_F32;
        split_dim = f16s ? t_split_dim : 0;
        if (t_split_dim && !f16s) return MADTP_E_BADARG;  // (the caller asks for planes only when the f16s kernels are on)
    } else if (t_split_dim) {
        return MADTP_E_BADARG;
    }
    if (colsum_part && (!p0 || !onorm || Nq != Nk)) return MADTP_E_BADARG;
    const int esz = (io_dtype == MADTP_BF16 || f16) ? 2 : 4;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (ldq * esz) % 16 || (ldk * esz) % 16 || (ldv * esz) % 16)
        return MADTP_E_ALIGN;
    if (Nk > 1024) return MADTP_E_SHAPE;
    AttnArgs a;
    a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.out = (char*)out; a.mask = add_mask;
    a.colsum = colsum_part; a.p0 = p0; a.onorm = onorm;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.nrt = (Nq + 15) / 16;
    a.scale = scale;
    a.kvidx = kv_batch_index;
    a.pair = 0; a.q2 = a.k2 = a.v2 = nullptr; a.out2 = nullptr; a.mask2 = nullptr; a.hm_ws = nullptr; a.hm_tick = nullptr; a.o_part = nullptr;
    a.mask_qk = mask_qk; a.ld_mqk = ld_mask_qk;
    a.n_dev = n_dev; a.dev_q_only = dev_q_only;
    a.kvb = kv_block_rows ? kv_block_rows : Nk;
    a.split_dim = split_dim; a.range_flag = split_dim ? madtp_internal_range_flag() : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const bool scores = colsum_part != nullptr;
    if (split_dim) {  // planes: 8-byte stores at f16 offsets
        if ((ldo % 4) || (split_dim % 4) || ((uintptr_t)out & 7)) return MADTP_E_ALIGN;
        t_split_done = 1;
        if (Nk > 256) return scores ? dispatch_large_f16s<true>(a, s) : dispatch_large_f16s<false>(a, s);
        return scores ? dispatch_nt_f16s<true>(a, s) : dispatch_nt_f16s<false>(a, s);
    }
    if (Nk > 256) {  // long sequences (384^2 / 480^2 images): two-pass kernels
        if (f16s && (ldo * 4) % 16 == 0 && aligned16(out))
            return scores ? dispatch_large_f16s<true>(a, s) : dispatch_large_f16s<false>(a, s);
        if (io_dtype == MADTP_F32) return scores ? dispatch_large<float, true>(a, s) : dispatch_large<float, false>(a, s);
        static int large_env = -1;  // MADTP_ATTN_LARGE_F32=1: bf16 storage on the exact-f32 MFMA kernel (A/B runs)
        if (large_env < 0) { const char* e = getenv("MADTP_ATTN_LARGE_F32"); large_env = e ? atoi(e) : 0; }
        if (f16) return scores ? dispatch_bf16_large<true, true>(a, s) : dispatch_bf16_large<false, true>(a, s);
        if (large_env) return scores ? dispatch_large<bf16_t, true>(a, s) : dispatch_large<bf16_t, false>(a, s);
        return scores ? dispatch_bf16_large<true>(a, s) : dispatch_bf16_large<false>(a, s);  // fast mode: bf16 MFMA, LDS-DMA ring
    }
    if (f16s && (ldo * 4) % 16 == 0 && aligned16(out))
        return scores ? dispatch_nt_f16s<true>(a, s) : dispatch_nt_f16s<false>(a, s);
    if (io_dtype == MADTP_F32) return scores ? dispatch_nt<float, true>(a, s) : dispatch_nt<float, false>(a, s);
    if ((ldk * 2) % 16 || (ldv * 2) % 16) return MADTP_E_ALIGN;
    if (scores && Nk <= 32 && !mask_qk && !n_dev) {  // short text sequences: one sample per workgroup, heads spread over the waves
        if (f16) hipLaunchKernelGGL(attn_bf16_small_kernel<true>, dim3(B), dim3(64 * SMALL_NW), 0, s, a);
        else hipLaunchKernelGGL(attn_bf16_small_kernel<false>, dim3(B), dim3(64 * SMALL_NW), 0, s, a);
        MADTP_LAUNCH_CHECK();
        return 0;
    }
    if (f16) return scores ? dispatch_nt_bf16<true, true>(a, s) : dispatch_nt_bf16<false, true>(a, s);
    return scores ? dispatch_nt_bf16<true>(a, s) : dispatch_nt_bf16<false>(a, s);
}

// Two attention problems of identical shape (no score side outputs) in one launch when they run on the bf16 kernel for
// <= 256 keys, two madtp_attention_indexed launches otherwise: the twin cross-attention branches of an NLVR text layer
// (nlvr_encoder.py:314-333: self0 against image 0, self1 against image 1) are 6.7 us launches of 768 small workgroups each.
extern "C" int madtp_attention_pair(const void* q0, const void* q1, const void* k0, const void* k1, const void* v0, const void* v1,
                                    const int32_t* kv_batch_index, void* out0, void* out1, const float* add_mask0,
                                    const float* add_mask1, int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo,
                                    float scale, int io_dtype, void* stream) {
    return madtp_i_attention_pair(q0, q1, k0, k1, v0, v1, kv_batch_index, out0, out1, add_mask0, add_mask1, B, H, Nq, Nk, ldq, ldk, ldv,
                                  ldo, scale, io_dtype, nullptr, stream);
}
// nq_dev != NULL: the sync-free text encoder - *nq_dev query tokens per sample (Nq is the worst case), <= 256 keys
int madtp_i_attention_pair(const void* q0, const void* q1, const void* k0, const void* k1, const void* v0, const void* v1,
                           const int32_t* kv_batch_index, void* out0, void* out1, const float* add_mask0, const float* add_mask1, int B,
                           int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype, const int32_t* nq_dev,
                           void* stream) {
    if (nq_dev && (Nk > 256 || Nq > 256)) return MADTP_E_SHAPE;
    if (!q0 || !q1 || !k0 || !k1 || !v0 || !v1 || !out0 || !out1 || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return MADTP_E_BADARG;
    static int pair_env = -1;  // MADTP_ATTN_PAIR=0: always two launches (A/B runs)
    if (pair_env < 0) { const char* e = getenv("MADTP_ATTN_PAIR"); pair_env = e ? atoi(e) : 1; }
    const bool one = pair_env && (io_dtype == MADTP_BF16 || io_dtype == MADTP_F16) && Nk <= 256 && (!add_mask0) == (!add_mask1) && aligned16(q0) &&
                     aligned16(q1) && aligned16(k0) && aligned16(k1) && aligned16(v0) && aligned16(v1) && (ldq * 2) % 16 == 0 &&
                     (ldk * 2) % 16 == 0 && (ldv * 2) % 16 == 0;
    if (!one) {
        const int rc = attention_launch(q0, k0, v0, kv_batch_index, out0, add_mask0, nullptr, 0, nullptr, nullptr, nullptr, B, H, Nq, Nk,
                                        ldq, ldk, ldv, ldo, scale, io_dtype, stream, nq_dev, nq_dev ? 1 : 0);
        if (rc) return rc;
        return attention_launch(q1, k1, v1, kv_batch_index, out1, add_mask1, nullptr, 0, nullptr, nullptr, nullptr, B, H, Nq, Nk, ldq, ldk,
                                ldv, ldo, scale, io_dtype, stream, nq_dev, nq_dev ? 1 : 0);
    }
    AttnArgs a;
    a.q = (const char*)q0; a.k = (const char*)k0; a.v = (const char*)v0; a.out = (char*)out0; a.mask = add_mask0;
    a.q2 = (const char*)q1; a.k2 = (const char*)k1; a.v2 = (const char*)v1; a.out2 = (char*)out1; a.mask2 = add_mask1;
    a.pair = 1;
    a.mask_qk = nullptr; a.ld_mqk = 0; a.hm_ws = nullptr; a.hm_tick = nullptr; a.n_dev = nq_dev; a.dev_q_only = 1; a.o_part = nullptr;
    a.colsum = nullptr; a.p0 = nullptr; a.onorm = nullptr; a.split_dim = 0; a.range_flag = nullptr;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.kvb = Nk;
    a.nrt = (Nq + 15) / 16;
    a.scale = scale;
    a.kvidx = kv_batch_index;
    if (io_dtype == MADTP_F16) return dispatch_nt_bf16<false, true>(a, (hipStream_t)stream);
    return dispatch_nt_bf16<false>(a, (hipStream_t)stream);
}

#ifdef MADTP_TS_TIMING
extern "C" int madtp_debug_read_attn_ts(long long* out) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_dbg), sizeof(long long) * 8);
}
#endif
