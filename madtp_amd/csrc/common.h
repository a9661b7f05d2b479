// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/madtp_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;  // raw bfloat16 bits

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hwbf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- f16-split ("x3") operands: fp32-accurate products on the f16 MFMA (dtype code MADTP_F16S) ---------------------------
// An f32 matrix X[R, K] is stored as f16 planes side by side in one row.
//   activations, 2 planes (row = [P0 | P1]: 2K f16 = the bytes of the f32 row):
//       P0 = f16(x),  P1 = f16((x - P0) * 2^11)            ->  x = P0 + 2^-11 P1 to ~2^-24 relative (|x| < 65504)
//   weights, pre-scaled per tensor by a power of two (w~ = w 2^s, max|w~| in (2^13, 2^14]), 2 planes (row = [Q0 | Q1]):
//       Q0 = f16(w~),  Q1 = f16(w~ - Q0)     (Q0 2^-11, formed in registers by the GEMM, is a normal f16 number for every
//                                              weight above 2^-17 of the tensor's maximum)
// so that   x w~ = P0 Q1 + P0 Q0 + P1 (Q0 2^-11)   (+ 2^-11 P1 Q1, dropped: < 2^-23 relative)
// is THREE f16 MFMA products with exact partial products and f32 accumulation: 3/16 of the cost of the exact-f32 MFMA.
constexpr float F16S_LO_SCALE = 2048.0f;  // 2^11
__device__ __forceinline__ void split_f16x8(f32x4 a, f32x4 b, u32x4& p0, u32x4& p1) {
    const f32x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const f16x8 h = __builtin_convertvector(v, f16x8);
    const f32x8 r = (v - __builtin_convertvector(h, f32x8)) * F16S_LO_SCALE;  // v - h is exact in f32
    const f16x8 l = __builtin_convertvector(r, f16x8);
    p0 = __builtin_bit_cast(u32x4, h);
    p1 = __builtin_bit_cast(u32x4, l);
}
__device__ __forceinline__ void split_f16x4(f32x4 v, f16x4& h, f16x4& l) {
    h = __builtin_convertvector(v, f16x4);
    l = __builtin_convertvector((v - __builtin_convertvector(h, f32x4)) * F16S_LO_SCALE, f16x4);
}

// round-to-nearest-even through the hardware converter (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ bf16x8 pack_bf16x8(f32x4 lo, f32x4 hi) {
    f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, __builtin_convertvector(v, hwbf16x8));
}
__device__ __forceinline__ bf16x4 pack_bf16x4(f32x4 v) {
    return __builtin_bit_cast(bf16x4, __builtin_convertvector(v, hwbf16x4));
}
// ---- plain f16 operands (dtype code MADTP_F16): the 2-byte storage of the bf16 kernels with IEEE half elements ------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// (pairwise: one v_cvt_pk_f16_f32 - round to nearest even - per two values; the 8-wide convertvector came out as a scalar
//  v_cvt_f16_f32 per value plus packing)
__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
}
__device__ __forceinline__ bf16x8 pack_f16x8(f32x4 lo, f32x4 hi) {
    const u32x4 r = {pack_f16x2(lo[0], lo[1]), pack_f16x2(lo[2], lo[3]), pack_f16x2(hi[0], hi[1]), pack_f16x2(hi[2], hi[3])};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x4 pack_f16x4(f32x4 v) {
    const u32x2 r = {pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3])};
    return __builtin_bit_cast(bf16x4, r);
}
template <bool F16> __device__ __forceinline__ bf16x8 pack_lp8(f32x4 lo, f32x4 hi) { return F16 ? pack_f16x8(lo, hi) : pack_bf16x8(lo, hi); }
template <bool F16> __device__ __forceinline__ bf16x4 pack_lp4(f32x4 v) { return F16 ? pack_f16x4(v) : pack_bf16x4(v); }
// Range flag of the f16 formats (include/madtp_hip.h madtp_range_status): producers OR "not (|v| < 65520)" - true for NaN and for
// everything that rounds to an f16 infinity - over the values they convert and store 1 to the flag when it is set.
int* madtp_internal_range_flag();  // host side: device-visible pointer to the pinned flag (norm.hip)
__device__ __forceinline__ bool f16_range_bad(float v) { return !(fabsf(v) < 65520.0f); }
__device__ __forceinline__ bool f16_range_bad(f32x4 v) {
    return f16_range_bad(v[0]) | f16_range_bad(v[1]) | f16_range_bad(v[2]) | f16_range_bad(v[3]);
}
__device__ __forceinline__ void f16_range_raise(int* flag, bool bad) {
    if (flag && bad) *(volatile int*)flag = 1;
}
// the cheap form for register-resident tiles (GEMM epilogues): running maximum of |v| (one v_max per value), tested once.  A NaN
// does not survive fmaxf - but a NaN in a GEMM output needs a non-finite operand, whose own producer has raised the flag.
__device__ __forceinline__ float f16_range_acc(float mx, f32x4 v) {
    mx = fmaxf(fmaxf(mx, fabsf(v[0])), fabsf(v[1]));  // (v_max3_f32 with |.| source modifiers)
    return fmaxf(fmaxf(mx, fabsf(v[2])), fabsf(v[3]));
}

// ds_read_b64_tr_b16: within each 16-lane group, lane 4r+q supplies the LDS address of the 8-byte piece (row r, columns
// 4q..4q+3) of a 4x16 bf16 block and lane i receives column i of that block (rows 0..3) - verified on MI355X with
// tools/probes/probe_tr.hip.  It turns a ROW-major LDS image into MFMA operands whose k index runs along the rows.
__device__ __forceinline__ bf16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}
__device__ __forceinline__ bf16x8 cat_bf16x4(bf16x4 a, bf16x4 b) { return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a full release/acquire fence, which on gfx950
// drains EVERY outstanding vector-memory operation (s_waitcnt vmcnt(0)) - including global loads and LDS-DMAs a kernel
// issued ahead on purpose.  This one waits for the wave's LDS operations (lgkmcnt) and leaves vmcnt alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// Exchanges between the 16-lane rows of a wave on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap) instead of
// ds_bpermute: no LDS-pipe round trip and no lgkmcnt wait.  xor16_pair / xor32_pair return (a, b) with {a, b} =
// {v[lane], v[lane ^ 16 (32)]} in some order - enough for the commutative reductions below, whose results are bit-identical
// to the __shfl_xor butterflies they replace.
__device__ __forceinline__ void xor16_pair(float v, float& a, float& b) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xor32_pair(float v, float& a, float& b) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// reductions over the 4 lanes l, l^16, l^32, l^48 (the four row groups of an MFMA 16x16 tile), xor-16 step first
__device__ __forceinline__ float rows4_max(float v) {
    float a, b;
    xor16_pair(v, a, b);
    v = fmaxf(a, b);
    xor32_pair(v, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float rows4_sum(float v) {
    float a, b;
    xor16_pair(v, a, b);
    v = a + b;
    xor32_pair(v, a, b);
    return a + b;
}

// butterfly reductions over the 64-lane wave (xor 32 and 16 on the swaps above, the rest through ds_bpermute)
__device__ __forceinline__ float wave_sum(float v) {
    float a, b;
    xor32_pair(v, a, b);
    v = a + b;
    xor16_pair(v, a, b);
    v = a + b;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    float a, b;
    xor32_pair(v, a, b);
    v = fmaxf(a, b);
    xor16_pair(v, a, b);
    v = fmaxf(a, b);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
    float a, b;
    xor32_pair(v, a, b);
    v = fminf(a, b);
    xor16_pair(v, a, b);
    v = fminf(a, b);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce across the 16 lanes that share lane>>4 (an MFMA 16x16 column group)
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE setting: each call site remembers the devices it has set it on
// (bit per device ordinal), so a process that drives several GPUs configures every one of them.
#define MADTP_ENSURE_MAX_LDS(FN, BYTES)                                                                                      \
    do {                                                                                                                     \
        static std::atomic<unsigned long long> done__{0};                                                                    \
        int dev__ = 0;                                                                                                       \
        (void)hipGetDevice(&dev__);                                                                                          \
        const unsigned long long bit__ = 1ull << (dev__ & 63);                                                               \
        if (!(done__.load(std::memory_order_acquire) & bit__)) {                                                             \
            hipError_t e__ = hipFuncSetAttribute((const void*)(FN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)); \
            if (e__ != hipSuccess) return (int)e__;                                                                          \
            done__.fetch_or(bit__, std::memory_order_release);                                                               \
        }                                                                                                                    \
    } while (0)

// ---- device-side token counts (the sync-free encoder path; layers.hip madtp_vit_encoder_async) --------
// A kernel that takes a DevN reads the size it names from DEVICE memory when p != nullptr: value = mul * p[0] + add, written by an
// earlier kernel of the same stream (token_score publishes the layer's k and the next layer's token count).  Launch geometry and
// buffers are sized by the host for the worst case (the unpruned sequence); workgroups past the actual size exit or idle.
struct DevN { const int32_t* p; int mul, add; };
__host__ __device__ __forceinline__ int devn(const DevN& d, int host_value) {
#if defined(__HIP_DEVICE_COMPILE__)
    return d.p ? d.mul * *d.p + d.add : host_value;
#else
    return host_value;
#endif
}
// per-layer record of the device-side pruning decision: dims[l] = {tokens per sample entering layer l, k = max_b count of the
// layer, k applied (0 = not pruned), tokens per sample leaving it}
constexpr int DIMS_STRIDE = 4;

#define MADTP_LAUNCH_CHECK()                          \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- LayerNorm of one row held by one wave (shared by norm.hip and the gather+LayerNorm fusion in prune.hip) ----
constexpr int LN_MAX_CHUNKS = 4;  // dim <= 1024

// normalise a row held as float4 chunks in registers; two-pass (mean, then centred variance) in f32
__device__ __forceinline__ void ln_row(float4 (&v)[LN_MAX_CHUNKS], int nchunk_lane, int dim, float eps, float& mean,
                                       float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c)
        if (c < nchunk_lane) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c)
        if (c < nchunk_lane) {
            const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    const float var = wave_sum(q) / (float)dim;
    rstd = 1.0f / sqrtf(var + eps);
}

// gamma / beta chunks of this lane, fetched BEFORE the row reductions (their latency hides under the x loads; a wave that
// normalises several rows fetches them once)
struct LnParams { float4 gm[LN_MAX_CHUNKS], bt[LN_MAX_CHUNKS]; };
__device__ __forceinline__ LnParams ln_params(const float* gamma, const float* beta, int lane, int dim) {
    LnParams p;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int col = (lane + 64 * c) * 4;
        const bool ok = col < dim;
        p.gm[c] = *(const float4*)(gamma + (ok ? col : 0));
        p.bt[c] = *(const float4*)(beta + (ok ? col : 0));
    }
    return p;
}

// ylp: row base of the low-precision copy - bf16 [dim], or (lp_f16s) the f16-split planes [P0 | P1] of 2*dim f16
// lp_fmt: MADTP_BF16 / MADTP_F16S (split planes) / MADTP_F16 (plain f16); range_flag: see f16_range_raise
__device__ __forceinline__ void ln_store(const float4 (&v)[LN_MAX_CHUNKS], int lane, int dim, float mean, float rstd,
                                         const LnParams& p, float* y32, bf16_t* ylp, int lp_fmt = MADTP_BF16, int* range_flag = nullptr) {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int col = (lane + 64 * c) * 4;
        if (col >= dim) break;
        const float4 gm = p.gm[c], bt = p.bt[c];
        float4 o;
        o.x = (v[c].x - mean) * rstd * gm.x + bt.x;
        o.y = (v[c].y - mean) * rstd * gm.y + bt.y;
        o.z = (v[c].z - mean) * rstd * gm.z + bt.z;
        o.w = (v[c].w - mean) * rstd * gm.w + bt.w;
        if (y32) *(float4*)(y32 + col) = o;
        if (ylp) {
            if (lp_fmt == MADTP_F16S) {
                f16x4 h, l;
                split_f16x4((f32x4){o.x, o.y, o.z, o.w}, h, l);
                *(f16x4*)(ylp + col) = h;
                *(f16x4*)(ylp + dim + col) = l;
                bad |= f16_range_bad((f32x4){o.x, o.y, o.z, o.w});
            } else if (lp_fmt == MADTP_F16) {
                *(bf16x4*)(ylp + col) = pack_f16x4((f32x4){o.x, o.y, o.z, o.w});
                bad |= f16_range_bad((f32x4){o.x, o.y, o.z, o.w});
            } else {
                *(bf16x4*)(ylp + col) = pack_bf16x4((f32x4){o.x, o.y, o.z, o.w});
            }
        }
    }
    f16_range_raise(range_flag, bad);
}
__host__ __device__ static inline int lp_row_mul(int lp_fmt) { return lp_fmt == MADTP_F16S ? 2 : 1; }  // row width of the lp copy in units of dim


