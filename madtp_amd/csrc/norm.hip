// HBM-bound row kernels: LayerNorm, BERT embeddings(+LN), patch im2col, CLS/pos assembly, small elementwise.
// One 64-lane wave owns one row; each lane moves 16-byte vectors (float4), so a wave instruction covers 1 KiB
// of a row and consecutive waves cover consecutive rows (fully coalesced).  Roofline: HBM bandwidth,
// algorithmic bytes per row = dim*4 read + dim*(4 and/or 2) written.
#include "common.h"
#include "internal.h"
#include <type_traits>

namespace {
struct F16Plain { unsigned short bits; };  // element tag: plain f16 (MADTP_F16)

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* y32, bf16_t* ylp,
                                                        int lp_fmt, int* range_flag, int rows_h, int dim, float eps, DevN rows_dev) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = devn(rows_dev, rows_h);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * dim;
    const LnParams prm = ln_params(gamma, beta, lane, dim);
    float4 v[LN_MAX_CHUNKS];
    int n = 0;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int col = (lane + 64 * c) * 4;
        if (col < dim) { v[c] = *(const float4*)(xr + col); n = c + 1; } else v[c] = make_float4(0, 0, 0, 0);
    }
    float mean, rstd;
    ln_row(v, n, dim, eps, mean, rstd);
    ln_store(v, lane, dim, mean, rstd, prm, y32 ? y32 + (size_t)row * dim : nullptr,
             ylp ? ylp + (size_t)row * dim * lp_row_mul(lp_fmt) : nullptr, lp_fmt, range_flag);
}

__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ wemb,
                                                         const float* __restrict__ pemb, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* y32, bf16_t* ylp,
                                                         int lp_fmt, int* range_flag, int rows, int L, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t id = ids[row];
    const float* wr = wemb + (size_t)id * dim;
    const float* pr = pemb + (size_t)(row % L) * dim;
    const LnParams prm = ln_params(gamma, beta, lane, dim);
    float4 v[LN_MAX_CHUNKS];
    int n = 0;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int col = (lane + 64 * c) * 4;
        if (col < dim) {
            const float4 a = *(const float4*)(wr + col), b = *(const float4*)(pr + col);
            v[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            n = c + 1;
        } else v[c] = make_float4(0, 0, 0, 0);
    }
    float mean, rstd;
    ln_row(v, n, dim, eps, mean, rstd);
    ln_store(v, lane, dim, mean, rstd, prm, y32 ? y32 + (size_t)row * dim : nullptr,
             ylp ? ylp + (size_t)row * dim * lp_row_mul(lp_fmt) : nullptr, lp_fmt, range_flag);
}

// y = LayerNorm(scale * (sum_s part[s] + bias) + residual): the fixed-order reduction of split-K partials fused with the
// bias / residual / LayerNorm that follows every small-M projection of the BERT layers.
__global__ __launch_bounds__(256) void splitk_ln_kernel(const float* __restrict__ part, int S, size_t pstride,
                                                        const float* __restrict__ bias, const float* __restrict__ residual,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* y32, bf16_t* ylp, int lp_fmt, int* range_flag, int rows, int dim, float eps,
                                                        float acc_scale, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const LnParams prm = ln_params(gamma, beta, lane, dim);
    float4 v[LN_MAX_CHUNKS];
    int n = 0;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int col = (lane + 64 * c) * 4;
        v[c] = make_float4(0, 0, 0, 0);
        if (col < dim) {
            float4 a = *(const float4*)(part + (size_t)row * dim + col);
            for (int s = 1; s < S; ++s) {
                const float4 p = *(const float4*)(part + s * pstride + (size_t)row * dim + col);
                a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
            }
            float4 b = make_float4(0, 0, 0, 0);
            if (bias) b = *(const float4*)(bias + col);
            // fmaf(a, 1, b) == a + b: the exact-f32 and bf16 paths are unchanged by the accumulator scale
            a.x = fmaf(a.x, acc_scale, b.x); a.y = fmaf(a.y, acc_scale, b.y); a.z = fmaf(a.z, acc_scale, b.z); a.w = fmaf(a.w, acc_scale, b.w);
            a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
            if (residual) {
                const float4 r = *(const float4*)(residual + (size_t)row * dim + col);
                a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
            }
            v[c] = a;
            n = c + 1;
        }
    }
    float mean, rstd;
    ln_row(v, n, dim, eps, mean, rstd);
    ln_store(v, lane, dim, mean, rstd, prm, y32 ? y32 + (size_t)row * dim : nullptr,
             ylp ? ylp + (size_t)row * dim * lp_row_mul(lp_fmt) : nullptr, lp_fmt, range_flag);
}

// one thread per 4 consecutive kx of one (b, patch, c, ky): reads 16 B of an image row, writes 16 B / 8 B
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, T* __restrict__ cols, int B, int S,
                                                       int P, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int kcols = 3 * P * P;
    const int g = S / P;
    const size_t e = i * 4;
    const int col = (int)(e % kcols);
    const size_t prow = e / kcols;
    const int kx = col % P, ky = (col / P) % P, c = col / (P * P);
    const int px = (int)(prow % g), py = (int)((prow / g) % g);
    const int b = (int)(prow / ((size_t)g * g));
    const float4 v = *(const float4*)(img + (((size_t)b * 3 + c) * S + (py * P + ky)) * S + px * P + kx);
    if constexpr (std::is_same<T, _Float16>::value) {  // f16-split planes [P0 | P1], row stride 2*kcols
        f16x4 h, l;
        split_f16x4((f32x4){v.x, v.y, v.z, v.w}, h, l);
        _Float16* o = cols + prow * 2 * kcols + col;
        *(f16x4*)o = h;
        *(f16x4*)(o + kcols) = l;
    } else if constexpr (std::is_same<T, F16Plain>::value) {  // plain f16 (MADTP_F16)
        *(bf16x4*)((bf16_t*)cols + e) = pack_f16x4((f32x4){v.x, v.y, v.z, v.w});
    } else {
        T* o = cols + e;
        o[0] = from_f32<T>(v.x); o[1] = from_f32<T>(v.y); o[2] = from_f32<T>(v.z); o[3] = from_f32<T>(v.w);
    }
}

__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ patches, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, float* __restrict__ x, int B, int np,
                                                       int dim4, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % dim4);
    const size_t row = i / dim4;
    const int t = (int)(row % (np + 1));
    const size_t b = row / (np + 1);
    const float4 p = ((const float4*)pos)[(size_t)t * dim4 + c];
    const float4 s = t == 0 ? ((const float4*)cls)[c] : ((const float4*)patches)[(b * np + (t - 1)) * dim4 + c];
    ((float4*)x)[i] = make_float4(s.x + p.x, s.y + p.y, s.z + p.z, s.w + p.w);
}

__global__ __launch_bounds__(256) void add_scale_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ o, float scale, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i];
    ((float4*)o)[i] = make_float4((x.x + y.x) * scale, (x.y + y.y) * scale, (x.z + y.z) * scale, (x.w + y.w) * scale);
}

// f32 [rows, K] -> f16-split activation planes [rows, 2K]; one thread per 4 consecutive columns
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ src, int ld_src, _Float16* __restrict__ dst,
                                                        int ld_dst, int K4, size_t total4_h, int* range_flag, DevN rows_dev) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total4 = rows_dev.p ? (size_t)devn(rows_dev, 0) * K4 : total4_h;
    if (i >= total4) return;
    const size_t row = i / K4;
    const int col = (int)(i - row * K4) * 4;
    const float4 v = *(const float4*)(src + row * ld_src + col);
    f16x4 h, l;
    split_f16x4((f32x4){v.x, v.y, v.z, v.w}, h, l);
    _Float16* o = dst + row * ld_dst + col;
    *(f16x4*)o = h;
    *(f16x4*)(o + K4 * 4) = l;
    f16_range_raise(range_flag, f16_range_bad((f32x4){v.x, v.y, v.z, v.w}));  // P0 would be an infinity: flag it (madtp_range_status)
}

// weight planes [Q0 | Q1] of w * inv_scale (common.h): Q0 = f16(w~), Q1 = f16(w~ - Q0)
__global__ __launch_bounds__(256) void split_f16_weight_kernel(const float* __restrict__ w, int ldw, _Float16* __restrict__ dst,
                                                               int K4, size_t total4, float inv_scale, int* range_flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const size_t row = i / K4;
    const int col = (int)(i - row * K4) * 4;
    const float4 v4 = *(const float4*)(w + row * ldw + col);
    const f32x4 v = (f32x4){v4.x, v4.y, v4.z, v4.w} * inv_scale;
    const f16x4 q0 = __builtin_convertvector(v, f16x4);
    const f32x4 q0f = __builtin_convertvector(q0, f32x4);
    const f16x4 q1 = __builtin_convertvector(v - q0f, f16x4);
    _Float16* o = dst + row * (size_t)(2 * K4 * 4) + col;
    *(f16x4*)o = q0;
    *(f16x4*)(o + K4 * 4) = q1;
    // a REUSED scale (runtime.prepare_linear: the previous version's 2^s) that the weight has outgrown: Q0 would be an infinity and
    // Q1 a NaN - flag it (madtp_range_status) instead of handing the GEMMs non-finite planes
    f16_range_raise(range_flag, f16_range_bad(v));
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *(const float4*)(src + i);
        *(bf16x4*)(dst + i) = pack_bf16x4((f32x4){v.x, v.y, v.z, v.w});
    } else {
        for (size_t j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
    }
}

// f32 -> 2-byte elements (bf16 or plain f16) of src * scale
template <bool F16>
__global__ __launch_bounds__(256) void cast_lp_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n, float scale,
                                                      int* range_flag, DevN n_dev = DevN{nullptr, 0, 0}) {
    if (n_dev.p) n = (size_t)devn(n_dev, 0);  // sync-free encoder path: the element count comes from the layer's device-side record
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const float4 v4 = *(const float4*)(src + i);
        const f32x4 v = (f32x4){v4.x, v4.y, v4.z, v4.w} * scale;
        *(bf16x4*)(dst + i) = pack_lp4<F16>(v);
        if constexpr (F16) f16_range_raise(range_flag, f16_range_bad(v));
    } else {
        for (size_t j = i; j < n; ++j) {
            const float v = src[j] * scale;
            if constexpr (F16) { ((_Float16*)dst)[j] = (_Float16)v; f16_range_raise(range_flag, f16_range_bad(v)); }
            else dst[j] = f32_to_bf16(v);
        }
    }
}

}  // namespace

static inline bool lp_dtype_ok(int d) { return d == MADTP_BF16 || d == MADTP_F16S || d == MADTP_F16; }

// The f16 range flag: one int of pinned (host-coherent) memory per process, written by device kernels, read by the host.
int* madtp_internal_range_flag() {
    static int* flag = [] {
        int* p = nullptr;
        if (hipHostMalloc((void**)&p, 64, hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return (int*)nullptr; }
        *p = 0;
        return p;
    }();
    return flag;
}
extern "C" int madtp_range_status(int reset, void* stream) {
    int* f = madtp_internal_range_flag();
    if (!f) return 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) (void)hipGetLastError();
    const int v = *(volatile int*)f;
    if (reset) *(volatile int*)f = 0;
    return v;
}

extern "C" int madtp_layernorm(const float* x, const float* gamma, const float* beta, float* y32, void* ylp, int lp_dtype,
                               int rows, int dim, float eps, void* stream) {
    return madtp_i_layernorm(x, gamma, beta, y32, ylp, lp_dtype, rows, dim, eps, DevN{nullptr, 0, 0}, stream);
}
int madtp_i_layernorm(const float* x, const float* gamma, const float* beta, float* y32, void* ylp, int lp_dtype, int rows, int dim,
                      float eps, DevN rows_dev, void* stream) {
    if (!x || !gamma || !beta || (!y32 && !ylp) || rows <= 0 || dim <= 0) return MADTP_E_BADARG;
    if (ylp && !lp_dtype_ok(lp_dtype)) return MADTP_E_DTYPE;
    if (dim % 4 || dim > 256 * LN_MAX_CHUNKS) return MADTP_E_SHAPE;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || (y32 && !aligned16(y32)) || (ylp && ((uintptr_t)ylp & 7)))
        return MADTP_E_ALIGN;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y32,
                       (bf16_t*)ylp, lp_dtype, madtp_internal_range_flag(), rows, dim, eps, rows_dev);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_splitk_ln(const float* part, int splits, const float* bias, const float* residual, const float* gamma,
                               const float* beta, float* y32, void* ylp, int lp_dtype, int rows, int dim, float eps,
                               float acc_scale, float scale, void* stream) {
    if (!part || !gamma || !beta || (!y32 && !ylp) || rows <= 0 || dim <= 0 || splits < 1) return MADTP_E_BADARG;
    if (ylp && !lp_dtype_ok(lp_dtype)) return MADTP_E_DTYPE;
    if (dim % 4 || dim > 256 * LN_MAX_CHUNKS) return MADTP_E_SHAPE;
    hipLaunchKernelGGL(splitk_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, splits,
                       (size_t)rows * dim, bias, residual, gamma, beta, y32, (bf16_t*)ylp, lp_dtype, madtp_internal_range_flag(), rows, dim, eps,
                       acc_scale, scale);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_bert_embed(const int64_t* ids, const float* word_emb, const float* pos_emb, const float* gamma,
                                const float* beta, float* y32, void* ylp, int lp_dtype, int B, int L, int dim, float eps,
                                void* stream) {
    if (!ids || !word_emb || !pos_emb || !gamma || !beta || (!y32 && !ylp) || B <= 0 || L <= 0) return MADTP_E_BADARG;
    if (ylp && !lp_dtype_ok(lp_dtype)) return MADTP_E_DTYPE;
    if (dim % 4 || dim > 256 * LN_MAX_CHUNKS) return MADTP_E_SHAPE;
    const int rows = B * L;
    hipLaunchKernelGGL(bert_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, word_emb, pos_emb,
                       gamma, beta, y32, (bf16_t*)ylp, lp_dtype, madtp_internal_range_flag(), rows, L, dim, eps);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_patchify(const float* img, void* cols, int B, int S, int P, int out_dtype, void* stream) {
    if (!img || !cols || B <= 0 || S <= 0 || P <= 0) return MADTP_E_BADARG;
    if (S % P || P % 4) return MADTP_E_SHAPE;
    const size_t total4 = (size_t)B * 3 * S * S / 4;
    const dim3 grid((unsigned)((total4 + 255) / 256));
    if (out_dtype == MADTP_F32)
        hipLaunchKernelGGL(patchify_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img, (float*)cols, B, S, P, total4);
    else if (out_dtype == MADTP_BF16)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)cols, B, S, P, total4);
    else if (out_dtype == MADTP_F16)
        hipLaunchKernelGGL(patchify_kernel<F16Plain>, grid, dim3(256), 0, (hipStream_t)stream, img, (F16Plain*)cols, B, S, P, total4);
    else if (out_dtype == MADTP_F16S)
        hipLaunchKernelGGL(patchify_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, img, (_Float16*)cols, B, S, P, total4);
    else
        return MADTP_E_DTYPE;
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_assemble_tokens(const float* patches, const float* cls, const float* pos, float* x, int B, int np,
                                     int dim, void* stream) {
    if (!patches || !cls || !pos || !x || B <= 0 || np <= 0 || dim <= 0) return MADTP_E_BADARG;
    if (dim % 4) return MADTP_E_SHAPE;
    const size_t total4 = (size_t)B * (np + 1) * (dim / 4);
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, patches,
                       cls, pos, x, B, np, dim / 4, total4);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_add_scale(const float* a, const float* b, float* out, float scale, size_t n, void* stream) {
    if (!a || !b || !out || n == 0) return MADTP_E_BADARG;
    if (n % 4) return MADTP_E_SHAPE;
    hipLaunchKernelGGL(add_scale_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                       scale, n / 4);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_cast_lp(const float* src, void* dst, size_t n, int lp_dtype, float scale, void* stream) {
    return madtp_i_cast_lp(src, dst, n, lp_dtype, scale, DevN{nullptr, 0, 0}, stream);
}
int madtp_i_cast_lp(const float* src, void* dst, size_t n, int lp_dtype, float scale, DevN n_dev, void* stream) {
    if (!src || !dst || n == 0) return MADTP_E_BADARG;
    if (!aligned16(src) || (((uintptr_t)dst) & 7u)) return MADTP_E_ALIGN;
    const dim3 grid((unsigned)(((n + 3) / 4 + 255) / 256));
    if (lp_dtype == MADTP_BF16)
        hipLaunchKernelGGL(cast_lp_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n, scale, (int*)nullptr, n_dev);
    else if (lp_dtype == MADTP_F16)
        hipLaunchKernelGGL(cast_lp_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n, scale, madtp_internal_range_flag(), n_dev);
    else
        return MADTP_E_DTYPE;
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_cast_bf16(const float* src, void* dst, size_t n, void* stream) {
    if (!src || !dst || n == 0) return MADTP_E_BADARG;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, n);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_split_f16(const float* src, int ld_src, void* dst, int ld_dst, int rows, int K, void* stream) {
    return madtp_i_split_f16(src, ld_src, dst, ld_dst, rows, K, DevN{nullptr, 0, 0}, stream);
}
int madtp_i_split_f16(const float* src, int ld_src, void* dst, int ld_dst, int rows, int K, DevN rows_dev, void* stream) {
    if (!src || !dst || rows <= 0 || K <= 0) return MADTP_E_BADARG;
    if (K % 4 || ld_src < K || ld_dst < 2 * K) return MADTP_E_SHAPE;
    if (!aligned16(src) || ld_src % 4 || ((uintptr_t)dst & 7) || ld_dst % 4) return MADTP_E_ALIGN;
    const size_t total4 = (size_t)rows * (K / 4);
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                       (_Float16*)dst, ld_dst, K / 4, total4, madtp_internal_range_flag(), rows_dev);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_split_f16_weight(const float* w, int ldw, void* dst, int n, int K, float inv_scale, void* stream) {
    if (!w || !dst || n <= 0 || K <= 0 || !(inv_scale > 0.f)) return MADTP_E_BADARG;
    if (K % 4 || ldw < K) return MADTP_E_SHAPE;
    if (!aligned16(w) || ldw % 4 || ((uintptr_t)dst & 7)) return MADTP_E_ALIGN;
    const size_t total4 = (size_t)n * (K / 4);
    hipLaunchKernelGGL(split_f16_weight_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw,
                       (_Float16*)dst, K / 4, total4, inv_scale, madtp_internal_range_flag());
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_abi_version(void) { return 29; }

extern "C" const char* madtp_strerror(int code) {
    switch (code) {
        case 0: return "ok";
        case MADTP_E_BADARG: return "bad argument (null pointer or non-positive size)";
        case MADTP_E_SHAPE: return "unsupported shape";
        case MADTP_E_DTYPE: return "unknown dtype";
        case MADTP_E_ALIGN: return "pointer / leading dimension not 16-byte aligned";
        case MADTP_E_BUSY: return "all host hand-over slots of the device are pending";
        case MADTP_E_RANGE: return "a value left the f16 range (|x| >= 65504 or NaN) in an f16 precision mode - use the fp32 / bf16 mode for this model";
        default: return code > 0 ? "HIP launch error (hipError_t)" : "unknown error";
    }
}
