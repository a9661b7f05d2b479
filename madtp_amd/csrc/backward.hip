// Backward of the pruned ViT block (first half): the kernels behind Block.forward's gradient
// (reference: loss.backward() in compress_nlvr_dtp.py:46-58 through models/vit.py:75-103 Attention.forward, :123-163
// Reduce_token and :183-207 Block.forward).  fp32 ("parity") arithmetic throughout; every reduction runs in a fixed order, so
// repeated calls give identical bits.
//
// What autograd differentiates in the reference, and therefore here:
//   * the two residual branches (LayerNorm, Linear, erf-GELU, softmax attention);
//   * the pruning step's VALUES: x_topk = gather(x, indices) and x_combine = sum_dropped w_t x_t with
//     w_t = I_t / (sum_dropped I + 1e-8) - `indices` (topk) carry no gradient, the merge weights do: through
//     Importance_score = (self_attn_w + token_attn_w + cls_attn) / 3 they reach the attention probabilities (head-max column
//     mass, vit.py:126-128; CLS row x head-diversity weights, :95-101) and the alignment logits (row max, :131-132);
//   * nothing through the temperature softmax / threshold / count (only compared, :137-145).
// The GEMMs of the backward (dgrad = dY W, wgrad = dY^T X) run on madtp_gemm's exact-f32 MFMA kernel with operands
// transposed by madtp_transpose_pad; this file holds everything else.
#include "common.h"

namespace {

constexpr int HD = 64;  // head dim

// fixed-order block sum over 256 threads (4 waves): wave butterfly, then the four partials in wave order
__device__ __forceinline__ float block_sum256(float v, float* red /* >= 4 floats of LDS */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- dst[c, r] = src[r, c] for r < R, c < C; zero elsewhere of dst[Cp, Rp] ---------------------------------------------
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ src, int ld_src, int R, int C,
                                                            float* __restrict__ dst, int ld_dst, int Rp, int Cp) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + ty + 8 * q, c = c0 + tx;
        tile[ty + 8 * q][tx] = (r < R && c < C) ? src[(size_t)r * ld_src + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + ty + 8 * q, r = r0 + tx;
        if (c < Cp && r < Rp) dst[(size_t)c * ld_dst + r] = tile[tx][ty + 8 * q];
    }
}

// ---- the same transpose straight into f16-split planes (round 5, the f16x3 backward's wgrad operands): dst[c, :] = the planes of
// src[:, c] - WEIGHT false: activation format [P0 | P1] (P1 = (x - P0) 2^11), true: weight format [Q0 | Q1] at scale 1 (common.h) -
// dst f16 [Cp, 2 Rp], zero beyond R / C.  64 x 64 tiles; one pass instead of transpose_pad + split_f16 (f32 written and re-read).
template <bool WEIGHT>
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ src, int ld_src, int R, int C,
                                                              _Float16* __restrict__ dst, int Rp, int Cp,
                                                              float* __restrict__ colsum_part) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = r0 + ty + 4 * q, c = c0 + tx;
        tile[ty + 4 * q][tx] = (r < R && c < C) ? src[(size_t)r * ld_src + c] : 0.0f;
    }
    __syncthreads();
    if (colsum_part) {  // the tile's column sums (the bias gradient's partials, rows in order): part[row tile, c]
        __shared__ float red[4][64];
        float cs = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) cs += tile[16 * ty + q][tx];
        red[ty][tx] = cs;
        __syncthreads();
        if (ty == 0 && c0 + tx < C) colsum_part[(size_t)blockIdx.x * C + c0 + tx] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    }
    const int r4 = (threadIdx.x & 15) * 4, cy = threadIdx.x >> 4;  // 16 row quads x 16 columns
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cl = cy + 16 * q, c = c0 + cl, r = r0 + r4;
        if (c >= Cp || r >= Rp) continue;
        const f32x4 v = (f32x4){tile[r4][cl], tile[r4 + 1][cl], tile[r4 + 2][cl], tile[r4 + 3][cl]};
        f16x4 h, l;
        if constexpr (WEIGHT) {
            h = __builtin_convertvector(v, f16x4);
            l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
        } else {
            split_f16x4(v, h, l);
        }
        _Float16* o = dst + (size_t)c * (2 * Rp) + r;
        *(f16x4*)o = h;
        *(f16x4*)(o + Rp) = l;
    }
}

// ---- both f16-split operand forms of a weight in one pass (round 5: a training step re-prepares every weight after the optimizer
// step - pad copy, zero fill, transpose, two splits: ~1500 small launches, ~10 ms of a b64 step): w f32 [N, K] (row stride ldw) x
// 2^s -> planes [.., 2 K] rows row0 .. row0 + N - 1 as [Q0 | Q1] (the forward's weight operand; K % 4 == 0) and planes_t
// [.., 2 Ntp] with (k, col0 + n) = Q0 / (k, Ntp + col0 + n) = Q1 of w[n, k] (dgrad's operand W^T).  Padding rows / columns are not
// touched (the caller's buffers are zero-initialised once and reused across parameter versions).
__global__ __launch_bounds__(256) void weight_planes_kernel(const float* __restrict__ w, int ldw, int N, int K, float inv_scale,
                                                            _Float16* __restrict__ planes, int row0, _Float16* __restrict__ planes_t,
                                                            int Ntp, int col0, int* range_flag) {
    __shared__ float tile[64][65];
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    bool bad = false;  // a reused scale the weight has outgrown (see split_f16_weight_kernel): raise the range flag
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int n = n0 + ty + 4 * q, k = k0 + tx;
        const float v = (n < N && k < K) ? w[(size_t)n * ldw + k] * inv_scale : 0.0f;
        tile[ty + 4 * q][tx] = v;
        bad |= f16_range_bad(v);
    }
    f16_range_raise(range_flag, bad);
    __syncthreads();
    const int c4 = (threadIdx.x & 15) * 4, ry = threadIdx.x >> 4;  // 16 quads x 16 rows
    if (planes) {  // row-major: row n, four consecutive k
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl = ry + 16 * q, n = n0 + nl, k = k0 + c4;
            if (n >= N || k >= K) continue;
            const f32x4 v = (f32x4){tile[nl][c4], tile[nl][c4 + 1], tile[nl][c4 + 2], tile[nl][c4 + 3]};
            const f16x4 h = __builtin_convertvector(v, f16x4);
            const f16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
            _Float16* o = planes + (size_t)(row0 + n) * (2 * K) + k;
            *(f16x4*)o = h;
            *(f16x4*)(o + K) = l;
        }
    }
    if (planes_t) {  // transposed: row k, four consecutive n
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kl = ry + 16 * q, k = k0 + kl, n = n0 + c4;
            if (k >= K || n >= N) continue;
            const f32x4 v = (f32x4){tile[c4][kl], tile[c4 + 1][kl], tile[c4 + 2][kl], tile[c4 + 3][kl]};
            const f16x4 h = __builtin_convertvector(v, f16x4);
            const f16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
            _Float16* o = planes_t + (size_t)k * (2 * Ntp) + col0 + n;
            if (n + 3 < N) {
                *(f16x4*)o = h;
                *(f16x4*)(o + Ntp) = l;
            } else {
                for (int e = 0; e < 4 && n + e < N; ++e) { o[e] = h[e]; o[Ntp + e] = l[e]; }
            }
        }
    }
}

// ---- column reductions: out[c] = sum_r f(r, c); XHAT: f = dy * (x - mean_r) * rstd_r (LayerNorm gamma grad), else f = dy --
// stage 1: grid (ceil(N/64), P): block (64 columns x 4 row lanes), rows r = chunk start + lane, +4, ... in order;
// stage 2: the P partials of a column in order.
template <bool XHAT>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ dy, int ld, const float* __restrict__ x,
                                                         int ldx, const float* __restrict__ stats, int M, int N, int rows_per,
                                                         float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int r_begin = blockIdx.y * rows_per, r_end = min(M, r_begin + rows_per);
    float s = 0.f;
    if (c < N)
        for (int r = r_begin + sub; r < r_end; r += 4) {
            float v = dy[(size_t)r * ld + c];
            if constexpr (XHAT) v *= (x[(size_t)r * ldx + c] - stats[2 * r]) * stats[2 * r + 1];
            s += v;
        }
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < N) part[(size_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// (64 columns x 4 partial lanes per block: lane q adds partials q, q + 4, ... in order - four independent loads in flight per
//  step instead of one chain of P dependent ones, which cost ~9 us for P = 64 - then the four lane sums in order)
__global__ __launch_bounds__(256) void col_reduce_final_kernel(const float* __restrict__ part, int P, int N, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N) {
        int p = sub;
        for (; p + 12 < P; p += 16) {
            const float a0 = part[(size_t)p * N + c], a1 = part[(size_t)(p + 4) * N + c], a2 = part[(size_t)(p + 8) * N + c],
                        a3 = part[(size_t)(p + 12) * N + c];
            s = (((s + a0) + a1) + a2) + a3;
        }
        for (; p < P; p += 4) s += part[(size_t)p * N + c];
    }
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < N) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- activations -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float u, int act) {
    switch (act) {
        case MADTP_ACT_GELU_ERF: {
            const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
            return cdf + u * 0.39894228040143267794f * expf(-0.5f * u * u);
        }
        case MADTP_ACT_QUICK_GELU: {
            const float s = 1.0f / (1.0f + expf(-1.702f * u));
            return s + 1.702f * u * s * (1.0f - s);
        }
        case MADTP_ACT_RELU: return u > 0.f ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}
__device__ __forceinline__ float act_val(float v, int act) {
    switch (act) {
        case MADTP_ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case MADTP_ACT_QUICK_GELU: return v / (1.0f + expf(-1.702f * v));
        case MADTP_ACT_RELU: return fmaxf(v, 0.0f);
        default: return v;
    }
}
// g = act(u) (g != NULL) and / or du = dg * act'(u) (dg, du != NULL)
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ u, const float* __restrict__ dg, float* __restrict__ g,
                                                  float* __restrict__ du, size_t n4, int act) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 uv = ((const float4*)u)[i];
    if (g) ((float4*)g)[i] = make_float4(act_val(uv.x, act), act_val(uv.y, act), act_val(uv.z, act), act_val(uv.w, act));
    if (du) {
        const float4 d = ((const float4*)dg)[i];
        ((float4*)du)[i] = make_float4(d.x * act_grad(uv.x, act), d.y * act_grad(uv.y, act), d.z * act_grad(uv.z, act), d.w * act_grad(uv.w, act));
    }
}

// ---- LayerNorm backward, one wave per row: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma (+ add) -----------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, const float* __restrict__ add,
                                                            float* __restrict__ dx, float* __restrict__ stats, int rows, int dim,
                                                            float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = (dim / 4 - lane + 63) / 64;
    float4 v[LN_MAX_CHUNKS], gd[LN_MAX_CHUNKS];
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c)
        if (c < nchunk) {
            const int col = (lane + 64 * c) * 4;
            v[c] = *(const float4*)(x + (size_t)row * dim + col);
            const float4 d = *(const float4*)(dy + (size_t)row * dim + col), gm = *(const float4*)(gamma + col);
            gd[c] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
        }
    float mean, rstd;
    ln_row(v, nchunk, dim, eps, mean, rstd);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c)
        if (c < nchunk) {
            s1 += (gd[c].x + gd[c].y) + (gd[c].z + gd[c].w);
            s2 += (gd[c].x * (v[c].x - mean) + gd[c].y * (v[c].y - mean)) + (gd[c].z * (v[c].z - mean) + gd[c].w * (v[c].w - mean));
        }
    const float c1 = wave_sum(s1) / (float)dim, c2 = wave_sum(s2) * rstd / (float)dim;  // mean(g), mean(g xhat)
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c)
        if (c < nchunk) {
            const int col = (lane + 64 * c) * 4;
            float4 o;
            o.x = rstd * (gd[c].x - c1 - (v[c].x - mean) * rstd * c2);
            o.y = rstd * (gd[c].y - c1 - (v[c].y - mean) * rstd * c2);
            o.z = rstd * (gd[c].z - c1 - (v[c].z - mean) * rstd * c2);
            o.w = rstd * (gd[c].w - c1 - (v[c].w - mean) * rstd * c2);
            if (add) {
                const float4 a = *(const float4*)(add + (size_t)row * dim + col);
                o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            }
            *(float4*)(dx + (size_t)row * dim + col) = o;
        }
    if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// ---- gather / merge backward (vit.py:153-161): one wave per row of dx [B, N, dim] ------------------------------------
//   dx[b,0] = dy[b,0]; kept t: dx[b,1+t] = dy[b,1+dst_pos]; dropped t: dx[b,1+t] = merge_w[b,t] dy[b,k+1],
//   dw[b,t] = <dy[b,k+1], x[b,1+t]> (0 for kept tokens)
__global__ __launch_bounds__(256) void token_gather_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const int32_t* __restrict__ dst_pos, const float* __restrict__ merge_w,
                                                               float* __restrict__ dx, float* __restrict__ dw, int B, int N, int k,
                                                               int dim) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * N) return;
    const int b = (int)(row / N), tok = (int)(row % N);
    const float* dyb = dy + (size_t)b * (k + 2) * dim;
    float* o = dx + (size_t)row * dim;
    if (tok == 0) {
        for (int c = lane * 4; c < dim; c += 256) *(float4*)(o + c) = *(const float4*)(dyb + c);
        return;
    }
    const int t = tok - 1, dp = dst_pos[(size_t)b * (N - 1) + t];
    if (dp >= 0) {
        const float* s = dyb + (size_t)(1 + dp) * dim;
        for (int c = lane * 4; c < dim; c += 256) *(float4*)(o + c) = *(const float4*)(s + c);
        if (lane == 0) dw[(size_t)b * (N - 1) + t] = 0.f;
    } else {
        const float w = merge_w[(size_t)b * (N - 1) + t];
        const float* s = dyb + (size_t)(k + 1) * dim;
        const float* xr = x + (size_t)row * dim;
        float acc = 0.f;
        for (int c = lane * 4; c < dim; c += 256) {
            const float4 d = *(const float4*)(s + c), xv = *(const float4*)(xr + c);
            *(float4*)(o + c) = make_float4(w * d.x, w * d.y, w * d.z, w * d.w);
            acc += (d.x * xv.x + d.y * xv.y) + (d.z * xv.z + d.w * xv.w);
        }
        acc = wave_sum(acc);
        if (lane == 0) dw[(size_t)b * (N - 1) + t] = acc;
    }
}

// ---- importance-score backward (vit.py:126-134 + :95-101), one workgroup per sample -----------------------------------
// in : dw [B,n] (grad of the merge weights), score [B,n], dst_pos, merge_w, colsum_part [B,nrt,N], p0 [B,H,N], onorm [B,H,N],
//      token_attn (strided [B,n,K])
// out: da [B,N] (grad of the UN-normalised head-max column mass a_j, j >= 1; da[b,0] = 0), dp0 [B,H,N] (grad of P[b,h,0,j]),
//      dnrm_scale [B,H,N] (grad of ||out[b,h,j,:]|| divided by that norm: d out += dnrm_scale * out), dta [B,n,K] dense
__global__ __launch_bounds__(256) void score_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ score,
                                                        const int32_t* __restrict__ dst_pos, const float* __restrict__ merge_w,
                                                        const float* __restrict__ colsum_part, int nrt, const float* __restrict__ p0,
                                                        const float* __restrict__ onorm, const float* __restrict__ ta, int ldt_row,
                                                        int ldt_batch, int K, float* __restrict__ da, float* __restrict__ dp0,
                                                        float* __restrict__ dnrm_scale, float* __restrict__ dta, int H, int N) {
    __shared__ float red[4];
    const int b = blockIdx.x, n = N - 1, tid = threadIdx.x;
    const float* cs = colsum_part + (size_t)b * nrt * N;
    // pass 1: sums
    float sS = 0.f, sC = 0.f, sA = 0.f, sT = 0.f;
    for (int j = tid; j < n; j += 256) {
        const size_t bj = (size_t)b * n + j;
        if (dst_pos[bj] < 0) { sS += score[bj]; sC += dw[bj] * merge_w[bj]; }
        float a = 0.f;
        for (int rt = 0; rt < nrt; ++rt) a += cs[(size_t)rt * N + 1 + j];
        sA += a;
        const float* row = ta + (size_t)b * ldt_batch + (size_t)j * ldt_row;
        float m = row[0];
        for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
        sT += m;
    }
    const float S = block_sum256(sS, red), Cw = block_sum256(sC, red), SA = block_sum256(sA, red), ST = block_sum256(sT, red);
    const float invS = 1.0f / (S + 1e-8f), invA = 1.0f / (SA + 1e-8f), invT = 1.0f / (ST + 1e-8f);
    // pass 2: coupling terms of the two normalisations: ca = sum_j g_j a_j/(SA+eps), ct likewise
    float sca = 0.f, sct = 0.f;
    for (int j = tid; j < n; j += 256) {
        const size_t bj = (size_t)b * n + j;
        const float g = dst_pos[bj] < 0 ? (dw[bj] - Cw) * invS * (1.0f / 3.0f) : 0.f;
        float a = 0.f;
        for (int rt = 0; rt < nrt; ++rt) a += cs[(size_t)rt * N + 1 + j];
        const float* row = ta + (size_t)b * ldt_batch + (size_t)j * ldt_row;
        float m = row[0];
        for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
        sca += g * a * invA;
        sct += g * m * invT;
    }
    const float CA = block_sum256(sca, red), CT = block_sum256(sct, red);
    if (tid == 0) da[(size_t)b * N] = 0.f;
    for (int h = tid; h < H; h += 256) { dp0[((size_t)b * H + h) * N] = 0.f; dnrm_scale[((size_t)b * H + h) * N] = 0.f; }
    for (int j = tid; j < n; j += 256) {
        const size_t bj = (size_t)b * n + j;
        const float g = dst_pos[bj] < 0 ? (dw[bj] - Cw) * invS * (1.0f / 3.0f) : 0.f;
        da[(size_t)b * N + 1 + j] = (g - CA) * invA;
        // alignment logits: gradient lands on the row maximum (first maximum, as torch.max)
        const float* row = ta + (size_t)b * ldt_batch + (size_t)j * ldt_row;
        float m = row[0];
        int am = 0;
        for (int c = 1; c < K; ++c)
            if (row[c] > m) { m = row[c]; am = c; }
        float* drow = dta + ((size_t)b * n + j) * K;
        const float dt = (g - CT) * invT;
        for (int c = 0; c < K; ++c) drow[c] = c == am ? dt : 0.f;
        // cls_attn[j] = sum_h P[h,0,j] hi[h,j], hi = nr_h / (sum_h nr_h + eps)
        float sn = 0.f;
        for (int h = 0; h < H; ++h) sn += onorm[((size_t)b * H + h) * N + 1 + j];
        const float invN = 1.0f / (sn + 1e-8f);
        float cc = 0.f;
        for (int h = 0; h < H; ++h) {
            const size_t o = ((size_t)b * H + h) * N + 1 + j;
            cc += g * p0[o] * onorm[o] * invN;
        }
        for (int h = 0; h < H; ++h) {
            const size_t o = ((size_t)b * H + h) * N + 1 + j;
            const float nr = onorm[o];
            dp0[o] = g * nr * invN;
            const float dnr = (g * p0[o] - cc) * invN;
            dnrm_scale[o] = nr > 0.f ? dnr / nr : 0.f;
        }
    }
}

// ---- dropout / DropPath (round 5: the training forward's stochastic regularisers, models/med.py:55,111,244,323 nn.Dropout at
// hidden_dropout_prob / attention_probs_dropout_prob = 0.1, models/vit.py:114,186,205 DropPath) -----------------------------------
// Counter-based masks: Philox4x32-10 keyed by the caller's 64-bit seed, counter = (element index / 4, site id); element i takes word
// i & 3 of its counter's output, keep = u >= p with u = (word >> 8) 2^-24.  Nothing is stored: the backward regenerates the mask
// of a site from (seed, site id).  tests/test_backward_gpu.py restates the generator in numpy.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
struct DropArgs { float p, inv_keep; unsigned long long seed, site; };
// keep factor (0 or 1 / (1 - p)) of element idx of a site
__device__ __forceinline__ float drop_factor(const DropArgs& d, unsigned long long idx) {
    unsigned o[4];
    const unsigned long long c = idx >> 2;
    philox4x32_10((unsigned)c, (unsigned)(c >> 32), (unsigned)d.site, (unsigned)(d.site >> 32), (unsigned)d.seed, (unsigned)(d.seed >> 32), o);
    const float u = (float)(o[idx & 3] >> 8) * (1.0f / 16777216.0f);
    return u >= d.p ? d.inv_keep : 0.f;
}
// y = residual + x * keep / (1 - p); per_sample > 0: one draw per run of per_sample elements (DropPath: a sample's whole branch)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ residual, float* __restrict__ y,
                                                      size_t n4, size_t per_sample, DropArgs d) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    float f[4];
    if (per_sample) {
        f[0] = f[1] = f[2] = f[3] = drop_factor(d, (4 * i4) / per_sample);
    } else {
        unsigned o[4];
        philox4x32_10((unsigned)i4, (unsigned)(i4 >> 32), (unsigned)d.site, (unsigned)(d.site >> 32), (unsigned)d.seed, (unsigned)(d.seed >> 32), o);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = (float)(o[e] >> 8) * (1.0f / 16777216.0f) >= d.p ? d.inv_keep : 0.f;
    }
    const float4 v = ((const float4*)x)[i4];
    float4 r = make_float4(v.x * f[0], v.y * f[1], v.z * f[2], v.w * f[3]);
    if (residual) { const float4 a = ((const float4*)residual)[i4]; r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w; }
    ((float4*)y)[i4] = r;
}

// ---- attention backward ---------------------------------------------------------------------------------------------
// q/k/v: rows (b*N + i) of a [B*N, ld] f32 matrix, head h at columns [h*64, h*64+64) of each operand's base pointer.
struct AttnBwdArgs {
    const float *q, *k, *v; int ld;      // forward operands
    const float* dout; int ldo;          // grad of the attention output [B*N, H*64]
    const float* out; int ldout;         // forward attention output (for the norm term), may be NULL with dnrm_scale
    const float* dnrm_scale;             // [B,H,N] or NULL
    const float* da;                     // [B,N] or NULL: grad of a_j = sum_{i>=1} max_h P[h,i,j]
    const float* dp0;                    // [B,H,N] or NULL: grad of P[h,0,j]
    float* P; float* dS;                 // scratch [B,H,N,N] each
    unsigned char* hm;                   // scratch [B,N,N]: argmax_h P[b,h,i,j]
    float *dq, *dk, *dv; int ldd;        // outputs, same layout as q/k/v
    int B, H, N; float scale;            // N = Nq (query rows per sample)
    const float* key_mask;               // [B,Nk] additive key mask (BERT padding mask, med.py:197-199) or NULL
    const float* mask_qk; int ld_mqk;    // [N, ld_mqk] additive mask over (query, key) pairs (the decoder's causal mask) or NULL
    float* dp_out;                       // [B,H,N,Nk] or NULL: the gradient of the attention probabilities themselves (Grad-CAM hook)
    DropArgs drop;                       // attention_probs dropout of the forward (p = 0: none): out = (P o mask / (1 - p)) V
    int Nk; int ldk; int lddk;           // keys per sample, leading dimension of k / v and of dk / dv (cross-attention: the keys
                                         // come from another sequence and projection; self-attention: Nk = N, ldk = ld, lddk = ldd)
};

// The three attention-backward kernels form their matrix products - S = Q K^T, dP = dO V^T, dQ = dS K, dV = P^T dO, dK = dS^T Q -
// on the exact-f32 MFMA (16x16x4 f32; round 5: the first versions were VALU loops with one LDS broadcast read per FMA and cost a
// f16x3 training step 12 of its ~100 ms).  A operand: lane (l16, g) holds row l16, contraction index g; B operand: contraction
// index g, column l16; a float4 along the contraction dimension feeds four MFMA steps (the same permutation of the contraction
// index on both operands).  Summation orders are fixed (by the instruction, tiles in ascending order, waves' partials in order).
typedef float af_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ af_f32x4 mfma_f32_4(float4 a, float4 b, af_f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
}
__host__ __device__ constexpr int attn_ns(int nk) { return (nk + 15) & ~15; }  // LDS row stride of a 16 x Nk score tile

// P[b,h,i,:] = softmax_j(scale q_i . k_j) for 16 query rows per workgroup; a wave forms the 16 x 16 score tiles of key tiles
// wave, wave + 4, ...
__global__ __launch_bounds__(256) void attn_probs_kernel(AttnBwdArgs a) {
    extern __shared__ float sm[];
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, i0 = blockIdx.y * 16, NQ = a.N, N = a.Nk, tid = threadIdx.x;
    const int NS = attn_ns(N);
    float* Qs = sm;                 // [16][64]
    float* S = sm + 16 * HD;        // [16][NS]
    for (int e = tid; e < 16 * HD; e += 256) {
        const int r = e >> 6, d = e & 63, i = min(i0 + r, NQ - 1);
        Qs[e] = a.q[(size_t)(b * NQ + i) * a.ld + h * HD + d];
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
        float4 qa[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) qa[c] = *(const float4*)&Qs[l16 * HD + 16 * c + 4 * g];
        for (int jt = wave; jt * 16 < N; jt += 4) {
            const int jj = jt * 16 + l16, j = min(jj, N - 1);
            const float* kp = a.k + (size_t)(b * N + j) * a.ldk + h * HD + 4 * g;
            af_f32x4 acc = (af_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = mfma_f32_4(qa[c], *(const float4*)(kp + 16 * c), acc);
            const float km = a.key_mask ? a.key_mask[(size_t)b * N + j] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // acc[r] = <q row 4 g + r, k row jj>
                const int row = 4 * g + r;
                float sv = a.key_mask ? fmaf(acc[r], a.scale, km) : acc[r] * a.scale;
                if (a.mask_qk) sv += a.mask_qk[(size_t)min(i0 + row, NQ - 1) * a.ld_mqk + j];
                S[row * NS + jj] = sv;
            }
        }
    }
    __syncthreads();
    const int r = tid >> 4, l = tid & 15;  // 16 lanes per row
    float m = -INFINITY;
    for (int j = l; j < N; j += 16) m = fmaxf(m, S[r * NS + j]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int j = l; j < N; j += 16) { const float e = expf(S[r * NS + j] - m); S[r * NS + j] = e; sum += e; }
    sum = row16_sum(sum);
    if (i0 + r < NQ) {
        float* Pr = a.P + ((size_t)bh * NQ + i0 + r) * N;
        for (int j = l; j < N; j += 16) Pr[j] = S[r * NS + j] / sum;
    }
}

// hm[b,i,j] = argmax_h P[b,h,i,j] (first maximum)
__global__ __launch_bounds__(256) void attn_headmax_kernel(AttnBwdArgs a) {
    const int b = blockIdx.x, i = blockIdx.y, N = a.N;
    for (int j = threadIdx.x; j < N; j += 256) {
        float m = a.P[(((size_t)b * a.H) * N + i) * N + j];
        int am = 0;
        for (int h = 1; h < a.H; ++h) {
            const float p = a.P[(((size_t)b * a.H + h) * N + i) * N + j];
            if (p > m) { m = p; am = h; }
        }
        a.hm[((size_t)b * N + i) * N + j] = (unsigned char)am;
    }
}

// rows pass: dP = dO V^T (+ score terms), dS = P (dP - rowsum(P dP)), dQ = scale dS K
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(AttnBwdArgs a) {
    extern __shared__ float sm[];
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, i0 = blockIdx.y * 16, NQ = a.N, N = a.Nk, tid = threadIdx.x;
    const int NS = attn_ns(N);
    const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    float* dOs = sm;                  // [16][64]
    float* Ps = sm + 16 * HD;         // [16][NS]
    float* Ds = Ps + 16 * NS;         // [16][NS]: dP, then dS (zero beyond key N)
    float* red = Ds + 16 * NS;        // [4][16][64]: the waves' partial dQ tiles
    for (int e = tid; e < 16 * HD; e += 256) {
        const int r = e >> 6, d = e & 63, i = i0 + r;
        float v = 0.f;
        if (i < NQ) {
            v = a.dout[(size_t)(b * NQ + i) * a.ldo + h * HD + d];
            if (a.dnrm_scale) v += a.dnrm_scale[((size_t)b * a.H + h) * NQ + i] * a.out[(size_t)(b * NQ + i) * a.ldout + h * HD + d];
        }
        dOs[e] = v;
    }
    __syncthreads();
    {
        float4 oa[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) oa[c] = *(const float4*)&dOs[l16 * HD + 16 * c + 4 * g];
        for (int jt = wave; jt * 16 < N; jt += 4) {
            const int jj = jt * 16 + l16, j = min(jj, N - 1);
            const float* vp = a.v + (size_t)(b * N + j) * a.ldk + h * HD + 4 * g;
            af_f32x4 acc = (af_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = mfma_f32_4(oa[c], *(const float4*)(vp + 16 * c), acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // acc[r] = <dO row 4 g + r, v row jj>
                const int row = 4 * g + r, i = i0 + row;
                float dp = 0.f, p = 0.f;
                if (i < NQ && jj < N) {
                    dp = acc[r];
                    if (a.drop.p > 0.f) dp *= drop_factor(a.drop, ((unsigned long long)bh * NQ + i) * N + jj);  // d P = mask o d(P dropped)
                    p = a.P[((size_t)bh * NQ + i) * N + jj];
                    if (jj >= 1) {
                        if (i == 0) { if (a.dp0) dp += a.dp0[((size_t)b * a.H + h) * N + jj]; }
                        else if (a.da && a.hm[((size_t)b * N + i) * N + jj] == h) dp += a.da[(size_t)b * N + jj];
                    }
                    if (a.dp_out) a.dp_out[((size_t)bh * NQ + i) * N + jj] = dp;
                }
                Ps[row * NS + jj] = p;
                Ds[row * NS + jj] = dp;
            }
        }
    }
    __syncthreads();
    {
        const int r = tid >> 4, l = tid & 15;
        float dsum = 0.f;
        for (int j = l; j < N; j += 16) dsum += Ps[r * NS + j] * Ds[r * NS + j];
        dsum = row16_sum(dsum);
        for (int j = l; j < N; j += 16) {
            const float ds = Ps[r * NS + j] * (Ds[r * NS + j] - dsum);
            Ds[r * NS + j] = ds;
            if (i0 + r < NQ) a.dS[((size_t)bh * NQ + i0 + r) * N + j] = ds;
        }
    }
    __syncthreads();
    {
        // dQ tile [16 rows][64] = sum over key tiles of dS[16][16 keys] K[16 keys][64]; a wave takes key tiles wave, wave + 4, ...
        af_f32x4 accq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) accq[dt] = (af_f32x4){0.f, 0.f, 0.f, 0.f};
        for (int jt = wave; jt * 16 < N; jt += 4) {
            const float4 da = *(const float4*)&Ds[l16 * NS + jt * 16 + 4 * g];  // dS[row l16][keys jt 16 + 4 g ..+3]
            const float dav[4] = {da.x, da.y, da.z, da.w};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float* kp = a.k + (size_t)(b * N + min(jt * 16 + 4 * g + s4, N - 1)) * a.ldk + h * HD + l16;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) accq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dav[s4], kp[16 * dt], accq[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * g + r) * HD + 16 * dt + l16] = accq[dt][r];
    }
    __syncthreads();
    {
        const int r = tid >> 4, dq = (tid & 15) * 4;
        const float4 p0 = *(const float4*)&red[(0 * 16 + r) * HD + dq], p1 = *(const float4*)&red[(1 * 16 + r) * HD + dq],
                     p2 = *(const float4*)&red[(2 * 16 + r) * HD + dq], p3 = *(const float4*)&red[(3 * 16 + r) * HD + dq];
        if (i0 + r < NQ)
            *(float4*)(a.dq + (size_t)(b * NQ + i0 + r) * a.ldd + h * HD + dq) =
                make_float4(((p0.x + p1.x) + (p2.x + p3.x)) * a.scale, ((p0.y + p1.y) + (p2.y + p3.y)) * a.scale,
                            ((p0.z + p1.z) + (p2.z + p3.z)) * a.scale, ((p0.w + p1.w) + (p2.w + p3.w)) * a.scale);
    }
}

// columns pass: dV_j = sum_i P_ij dO_i, dK_j = scale sum_i dS_ij Q_i; a workgroup takes 64 keys, each wave one 16-key tile over ALL
// query tiles in ascending order (operands straight from global memory; the four waves read neighbouring 64-byte runs of the same P /
// dS rows), so a tile's sum is one accumulation chain - no cross-wave reduction
__global__ __launch_bounds__(256) void attn_bwd_cols_kernel(AttnBwdArgs a) {
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, NQ = a.N, N = a.Nk, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int j0 = (blockIdx.y * 4 + wave) * 16;
    if (j0 >= N) return;
    af_f32x4 accv[4], acck[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { accv[dt] = (af_f32x4){0.f, 0.f, 0.f, 0.f}; acck[dt] = accv[dt]; }
    const int jc = min(j0 + l16, N - 1);
    for (int it = 0; it * 16 < NQ; ++it) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int ii = it * 16 + 4 * g + s4, i = min(ii, NQ - 1);
            const bool ok = ii < NQ;
            // A operands: P^T / dS^T - row (key) l16, contraction index (query) ii
            float pv = ok ? a.P[((size_t)bh * NQ + i) * N + jc] : 0.f;
            if (a.drop.p > 0.f) pv *= drop_factor(a.drop, ((unsigned long long)bh * NQ + i) * N + jc);  // dV = (P dropped)^T dO
            const float sv = ok ? a.dS[((size_t)bh * NQ + i) * N + jc] : 0.f;
            const float* op = a.dout + (size_t)(b * NQ + i) * a.ldo + h * HD + l16;
            const float* qp = a.q + (size_t)(b * NQ + i) * a.ld + h * HD + l16;
            float ov[4], qv[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { ov[dt] = op[16 * dt]; qv[dt] = qp[16 * dt]; }
            if (a.dnrm_scale) {
                const float sc = a.dnrm_scale[((size_t)b * a.H + h) * NQ + i];
                const float* o2 = a.out + (size_t)(b * NQ + i) * a.ldout + h * HD + l16;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) ov[dt] += sc * o2[16 * dt];
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                accv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, ov[dt], accv[dt], 0, 0, 0);
                acck[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv, qv[dt], acck[dt], 0, 0, 0);
            }
        }
    }
    // lane holds keys j0 + 4 g + r, columns 16 dt + l16
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + 4 * g + r;
        if (j >= N) continue;
        float* dvp = a.dv + (size_t)(b * N + j) * a.lddk + h * HD + l16;
        float* dkp = a.dk + (size_t)(b * N + j) * a.lddk + h * HD + l16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dvp[16 * dt] = accv[dt][r]; dkp[16 * dt] = acck[dt][r] * a.scale; }
    }
}


// ---- attention of the TRAINING forward with attention_probs dropout (med.py:202-222): P materialised by attn_probs_kernel, then
// out[b, i, h, :] = sum_j P[bh, i, j] mask_ij / (1 - p) V[b, j, h, :]  - the dS K product of the rows pass with V in K's place - and
// the pruning score's side outputs from the UNDROPPED P and the dropped out (med.py:227-233: cls_attn from attention_probs, the head
// importance from attn_out), in the layouts of madtp_attention.
__global__ __launch_bounds__(256) void attn_pv_kernel(AttnBwdArgs a, float* __restrict__ out, int ldo) {
    extern __shared__ float sm[];
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, i0 = blockIdx.y * 16, NQ = a.N, N = a.Nk, tid = threadIdx.x;
    const int NS = attn_ns(N);
    const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    float* Ps = sm;              // [16][NS]: P o mask / (1 - p), zero beyond the tile's rows / keys
    float* red = Ps + 16 * NS;   // [4][16][64]
    for (int e = tid; e < 16 * NS; e += 256) {
        const int r = e / NS, j = e - r * NS, i = i0 + r;
        float v = 0.f;
        if (i < NQ && j < N) {
            const unsigned long long idx = ((unsigned long long)bh * NQ + i) * N + j;
            v = a.P[idx];
            if (a.drop.p > 0.f) v *= drop_factor(a.drop, idx);
        }
        Ps[e] = v;
    }
    __syncthreads();
    af_f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = (af_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int jt = wave; jt * 16 < N; jt += 4) {
        const float4 pa = *(const float4*)&Ps[l16 * NS + jt * 16 + 4 * g];
        const float pav[4] = {pa.x, pa.y, pa.z, pa.w};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float* vp = a.v + (size_t)(b * N + min(jt * 16 + 4 * g + s4, N - 1)) * a.ldk + h * HD + l16;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pav[s4], vp[16 * dt], acc[dt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * g + r) * HD + 16 * dt + l16] = acc[dt][r];
    __syncthreads();
    const int r = tid >> 4, dq = (tid & 15) * 4;
    const float4 p0 = *(const float4*)&red[(0 * 16 + r) * HD + dq], p1 = *(const float4*)&red[(1 * 16 + r) * HD + dq],
                 p2 = *(const float4*)&red[(2 * 16 + r) * HD + dq], p3 = *(const float4*)&red[(3 * 16 + r) * HD + dq];
    const float4 o = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                                 (p0.w + p1.w) + (p2.w + p3.w));
    if (i0 + r < NQ) *(float4*)(out + (size_t)(b * NQ + i0 + r) * ldo + h * HD + dq) = o;
    // onorm[b, h, i] = || out[b, i, h, :] ||: 16 lanes per row
    float sq = o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
    sq = row16_sum(sq);
    if (!a.hm) return;  // (no side outputs asked; see the launcher: the onorm pointer travels in a.dq)
    if ((tid & 15) == 0 && i0 + r < NQ) a.dq[((size_t)b * a.H + h) * NQ + i0 + r] = sqrtf(sq);
}
// p0[b, h, j] = P[bh, 0, j];  colsum_part[b, rt, j] = sum over rows i of tile rt, i >= 1, of max_h P[b, h, i, j]  (self-attention)
__global__ __launch_bounds__(256) void attn_side_kernel(const float* __restrict__ P, float* __restrict__ colsum_part, float* __restrict__ p0,
                                                        int H, int N) {
    const int b = blockIdx.x, rt = blockIdx.y, nrt = gridDim.y;
    for (int j = threadIdx.x; j < N; j += 256) {
        float s = 0.f;
        for (int i = max(rt * 16, 1); i < min(rt * 16 + 16, N); ++i) {
            float m = P[(((size_t)b * H) * N + i) * N + j];
            for (int h = 1; h < H; ++h) m = fmaxf(m, P[(((size_t)b * H + h) * N + i) * N + j]);
            s += m;
        }
        colsum_part[((size_t)b * nrt + rt) * N + j] = s;
        if (rt == 0)
            for (int h = 0; h < H; ++h) p0[((size_t)b * H + h) * N + j] = P[(((size_t)b * H + h) * N) * N + j];
    }
}

// ------------------------------------------------------------------------------------------------ att_ft backward
// Query_model (models/utils.py:170-178):  W[b,k,:] = softmax_n(inner[b,:,k] / sqrt(sd_dim)),  att_ft[b,k,:] = sum_n W[b,k,n] q[b,n,:].
// Given dA = d att_ft:  dW[k,n] = <dA[k,:], q[n,:]>,  dS[k,n] = W (dW - sum_n W dW) / sqrt(sd_dim)  (added to d inner[b,n,k]),
// dq[n,:] += sum_k W[k,n] dA[k,:].  Exact f32, fixed summation orders.
// kernel 0 (round 5; D % 16 == 0): dW[b, k, t] = <dA[b,k,:], q[b,t,:]> as a batched exact-f32 MFMA product (16x16x4 f32) - the
// first version re-read q[b] (n x D) once per dictionary column, 100 x the operand, and was the second-largest kernel of a
// training step.  One workgroup per (64 tokens, sample); a wave owns one 16-token tile x all (<= 8) 16-column dictionary tiles (with
// two tiles per wave the launch had 512 waves for 1024 SIMDs and took 90 us).
// Operands come straight from global memory as float4 (contraction index d0 + 4 (lane >> 4) + s in MFMA step s - the same
// permutation of d on both operands); summation order: fixed by the instruction, chunks of 16 d in ascending order.
constexpr int AF_MAXN = 1024, AF_KT = 8;
__global__ __launch_bounds__(256) void att_ft_bwd_dw_kernel(const float* __restrict__ q, const float* __restrict__ dA,
                                                            float* __restrict__ dW, int n, int K, int D) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, t0 = blockIdx.x * 64 + wave * 16;
    if (t0 >= n) return;
    const int nkt = (K + 15) / 16;
    const float* qr = q + ((size_t)b * n + min(t0 + l16, n - 1)) * D + 4 * g;
    const float* ar[AF_KT];
#pragma unroll
    for (int i = 0; i < AF_KT; ++i) ar[i] = dA + ((size_t)b * K + min(16 * i + l16, K - 1)) * D + 4 * g;
    af_f32x4 acc[AF_KT];
#pragma unroll
    for (int i = 0; i < AF_KT; ++i) acc[i] = (af_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D; d0 += 16) {
        float4 av[AF_KT];
        const float4 qv = *(const float4*)(qr + d0);
#pragma unroll
        for (int i = 0; i < AF_KT; ++i) if (i < nkt) av[i] = *(const float4*)(ar[i] + d0);
#pragma unroll
        for (int i = 0; i < AF_KT; ++i) {
            if (i >= nkt) continue;
            acc[i] = mfma_f32_4(av[i], qv, acc[i]);
        }
    }
    // lane holds rows (dictionary columns) 16 i + 4 g + r, column (token) t0 + l16
    const int t = t0 + l16;
#pragma unroll
    for (int i = 0; i < AF_KT; ++i) {
        if (i >= nkt) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * i + 4 * g + r;
            if (k < K && t < n) dW[((size_t)b * K + k) * n + t] = acc[i][r];
        }
    }
}
// kernel 1: one workgroup per (dictionary column k, sample b); W is written to ws for kernel 2.  dw_pre: kernel 0's dW (else the
// dot products are formed here, a wave per token row).
__global__ __launch_bounds__(256) void att_ft_bwd_logits_kernel(const float* __restrict__ inner, const float* __restrict__ q,
                                                                const float* __restrict__ dA, float inv_sqrt_d,
                                                                float* __restrict__ dinner, float* __restrict__ Wws,
                                                                const float* __restrict__ dw_pre, int n, int K, int D) {
    __shared__ float w_s[AF_MAXN], dw_s[AF_MAXN], red[8];
    extern __shared__ float da_s[];  // [D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x, b = blockIdx.y;
    const float* in_b = inner + (size_t)b * n * K;
    if (!dw_pre)
        for (int d = tid; d < D; d += 256) da_s[d] = dA[((size_t)b * K + k) * D + d];
    float m = -INFINITY;
    for (int t = tid; t < n; t += 256) { const float v = in_b[(size_t)t * K + k] * inv_sqrt_d; w_s[t] = v; m = fmaxf(m, v); }
    m = wave_max(m);
    __syncthreads();
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float z = 0.f;
    for (int t = tid; t < n; t += 256) { const float e = expf(w_s[t] - m); w_s[t] = e; z += e; }
    const float Z = block_sum256(z, red);
    if (dw_pre) {
        for (int t = tid; t < n; t += 256) dw_s[t] = dw_pre[((size_t)b * K + k) * n + t];
    } else {
        // dW[n] = <dA[k,:], q[n,:]>: a wave per token row, lanes over d
        const float* q_b = q + (size_t)b * n * D;
        for (int t = wave; t < n; t += 4) {
            float acc = 0.f;
            for (int d = lane; d < D; d += 64) acc = fmaf(da_s[d], q_b[(size_t)t * D + d], acc);
            acc = wave_sum(acc);
            if (lane == 0) dw_s[t] = acc;
        }
    }
    __syncthreads();
    float c = 0.f;
    for (int t = tid; t < n; t += 256) { const float w = w_s[t] / Z; w_s[t] = w; c = fmaf(w, dw_s[t], c); }
    const float C = block_sum256(c, red);
    for (int t = tid; t < n; t += 256) {
        const float w = w_s[t];
        dinner[((size_t)b * n + t) * K + k] += w * (dw_s[t] - C) * inv_sqrt_d;
        Wws[((size_t)b * K + k) * n + t] = w;
    }
}
// kernel 2: dq[b,n,:] += sum_k W[b,k,n] dA[b,k,:]; one workgroup per (8 token rows, sample), a thread per 1/256 of D
__global__ __launch_bounds__(256) void att_ft_bwd_q_kernel(const float* __restrict__ Wws, const float* __restrict__ dA,
                                                           float* __restrict__ dq, int n, int K, int D) {
    constexpr int R = 8, DMAX = 4;  // D <= 1024
    __shared__ float w_s[R][128];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * R;
    for (int i = tid; i < R * K; i += 256) {
        const int r = i / K, k = i - r * K;
        w_s[r][k] = t0 + r < n ? Wws[((size_t)b * K + k) * n + t0 + r] : 0.f;
    }
    __syncthreads();
    float acc[R][DMAX];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < DMAX; ++j) acc[r][j] = 0.f;
    for (int k = 0; k < K; ++k) {
        float a[DMAX];
#pragma unroll
        for (int j = 0; j < DMAX; ++j) { const int d = tid + 256 * j; a[j] = d < D ? dA[((size_t)b * K + k) * D + d] : 0.f; }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < DMAX; ++j) acc[r][j] = fmaf(w_s[r][k], a[j], acc[r][j]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (t0 + r < n)
#pragma unroll
            for (int j = 0; j < DMAX; ++j) { const int d = tid + 256 * j; if (d < D) dq[((size_t)b * n + t0 + r) * D + d] += acc[r][j]; }
}

}  // namespace

extern "C" int madtp_transpose_pad(const float* src, int ld_src, int R, int C, float* dst, int ld_dst, int Rp, int Cp, void* stream) {
    if (!src || !dst || R <= 0 || C <= 0 || Rp < R || Cp < C || ld_src < C || ld_dst < Rp) return MADTP_E_BADARG;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((Rp + 31) / 32, (Cp + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, ld_src, R, C,
                       dst, ld_dst, Rp, Cp);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_transpose_split(const float* src, int ld_src, int R, int C, void* dst, int Rp, int Cp, int weight_format,
                                     float* colsum_out, float* colsum_ws, void* stream) {
    if (!src || !dst || R <= 0 || C <= 0 || Rp < R || Cp < C || ld_src < C) return MADTP_E_BADARG;
    if (Rp % 4) return MADTP_E_SHAPE;
    if ((uintptr_t)dst & 7) return MADTP_E_ALIGN;
    if (colsum_out && !colsum_ws) return MADTP_E_BADARG;
    const dim3 grid((Rp + 63) / 64, (Cp + 63) / 64);
    float* part = colsum_out ? colsum_ws : nullptr;
    if (weight_format) hipLaunchKernelGGL(transpose_split_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, R, C, (_Float16*)dst, Rp, Cp, part);
    else hipLaunchKernelGGL(transpose_split_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, R, C, (_Float16*)dst, Rp, Cp, part);
    MADTP_LAUNCH_CHECK();
    if (colsum_out) {  // column sums of src (sum over its R rows) from the per-row-tile partials, in order
        hipLaunchKernelGGL(col_reduce_final_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, part, (Rp + 63) / 64, C, colsum_out);
        MADTP_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int madtp_weight_planes(const float* w, int ldw, int N, int K, float inv_scale, void* planes, int row0, void* planes_t, int Ntp,
                                   int col0, void* stream) {
    if (!w || (!planes && !planes_t) || N <= 0 || K <= 0 || ldw < K || !(inv_scale > 0.f) || row0 < 0 || col0 < 0) return MADTP_E_BADARG;
    if (K % 4 || (planes_t && (Ntp % 4 || col0 % 4 || col0 + N > Ntp))) return MADTP_E_SHAPE;
    if (((uintptr_t)planes & 7) || ((uintptr_t)planes_t & 7)) return MADTP_E_ALIGN;
    hipLaunchKernelGGL(weight_planes_kernel, dim3((N + 63) / 64, (K + 63) / 64), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K, inv_scale,
                       (_Float16*)planes, row0, (_Float16*)planes_t, Ntp, col0, madtp_internal_range_flag());
    MADTP_LAUNCH_CHECK();
    return 0;
}

static int col_reduce(const float* dy, int ld, const float* x, int ldx, const float* stats, int M, int N, float* out, float* part,
                      hipStream_t s) {
    const int P = M >= 4096 ? 64 : (M >= 256 ? 16 : 1);
    const int rows_per = (M + P - 1) / P;
    if (x) hipLaunchKernelGGL(col_reduce_kernel<true>, dim3((N + 63) / 64, P), dim3(256), 0, s, dy, ld, x, ldx, stats, M, N, rows_per, part);
    else hipLaunchKernelGGL(col_reduce_kernel<false>, dim3((N + 63) / 64, P), dim3(256), 0, s, dy, ld, x, ldx, stats, M, N, rows_per, part);
    hipLaunchKernelGGL(col_reduce_final_kernel, dim3((N + 63) / 64), dim3(256), 0, s, part, P, N, out);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_colsum(const float* dy, int ld, int M, int N, float* out, float* part_ws, void* stream) {
    if (!dy || !out || !part_ws || M <= 0 || N <= 0 || ld < N) return MADTP_E_BADARG;
    return col_reduce(dy, ld, nullptr, 0, nullptr, M, N, out, part_ws, (hipStream_t)stream);
}

extern "C" int madtp_act_fwd_bwd(const float* u, const float* dg, float* g, float* du, size_t n, int act, void* stream) {
    if (!u || (!g && !du) || (du && !dg) || n == 0 || (n & 3)) return MADTP_E_BADARG;
    if (!aligned16(u) || (g && !aligned16(g)) || (du && (!aligned16(du) || !aligned16(dg)))) return MADTP_E_ALIGN;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, dg, g, du, n4, act);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* add, float* dx, float* dgamma,
                                   float* dbeta, float* ws, int rows, int dim, float eps, void* stream) {
    if (!x || !gamma || !dy || !dx || !ws || rows <= 0 || dim <= 0) return MADTP_E_BADARG;
    if (dim % 4 || dim > 256 * LN_MAX_CHUNKS) return MADTP_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    float* stats = ws;                       // [rows, 2]
    float* part = ws + (size_t)2 * rows;     // [64, dim]
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, dy, add, dx, stats, rows, dim, eps);
    MADTP_LAUNCH_CHECK();
    if (dgamma) { const int rc = col_reduce(dy, dim, x, dim, stats, rows, dim, dgamma, part, s); if (rc) return rc; }
    if (dbeta) { const int rc = col_reduce(dy, dim, nullptr, 0, nullptr, rows, dim, dbeta, part, s); if (rc) return rc; }
    return 0;
}

extern "C" int madtp_token_gather_bwd(const float* dy, const float* x, const int32_t* dst_pos, const float* merge_w, float* dx,
                                      float* dw, int B, int N, int k, int dim, void* stream) {
    if (!dy || !x || !dst_pos || !merge_w || !dx || !dw || B <= 0 || N < 3 || k < 1 || k > N - 1 || dim <= 0 || dim % 4) return MADTP_E_BADARG;
    const long rows = (long)B * N;
    hipLaunchKernelGGL(token_gather_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, x, dst_pos,
                       merge_w, dx, dw, B, N, k, dim);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_token_score_bwd(const float* dw, const float* score, const int32_t* dst_pos, const float* merge_w,
                                     const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                                     const float* token_attn, int ldt_row, int ldt_batch, int K, float* da, float* dp0,
                                     float* dnrm_scale, float* dtoken_attn, int B, int H, int N, void* stream) {
    if (!dw || !score || !dst_pos || !merge_w || !colsum_part || !p0 || !onorm || !token_attn || !da || !dp0 || !dnrm_scale ||
        !dtoken_attn || B <= 0 || H <= 0 || N < 3 || K <= 0)
        return MADTP_E_BADARG;
    hipLaunchKernelGGL(score_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dw, score, dst_pos, merge_w, colsum_part,
                       n_row_tiles, p0, onorm, token_attn, ldt_row, ldt_batch, K, da, dp0, dnrm_scale, dtoken_attn, H, N);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_attention_probs(const float* q, const float* k, int ld, const float* key_mask, float* P, int B, int H, int N,
                                     float scale, void* stream) {
    if (!q || !k || !P || B <= 0 || H <= 0 || N <= 0) return MADTP_E_BADARG;
    if (N > 1024) return MADTP_E_SHAPE;
    if (ld % 4 || !aligned16(q) || !aligned16(k)) return MADTP_E_ALIGN;
    AttnBwdArgs a = {};
    a.q = q; a.k = k; a.ld = ld; a.P = P; a.B = B; a.H = H; a.N = N; a.scale = scale; a.key_mask = key_mask;
    a.Nk = N; a.ldk = ld; a.mask_qk = nullptr; a.ld_mqk = 0;
    const size_t lds_p = (size_t)(16 * HD + 16 * attn_ns(N)) * sizeof(float);
    MADTP_ENSURE_MAX_LDS(attn_probs_kernel, lds_p);
    hipLaunchKernelGGL(attn_probs_kernel, dim3(B * H, (N + 15) / 16), dim3(256), lds_p, (hipStream_t)stream, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_attention_probs_x(const float* q, const float* k, int ldq, int ldk, const float* key_mask, const float* mask_qk,
                                       int ld_mqk, float* P, int B, int H, int Nq, int Nk, float scale, void* stream) {
    if (!q || !k || !P || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return MADTP_E_BADARG;
    if (Nq > 1024 || Nk > 1024 || (mask_qk && ld_mqk < Nk)) return MADTP_E_SHAPE;
    if (ldq % 4 || ldk % 4 || !aligned16(q) || !aligned16(k)) return MADTP_E_ALIGN;
    AttnBwdArgs a = {};
    a.q = q; a.k = k; a.ld = ldq; a.ldk = ldk; a.P = P; a.B = B; a.H = H; a.N = Nq; a.Nk = Nk; a.scale = scale; a.key_mask = key_mask;
    a.mask_qk = mask_qk; a.ld_mqk = ld_mqk;
    const size_t lds_p = (size_t)(16 * HD + 16 * attn_ns(Nk)) * sizeof(float);
    MADTP_ENSURE_MAX_LDS(attn_probs_kernel, lds_p);
    hipLaunchKernelGGL(attn_probs_kernel, dim3(B * H, (Nq + 15) / 16), dim3(256), lds_p, (hipStream_t)stream, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t madtp_attention_bwd_workspace(int B, int H, int N) {
    const size_t pn = (size_t)B * H * N * N * sizeof(float);
    return 2 * pn + (((size_t)B * N * N + 255) & ~(size_t)255);
}

extern "C" int madtp_attention_bwd(const float* q, const float* k, const float* v, int ld, const float* key_mask, const float* mask_qk,
                                   int ld_mqk, const float* dout, int ldo,
                                   const float* out, int ldout, const float* dnrm_scale, const float* da, const float* dp0,
                                   float* dq, float* dk, float* dv, int ldd, void* ws, size_t ws_bytes, float* dp_out, int B, int H,
                                   int N, float scale, float p_drop, unsigned long long seed, unsigned long long site, void* stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || !ws || B <= 0 || H <= 0 || N <= 0) return MADTP_E_BADARG;
    if (dnrm_scale && !out) return MADTP_E_BADARG;
    if (N > 1024) return MADTP_E_SHAPE;  // the row kernels keep 16 x N score rows in LDS (2 x 64 KiB at N = 1024)
    if (ws_bytes < madtp_attention_bwd_workspace(B, H, N)) return MADTP_E_BADARG;
    if (ld % 4 || ldd % 4 || !aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(dq) || !aligned16(dk) || !aligned16(dv)) return MADTP_E_ALIGN;
    AttnBwdArgs a;
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.dout = dout; a.ldo = ldo; a.out = out; a.ldout = ldout; a.dnrm_scale = dnrm_scale;
    a.da = da; a.dp0 = dp0; a.dq = dq; a.dk = dk; a.dv = dv; a.ldd = ldd; a.B = B; a.H = H; a.N = N; a.scale = scale;
    a.key_mask = key_mask; a.Nk = N; a.ldk = ld; a.lddk = ldd; a.mask_qk = mask_qk; a.ld_mqk = ld_mqk; a.dp_out = dp_out;
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) return MADTP_E_BADARG;
    a.drop = DropArgs{p_drop, 1.0f / (1.0f - p_drop), seed, site};
    const size_t pn = (size_t)B * H * N * N;
    a.P = (float*)ws; a.dS = a.P + pn; a.hm = (unsigned char*)(a.dS + pn);
    hipStream_t s = (hipStream_t)stream;
    const int nrt = (N + 15) / 16;
    const size_t lds_p = (size_t)(16 * HD + 16 * attn_ns(N)) * sizeof(float), lds_r = (size_t)(16 * HD + 32 * attn_ns(N) + 64 * HD) * sizeof(float);
    MADTP_ENSURE_MAX_LDS(attn_probs_kernel, lds_p);
    MADTP_ENSURE_MAX_LDS(attn_bwd_rows_kernel, lds_r);
    hipLaunchKernelGGL(attn_probs_kernel, dim3(B * H, nrt), dim3(256), lds_p, s, a);
    if (da) hipLaunchKernelGGL(attn_headmax_kernel, dim3(B, N), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_rows_kernel, dim3(B * H, nrt), dim3(256), lds_r, s, a);
    hipLaunchKernelGGL(attn_bwd_cols_kernel, dim3(B * H, (N + 63) / 64), dim3(256), 0, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}


// Cross-attention backward (med.py:143-236 with encoder_hidden_states: Nq text queries against Nk encoder tokens, no score terms):
// q [B*Nq, ldq], k / v [B*Nk, ldkv] (e.g. the two halves of a fused [k|v] projection of the encoder tokens), dout [B*Nq, ldo] ->
// dq [B*Nq, lddq], dk / dv [B*Nk, lddkv].  key_mask: additive [B,Nk] or NULL (MED ignores the encoder mask, nlvr_encoder applies
// it).  ws: 2 * B*H*Nq*Nk floats.  Nq, Nk <= 1024.
extern "C" size_t madtp_attention_bwd_cross_workspace(int B, int H, int Nq, int Nk) {
    return 2 * (size_t)B * H * Nq * Nk * sizeof(float);
}
extern "C" int madtp_attention_bwd_cross(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* key_mask,
                                         const float* dout, int ldo, float* dq, int lddq, float* dk, float* dv, int lddkv, void* ws,
                                         size_t ws_bytes, int B, int H, int Nq, int Nk, float scale, float p_drop,
                                         unsigned long long seed, unsigned long long site, void* stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || !ws || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return MADTP_E_BADARG;
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) return MADTP_E_BADARG;
    if (Nq > 1024 || Nk > 1024) return MADTP_E_SHAPE;
    if (ws_bytes < madtp_attention_bwd_cross_workspace(B, H, Nq, Nk)) return MADTP_E_BADARG;
    if (ldq % 4 || ldkv % 4 || lddq % 4 || lddkv % 4 || !aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(dq) ||
        !aligned16(dk) || !aligned16(dv)) return MADTP_E_ALIGN;
    AttnBwdArgs a = {};
    a.q = q; a.k = k; a.v = v; a.ld = ldq; a.ldk = ldkv; a.dout = dout; a.ldo = ldo; a.dq = dq; a.dk = dk; a.dv = dv;
    a.ldd = lddq; a.lddk = lddkv; a.B = B; a.H = H; a.N = Nq; a.Nk = Nk; a.scale = scale; a.key_mask = key_mask;
    a.drop = DropArgs{p_drop, 1.0f / (1.0f - p_drop), seed, site};
    const size_t pn = (size_t)B * H * Nq * Nk;
    a.P = (float*)ws; a.dS = a.P + pn;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_p = (size_t)(16 * HD + 16 * attn_ns(Nk)) * sizeof(float), lds_r = (size_t)(16 * HD + 32 * attn_ns(Nk) + 64 * HD) * sizeof(float);
    MADTP_ENSURE_MAX_LDS(attn_probs_kernel, lds_p);
    MADTP_ENSURE_MAX_LDS(attn_bwd_rows_kernel, lds_r);
    hipLaunchKernelGGL(attn_probs_kernel, dim3(B * H, (Nq + 15) / 16), dim3(256), lds_p, s, a);
    hipLaunchKernelGGL(attn_bwd_rows_kernel, dim3(B * H, (Nq + 15) / 16), dim3(256), lds_r, s, a);
    hipLaunchKernelGGL(attn_bwd_cols_kernel, dim3(B * H, (Nk + 63) / 64), dim3(256), 0, s, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_dropout(const float* x, const float* residual, float* y, size_t n, size_t per_sample, float p, unsigned long long seed,
                             unsigned long long site, void* stream) {
    if (!x || !y || n == 0 || !(p >= 0.f) || !(p < 1.f)) return MADTP_E_BADARG;
    if (n % 4 || per_sample % 4) return MADTP_E_SHAPE;
    if (!aligned16(x) || !aligned16(y) || (residual && !aligned16(residual))) return MADTP_E_ALIGN;
    const DropArgs d{p, 1.0f / (1.0f - p), seed, site};
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, residual, y, n / 4, per_sample, d);
    MADTP_LAUNCH_CHECK();
    return 0;
}

// Attention of the training forward (see attn_pv_kernel): P [B,H,Nq,Nk] f32 is written (the caller's scratch - the backward
// recomputes it), out [B*Nq, ldo]; side outputs as madtp_attention's (self-attention, Nq == Nk) or NULL.
extern "C" int madtp_attention_train(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* key_mask,
                                     const float* mask_qk, int ld_mqk, float* P, float* out, int ldo, float* colsum_part, float* p0,
                                     float* onorm, int B, int H, int Nq, int Nk, float scale, float p_drop, unsigned long long seed,
                                     unsigned long long site, void* stream) {
    if (!q || !k || !v || !P || !out || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || !(p_drop >= 0.f) || !(p_drop < 1.f)) return MADTP_E_BADARG;
    if (Nq > 1024 || Nk > 1024) return MADTP_E_SHAPE;
    if (colsum_part && (!p0 || !onorm || Nq != Nk)) return MADTP_E_BADARG;
    if (ldq % 4 || ldkv % 4 || ldo % 4 || !aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out)) return MADTP_E_ALIGN;
    AttnBwdArgs a = {};
    a.q = q; a.k = k; a.v = v; a.ld = ldq; a.ldk = ldkv; a.B = B; a.H = H; a.N = Nq; a.Nk = Nk; a.scale = scale; a.key_mask = key_mask;
    a.mask_qk = mask_qk; a.ld_mqk = ld_mqk; a.P = P;
    a.drop = DropArgs{p_drop, 1.0f / (1.0f - p_drop), seed, site};
    // (attn_pv_kernel's optional output onorm rides in a.dq, switched on by a non-null a.hm)
    a.dq = onorm; a.hm = colsum_part ? (unsigned char*)P : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int nrt = (Nq + 15) / 16;
    const size_t lds_p = (size_t)(16 * HD + 16 * attn_ns(Nk)) * sizeof(float), lds_v = (size_t)(16 * attn_ns(Nk) + 64 * HD) * sizeof(float);
    MADTP_ENSURE_MAX_LDS(attn_probs_kernel, lds_p);
    MADTP_ENSURE_MAX_LDS(attn_pv_kernel, lds_v);
    hipLaunchKernelGGL(attn_probs_kernel, dim3(B * H, nrt), dim3(256), lds_p, s, a);
    hipLaunchKernelGGL(attn_pv_kernel, dim3(B * H, nrt), dim3(256), lds_v, s, a, out, ldo);
    if (colsum_part) hipLaunchKernelGGL(attn_side_kernel, dim3(B, nrt), dim3(256), 0, s, P, colsum_part, p0, H, Nq);
    MADTP_LAUNCH_CHECK();
    return 0;
}

// d att_ft -> (d inner += ..., d q += ...), see att_ft_bwd_logits_kernel.  inner / dinner: dense [B, n, K]; q / dq: dense [B, n, D];
// dA: [B, K, D]; ws: 2 B K n floats (W, dW).  inv_sqrt_d = 1 / sqrt(sd_dim) (models/utils.py:174).
extern "C" int madtp_att_ft_bwd(const float* inner, const float* q, const float* dA, float inv_sqrt_d, float* dinner, float* dq,
                                float* ws, int B, int n, int K, int D, void* stream) {
    if (!inner || !q || !dA || !dinner || !dq || !ws || B <= 0 || n <= 0 || K <= 0 || D <= 0) return MADTP_E_BADARG;
    if (n > AF_MAXN || K > 128 || D > 1024) return MADTP_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    float* dw_pre = nullptr;
    if (D % 16 == 0 && aligned16(q) && aligned16(dA)) {
        dw_pre = ws + (size_t)B * K * n;
        hipLaunchKernelGGL(att_ft_bwd_dw_kernel, dim3((n + 63) / 64, B), dim3(256), 0, s, q, dA, dw_pre, n, K, D);
        MADTP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(att_ft_bwd_logits_kernel, dim3(K, B), dim3(256), (size_t)D * sizeof(float), s, inner, q, dA, inv_sqrt_d, dinner, ws,
                       dw_pre, n, K, D);
    MADTP_LAUNCH_CHECK();
    hipLaunchKernelGGL(att_ft_bwd_q_kernel, dim3((n + 7) / 8, B), dim3(256), 0, s, ws, dA, dq, n, K, D);
    MADTP_LAUNCH_CHECK();
    return 0;
}
