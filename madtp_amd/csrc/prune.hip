// The memory-bound half of MADTP's dynamic token pruning (k8, k9, k12, k7b):
//   token_score  - importance score, alignment-guided threshold, survivor count, batch max   (vit.py:125-145)
//   token_select - rank-based top-k, compaction map, merge weights                          (vit.py:153-159)
//   token_gather - gather kept rows + weighted merge of dropped rows into one token          (vit.py:154-161)
//   mask_gather  - additive-mask compaction for the text encoders                             (nlvr_encoder.py:451-452)
//   query_att_ft - Query_model's att_ft = softmax_t(x.sd^T/sqrt(d)) @ x                       (models/utils.py:174-178)
// No MFMA except att_ft (a 100 x n x 768 batched product).  These kernels are HBM/L2 bound integer+f32 work:
// coalesced row reads, wave shuffle reductions, ballot/popcount prefix sums, fixed reduction orders (no float
// atomics, so results are run-to-run deterministic and independent of workgroup placement).
#include "common.h"
#include "internal.h"
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

#ifndef TRY
#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
#endif

namespace {

constexpr int MAXN = 1024;  // max patch tokens per sample handled by the LDS-resident kernels

__device__ __forceinline__ float block_sum(float v, float* red, int tid, int nwaves) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < nwaves; ++w) s += red[w];
    return s;
}

__device__ __forceinline__ float block_max_f(float v, float* red, int tid, int nwaves) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float m = red[0];
    for (int w = 1; w < nwaves; ++w) m = fmaxf(m, red[w]);
    return m;
}

// ------------------------------------------------------------------------------------------------ token_score
// one 512-thread workgroup per sample.  The sample's logits token_attn[b] (n x K f32, <= 150 KB) are staged in LDS
// once with coalesced float4 loads (STAGED; falls back to reading global memory for very long sequences), then
//   phase A: per-token terms (row max of the logits, column mass of head-max attention, CLS attention);
//   phase B: per-dictionary-column softmax over tokens (4 token slices x 128 columns) and
//            threshold = min_k sum_t softmax_t(x/T)[t,k] * I[t]; survivor count; optional batch max (atomicMax).
#ifdef MADTP_TS_TIMING
__device__ long long g_ts_dbg[16];
#define TS_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_ts_dbg[i] = wall_clock64(); } while (0)
#else
#define TS_MARK(i)
#endif
// FAST (fast precision modes only, madtp_set_score_fast): phase B in log2 units with v_exp_f32 and one reciprocal per column
// instead of two IEEE divisions and a precise expf per logit - phase B is VALU-bound (10 of 17 us at 197 tokens) and the parity
// modes' arithmetic (FAST = false) has to stay what the reference computes.
template <bool STAGED, bool FAST = false>
__global__ __launch_bounds__(512) void token_score_kernel(const float* __restrict__ colsum, int nrt,
                                                          const float* __restrict__ p0, const float* __restrict__ onorm,
                                                          const float* __restrict__ ta, int ldt_g, int ldb, int K, float temperature,
                                                          float* __restrict__ score, float* __restrict__ threshold,
                                                          int32_t* __restrict__ count, int32_t* __restrict__ kmax, int H,
                                                          int N, int32_t* done_ctr, int32_t* host_slot, int seq,
                                                          int32_t* dims_l = nullptr) {
    if (dims_l) {  // sync-free encoder path: the token count comes from the device-side record of this layer (common.h DevN)
        N = dims_l[0];
        nrt = (N + 15) / 16;
        ldb = N * ldt_g;  // token_attn = rows 1.. of each sample of a dense [B * N, ldt] logits buffer
    }
    __shared__ float I_s[MAXN];
    __shared__ int last_s;
    __shared__ float tw_s[MAXN];
    __shared__ float red[8];
    __shared__ float colred[4][128];
    __shared__ float colstat[128];
    extern __shared__ __attribute__((aligned(16))) float ta_s[];  // [n][K] when STAGED
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, n = N - 1;
    const float* ta_g = ta + (size_t)b * ldb;  // row t <-> patch token t
    TS_MARK(0);
    // The attention-side operands of this thread's tokens (column-mass partials, CLS attention, context norms: <= 48 values per
    // token) are requested BEFORE the logits are staged, so that their latency runs under the staging loads instead of after the
    // row maxima (short sequences; the long-sequence form below loads in batches where it uses them).
    constexpr int HMAX = 16, RMAX = 16;
    const bool pre = H <= HMAX && nrt <= RMAX;
    float on_pre[2][HMAX], pz_pre[2][HMAX], cs_pre[2][RMAX];
    if (pre) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = min(tid + u * 512, n - 1);  // clamped: straight-line loads, the surplus lanes' values are never used
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                const size_t o = ((size_t)b * H + min(h, H - 1)) * N + t + 1;
                on_pre[u][h] = onorm[o];
                pz_pre[u][h] = p0[o];
            }
#pragma unroll
            for (int r = 0; r < RMAX; ++r) cs_pre[u][r] = colsum[((size_t)b * nrt + min(r, nrt - 1)) * N + t + 1];
        }
    }
    if constexpr (STAGED) {
        const int k4 = K >> 2;  // K % 4 == 0 on this path
        for (int idx = tid; idx < n * k4; idx += 512) {
            const int t = idx / k4, c4 = (idx - t * k4) * 4;
            *(float4*)(ta_s + t * K + c4) = *(const float4*)(ta_g + (size_t)t * ldt_g + c4);
        }
        __syncthreads();
    }
    TS_MARK(1);
    const int ldt = STAGED ? K : ldt_g;
    auto TA = [&](int t, int c) -> float { return STAGED ? ta_s[t * ldt + c] : ta_g[(size_t)t * ldt + c]; };

    // token_attn_w = max over dictionary columns (vit.py:131)
    if constexpr (STAGED) {
        // four lanes per row, float4 reads, two quad shuffles: 128 rows per pass (a wave-wide reduction per row costs six
        // dependent cross-lane steps - it was a third of the kernel)
        const int k4 = K >> 2, q = tid & 3;
        for (int t = tid >> 2; t < n; t += 128) {
            float m = -INFINITY;
            for (int c = q; c < k4; c += 4) {
                const float4 v = *(const float4*)(ta_s + t * K + 4 * c);
                m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            }
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            if (q == 0) tw_s[t] = m;
        }
    } else {
        for (int t = wave; t < n; t += 8) {
            float m = -INFINITY;
            for (int k = lane; k < K; k += 64) m = fmaxf(m, TA(t, k));
            m = wave_max(m);
            if (lane == 0) tw_s[t] = m;
        }
    }
    __syncthreads();

    TS_MARK(2);
    // a = column mass of head-max attention (vit.py:126-127), c = head-diversity weighted CLS attention (vit.py:96-101)
    // All global operands of a token are requested before any is used (fully unrolled, predicated loads: the head and
    // row-tile counts are small): one memory latency instead of one per group of four.
    float a_loc[2], c_loc[2], suma_l = 0.f, sumt_l = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + u * 512;
        a_loc[u] = 0.f; c_loc[u] = 0.f;
        if (t < n) {
            float a = 0.f, hs = 0.f, c = 0.f;
            if (pre) {  // (the same sums in the same order as before the loads moved up)
#pragma unroll
                for (int r = 0; r < RMAX; ++r) if (r < nrt) a += cs_pre[u][r];
#pragma unroll
                for (int h = 0; h < HMAX; ++h) if (h < H) hs += on_pre[u][h];
#pragma unroll
                for (int h = 0; h < HMAX; ++h) if (h < H) c += pz_pre[u][h] * (on_pre[u][h] / (hs + 1e-8f));
            } else {
                // long sequences (nrt = 38..57 row tiles at 605..901 tokens): the same sums in the same order, with the loads of
                // sixteen row tiles requested before the first add (one load per dependent add was ~60 L2 round trips per token)
                for (int r0 = 0; r0 < nrt; r0 += RMAX) {
                    float cs[RMAX];
#pragma unroll
                    for (int r = 0; r < RMAX; ++r) cs[r] = r0 + r < nrt ? colsum[((size_t)b * nrt + r0 + r) * N + t + 1] : 0.f;
#pragma unroll
                    for (int r = 0; r < RMAX; ++r) if (r0 + r < nrt) a += cs[r];
                }
                if (H <= HMAX) {
                    float on[HMAX], pz[HMAX];
#pragma unroll
                    for (int h = 0; h < HMAX; ++h) {
                        const size_t o = ((size_t)b * H + h) * N + t + 1;
                        on[h] = h < H ? onorm[o] : 0.f;
                        pz[h] = h < H ? p0[o] : 0.f;
                    }
#pragma unroll
                    for (int h = 0; h < HMAX; ++h) if (h < H) hs += on[h];
#pragma unroll
                    for (int h = 0; h < HMAX; ++h) if (h < H) c += pz[h] * (on[h] / (hs + 1e-8f));
                } else {
                    for (int h = 0; h < H; ++h) hs += onorm[((size_t)b * H + h) * N + t + 1];
                    for (int h = 0; h < H; ++h) {
                        const size_t o = ((size_t)b * H + h) * N + t + 1;
                        c += p0[o] * (onorm[o] / (hs + 1e-8f));
                    }
                }
            }
            a_loc[u] = a; c_loc[u] = c;
            suma_l += a; sumt_l += tw_s[t];
        }
    }
    TS_MARK(3);
    const float suma = block_sum(suma_l, red, tid, 8);
    const float sumt = block_sum(sumt_l, red, tid, 8);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + u * 512;
        if (t < n) {
            const float aw = a_loc[u] / (suma + 1e-8f);
            const float tw = tw_s[t] / (sumt + 1e-8f);
            const float sc = (aw + tw + c_loc[u]) / 3.0f;  // vit.py:134
            I_s[t] = sc;
            score[(size_t)b * n + t] = sc;
        }
    }
    __syncthreads();

    TS_MARK(4);
    // phase B: softmax over tokens of token_attn/T per column (vit.py:137-139), 4 slices x 128 columns
    const int col = tid & 127, slice = tid >> 7;
    const int t0 = (n * slice) / 4, t1 = (n * (slice + 1)) / 4;
    const bool cval = col < K;
    // (STAGED: the LDS copy is private to this kernel and element [t][col] is only touched by this thread from here on,
    //  so x/T and then exp(x/T - max) overwrite it in place - the arithmetic of each pass is unchanged, it just is not
    //  repeated by the next one)
    float m = -INFINITY;
    const float c2 = 1.44269504088896341f / temperature;  // FAST: logits in log2 units
    if constexpr (FAST && STAGED) {
        // fast modes: TWO passes over the staged column and no stores - the maximum of the raw logits (a positive scale commutes
        // with max), then exp2(fma(x, c2, -c2 max)), its sum and its I-weighted sum together (sw / sum = the softmax-weighted score)
        if (cval) {
#pragma unroll 8
            for (int t = t0; t < t1; ++t) m = fmaxf(m, ta_s[t * K + col]);
        }
        colred[slice][col] = m;
        __syncthreads();
        m = fmaxf(fmaxf(colred[0][col], colred[1][col]), fmaxf(colred[2][col], colred[3][col]));
        const float off = -m * c2;
        float se_f = 0.f, sw_f = 0.f;
        if (cval) {
#pragma unroll 8
            for (int t = t0; t < t1; ++t) {
                const float e = __builtin_amdgcn_exp2f(fmaf(ta_s[t * K + col], c2, off));
                se_f += e;
                sw_f = fmaf(e, I_s[t], sw_f);
            }
        }
        __syncthreads();
        colred[slice][col] = se_f;
        tw_s[slice * 128 + col] = sw_f;  // (tw_s is free from here on: 4 x 128 partial weighted sums)
        __syncthreads();
        if (tid < 128) {
            const float sum = ((colred[0][col] + colred[1][col]) + colred[2][col]) + colred[3][col];
            const float sw = ((tw_s[col] + tw_s[128 + col]) + tw_s[256 + col]) + tw_s[384 + col];
            colstat[tid] = cval ? sw * __builtin_amdgcn_rcpf(sum) : INFINITY;
        }
        __syncthreads();
    } else {
    if (cval) {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            const float v = TA(t, col) / temperature;
            if constexpr (STAGED) ta_s[t * K + col] = v;
            m = fmaxf(m, v);
        }
    }
    colred[slice][col] = m;
    __syncthreads();
    m = fmaxf(fmaxf(colred[0][col], colred[1][col]), fmaxf(colred[2][col], colred[3][col]));
    __syncthreads();
    float se = 0.f;
    if (cval) {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            float e;
            if constexpr (STAGED) { e = expf(ta_s[t * K + col] - m); ta_s[t * K + col] = e; }
            else e = expf(TA(t, col) / temperature - m);
            se += e;
        }
    }
    colred[slice][col] = se;
    __syncthreads();
    const float sum = ((colred[0][col] + colred[1][col]) + colred[2][col]) + colred[3][col];
    __syncthreads();
    float sw = 0.f;
    if (cval) {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            const float e = STAGED ? ta_s[t * K + col] : expf(TA(t, col) / temperature - m);
            sw += (e / sum) * I_s[t];
        }
    }
    colred[slice][col] = sw;
    __syncthreads();
    if (tid < 128) colstat[tid] = cval ? ((colred[0][col] + colred[1][col]) + colred[2][col]) + colred[3][col] : INFINITY;
    __syncthreads();
    }
    TS_MARK(5);
    // threshold = min over columns (vit.py:141)
    float thr = INFINITY;
    for (int k = lane; k < 128; k += 64) thr = fminf(thr, colstat[k]);
    thr = wave_min(thr);
    // survivors (vit.py:143-145)
    int cnt = 0;
    for (int t = tid; t < n; t += 512) cnt += I_s[t] > thr ? 1 : 0;
    const float total = block_sum((float)cnt, red, tid, 8);  // exact: counts <= 1024
    TS_MARK(6);
    if (tid == 0) {
        threshold[b] = thr;
        count[b] = (int)total;
        if (kmax) atomicMax(kmax, (int)total);
        int last = 0;
        if (done_ctr) {
            __threadfence();  // count[b] is visible before the ticket
            last = atomicAdd(done_ctr, 1) == (int)gridDim.x - 1;
        }
        last_s = last;
    }
    if (!done_ctr) return;
    __syncthreads();
    if (!last_s) return;
    // Last workgroup of the launch: k = max_b count goes straight to host-visible pinned memory (slot[0] = k, then
    // slot[1] = sequence number with system-scope release) - the host spins on the sequence number instead of paying a
    // device-to-host copy plus a stream synchronisation for the one value it needs per layer (vit.py:145 `.item()`).
    int mloc = 0;
    for (int i = tid; i < (int)gridDim.x; i += 512) mloc = max(mloc, __hip_atomic_load(count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int kk = (int)block_max_f((float)mloc, red, tid, 8);  // exact: counts <= 1024
    if (tid == 0) {
        *done_ctr = 0;
        if (dims_l) {
            // the layer's decision stays on the device: k, the k applied under the BLIP rule (vit.py:148-149: no pruning when
            // k < 1 or N - 1 - k <= 1) and the next layer's token count; later kernels of the stream read them (kernel boundary)
            const int k_used = (kk < 1 || (N - 1 - kk) <= 1) ? 0 : kk;
            dims_l[1] = kk;
            dims_l[2] = k_used;
            dims_l[3] = k_used ? k_used + 2 : N;
            dims_l[DIMS_STRIDE] = k_used ? k_used + 2 : N;
        }
        if (host_slot) {
            __hip_atomic_store(host_slot, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_slot + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Long sequences (the logits of a sample no longer fit the LDS: 577 / 901 tokens at 384^2 / 480^2 images) at small batches: one
// workgroup per sample leaves most of the chip idle (VQA: 32 samples -> 32 of 256 CUs, 124 us per launch).  Here a sample is
// split over G workgroups by DICTIONARY COLUMNS: every workgroup repeats the cheap per-token phase A (so it owns the complete
// score vector I), runs the three softmax-over-tokens passes of phase B for its 128/G columns with 512/(128/G) token slices,
// and publishes the minimum of its columns' sums; the last workgroup of the sample to arrive (agent-scope ticket) takes the
// minimum over the G partial minima (exact, order-free), counts the survivors and takes part in the launch-wide ticket that
// hands k to the host.  tick / part: per-sample scratch of the hand-over slot (zeroed once, every ticket resets itself).
template <int G, bool FAST = false>
__global__ __launch_bounds__(512, 2) void token_score_split_kernel(const float* __restrict__ colsum, int nrt,
                                                                const float* __restrict__ p0, const float* __restrict__ onorm,
                                                                const float* __restrict__ ta, int ldt, int ldb, int K,
                                                                float temperature, float* __restrict__ score,
                                                                float* __restrict__ threshold, int32_t* __restrict__ count, int H, int N,
                                                                int32_t* done_ctr, int32_t* host_slot, int seq, int32_t* tick,
                                                                float* part, int nsamp) {
    constexpr int KC = 128 / G, S = 512 / KC;  // columns per workgroup, token slices
    __shared__ float I_s[MAXN];
    __shared__ int last_s, lastg_s;
    __shared__ float tw_s[MAXN];
    __shared__ float red[8];
    __shared__ float colred[S][KC];
    __shared__ float colstat[KC];
    const int tid = threadIdx.x, lane = tid & 63;
    // The G workgroups of a sample share its logits and attention statistics, so they sit on ONE XCD (workgroup id % 8 is the
    // XCD: sample b lives on XCD b % 8, its workgroups are that XCD's slots (b / 8) G .. + G - 1).  With the sample's workgroups
    // dealt round-robin over the XCDs every L2 fetched every sample (8 x 11.5 MB through the fabric at 32 x 901 tokens: the row
    // max alone ran 17 us at ~5 TB/s of fabric traffic).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = slot % G, b = (slot / G) * 8 + xcd, n = N - 1;
    if (b >= nsamp) return;
    const float* ta_g = ta + (size_t)b * ldb;
#ifdef MADTP_TS_TIMING
#define TSS_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_ts_dbg[i] = wall_clock64(); } while (0)
#else
#define TSS_MARK(i)
#endif
    TSS_MARK(0);
    // this workgroup's KC columns of the sample's logits, staged once ([n][KC] f32: <= 57 KiB at 901 tokens and G = 8) - the three
    // softmax passes of phase B then read LDS instead of walking 64-byte pieces of global rows three times
    extern __shared__ __attribute__((aligned(16))) float cols_s[];
    auto TA = [&](int t, int c) -> float { return cols_s[t * KC + (c - g * KC)]; };
    {
        constexpr int C4 = KC / 4;
        for (int idx = tid; idx < n * C4; idx += 512) {
            const int t = idx / C4, c = (idx - t * C4) * 4;
            if (g * KC + c < K)  // K % 4 == 0: a float4 is inside the row or not at all
                *(float4*)(cols_s + t * KC + c) = *(const float4*)(ta_g + (size_t)t * ldt + g * KC + c);
        }
    }

    TSS_MARK(1);
    // ---- phase A (identical in every workgroup of the sample): row max of the logits, four lanes per row, float4 reads (all of
    //      a lane's <= 8 loads of a row are requested before the first max)
    {
        // eight lanes per row, four float4 per lane (K <= 128), 64 rows per pass and EIGHT passes requested at once (32 loads per
        // lane in flight: the sample's 360 KB at 901 tokens are two round trips instead of one per 128 rows - the row max was a
        // third of this kernel).  max is exact and order-free.
        const int k4 = K >> 2, q = tid & 7;  // K % 4 == 0, ldt % 4 == 0, 16-byte aligned rows (launch condition)
        for (int tb = tid >> 3; tb < n; tb += 512) {
            float4 v[8][4];
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int t = tb + 64 * ps;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = q + 8 * i;
                    v[ps][i] = (t < n && c < k4) ? *(const float4*)(ta_g + (size_t)t * ldt + 4 * c)
                                                 : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                }
            }
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int t = tb + 64 * ps;
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i) m = fmaxf(fmaxf(m, fmaxf(v[ps][i].x, v[ps][i].y)), fmaxf(v[ps][i].z, v[ps][i].w));
                m = fmaxf(m, __shfl_xor(m, 1));
                m = fmaxf(m, __shfl_xor(m, 2));
                m = fmaxf(m, __shfl_xor(m, 4));
                if (q == 0 && t < n) tw_s[t] = m;
            }
        }
    }
    __syncthreads();
    TSS_MARK(2);
    // a = column mass of head-max attention, c = head-diversity weighted CLS attention (as in token_score_kernel: the same sums in
    // the same order).  Every load of BOTH tokens of a thread is requested before the first add - clamped indices instead of
    // predicates: straight-line code, no waits at control-flow merges - in batches of 32 row tiles (nrt = 38..57 here).
    constexpr int HMAX = 16, RB = 32;
    float a_loc[2] = {0.f, 0.f}, c_loc[2] = {0.f, 0.f}, suma_l = 0.f, sumt_l = 0.f;
    const int tt[2] = {min(tid, n - 1), min(tid + 512, n - 1)};
    for (int r0 = 0; r0 < nrt; r0 += RB) {
        float cs[2][RB];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < RB; ++r) cs[u][r] = colsum[((size_t)b * nrt + min(r0 + r, nrt - 1)) * N + tt[u] + 1];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < RB; ++r) if (r0 + r < nrt) a_loc[u] += cs[u][r];
    }
    if (H <= HMAX) {
        float on[2][HMAX], pz[2][HMAX];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                const size_t o = ((size_t)b * H + min(h, H - 1)) * N + tt[u] + 1;
                on[u][h] = onorm[o];
                pz[u][h] = p0[o];
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float hs = 0.f, c = 0.f;
#pragma unroll
            for (int h = 0; h < HMAX; ++h) if (h < H) hs += on[u][h];
#pragma unroll
            for (int h = 0; h < HMAX; ++h) if (h < H) c += pz[u][h] * (on[u][h] / (hs + 1e-8f));
            c_loc[u] = c;
        }
    } else {
        for (int u = 0; u < 2; ++u) {
            float hs = 0.f, c = 0.f;
            for (int h = 0; h < H; ++h) hs += onorm[((size_t)b * H + h) * N + tt[u] + 1];
            for (int h = 0; h < H; ++h) {
                const size_t o = ((size_t)b * H + h) * N + tt[u] + 1;
                c += p0[o] * (onorm[o] / (hs + 1e-8f));
            }
            c_loc[u] = c;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + u * 512;
        if (t < n) { suma_l += a_loc[u]; sumt_l += tw_s[t]; }
        else { a_loc[u] = 0.f; c_loc[u] = 0.f; }
    }
    TSS_MARK(3);
    const float suma = block_sum(suma_l, red, tid, 8);
    const float sumt = block_sum(sumt_l, red, tid, 8);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + u * 512;
        if (t < n) {
            const float aw = a_loc[u] / (suma + 1e-8f);
            const float tw = tw_s[t] / (sumt + 1e-8f);
            const float sc = (aw + tw + c_loc[u]) / 3.0f;  // vit.py:134
            I_s[t] = sc;
            if (g == 0) score[(size_t)b * n + t] = sc;
        }
    }
    __syncthreads();

    TSS_MARK(4);
    // ---- phase B on columns [g KC, (g+1) KC): softmax over tokens of token_attn / T (vit.py:137-139), S token slices
    const int cl = tid % KC, slice = tid / KC, col = g * KC + cl;
    const int t0 = (n * slice) / S, t1 = (n * (slice + 1)) / S;
    const bool cval = col < K;
    // (the LDS copy is private to this workgroup and element [t][col] is only touched by this thread: x / T and then
    //  exp(x / T - max) overwrite it in place - each pass's arithmetic is unchanged, it just is not repeated by the next)
    float m = -INFINITY;
    if (cval) {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            const float v = FAST ? TA(t, col) * (1.44269504088896341f / temperature) : TA(t, col) / temperature;  // (FAST: log2 units)
            cols_s[t * KC + cl] = v;
            m = fmaxf(m, v);
        }
    }
    colred[slice][cl] = m;
    __syncthreads();
    m = colred[0][cl];
#pragma unroll
    for (int s2 = 1; s2 < S; ++s2) m = fmaxf(m, colred[s2][cl]);
    __syncthreads();
    float se = 0.f;
    if (cval) {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            const float e = FAST ? __builtin_amdgcn_exp2f(cols_s[t * KC + cl] - m) : expf(cols_s[t * KC + cl] - m);
            cols_s[t * KC + cl] = e;
            se += e;
        }
    }
    colred[slice][cl] = se;
    __syncthreads();
    float sum = colred[0][cl];
#pragma unroll
    for (int s2 = 1; s2 < S; ++s2) sum += colred[s2][cl];  // fixed order
    __syncthreads();
    float sw = 0.f;
    if (cval) {
        if constexpr (FAST) {
#pragma unroll 8
            for (int t = t0; t < t1; ++t) sw = fmaf(cols_s[t * KC + cl], I_s[t], sw);
            sw *= __builtin_amdgcn_rcpf(sum);
        } else {
#pragma unroll 8
            for (int t = t0; t < t1; ++t) sw += (cols_s[t * KC + cl] / sum) * I_s[t];
        }
    }
    colred[slice][cl] = sw;
    __syncthreads();
    if (tid < KC) {
        float v = colred[0][tid];
#pragma unroll
        for (int s2 = 1; s2 < S; ++s2) v += colred[s2][tid];
        colstat[tid] = (g * KC + tid) < K ? v : INFINITY;
    }
    __syncthreads();
    TSS_MARK(5);
    float pm = INFINITY;
    for (int k = lane; k < KC; k += 64) pm = fminf(pm, colstat[k]);
    pm = wave_min(pm);
    // ---- publish this workgroup's minimum, the last arriver of the sample finishes (MI355X_MICROARCH.md: agent-scope
    //      release before the ticket, acquire after it, the partials read past the L1 with agent-scope atomic loads)
    if (tid == 0) {
        __hip_atomic_store(part + (size_t)b * G + g, pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        lastg_s = atomicAdd(tick + b, 1) == G - 1;
    }
    __syncthreads();
    TSS_MARK(6);
    if (!lastg_s) return;
    if (tid == 0) __threadfence();
    __syncthreads();
    float thr = INFINITY;
#pragma unroll
    for (int i = 0; i < G; ++i) thr = fminf(thr, __hip_atomic_load(part + (size_t)b * G + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    int cnt = 0;
    for (int t = tid; t < n; t += 512) cnt += I_s[t] > thr ? 1 : 0;
    const float total = block_sum((float)cnt, red, tid, 8);  // exact: counts <= 1024
    if (tid == 0) {
        tick[b] = 0;  // the scratch is clean again for the next launch on this slot
        threshold[b] = thr;
        count[b] = (int)total;
        int last = 0;
        if (done_ctr) {
            __threadfence();
            last = atomicAdd(done_ctr, 1) == nsamp - 1;
        }
        last_s = last;
    }
    if (!done_ctr) return;
    __syncthreads();
    if (!last_s) return;
    int mloc = 0;
    for (int i = tid; i < nsamp; i += 512) mloc = max(mloc, __hip_atomic_load(count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int kk = (int)block_max_f((float)mloc, red, tid, 8);
    if (tid == 0) {
        *done_ctr = 0;
        __hip_atomic_store(host_slot, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_slot + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ----------------------------------------------------------------------------------------------- token_select
// NTHR threads per sample: 256, or 1024 for long sequences (the ranking is n^2 / NTHR compare steps per thread: 35 us at 900
// tokens with 256 threads).  The sum of the dropped scores keeps the 256-thread association in both variants (same bits).
// Long sequences (n > 320: 577 / 901 tokens) at small batches: the n^2 ranking of one sample on ONE workgroup was 20 us of
// VALU work on 32 of 256 CUs (VQA).  token_rank_kernel spreads it: a workgroup ranks 64 tokens, four lanes per token (each counts a
// quarter of the other tokens, two quad shuffles add up), grid (ceil(n / 64), B); the ranks go to `rank_out` [B, n] (the caller
// passes dst_pos, which token_select_kernel overwrites afterwards) and indices_sort is scattered here.  Same keys, same tie rule:
// the same permutation as the in-kernel ranking.
__global__ __launch_bounds__(256) void token_rank_kernel(const float* __restrict__ score, int64_t* __restrict__ indices_sort,
                                                         int32_t* __restrict__ rank_out, int n) {
    __shared__ unsigned key_s[MAXN];
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int t = tid; t < n; t += 256) {
        const float v = score[(size_t)b * n + t];
        const unsigned u = __float_as_uint(v + 0.0f);
        key_s[t] = (v != v) ? 0u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
    }
    __syncthreads();
    const int t = blockIdx.x * 64 + (tid >> 2), q = tid & 3;
    const bool ok = t < n;
    const unsigned kv = ok ? key_s[t] : 0u;
    int r = 0;
    if (ok)
        for (int u = q; u < n; u += 4) {
            const unsigned kw = key_s[u];
            r += (kw > kv || (kw == kv && u < t)) ? 1 : 0;
        }
    r += __shfl_xor(r, 1);
    r += __shfl_xor(r, 2);
    if (ok && q == 0) {
        rank_out[(size_t)b * n + t] = r;
        indices_sort[(size_t)b * n + r] = t;
    }
}

template <int NTHR>
__global__ __launch_bounds__(NTHR) void token_select_kernel(const float* __restrict__ score, int k,
                                                            int64_t* __restrict__ indices, int64_t* __restrict__ indices_sort,
                                                            int32_t* __restrict__ dst_pos, float* __restrict__ merge_w, int n,
                                                            const int32_t* dims_l = nullptr, bool ranked = false) {
    // sync-free encoder path: n and k from the layer's device-side record; k == 0 there means "not pruned" (vit.py:148-149): the
    // kernel then writes the identity map (every token kept in place, no merge weights) for the gather that follows
    bool ident = false;
    if (dims_l) { n = dims_l[0] - 1; k = dims_l[2]; ident = k == 0; }
    constexpr int NW = NTHR / 64;
    __shared__ float s[MAXN];
    __shared__ unsigned key_s[MAXN];
    __shared__ int rank_s[MAXN];
    __shared__ float red[NW];
    __shared__ int wsum[NW];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    // The ranking runs on an ORDER-PRESERVING integer key of the score, so it is a total order whatever the scores are: for
    // ordinary numbers key(w) > key(v) <=> w > v (-0 is folded onto +0 first), and a NaN score (NaN / Inf upstream) ranks below
    // everything instead of tying with every token - every rank 0..n-1 is produced exactly once, indices / indices_sort /
    // dst_pos are always fully written permutations and no later kernel reads an uninitialised index.
    for (int t = tid; t < n; t += NTHR) {
        const float v = score[(size_t)b * n + t];
        s[t] = v;
        const unsigned u = __float_as_uint(v + 0.0f);
        key_s[t] = (v != v) ? 0u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
    }
    if (tid == 0) base_s = 0;
    __syncthreads();
    // rank by counting: #tokens with a larger score (ties: lower index first) == position in a stable descending sort
    // (ranked: token_rank_kernel did it and left the ranks in dst_pos - read them all before anything below overwrites dst_pos)
    for (int t = tid; t < n; t += NTHR) {
        if (ranked) {
            rank_s[t] = dst_pos[(size_t)b * n + t];
            continue;
        }
        const unsigned kv = key_s[t];
        int r = 0;
        for (int u = 0; u < n; ++u) {
            const unsigned kw = key_s[u];
            r += (kw > kv || (kw == kv && u < t)) ? 1 : 0;
        }
        rank_s[t] = r;
        indices_sort[(size_t)b * n + r] = t;
    }
    __syncthreads();
    float dsum_l = 0.f;  // thread tid < 256 sums its tokens tid, tid + 256, ... in that order (the other threads add +0)
    if (tid < 256)
        for (int t = tid; t < n; t += 256)
            if (rank_s[t] >= k) dsum_l += s[t];
    const float dsum = block_sum(dsum_l, red, tid, NW);  // sum of dropped scores (vit.py:158-159)
    // stable compaction of kept tokens in ascending token order: ballot + popcount prefix per wave, serial over chunks
    for (int c0 = 0; c0 < n; c0 += NTHR) {
        const int t = c0 + tid;
        const bool keep = t < n && (ident || rank_s[t] < k);
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (t < n) {
            if (keep) {
                if (!ident) indices[(size_t)b * k + off + before] = t;
                dst_pos[(size_t)b * n + t] = off + before;
                merge_w[(size_t)b * n + t] = 0.f;
            } else {
                dst_pos[(size_t)b * n + t] = -1;
                merge_w[(size_t)b * n + t] = s[t] / (dsum + 1e-8f);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < NW; ++w) tot += wsum[w];
            base_s += tot;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------- token_gather
// grid (chunks+1, B): chunk c copies source tokens [16c,16c+16) that survive (token 0 = CLS -> slot 0); the extra
// last workgroup of each sample builds the merged token (fixed wave/token order, LDS combine).
constexpr int GATHER_ROWS = 16;
// Optional fused LayerNorm (the Block's norm2, vit.py:195): the wave that copies a row holds it in registers, so it also
// emits LN(row) (f32 and/or compute dtype) - the separate LayerNorm kernel re-read the whole gathered tensor.  The
// arithmetic is the row routine of layernorm_kernel (common.h), so the result is bit-identical to the two-kernel path.
__global__ __launch_bounds__(256) void token_gather_kernel(const float* __restrict__ x, const int32_t* __restrict__ dst_pos,
                                                           const float* __restrict__ merge_w, float* __restrict__ y, int N,
                                                           int k, int dim4, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float* h32, bf16_t* hlp,
                                                           int lp_fmt, int* range_flag, const int32_t* dims_l = nullptr) {
    __shared__ float4 part[4][256];  // dim <= 1024
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool ident = false;  // sync-free encoder path: N, k from the device-side record; k == 0: not pruned, every token is copied
    if (dims_l) { N = dims_l[0]; k = dims_l[2]; ident = k == 0; }
    const int b = blockIdx.y, n = N - 1, No = ident ? N : k + 2, dim = dim4 * 4;
    const float4* xb = (const float4*)x + (size_t)b * N * dim4;
    float4* yb = (float4*)y + (size_t)b * No * dim4;
    const bool ln = gamma != nullptr;
    LnParams prm;
    if (ln) prm = ln_params(gamma, beta, lane, dim);
    auto ln_out = [&](float4 (&v)[LN_MAX_CHUNKS], int nch, int dst) {
        float mean, rstd;
        ln_row(v, nch, dim, eps, mean, rstd);
        const size_t off = ((size_t)b * No + dst) * dim;
        ln_store(v, lane, dim, mean, rstd, prm, h32 ? h32 + off : nullptr, hlp ? hlp + off * lp_row_mul(lp_fmt) : nullptr, lp_fmt, range_flag);
    };
    if ((int)blockIdx.x < (int)gridDim.x - 1) {
        // A wave copies the source tokens t0 + wave + 4 i (i < 4).  The straightforward loop was a dependent chain per row
        // (destination slot, then the row, then the LayerNorm reductions and stores): the four slots are fetched up front now
        // and row i + 1 is in flight while row i is normalised and stored (same arithmetic per row: bit-identical).
        constexpr int RPW = GATHER_ROWS / 4;
        int dsts[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int t = blockIdx.x * GATHER_ROWS + wave + 4 * i;  // source token incl. CLS
            dsts[i] = t >= N ? -1 : (t == 0 ? 0 : dst_pos[(size_t)b * n + t - 1] + 1);
            if (t != 0 && dsts[i] <= 0) dsts[i] = -1;  // dropped (or past the end)
        }
        float4 v[2][LN_MAX_CHUNKS];
        auto load_row = [&](float4 (&r)[LN_MAX_CHUNKS], int i) {
            const int t = blockIdx.x * GATHER_ROWS + wave + 4 * i;
#pragma unroll
            for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
                const int cc = lane + 64 * c;
                r[c] = (dsts[i] >= 0 && cc < dim4) ? xb[(size_t)t * dim4 + cc] : make_float4(0, 0, 0, 0);
            }
        };
        const int nch = lane < dim4 ? (dim4 - lane + 63) / 64 : 0;  // chunks lane + 64 c < dim4 of this lane
        load_row(v[0], 0);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            if (i + 1 < RPW) load_row(v[(i + 1) & 1], i + 1);
            if (dsts[i] >= 0) {  // wave-uniform
#pragma unroll
                for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
                    const int cc = lane + 64 * c;
                    if (cc < dim4) yb[(size_t)dsts[i] * dim4 + cc] = v[i & 1][c];
                }
                if (ln) ln_out(v[i & 1], nch, dsts[i]);
            }
        }
    } else if (!ident) {
        // Merged token: wave w sums its dropped tokens t = w, w+4, .. in increasing order (the association of the result is fixed
        // by that).  The straightforward loop was a chain of dependent loads per token (weight / position, then the row): ~1 us
        // per token, 49 (224 at 901 tokens) iterations per wave - the longest workgroup of the launch.  Now each wave first
        // compacts the list of its dropped tokens (ballot over 64 candidates at a time, list in LDS), then walks the list with
        // the rows of FOUR tokens in flight; the sums are accumulated in list order, so the result is bit-identical.
        __shared__ int lst_t[4][256];
        __shared__ float lst_w[4][256];
        int cnt = 0;
        for (int base = 0; base < n; base += 256) {  // candidates t = wave + 4 * lane + base, 64 per pass
            const int t = base + wave + 4 * lane;
            bool drop = false;
            float w = 0.f;
            if (t < n) {
                drop = dst_pos[(size_t)b * n + t] < 0;
                w = merge_w[(size_t)b * n + t];
            }
            const unsigned long long bal = __ballot(drop);
            if (drop) {
                const int pos = cnt + __popcll(bal & ((1ull << lane) - 1ull));
                lst_t[wave][pos] = t;
                lst_w[wave][pos] = w;
            }
            cnt += __popcll(bal);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // the wave's own list writes (same wave reads them: no barrier needed)
        float4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = make_float4(0, 0, 0, 0);
        constexpr int DEPTH = 4;
        for (int i0 = 0; i0 < cnt; i0 += DEPTH) {
            float4 v[DEPTH][4];
            float wv[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int i = min(i0 + u, cnt - 1);
                const int t = lst_t[wave][i];
                wv[u] = (i0 + u < cnt) ? lst_w[wave][i] : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cc = lane + 64 * c;
                    v[u][c] = cc < dim4 ? xb[(size_t)(t + 1) * dim4 + cc] : make_float4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                if (i0 + u < cnt) {  // wave-uniform
                    const float w = wv[u];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c].x += w * v[u][c].x; acc[c].y += w * v[u][c].y; acc[c].z += w * v[u][c].z; acc[c].w += w * v[u][c].w;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) part[wave][lane + 64 * c] = acc[c];
        __syncthreads();
        for (int c = tid; c < dim4; c += 256) {
            const float4 p0 = part[0][c], p1 = part[1][c], p2 = part[2][c], p3 = part[3][c];
            const float4 f = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                         ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
            yb[(size_t)(k + 1) * dim4 + c] = f;
            part[0][c] = f;  // only this thread touches column c of part[0] from here on
        }
        if (ln) {
            __syncthreads();
            if (wave == 0) {
                float4 v[LN_MAX_CHUNKS];
                int nch = 0;
#pragma unroll
                for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
                    const int cc = lane + 64 * c;
                    if (cc < dim4) { v[c] = part[0][cc]; nch = c + 1; } else v[c] = make_float4(0, 0, 0, 0);
                }
                ln_out(v, nch, k + 1);
            }
        }
    }
}

// order2 == NULL: slot p takes mask[1+order[p]] for p in [0,k]            (NLVR: order = indices_sort)
// order2 != NULL: slots p<k take mask[1+order[p]] (order = indices), slot k takes mask[1+order2[k]] (order2 =
//                 indices_sort: the (k+1)-th ranked token, med.py:377,388-390)
// dims_l != NULL (sync-free encoder path): N, k from the layer's device-side record, the index rows at their DEVICE strides
// (order2 given: order = indices [B, k], order2 = indices_sort [B, N-1]; else order = indices_sort [B, N-1]); k == 0: the layer is
// not pruned and the mask is copied.
__global__ void mask_gather_kernel(const float* __restrict__ mask, const int64_t* __restrict__ order, int ld_order,
                                   const int64_t* __restrict__ order2, int ld_order2, float* __restrict__ out, int N, int k,
                                   const int32_t* dims_l = nullptr) {
    const int b = blockIdx.x;
    if (dims_l) {
        N = dims_l[0]; k = dims_l[2];
        if (k == 0) {
            for (int p = threadIdx.x; p < N; p += blockDim.x) out[(size_t)b * N + p] = mask[(size_t)b * N + p];
            return;
        }
        ld_order = order2 ? k : N - 1;
        ld_order2 = N - 1;
    }
    for (int p = threadIdx.x; p < k + 2; p += blockDim.x) {
        float v;
        if (p == 0) v = mask[(size_t)b * N];
        else if (order2 && p - 1 == k) v = mask[(size_t)b * N + 1 + order2[(size_t)b * ld_order2 + k]];
        else v = mask[(size_t)b * N + 1 + order[(size_t)b * ld_order + p - 1]];
        out[(size_t)b * (k + 2) + p] = v;
    }
}

__global__ __launch_bounds__(256) void vector_gather_kernel(const float* __restrict__ v, const int64_t* __restrict__ idx,
                                                            float* __restrict__ out, int L, int K, int dim4, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / K;
    const int64_t src = idx[row];
    const float4* s = (const float4*)v + ((size_t)b * L + src) * dim4;
    float4* o = (float4*)out + (size_t)row * dim4;
    for (int c = lane; c < dim4; c += 64) o[c] = s[c];
}

// ----------------------------------------------------------------------------------------------- query_att_ft
// grid (dim/256, B); 4 waves, wave w owns output columns [256*bx + 64w, +64) (4 MFMA column tiles) for all K<=112
// dictionary rows (7 row tiles): C[c, d] = sum_t w[c,t] * ft[t, d] with the exact-f32 MFMA 16x16x4
// (A = w: row c, k-slot g <-> t = 4*step+g;  B = ft: k-slot g, column d).  The softmax weights of a 128-token chunk
// are computed ONCE per workgroup into LDS (they used to be re-evaluated by every wave of every column block) so the
// MFMA loop only issues ds_read_b32 + coalesced 64-byte row-segment loads.
constexpr int AF_TCHUNK = 128, AF_KP = 112;
__global__ __launch_bounds__(256) void query_att_ft_kernel(const float* __restrict__ ta, int ldt, int ldb, int K,
                                                           const float* __restrict__ ft, int ldf, int ldfb,
                                                           float* __restrict__ out, float inv_sqrt_sd, int accumulate,
                                                           int n, int dim) {
    __shared__ float mx[128], sm[128];
    __shared__ float part[2][128];
    __shared__ float wl[AF_TCHUNK * AF_KP];  // [t][c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const float* ta_b = ta + (size_t)b * ldb;
    // column softmax statistics over tokens: 2 token slices x 128 columns
    {
        const int c = tid & 127, sl = tid >> 7;
        const int t0 = sl ? n / 2 : 0, t1 = sl ? n : n / 2;
        float m = -INFINITY;
        if (c < K) {
#pragma unroll 8
            for (int t = t0; t < t1; ++t) m = fmaxf(m, ta_b[(size_t)t * ldt + c] * inv_sqrt_sd);
        }
        part[sl][c] = m;
        __syncthreads();
        m = fmaxf(part[0][c], part[1][c]);
        __syncthreads();
        float s = 0.f;
        if (c < K) {
#pragma unroll 8
            for (int t = t0; t < t1; ++t) s += expf(ta_b[(size_t)t * ldt + c] * inv_sqrt_sd - m);
        }
        part[sl][c] = s;
        __syncthreads();
        if (tid < 128) { mx[tid] = m; sm[tid] = c < K ? part[0][c] + part[1][c] : 1.f; }
        __syncthreads();
    }
    const int d0 = blockIdx.x * 256 + wave * 64 + l16;
    f32x4 acc[7][4];
#pragma unroll
    for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xb = ft + (size_t)b * ldfb;
    for (int tc = 0; tc < n; tc += AF_TCHUNK) {
        const int tn = min(AF_TCHUNK, n - tc);
        const int tn4 = (tn + 3) & ~3;
        __syncthreads();
        for (int idx = tid; idx < tn4 * AF_KP; idx += 256) {
            const int t = idx / AF_KP, c = idx % AF_KP;
            float w = 0.f;
            if (t < tn && c < K) w = expf(ta_b[(size_t)(tc + t) * ldt + c] * inv_sqrt_sd - mx[c]) / sm[c];
            wl[idx] = w;
        }
        __syncthreads();
        for (int t4 = 0; t4 < tn; t4 += 4) {
            const int t = t4 + g;
            float xv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) xv[nt] = t < tn ? xb[(size_t)(tc + t) * ldf + d0 + 16 * nt] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 7; ++mt) {
                const float w = wl[t * AF_KP + mt * 16 + l16];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = mt * 16 + g * 4 + r;
            if (c < K) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float* o = out + ((size_t)b * K + c) * dim + d0 + 16 * nt;
                    *o = accumulate ? *o + acc[mt][nt][r] : acc[mt][nt][r];
                }
            }
        }
}

// ----------------------------------------------------------------------------------------- query_att_ft (bf16 MFMA)
// Fast-mode variant: same decomposition, but the product runs on v_mfma_f32_16x16x32_bf16.  Both operands need the
// TOKEN index along the MFMA k slots while the data is token-major in memory, so per 64-token chunk the softmax
// weights Wt[t][c] and the token rows Xs[t][d] are staged ROW-major as bf16 (coalesced reads, conflict-free 8-byte
// LDS writes) and BOTH fragments are fetched with the hardware transpose read ds_read_b64_tr_b16.  Pitches 144 /
// 272 elements put the 8 rows of a 32-lane access on disjoint bank octets.
// The kernel walks a LIST of (logits, token rows) segments - the layers of an encoder - and keeps the [K, 256] output
// block in registers across them: att_ft_all = sum_l softmax_t(logits_l)^T x_l is written once instead of being
// read-modified-written (B*K*dim f32 = 39 MB at B=128) after every layer.
// Two kernels: att_ft_stats_kernel (grid nseg x B) computes the softmax statistics of every (segment, sample) - the
// maximum and 1/sum over tokens of each dictionary column, two passes with four independent float4 loads in flight per
// thread; query_att_ft_mma_kernel then streams ALL 32-token chunks of the sample's segments as one sequence: the next
// chunk's logits and token rows are fetched into registers while the current one is multiplied, converted (softmax weight /
// bf16) into one of two LDS buffers, one barrier per chunk.
// Workgroup = one sample x 128 output columns (wave w: columns [32w, 32w+32), 2 MFMA column tiles): 6 x B workgroups of
// 56 accumulator registers, three resident per CU.
constexpr int AB_TCH = 32, AB_WP = 144, AB_COLS = 128, AB_XP = AB_COLS + 16, AB_NT = AB_COLS / 64, AB_MAXSEG = 16, AB_SL = 9;
struct AttFtSeg { const float* ta; const float* ft; int n, ldt, ldb, ldf, ldfb; };
struct AttFtSegs { AttFtSeg s[AB_MAXSEG]; int nseg; };

// Parity-mode counterpart of the segment walk (exact-f32 MFMA, expf, true division - the arithmetic of query_att_ft_kernel):
// per segment the block [K, 256] of the sample is accumulated from zero in one register set and then added to the running
// sum in a second one, i.e. exactly the values `out += att_ft_l` produced layer by layer - without the read-modify-write of the
// [B, K, dim] sum per layer and in one launch (it can run on the auxiliary stream like the fast-mode kernel).
__global__ __launch_bounds__(256) void query_att_ft_exact_multi_kernel(AttFtSegs segs, int K, float* __restrict__ out,
                                                                       float inv_sqrt_sd, int accumulate, int dim) {
    __shared__ float mx[128], sm[128];
    __shared__ float part[2][128];
    __shared__ float wl[AF_TCHUNK * AF_KP];  // [t][c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int d0 = blockIdx.x * 256 + wave * 64 + l16;
    f32x4 tot[7][4];
#pragma unroll
    for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) tot[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int si = 0; si < segs.nseg; ++si) {
        const AttFtSeg sg = segs.s[si];
        const int n = sg.n, ldt = sg.ldt, ldf = sg.ldf;
        const float* ta_b = sg.ta + (size_t)b * sg.ldb;
        const float* xb = sg.ft + (size_t)b * sg.ldfb;
        __syncthreads();
        {
            const int c = tid & 127, sl = tid >> 7;
            const int t0 = sl ? n / 2 : 0, t1 = sl ? n : n / 2;
            float m = -INFINITY;
            if (c < K) {
#pragma unroll 8
                for (int t = t0; t < t1; ++t) m = fmaxf(m, ta_b[(size_t)t * ldt + c] * inv_sqrt_sd);
            }
            part[sl][c] = m;
            __syncthreads();
            m = fmaxf(part[0][c], part[1][c]);
            __syncthreads();
            float s = 0.f;
            if (c < K) {
#pragma unroll 8
                for (int t = t0; t < t1; ++t) s += expf(ta_b[(size_t)t * ldt + c] * inv_sqrt_sd - m);
            }
            part[sl][c] = s;
            __syncthreads();
            if (tid < 128) { mx[tid] = m; sm[tid] = c < K ? part[0][c] + part[1][c] : 1.f; }
            __syncthreads();
        }
        f32x4 acc[7][4];
#pragma unroll
        for (int mt = 0; mt < 7; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int tc = 0; tc < n; tc += AF_TCHUNK) {
            const int tn = min(AF_TCHUNK, n - tc);
            const int tn4 = (tn + 3) & ~3;
            __syncthreads();
            for (int idx = tid; idx < tn4 * AF_KP; idx += 256) {
                const int t = idx / AF_KP, c = idx % AF_KP;
                float w = 0.f;
                if (t < tn && c < K) w = expf(ta_b[(size_t)(tc + t) * ldt + c] * inv_sqrt_sd - mx[c]) / sm[c];
                wl[idx] = w;
            }
            __syncthreads();
            for (int t4 = 0; t4 < tn; t4 += 4) {
                const int t = t4 + g;
                float xv[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) xv[nt] = t < tn ? xb[(size_t)(tc + t) * ldf + d0 + 16 * nt] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 7; ++mt) {
                    const float w = wl[t * AF_KP + mt * 16 + l16];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xv[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        // running sum in layer order: the first segment of a non-accumulating call initialises it (x + 0 would also be exact,
        // but -0 + 0 is not -0), later ones add - the same roundings as `out = out + att_ft_l` per layer
        const bool first = si == 0 && !accumulate;
#pragma unroll
        for (int mt = 0; mt < 7; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) tot[mt][nt] = first ? acc[mt][nt] : tot[mt][nt] + acc[mt][nt];
        if (si == 0 && accumulate) {  // continue a sum that already lives in `out`: out + att_ft_0 first
#pragma unroll
            for (int mt = 0; mt < 7; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = mt * 16 + g * 4 + r;
                    if (c < K) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) tot[mt][nt][r] = out[((size_t)b * K + c) * dim + d0 + 16 * nt] + acc[mt][nt][r];
                    }
                }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = mt * 16 + g * 4 + r;
            if (c < K) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) out[((size_t)b * K + c) * dim + d0 + 16 * nt] = tot[mt][nt][r];
            }
        }
}

// stats[(seg*B + b)*256 + c] = max_t logit*inv, [.. + 128 + c] = 1 / sum_t exp(logit*inv - max)  (0 for c >= K)
// EXACT (the f16-split att_ft of the f16x3 mode): expf, and the SUM itself is stored (the weights are then true quotients)
template <bool EXACT>
__global__ __launch_bounds__(256) void att_ft_stats_kernel(AttFtSegs segs, int K, float inv_sqrt_sd, float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float part[AB_SL][128];
    __shared__ __attribute__((aligned(16))) float mx[128];
    const int tid = threadIdx.x, si = blockIdx.x, b = blockIdx.y;
    const int K4 = (K + 3) & ~3;
    const int c4 = (tid % 28) * 4, sl = tid / 28;
    const float* ta_b = segs.s[si].ta + (size_t)b * segs.s[si].ldb + c4;
    const int n = segs.s[si].n, ldt = segs.s[si].ldt;
    const bool act = sl < AB_SL && c4 < K4;
    const f32x4 ninf = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 m = ninf;
    if (act)
        for (int t = sl; t < n; t += 4 * AB_SL) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t + u * AB_SL < n ? *(const f32x4*)(ta_b + (size_t)(t + u * AB_SL) * ldt) : ninf;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[u][e] * inv_sqrt_sd);
        }
    if (sl < AB_SL) *(f32x4*)(&part[sl][c4]) = m;
    __syncthreads();
    if (tid < 128) {
        float mm = part[0][tid];
#pragma unroll
        for (int q = 1; q < AB_SL; ++q) mm = fmaxf(mm, part[q][tid]);
        mx[tid] = mm;
    }
    __syncthreads();
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (act) {
        const f32x4 mc = *(const f32x4*)(&mx[c4]);
        for (int t = sl; t < n; t += 4 * AB_SL) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t + u * AB_SL < n ? *(const f32x4*)(ta_b + (size_t)(t + u * AB_SL) * ldt) : ninf;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += EXACT ? expf(v[u][e] * inv_sqrt_sd - mc[e]) : __expf(v[u][e] * inv_sqrt_sd - mc[e]);
        }
    }
    if (sl < AB_SL) *(f32x4*)(&part[sl][c4]) = sum;
    __syncthreads();
    if (tid < 128) {
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < AB_SL; ++q) ss += part[q][tid];
        float* o = stats + ((size_t)si * gridDim.y + b) * 256;
        o[tid] = tid < K ? mx[tid] : 0.f;
        // fast: 1 / sum (weight 0 for the padding columns K..127); exact: the sum itself (the kernel zeroes the padding columns)
        o[128 + tid] = EXACT ? (tid < K ? ss : 1.f) : (tid < K ? 1.0f / ss : 0.f);
    }
}

// SPLIT = false: the fast mode's kernel (bf16 operands, __expf, reciprocal sums).  SPLIT = true: the f16x3 mode's - token rows and
// softmax weights (expf, true division) as f16-split planes, three f16 MFMA products per (weight tile, column tile) in two
// accumulator sets (hi, cross terms), out = hi + 2^-11 lo: the rounding class of the exact-f32 kernel at a fraction of its time
// (that one issues its token-row loads right in front of the MFMAs that use them: 1.2 ms per ViT encoder at the headline batch).
template <bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 1 : 3) void query_att_ft_mma_kernel(AttFtSegs segs, int K, const float* __restrict__ stats,
                                                                              float* __restrict__ out, float inv_sqrt_sd,
                                                                              int accumulate, int dim) {
    constexpr int NP = SPLIT ? 2 : 1;  // operand planes
    extern __shared__ __attribute__((aligned(16))) char smem_af[];
    float (*st)[256] = (float (*)[256])smem_af;                                     // per segment: max[128] | 1/sum or sum [128]
    bf16_t* const Wt = (bf16_t*)(smem_af + AB_MAXSEG * 256 * 4);                    // [2 buffers][NP planes][AB_TCH * AB_WP]
    bf16_t* const Xs = Wt + 2 * NP * AB_TCH * AB_WP;                                // [2 buffers][NP planes][AB_TCH * AB_XP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, B = gridDim.y;
    const int dblk = blockIdx.x * AB_COLS;
    const int K4 = (K + 3) & ~3;  // logits are read as float4s: columns K..K4-1 exist (row pitch >= K4) and get weight 0
    f32x4 acc[7][AB_NT], acl[SPLIT ? 7 : 1][AB_NT];
#pragma unroll
    for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int nt = 0; nt < AB_NT; ++nt) {
            acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) acl[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    int nchunks = 0;
    for (int si = 0; si < segs.nseg; ++si) {
        st[si][tid] = stats[((size_t)si * B + b) * 256 + tid];
        nchunks += (segs.s[si].n + AB_TCH - 1) / AB_TCH;
    }
    constexpr int XL = AB_TCH * AB_COLS / 4 / 256, XC4 = AB_COLS / 4;  // float4 loads per thread, float4s per row
    f32x4 xr[XL], lr[4];
    int f_si = 0, f_tc = 0, r_si = 0;  // next chunk to fetch; segment of the chunk held in registers
    auto fetch = [&]() {
        const float* ta_b = segs.s[f_si].ta + (size_t)b * segs.s[f_si].ldb;
        const float* xb = segs.s[f_si].ft + (size_t)b * segs.s[f_si].ldfb + dblk;
        const int n = segs.s[f_si].n, ldt = segs.s[f_si].ldt, ldf = segs.s[f_si].ldf;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int t = f_tc + tid / XC4 + (256 / XC4) * i;
            xr[i] = t < n ? *(const f32x4*)(xb + (size_t)t * ldf + (tid % XC4) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, t = f_tc + idx / 28, c4 = (idx % 28) * 4;
            lr[i] = (idx < AB_TCH * 28 && t < n && c4 < K4) ? *(const f32x4*)(ta_b + (size_t)t * ldt + c4)
                                                            : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        r_si = f_si;
        f_tc += AB_TCH;
        if (f_tc >= n) { f_tc = 0; ++f_si; }
    };
    auto commit = [&](int buf) {  // registers -> LDS: token rows and softmax weights (exp(-inf) = 0 pads) in the operand format
        bf16_t* const xs = Xs + buf * NP * AB_TCH * AB_XP;
        bf16_t* const wt = Wt + buf * NP * AB_TCH * AB_WP;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int o = (tid / XC4 + (256 / XC4) * i) * AB_XP + (tid % XC4) * 4;
            if constexpr (SPLIT) {
                f16x4 h4, l4;
                split_f16x4(xr[i], h4, l4);
                *(f16x4*)(xs + o) = h4;
                *(f16x4*)(xs + AB_TCH * AB_XP + o) = l4;
            } else {
                *(bf16x4*)(xs + o) = pack_bf16x4(xr[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, t = idx / 28, c4 = (idx % 28) * 4;
            if (idx < AB_TCH * 28) {
                const f32x4 mc = *(const f32x4*)(&st[r_si][c4]), sc = *(const f32x4*)(&st[r_si][128 + c4]);
                f32x4 w;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = c4 + e < K ? expf(lr[i][e] * inv_sqrt_sd - mc[e]) / sc[e] : 0.f;
                    f16x4 h4, l4;
                    split_f16x4(w, h4, l4);
                    *(f16x4*)(wt + t * AB_WP + c4) = h4;
                    *(f16x4*)(wt + AB_TCH * AB_WP + t * AB_WP + c4) = l4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = __expf(lr[i][e] * inv_sqrt_sd - mc[e]) * sc[e];
                    *(bf16x4*)(wt + t * AB_WP + c4) = pack_bf16x4(w);
                }
            }
        }
    };
    fetch();
    __syncthreads();  // st[] visible
    commit(0);
    if (nchunks > 1) fetch();
    lds_barrier();  // (not __syncthreads(): the fetches stay in flight across the barriers)
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) {
            commit((ch + 1) & 1);  // that buffer's readers finished before the barrier that ended chunk ch-1
            if (ch + 2 < nchunks) fetch();
        }
        const bf16_t* xs = Xs + (ch & 1) * NP * AB_TCH * AB_XP;
        const bf16_t* wt = Wt + (ch & 1) * NP * AB_TCH * AB_WP;
        // lane 4r+q of a 16-lane group addresses (token row r, columns 4q..4q+3); k slots 8g..8g+7 = two reads
        const int trow = 8 * g + (l16 >> 2), cpart = 4 * (l16 & 3);
        bf16x8 xf[AB_NT], xl[SPLIT ? AB_NT : 1];
#pragma unroll
        for (int nt = 0; nt < AB_NT; ++nt) {
            const bf16_t* p = xs + trow * AB_XP + wave * (16 * AB_NT) + nt * 16 + cpart;
            xf[nt] = cat_bf16x4(lds_read_tr16(p), lds_read_tr16(p + 4 * AB_XP));
            if constexpr (SPLIT) xl[nt] = cat_bf16x4(lds_read_tr16(p + AB_TCH * AB_XP), lds_read_tr16(p + AB_TCH * AB_XP + 4 * AB_XP));
        }
#pragma unroll
        for (int mt = 0; mt < 7; ++mt) {
            const bf16_t* p = wt + trow * AB_WP + mt * 16 + cpart;
            const bf16x8 wf = cat_bf16x4(lds_read_tr16(p), lds_read_tr16(p + 4 * AB_WP));
            if constexpr (SPLIT) {
                const bf16x8 wl = cat_bf16x4(lds_read_tr16(p + AB_TCH * AB_WP), lds_read_tr16(p + AB_TCH * AB_WP + 4 * AB_WP));
#pragma unroll
                for (int nt = 0; nt < AB_NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xf[nt]), __builtin_bit_cast(f16x8, wf), acc[mt][nt], 0, 0, 0);
                    acl[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xf[nt]), __builtin_bit_cast(f16x8, wl), acl[mt][nt], 0, 0, 0);
                    acl[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xl[nt]), __builtin_bit_cast(f16x8, wf), acl[mt][nt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < AB_NT; ++nt)  // operands swapped: lane (c = l16, g) ends up with 4 consecutive d
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[nt], wf, acc[mt][nt], 0, 0, 0);
            }
        }
        lds_barrier();
    }
    // epilogue: D[row = d-fragment row 4g+r][col = c]: one float4 (read-modify-)write per (mt, nt)
#pragma unroll
    for (int mt = 0; mt < 7; ++mt) {
        const int c = mt * 16 + l16;
        if (c >= K) continue;
        float* orow = out + ((size_t)b * K + c) * dim + dblk + wave * (16 * AB_NT) + 4 * g;
#pragma unroll
        for (int nt = 0; nt < AB_NT; ++nt) {
            f32x4* o = (f32x4*)(orow + 16 * nt);
            f32x4 v = acc[mt][nt];
            if constexpr (SPLIT) v += acl[mt][nt] * (1.0f / F16S_LO_SCALE);
            *o = accumulate ? *o + v : v;
        }
    }
}
constexpr size_t att_ft_mma_lds(bool split) {
    return (size_t)AB_MAXSEG * 256 * 4 + (size_t)2 * (split ? 2 : 1) * AB_TCH * (AB_WP + AB_XP) * 2;
}

// ------------------------------------------------------------------------------------- alignment logits (bf16 x 3)
// token_attn = x @ sd^T for the fast mode: f32 accuracy class on the bf16 matrix cores via a two-term split
// x = xh + xl, sd = sh + sl (bf16 each):  x.sd ~= xh.sh + xl.sh + xh.sl  (the dropped xl.sl term is 2^-16 relative).
// x rows are read as f32 straight into registers and split there; the 128 dictionary rows (hi and lo slabs) are
// LDS-DMA'd in 128-byte K slabs with the GEMM kernel's swizzle; workgroup = 64 token rows x 128 columns, one 16-row
// MFMA fragment per wave, operand-swapped MFMA so each lane stores float4s.
constexpr int AL_ROWB = 128, AL_TILE = 128 * AL_ROWB, AL_STAGES = 3;
// MADTP_AL_ABLATE (tools/build_ablate.py, timing experiments only): bit 0 drops the MFMAs, bit 1 the x loads, bit 2 the
// LDS-DMAs, bit 3 the fragment reads.
#ifndef MADTP_AL_ABLATE
#define MADTP_AL_ABLATE 0
#endif
// NK = dim/64 as a compile-time constant fully unrolls the slab loop: in a rolled loop the compiler carries the in-flight
// x registers through copies at the back edge, and a copy of a pending load is a wait for it (NK = 0: runtime loop).
template <int NK>
__global__ __launch_bounds__(256, 1) void align_logits_kernel(const float* __restrict__ x, const char* __restrict__ sd_hi,
                                                              const char* __restrict__ sd_lo, float* __restrict__ out, int M,
                                                              int dim, DevN m_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // AL_STAGES x (hi tile, lo tile) = 96 KiB
    M = devn(m_dev, M);
    if ((int)blockIdx.x * 64 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 64 + wave * 16;
    const int row = min(m0 + l16, M - 1);
    const float* xr = x + (size_t)row * dim;
    const int nk = NK ? NK : dim * 2 / AL_ROWB;
    const int sub = lane >> 3, chunk_src = (lane & 7) ^ sub;
    auto stage = [&](int kt) {  // 8 LDS-DMA instructions per wave
        char* base = smem + (kt % AL_STAGES) * 2 * AL_TILE;
        if (MADTP_AL_ABLATE & 4) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int grp = wave * 4 + q;
            const size_t src = ((size_t)(grp * 8 + sub) * dim) * 2 + (size_t)kt * AL_ROWB + chunk_src * 16;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sd_hi + src), LDS_PTR(base + grp * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sd_lo + src), LDS_PTR(base + AL_TILE + grp * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's x slice of a slab: k = 64kt + 32kk + 8g .. +7, read as f32 straight into registers TWO slabs ahead
    // (the HBM latency of the token rows hides under two slabs of MFMAs); the dictionary slabs run two ahead as well,
    // through a 3-stage LDS ring synchronised with a counted s_waitcnt.
    f32x4 xq[2][2][2];  // register set kt&1 holds x(kt); it is refilled with x(kt+2) as soon as it has been split
    auto load_x = [&](int kt, f32x4 (&d)[2][2]) {
        if ((MADTP_AL_ABLATE & 2) && kt > 1) return;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const float* p = xr + kt * 64 + kk * 32 + g * 8;
            d[kk][0] = *(const f32x4*)p;
            d[kk][1] = *(const f32x4*)(p + 4);
        }
    };
    load_x(0, xq[0]);
    stage(0);
    load_x(1, xq[1]);  // nk is even (dim % 128 == 0, checked by the launcher)
    stage(1);
    auto slab = [&](int kt, f32x4 (&xs)[2][2]) {
        // vmcnt is one in-order counter: slab kt (and x(kt)) were issued two iterations ago, so everything issued in
        // the previous iteration - 4 x loads + 8 DMAs - may still be in flight
        if (MADTP_AL_ABLATE & 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // raw barrier, no fence: a fence would drain the LDS-DMAs in flight (the compiler tracks them as pending LDS
        // writes); the counted wait above is what makes slab kt visible, and every ds_read of slab kt-1 has been
        // consumed by an MFMA before its wave gets here
        __builtin_amdgcn_s_barrier();
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            ah[kk] = pack_bf16x8(xs[kk][0], xs[kk][1]);
            f32x4 r0, r1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r0[e] = xs[kk][0][e] - bf16_to_f32((bf16_t)ah[kk][e]);
                r1[e] = xs[kk][1][e] - bf16_to_f32((bf16_t)ah[kk][4 + e]);
            }
            al[kk] = pack_bf16x8(r0, r1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) { load_x(kt + 2, xs); stage(kt + 2); }
        const char* sh = smem + (kt % AL_STAGES) * 2 * AL_TILE;
        const char* sl = sh + AL_TILE;
        // Each accumulator's three products are issued in three passes over the 8 column tiles, so consecutive MFMAs
        // never depend on each other (the order per accumulator - hi.hi, hi.lo, lo.hi - and therefore the result is
        // unchanged); the kk=1 fragments are read under the first kk=0 pass (lgkmcnt is a 4-bit counter: at most 16
        // reads are kept in flight).
        bf16x8 bh[2][8], bl[2][8];
        auto read_frags = [&](int kk) {
            if ((MADTP_AL_ABLATE & 8) && kt > 0) return;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rb = j * 16 + l16;
                const int off = rb * AL_ROWB + (((kk * 4 + g) ^ (rb & 7)) << 4);
                bh[kk][j] = *(const bf16x8*)(sh + off);
                bl[kk][j] = *(const bf16x8*)(sl + off);
            }
        };
        read_frags(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(MADTP_AL_ABLATE & 1))
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[0][j], ah[0], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(MADTP_AL_ABLATE & 1)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[0][j], al[0], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[0][j], ah[0], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[1][j], ah[1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[1][j], al[1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[1][j], ah[1], acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NK > 0) {
#pragma unroll
        for (int kt = 0; kt < NK; kt += 2) {
            slab(kt, xq[0]);
            slab(kt + 1, xq[1]);
        }
    } else {
        for (int kt = 0; kt < nk; kt += 2) {
            slab(kt, xq[0]);
            slab(kt + 1, xq[1]);
        }
    }
    if (m0 + l16 < M) {
        float* o = out + (size_t)(m0 + l16) * 128 + 4 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) *(f32x4*)(o + 16 * j) = acc[j];
    }
}

// Wave-specialised variant (the one the fast mode runs): workgroup = 64 token rows x 128 columns as 8 CONSUMER waves (4 row
// tiles x 2 column halves: two MFMA instruction streams per SIMD) + 4 LOADER waves that LDS-DMA the f32 token rows (64 x
// 256 B per K slab) and the dictionary hi/lo slabs into a 3-stage 144 KiB ring with counted vmcnt waits, one raw barrier per
// slab.  The consumers never touch vector memory until their epilogue, so no HBM latency sits in the MFMA stream (the
// register-prefetching kernel above loses ~7 us per launch to it).  x rows are swizzled with chunk ^= row&15 (256-byte rows
// span all 64 banks), the dictionary with the GEMM swizzle.
// F16 = true is the fp32-ACCURATE flavour (precision mode "f16x3"): the dictionary arrives as the f16 planes Q0 / Q1 of
// sd * 2^s (common.h), x is split in registers into P0 = f16(x), P1 = f16((x - P0) 2^11), the three products
// P0 Q0 + P0 Q1 + P1 (Q0 2^-11) run on the f16 MFMA and the accumulators are scaled by out_scale = 2^-s.
constexpr int AW_XT = 64 * 256, AW_STAGE = AW_XT + 2 * AL_TILE;  // 16 KiB + 2 x 16 KiB
template <bool F16>
__global__ __launch_bounds__(768, 1) void align_ws_kernel(const float* __restrict__ x, const char* __restrict__ sd_hi,
                                                          const char* __restrict__ sd_lo, float* __restrict__ out, int M, int dim,
                                                          float out_scale, DevN m_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    M = devn(m_dev, M);
    if ((int)blockIdx.x * 64 >= M) return;
    constexpr int STAGES = 3, PER = 12;  // 48 one-KiB DMA instructions per slab, 12 per loader wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = dim / 64;
    const int row0 = blockIdx.x * 64;
    if (wave >= 8) {
        // ------------------------------------------ loader ------------------------------------------
        const int lw = wave - 8;
        const char* srcp[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = lw * PER + q;  // 0..15: x groups of 4 rows; 16..31: hi groups of 8 rows; 32..47: lo groups
            if (idx < 16) {
                const int r = idx * 4 + (lane >> 4);
                int row = row0 + r;
                row = row < M ? row : M - 1;
                srcp[q] = (const char*)(x + (size_t)row * dim) + (((lane & 15) ^ (r & 15)) << 4);
            } else {
                const int grp = (idx - 16) & 15, r = grp * 8 + (lane >> 3);
                const char* base = idx < 32 ? sd_hi : sd_lo;
                srcp[q] = base + (size_t)r * dim * 2 + (((lane & 7) ^ (r & 7)) << 4);
            }
        }
        auto issue = [&](int kt) {
            char* st = smem + (kt % STAGES) * AW_STAGE + lw * PER * 1024;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int idx = lw * PER + q;
                const int koff = idx < 16 ? kt * 256 : kt * 128;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(srcp[q] + koff), LDS_PTR(st + q * 1024), 16, 0, 0);
            }
        };
        issue(0);
        if (nk > 1) issue(1);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // only slab kt+1 may still fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) issue(kt + 2);  // into the stage slab kt-1 used: every consumer is past its reads
        }
        return;
    }
    // ------------------------------------------ consumer ------------------------------------------
    const int l16 = lane & 15, g = lane >> 4;
    const int rw = wave & 3, cw = wave >> 2;
    const int xr = rw * 16 + l16;  // row of the x tile
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + (kt % STAGES) * AW_STAGE;
        const char* sh = st + AW_XT;
        const char* sl = sh + AL_TILE;
        __builtin_amdgcn_s_barrier();  // slab kt landed (the loaders waited for it before arriving)
        f32x4 xa[2][2];
        bf16x8 bh[2][4], bl[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = kk * 8 + g * 2;  // 16-byte chunk of k = 32kk + 8g
            xa[kk][0] = *(const f32x4*)(st + xr * 256 + (((c + 0) ^ (xr & 15)) << 4));
            xa[kk][1] = *(const f32x4*)(st + xr * 256 + (((c + 1) ^ (xr & 15)) << 4));
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = (cw * 4 + j) * 16 + l16;
                const int off = rb * AL_ROWB + (((kk * 4 + g) ^ (rb & 7)) << 4);
                bh[kk][j] = *(const bf16x8*)(sh + off);
                bl[kk][j] = *(const bf16x8*)(sl + off);
            }
        bf16x8 ah[2], al[2];
        if constexpr (F16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 p0, p1;
                split_f16x8(xa[kk][0], xa[kk][1], p0, p1);
                ah[kk] = __builtin_bit_cast(bf16x8, p0);
                al[kk] = __builtin_bit_cast(bf16x8, p1);
            }
            // per accumulator: P0 Q1, P0 Q0, P1 (Q0 2^-11)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bl[kk][j]), __builtin_bit_cast(f16x8, ah[kk]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[kk][j]), __builtin_bit_cast(f16x8, ah[kk]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x8 q2 = __builtin_bit_cast(f16x8, bh[kk][j]) * (_Float16)(1.0f / F16S_LO_SCALE);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q2, __builtin_bit_cast(f16x8, al[kk]), acc[j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            ah[kk] = pack_bf16x8(xa[kk][0], xa[kk][1]);
            f32x4 r0, r1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r0[e] = xa[kk][0][e] - bf16_to_f32((bf16_t)ah[kk][e]);
                r1[e] = xa[kk][1][e] - bf16_to_f32((bf16_t)ah[kk][4 + e]);
            }
            al[kk] = pack_bf16x8(r0, r1);
        }
        // per accumulator: hi.hi, hi.lo, lo.hi (kk = 0), then the same for kk = 1 - the order of the kernel above
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kk][j], ah[kk], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kk][j], al[kk], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[kk][j], ah[kk], acc[j], 0, 0, 0);
        }
        }
    }
    const int m = row0 + xr;
    if (m < M) {
        float* o = out + (size_t)m * 128 + cw * 64 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(o + 16 * j) = F16 ? acc[j] * out_scale : acc[j];
    }
}

// 128-row variant of align_ws_kernel (round 6; the "taller row tile" of DESIGN section 9): one workgroup streams the split dictionary
// for 128 token rows instead of 64 - half the workgroups and half the dictionary bytes per launch.  Per K slab: 32 KiB of x rows +
// 2 x 16 KiB of dictionary = 64 KiB, TWO stages (three would not fit 160 KiB): consumers signal "fragments of slab kt are in
// registers" with a second barrier, after which the loaders refill that stage under the slab's MFMAs.  8 consumer waves = 4 row
// groups of 32 rows x 2 column halves: a wave's dictionary fragments serve two row tiles (half the LDS fragment reads per MFMA).
// Per accumulator the product order over (slab, kk, term) is align_ws_kernel's: the same bits.
constexpr int AW2_XT = 128 * 256, AW2_STAGE = AW2_XT + 2 * AL_TILE;  // 32 KiB + 2 x 16 KiB
template <bool F16>
__global__ __launch_bounds__(768, 1) void align_ws2_kernel(const float* __restrict__ x, const char* __restrict__ sd_hi,
                                                           const char* __restrict__ sd_lo, float* __restrict__ out, int M, int dim,
                                                           float out_scale, DevN m_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    M = devn(m_dev, M);
    if ((int)blockIdx.x * 128 >= M) return;
    constexpr int STAGES = 2, PER = 16;  // 64 one-KiB DMA instructions per slab, 16 per loader wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = dim / 64;
    const int row0 = blockIdx.x * 128;
    if (wave >= 8) {
        // ------------------------------------------ loader ------------------------------------------
        const int lw = wave - 8;
        const char* srcp[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = lw * PER + q;  // 0..31: x groups of 4 rows; 32..47: hi groups of 8 rows; 48..63: lo groups
            if (idx < 32) {
                const int r = idx * 4 + (lane >> 4);
                int row = row0 + r;
                row = row < M ? row : M - 1;
                srcp[q] = (const char*)(x + (size_t)row * dim) + (((lane & 15) ^ (r & 15)) << 4);
            } else {
                const int grp = (idx - 32) & 15, r = grp * 8 + (lane >> 3);
                const char* base = idx < 48 ? sd_hi : sd_lo;
                srcp[q] = base + (size_t)r * dim * 2 + (((lane & 7) ^ (r & 7)) << 4);
            }
        }
        auto issue = [&](int kt) {
            char* st = smem + (kt % STAGES) * AW2_STAGE + lw * PER * 1024;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int idx = lw * PER + q;
                const int koff = idx < 32 ? kt * 256 : kt * 128;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(srcp[q] + koff), LDS_PTR(st + q * 1024), 16, 0, 0);
            }
        };
        issue(0);
        if (nk > 1) issue(1);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // only slab kt+1 may still fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // A: slab kt landed
            __builtin_amdgcn_s_barrier();  // B: every consumer holds slab kt's fragments in registers
            if (kt + 2 < nk) issue(kt + 2);  // into the stage slab kt used
        }
        return;
    }
    // ------------------------------------------ consumer ------------------------------------------
    const int l16 = lane & 15, g = lane >> 4;
    const int rg = wave & 3, cw = wave >> 2;
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + (kt % STAGES) * AW2_STAGE;
        const char* sh = st + AW2_XT;
        const char* sl = sh + AL_TILE;
        __builtin_amdgcn_s_barrier();  // A
        f32x4 xa[2][2][2];
        bf16x8 bh[2][4], bl[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int xr = rg * 32 + t * 16 + l16;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = kk * 8 + g * 2;
                xa[t][kk][0] = *(const f32x4*)(st + xr * 256 + (((c + 0) ^ (xr & 15)) << 4));
                xa[t][kk][1] = *(const f32x4*)(st + xr * 256 + (((c + 1) ^ (xr & 15)) << 4));
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = (cw * 4 + j) * 16 + l16;
                const int off = rb * AL_ROWB + (((kk * 4 + g) ^ (rb & 7)) << 4);
                bh[kk][j] = *(const bf16x8*)(sh + off);
                bl[kk][j] = *(const bf16x8*)(sl + off);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // B: the stage may be refilled
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 ah[2], al[2];
            if constexpr (F16) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 p0, p1;
                    split_f16x8(xa[t][kk][0], xa[t][kk][1], p0, p1);
                    ah[kk] = __builtin_bit_cast(bf16x8, p0);
                    al[kk] = __builtin_bit_cast(bf16x8, p1);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bl[kk][j]), __builtin_bit_cast(f16x8, ah[kk]), acc[t][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[kk][j]), __builtin_bit_cast(f16x8, ah[kk]), acc[t][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f16x8 q2 = __builtin_bit_cast(f16x8, bh[kk][j]) * (_Float16)(1.0f / F16S_LO_SCALE);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q2, __builtin_bit_cast(f16x8, al[kk]), acc[t][j], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    ah[kk] = pack_bf16x8(xa[t][kk][0], xa[t][kk][1]);
                    f32x4 r0, r1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        r0[e] = xa[t][kk][0][e] - bf16_to_f32((bf16_t)ah[kk][e]);
                        r1[e] = xa[t][kk][1][e] - bf16_to_f32((bf16_t)ah[kk][4 + e]);
                    }
                    al[kk] = pack_bf16x8(r0, r1);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kk][j], ah[kk], acc[t][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kk][j], al[kk], acc[t][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[kk][j], ah[kk], acc[t][j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = row0 + rg * 32 + t * 16 + l16;
        if (m < M) {
            float* o = out + (size_t)m * 128 + cw * 64 + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(o + 16 * j) = F16 ? acc[t][j] * out_scale : acc[t][j];
        }
    }
}

}  // namespace

constexpr int SPLIT_MAX_B = 1024, SPLIT_MAX_G = 8;  // per-slot scratch of token_score_split_kernel
// Score arithmetic of the fast precision modes (token_score_kernel<.., FAST>): a process-wide switch like the GEMM hints, set by
// the caller together with its precision mode (madtp_amd/runtime.py).  -> previous value.
static std::atomic<int> g_score_fast{0};
static bool score_fast() { return g_score_fast.load(std::memory_order_relaxed) != 0; }
extern "C" int madtp_set_score_fast(int on) { return g_score_fast.exchange(on ? 1 : 0, std::memory_order_relaxed); }

static bool split_enabled() {  // MADTP_TS_SPLIT=0: always one workgroup per sample (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("MADTP_TS_SPLIT"); v = e ? atoi(e) : 1; }
    return v != 0;
}

static int token_score_launch(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                              const float* token_attn, int ldt, int ldb, int K, float temperature, float* score,
                              float* threshold, int32_t* count, int32_t* kmax, int B, int H, int N, int32_t* done_ctr,
                              int32_t* host_slot, int seq, void* stream, int32_t* tick = nullptr, float* part = nullptr) {
    if (!colsum_part || !p0 || !onorm || !token_attn || !score || !threshold || !count) return MADTP_E_BADARG;
    if (B <= 0 || H <= 0 || N < 2 || n_row_tiles <= 0 || !(temperature > 0.f)) return MADTP_E_BADARG;
    if (N - 1 > MAXN || K > 128 || K <= 0 || ldt < K) return MADTP_E_SHAPE;
    const size_t stage_bytes = (size_t)(N - 1) * K * sizeof(float);
    const bool staged = K % 4 == 0 && ldt % 4 == 0 && ldb % 4 == 0 && aligned16(token_attn) && stage_bytes <= 140 * 1024;
    if (staged) {
        if (score_fast()) {
            MADTP_ENSURE_MAX_LDS((token_score_kernel<true, true>), 140 * 1024);
            hipLaunchKernelGGL((token_score_kernel<true, true>), dim3(B), dim3(512), stage_bytes, (hipStream_t)stream, colsum_part,
                               n_row_tiles, p0, onorm, token_attn, ldt, ldb, K, temperature, score, threshold, count, kmax, H, N,
                               done_ctr, host_slot, seq);
        } else {
            MADTP_ENSURE_MAX_LDS(token_score_kernel<true>, 140 * 1024);
            hipLaunchKernelGGL(token_score_kernel<true>, dim3(B), dim3(512), stage_bytes, (hipStream_t)stream, colsum_part,
                               n_row_tiles, p0, onorm, token_attn, ldt, ldb, K, temperature, score, threshold, count, kmax, H, N,
                               done_ctr, host_slot, seq);
        }
    } else if (tick && part && !kmax && B <= SPLIT_MAX_B && B <= 64 && K % 4 == 0 && ldt % 4 == 0 && ldb % 4 == 0 &&
               aligned16(token_attn) && split_enabled()) {
        // long sequence, small batch: G workgroups per sample (column split), so that the launch covers the chip (measured: VQA,
        // 32 samples x 901 tokens, +6.5 % on the whole forward; at 128 samples the one-workgroup kernel already fills half the
        // chip and the repeated phase A costs 2 %)
#define TS_SPLIT_LAUNCH(G_, FAST_)                                                                                                          \
    do {                                                                                                                                   \
        MADTP_ENSURE_MAX_LDS((token_score_split_kernel<G_, FAST_>), (size_t)MAXN * (128 / G_) * sizeof(float));                             \
        hipLaunchKernelGGL((token_score_split_kernel<G_, FAST_>), dim3(8 * ((B + 7) / 8) * G_), dim3(512),                                  \
                           (size_t)(N - 1) * (128 / G_) * sizeof(float), (hipStream_t)stream, colsum_part, n_row_tiles, p0, onorm,        \
                           token_attn, ldt, ldb, K, temperature, score, threshold, count, H, N, done_ctr, host_slot, seq, tick, part, B); \
    } while (0)
        const bool fast = score_fast();  // (the fast precision modes' score arithmetic, madtp_set_score_fast)
        if (B <= 32) { if (fast) TS_SPLIT_LAUNCH(8, true); else TS_SPLIT_LAUNCH(8, false); }
        else { if (fast) TS_SPLIT_LAUNCH(4, true); else TS_SPLIT_LAUNCH(4, false); }
#undef TS_SPLIT_LAUNCH
    } else {
        hipLaunchKernelGGL(token_score_kernel<false>, dim3(B), dim3(512), 0, (hipStream_t)stream, colsum_part, n_row_tiles, p0,
                           onorm, token_attn, ldt, ldb, K, temperature, score, threshold, count, kmax, H, N, done_ctr, host_slot,
                           seq);
    }
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_token_score(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                                 const float* token_attn, int ldt, int ldb, int K, float temperature, float* score,
                                 float* threshold, int32_t* count, int32_t* kmax, int B, int H, int N, void* stream) {
    return token_score_launch(colsum_part, n_row_tiles, p0, onorm, token_attn, ldt, ldb, K, temperature, score, threshold, count,
                              kmax, B, H, N, nullptr, nullptr, 0, stream);
}

// Host-visible batch maximum.  Per DEVICE a small ring of hand-over slots, each = {pinned host pair (k, sequence number),
// device ticket counter}.  _publish claims a free slot of the current device under a short lock (never held across calls),
// launches token_score with it armed and returns the sequence number, which names the slot; _wait spins on that slot and
// frees it.  A caller that publishes and never waits leaks one slot (MADTP_E_BUSY once all are taken) instead of blocking
// every later call; slots of different devices / streams never alias.
namespace {
constexpr int SYNC_SLOTS = 16, SYNC_MAX_DEV = 64;
struct SyncSlot {
    int32_t* host = nullptr;      // pinned (portable, mapped): [0] = k, [1] = sequence number of the launch that wrote it
    int32_t* host_dev = nullptr;  // the same memory as this device sees it
    int32_t* ctr = nullptr;       // device ticket counter (the last workgroup resets it)
    int32_t* tick = nullptr;      // per-sample tickets of token_score_split_kernel [SPLIT_MAX_B] (self-resetting)
    float* part = nullptr;        // per-sample partial minima [SPLIT_MAX_B, SPLIT_MAX_G]
    bool busy = false;
    int seq = 0;
};
struct DevSync {
    std::mutex mu;
    bool ready = false;
    unsigned next_seq = 0;
    SyncSlot slot[SYNC_SLOTS];
};
DevSync g_dev_sync[SYNC_MAX_DEV];

int dev_sync_init(DevSync& d) {  // caller holds d.mu and has the device current
    if (d.ready) return 0;
    int32_t* host = nullptr;
    hipError_t e = hipHostMalloc((void**)&host, SYNC_SLOTS * 64, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return (int)e;
    memset(host, 0, SYNC_SLOTS * 64);
    int32_t* host_dev = nullptr;
    e = hipHostGetDevicePointer((void**)&host_dev, host, 0);
    if (e != hipSuccess) return (int)e;
    int32_t* ctr = nullptr;
    e = hipMalloc((void**)&ctr, SYNC_SLOTS * 64);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(ctr, 0, SYNC_SLOTS * 64);
    if (e != hipSuccess) return (int)e;
    const size_t per_slot = (size_t)SPLIT_MAX_B * (1 + SPLIT_MAX_G) * 4;
    char* scratch = nullptr;
    e = hipMalloc((void**)&scratch, SYNC_SLOTS * per_slot);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(scratch, 0, SYNC_SLOTS * per_slot);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < SYNC_SLOTS; ++i) {  // one 64-byte line per slot on both sides
        d.slot[i].host = host + 16 * i;
        d.slot[i].host_dev = host_dev + 16 * i;
        d.slot[i].ctr = ctr + 16 * i;
        d.slot[i].tick = (int32_t*)(scratch + i * per_slot);
        d.slot[i].part = (float*)(scratch + i * per_slot + (size_t)SPLIT_MAX_B * 4);
    }
    d.ready = true;
    return 0;
}
// sequence number handed to the caller: (device << 24) | (slot << 20) | 20-bit counter, never 0
inline int pack_seq(int dev, int slot, unsigned n) { return (dev << 24) | (slot << 20) | (int)(n & 0xFFFFF); }
}  // namespace

// Two-step form used by the layer-level calls: publish = launch with a host slot armed; wait = spin until the last workgroup
// has written k, then free the slot.  Work enqueued between the two calls runs on the GPU while the host waits - the layers
// put the projection GEMM there, so the read-back costs no GPU idle time.
extern "C" int madtp_token_score_publish(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                                         const float* token_attn, int ldt, int ldb, int K, float temperature, float* score,
                                         float* threshold, int32_t* count, int B, int H, int N, int* seq_out, void* stream) {
    if (!seq_out) return MADTP_E_BADARG;
    int dev = 0;
    hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) return (int)he;
    if (dev < 0 || dev >= SYNC_MAX_DEV) return MADTP_E_SHAPE;
    DevSync& d = g_dev_sync[dev];
    int si = -1, seq = 0;
    {
        std::lock_guard<std::mutex> lk(d.mu);
        const int rc = dev_sync_init(d);
        if (rc) return rc;
        for (int i = 0; i < SYNC_SLOTS && si < 0; ++i)
            if (!d.slot[i].busy) si = i;
        if (si < 0) return MADTP_E_BUSY;
        if ((++d.next_seq & 0xFFFFF) == 0) ++d.next_seq;
        seq = pack_seq(dev, si, d.next_seq);
        d.slot[si].busy = true;
        d.slot[si].seq = seq;
    }
    const int rc = token_score_launch(colsum_part, n_row_tiles, p0, onorm, token_attn, ldt, ldb, K, temperature, score, threshold,
                                      count, nullptr, B, H, N, d.slot[si].ctr, d.slot[si].host_dev, seq, stream, d.slot[si].tick,
                                      d.slot[si].part);
    if (rc) {
        std::lock_guard<std::mutex> lk(d.mu);
        d.slot[si].busy = false;
        return rc;
    }
    *seq_out = seq;
    return 0;
}

extern "C" int madtp_token_score_wait(int seq, const int32_t* count, int B, int32_t* k_host, void* stream) {
    const int dev = (seq >> 24) & 0x3F, si = (seq >> 20) & 0xF;
    if (!k_host || !count || seq == 0 || dev >= SYNC_MAX_DEV) return MADTP_E_BADARG;
    DevSync& d = g_dev_sync[dev];
    {
        std::lock_guard<std::mutex> lk(d.mu);
        if (!d.ready || !d.slot[si].busy || d.slot[si].seq != seq) return MADTP_E_BADARG;  // not a pending publish
    }
    SyncSlot& sl = d.slot[si];
    int rc = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; __atomic_load_n(&sl.host[1], __ATOMIC_ACQUIRE) != seq; ++spins) {
        if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
            // never signalled (failed launch / lost stream): fall back to the ordinary path so the error surfaces.  The ticket
            // counter is reset only AFTER the stream has drained, so a late kernel cannot race the memset.
            hipError_t e = hipStreamSynchronize((hipStream_t)stream);
            std::vector<int32_t> h(B);
            if (e == hipSuccess) e = hipMemcpy(h.data(), count, sizeof(int32_t) * B, hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                int m = 0;
                for (int v : h) m = v > m ? v : m;
                *k_host = m;
                if (__atomic_load_n(&sl.host[1], __ATOMIC_ACQUIRE) != seq) {
                    e = hipMemset(sl.ctr, 0, sizeof(int32_t));
                    if (e == hipSuccess) e = hipMemset(sl.tick, 0, sizeof(int32_t) * SPLIT_MAX_B);
                }
            }
            rc = (int)e;
            std::lock_guard<std::mutex> lk(d.mu);
            sl.busy = false;
            return rc;
        }
    }
    *k_host = __atomic_load_n(&sl.host[0], __ATOMIC_RELAXED);
    {
        std::lock_guard<std::mutex> lk(d.mu);
        sl.busy = false;
    }
    // the f16 range flag (madtp_range_status): kernels of this layer that ran before token_score have long completed, so a value
    // that left the f16 range in an f16 precision mode surfaces here, at the layer's one host synchronisation, as an error code
    // instead of NaNs further down (sticky until madtp_range_status(reset = 1))
    const int* rf = madtp_internal_range_flag();
    return (rf && *(const volatile int*)rf) ? MADTP_E_RANGE : 0;
}

extern "C" int madtp_token_score_sync(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                                      const float* token_attn, int ldt, int ldb, int K, float temperature, float* score,
                                      float* threshold, int32_t* count, int32_t* k_host, int B, int H, int N, void* stream) {
    if (!k_host) return MADTP_E_BADARG;
    int seq = 0;
    TRY(madtp_token_score_publish(colsum_part, n_row_tiles, p0, onorm, token_attn, ldt, ldb, K, temperature, score, threshold,
                                  count, B, H, N, &seq, stream));
    return madtp_token_score_wait(seq, count, B, k_host, stream);
}

#ifdef MADTP_TS_TIMING
extern "C" int madtp_debug_read_ts(long long* out) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ts_dbg), sizeof(long long) * 16);
}
#endif

// ---- sync-free encoder path (internal.h): the same kernels with their sizes read from the layer's device-side record ----
int madtp_i_token_score_dev(const float* colsum_part, const float* p0, const float* onorm, const float* logits, int ldt, int K,
                            float temperature, float* score, float* threshold, int32_t* count, int B, int H, int N_max,
                            int32_t* dims_l, int32_t* ticket, void* stream) {
    if (!colsum_part || !p0 || !onorm || !logits || !score || !threshold || !count || !dims_l || !ticket) return MADTP_E_BADARG;
    if (B <= 0 || H <= 0 || N_max < 2 || !(temperature > 0.f)) return MADTP_E_BADARG;
    if (N_max - 1 > MAXN || K > 128 || K <= 0 || ldt < K) return MADTP_E_SHAPE;
    const size_t stage_bytes = (size_t)(N_max - 1) * K * sizeof(float);
    const bool staged = K % 4 == 0 && ldt % 4 == 0 && aligned16(logits) && stage_bytes <= 140 * 1024;
    const float* ta = logits + ldt;  // row 0 of every sample is the CLS token
    if (staged) {  // (the same kernel variant as the host-side path takes at any n <= N_max - 1: same arithmetic)
        if (score_fast()) {
            MADTP_ENSURE_MAX_LDS((token_score_kernel<true, true>), 140 * 1024);
            hipLaunchKernelGGL((token_score_kernel<true, true>), dim3(B), dim3(512), stage_bytes, (hipStream_t)stream, colsum_part,
                               (N_max + 15) / 16, p0, onorm, ta, ldt, N_max * ldt, K, temperature, score, threshold, count,
                               (int32_t*)nullptr, H, N_max, ticket, (int32_t*)nullptr, 0, dims_l);
        } else {
            MADTP_ENSURE_MAX_LDS(token_score_kernel<true>, 140 * 1024);
            hipLaunchKernelGGL(token_score_kernel<true>, dim3(B), dim3(512), stage_bytes, (hipStream_t)stream, colsum_part,
                               (N_max + 15) / 16, p0, onorm, ta, ldt, N_max * ldt, K, temperature, score, threshold, count,
                               (int32_t*)nullptr, H, N_max, ticket, (int32_t*)nullptr, 0, dims_l);
        }
    } else {
        hipLaunchKernelGGL(token_score_kernel<false>, dim3(B), dim3(512), 0, (hipStream_t)stream, colsum_part, (N_max + 15) / 16, p0,
                           onorm, ta, ldt, N_max * ldt, K, temperature, score, threshold, count, (int32_t*)nullptr, H, N_max, ticket,
                           (int32_t*)nullptr, 0, dims_l);
    }
    MADTP_LAUNCH_CHECK();
    return 0;
}

int madtp_i_token_select_dev(const float* score, int64_t* indices, int64_t* indices_sort, int32_t* dst_pos, float* merge_w, int B,
                             int n_max, const int32_t* dims_l, void* stream) {
    if (!score || !indices || !indices_sort || !dst_pos || !merge_w || !dims_l || B <= 0 || n_max <= 0) return MADTP_E_BADARG;
    if (n_max > 320) return MADTP_E_SHAPE;  // (the 1024-thread variant of long sequences keeps host-side sizes)
    hipLaunchKernelGGL(token_select_kernel<256>, dim3(B), dim3(256), 0, (hipStream_t)stream, score, 0, indices, indices_sort, dst_pos,
                       merge_w, n_max, dims_l);
    MADTP_LAUNCH_CHECK();
    return 0;
}

int madtp_i_token_gather_ln_dev(const float* x, const int32_t* dst_pos, const float* merge_w, float* y, int B, int N_max, int dim,
                                const float* gamma, const float* beta, float eps, float* h32, void* h_lp, int lp_dtype,
                                const int32_t* dims_l, void* stream) {
    if (h_lp && lp_dtype != MADTP_BF16 && lp_dtype != MADTP_F16S && lp_dtype != MADTP_F16) return MADTP_E_DTYPE;
    if (!x || !dst_pos || !merge_w || !y || !dims_l || B <= 0 || N_max < 2) return MADTP_E_BADARG;
    if (gamma && (!beta || (!h32 && !h_lp))) return MADTP_E_BADARG;
    if (dim % 4 || dim > 1024) return MADTP_E_SHAPE;
    const int chunks = (N_max + GATHER_ROWS - 1) / GATHER_ROWS;
    hipLaunchKernelGGL(token_gather_kernel, dim3(chunks + 1, B), dim3(256), 0, (hipStream_t)stream, x, dst_pos, merge_w, y, N_max, 0,
                       dim / 4, gamma, beta, eps, h32, (bf16_t*)h_lp, lp_dtype, madtp_internal_range_flag(), dims_l);
    MADTP_LAUNCH_CHECK();
    return 0;
}

static bool rank_split_enabled() {  // MADTP_RANK_SPLIT=0: the one-workgroup ranking at every length (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("MADTP_RANK_SPLIT"); v = e ? atoi(e) : 1; }
    return v != 0;
}

extern "C" int madtp_token_select(const float* score, int k, int64_t* indices, int64_t* indices_sort, int32_t* dst_pos,
                                  float* merge_w, int B, int n, void* stream) {
    if (!score || !indices || !indices_sort || !dst_pos || !merge_w || B <= 0 || n <= 0) return MADTP_E_BADARG;
    if (k < 1 || k > n || n > MAXN) return MADTP_E_SHAPE;
    if (n > 320 && B <= 64 && rank_split_enabled()) {
        // long sequences at small batches: the ranking on ceil(n / 64) workgroups per sample, then compaction + merge weights
        hipLaunchKernelGGL(token_rank_kernel, dim3((n + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, score, indices_sort, dst_pos, n);
        hipLaunchKernelGGL(token_select_kernel<256>, dim3(B), dim3(256), 0, (hipStream_t)stream, score, k, indices, indices_sort,
                           dst_pos, merge_w, n, (const int32_t*)nullptr, true);
    } else if (n > 320)
        hipLaunchKernelGGL(token_select_kernel<1024>, dim3(B), dim3(1024), 0, (hipStream_t)stream, score, k, indices, indices_sort,
                           dst_pos, merge_w, n);
    else
        hipLaunchKernelGGL(token_select_kernel<256>, dim3(B), dim3(256), 0, (hipStream_t)stream, score, k, indices, indices_sort,
                           dst_pos, merge_w, n);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_token_gather_ln(const float* x, const int32_t* dst_pos, const float* merge_w, float* y, int B, int N,
                                     int k, int dim, const float* gamma, const float* beta, float eps, float* h32, void* h_lp,
                                     int lp_dtype, void* stream) {
    if (h_lp && lp_dtype != MADTP_BF16 && lp_dtype != MADTP_F16S && lp_dtype != MADTP_F16) return MADTP_E_DTYPE;
    if (!x || !dst_pos || !merge_w || !y || B <= 0 || N < 2 || k < 1 || k > N - 1) return MADTP_E_BADARG;
    if (gamma && (!beta || (!h32 && !h_lp))) return MADTP_E_BADARG;
    if (dim % 4 || dim > 1024) return MADTP_E_SHAPE;
    if (!aligned16(x) || !aligned16(y) || (gamma && (!aligned16(gamma) || !aligned16(beta) || !aligned16(h32) || !aligned16(h_lp))))
        return MADTP_E_ALIGN;
    const int chunks = (N + GATHER_ROWS - 1) / GATHER_ROWS;
    hipLaunchKernelGGL(token_gather_kernel, dim3(chunks + 1, B), dim3(256), 0, (hipStream_t)stream, x, dst_pos, merge_w, y, N,
                       k, dim / 4, gamma, beta, eps, h32, (bf16_t*)h_lp, lp_dtype, madtp_internal_range_flag());
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_token_gather(const float* x, const int32_t* dst_pos, const float* merge_w, float* y, int B, int N,
                                  int k, int dim, void* stream) {
    return madtp_token_gather_ln(x, dst_pos, merge_w, y, B, N, k, dim, nullptr, nullptr, 0.f, nullptr, nullptr, MADTP_BF16, stream);
}

extern "C" int madtp_mask_gather(const float* mask, const int64_t* order, int ld_order, const int64_t* order2,
                                 int ld_order2, float* out, int B, int N, int k, void* stream) {
    if (!mask || !order || !out || B <= 0 || N < 2 || k < 1) return MADTP_E_BADARG;
    if (order2 ? (k > ld_order || k + 1 > ld_order2) : (k + 1 > ld_order)) return MADTP_E_BADARG;
    hipLaunchKernelGGL(mask_gather_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mask, order, ld_order, order2, ld_order2,
                       out, N, k);
    MADTP_LAUNCH_CHECK();
    return 0;
}

int madtp_i_mask_gather_dev(const float* mask, const int64_t* indices, const int64_t* indices_sort, int variant_nlvr, float* out, int B,
                            const int32_t* dims_l, void* stream) {
    if (!mask || !indices || !indices_sort || !out || !dims_l || B <= 0) return MADTP_E_BADARG;
    if (variant_nlvr)  // nlvr_encoder.py:451-452: the first k+1 entries of the full sort order
        hipLaunchKernelGGL(mask_gather_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mask, indices_sort, 0, (const int64_t*)nullptr, 0,
                           out, 0, 0, dims_l);
    else               // med.py:377,388-390: the kept indices, then the (k+1)-th ranked token
        hipLaunchKernelGGL(mask_gather_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mask, indices, 0, indices_sort, 0, out, 0, 0, dims_l);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_query_att_ft(const float* token_attn, int ldt, int ldb, int K, const float* x, int ldf, int ldfb,
                                  float* out, float inv_sqrt_sd, int accumulate, int B, int n, int dim, int fast,
                                  float* stats_ws, void* stream) {
    if (!token_attn || !x || !out || B <= 0 || n < 1) return MADTP_E_BADARG;
    if (K <= 0 || K > 112 || dim % 256 || ldt < K) return MADTP_E_SHAPE;
    if (fast) {
        const madtp_att_ft_seg one = {token_attn, x, n, ldt, ldb, ldf, ldfb};
        return madtp_query_att_ft_multi(&one, 1, K, out, stats_ws, inv_sqrt_sd, accumulate, B, dim, stream);
    }
    hipLaunchKernelGGL(query_att_ft_kernel, dim3(dim / 256, B), dim3(256), 0, (hipStream_t)stream, token_attn, ldt, ldb, K, x,
                       ldf, ldfb, out, inv_sqrt_sd, accumulate, n, dim);
    MADTP_LAUNCH_CHECK();
    return 0;
}

static int att_ft_multi_impl(const madtp_att_ft_seg* segs, int nseg, int K, float* out, float* stats_ws, float inv_sqrt_sd,
                             int accumulate, int B, int dim, bool split, void* stream);

extern "C" int madtp_query_att_ft_multi(const madtp_att_ft_seg* segs, int nseg, int K, float* out, float* stats_ws,
                                        float inv_sqrt_sd, int accumulate, int B, int dim, void* stream) {
    return att_ft_multi_impl(segs, nseg, K, out, stats_ws, inv_sqrt_sd, accumulate, B, dim, false, stream);
}

extern "C" int madtp_query_att_ft_multi_split(const madtp_att_ft_seg* segs, int nseg, int K, float* out, float* stats_ws,
                                              float inv_sqrt_sd, int accumulate, int B, int dim, void* stream) {
    if (!stats_ws) return MADTP_E_BADARG;
    return att_ft_multi_impl(segs, nseg, K, out, stats_ws, inv_sqrt_sd, accumulate, B, dim, true, stream);
}

static int att_ft_multi_impl(const madtp_att_ft_seg* segs, int nseg, int K, float* out, float* stats_ws, float inv_sqrt_sd,
                             int accumulate, int B, int dim, bool split, void* stream) {
    if (!segs || !out || nseg < 1 || B <= 0) return MADTP_E_BADARG;
    if (K <= 0 || K > 112 || dim % AB_COLS) return MADTP_E_SHAPE;
    if (!stats_ws) {  // parity modes: exact-f32 arithmetic, the per-layer summation order kept (see the kernel)
        if (dim % 256) return MADTP_E_SHAPE;
        for (int first = 0; first < nseg; first += AB_MAXSEG) {
            AttFtSegs a;
            a.nseg = nseg - first < AB_MAXSEG ? nseg - first : AB_MAXSEG;
            for (int i = 0; i < a.nseg; ++i) {
                const madtp_att_ft_seg& g = segs[first + i];
                if (!g.token_attn || !g.ft || g.n < 1) return MADTP_E_BADARG;
                if (g.ldt_row < K) return MADTP_E_SHAPE;
                a.s[i] = AttFtSeg{g.token_attn, g.ft, g.n, g.ldt_row, g.ldt_batch, g.ldf_row, g.ldf_batch};
            }
            hipLaunchKernelGGL(query_att_ft_exact_multi_kernel, dim3(dim / 256, B), dim3(256), 0, (hipStream_t)stream, a, K, out,
                               inv_sqrt_sd, (accumulate || first > 0) ? 1 : 0, dim);
            MADTP_LAUNCH_CHECK();
        }
        return 0;
    }
    for (int first = 0; first < nseg; first += AB_MAXSEG) {
        AttFtSegs a;
        a.nseg = nseg - first < AB_MAXSEG ? nseg - first : AB_MAXSEG;
        for (int i = 0; i < a.nseg; ++i) {
            const madtp_att_ft_seg& g = segs[first + i];
            if (!g.token_attn || !g.ft || g.n < 1) return MADTP_E_BADARG;
            if (g.ldt_row < ((K + 3) & ~3)) return MADTP_E_SHAPE;
            if (g.ldf_row % 4 || g.ldf_batch % 4 || !aligned16(g.ft)) return MADTP_E_ALIGN;
            if (g.ldt_row % 4 || g.ldt_batch % 4 || !aligned16(g.token_attn)) return MADTP_E_ALIGN;
            a.s[i] = AttFtSeg{g.token_attn, g.ft, g.n, g.ldt_row, g.ldt_batch, g.ldf_row, g.ldf_batch};
        }
        float* stats = stats_ws + (size_t)first * B * 256;
        const int acc_flag = (accumulate || first > 0) ? 1 : 0;
        if (split) {
            MADTP_ENSURE_MAX_LDS(query_att_ft_mma_kernel<true>, att_ft_mma_lds(true));
            hipLaunchKernelGGL(att_ft_stats_kernel<true>, dim3(a.nseg, B), dim3(256), 0, (hipStream_t)stream, a, K, inv_sqrt_sd, stats);
            hipLaunchKernelGGL(query_att_ft_mma_kernel<true>, dim3(dim / AB_COLS, B), dim3(256), att_ft_mma_lds(true),
                               (hipStream_t)stream, a, K, stats, out, inv_sqrt_sd, acc_flag, dim);
        } else {
            MADTP_ENSURE_MAX_LDS(query_att_ft_mma_kernel<false>, att_ft_mma_lds(false));
            hipLaunchKernelGGL(att_ft_stats_kernel<false>, dim3(a.nseg, B), dim3(256), 0, (hipStream_t)stream, a, K, inv_sqrt_sd, stats);
            hipLaunchKernelGGL(query_att_ft_mma_kernel<false>, dim3(dim / AB_COLS, B), dim3(256), att_ft_mma_lds(false),
                               (hipStream_t)stream, a, K, stats, out, inv_sqrt_sd, acc_flag, dim);
        }
        MADTP_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int madtp_vector_gather(const float* vectors, const int64_t* indices, float* out, int B, int L, int K, int D,
                                   void* stream) {
    if (!vectors || !indices || !out || B <= 0 || L <= 0 || K <= 0 || D <= 0) return MADTP_E_BADARG;
    if (D % 4) return MADTP_E_SHAPE;
    if (!aligned16(vectors) || !aligned16(out)) return MADTP_E_ALIGN;
    const int rows = B * K;
    hipLaunchKernelGGL(vector_gather_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, vectors, indices, out, L,
                       K, D / 4, rows);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_align_logits(const float* x, const void* sd_hi, const void* sd_lo, float* out, int M, int dim,
                                  int split_dtype, float out_scale, void* stream) {
    return madtp_i_align_logits(x, sd_hi, sd_lo, out, M, dim, split_dtype, out_scale, DevN{nullptr, 0, 0}, stream);
}
// tile height of the wave-specialised alignment kernel when MADTP_ALIGN_ROWS does not force one (set from the measurement,
// profiles/r06_align_rows_ab.txt)
static int align_rows_auto(int M) { (void)M; return 64; }

int madtp_i_align_logits(const float* x, const void* sd_hi, const void* sd_lo, float* out, int M, int dim, int split_dtype,
                         float out_scale, DevN m_dev, void* stream) {
    if (!x || !sd_hi || !sd_lo || !out || M <= 0) return MADTP_E_BADARG;
    if (split_dtype != MADTP_BF16 && split_dtype != MADTP_F16S) return MADTP_E_DTYPE;
    if (dim % 128) return MADTP_E_SHAPE;
    if (!aligned16(x) || !aligned16(sd_hi) || !aligned16(sd_lo) || !aligned16(out)) return MADTP_E_ALIGN;
    constexpr int lds = AL_STAGES * 2 * AL_TILE;
    MADTP_ENSURE_MAX_LDS(align_logits_kernel<12>, lds);
    MADTP_ENSURE_MAX_LDS(align_logits_kernel<8>, lds);
    MADTP_ENSURE_MAX_LDS(align_logits_kernel<0>, lds);
    static int variant = -1;  // MADTP_ALIGN_KERNEL=1 selects the register-prefetching kernel (A/B measurements)
    if (variant < 0) { const char* e = getenv("MADTP_ALIGN_KERNEL"); variant = e ? atoi(e) : 0; }
    // MADTP_ALIGN_ROWS: 64 (align_ws_kernel), 128 (align_ws2_kernel), 0 = automatic (see align_rows_auto)
    static int rows_env = -1;
    if (rows_env < 0) { const char* e = getenv("MADTP_ALIGN_ROWS"); rows_env = e ? atoi(e) : 0; }
    const int rows_sel = rows_env == 64 || rows_env == 128 ? rows_env : align_rows_auto(M);
    if ((variant == 0 || split_dtype == MADTP_F16S) && rows_sel == 128 && !m_dev.p) {
        MADTP_ENSURE_MAX_LDS(align_ws2_kernel<false>, 2 * AW2_STAGE);
        MADTP_ENSURE_MAX_LDS(align_ws2_kernel<true>, 2 * AW2_STAGE);
        if (split_dtype == MADTP_F16S)
            hipLaunchKernelGGL(align_ws2_kernel<true>, dim3((M + 127) / 128), dim3(768), 2 * AW2_STAGE, (hipStream_t)stream, x,
                               (const char*)sd_hi, (const char*)sd_lo, out, M, dim, out_scale, m_dev);
        else
            hipLaunchKernelGGL(align_ws2_kernel<false>, dim3((M + 127) / 128), dim3(768), 2 * AW2_STAGE, (hipStream_t)stream, x,
                               (const char*)sd_hi, (const char*)sd_lo, out, M, dim, 1.f, m_dev);
        MADTP_LAUNCH_CHECK();
        return 0;
    }
    if (variant == 0 || split_dtype == MADTP_F16S) {
        MADTP_ENSURE_MAX_LDS(align_ws_kernel<false>, 3 * AW_STAGE);
        MADTP_ENSURE_MAX_LDS(align_ws_kernel<true>, 3 * AW_STAGE);
        if (split_dtype == MADTP_F16S)
            hipLaunchKernelGGL(align_ws_kernel<true>, dim3((M + 63) / 64), dim3(768), 3 * AW_STAGE, (hipStream_t)stream, x,
                               (const char*)sd_hi, (const char*)sd_lo, out, M, dim, out_scale, m_dev);
        else
            hipLaunchKernelGGL(align_ws_kernel<false>, dim3((M + 63) / 64), dim3(768), 3 * AW_STAGE, (hipStream_t)stream, x,
                               (const char*)sd_hi, (const char*)sd_lo, out, M, dim, 1.f, m_dev);
        MADTP_LAUNCH_CHECK();
        return 0;
    }
    auto kern = dim == 768 ? align_logits_kernel<12> : dim == 512 ? align_logits_kernel<8> : align_logits_kernel<0>;
    hipLaunchKernelGGL(kern, dim3((M + 63) / 64), dim3(256), lds, (hipStream_t)stream, x, (const char*)sd_hi,
                       (const char*)sd_lo, out, M, dim, m_dev);
    MADTP_LAUNCH_CHECK();
    return 0;
}
