// Answer ranking on the LM head's prediction scores (inference half):
//   madtp_lm_loss    - models/med.py BertLMHeadModel.forward :1036-1042: label-smoothed next-token cross-entropy, summed per sequence
//   madtp_token_prob - models/blip_vqa.py rank_answer :170-171: softmax probability of each candidate's first token
//   madtp_beam_topk  - the candidate selection of text_decoder.generate(num_beams=..) (models/blip_vqa.py:134, models/blip.py:189)
// Both are HBM-bound row kernels over the vocabulary (V = 30524 f32 scores = 119 KiB per row): one 256-thread workgroup per
// sequence / question, float4 loads, two passes per row (maximum + plain sum, then the exponential sum: the second pass reads
// the row back from L2), wave-shuffle + LDS reductions in a fixed order (deterministic).
// Algorithmic bytes: rows * V * 4 (scores read once from HBM), outputs negligible.
#include "common.h"

namespace {

constexpr int LM_THREADS = 256;

struct RowStats { float mx, sum_z, sum_e; };

// block-wide reductions through a 4-entry LDS scratch (one slot per wave); every thread gets the result
template <int NT = LM_THREADS>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = fmaxf(r, scratch[w]);
    return r;
}
template <int NT = LM_THREADS>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    if constexpr (NT == 256) return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);  // (the order lm_loss was pinned with)
    float r = scratch[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r += scratch[w];
    return r;
}

// max, sum of scores and sum of exp(score - max) of one row of V scores (row is 16-byte aligned, V need not be a multiple of 4)
template <int NT = LM_THREADS>
__device__ __forceinline__ RowStats row_stats(const float* row, int V, float* scratch) {
    const int V4 = V >> 2, tid = threadIdx.x;
    float mx = -INFINITY, sz = 0.f;
#pragma unroll 8
    for (int i = tid; i < V4; i += NT) {
        const float4 v = ((const float4*)row)[i];
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        sz += (v.x + v.y) + (v.z + v.w);
    }
    for (int i = (V4 << 2) + tid; i < V; i += NT) { mx = fmaxf(mx, row[i]); sz += row[i]; }
    mx = block_max<NT>(mx, scratch);
    sz = block_sum<NT>(sz, scratch);
    float se = 0.f;
#pragma unroll 8
    for (int i = tid; i < V4; i += NT) {
        const float4 v = ((const float4*)row)[i];
        se += (expf(v.x - mx) + expf(v.y - mx)) + (expf(v.z - mx) + expf(v.w - mx));
    }
    for (int i = (V4 << 2) + tid; i < V; i += NT) se += expf(row[i] - mx);
    se = block_sum<NT>(se, scratch);
    return {mx, sz, se};
}

__global__ __launch_bounds__(LM_THREADS) void lm_loss_kernel(const float* logits, int ld, int rows_per_seq, int n_pred, int V,
                                                             const int64_t* labels, int ld_labels, float eps, float* loss) {
    __shared__ float scratch[4];
    const int b = blockIdx.x;
    float total = 0.f;
    for (int t = 0; t < n_pred; ++t) {
        const long long y = labels[(size_t)b * ld_labels + t + 1];
        if (y < 0 || y >= V) continue;  // ignore_index (-100): the row contributes nothing (uniform across the workgroup)
        const float* row = logits + ((size_t)b * rows_per_seq + t) * ld;
        const RowStats st = row_stats(row, V, scratch);
        const float lse = st.mx + logf(st.sum_e);
        // CrossEntropyLoss(label_smoothing = eps): (1 - eps) * (-log p_y) + eps / V * sum_j (-log p_j)
        const float nll = lse - row[y];
        const float smooth = (float)V * lse - st.sum_z;
        total += (1.0f - eps) * nll + (eps / (float)V) * smooth;
    }
    if (threadIdx.x == 0) loss[b] = total;
}

__global__ __launch_bounds__(LM_THREADS) void token_prob_kernel(const float* logits, int ld, int V, const int64_t* tok, int A,
                                                                float* out) {
    __shared__ float scratch[4];
    const int q = blockIdx.x;
    const float* row = logits + (size_t)q * ld;
    const RowStats st = row_stats(row, V, scratch);
    for (int a = threadIdx.x; a < A; a += LM_THREADS) {
        const long long t = tok[a];
        out[(size_t)q * A + a] = (t >= 0 && t < V) ? expf(row[t] - st.mx) / st.sum_e : 0.f;
    }
}

// Beam-search candidate selection (transformers 4.15 generation_utils.py beam_search: log_softmax of the last-position scores,
// + the running beam score, view [B, num_beams * V], topk(2 * num_beams) sorted; models/blip_vqa.py:134, models/blip.py:189).
// One workgroup per batch item: log-sum-exp of its num_beams rows (row_stats), then every thread scans a strided share of the
// num_beams * V candidates keeping its n_top best in its own LDS slice, then n_top rounds of a block-wide arg-max (ties: the
// lower flat index) emit the winners in descending order.  HBM-bound: num_beams * V * 4 bytes per item, read twice from L2/HBM.
// 1024 threads per item: the launch has only B workgroups (32 for a caption batch), so the memory-level parallelism has to come
// from inside the workgroup - with 256 threads and one float4 in flight per thread the same kernel took 197 us per step.
constexpr int BT_MAX_TOP = 16, BT_MAX_BEAMS = 8, BT_THREADS = 1024;

__global__ __launch_bounds__(BT_THREADS) void beam_topk_kernel(const float* logits, int ld, int V, const float* beam_scores,
                                                               int num_beams, int n_top, int suppress, float* out_scores,
                                                               int32_t* out_index, const int64_t* prev_ids, int ld_prev,
                                                               int cur_len, float penalty) {
    __shared__ float scratch[BT_THREADS / 64];
    __shared__ float lse_s[BT_MAX_BEAMS];
    extern __shared__ __attribute__((aligned(16))) char bt_dyn[];  // [BT_THREADS][n_top] values, then [BT_THREADS][n_top] indices
    float* c_val = (float*)bt_dyn;
    int* c_idx = (int*)(c_val + BT_THREADS * n_top);
    __shared__ float w_val[BT_THREADS / 64];
    __shared__ int w_idx[BT_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    // RepetitionPenaltyLogitsProcessor (transformers 4.15, applied by beam_search to the LOG-PROBABILITIES): tokens already in a
    // beam's sequence (prompt included) get lp < 0 ? lp * penalty : lp / penalty.  One bit per (beam row, token) in LDS.
    unsigned* seen = (unsigned*)(c_idx + BT_THREADS * n_top);  // [num_beams][(V + 31) / 32] when prev_ids != NULL
    const int vw = (V + 31) / 32;
    if (prev_ids) {
        for (int i = tid; i < num_beams * vw; i += BT_THREADS) seen[i] = 0u;
        __syncthreads();
        for (int i = tid; i < num_beams * cur_len; i += BT_THREADS) {
            const int j = i / cur_len, c = i - j * cur_len;
            const long long t = prev_ids[((size_t)b * num_beams + j) * ld_prev + c];
            if (t >= 0 && t < V) atomicOr(&seen[j * vw + (int)(t >> 5)], 1u << (t & 31));
        }
        __syncthreads();
    }
    for (int j = 0; j < num_beams; ++j) {
        const RowStats st = row_stats<BT_THREADS>(logits + ((size_t)b * num_beams + j) * ld, V, scratch);
        if (tid == 0) lse_s[j] = st.mx + logf(st.sum_e);
    }
    __syncthreads();
    float* mv = c_val + tid * n_top;
    int* mi = c_idx + tid * n_top;
    for (int q = 0; q < n_top; ++q) { mv[q] = -INFINITY; mi[q] = 0x7fffffff; }
    float cur_min = -INFINITY;
    int min_pos = 0, filled = 0;
    for (int j = 0; j < num_beams; ++j) {
        const float* row = logits + ((size_t)b * num_beams + j) * ld;
        const float add = beam_scores[b * num_beams + j], lse = lse_s[j];
        // a thread visits its elements in ascending flat index (ties keep the lower index): four consecutive tokens per 16-byte load
        // (the launcher requires 16-byte aligned rows), two loads in flight; V % 4 tokens at the end one by one
        auto take = [&](float x, int t) {
            float lp = x - lse;
            if (prev_ids && ((seen[j * vw + (t >> 5)] >> (t & 31)) & 1u)) lp = lp < 0.f ? lp * penalty : lp / penalty;
            float v = lp + add;
            if (t == suppress) v = -INFINITY;
            if (!(v > -INFINITY)) return;  // -inf and NaN never become candidates
            if (filled < n_top) {
                mv[filled] = v; mi[filled] = j * V + t; ++filled;
                if (filled == n_top) {
                    cur_min = mv[0]; min_pos = 0;
                    for (int q = 1; q < n_top; ++q) if (mv[q] < cur_min) { cur_min = mv[q]; min_pos = q; }
                }
            } else if (v > cur_min) {
                mv[min_pos] = v; mi[min_pos] = j * V + t;
                cur_min = mv[0]; min_pos = 0;
                for (int q = 1; q < n_top; ++q) if (mv[q] < cur_min) { cur_min = mv[q]; min_pos = q; }
            }
        };
        const int V4 = V >> 2;
        int t4 = tid;
        for (; t4 + BT_THREADS < V4; t4 += 2 * BT_THREADS) {
            const float4 x0 = ((const float4*)row)[t4], x1 = ((const float4*)row)[t4 + BT_THREADS];
            take(x0.x, 4 * t4); take(x0.y, 4 * t4 + 1); take(x0.z, 4 * t4 + 2); take(x0.w, 4 * t4 + 3);
            const int u4 = t4 + BT_THREADS;
            take(x1.x, 4 * u4); take(x1.y, 4 * u4 + 1); take(x1.z, 4 * u4 + 2); take(x1.w, 4 * u4 + 3);
        }
        for (; t4 < V4; t4 += BT_THREADS) {
            const float4 x0 = ((const float4*)row)[t4];
            take(x0.x, 4 * t4); take(x0.y, 4 * t4 + 1); take(x0.z, 4 * t4 + 2); take(x0.w, 4 * t4 + 3);
        }
        for (int t = 4 * V4 + tid; t < V; t += BT_THREADS) take(row[t], t);
    }
    for (int r = 0; r < n_top; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int q = 0; q < n_top; ++q)
            if (mv[q] > bv || (mv[q] == bv && mi[q] < bi)) { bv = mv[q]; bi = mi[q]; }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { w_val[tid >> 6] = bv; w_idx[tid >> 6] = bi; }
        __syncthreads();
        bv = w_val[0]; bi = w_idx[0];
        for (int w = 1; w < BT_THREADS / 64; ++w)
            if (w_val[w] > bv || (w_val[w] == bv && w_idx[w] < bi)) { bv = w_val[w]; bi = w_idx[w]; }
        if (tid == 0) {
            out_scores[(size_t)b * n_top + r] = bv;
            out_index[(size_t)b * n_top + r] = bi == 0x7fffffff ? -1 : bi;
        }
        for (int q = 0; q < n_top; ++q)
            if (mi[q] == bi && bi != 0x7fffffff) { mv[q] = -INFINITY; mi[q] = 0x7fffffff; }
    }
}


// ---- nucleus sampling step (models/blip.py:175-186: text_decoder.generate(do_sample=True, top_p=0.9, repetition_penalty=1.1)) ----
// transformers 4.15 `sample`: the processors run on the raw last-position scores (RepetitionPenaltyLogitsProcessor: a token that
// already occurs in the row gets s < 0 ? s * penalty : s / penalty; MinLengthLogitsProcessor: EOS = -inf below min_length), then the
// warpers - TopKLogitsWarper(top_k = config.top_k = 50) keeps the 50 best scores, TopPLogitsWarper(top_p) sorts them descending,
// softmaxes, and drops every token whose PREDECESSORS already hold more than top_p of the mass (so the first token that crosses
// top_p stays) - then softmax over what is left and one multinomial draw.  Here: one 1024-thread workgroup per row, the row in LDS
// (V <= 36 k); the k-th largest score by bisection over the order-preserving integer image of the floats (32 counting passes over
// LDS), the survivors collected, ranked by (score descending, index ascending) and walked by one wave: softmax, top-p prefix,
// and the draw as the inverse CDF of that prefix at u[row] (the caller's uniform number: the library has no generator state).
constexpr int SP_THREADS = 1024, SP_MAX_K = 64, SP_MAX_V = 36864;
__device__ __forceinline__ unsigned f32_order_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(SP_THREADS) void sample_top_kp_kernel(const float* logits, int ld, int V, const int64_t* prev_ids, int ld_prev,
                                                                  int cur_len, float penalty, int suppress, int top_k, float top_p,
                                                                  const float* u, int64_t* out_token, float* out_prob) {
    extern __shared__ __attribute__((aligned(16))) char sp_dyn[];
    float* row_s = (float*)sp_dyn;  // [V]
    __shared__ int cnt_s[SP_THREADS / 64];
    __shared__ int n_sel;
    __shared__ float sel_v[2 * SP_MAX_K];
    __shared__ int sel_i[2 * SP_MAX_K];
    __shared__ float ord_v[2 * SP_MAX_K];
    __shared__ int ord_i[2 * SP_MAX_K];
    __shared__ int tie_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* row = logits + (size_t)b * ld;
    for (int t = tid; t < V; t += SP_THREADS) row_s[t] = row[t];
    __syncthreads();
    if (prev_ids && penalty != 1.0f) {  // (a token may occur several times: the penalty is applied once, as the gather / scatter of the library does)
        for (int c = tid; c < cur_len; c += SP_THREADS) {
            const long long t = prev_ids[(size_t)b * ld_prev + c];
            bool first = t >= 0 && t < V;
            for (int c2 = 0; first && c2 < c; ++c2) first = prev_ids[(size_t)b * ld_prev + c2] != t;
            if (first) { const float s0 = row[t]; row_s[t] = s0 < 0.f ? s0 * penalty : s0 / penalty; }
        }
    }
    if (tid == 0 && suppress >= 0 && suppress < V) row_s[suppress] = -INFINITY;
    __syncthreads();
    // k-th largest key: the largest key K with count(key >= K) >= k
    const int k = min(top_k, V);
    unsigned lo = 0u, hi = 0xFFFFFFFFu;
    while (lo < hi) {
        const unsigned mid = lo + (unsigned)(((unsigned long long)hi - lo + 1ull) >> 1);
        int c = 0;
        for (int t = tid; t < V; t += SP_THREADS) c += f32_order_key(row_s[t]) >= mid ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        __syncthreads();
        if (lane == 0) cnt_s[tid >> 6] = c;
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < SP_THREADS / 64; ++w) tot += cnt_s[w];
        if (tot >= k) lo = mid; else hi = mid - 1u;
    }
    if (tid == 0) n_sel = 0;
    __syncthreads();
    // Survivors = every score >= the k-th largest: TopKLogitsWarper removes `scores < kth`, so ties AT the k-th score all stay.
    // (1) the strictly greater ones - fewer than k, any arrival order (they are ranked below); (2) the ties in ascending index order,
    // as many as the 2 * SP_MAX_K slots hold: chunks of SP_THREADS consecutive indices, a workgroup-wide exclusive count per chunk -
    // the same survivors whatever the scheduling (arrival order used to decide, and ties were cut to exactly k).
    for (int t = tid; t < V; t += SP_THREADS) {
        const float v = row_s[t];
        if (f32_order_key(v) > lo && v > -INFINITY) {
            const int p = atomicAdd(&n_sel, 1);
            if (p < 2 * SP_MAX_K) { sel_v[p] = v; sel_i[p] = t; }
        }
    }
    __syncthreads();
    if (tid == 0) tie_base = min(n_sel, 2 * SP_MAX_K);
    __syncthreads();
    for (int base = 0; base < V; base += SP_THREADS) {
        const int t = base + tid;
        const float v = t < V ? row_s[t] : 0.f;
        const bool tie = t < V && f32_order_key(v) == lo && v > -INFINITY;
        const unsigned long long bal = __ballot(tie);
        if (lane == 0) cnt_s[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < (tid >> 6); ++w) before += cnt_s[w];
        int chunk = 0;
        for (int w = 0; w < SP_THREADS / 64; ++w) chunk += cnt_s[w];
        const int p = tie_base + before;
        if (tie && p < 2 * SP_MAX_K) { sel_v[p] = v; sel_i[p] = t; }
        __syncthreads();
        if (tid == 0) tie_base = min(tie_base + chunk, 2 * SP_MAX_K);
        __syncthreads();
        if (tie_base >= 2 * SP_MAX_K) break;  // (uniform: every thread reads the same shared value)
    }
    const int ns = tie_base;
    // rank by (score descending, index ascending)
    for (int p = tid; p < ns; p += SP_THREADS) {
        int r = 0;
        for (int q = 0; q < ns; ++q) r += (sel_v[q] > sel_v[p] || (sel_v[q] == sel_v[p] && sel_i[q] < sel_i[p])) ? 1 : 0;
        ord_v[r] = sel_v[p]; ord_i[r] = sel_i[p];
    }
    __syncthreads();
    if (tid == 0) {
        const int n = ns;
        if (n <= 0) { out_token[b] = suppress == 0 ? 1 : 0; if (out_prob) out_prob[b] = 0.f; return; }
        const float mx = ord_v[0];
        float z = 0.f;
        for (int r = 0; r < n; ++r) z += expf(ord_v[r] - mx);
        // top-p prefix: token r stays iff the mass of tokens 0..r-1 is <= top_p (the first one always stays)
        int m = 0;
        float cum = 0.f, kept = 0.f;
        for (int r = 0; r < n; ++r) {
            if (r > 0 && cum > top_p) break;
            const float pr = expf(ord_v[r] - mx) / z;
            cum += pr; kept += pr; m = r + 1;
        }
        // inverse CDF of the re-normalised prefix at u
        const float target = u[b] * kept;
        float acc = 0.f;
        int pick = m - 1;
        for (int r = 0; r < m; ++r) {
            acc += expf(ord_v[r] - mx) / z;
            if (target < acc) { pick = r; break; }
        }
        out_token[b] = ord_i[pick];
        if (out_prob) out_prob[b] = expf(ord_v[pick] - mx) / z / kept;
    }
}

}  // namespace

static int beam_topk_launch(const float* logits, int ld, int V, const float* beam_scores, int num_beams, int n_top,
                            int suppress_token, float* out_scores, int32_t* out_index, int B, const int64_t* prev_ids, int ld_prev,
                            int cur_len, float penalty, void* stream) {
    if (!logits || !beam_scores || !out_scores || !out_index || B <= 0 || V <= 0 || ld < V) return MADTP_E_BADARG;
    if (num_beams < 1 || num_beams > BT_MAX_BEAMS || n_top < 1 || n_top > BT_MAX_TOP || (long long)num_beams * V > 0x7ffffffeLL)
        return MADTP_E_SHAPE;
    if (!aligned16(logits) || (ld & 3)) return MADTP_E_ALIGN;
    if (prev_ids && (cur_len < 1 || ld_prev < cur_len || !(penalty > 0.f))) return MADTP_E_BADARG;
    const size_t lds = (size_t)BT_THREADS * n_top * 8 + (prev_ids ? (size_t)num_beams * ((V + 31) / 32) * 4 : 0);
    if (lds > 150 * 1024) return MADTP_E_SHAPE;  // (the kernel's static LDS comes on top)
    MADTP_ENSURE_MAX_LDS(beam_topk_kernel, 150 * 1024);
    hipLaunchKernelGGL(beam_topk_kernel, dim3(B), dim3(BT_THREADS), lds, (hipStream_t)stream, logits, ld, V, beam_scores, num_beams,
                       n_top, suppress_token, out_scores, out_index, prev_ids, ld_prev, cur_len, penalty);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_beam_topk(const float* logits, int ld, int V, const float* beam_scores, int num_beams, int n_top,
                               int suppress_token, float* out_scores, int32_t* out_index, int B, void* stream) {
    return beam_topk_launch(logits, ld, V, beam_scores, num_beams, n_top, suppress_token, out_scores, out_index, B, nullptr, 0, 0, 1.f,
                            stream);
}

extern "C" int madtp_beam_topk_penalty(const float* logits, int ld, int V, const float* beam_scores, int num_beams, int n_top,
                                       int suppress_token, const int64_t* prev_ids, int ld_prev, int cur_len, float repetition_penalty,
                                       float* out_scores, int32_t* out_index, int B, void* stream) {
    if (!prev_ids) return MADTP_E_BADARG;
    return beam_topk_launch(logits, ld, V, beam_scores, num_beams, n_top, suppress_token, out_scores, out_index, B, prev_ids, ld_prev,
                            cur_len, repetition_penalty, stream);
}

// ---- the hypothesis book-keeping of one beam-search step on the device (round 5) ----------------------------------------------
// transformers 4.15 generation_beam_search.py BeamSearchScorer.process / BeamHypotheses.add / is_done, restated per batch item: the
// 2 * num_beams candidates in rank order - an EOS candidate of rank < num_beams closes a hypothesis (score = sum_logprobs /
// cur_len ** length_penalty in double, kept while among the num_beams best), the others continue as next step's beams until
// num_beams are found.  With it the search has no host round trip per step: candidates, sequences, scores and the beams' source
// rows stay in device memory, the host reads the hypotheses once at the end (madtp_amd/generation.py).
namespace {
constexpr int BU_MAXNB = 8;
struct BeamUpdateArgs {
    const float* sc; const int32_t* ix; int n_top, V;
    const int64_t* ids_in; int64_t* ids_out; int ld_ids, cur_len;
    float* beam_scores; int64_t* beam_src;
    int32_t* hyp_n; int32_t* hyp_order; double* hyp_score; int32_t* hyp_len; int64_t* hyp_tok; double* worst; int32_t* done; int32_t* err;
    double denom; int nb, eos, pad, early_stopping;
};
__global__ __launch_bounds__(64) void beam_update_kernel(BeamUpdateArgs a) {
    // every lane runs the same (uniform) scalar logic; the lanes share the copies of token rows
    const int b = blockIdx.x, lane = threadIdx.x, nb = a.nb, S = nb + 1;
    if (a.done[b]) {  // a finished item: pad tokens, zero scores, source row 0 (as the library's zero-initialised next_beam_* do)
        for (int j = 0; j < nb; ++j) {
            const size_t ro = (size_t)(b * nb + j) * a.ld_ids;
            for (int t = lane; t < a.cur_len; t += 64) a.ids_out[ro + t] = a.ids_in[t];
            if (lane == 0) { a.ids_out[ro + a.cur_len] = a.pad; a.beam_scores[b * nb + j] = 0.f; a.beam_src[b * nb + j] = 0; }
        }
        return;
    }
    int n = a.hyp_n[b];
    double w = a.worst[b];
    int order[BU_MAXNB + 1];
    double score[BU_MAXNB + 1];
#pragma unroll
    for (int i = 0; i <= BU_MAXNB; ++i) { order[i] = i < S ? a.hyp_order[b * S + i] : 0; score[i] = i < S ? a.hyp_score[b * S + i] : 0.0; }
    float nb_s[BU_MAXNB];
    int nb_t[BU_MAXNB], nb_r[BU_MAXNB];
    int slot = 0;
    float best = -INFINITY;
    for (int r = 0; r < a.n_top; ++r) best = fmaxf(best, a.sc[b * a.n_top + r]);
    for (int r = 0; r < a.n_top && slot < nb; ++r) {
        const int idx = a.ix[b * a.n_top + r];
        if (idx < 0) continue;
        const int beam = idx / a.V, tok = idx - beam * a.V, row = b * nb + beam;
        const float s = a.sc[b * a.n_top + r];
        if (tok == a.eos) {
            if (r >= nb) continue;
            const double hs = (double)s / a.denom;
            if (n < nb || hs > w) {
                unsigned used = 0;  // physical slots in use
                for (int i = 0; i < n; ++i) used |= 1u << order[i];
                int p = 0;
                while (used & (1u << p)) ++p;
                for (int t = lane; t < a.cur_len; t += 64) a.hyp_tok[((size_t)b * S + p) * a.ld_ids + t] = a.ids_in[(size_t)row * a.ld_ids + t];
                if (lane == 0) a.hyp_len[b * S + p] = a.cur_len;
                score[p] = hs;
                order[n++] = p;
                if (n > nb) {  // drop the lowest (first of equals), the new worst is the lowest of the rest
                    int lo = 0;
                    for (int i = 1; i < n; ++i) if (score[order[i]] < score[order[lo]]) lo = i;
                    for (int i = lo; i + 1 < n; ++i) order[i] = order[i + 1];
                    --n;
                    w = score[order[0]];
                    for (int i = 1; i < n; ++i) w = fmin(w, score[order[i]]);
                } else {
                    w = fmin(hs, w);
                }
            }
        } else {
            nb_s[slot] = s; nb_t[slot] = tok; nb_r[slot] = row;
            ++slot;
        }
    }
    if (slot < nb) {  // fewer than num_beams open continuations among the candidates (the library asserts the same)
        if (lane == 0) *a.err = 1;
        for (; slot < nb; ++slot) { nb_s[slot] = 0.f; nb_t[slot] = a.pad; nb_r[slot] = b * nb; }
    }
    const bool dn = n >= nb && (a.early_stopping || w >= (double)best / a.denom);
    for (int j = 0; j < nb; ++j) {
        const size_t ro = (size_t)(b * nb + j) * a.ld_ids, ri = (size_t)nb_r[j] * a.ld_ids;
        for (int t = lane; t < a.cur_len; t += 64) a.ids_out[ro + t] = a.ids_in[ri + t];
        if (lane == 0) { a.ids_out[ro + a.cur_len] = nb_t[j]; a.beam_scores[b * nb + j] = nb_s[j]; a.beam_src[b * nb + j] = nb_r[j]; }
    }
    if (lane == 0) {
        a.hyp_n[b] = n; a.worst[b] = w; a.done[b] = dn ? 1 : 0;
        for (int i = 0; i < S; ++i) { a.hyp_order[b * S + i] = order[i]; a.hyp_score[b * S + i] = score[i]; }
    }
}
}  // namespace

extern "C" int madtp_beam_update(const float* cand_scores, const int32_t* cand_index, int n_top, int V, const int64_t* ids_in,
                                 int64_t* ids_out, int ld_ids, int cur_len, float* beam_scores, int64_t* beam_src, int32_t* hyp_n,
                                 int32_t* hyp_order, double* hyp_score, int32_t* hyp_len, int64_t* hyp_tok, double* worst, int32_t* done,
                                 int32_t* err, double denom, int num_beams, int eos_token, int pad_token, int early_stopping, int B,
                                 void* stream) {
    if (!cand_scores || !cand_index || !ids_in || !ids_out || !beam_scores || !beam_src || !hyp_n || !hyp_order || !hyp_score || !hyp_len ||
        !hyp_tok || !worst || !done || !err || B <= 0 || V <= 0)
        return MADTP_E_BADARG;
    if (num_beams < 1 || num_beams > BU_MAXNB || n_top < num_beams || n_top > 2 * BU_MAXNB || cur_len < 1 || cur_len >= ld_ids || !(denom > 0.0))
        return MADTP_E_SHAPE;
    BeamUpdateArgs a{cand_scores, cand_index, n_top, V, ids_in, ids_out, ld_ids, cur_len, beam_scores, beam_src, hyp_n, hyp_order, hyp_score,
                     hyp_len, hyp_tok, worst, done, err, denom, num_beams, eos_token, pad_token, early_stopping};
    hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_sample_top_p(const float* logits, int ld, int V, const int64_t* prev_ids, int ld_prev, int cur_len,
                                  float repetition_penalty, int suppress_token, int top_k, float top_p, const float* u, int64_t* out_token,
                                  float* out_prob, int B, void* stream) {
    if (!logits || !u || !out_token || B <= 0 || V <= 0 || ld < V) return MADTP_E_BADARG;
    if (V > SP_MAX_V || top_k < 1 || top_k > SP_MAX_K || !(top_p > 0.f) || !(repetition_penalty > 0.f)) return MADTP_E_SHAPE;
    if (prev_ids && (cur_len < 1 || ld_prev < cur_len)) return MADTP_E_BADARG;
    const size_t lds = (size_t)V * sizeof(float);
    MADTP_ENSURE_MAX_LDS(sample_top_kp_kernel, 150 * 1024);
    hipLaunchKernelGGL(sample_top_kp_kernel, dim3(B), dim3(SP_THREADS), lds, (hipStream_t)stream, logits, ld, V, prev_ids, ld_prev, cur_len,
                       repetition_penalty, suppress_token, top_k, top_p, u, out_token, out_prob);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_lm_loss(const float* logits, int ld, int rows_per_seq, int n_pred, int V, const int64_t* labels,
                             int ld_labels, float label_smoothing, float* loss, int B, void* stream) {
    if (!logits || !labels || !loss || B <= 0 || V <= 0 || n_pred < 0 || n_pred > rows_per_seq || ld < V ||
        ld_labels < n_pred + 1)
        return MADTP_E_BADARG;
    if (!aligned16(logits) || (ld & 3)) return MADTP_E_ALIGN;
    hipLaunchKernelGGL(lm_loss_kernel, dim3(B), dim3(LM_THREADS), 0, (hipStream_t)stream, logits, ld, rows_per_seq, n_pred, V,
                       labels, ld_labels, label_smoothing, loss);
    MADTP_LAUNCH_CHECK();
    return 0;
}

extern "C" int madtp_token_prob(const float* logits, int ld, int V, const int64_t* tok, int A, float* out, int Q, void* stream) {
    if (!logits || !tok || !out || Q <= 0 || A <= 0 || V <= 0 || ld < V) return MADTP_E_BADARG;
    if (!aligned16(logits) || (ld & 3)) return MADTP_E_ALIGN;
    hipLaunchKernelGGL(token_prob_kernel, dim3(Q), dim3(LM_THREADS), 0, (hipStream_t)stream, logits, ld, V, tok, A, out);
    MADTP_LAUNCH_CHECK();
    return 0;
}
