// Layer-level entry points: one C call enqueues the whole kernel sequence of half a transformer layer, so the host
// pays one FFI crossing instead of ~15-40 (the text encoder at B=64 x 20 tokens is launch-bound: its kernels run
// for 3-15 us each, less than a Python-side launch costs).  Pure host code; every op is one of the kernels of this
// library; nothing allocates - scratch comes from a caller-provided workspace whose size the *_workspace() queries
// return.  The only host<->device round trip of a layer stays where the reference has it: the read-back of
// k = max_b count between the two halves (vit.py:145 `.item()`).
#include "common.h"
#include "internal.h"
#include <stdlib.h>
#include <map>
#include <mutex>

namespace {

struct Carver {
    char* base; size_t off; size_t cap;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
    bool fits() const { return !base || off <= cap; }
};

// bytes per logical element of an activation in compute dtype dt (F16S: two f16 planes = 4 bytes, like f32)
inline size_t esz_of(int dt) { return (dt == MADTP_BF16 || dt == MADTP_F16) ? 2 : 4; }
// physical leading dimension of an operand with logical row length ld (f16-split rows hold 2 planes)
inline int pld(int dt, int ld) { return dt == MADTP_F16S ? 2 * ld : ld; }
// dtype the attention kernels run in: the f16-split mode keeps attention on the exact-f32 kernels (q/k/v/out f32)
inline int attn_dt(int dt) { return dt == MADTP_F16S ? MADTP_F32 : dt; }
// io_dtype of the madtp_attention* calls: the f16x3 mode keeps f32 STORAGE (attn_dt) but asks for the f16-split products
inline int attn_io(int dt) { return dt == MADTP_F16S ? MADTP_F16S : dt; }
// 128-byte K slabs a GEMM with operand dtype dt walks
inline int slabs_of(int K, int dt) { return dt == MADTP_F32 ? K / 32 : K / 64; }  // (f16-split: k-slabs; 2 staged steps each)

#define TRY(call)                  \
    do {                           \
        int rc__ = (call);         \
        if (rc__ != 0) return rc__; \
    } while (0)

// lda / ldc: LOGICAL row lengths (elements of the f32 matrix); split operands get their physical stride here
inline int lin(const void* a, int lda, const madtp_lin& L, const float* residual, int ldr, void* c, int ldc, int M,
               int dt, int c_dt, int act, float scale, void* stream) {
    return madtp_gemm(a, L.w, L.b, residual, c, M, L.n, L.k, pld(dt, lda), (dt == MADTP_F16S ? 2 : 1) * L.k, pld(c_dt, ldc), ldr,
                      dt, c_dt, act, L.w_scale, scale, stream);
}

// LayerNorm into the compute dtype (and optionally an f32 copy)
inline int ln_to(const float* x, const float* g, const float* b, float* y32, void* yc, int rows, int dim, float eps, int dt,
                 void* stream) {
    if (dt == MADTP_F32) return madtp_layernorm(x, g, b, (float*)yc, nullptr, MADTP_BF16, rows, dim, eps, stream);  // yc doubles as y32
    return madtp_layernorm(x, g, b, y32, yc, dt, rows, dim, eps, stream);
}

// f32 [rows, dim] (contiguous rows of ld_src) -> the compute dtype's GEMM operand format (bf16 cast or f16 split)
inline int to_lp(const float* src, int ld_src, void* dst, int rows, int dim, int dt, void* stream) {
    if (dt == MADTP_F16S) return madtp_split_f16(src, ld_src, dst, 2 * dim, rows, dim, stream);
    if (ld_src != dim) return MADTP_E_SHAPE;
    return madtp_cast_lp(src, dst, (size_t)rows * dim, dt, 1.0f, stream);
}

// split-K factor for a small-M projection that feeds a LayerNorm: only when the tile count leaves most CUs idle and the
// K loop is long enough that cutting it beats the extra partial traffic (S*M*N*4 bytes written and re-read)
inline int choose_splits(int M, int N, int K, int dt) {
    const int nk = slabs_of(K, dt);
    if (dt != MADTP_F32 && M < 4096) {
        // small bf16 problem: madtp_gemm runs it on 64x64 tiles, 3 workgroups per CU; split K while the grid still fits
        // one round (768 workgroups) and every split keeps >= 12 slabs
        const int t64 = ((M + 63) / 64) * ((N + 63) / 64);
        int best = 1;
        const int steps = dt == MADTP_F16S ? 2 * nk : nk;  // staged slab steps of the K loop
        for (int sp = 2; sp <= 4; ++sp)
            if (nk % sp == 0 && steps / sp >= 12 && t64 * sp <= 768) best = sp;
        return best;
    }
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    if (tiles > 128 || nk < 24) return 1;
    int best = 1;
    for (int sp = 2; sp <= 4; ++sp)
        if (nk % sp == 0 && nk / sp >= 6 && tiles * sp <= 512) best = sp;
    return best;
}

// y = LayerNorm(scale*(a @ W^T + b) + residual) -> y32 (f32) and/or ylp (bf16); `part` is scratch of >= 4*M*N floats
inline int lin_ln(const void* a, int lda, const madtp_lin& L, const float* residual, float scale, const float* gamma,
                  const float* beta, float* y32, void* ylp, int M, int dt, float eps, float* part, void* stream) {
    const int S = choose_splits(M, L.n, L.k, dt);
    const int ldw = (dt == MADTP_F16S ? 2 : 1) * L.k;
    if (S > 1) {
        TRY(madtp_gemm_splitk(a, L.w, part, M, L.n, L.k, pld(dt, lda), ldw, S, dt, stream));
        return madtp_splitk_ln(part, S, L.b, residual, gamma, beta, y32, ylp, dt, M, L.n, eps, L.w_scale, scale, stream);
    }
    TRY(madtp_gemm(a, L.w, L.b, residual, part, M, L.n, L.k, pld(dt, lda), ldw, L.n, L.n, dt, MADTP_F32, MADTP_ACT_NONE, L.w_scale,
                   scale, stream));
    return madtp_layernorm(part, gamma, beta, y32, ylp, dt, M, L.n, eps, stream);
}

struct VitWs {
    void *h, *qkv, *o, *mid;
    float *colsum, *p0, *onorm, *xp, *merge_w;
    int32_t* dst_pos;
    size_t bytes;
};

VitWs vit_carve(char* base, size_t cap, int B, int N, int dim, int hidden, int heads, int dt, bool* ok) {
    Carver c{base, 0, cap};
    const size_t M = (size_t)B * N, e = esz_of(dt);
    VitWs w;
    w.h = c.take(M * dim * e);
    w.qkv = c.take(M * 3 * dim * e);
    w.o = c.take(M * dim * e);
    w.mid = c.take(M * hidden * e);
    w.colsum = (float*)c.take((size_t)B * ((N + 15) / 16) * N * 4);
    w.p0 = (float*)c.take((size_t)B * heads * N * 4);
    w.onorm = (float*)c.take((size_t)B * heads * N * 4);
    w.xp = (float*)c.take(M * dim * 4);
    w.merge_w = (float*)c.take((size_t)B * N * 4);
    w.dst_pos = (int32_t*)c.take((size_t)B * N * 4);
    w.bytes = c.off;
    *ok = c.fits();
    return w;
}

struct BertWs {
    void *hc, *qkv, *ctx, *q, *q2, *kv, *kv1, *c0, *c1, *cat, *mid, *attc;
    float *t, *xp, *s, *att2, *merge_w, *colsum, *p0, *onorm, *part;
    int32_t* dst_pos;
    size_t bytes;
};

BertWs bert_carve(char* base, size_t cap, int B, int L, int Nk, int dim, int hidden, int heads, int dt, bool* ok) {
    Carver c{base, 0, cap};
    const size_t M = (size_t)B * L, MK = (size_t)B * (Nk > 0 ? Nk : 1), e = esz_of(dt);
    BertWs w;
    w.hc = c.take(M * dim * e);
    w.qkv = c.take(M * 3 * dim * e);
    w.ctx = c.take(M * dim * e);
    w.q = c.take(M * dim * e);
    w.q2 = c.take(M * 2 * dim * e);
    w.kv = c.take(MK * 2 * dim * e);
    w.kv1 = c.take(MK * 2 * dim * e);  // second branch (twin cross-attention: both projections in one launch)
    w.c0 = c.take(M * dim * e);
    w.c1 = c.take(M * dim * e);
    w.cat = c.take(M * 2 * dim * e);
    w.mid = c.take(M * hidden * e);
    w.attc = c.take(M * dim * e);
    w.t = (float*)c.take(M * dim * 4);
    w.xp = (float*)c.take(M * dim * 4);
    w.s = (float*)c.take(M * dim * 4);
    w.att2 = (float*)c.take(M * dim * 4);
    w.part = (float*)c.take(4 * M * dim * 4);
    w.merge_w = (float*)c.take((size_t)B * L * 4);
    w.colsum = (float*)c.take((size_t)B * ((L + 15) / 16) * L * 4);
    w.p0 = (float*)c.take((size_t)B * heads * L * 4);
    w.onorm = (float*)c.take((size_t)B * heads * L * 4);
    w.dst_pos = (int32_t*)c.take((size_t)B * L * 4);
    w.bytes = c.off;
    *ok = c.fits();
    return w;
}

}  // namespace

extern "C" size_t madtp_vit_block_workspace(int B, int N, int dim, int hidden, int heads, int dtype) {
    bool ok;
    return vit_carve(nullptr, 0, B, N, dim, hidden, heads, dtype, &ok).bytes;
}

// x = x + proj(attention(qkv(LN1(x))))  [+ importance score / threshold / count / kmax when temperature > 0]
static int vit_attn_impl(const madtp_vit_block_w* w, const float* x, float* x_out, void* ws, size_t ws_bytes, int B, int N,
                         const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature, float* score,
                         float* threshold, int32_t* count, int32_t* kmax, int32_t* k_host, void* stream) {
    if (!w || !x || !x_out || !ws || B <= 0 || N <= 0) return MADTP_E_BADARG;
    bool ok;
    VitWs s = vit_carve((char*)ws, ws_bytes, B, N, w->dim, w->fc1.n, w->heads, w->dtype, &ok);
    if (!ok) return MADTP_E_SHAPE;
    const int M = B * N, D = w->dim, dt = w->dtype, adt = attn_dt(dt);
    const size_t e = esz_of(adt);
    const bool prune = temperature > 0.f;
    if (prune && (!token_attn || !score || !threshold || !count)) return MADTP_E_BADARG;
    TRY(ln_to(x, w->ln1_g, w->ln1_b, nullptr, s.h, M, D, w->eps, dt, stream));
    TRY(lin(s.h, D, w->qkv, nullptr, 0, s.qkv, 3 * D, M, dt, adt, MADTP_ACT_NONE, 1.f, stream));
    const char* q = (const char*)s.qkv;
    // f16x3: the projection GEMM takes the attention output as f16 planes (s.h is free again) - written by the attention kernel itself
    // (round 6; a separate split_f16 launch re-read the f32 context before), unless the f16s attention kernels are switched off
    const bool ctx_planes = dt == MADTP_F16S && madtp_internal_attn_f16s_enabled();
    if (ctx_planes) madtp_internal_attn_split_out(D);
    const int rc_att = madtp_attention_qk_mask(q, q + (size_t)D * e, q + (size_t)2 * D * e, ctx_planes ? s.h : s.o, nullptr, w->attn_mask,
                                               w->ld_attn_mask, prune ? s.colsum : nullptr, s.p0, s.onorm, B, w->heads, N, N, 3 * D, 3 * D,
                                               3 * D, ctx_planes ? 2 * D : D, w->scale, attn_io(dt), stream);
    madtp_internal_attn_split_out(0);
    TRY(rc_att);
    if (ctx_planes) s.o = s.h;
    else if (dt == MADTP_F16S) {
        TRY(to_lp((const float*)s.o, D, s.h, M, D, dt, stream));
        s.o = s.h;
    }
    if (prune && k_host) {
        // k = max_b count is delivered through pinned host memory.  token_score only needs the attention statistics, so it
        // is launched BEFORE the projection GEMM and the host waits for k while that GEMM runs.
        int seq = 0;
        TRY(madtp_token_score_publish(s.colsum, (N + 15) / 16, s.p0, s.onorm, token_attn, ldt_row, ldt_batch, K, temperature,
                                      score, threshold, count, B, w->heads, N, &seq, stream));
        const int rc = lin(s.o, D, w->proj, x, D, x_out, D, M, dt, MADTP_F32, MADTP_ACT_NONE, 1.f, stream);
        const int rw = madtp_token_score_wait(seq, count, B, k_host, stream);
        return rc ? rc : rw;
    }
    TRY(lin(s.o, D, w->proj, x, D, x_out, D, M, dt, MADTP_F32, MADTP_ACT_NONE, 1.f, stream));
    if (prune) {
        if (kmax) {
            hipError_t he = hipMemsetAsync(kmax, 0, sizeof(int32_t), (hipStream_t)stream);
            if (he != hipSuccess) return (int)he;
        }
        TRY(madtp_token_score(s.colsum, (N + 15) / 16, s.p0, s.onorm, token_attn, ldt_row, ldt_batch, K, temperature, score,
                              threshold, count, kmax, B, w->heads, N, stream));
    }
    return 0;
}

extern "C" int madtp_vit_block_attn(const madtp_vit_block_w* w, const float* x, float* x_out, void* ws, size_t ws_bytes,
                                    int B, int N, const float* token_attn, int ldt_row, int ldt_batch, int K,
                                    float temperature, float* score, float* threshold, int32_t* count, int32_t* kmax,
                                    void* stream) {
    return vit_attn_impl(w, x, x_out, ws, ws_bytes, B, N, token_attn, ldt_row, ldt_batch, K, temperature, score, threshold,
                         count, kmax, nullptr, stream);
}

// [select top-k, gather + merge] ; y = x' + fc2(GELU(fc1(LN2(x'))))    k == 0: no pruning (x' = x), y is [B,N,dim];
// k > 0: y is [B,k+2,dim], indices [B,k], indices_sort [B,N-1] are written.
extern "C" int madtp_vit_block_mlp(const madtp_vit_block_w* w, const float* x, float* y, void* ws, size_t ws_bytes, int B,
                                   int N, int k, const float* score, int64_t* indices, int64_t* indices_sort, void* stream) {
    if (!w || !x || !y || !ws || B <= 0 || N <= 0 || k < 0) return MADTP_E_BADARG;
    bool ok;
    VitWs s = vit_carve((char*)ws, ws_bytes, B, N, w->dim, w->fc1.n, w->heads, w->dtype, &ok);
    if (!ok) return MADTP_E_SHAPE;
    const int D = w->dim, dt = w->dtype;
    const float* xr = x;
    int Np = N;
    if (k > 0) {
        if (!score || !indices || !indices_sort) return MADTP_E_BADARG;
        TRY(madtp_token_select(score, k, indices, indices_sort, s.dst_pos, s.merge_w, B, N - 1, stream));
        // gather + merge with norm2 fused in: the copying wave holds the row, so it emits LN(row) as well
        TRY(madtp_token_gather_ln(x, s.dst_pos, s.merge_w, s.xp, B, N, k, D, w->ln2_g, w->ln2_b, w->eps,
                                  dt == MADTP_F32 ? (float*)s.h : nullptr, dt == MADTP_F32 ? nullptr : s.h, dt, stream));
        xr = s.xp;
        Np = k + 2;
    }
    const int M = B * Np;
    if (k == 0) TRY(ln_to(xr, w->ln2_g, w->ln2_b, nullptr, s.h, M, D, w->eps, dt, stream));
    TRY(lin(s.h, D, w->fc1, nullptr, 0, s.mid, w->fc1.n, M, dt, dt, w->act, 1.f, stream));
    TRY(lin(s.mid, w->fc1.n, w->fc2, xr, D, y, D, M, dt, MADTP_F32, MADTP_ACT_NONE, 1.f, stream));
    return 0;
}

// Whole Block.forward (vit.py:184-205) in one call: attention half, k = max_b count read on the host (the reference's one
// synchronisation per layer, vit.py:145), the pruning rule of vit.py:148-149, MLP half launched straight away.
// y has room for [B,N,dim], indices for [B,N-1]; on return *k_used > 0 means y is [B,k_used+2,dim] and indices [B,k_used].
static int vit_block_impl(const madtp_vit_block_w* w, const float* x, float* x_attn, float* y, void* ws, size_t ws_bytes,
                          int B, int N, const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature,
                          float* score, float* threshold, int32_t* count, int64_t* indices, int64_t* indices_sort,
                          int max_keep, int* k_out, int* k_used, void* stream) {
    if (!k_out || !k_used) return MADTP_E_BADARG;
    *k_out = 0; *k_used = 0;
    int32_t k = 0;
    TRY(vit_attn_impl(w, x, x_attn, ws, ws_bytes, B, N, token_attn, ldt_row, ldt_batch, K, temperature, score, threshold, count,
                      nullptr, temperature > 0.f ? &k : nullptr, stream));
    if (temperature > 0.f) {
        *k_out = k;
        // vit.py:148-149 `k < 1` is max_keep = 0; clip/model.py:220-221 `k <= max_keep`
        if (!(k <= max_keep || (N - 1 - k) <= 1)) *k_used = k;
    }
    return madtp_vit_block_mlp(w, x_attn, y, ws, ws_bytes, B, N, *k_used, score, indices, indices_sort, stream);
}

extern "C" int madtp_vit_block(const madtp_vit_block_w* w, const float* x, float* x_attn, float* y, void* ws, size_t ws_bytes,
                               int B, int N, const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature,
                               float* score, float* threshold, int32_t* count, int64_t* indices, int64_t* indices_sort,
                               int* k_out, int* k_used, void* stream) {
    return vit_block_impl(w, x, x_attn, y, ws, ws_bytes, B, N, token_attn, ldt_row, ldt_batch, K, temperature, score, threshold,
                          count, indices, indices_sort, 0, k_out, k_used, stream);
}

extern "C" int madtp_vit_block_keep(const madtp_vit_block_w* w, const float* x, float* x_attn, float* y, void* ws, size_t ws_bytes,
                                    int B, int N, const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature,
                                    float* score, float* threshold, int32_t* count, int64_t* indices, int64_t* indices_sort,
                                    int max_keep, int* k_out, int* k_used, void* stream) {
    if (max_keep < 0) return MADTP_E_BADARG;
    return vit_block_impl(w, x, x_attn, y, ws, ws_bytes, B, N, token_attn, ldt_row, ldt_batch, K, temperature, score, threshold,
                          count, indices, indices_sort, max_keep, k_out, k_used, stream);
}

// Query_model (models/utils.py:147-183) on the token buffer in place: logits of ALL rows of x (the CLS row is computed
// and ignored) with the exact-f32 MFMA, then att_ft over the patch rows.
extern "C" int madtp_query_model(const float* x, const void* sd_w, const void* sd_hi, const void* sd_lo, int split_dtype,
                                 float sd_scale, int K, float* token_attn_full, float* att_ft, float* stats_ws, int accumulate,
                                 float inv_sqrt_sd, int B, int N, int dim, void* stream) {
    if (!x || !token_attn_full || B <= 0 || N < 2) return MADTP_E_BADARG;
    const int kp = (K + 127) / 128 * 128;
    const bool split = sd_hi && sd_lo;
    const bool fast = split && split_dtype == MADTP_BF16;  // bf16 att_ft kernel as well
    if (split) {
        if (kp != 128) return MADTP_E_SHAPE;
        TRY(madtp_align_logits(x, sd_hi, sd_lo, token_attn_full, B * N, dim, split_dtype, sd_scale, stream));
    } else {
        if (!sd_w) return MADTP_E_BADARG;
        TRY(madtp_gemm(x, sd_w, nullptr, nullptr, token_attn_full, B * N, kp, dim, dim, dim, kp, 0, MADTP_F32, MADTP_F32,
                       MADTP_ACT_NONE, 1.f, 1.f, stream));
    }
    if (att_ft)
        TRY(madtp_query_att_ft(token_attn_full + kp, kp, N * kp, K, x + dim, dim, N * dim, att_ft, inv_sqrt_sd, accumulate, B,
                               N - 1, dim, fast ? 1 : 0, stats_ws, stream));
    return 0;
}

extern "C" size_t madtp_bert_layer_workspace(int B, int L, int Nk, int dim, int hidden, int heads, int dtype) {
    bool ok;
    return bert_carve(nullptr, 0, B, L, Nk, dim, hidden, heads, dtype, &ok).bytes;
}

// att = LayerNorm(dense(self_attention(hidden)) + hidden)   [+ score/threshold/count/kmax when temperature > 0]
static int bert_attn_impl(const madtp_bert_layer_w* w, const float* hidden, const float* mask2d, float* att, void* ws,
                          size_t ws_bytes, int B, int L, int Nk, const float* token_attn, int ldt_row, int ldt_batch, int K,
                          float temperature, float* score, float* threshold, int32_t* count, int32_t* kmax, int32_t* k_host,
                          const void* hidden_lp, void* stream) {
    if (!w || !hidden || !att || !ws || B <= 0 || L <= 0) return MADTP_E_BADARG;
    bool ok;
    BertWs s = bert_carve((char*)ws, ws_bytes, B, L, Nk, w->dim, w->inter.n, w->heads, w->dtype, &ok);
    if (!ok) return MADTP_E_SHAPE;
    const int M = B * L, D = w->dim, dt = w->dtype, adt = attn_dt(dt);
    const size_t e = esz_of(adt);
    const bool prune = temperature > 0.f;
    if (prune && (!token_attn || !score || !threshold || !count || !mask2d)) return MADTP_E_BADARG;
    const void* hc = hidden;
    if (dt != MADTP_F32) {
        if (hidden_lp) hc = hidden_lp;  // the previous layer's LayerNorm already emitted the compute-dtype copy
        else {
            TRY(to_lp(hidden, D, s.hc, M, D, dt, stream));
            hc = s.hc;
        }
    }
    void* att_lp = dt != MADTP_F32 ? s.attc : nullptr;  // compute-dtype copy of att for the second half (same workspace)
    TRY(lin(hc, D, w->qkv, nullptr, 0, s.qkv, 3 * D, M, dt, adt, MADTP_ACT_NONE, 1.f, stream));
    const char* q = (const char*)s.qkv;
    // f16x3: attention.output.dense takes the context as f16 planes (s.q is scratch of the second half), written by the attention kernel
    const bool ctx_planes = dt == MADTP_F16S && madtp_internal_attn_f16s_enabled();
    void* ctx_out = ctx_planes ? s.q : s.ctx;
    const int ld_ctx = ctx_planes ? 2 * D : D;
    if (ctx_planes) madtp_internal_attn_split_out(D);
    int rc_att;
    if (w->self_mask_qk)  // decoder layer (BertModel(is_decoder=True), med.py:752-768): causal [L,L] mask next to the padding mask
        rc_att = madtp_attention_qk_mask(q, q + (size_t)D * e, q + (size_t)2 * D * e, ctx_out, mask2d, w->self_mask_qk, w->ld_self_mask_qk,
                                         prune ? s.colsum : nullptr, s.p0, s.onorm, B, w->heads, L, L, 3 * D, 3 * D, 3 * D, ld_ctx, w->scale,
                                         attn_io(dt), stream);
    else
        rc_att = madtp_attention(q, q + (size_t)D * e, q + (size_t)2 * D * e, ctx_out, mask2d, prune ? s.colsum : nullptr, s.p0, s.onorm,
                                 B, w->heads, L, L, 3 * D, 3 * D, 3 * D, ld_ctx, w->scale, attn_io(dt), stream);
    madtp_internal_attn_split_out(0);
    TRY(rc_att);
    if (ctx_planes) s.ctx = s.q;
    else if (dt == MADTP_F16S) {
        TRY(to_lp((const float*)s.ctx, D, s.q, M, D, dt, stream));
        s.ctx = s.q;
    }
    if (prune && k_host) {  // as in the ViT block: score first, the output projection + LayerNorm run while the host waits
        int seq = 0;
        TRY(madtp_token_score_publish(s.colsum, (L + 15) / 16, s.p0, s.onorm, token_attn, ldt_row, ldt_batch, K, temperature,
                                      score, threshold, count, B, w->heads, L, &seq, stream));
        const int rc = lin_ln(s.ctx, D, w->attn_out, hidden, 1.f, w->ln_att_g, w->ln_att_b, att, att_lp, M, dt, w->eps, s.part,
                              stream);
        const int rw = madtp_token_score_wait(seq, count, B, k_host, stream);
        return rc ? rc : rw;
    }
    TRY(lin_ln(s.ctx, D, w->attn_out, hidden, 1.f, w->ln_att_g, w->ln_att_b, att, att_lp, M, dt, w->eps, s.part, stream));
    if (prune) {
        if (kmax) {
            hipError_t he = hipMemsetAsync(kmax, 0, sizeof(int32_t), (hipStream_t)stream);
            if (he != hipSuccess) return (int)he;
        }
        TRY(madtp_token_score(s.colsum, (L + 15) / 16, s.p0, s.onorm, token_attn, ldt_row, ldt_batch, K, temperature, score,
                              threshold, count, kmax, B, w->heads, L, stream));
    }
    return 0;
}

extern "C" int madtp_bert_layer_attn(const madtp_bert_layer_w* w, const float* hidden, const float* mask2d, float* att,
                                     void* ws, size_t ws_bytes, int B, int L, int Nk, const float* token_attn, int ldt_row,
                                     int ldt_batch, int K, float temperature, float* score, float* threshold, int32_t* count,
                                     int32_t* kmax, void* stream) {
    return bert_attn_impl(w, hidden, mask2d, att, ws, ws_bytes, B, L, Nk, token_attn, ldt_row, ldt_batch, K, temperature, score,
                          threshold, count, kmax, nullptr, nullptr, stream);
}

// [prune att + mask] ; [cross-attention to the image tokens] ; y = LayerNorm(output(GELU(intermediate(a))) + a)
// cross_mode: 0 = text mode (no cross-attention), otherwise w->cross selects single (MED) or twin (NLVR).
// enc0/enc1: image tokens [B*Nk, dim] in the compute dtype; enc_mask0/1: additive f32 [B,Nk] or NULL.
// att_lp_ready: the workspace's bf16 copy of att was written by the first half of the SAME call (fused layer entry point);
// y_lp (optional): bf16 copy of y for the next layer.
static int bert_rest_impl(const madtp_bert_layer_w* w, const float* att, const float* mask2d, float* y, float* mask_out, void* ws,
                          size_t ws_bytes, int B, int L, int k, const float* score, int64_t* indices, int64_t* indices_sort,
                          int cross_mode, const void* enc0, const void* enc1, int Nk, const float* enc_mask0,
                          const float* enc_mask1, bool att_lp_ready, void* y_lp, const void* kv_pre0, const void* kv_pre1,
                          const int32_t* kv_index, int kv_ld, void* stream) {
    if (!w || !att || !y || !ws || B <= 0 || L <= 0 || k < 0) return MADTP_E_BADARG;
    bool ok;
    BertWs s = bert_carve((char*)ws, ws_bytes, B, L, Nk, w->dim, w->inter.n, w->heads, w->dtype, &ok);
    if (!ok) return MADTP_E_SHAPE;
    const int D = w->dim, dt = w->dtype, adt = attn_dt(dt);
    const size_t e = esz_of(adt);
    const bool lpm = dt != MADTP_F32;  // a compute-dtype copy of the f32 activations feeds the GEMMs
    const float* a32 = att;
    const float* m2 = mask2d;
    int Lp = L;
    if (k > 0) {
        if (!score || !indices || !indices_sort) return MADTP_E_BADARG;
        TRY(madtp_token_select(score, k, indices, indices_sort, s.dst_pos, s.merge_w, B, L - 1, stream));
        TRY(madtp_token_gather(att, s.dst_pos, s.merge_w, s.xp, B, L, k, D, stream));
        if (mask2d) {
            if (!mask_out) return MADTP_E_BADARG;
            if (w->variant_nlvr) TRY(madtp_mask_gather(mask2d, indices_sort, L - 1, nullptr, 0, mask_out, B, L, k, stream));
            else TRY(madtp_mask_gather(mask2d, indices, k, indices_sort, L - 1, mask_out, B, L, k, stream));
            m2 = mask_out;
        }
        a32 = s.xp;
        Lp = k + 2;
    }
    (void)m2;  // the text-side padding mask only feeds the NEXT layer's self-attention
    const int M = B * Lp;
    const void* ac = a32;
    if (lpm) {
        if (!(att_lp_ready && k == 0)) TRY(to_lp(a32, D, s.attc, M, D, dt, stream));
        ac = s.attc;
    }
    if (cross_mode && w->cross) {
        const int nbr = w->cross == 2 ? 2 : 1;
        if (Nk <= 0 || (!enc0 && !kv_pre0) || (nbr == 2 && !enc1 && !kv_pre1)) return MADTP_E_BADARG;
        if (nbr == 2 && w->fused_twin) {
            // twin branches with fused projections: one q GEMM ([q0|q1], N = 2D), the two context tensors written
            // side by side ([c0|c1], ld 2D) and ONE output GEMM over K = 2D (dense0|dense1, merge_layer folded in)
            TRY(lin(ac, D, w->cq_fused, nullptr, 0, s.q2, 2 * D, M, dt, adt, MADTP_ACT_NONE, 1.f, stream));
            // [k|v] = enc @ [Wk|Wv]^T of BOTH branches in one launch (or read from the caller's cache of projected blocks)
            const bool both_here = !kv_pre0 && !kv_pre1 && w->ckv[0].n == w->ckv[1].n && w->ckv[0].k == w->ckv[1].k;
            if (both_here)
                TRY(madtp_gemm_pair(enc0, enc1, w->ckv[0].w, w->ckv[1].w, w->ckv[0].b, w->ckv[1].b, s.kv, s.kv1, B * Nk,
                                    w->ckv[0].n, w->ckv[0].k, pld(dt, D), (dt == MADTP_F16S ? 2 : 1) * w->ckv[0].k, 2 * D, dt, adt,
                                    w->ckv[0].w_scale, w->ckv[1].w_scale, stream));
            const char* kvp[2];
            int ldkv[2];
            for (int br = 0; br < 2; ++br) {
                const void* enc = br ? enc1 : enc0;
                const char* kv = (const char*)(br ? kv_pre1 : kv_pre0);
                ldkv[br] = (kv && kv_ld) ? kv_ld : 2 * D;
                if (!kv) {
                    void* dst = br ? s.kv1 : s.kv;
                    if (!both_here) TRY(lin(enc, D, w->ckv[br], nullptr, 0, dst, 2 * D, B * Nk, dt, adt, MADTP_ACT_NONE, 1.f, stream));
                    kv = (const char*)dst;
                }
                kvp[br] = kv;
            }
            const float* em0 = w->variant_nlvr ? enc_mask0 : nullptr;
            const float* em1 = w->variant_nlvr ? enc_mask1 : nullptr;
            // f16x3: [c0|c1] as the f16 planes of ONE [M, 2D] matrix for the fused output GEMM (s.mid is free until the FFN): branch br
            // writes columns br D .. of both planes (plane offset 2D, row length 4D f16) from its attention kernel
            const bool cat_planes = dt == MADTP_F16S && madtp_internal_attn_f16s_enabled();
            char* cat0 = cat_planes ? (char*)s.mid : (char*)s.cat;
            char* cat1 = cat_planes ? (char*)s.mid + (size_t)D * 2 : (char*)s.cat + (size_t)D * e;
            const int ld_cat = cat_planes ? 4 * D : 2 * D;
            if (cat_planes) madtp_internal_attn_split_out(2 * D);
            int rc_att = 0;
            if (ldkv[0] == ldkv[1] && (!kv_pre0) == (!kv_pre1)) {
                // both branches in one launch ([c0|c1] side by side, ld 2D)
                rc_att = madtp_attention_pair(s.q2, (const char*)s.q2 + (size_t)D * e, kvp[0], kvp[1], kvp[0] + (size_t)D * e,
                                              kvp[1] + (size_t)D * e, kv_pre0 ? kv_index : nullptr, cat0, cat1,
                                              em0, em1, B, w->heads, Lp, Nk, 2 * D, ldkv[0], ldkv[0], ld_cat, w->scale, attn_io(dt), stream);
            } else {
                for (int br = 0; br < 2 && !rc_att; ++br)
                    rc_att = madtp_attention_indexed((const char*)s.q2 + (size_t)br * D * e, kvp[br], kvp[br] + (size_t)D * e,
                                                     (br ? kv_pre1 : kv_pre0) ? kv_index : nullptr, br ? cat1 : cat0,
                                                     br ? em1 : em0, nullptr, nullptr, nullptr, B, w->heads, Lp, Nk, 2 * D, ldkv[br],
                                                     ldkv[br], ld_cat, w->scale, attn_io(dt), stream);
            }
            madtp_internal_attn_split_out(0);
            TRY(rc_att);
            const void* catc = s.cat;
            if (cat_planes) catc = s.mid;
            else if (dt == MADTP_F16S) {  // [c0|c1] f32 -> f16 planes
                TRY(to_lp((const float*)s.cat, 2 * D, s.mid, M, 2 * D, dt, stream));
                catc = s.mid;
            }
            TRY(lin_ln(catc, 2 * D, w->cdense_fused, a32, w->fused_twin == 2 ? 1.f : 0.5f, w->ln_cross_g, w->ln_cross_b, s.att2,
                       lpm ? s.attc : nullptr, M, dt, w->eps, s.part, stream));
            a32 = s.att2;
            ac = lpm ? (const void*)s.attc : (const void*)s.att2;
            goto ffn;
        }
        void* cbuf[2] = {s.c0, s.c1};
        for (int br = 0; br < nbr; ++br) {
            const void* enc = br ? enc1 : enc0;
            // med.py:197-199 drops the encoder mask in cross-attention; nlvr_encoder.py:196-198 applies it
            const float* em = w->variant_nlvr ? (br ? enc_mask1 : enc_mask0) : nullptr;
            TRY(lin(ac, D, w->cq[br], nullptr, 0, s.q, D, M, dt, adt, MADTP_ACT_NONE, 1.f, stream));
            const char* kv = (const char*)(br ? kv_pre1 : kv_pre0);
            if (!kv) {
                TRY(lin(enc, D, w->ckv[br], nullptr, 0, s.kv, 2 * D, B * Nk, dt, adt, MADTP_ACT_NONE, 1.f, stream));
                kv = (const char*)s.kv;
            }
            const int ldkv = ((br ? kv_pre1 : kv_pre0) && kv_ld) ? kv_ld : 2 * D;
            // f16x3: the context as f16 planes for crossattention.output.dense, in s.mid (free until the FFN; hidden >= 2*dim, so both
            // branches fit side by side) - written by the attention kernel, else split from the f32 context
            const bool ctx_planes = dt == MADTP_F16S && madtp_internal_attn_f16s_enabled();
            void* sp = nullptr;
            if (dt == MADTP_F16S) {
                if (w->inter.n < 2 * D) return MADTP_E_SHAPE;
                sp = (char*)s.mid + (size_t)br * M * D * esz_of(dt);
            }
            if (ctx_planes) madtp_internal_attn_split_out(D);
            const int rc_att = madtp_attention_indexed(s.q, kv, kv + (size_t)D * e, (br ? kv_pre1 : kv_pre0) ? kv_index : nullptr,
                                                       ctx_planes ? sp : cbuf[br], em, nullptr, nullptr, nullptr, B, w->heads, Lp, Nk, D, ldkv,
                                                       ldkv, ctx_planes ? 2 * D : D, w->scale, attn_io(dt), stream);
            madtp_internal_attn_split_out(0);
            TRY(rc_att);
            if (dt == MADTP_F16S) {
                if (!ctx_planes) TRY(to_lp((const float*)cbuf[br], D, sp, M, D, dt, stream));
                cbuf[br] = sp;
            }
        }
        if (nbr == 2) {
            if (w->has_merge) {  // nlvr_encoder.py:263-264
                // [dense0(c0) | dense1(c1)] is written as f32 in the f16-split mode (the planes of a split matrix are
                // separated by ITS row length, which two column-block writers of half the width cannot produce) and split once
                const int cdt = dt == MADTP_F16S ? MADTP_F32 : dt;
                char* cat = (char*)s.cat;
                TRY(lin(cbuf[0], D, w->cdense[0], nullptr, 0, cat, 2 * D, M, dt, cdt, MADTP_ACT_NONE, 1.f, stream));
                TRY(lin(cbuf[1], D, w->cdense[1], nullptr, 0, cat + (size_t)D * esz_of(cdt), 2 * D, M, dt, cdt, MADTP_ACT_NONE, 1.f, stream));
                const void* catc = s.cat;
                if (dt == MADTP_F16S) {
                    TRY(to_lp((const float*)s.cat, 2 * D, s.mid, M, 2 * D, dt, stream));
                    catc = s.mid;
                }
                TRY(lin(catc, 2 * D, w->merge, a32, D, s.s, D, M, dt, MADTP_F32, MADTP_ACT_NONE, 1.f, stream));
            } else {  // :266 (h0+h1)/2 folded into the epilogues
                TRY(lin(cbuf[0], D, w->cdense[0], a32, D, s.t, D, M, dt, MADTP_F32, MADTP_ACT_NONE, 0.5f, stream));
                TRY(lin(cbuf[1], D, w->cdense[1], s.t, D, s.s, D, M, dt, MADTP_F32, MADTP_ACT_NONE, 0.5f, stream));
            }
        } else {
            TRY(lin(cbuf[0], D, w->cdense[0], a32, D, s.s, D, M, dt, MADTP_F32, MADTP_ACT_NONE, 1.f, stream));
        }
        TRY(madtp_layernorm(s.s, w->ln_cross_g, w->ln_cross_b, s.att2, lpm ? s.attc : nullptr, dt, M, D, w->eps, stream));
        a32 = s.att2;
        ac = lpm ? (const void*)s.attc : (const void*)s.att2;
    }
ffn:
    TRY(lin(ac, D, w->inter, nullptr, 0, s.mid, w->inter.n, M, dt, dt, MADTP_ACT_GELU_ERF, 1.f, stream));
    TRY(lin_ln(s.mid, w->inter.n, w->out, a32, 1.f, w->ln_out_g, w->ln_out_b, y, lpm ? y_lp : nullptr, M, dt, w->eps,
               s.part, stream));
    return 0;
}

extern "C" int madtp_bert_layer_rest(const madtp_bert_layer_w* w, const float* att, const float* mask2d, float* y,
                                     float* mask_out, void* ws, size_t ws_bytes, int B, int L, int k, const float* score,
                                     int64_t* indices, int64_t* indices_sort, int cross_mode, const void* enc0,
                                     const void* enc1, int Nk, const float* enc_mask0, const float* enc_mask1, void* stream) {
    return bert_rest_impl(w, att, mask2d, y, mask_out, ws, ws_bytes, B, L, k, score, indices, indices_sort, cross_mode, enc0, enc1,
                          Nk, enc_mask0, enc_mask1, false, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// Whole BertLayer.forward in one call (med.py:393-467 / nlvr_encoder.py:484-559): self-attention half, host read of
// k = max_b count, the pruning rule of med.py:374-375, then the rest.  y has room for [B,L,dim], mask_out for [B,L],
// indices for [B,L-1]; *k_used > 0 means y is [B,k_used+2,dim], mask_out [B,k_used+2], indices [B,k_used].
extern "C" int madtp_bert_layer(const madtp_bert_layer_w* w, const float* hidden, const float* mask2d, float* att, float* y,
                                float* mask_out, void* ws, size_t ws_bytes, int B, int L, int Nk, const float* token_attn,
                                int ldt_row, int ldt_batch, int K, float temperature, float* score, float* threshold,
                                int32_t* count, int64_t* indices, int64_t* indices_sort, int cross_mode, const void* enc0,
                                const void* enc1, const float* enc_mask0, const float* enc_mask1, const void* hidden_lp,
                                void* y_lp, const void* kv_pre0, const void* kv_pre1, const int32_t* kv_index, int kv_ld, int* k_out,
                                int* k_used, void* stream) {
    if (!k_out || !k_used) return MADTP_E_BADARG;
    *k_out = 0; *k_used = 0;
    int32_t k = 0;
    TRY(bert_attn_impl(w, hidden, mask2d, att, ws, ws_bytes, B, L, Nk, token_attn, ldt_row, ldt_batch, K, temperature, score,
                       threshold, count, nullptr, temperature > 0.f ? &k : nullptr, hidden_lp, stream));
    if (temperature > 0.f) {
        *k_out = k;
        if (!(k < 1 || (L - 1 - k) <= 1)) *k_used = k;
    }
    return bert_rest_impl(w, att, mask2d, y, mask_out, ws, ws_bytes, B, L, *k_used, score, indices, indices_sort, cross_mode, enc0,
                          enc1, Nk, enc_mask0, enc_mask1, true, y_lp, kv_pre0, kv_pre1, kv_index, kv_ld, stream);
}

// ---- incremental decoding (models/med.py:1071-1094 prepare_inputs_for_generation / past_key_values) ---------------------------
// One decoder step for ONE new token per row: every layer projects the new token to q|k|v, appends k|v to its self-attention cache
// at position t (cache [n_layers][rows][Lmax][2 dim] in the attention dtype), attends the new query to positions 0..t of its row
// (no causal mask needed: everything in the cache is in the past), then output projection + LayerNorm, cross-attention against the
// caller's cached encoder K/V (kv_pre / kv_index: the EncoderKVCache of f1) and the FFN - the second half is bert_rest_impl on
// [rows / group, group, dim].  x: embedded new tokens [rows, dim] f32 (position t), y: [rows, dim] f32.  Beam re-ordering of the cache between
// steps (_reorder_cache, :1091-1094) is the caller's gather over the rows.
extern "C" int madtp_bert_decode_step(const madtp_bert_layer_w* const* layers, int n_layers, const float* x, void* kv_cache, int rows,
                                      int t, int Lmax, const void* const* kv_pre, const int32_t* kv_index, int kv_ld, int Nk, int group,
                                      float* y, void* ws, size_t ws_bytes, void* stream) {
    if (!layers || !x || !kv_cache || !kv_pre || !y || !ws || n_layers <= 0 || rows <= 0 || t < 0 || t >= Lmax || Lmax > 256 || Nk <= 0)
        return MADTP_E_BADARG;
    if (group < 1 || rows % group) return MADTP_E_BADARG;
    const float* h = x;
    for (int l = 0; l < n_layers; ++l) {
        const madtp_bert_layer_w* w = layers[l];
        if (!w || w->cross != 1 || !kv_pre[l]) return MADTP_E_BADARG;  // MED decoder layers with single cross-attention
        bool ok;
        // (the carve of the second half - bert_rest_impl on [rows / group, group, dim] - so that both halves agree on the buffers)
        BertWs s = bert_carve((char*)ws, ws_bytes, rows / group, group, Nk, w->dim, w->inter.n, w->heads, w->dtype, &ok);
        if (!ok) return MADTP_E_SHAPE;
        const int D = w->dim, dt = w->dtype, adt = attn_dt(dt);
        const size_t e = esz_of(adt);
        // the compute-dtype copy of the layer input: layer 0 casts x; every later layer finds it in s.hc, where the layer before
        // left it with its last kernel (the output LayerNorm emits it - same carve, same offset, nothing in between writes there)
        const void* hc = h;
        if (dt != MADTP_F32) { if (l == 0) TRY(to_lp(h, D, s.hc, rows, D, dt, stream)); hc = s.hc; }
        TRY(lin(hc, D, w->qkv, nullptr, 0, s.qkv, 3 * D, rows, dt, adt, MADTP_ACT_NONE, 1.f, stream));
        char* cache_l = (char*)kv_cache + (size_t)l * rows * Lmax * 2 * D * e;
        hipError_t he = hipMemcpy2DAsync(cache_l + (size_t)t * 2 * D * e, (size_t)Lmax * 2 * D * e, (const char*)s.qkv + (size_t)D * e,
                                         (size_t)3 * D * e, (size_t)2 * D * e, rows, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (he != hipSuccess) return (int)he;
        TRY(madtp_i_attention_cached(s.qkv, cache_l, cache_l + (size_t)D * e, Lmax, s.ctx, rows, w->heads, 1, t + 1, 3 * D, 2 * D, 2 * D, D,
                                     w->scale, attn_io(dt), stream));
        const void* ctx = s.ctx;
        if (dt == MADTP_F16S) { TRY(to_lp((const float*)s.ctx, D, s.q, rows, D, dt, stream)); ctx = s.q; }
        float* att = s.t;  // (f32 scratch the second half does not use in the single cross-attention path)
        TRY(lin_ln(ctx, D, w->attn_out, h, 1.f, w->ln_att_g, w->ln_att_b, att, dt != MADTP_F32 ? s.attc : nullptr, rows, dt, w->eps, s.part,
                   stream));
        // the caller's y serves every layer: layer l + 1 reads it as its input (operand copy, residual of the output projection) and
        // writes it again only with its very last kernel
        // group > 1: `group` consecutive rows (the beams of one item) read the same cached encoder [k|v] block - the second half
        // runs them as ONE sequence of `group` query rows (its row-wise kernels do not care, the cross-attention reads the item's
        // K/V once instead of once per beam); kv_index then has one entry per item
        TRY(bert_rest_impl(w, att, nullptr, y, nullptr, ws, ws_bytes, rows / group, group, 0, nullptr, nullptr, nullptr, 1, nullptr, nullptr,
                           Nk, nullptr, nullptr, true, (dt != MADTP_F32 && l + 1 < n_layers) ? s.hc : nullptr, kv_pre[l], nullptr, kv_index,
                           kv_ld, stream));
        h = y;
    }
    return 0;
}

// _reorder_cache (models/med.py:1091-1094) for the cache of madtp_bert_decode_step: dst[l, r, 0..t) = src[l, beam_src[r], 0..t) - only
// the t positions filled so far (an index_select over the whole [layers, rows, Lmax, 2 dim] tensor moved Lmax / t times the bytes).
namespace {
__global__ __launch_bounds__(256) void kv_reorder_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int64_t* __restrict__ beam_src,
                                                         int rows, size_t row_u4, size_t used_u4) {
    const int r = blockIdx.x, l = blockIdx.y;
    const uint4* s = src + ((size_t)l * rows + (size_t)beam_src[r]) * row_u4;
    uint4* d = dst + ((size_t)l * rows + r) * row_u4;
    for (size_t i = threadIdx.x; i < used_u4; i += 256) d[i] = s[i];
}
}  // namespace
extern "C" int madtp_kv_cache_reorder(const void* src, void* dst, const int64_t* beam_src, int n_layers, int rows, int Lmax, int t,
                                      int row_bytes, void* stream) {
    if (!src || !dst || !beam_src || src == dst || n_layers <= 0 || rows <= 0 || t < 0 || t > Lmax || row_bytes <= 0) return MADTP_E_BADARG;
    if (row_bytes % 16 || !aligned16(src) || !aligned16(dst)) return MADTP_E_ALIGN;
    if (t == 0) return 0;
    hipLaunchKernelGGL(kv_reorder_kernel, dim3(rows, n_layers), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, beam_src, rows,
                       (size_t)Lmax * row_bytes / 16, (size_t)t * row_bytes / 16);
    MADTP_LAUNCH_CHECK();
    return 0;
}

// ---- encoder-level entry points (include/madtp_hip.h): the layer loops of the two encoders in C --------------------------
static int query_step(const madtp_query_w* q, const float* x, float* logits, int layer, int B, int N, int dim, void* stream) {
    return madtp_query_model(x, q->sd_w, q->sd_hi, q->sd_lo, q->split_dtype, q->sd_scale, q->K, logits, q->att_ft, q->stats_ws,
                             layer > 0 ? 1 : 0, q->inv_sqrt_sd, B, N, dim, stream);
}

extern "C" int madtp_vit_encoder(const madtp_vit_block_w* const* layers, int n_layers, const madtp_query_w* q, const float* x0,
                                 madtp_layer_io* io, void* ws, size_t ws_bytes, int B, int N0, float temperature, void* stream) {
    if (!layers || !io || !x0 || n_layers <= 0 || B <= 0 || N0 <= 0) return MADTP_E_BADARG;
    const float* x = x0;
    int N = N0;
    const bool prune = q && temperature > 0.f;
    const int kp = q ? (q->K + 127) / 128 * 128 : 0;
    for (int l = 0; l < n_layers; ++l) {
        madtp_layer_io& o = io[l];
        const madtp_vit_block_w* w = layers[l];
        if (!w || !o.x_attn || !o.y) return MADTP_E_BADARG;
        if (q) {
            if (!o.logits) return MADTP_E_BADARG;
            TRY(query_step(q, x, o.logits, l, B, N, w->dim, stream));
        }
        int k_out = 0, k_used = 0;
        if (prune)
            TRY(madtp_vit_block(w, x, o.x_attn, o.y, ws, ws_bytes, B, N, o.logits + kp, kp, N * kp, q->K, temperature, o.score,
                                o.threshold, o.count, o.indices, o.indices_sort, &k_out, &k_used, stream));
        else
            TRY(madtp_vit_block(w, x, o.x_attn, o.y, ws, ws_bytes, B, N, nullptr, 0, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr,
                                nullptr, &k_out, &k_used, stream));
        if (k_used > 0) N = k_used + 2;
        o.k_out = k_out; o.k_used = k_used; o.n_out = N;
        x = o.y;
    }
    return 0;
}

// VisionTransformer.forward's block loop WITHOUT the per-layer host read of k (include/madtp_hip.h):
// token_score's last workgroup leaves the layer's decision in a device-side record (dims[l] = {N_l, k, k applied, N_{l+1}}) and
// every later kernel of the stream - top-k select, gather + norm2, the MLP GEMMs' M, the next layer's LayerNorm / alignment
// logits / qkv GEMM / attention - reads its size from there; grids, key-tile instantiations and buffers are the unpruned case's.
// The arithmetic per element is that of the per-layer path (same kernels, same k order, same reduction trees).
static int lin_dev(const void* a, int lda, const madtp_lin& L, const float* residual, int ldr, void* c, int ldc, int M_max, int dt,
                   int c_dt, int act, DevN m, void* stream) {
    return madtp_i_gemm(a, L.w, L.b, residual, c, M_max, L.n, L.k, pld(dt, lda), (dt == MADTP_F16S ? 2 : 1) * L.k, pld(c_dt, ldc), ldr, dt,
                        c_dt, act, L.w_scale, 1.f, m, stream);
}

extern "C" int madtp_vit_encoder_async(const madtp_vit_block_w* const* layers, int n_layers, const madtp_query_w* q, const float* x0,
                                       madtp_layer_io* io, void* ws, size_t ws_bytes, int B, int N0, float temperature,
                                       int32_t* dims_dev, int32_t* dims_host, void* stream) {
    if (!layers || !io || !x0 || !q || !dims_dev || n_layers <= 0 || B <= 0 || N0 < 3) return MADTP_E_BADARG;
    if (!(temperature > 0.f) || q->att_ft) return MADTP_E_BADARG;  // (att_ft: the caller's deferred sum, after this call)
    if ((long)B * N0 >= 4096 || N0 > 256) return MADTP_E_SHAPE;   // small-tile GEMMs and the <= 256-key attention kernels only
    hipStream_t s = (hipStream_t)stream;
    const size_t nrec = (size_t)(n_layers + 1) * DIMS_STRIDE;
    int32_t* ticket = dims_dev + nrec;  // token_score's arrival counter (resets itself)
    hipError_t he = hipMemsetAsync(dims_dev, 0, (nrec + 4) * sizeof(int32_t), s);
    if (he == hipSuccess) he = hipMemsetD32Async((hipDeviceptr_t)dims_dev, N0, 1, s);
    if (he != hipSuccess) return (int)he;
    const int kp = (q->K + 127) / 128 * 128;
    const bool split = q->sd_hi && q->sd_lo;
    if (split && kp != 128) return MADTP_E_SHAPE;
    const float* x = x0;
    const int Mmax = B * N0;
    for (int l = 0; l < n_layers; ++l) {
        madtp_layer_io& o = io[l];
        const madtp_vit_block_w* w = layers[l];
        if (!w || !o.x_attn || !o.y || !o.logits || !o.score || !o.threshold || !o.count || !o.indices || !o.indices_sort || w->attn_mask)
            return MADTP_E_BADARG;
        bool ok;
        VitWs v = vit_carve((char*)ws, ws_bytes, B, N0, w->dim, w->fc1.n, w->heads, w->dtype, &ok);
        if (!ok) return MADTP_E_SHAPE;
        const int D = w->dim, dt = w->dtype, adt = attn_dt(dt);
        const size_t e = esz_of(adt);
        int32_t* dl = dims_dev + (size_t)l * DIMS_STRIDE;
        const DevN m_in{dl, B, 0}, m_out{dl + DIMS_STRIDE, B, 0};
        // query model (models/utils.py:170): logits of all rows of x
        if (split) TRY(madtp_i_align_logits(x, q->sd_hi, q->sd_lo, o.logits, Mmax, D, q->split_dtype, q->sd_scale, m_in, stream));
        else TRY(madtp_i_gemm(x, q->sd_w, nullptr, nullptr, o.logits, Mmax, kp, D, D, D, kp, 0, MADTP_F32, MADTP_F32, MADTP_ACT_NONE, 1.f,
                              1.f, m_in, stream));
        // attention half (vit_attn_impl)
        if (dt == MADTP_F32) TRY(madtp_i_layernorm(x, w->ln1_g, w->ln1_b, (float*)v.h, nullptr, MADTP_BF16, Mmax, D, w->eps, m_in, stream));
        else TRY(madtp_i_layernorm(x, w->ln1_g, w->ln1_b, nullptr, v.h, dt, Mmax, D, w->eps, m_in, stream));
        TRY(lin_dev(v.h, D, w->qkv, nullptr, 0, v.qkv, 3 * D, Mmax, dt, adt, MADTP_ACT_NONE, m_in, stream));
        const char* qp = (const char*)v.qkv;
        TRY(madtp_i_attention(qp, qp + (size_t)D * e, qp + (size_t)2 * D * e, v.o, v.colsum, v.p0, v.onorm, B, w->heads, N0, 3 * D, 3 * D,
                              3 * D, D, w->scale, attn_io(dt), dl, stream));
        void* attn_o = v.o;
        if (dt == MADTP_F16S) {
            TRY(madtp_i_split_f16((const float*)v.o, D, v.h, 2 * D, Mmax, D, m_in, stream));
            attn_o = v.h;
        }
        TRY(madtp_i_token_score_dev(v.colsum, v.p0, v.onorm, o.logits, kp, q->K, temperature, o.score, o.threshold, o.count, B, w->heads,
                                    N0, dl, ticket, stream));
        TRY(lin_dev(attn_o, D, w->proj, x, D, o.x_attn, D, Mmax, dt, MADTP_F32, MADTP_ACT_NONE, m_in, stream));
        // pruning + MLP half (madtp_vit_block_mlp); a layer that is not pruned runs the gather as a copy (+ norm2)
        TRY(madtp_i_token_select_dev(o.score, o.indices, o.indices_sort, v.dst_pos, v.merge_w, B, N0 - 1, dl, stream));
        TRY(madtp_i_token_gather_ln_dev(o.x_attn, v.dst_pos, v.merge_w, v.xp, B, N0, D, w->ln2_g, w->ln2_b, w->eps,
                                        dt == MADTP_F32 ? (float*)v.h : nullptr, dt == MADTP_F32 ? nullptr : v.h, dt, dl, stream));
        TRY(lin_dev(v.h, D, w->fc1, nullptr, 0, v.mid, w->fc1.n, Mmax, dt, dt, w->act, m_out, stream));
        TRY(lin_dev(v.mid, w->fc1.n, w->fc2, v.xp, D, o.y, D, Mmax, dt, MADTP_F32, MADTP_ACT_NONE, m_out, stream));
        x = o.y;
    }
    // dims_host == NULL (round 6): ENQUEUE ONLY - nothing is copied or waited for, the caller reads dims_dev when it wants the shapes
    // (a stream capture into a hipGraph, tools/graph_replay_probe.py, or an enqueue-ahead scheduler)
    if (!dims_host) return 0;
    // the ONE host read of the call: every layer's decision, for the output shapes and the layers' pruning records
    he = hipMemcpyAsync(dims_host, dims_dev, nrec * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) return (int)he;
    for (int l = 0; l < n_layers; ++l) {
        io[l].k_out = dims_host[l * DIMS_STRIDE + 1];
        io[l].k_used = dims_host[l * DIMS_STRIDE + 2];
        io[l].n_out = dims_host[l * DIMS_STRIDE + 3];
    }
    const int* rf = madtp_internal_range_flag();
    return (rf && *(const volatile int*)rf) ? MADTP_E_RANGE : 0;
}

// BertEncoder.forward's layer loop WITHOUT the per-layer host read of k (the text side of
// madtp_vit_encoder_async).  Same device-side record per layer (dims[l] = {L_l, k, k applied under med.py:374-375, L_{l+1}}, written by
// token_score's last workgroup); every later kernel of the stream reads its size from it: the output projection + LayerNorm, top-k
// select, the gather (a copy when the layer is not pruned), the compaction of the additive padding mask (med.py:388-390 /
// nlvr_encoder.py:451-452), the compute-dtype copy, the cross-attention q projection and the attention's QUERY count (keys are the
// image tokens: host-side), the FFN GEMMs and LayerNorms, and the next layer's alignment logits / qkv GEMM / masked self-attention.
// Grids and buffers are the unpruned sequence's.  Linear + LayerNorm pairs run as GEMM (f32, residual in the epilogue) + LayerNorm:
// the split-K factor of the host path depends on the row count, which the host does not know here - so this path has its own
// (equally f32-accurate) summation order and is pinned against the reference fixtures, not bit-for-bit against the host-k path.
// Takes: text mode, MED single cross-attention, NLVR twin cross-attention; B * L0 < 4096 rows, L0 <= 256, <= 256 keys.
static int to_lp_dev(const float* src, void* dst, int rows_max, int dim, int dt, DevN rows, void* stream) {
    if (dt == MADTP_F16S) return madtp_i_split_f16(src, dim, dst, 2 * dim, rows_max, dim, rows, stream);
    return madtp_i_cast_lp(src, dst, (size_t)rows_max * dim, dt, 1.0f, DevN{rows.p, rows.mul * dim, 0}, stream);
}
// y = LayerNorm(scale * (a @ W^T + b) + residual) with M = m rows: f32 GEMM into `part`, then the row kernel
static int lin_ln_dev(const void* a, int lda, const madtp_lin& L, const float* residual, float scale, const float* gamma, const float* beta,
                      float* y32, void* ylp, int M_max, int dt, float eps, float* part, DevN m, void* stream) {
    TRY(madtp_i_gemm(a, L.w, L.b, residual, part, M_max, L.n, L.k, pld(dt, lda), (dt == MADTP_F16S ? 2 : 1) * L.k, L.n, L.n, dt, MADTP_F32,
                     MADTP_ACT_NONE, L.w_scale, scale, m, stream));
    return madtp_i_layernorm(part, gamma, beta, y32, ylp, dt == MADTP_F32 ? MADTP_BF16 : dt, M_max, L.n, eps, m, stream);
}

extern "C" int madtp_bert_encoder_async(const madtp_bert_layer_w* const* layers, int n_layers, const madtp_query_w* q, const float* hidden0,
                                        const void* hidden0_lp, const float* mask0, madtp_layer_io* io, void* ws, size_t ws_bytes, int B,
                                        int L0, int Nk, float temperature, int cross_mode, const void* enc0, const void* enc1,
                                        const float* enc_mask0, const float* enc_mask1, const void* const* kv_pre0,
                                        const void* const* kv_pre1, const int32_t* kv_index, int kv_ld, int32_t* dims_dev,
                                        int32_t* dims_host, void* stream) {
    if (!layers || !io || !hidden0 || !q || !mask0 || !dims_dev || !dims_host || n_layers <= 0 || B <= 0 || L0 < 3) return MADTP_E_BADARG;
    if (!(temperature > 0.f) || q->att_ft) return MADTP_E_BADARG;  // (att_ft: the caller's deferred sum, after this call)
    if ((long)B * L0 >= 4096 || L0 > 256 || (cross_mode && Nk > 256)) return MADTP_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const size_t nrec = (size_t)(n_layers + 1) * DIMS_STRIDE;
    int32_t* ticket = dims_dev + nrec;
    hipError_t he = hipMemsetAsync(dims_dev, 0, (nrec + 4) * sizeof(int32_t), s);
    if (he == hipSuccess) he = hipMemsetD32Async((hipDeviceptr_t)dims_dev, L0, 1, s);
    if (he != hipSuccess) return (int)he;
    const int kp = (q->K + 127) / 128 * 128;
    const bool split = q->sd_hi && q->sd_lo;
    if (split && kp != 128) return MADTP_E_SHAPE;
    const float* h = hidden0;
    const void* h_lp = hidden0_lp;
    const float* mask = mask0;
    const int Mmax = B * L0;
    for (int l = 0; l < n_layers; ++l) {
        madtp_layer_io& o = io[l];
        const madtp_bert_layer_w* w = layers[l];
        if (!w || !o.x_attn || !o.y || !o.logits || !o.score || !o.threshold || !o.count || !o.indices || !o.indices_sort || !o.mask_out)
            return MADTP_E_BADARG;
        if (w->self_mask_qk) return MADTP_E_SHAPE;  // decoder layers keep the per-layer path
        const int cross = (cross_mode && w->cross) ? w->cross : 0;
        bool ok;
        BertWs v = bert_carve((char*)ws, ws_bytes, B, L0, Nk, w->dim, w->inter.n, w->heads, w->dtype, &ok);
        if (!ok) return MADTP_E_SHAPE;
        const int D = w->dim, dt = w->dtype, adt = attn_dt(dt);
        const size_t e = esz_of(adt);
        const bool lpm = dt != MADTP_F32;
        if (lpm && !o.y_lp) return MADTP_E_BADARG;
        int32_t* dl = dims_dev + (size_t)l * DIMS_STRIDE;
        const DevN m_in{dl, B, 0}, m_out{dl + DIMS_STRIDE, B, 0};
        // query model on all rows of the hidden states (med.py:513-524, models/utils.py:170)
        if (split) TRY(madtp_i_align_logits(h, q->sd_hi, q->sd_lo, o.logits, Mmax, D, q->split_dtype, q->sd_scale, m_in, stream));
        else TRY(madtp_i_gemm(h, q->sd_w, nullptr, nullptr, o.logits, Mmax, kp, D, D, D, kp, 0, MADTP_F32, MADTP_F32, MADTP_ACT_NONE, 1.f,
                              1.f, m_in, stream));
        // ---- self-attention half (bert_attn_impl) ----
        const void* hc = h;
        if (lpm) {
            if (h_lp) hc = h_lp;
            else { TRY(to_lp_dev(h, v.hc, Mmax, D, dt, m_in, stream)); hc = v.hc; }
        }
        TRY(lin_dev(hc, D, w->qkv, nullptr, 0, v.qkv, 3 * D, Mmax, dt, adt, MADTP_ACT_NONE, m_in, stream));
        const char* qp = (const char*)v.qkv;
        TRY(madtp_i_attention_mask(qp, qp + (size_t)D * e, qp + (size_t)2 * D * e, v.ctx, mask, v.colsum, v.p0, v.onorm, B, w->heads, L0,
                                   3 * D, 3 * D, 3 * D, D, w->scale, attn_io(dt), dl, stream));
        const void* ctx = v.ctx;
        if (dt == MADTP_F16S) {
            TRY(madtp_i_split_f16((const float*)v.ctx, D, v.q, 2 * D, Mmax, D, m_in, stream));
            ctx = v.q;
        }
        TRY(madtp_i_token_score_dev(v.colsum, v.p0, v.onorm, o.logits, kp, q->K, temperature, o.score, o.threshold, o.count, B, w->heads,
                                    L0, dl, ticket, stream));
        // att = LayerNorm(dense(ctx) + hidden): still L_l tokens per sample
        TRY(lin_ln_dev(ctx, D, w->attn_out, h, 1.f, w->ln_att_g, w->ln_att_b, o.x_attn, nullptr, Mmax, dt, w->eps, v.part, m_in, stream));
        // ---- pruning of att and of the mask (bert_rest_impl); an unpruned layer copies both ----
        TRY(madtp_i_token_select_dev(o.score, o.indices, o.indices_sort, v.dst_pos, v.merge_w, B, L0 - 1, dl, stream));
        TRY(madtp_i_token_gather_ln_dev(o.x_attn, v.dst_pos, v.merge_w, v.xp, B, L0, D, nullptr, nullptr, 0.f, nullptr, nullptr, MADTP_BF16,
                                        dl, stream));
        TRY(madtp_i_mask_gather_dev(mask, o.indices, o.indices_sort, w->variant_nlvr, o.mask_out, B, dl, stream));
        const float* a32 = v.xp;
        const void* ac = a32;
        if (lpm) { TRY(to_lp_dev(a32, v.attc, Mmax, D, dt, m_out, stream)); ac = v.attc; }
        // ---- cross-attention to the image tokens ----
        if (cross == 2 && !w->fused_twin) {
            // twin branches with separate projections (the parity modes' layers >= 6: merge_layer is not folded into the dense
            // weights there) - the per-branch sequence of bert_rest_impl with device-side row counts
            void* cbuf[2] = {v.c0, v.c1};
            for (int br = 0; br < 2; ++br) {
                const void* enc = br ? enc1 : enc0;
                const void* const* pre = br ? kv_pre1 : kv_pre0;
                const float* em = w->variant_nlvr ? (br ? enc_mask1 : enc_mask0) : nullptr;
                TRY(lin_dev(ac, D, w->cq[br], nullptr, 0, v.q, D, Mmax, dt, adt, MADTP_ACT_NONE, m_out, stream));
                const void* kv = pre ? pre[l] : nullptr;
                int ldkv = 2 * D;
                if (!kv) {
                    if (!enc) return MADTP_E_BADARG;
                    TRY(lin(enc, D, w->ckv[br], nullptr, 0, v.kv, 2 * D, B * Nk, dt, adt, MADTP_ACT_NONE, 1.f, stream));
                    kv = v.kv;
                } else if (kv_ld) ldkv = kv_ld;
                TRY(madtp_i_attention_cross(v.q, kv, (const char*)kv + (size_t)D * e, (pre && pre[l]) ? kv_index : nullptr, cbuf[br], em, B,
                                            w->heads, L0, Nk, D, ldkv, ldkv, D, w->scale, attn_io(dt), dl + DIMS_STRIDE, stream));
                if (dt == MADTP_F16S) {
                    if (w->inter.n < 2 * D) return MADTP_E_SHAPE;
                    void* sp = (char*)v.mid + (size_t)br * Mmax * D * esz_of(dt);
                    TRY(madtp_i_split_f16((const float*)cbuf[br], D, sp, 2 * D, Mmax, D, m_out, stream));
                    cbuf[br] = sp;
                }
            }
            if (w->has_merge) {  // nlvr_encoder.py:263-264
                const int cdt = dt == MADTP_F16S ? MADTP_F32 : dt;
                char* cat = (char*)v.cat;
                TRY(lin_dev(cbuf[0], D, w->cdense[0], nullptr, 0, cat, 2 * D, Mmax, dt, cdt, MADTP_ACT_NONE, m_out, stream));
                TRY(lin_dev(cbuf[1], D, w->cdense[1], nullptr, 0, cat + (size_t)D * esz_of(cdt), 2 * D, Mmax, dt, cdt, MADTP_ACT_NONE, m_out, stream));
                const void* catc = v.cat;
                if (dt == MADTP_F16S) {
                    TRY(madtp_i_split_f16((const float*)v.cat, 2 * D, v.mid, 4 * D, Mmax, 2 * D, m_out, stream));
                    catc = v.mid;
                }
                TRY(lin_dev(catc, 2 * D, w->merge, a32, D, v.s, D, Mmax, dt, MADTP_F32, MADTP_ACT_NONE, m_out, stream));
            } else {  // :266 (h0 + h1) / 2 folded into the epilogues
                TRY(madtp_i_gemm(cbuf[0], w->cdense[0].w, w->cdense[0].b, a32, v.t, Mmax, D, w->cdense[0].k, pld(dt, D),
                                 (dt == MADTP_F16S ? 2 : 1) * w->cdense[0].k, D, D, dt, MADTP_F32, MADTP_ACT_NONE, w->cdense[0].w_scale, 0.5f,
                                 m_out, stream));
                TRY(madtp_i_gemm(cbuf[1], w->cdense[1].w, w->cdense[1].b, v.t, v.s, Mmax, D, w->cdense[1].k, pld(dt, D),
                                 (dt == MADTP_F16S ? 2 : 1) * w->cdense[1].k, D, D, dt, MADTP_F32, MADTP_ACT_NONE, w->cdense[1].w_scale, 0.5f,
                                 m_out, stream));
            }
            TRY(madtp_i_layernorm(v.s, w->ln_cross_g, w->ln_cross_b, v.att2, lpm ? v.attc : nullptr, dt == MADTP_F32 ? MADTP_BF16 : dt, Mmax,
                                  D, w->eps, m_out, stream));
            a32 = v.att2;
            ac = lpm ? (const void*)v.attc : (const void*)v.att2;
        } else if (cross == 2) {
            TRY(lin_dev(ac, D, w->cq_fused, nullptr, 0, v.q2, 2 * D, Mmax, dt, adt, MADTP_ACT_NONE, m_out, stream));
            const void* k0 = kv_pre0 ? kv_pre0[l] : nullptr;
            const void* k1 = kv_pre1 ? kv_pre1[l] : nullptr;
            if ((!k0) != (!k1) || (!k0 && (!enc0 || !enc1))) return MADTP_E_BADARG;
            int ldkv = 2 * D;
            if (!k0) {
                if (w->ckv[0].n == w->ckv[1].n && w->ckv[0].k == w->ckv[1].k)
                    TRY(madtp_gemm_pair(enc0, enc1, w->ckv[0].w, w->ckv[1].w, w->ckv[0].b, w->ckv[1].b, v.kv, v.kv1, B * Nk, w->ckv[0].n,
                                        w->ckv[0].k, pld(dt, D), (dt == MADTP_F16S ? 2 : 1) * w->ckv[0].k, 2 * D, dt, adt, w->ckv[0].w_scale,
                                        w->ckv[1].w_scale, stream));
                else {
                    TRY(lin(enc0, D, w->ckv[0], nullptr, 0, v.kv, 2 * D, B * Nk, dt, adt, MADTP_ACT_NONE, 1.f, stream));
                    TRY(lin(enc1, D, w->ckv[1], nullptr, 0, v.kv1, 2 * D, B * Nk, dt, adt, MADTP_ACT_NONE, 1.f, stream));
                }
                k0 = v.kv; k1 = v.kv1;
            } else if (kv_ld) ldkv = kv_ld;
            const float* em0 = w->variant_nlvr ? enc_mask0 : nullptr;
            const float* em1 = w->variant_nlvr ? enc_mask1 : nullptr;
            TRY(madtp_i_attention_pair(v.q2, (const char*)v.q2 + (size_t)D * e, k0, k1, (const char*)k0 + (size_t)D * e,
                                       (const char*)k1 + (size_t)D * e, (kv_pre0 && kv_pre0[l]) ? kv_index : nullptr, v.cat,
                                       (char*)v.cat + (size_t)D * e, em0, em1, B, w->heads, L0, Nk, 2 * D, ldkv, ldkv, 2 * D, w->scale,
                                       attn_io(dt), dl + DIMS_STRIDE, stream));
            const void* catc = v.cat;
            if (dt == MADTP_F16S) {
                TRY(madtp_i_split_f16((const float*)v.cat, 2 * D, v.mid, 4 * D, Mmax, 2 * D, m_out, stream));
                catc = v.mid;
            }
            TRY(lin_ln_dev(catc, 2 * D, w->cdense_fused, a32, w->fused_twin == 2 ? 1.f : 0.5f, w->ln_cross_g, w->ln_cross_b, v.att2,
                           lpm ? v.attc : nullptr, Mmax, dt, w->eps, v.part, m_out, stream));
            a32 = v.att2;
            ac = lpm ? (const void*)v.attc : (const void*)v.att2;
        } else if (cross == 1) {
            TRY(lin_dev(ac, D, w->cq[0], nullptr, 0, v.q, D, Mmax, dt, adt, MADTP_ACT_NONE, m_out, stream));
            const void* k0 = kv_pre0 ? kv_pre0[l] : nullptr;
            int ldkv = 2 * D;
            if (!k0) {
                if (!enc0) return MADTP_E_BADARG;
                TRY(lin(enc0, D, w->ckv[0], nullptr, 0, v.kv, 2 * D, B * Nk, dt, adt, MADTP_ACT_NONE, 1.f, stream));
                k0 = v.kv;
            } else if (kv_ld) ldkv = kv_ld;
            // (med.py:197-199 drops the encoder mask in cross-attention; nlvr_encoder.py:196-198 applies it)
            TRY(madtp_i_attention_cross(v.q, k0, (const char*)k0 + (size_t)D * e, (kv_pre0 && kv_pre0[l]) ? kv_index : nullptr, v.c0,
                                        w->variant_nlvr ? enc_mask0 : nullptr, B, w->heads, L0, Nk, D, ldkv, ldkv, D, w->scale, attn_io(dt),
                                        dl + DIMS_STRIDE, stream));
            const void* cc = v.c0;
            if (dt == MADTP_F16S) {
                TRY(madtp_i_split_f16((const float*)v.c0, D, v.mid, 2 * D, Mmax, D, m_out, stream));
                cc = v.mid;
            }
            TRY(lin_ln_dev(cc, D, w->cdense[0], a32, 1.f, w->ln_cross_g, w->ln_cross_b, v.att2, lpm ? v.attc : nullptr, Mmax, dt, w->eps,
                           v.part, m_out, stream));
            a32 = v.att2;
            ac = lpm ? (const void*)v.attc : (const void*)v.att2;
        }
        // ---- FFN ----
        TRY(lin_dev(ac, D, w->inter, nullptr, 0, v.mid, w->inter.n, Mmax, dt, dt, MADTP_ACT_GELU_ERF, m_out, stream));
        TRY(lin_ln_dev(v.mid, w->inter.n, w->out, a32, 1.f, w->ln_out_g, w->ln_out_b, o.y, lpm ? o.y_lp : nullptr, Mmax, dt, w->eps, v.part,
                       m_out, stream));
        h = o.y;
        h_lp = lpm ? o.y_lp : nullptr;
        mask = o.mask_out;
    }
    he = hipMemcpyAsync(dims_host, dims_dev, nrec * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) return (int)he;
    for (int l = 0; l < n_layers; ++l) {
        io[l].k_out = dims_host[l * DIMS_STRIDE + 1];
        io[l].k_used = dims_host[l * DIMS_STRIDE + 2];
        io[l].n_out = dims_host[l * DIMS_STRIDE + 3];
    }
    const int* rf = madtp_internal_range_flag();
    return (rf && *(const volatile int*)rf) ? MADTP_E_RANGE : 0;
}

// ---- cross-attention K/V of the image tokens on a SIDE stream (round 5) ------------------------------------------------------
// The twin cross-attention layers project the two images' tokens to [k|v] with one pair GEMM per layer (12 x ~31 us at the
// headline batch: the only chip-filling launches of the text encoder, whose other ~150 kernels are latency-bound chains on 1280
// rows).  Those projections depend on the vision encoder's output only, so the encoder-level call enqueues all of them on a side
// stream of its own (one per (device, caller stream); event-ordered behind the caller's stream on entry, one completion event
// per layer) into a library-owned buffer, and layer l of the caller's stream waits for event l and reads its K/V from there
// (the kv_pre path of madtp_bert_layer: same kernel, same operands, same bits).
// OFF by default (MADTP_KV_SIDE=1 turns it on): measured at the headline batch (profiles/r05_kv_side_ab.txt, same box, two rounds)
// the serial forward gains 2 % with the side GEMMs on half the chip (19.05 -> 19.43 k images/s; nothing with a chip-filling side
// GEMM - the persistent 256-row tiles leave the text kernels no CU to start on - and the per-layer Python path is still ahead at
// 19.68 k), while four forwards in flight LOSE 4 % (26.3 -> 25.1 k: eight streams contend kernel by kernel).
namespace {
constexpr int KV_SIDE_MAX_LAYERS = 32;
struct KvSide {
    hipStream_t side = nullptr;
    hipEvent_t ready = nullptr, done[KV_SIDE_MAX_LAYERS] = {};
    char* buf = nullptr;
    size_t cap = 0;
};
std::mutex g_kv_side_mu;
std::map<std::pair<int, hipStream_t>, KvSide> g_kv_side;

bool kv_side_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MADTP_KV_SIDE"); v = e ? atoi(e) : 0; }
    return v != 0;
}

// -> nullptr when the side resources cannot be had (the caller falls back to the in-layer projections)
KvSide* kv_side_get(hipStream_t main, size_t bytes, int n_layers) {
    int dev = 0;
    if (n_layers > KV_SIDE_MAX_LAYERS || hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_kv_side_mu);
    KvSide& k = g_kv_side[{dev, main}];
    if (!k.side) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&k.side, hipStreamNonBlocking, lo) != hipSuccess) { (void)hipGetLastError(); k.side = nullptr; return nullptr; }
        bool ok = hipEventCreateWithFlags(&k.ready, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < KV_SIDE_MAX_LAYERS; ++i) ok = hipEventCreateWithFlags(&k.done[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); return nullptr; }
    }
    if (k.cap < bytes) {
        // grow-only; the old buffer may still be read by kernels in flight on the caller's stream: free it behind them
        if (k.buf) { (void)hipStreamSynchronize(main); (void)hipStreamSynchronize(k.side); (void)hipFree(k.buf); k.buf = nullptr; k.cap = 0; }
        if (hipMalloc((void**)&k.buf, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        k.cap = bytes;
    }
    return &k;
}
}  // namespace

extern "C" int madtp_bert_encoder(const madtp_bert_layer_w* const* layers, int n_layers, const madtp_query_w* q, const float* hidden0,
                                  const void* hidden0_lp, const float* mask0, madtp_layer_io* io, void* ws, size_t ws_bytes, int B,
                                  int L0, int Nk, float temperature, int cross_mode, const void* enc0, const void* enc1,
                                  const float* enc_mask0, const float* enc_mask1, const void* const* kv_pre0,
                                  const void* const* kv_pre1, const int32_t* kv_index, int kv_ld, void* stream) {
    if (!layers || !io || !hidden0 || n_layers <= 0 || B <= 0 || L0 <= 0) return MADTP_E_BADARG;
    const float* h = hidden0;
    const void* h_lp = hidden0_lp;
    const float* mask = mask0;
    int L = L0;
    const bool prune = q && temperature > 0.f;
    const int kp = q ? (q->K + 127) / 128 * 128 : 0;
    // twin cross-attention without a caller-side K/V cache: all layers' [k|v] projections of the image tokens go to the side stream
    KvSide* side = nullptr;
    size_t kv_bytes = 0;
    if (cross_mode && enc0 && enc1 && !kv_pre0 && !kv_pre1 && Nk > 0 && kv_side_enabled()) {
        bool twin = true;
        for (int l = 0; l < n_layers && twin; ++l) {
            const madtp_bert_layer_w* w = layers[l];
            twin = w && w->cross == 2 && w->fused_twin && w->ckv[0].n == w->ckv[1].n && w->ckv[0].k == w->ckv[1].k &&
                   w->ckv[0].n == 2 * w->dim && w->dtype == layers[0]->dtype && w->dim == layers[0]->dim;
        }
        if (twin) {
            const int D = layers[0]->dim, dt = layers[0]->dtype, adt = attn_dt(dt);
            kv_bytes = (((size_t)B * Nk * 2 * D * esz_of(adt)) + 255) & ~(size_t)255;
            side = kv_side_get((hipStream_t)stream, 2 * kv_bytes * n_layers, n_layers);
            if (side) {
                hipError_t he = hipEventRecord(side->ready, (hipStream_t)stream);  // enc0 / enc1 (and the previous call's readers) are behind it
                if (he == hipSuccess) he = hipStreamWaitEvent(side->side, side->ready, 0);
                if (he != hipSuccess) return (int)he;
                // the projections are off the critical path (layer l needs its pair ~150 l us into the encoder): they run on HALF the
                // CUs (MADTP_KV_SIDE_CAP workgroups per XCD, default 16) - a chip-filling persistent GEMM would leave the main
                // stream's small kernels no CU to start on for its whole duration
                static int side_cap = -1;
                if (side_cap < 0) { const char* e = getenv("MADTP_KV_SIDE_CAP"); side_cap = e ? atoi(e) : 16; if (side_cap < 1 || side_cap > 32) side_cap = 16; }
                struct CapGuard { int prev; explicit CapGuard(int c) : prev(madtp_internal_gemm_wg_cap(c)) {} ~CapGuard() { madtp_internal_gemm_wg_cap(prev); } } cap_guard(side_cap);
                for (int l = 0; l < n_layers; ++l) {
                    const madtp_bert_layer_w* w = layers[l];
                    char* k0 = side->buf + (size_t)(2 * l) * kv_bytes;
                    TRY(madtp_gemm_pair(enc0, enc1, w->ckv[0].w, w->ckv[1].w, w->ckv[0].b, w->ckv[1].b, k0, k0 + kv_bytes, B * Nk,
                                        w->ckv[0].n, w->ckv[0].k, pld(dt, D), (dt == MADTP_F16S ? 2 : 1) * w->ckv[0].k, 2 * D, dt, adt,
                                        w->ckv[0].w_scale, w->ckv[1].w_scale, side->side));
                    he = hipEventRecord(side->done[l], side->side);
                    if (he != hipSuccess) return (int)he;
                }
            }
        }
    }
    for (int l = 0; l < n_layers; ++l) {
        madtp_layer_io& o = io[l];
        const madtp_bert_layer_w* w = layers[l];
        if (!w || !o.x_attn || !o.y) return MADTP_E_BADARG;
        if (q) {
            if (!o.logits) return MADTP_E_BADARG;
            TRY(query_step(q, h, o.logits, l, B, L, w->dim, stream));
        }
        int k_out = 0, k_used = 0;
        const void* kv0 = kv_pre0 ? kv_pre0[l] : nullptr;
        const void* kv1 = kv_pre1 ? kv_pre1[l] : nullptr;
        if (side) {
            // (the wait sits in front of the layer's first kernel: only layer 0 can actually stall on it - the side stream is
            //  ~380 us of GEMM in all, layer l starts ~150 l us into the encoder)
            const hipError_t he = hipStreamWaitEvent((hipStream_t)stream, side->done[l], 0);
            if (he != hipSuccess) return (int)he;
            kv0 = side->buf + (size_t)(2 * l) * kv_bytes;
            kv1 = (const char*)kv0 + kv_bytes;
        }
        if (prune)
            TRY(madtp_bert_layer(w, h, mask, o.x_attn, o.y, o.mask_out, ws, ws_bytes, B, L, Nk, o.logits + kp, kp, L * kp, q->K,
                                 temperature, o.score, o.threshold, o.count, o.indices, o.indices_sort, cross_mode, enc0, enc1,
                                 enc_mask0, enc_mask1, h_lp, o.y_lp, kv0, kv1, kv_index, kv_ld, &k_out, &k_used, stream));
        else
            TRY(madtp_bert_layer(w, h, mask, o.x_attn, o.y, nullptr, ws, ws_bytes, B, L, Nk, nullptr, 0, 0, 0, 0.f, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, cross_mode, enc0, enc1, enc_mask0, enc_mask1, h_lp, o.y_lp, kv0, kv1,
                                 kv_index, kv_ld, &k_out, &k_used, stream));
        if (k_used > 0) {
            L = k_used + 2;
            if (mask) mask = o.mask_out;
        }
        o.k_out = k_out; o.k_used = k_used; o.n_out = L;
        h = o.y;
        h_lp = o.y_lp;
    }
    return 0;
}
