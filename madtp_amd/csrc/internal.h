// Library-internal launchers shared between the translation units (hidden visibility: not part of the C-ABI).  The *_i_* forms are
// the public entry points plus a DevN (common.h): a size read from device memory by the kernel itself - the sync-free encoder path.
#pragma once
#include "common.h"

#define MADTP_INTERNAL __attribute__((visibility("hidden")))

MADTP_INTERNAL int madtp_i_layernorm(const float* x, const float* gamma, const float* beta, float* y32, void* ylp, int lp_dtype,
                                     int rows, int dim, float eps, DevN rows_dev, void* stream);
MADTP_INTERNAL int madtp_i_split_f16(const float* src, int ld_src, void* dst, int ld_dst, int rows, int K, DevN rows_dev, void* stream);
// madtp_gemm with M = m_dev (small-tile kernels only: the worst-case M must stay below 4096 rows)
MADTP_INTERNAL int madtp_i_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K,
                                int lda, int ldw, int ldc, int ldr, int ab_dtype, int c_dtype, int act, float acc_scale,
                                float out_scale, DevN m_dev, void* stream);
// madtp_attention (self-attention with the score side outputs) with Nq = Nk = n_dev tokens per sample
MADTP_INTERNAL int madtp_i_attention(const void* q, const void* k, const void* v, void* out, float* colsum_part, float* p0,
                                     float* onorm, int B, int H, int N, int ldq, int ldk, int ldv, int ldo, float scale,
                                     int io_dtype, const int32_t* n_dev, void* stream);
// token_score on n_dev tokens (token_attn = rows 1.. of a [B * n, ldt] logits buffer: batch stride n * ldt); the last workgroup
// writes the layer's decision to dims_l[1..3] and the next layer's token count to dims_l[DIMS_STRIDE] (BLIP rule vit.py:148-149)
MADTP_INTERNAL int madtp_i_token_score_dev(const float* colsum_part, const float* p0, const float* onorm, const float* logits, int ldt,
                                           int K, float temperature, float* score, float* threshold, int32_t* count, int B, int H,
                                           int N_max, int32_t* dims_l, int32_t* ticket, void* stream);
// token_select / token_gather_ln with n, k (k == 0: identity - nothing is pruned, the gather copies every token) from dims_l
MADTP_INTERNAL int madtp_i_token_select_dev(const float* score, int64_t* indices, int64_t* indices_sort, int32_t* dst_pos,
                                            float* merge_w, int B, int n_max, const int32_t* dims_l, void* stream);
MADTP_INTERNAL int madtp_i_token_gather_ln_dev(const float* x, const int32_t* dst_pos, const float* merge_w, float* y, int B, int N_max,
                                               int dim, const float* gamma, const float* beta, float eps, float* h32, void* h_lp,
                                               int lp_dtype, const int32_t* dims_l, void* stream);
MADTP_INTERNAL int madtp_i_align_logits(const float* x, const void* sd_hi, const void* sd_lo, float* out, int M, int dim, int split_dtype,
                                        float out_scale, DevN m_dev, void* stream);
// Workgroups per XCD of the big-GEMM kernels for the following launches of THIS thread (0 = the process-wide setting); returns the
// previous value.  The encoder call's side-stream K/V projections run on part of the chip so that the latency-bound text kernels
// of the main stream keep finding free CUs.
MADTP_INTERNAL int madtp_internal_gemm_wg_cap(int cap);
// ---- device-side lengths for the text encoders (madtp_bert_encoder_async) ----
MADTP_INTERNAL int madtp_i_cast_lp(const float* src, void* dst, size_t n, int lp_dtype, float scale, DevN n_dev, void* stream);
MADTP_INTERNAL int madtp_i_attention_mask(const void* q, const void* k, const void* v, void* out, const float* add_mask, float* colsum_part,
                                          float* p0, float* onorm, int B, int H, int N, int ldq, int ldk, int ldv, int ldo, float scale,
                                          int io_dtype, const int32_t* n_dev, void* stream);
MADTP_INTERNAL int madtp_i_attention_cross(const void* q, const void* k, const void* v, const int32_t* kv_batch_index, void* out,
                                           const float* add_mask, int B, int H, int Nq_max, int Nk, int ldq, int ldk, int ldv, int ldo,
                                           float scale, int io_dtype, const int32_t* nq_dev, void* stream);
MADTP_INTERNAL int madtp_i_attention_pair(const void* q0, const void* q1, const void* k0, const void* k1, const void* v0, const void* v1,
                                          const int32_t* kv_batch_index, void* out0, void* out1, const float* add_mask0,
                                          const float* add_mask1, int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo,
                                          float scale, int io_dtype, const int32_t* nq_dev, void* stream);
// additive-mask compaction with N, k from dims_l (k == 0: copy); MED: indices then the (k+1)-th of indices_sort, NLVR: indices_sort
MADTP_INTERNAL int madtp_i_mask_gather_dev(const float* mask, const int64_t* indices, const int64_t* indices_sort, int variant_nlvr,
                                           float* out, int B, const int32_t* dims_l, void* stream);
// incremental decoding: Nq queries per sample against the first Nk rows of the sample's K/V block of kv_block_rows rows
MADTP_INTERNAL int madtp_i_attention_cached(const void* q, const void* k, const void* v, int kv_block_rows, void* out, int B, int H, int Nq,
                                            int Nk, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype, void* stream);
// f16x3 layer calls: ask the next attention launches of this thread (io_dtype MADTP_F16S) to write their context as the f16-split
// operand planes of the consuming GEMM (out = _Float16 planes, ldo in f16 elements, P1 at column offset split_dim); 0 = off.
// _done(): 1 once a launch honoured the request.  _f16s_enabled(): the f16s attention kernels are on (MADTP_ATTN_F16S != 0).
MADTP_INTERNAL int madtp_internal_attn_split_out(int split_dim);
MADTP_INTERNAL int madtp_internal_attn_split_done();
MADTP_INTERNAL bool madtp_internal_attn_f16s_enabled();
