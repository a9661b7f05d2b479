/* madtp_hip.h - C-ABI of the MI355X (gfx950) kernels behind MADTP's pruned vision-language forward path.
 *
 * The reference (double125/MADTP) has no FFI/plugin layer of its own: its hot path is a chain of eager aten ops
 * issued from Python nn.Modules.  Each entry point below therefore names the reference code
 * span (file:line under the reference root) whose op sequence it replaces.  Conventions:
 *   - plain pointers + sizes only, no torch types; every pointer is DEVICE memory owned by the caller;
 *   - nothing allocates, frees or synchronises; work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, a negative MADTP_E_* code on a rejected argument, a positive hipError_t if the
 *     launch itself failed;
 *   - matrices are row-major; `dtype` arguments use MADTP_F32 / MADTP_BF16.
 */
#ifndef MADTP_HIP_H
#define MADTP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MADTP_F32 0
#define MADTP_BF16 1

#define MADTP_E_BADARG (-1)   /* null pointer / non-positive size                      */
#define MADTP_E_SHAPE (-2)    /* shape outside what the kernel family supports          */
#define MADTP_E_DTYPE (-3)    /* unknown dtype code                                     */
#define MADTP_E_ALIGN (-4)    /* pointer or leading dimension not 16-byte aligned       */

/* activation codes of the GEMM epilogue */
#define MADTP_ACT_NONE 0
#define MADTP_ACT_GELU_ERF 1   /* nn.GELU / ACT2FN['gelu']: vit.py:34, med.py:313        */
#define MADTP_ACT_QUICK_GELU 2 /* x*sigmoid(1.702x): clip/model.py:169-171               */
#define MADTP_ACT_RELU 3       /* cls_head: blip_nlvr.py:59                              */

/* Library/version probe; returns the ABI version (increments on any signature change). */
int madtp_abi_version(void);
/* Human-readable name of a negative MADTP_E_* code. */
const char* madtp_strerror(int code);

/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias[N]) * out_scale (+ residual[M,N])
 * Replaces every nn.Linear on the path (aten::addmm): vit.py:31-35,77,92; med.py:153-171,246-250,312-329;
 * nlvr_encoder.py:259-266; models/utils.py:170 (x @ space_dict^T); blip_nlvr.py:57-61.
 * ab_dtype: dtype of A and W (F32 -> exact-f32 MFMA 16x16x4; BF16 -> MFMA 16x16x32 with f32 accumulate).
 * W must be padded by the caller to a multiple of 128 rows (zero rows) - n_pad rows are read, N columns stored.
 * bias (f32, may be NULL), residual (f32 [M,ldr], may be NULL), C dtype c_dtype with leading dimension ldc.
 * K must be a multiple of 64 (bf16) / 32 (f32); lda, ldw in elements. */
int madtp_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C,
               int M, int N, int K, int lda, int ldw, int ldc, int ldr,
               int ab_dtype, int c_dtype, int act, float out_scale, void* stream);

/* y = LayerNorm(x) * gamma + beta over the last dim (dim % 4 == 0, dim <= 1024); x is f32.
 * Writes y32 (f32, may be NULL) and/or ylp (bf16, may be NULL).
 * vit.py:186,205,309 (eps 1e-6); med.py:79,249,328 (eps 1e-12); clip/model.py:160-166 (eps 1e-5). */
int madtp_layernorm(const float* x, const float* gamma, const float* beta, float* y32, void* ylp,
                    int rows, int dim, float eps, void* stream);

/* im2col of non-overlapping patches for the patch-embedding GEMM (timm PatchEmbed Conv2d k=s=P, call site
 * vit.py:241-242,283): img f32 [B,3,S,S] -> cols [B*(S/P)^2, 3*P*P] (dtype out_dtype), column = c*P*P+ky*P+kx. */
int madtp_patchify(const float* img, void* cols, int B, int S, int P, int out_dtype, void* stream);

/* x[b,0,:] = cls + pos[0]; x[b,1+p,:] = patches[b*np+p,:] + pos[1+p]   (vit.py:285-289). All f32. */
int madtp_assemble_tokens(const float* patches, const float* cls, const float* pos, float* x,
                          int B, int np, int dim, void* stream);

/* BERT embeddings: y = LayerNorm(word_emb[ids] + pos_emb[0..L))  (med.py:63-86 / nlvr_encoder.py:62-86).
 * ids int64 [B,L]; writes y32 (f32) and/or ylp (bf16). */
int madtp_bert_embed(const int64_t* ids, const float* word_emb, const float* pos_emb, const float* gamma,
                     const float* beta, float* y32, void* ylp, int B, int L, int dim, float eps, void* stream);

/* Multi-head attention core with the pruning-score side outputs.
 * q/k/v point at the first element of head 0 of token 0 for each operand; rows are tokens with row strides
 * ldq/ldk/ldv (elements), head h occupies columns [h*64, h*64+64).  io_dtype is the dtype of q,k,v and out.
 * scores = (q k^T) * scale (+ add_mask[b,j], f32 [B,Nk], may be NULL) ; P = softmax_j ; out = P v
 *   -> out[(b*Nq+i), h*64+d]  (ldo)                      vit.py:81-91; med.py:177-222; nlvr_encoder.py:176-223
 * Side outputs (all f32, pass NULL for colsum_part to skip them - cross-attention):
 *   colsum_part[b, rt, j] = sum over query rows i in 16-row tile rt, i>=1, of max_h P[b,h,i,j]   (vit.py:126-127)
 *   p0[b,h,j]   = P[b,h,0,j]                                                                      (vit.py:96)
 *   onorm[b,h,i]= || out[b,h,i,:] ||_2                                                            (vit.py:97)
 * Limits: head_dim 64; Nk <= 256 in this family. */
int madtp_attention(const void* q, const void* k, const void* v, void* out, const float* add_mask,
                    float* colsum_part, float* p0, float* onorm,
                    int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo,
                    float scale, int io_dtype, void* stream);

/* Alignment-guided token-importance score, per-sample threshold and survivor count
 * (Block.Reduce_token vit.py:125-145 == med.py:347-371 == nlvr_encoder.py:404-432 == clip/model.py:196-218).
 * n = N-1 patch tokens.  token_attn f32: element [b,t,c] at token_attn[b*ldt_batch + t*ldt_row + c], c < K (raw
 * x.sd^T logits of patch token t; any strided [B,n,K] view with unit column stride).
 * Outputs: score f32 [B,n]; threshold f32 [B]; count int32 [B]; kmax int32[1] = max_b count (must be zeroed by
 * the caller before the launch). */
int madtp_token_score(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                      const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature,
                      float* score, float* threshold, int32_t* count, int32_t* kmax,
                      int B, int H, int N, void* stream);

/* Top-k selection by rank + merge weights (vit.py:153-159).  For each sample: rank tokens by score (descending,
 * ties -> lower index first); kept = rank < k, emitted in ascending token order.
 *   indices      int64 [B,k]   kept token ids (the reference's `indices`, order implementation-defined there)
 *   indices_sort int64 [B,n]   full descending order (the reference's `indices_sort`)
 *   dst_pos      int32 [B,n]   output slot of a kept token, -1 for a dropped one
 *   merge_w      f32   [B,n]   score/(sum_dropped score + 1e-8) for dropped tokens, 0 for kept */
int madtp_token_select(const float* score, int k, int64_t* indices, int64_t* indices_sort, int32_t* dst_pos,
                       float* merge_w, int B, int n, void* stream);

/* Gather/compact + merge (vector_gather models/utils.py:13-33; vit.py:154-161,195-202):
 *   y[b,0,:] = x[b,0,:]; y[b,1+dst_pos[b,t],:] = x[b,1+t,:] for kept t; y[b,k+1,:] = sum_dropped merge_w*x[b,1+t,:]
 * x f32 [B,N,dim] -> y f32 [B,k+2,dim]. */
int madtp_token_gather(const float* x, const int32_t* dst_pos, const float* merge_w, float* y,
                       int B, int N, int k, int dim, void* stream);

/* Additive-mask compaction for the text encoders (nlvr_encoder.py:451-452,531-533; med.py:388-390,429-440):
 * out[b,0]=mask[b,0]; out[b,1+p] = mask[b,1+order[b,p]] for p in [0,k].  order = indices_sort (NLVR) . */
int madtp_mask_gather(const float* mask, const int64_t* order, int ld_order, float* out, int B, int N, int k,
                      void* stream);

/* Query_model's att_ft (models/utils.py:174-178): att_ft[b,c,:] (+)= sum_t softmax_t(token_attn[b,t,c]/sqrt(dim_sd)) * x[b,1+t,:]
 * token_attn as in madtp_token_score; ft f32: patch token t of sample b at ft[b*ldf_batch + t*ldf_row + d] (so
 * x[:,1:,:] of a [B,N,dim] tensor is passed without a copy); n patch tokens; out f32 [B,K,dim] contiguous;
 * accumulate!=0 adds into out (sd_img_ft_all += sd_img_ft, vit.py:300-303). */
int madtp_query_att_ft(const float* token_attn, int ldt_row, int ldt_batch, int K, const float* ft, int ldf_row,
                       int ldf_batch, float* out, float inv_sqrt_sd, int accumulate, int B, int n, int dim,
                       void* stream);

/* vector_gather (models/utils.py:13-33): out[b,k,:] = vectors[b, indices[b,k], :]; f32 [B,L,D], int64 [B,K]. */
int madtp_vector_gather(const float* vectors, const int64_t* indices, float* out, int B, int L, int K, int D,
                        void* stream);

/* (a+b)*scale elementwise, f32 (nlvr_encoder.py:266 average of the two cross-attention branches). */
int madtp_add_scale(const float* a, const float* b, float* out, float scale, size_t n, void* stream);

/* f32 -> bf16 copy (weight preparation, activations entering a bf16 GEMM). */
int madtp_cast_bf16(const float* src, void* dst, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MADTP_HIP_H */
