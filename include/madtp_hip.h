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
/* f16-split operands of the fp32-accurate GEMM on the f16 MFMA ("f16x3" precision mode): an f32 matrix [R,K] stored as
 * f16 planes side by side in a row - activations [P0 | P1] (2K f16 per row, P0 = f16(x), P1 = f16((x-P0) 2^11)), prepared
 * weights [Q0 | Q1] (2K f16 per row, w 2^s = Q0 + Q1 with the power of two in madtp_lin.w_scale); leading dimensions of such
 * operands count f16 elements.  Written by madtp_split_f16 / the LayerNorm and GEMM epilogues, read by madtp_gemm. */
#define MADTP_F16S 2
/* plain IEEE f16 operands on the f16 MFMA (the "f16" fast precision mode, round 4): the storage layout, shapes and kernels of
 * MADTP_BF16 with 11 significand bits instead of 8 at the same MFMA rate (v_mfma_f32_16x16x32_f16); accepted wherever
 * MADTP_BF16 is, unless an entry point says otherwise.  Range: |x| < 65504 - producers raise the range flag
 * (madtp_range_status) instead of handing on an infinity. */
#define MADTP_F16 3

#define MADTP_E_BADARG (-1)   /* null pointer / non-positive size                      */
#define MADTP_E_SHAPE (-2)    /* shape outside what the kernel family supports          */
#define MADTP_E_DTYPE (-3)    /* unknown dtype code                                     */
#define MADTP_E_ALIGN (-4)    /* pointer or leading dimension not 16-byte aligned       */
#define MADTP_E_BUSY (-5)     /* every host hand-over slot of the device is pending (publish without wait) */
#define MADTP_E_RANGE (-6)    /* a value left the f16 range in an f16 precision mode (madtp_range_status)   */

/* activation codes of the GEMM epilogue */
#define MADTP_ACT_NONE 0
#define MADTP_ACT_GELU_ERF 1   /* nn.GELU / ACT2FN['gelu']: vit.py:34, med.py:313        */
#define MADTP_ACT_QUICK_GELU 2 /* x*sigmoid(1.702x): clip/model.py:169-171               */
#define MADTP_ACT_RELU 3       /* cls_head: blip_nlvr.py:59                              */

/* Library/version probe; returns the ABI version (increments on any signature change). */
int madtp_abi_version(void);
/* Human-readable name of a negative MADTP_E_* code. */
const char* madtp_strerror(int code);

/* C[M,N] = act(acc_scale * (A[M,K] @ W[N,K]^T) + bias[N]) * out_scale (+ residual[M,N])
 * Replaces every nn.Linear on the path (aten::addmm): vit.py:31-35,77,92; med.py:153-171,246-250,312-329;
 * nlvr_encoder.py:259-266; models/utils.py:170 (x @ space_dict^T); blip_nlvr.py:57-61.
 * ab_dtype: dtype of A and W (F32 -> exact-f32 MFMA 16x16x4; BF16 -> MFMA 16x16x32 with f32 accumulate; F16S -> the
 * f16-split operands above: three f16 MFMA products per logical product, fp32-accurate; acc_scale = the weight's 2^-s).
 * W must be padded by the caller to a multiple of 128 rows (zero rows) - n_pad rows are read, N columns stored.
 * bias (f32, may be NULL), residual (f32 [M,ldr], may be NULL), C dtype c_dtype with leading dimension ldc
 * (BF16 output needs BF16 operands, F16S output F16S operands; F32 output is always available).
 * K must be a multiple of 64 (bf16, f16-split) / 32 (f32); lda, ldw in elements (f16 elements for F16S: both >= 2K). */
int madtp_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C,
               int M, int N, int K, int lda, int ldw, int ldc, int ldr,
               int ab_dtype, int c_dtype, int act, float acc_scale, float out_scale, void* stream);

/* Benchmark / test hook: force the tile configuration of the following madtp_gemm launches of this process (0 = automatic
 * dispatch (default), 1..4 = the 128x128 / 64x128 / 64x128x3 / 64x64 kernels, 5 = the wave-specialised 256x128 kernel,
 * 6 = the 256x256 kernel, 7 = the wave-specialised kernel without its stream-K tail; a configuration that cannot take a
 * problem falls back to the automatic choice).  The environment
 * variable MADTP_GEMM_CFG sets the initial value.  No reference counterpart (the reference has one GEMM: aten::addmm).
 * Returns the previous value. */
int madtp_gemm_set_config(int cfg);
/* Scheduling hint for the automatic dispatch: the relative per-round cost of a 256x256 tile against a 256x128 tile (default
 * 1.7 = an isolated launch; a caller that keeps several forwards in flight on the GPU lowers it - the other streams fill
 * sparse last rounds, so the more efficient tile wins more often).  cost <= 0 restores the default (MADTP_GEMM_SQ_COST or
 * 1.7).  Process-wide; results do not depend on it (same arithmetic per output element in both kernels).  Returns the previous
 * value.  No reference counterpart. */
float madtp_gemm_set_sq_cost(float cost);
/* Second scheduling hint of the same kind: the tile configuration of the SMALL problems (fewer than 200 tiles of 256x128, e.g.
 * the 1280-row GEMMs of the text encoders): -1 = automatic (the smallest tile that fits one round: lowest latency of a lone
 * launch), 0 = 128x128, 1 = 64x128, 2 = 64x128 with three stages, 3 = 64x64.  A caller with several forwards in flight sets 0
 * (fewer operand re-reads: less CU time per launch).  Process-wide; results do not depend on it.  Returns the previous value. */
int madtp_gemm_set_small_tile(int cfg);

/* Per-STREAM scheduling attributes (ABI 29; no reference counterpart - the reference runs one batch at a time on the default
 * stream, compress_nlvr_dtp.py:73-99).  A caller that keeps several forwards in flight on one GPU can give each forward its own
 * slice of the chip instead of letting the streams time-slice all 256 CUs kernel by kernel:
 *   madtp_stream_create_cumask  creates a HIP stream restricted to a CU mask (hipExtStreamCreateWithCUMask).  mask = `words`
 *       32-bit words; on MI355X bit i enables CU (i / 8) of XCD (i % 8) (measured: tools/probes/probe_cumask.hip,
 *       profiles/r06_cumask_probe.txt); an XCD whose bits are all zero is NOT excluded - the hardware dispatches workgroups round-robin
 *       over all eight XCDs and treats an empty per-XCD mask as "every CU" - so a partition is "CUs [c0, c0+n) of EVERY XCD".
 *   madtp_stream_set_sched      tells the library what the stream owns, so that launches on it are sized for it:
 *       cus_per_xcd (1..32; 0 = unchanged / the whole chip = 32): the persistent GEMM kernels launch 8 * cus_per_xcd workgroups
 *           and the dispatch rules count rounds over 8 * cus_per_xcd CUs;
 *       sq_cost (> 0) and small_tile (-1..3): the two hints above for launches on THIS stream only (<= 0 / -2 = follow the
 *           process-wide setting) - concurrent callers with different needs do not share state.
 *       Results never depend on these attributes (same arithmetic per output element in every kernel choice).
 *   madtp_stream_destroy        forgets the attributes and destroys a stream made by madtp_stream_create_cumask.
 * Attributes may be set for any stream (also one the caller created); up to 64 streams per process carry attributes. */
int madtp_stream_create_cumask(void** stream_out, const uint32_t* mask, int words);
int madtp_stream_set_sched(void* stream, int cus_per_xcd, float sq_cost, int small_tile);
int madtp_stream_get_sched(void* stream, int* cus_per_xcd, float* sq_cost, int* small_tile);
int madtp_stream_destroy(void* stream);

/* Split-K form for small-M projections (latency-bound at one workgroup per tile): part[s,M,N] (f32, contiguous) holds
 * the partial product of K range s; madtp_splitk_ln then computes
 *   y = LayerNorm(scale * (sum_s part[s] + bias) + residual)          (med.py:246-250,326-328; nlvr_encoder.py:259-271)
 * with the reduction in a fixed order.  (K*esz/128) must be divisible by `splits`; dim % 4 == 0, dim <= 1024. */
int madtp_gemm_splitk(const void* A, const void* W, float* part, int M, int N, int K, int lda, int ldw, int splits,
                      int ab_dtype, void* stream);
int madtp_splitk_ln(const float* part, int splits, const float* bias, const float* residual, const float* gamma,
                    const float* beta, float* y32, void* ylp, int lp_dtype, int rows, int dim, float eps, float acc_scale,
                    float scale, void* stream);

/* Split-K partials of a LONG-K product (ABI 28): the weight gradient dW = dY^T X of the f16x3 backward (compress_nlvr_dtp.py:46-53's
 * loss.backward(); K = every token row of the batch) on the 256x256 ping-pong kernel: part[s,M,N] (f32) = acc_scale * A[:, Ks] W[:, Ks]^T
 * for f16-split operands (MADTP_F16S layouts, lda / ldw >= 2K f16 elements).  K % (128*splits) == 0, N % 8 == 0.  madtp_splitk_sum
 * adds the slabs in order: out[i] = part[0][i] + part[1][i] + ...; count % 4 == 0. */
int madtp_gemm_splitk_pp(const void* A, const void* W, float* part, int M, int N, int K, int lda, int ldw, int splits, float acc_scale,
                         void* stream);
int madtp_splitk_sum(const float* part, int splits, size_t count, float* out, void* stream);
/* (with acc_scale: y = LayerNorm(scale * (acc_scale * sum_s part[s] + bias) + residual); ylp in lp_dtype BF16 or F16S) */

/* Two independent GEMMs of identical shape, leading dimensions and dtypes (C_i = A_i @ W_i^T + bias_i; bias0 and bias1 both
 * given or both NULL) in ONE launch when the shape runs on the wave-specialised bf16 kernel, two madtp_gemm launches otherwise.
 * Replaces the key/value nn.Linear pairs of the two cross-attention branches (nlvr_encoder.py:177-178 for self0 and self1). */
int madtp_gemm_pair(const void* A0, const void* A1, const void* W0, const void* W1, const float* bias0, const float* bias1,
                    void* C0, void* C1, int M, int N, int K, int lda, int ldw, int ldc, int ab_dtype, int c_dtype,
                    float acc_scale0, float acc_scale1, void* stream);

/* Optional profiling of madtp_gemm launches with HIP events recorded on the launch stream (bench.py roofline leg).
 * madtp_profile_begin() starts recording; madtp_profile_end() stops, waits for the events and writes one line per
 * (dtype, M, N, K): "dtype M N K launches total_ms flops algorithmic_bytes" into buf; returns the bytes written. */
int madtp_profile_begin(void);
int madtp_profile_end(char* buf, int cap);

/* y = LayerNorm(x) * gamma + beta over the last dim (dim % 4 == 0, dim <= 1024); x is f32.
 * Writes y32 (f32, may be NULL) and/or ylp (may be NULL; lp_dtype BF16: bf16 [rows,dim]; F16S: f16-split [rows,2*dim]).
 * vit.py:186,205,309 (eps 1e-6); med.py:79,249,328 (eps 1e-12); clip/model.py:160-166 (eps 1e-5). */
int madtp_layernorm(const float* x, const float* gamma, const float* beta, float* y32, void* ylp, int lp_dtype,
                    int rows, int dim, float eps, void* stream);

/* im2col of non-overlapping patches for the patch-embedding GEMM (timm PatchEmbed Conv2d k=s=P, call site
 * vit.py:241-242,283): img f32 [B,3,S,S] -> cols [B*(S/P)^2, 3*P*P] (dtype out_dtype), column = c*P*P+ky*P+kx. */
int madtp_patchify(const float* img, void* cols, int B, int S, int P, int out_dtype, void* stream);

/* x[b,0,:] = cls + pos[0]; x[b,1+p,:] = patches[b*np+p,:] + pos[1+p]   (vit.py:285-289). All f32. */
int madtp_assemble_tokens(const float* patches, const float* cls, const float* pos, float* x,
                          int B, int np, int dim, void* stream);

/* BERT embeddings: y = LayerNorm(word_emb[ids] + pos_emb[0..L))  (med.py:63-86 / nlvr_encoder.py:62-86).
 * ids int64 [B,L]; writes y32 (f32) and/or ylp (lp_dtype BF16 or F16S). */
int madtp_bert_embed(const int64_t* ids, const float* word_emb, const float* pos_emb, const float* gamma,
                     const float* beta, float* y32, void* ylp, int lp_dtype, int B, int L, int dim, float eps, void* stream);

/* Multi-head attention core with the pruning-score side outputs.
 * q/k/v point at the first element of head 0 of token 0 for each operand; rows are tokens with row strides
 * ldq/ldk/ldv (elements), head h occupies columns [h*64, h*64+64).  io_dtype is the dtype of q,k,v and out.
 * scores = (q k^T) * scale (+ add_mask[b,j], f32 [B,Nk], may be NULL) ; P = softmax_j ; out = P v
 *   -> out[(b*Nq+i), h*64+d]  (ldo)                      vit.py:81-91; med.py:177-222; nlvr_encoder.py:176-223
 * Side outputs (all f32, pass NULL for colsum_part to skip them - cross-attention):
 *   colsum_part[b, rt, j] = sum over query rows i in 16-row tile rt, i>=1, of max_h P[b,h,i,j]   (vit.py:126-127)
 *   p0[b,h,j]   = P[b,h,0,j]                                                                      (vit.py:96)
 *   onorm[b,h,i]= || out[b,h,i,:] ||_2                                                            (vit.py:97)
 * Limits: head_dim 64; Nk <= 1024 (Nk > 256 takes a two-pass kernel on the exact-f32 MFMA). */
int madtp_attention(const void* q, const void* k, const void* v, void* out, const float* add_mask,
                    float* colsum_part, float* p0, float* onorm,
                    int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo,
                    float scale, int io_dtype, void* stream);
/* madtp_attention with one more additive mask: mask_qk f32 [Nq, >=Nk] (row stride ld_mask_qk), shared by all samples and heads,
 * added to the scaled scores next to add_mask - the causal attn_mask of CLIP's text tower (clip/model.py:466-472, applied as
 * attn_mask[:L,:L] by the patched MultiheadAttention, clip/mock.py:309-310).  Nk <= 256. */
int madtp_attention_qk_mask(const void* q, const void* k, const void* v, void* out, const float* add_mask,
                            const float* mask_qk, int ld_mask_qk, float* colsum_part, float* p0, float* onorm,
                            int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo,
                            float scale, int io_dtype, void* stream);

/* Cross-attention against a CACHE of encoder K/V blocks: sample b attends to block kv_batch_index[b] (int32 [B], device) of
 * k / v, i.e. rows kv_batch_index[b]*Nk .. +Nk-1 (NULL = block b, which is madtp_attention).  Lets many queries share the
 * projected K/V of one image (retrieval re-ranking) without copying them; no score side outputs. */
int madtp_attention_indexed(const void* q, const void* k, const void* v, const int32_t* kv_batch_index, void* out,
                            const float* add_mask, float* colsum_part, float* p0, float* onorm,
                            int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo,
                            float scale, int io_dtype, void* stream);

/* Two attention problems of identical shape, without the score side outputs, in ONE launch where the bf16 kernel allows
 * (<= 256 keys), two madtp_attention_indexed launches otherwise.  The twin cross-attention branches of an NLVR text layer
 * (nlvr_encoder.py:314-333: self0 attends to image 0, self1 to image 1). */
int madtp_attention_pair(const void* q0, const void* q1, const void* k0, const void* k1, const void* v0, const void* v1,
                         const int32_t* kv_batch_index, void* out0, void* out1, const float* add_mask0, const float* add_mask1,
                         int B, int H, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, float scale, int io_dtype,
                         void* stream);

/* Alignment-guided token-importance score, per-sample threshold and survivor count
 * (Block.Reduce_token vit.py:125-145 == med.py:347-371 == nlvr_encoder.py:404-432 == clip/model.py:196-218).
 * n = N-1 patch tokens.  token_attn f32: element [b,t,c] at token_attn[b*ldt_batch + t*ldt_row + c], c < K (raw
 * x.sd^T logits of patch token t; any strided [B,n,K] view with unit column stride).
 * Outputs: score f32 [B,n]; threshold f32 [B]; count int32 [B]; kmax int32[1] (optional, may be NULL) = max_b count
 * by atomicMax - the caller zeroes it before the launch; callers that read `count` back can do the max on the host. */
int madtp_token_score(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                      const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature,
                      float* score, float* threshold, int32_t* count, int32_t* kmax,
                      int B, int H, int N, void* stream);

/* Same launch, but k = max_b count is handed to the HOST: the last workgroup writes it to pinned host memory and the call
 * returns once it has arrived (*k_host).  This is the reference's one synchronisation per layer (`topk_num.item()`,
 * vit.py:145) without a device-to-host copy and a stream synchronisation; work queued on `stream` before the call has
 * completed when it returns.  Each device has its own ring of 16 hand-over slots (pinned pair + ticket counter): calls on
 * different devices, streams or host threads do not interfere, and no lock is held while waiting. */
int madtp_token_score_sync(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                           const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature,
                           float* score, float* threshold, int32_t* count, int32_t* k_host,
                           int B, int H, int N, void* stream);

/* The same in two steps: _publish claims a slot of the CURRENT device, launches token_score with it armed and returns its
 * sequence number; _wait(seq) spins until k has arrived and frees the slot (a _publish that is never waited for leaks its
 * slot: MADTP_E_BUSY after 16 such leaks, never a deadlock).  Kernels enqueued between the two run while the host waits; the
 * layer-level calls put the projection GEMM there. */
int madtp_token_score_publish(const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                              const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature,
                              float* score, float* threshold, int32_t* count, int B, int H, int N, int* seq_out,
                              void* stream);
int madtp_token_score_wait(int seq, const int32_t* count, int B, int32_t* k_host, void* stream);

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
/* The same with the following LayerNorm fused in (Block.norm2, vit.py:195): h32 and/or h_lp (compute dtype) receive
 * LayerNorm(y) row by row, bit-identical to madtp_layernorm(y); gamma == NULL: plain gather. */
int madtp_token_gather_ln(const float* x, const int32_t* dst_pos, const float* merge_w, float* y, int B, int N, int k, int dim,
                          const float* gamma, const float* beta, float eps, float* h32, void* h_lp, int lp_dtype, void* stream);

/* Additive-mask compaction for the text encoders (nlvr_encoder.py:451-452,531-533; med.py:388-390,429-440):
 * out[b,0]=mask[b,0]; out[b,1+p] = mask[b,1+order[b,p]] for p in [0,k]  with order = indices_sort (NLVR, order2 = NULL);
 * MED (order = indices, order2 = indices_sort): slots p<k take mask[b,1+order[b,p]], slot k takes the mask of the
 * (k+1)-th ranked token mask[b,1+order2[b,k]]. */
int madtp_mask_gather(const float* mask, const int64_t* order, int ld_order, const int64_t* order2, int ld_order2,
                      float* out, int B, int N, int k, void* stream);

/* Query_model's att_ft (models/utils.py:174-178): att_ft[b,c,:] (+)= sum_t softmax_t(token_attn[b,t,c]/sqrt(dim_sd)) * x[b,1+t,:]
 * token_attn as in madtp_token_score; ft f32: patch token t of sample b at ft[b*ldf_batch + t*ldf_row + d] (so
 * x[:,1:,:] of a [B,N,dim] tensor is passed without a copy); n patch tokens; out f32 [B,K,dim] contiguous;
 * accumulate!=0 adds into out (sd_img_ft_all += sd_img_ft, vit.py:300-303). */
int madtp_query_att_ft(const float* token_attn, int ldt_row, int ldt_batch, int K, const float* ft, int ldf_row,
                       int ldf_batch, float* out, float inv_sqrt_sd, int accumulate, int B, int n, int dim, int fast,
                       float* stats_ws, void* stream);
/* fast != 0: bf16-MFMA variant (fast mode), needs stats_ws = B*256 floats of scratch; 0: exact-f32 MFMA (stats_ws unused) */

/* The encoders' running sum  sd_ft_all = sum over layers l of att_ft_l  (vit.py:297-303, nlvr_encoder.py:608-613) in ONE
 * launch: segment l is layer l's (token_attn, ft) pair with the strides of madtp_query_att_ft.  stats_ws != NULL (nseg*B*256
 * floats): fast mode, bf16 MFMA; stats_ws == NULL: the exact-f32 arithmetic of madtp_query_att_ft(fast = 0) with the
 * per-layer summation order kept, i.e. bit-identical to accumulating layer by layer (parity modes);
 * the [K,dim] block of a sample stays in registers across the segments and is written once.  The caller keeps every
 * layer's token buffer and logits alive until the stream has run this call. */
typedef struct madtp_att_ft_seg {
    const float* token_attn;
    const float* ft;
    int n, ldt_row, ldt_batch, ldf_row, ldf_batch;
} madtp_att_ft_seg;
int madtp_query_att_ft_multi(const madtp_att_ft_seg* segs, int nseg, int K, float* out, float* stats_ws, float inv_sqrt_sd,
                             int accumulate, int B, int dim, void* stream);  /* stats_ws: nseg*B*256 floats of scratch */
/* The same sum in the "f16x3" precision mode: token rows and softmax weights (expf, true division) as f16-split planes, three
 * f16 MFMA products per tile (the rounding class of the exact-f32 kernel; att_ft only feeds the training loss, vit.py:300-303,
 * no pruning decision depends on it).  stats_ws (nseg*B*256 floats) is required. */
int madtp_query_att_ft_multi_split(const madtp_att_ft_seg* segs, int nseg, int K, float* out, float* stats_ws, float inv_sqrt_sd,
                                   int accumulate, int B, int dim, void* stream);

/* Alignment logits out[M,128] = x[M,dim] @ sd^T with the dictionary given as two [128,dim] 2-byte planes (rows beyond the
 * dictionary size zero) and x split in registers.  models/utils.py:170.
 *   split_dtype MADTP_BF16 (fast mode): bf16 hi/lo planes, xh.sh + xl.sh + xh.sl on the bf16 MFMA (~2^-16 relative error);
 *   split_dtype MADTP_F16S ("f16x3" mode): the f16 planes Q0 / Q1 of sd * 2^s, three f16 MFMA products (fp32-accurate,
 *   see MADTP_F16S above), result scaled by out_scale = 2^-s. */
int madtp_align_logits(const float* x, const void* sd_hi, const void* sd_lo, float* out, int M, int dim, int split_dtype,
                       float out_scale, void* stream);

/* vector_gather (models/utils.py:13-33): out[b,k,:] = vectors[b, indices[b,k], :]; f32 [B,L,D], int64 [B,K]. */
int madtp_vector_gather(const float* vectors, const int64_t* indices, float* out, int B, int L, int K, int D,
                        void* stream);

/* (a+b)*scale elementwise, f32 (nlvr_encoder.py:266 average of the two cross-attention branches). */
int madtp_add_scale(const float* a, const float* b, float* out, float scale, size_t n, void* stream);

/* f32 -> bf16 copy (weight preparation, activations entering a bf16 GEMM). */
int madtp_cast_bf16(const float* src, void* dst, size_t n, void* stream);
/* f32 -> 2-byte copy in the element format lp_dtype (MADTP_BF16 or MADTP_F16), scaled by `scale` first (f16 weights are stored
 * as w * 2^s so that small weights stay clear of the f16 subnormals; the GEMM's acc_scale 2^-s undoes it). */
int madtp_cast_lp(const float* src, void* dst, size_t n, int lp_dtype, float scale, void* stream);

/* Score arithmetic of the FAST precision modes (bf16 / f16): with on != 0 `madtp_token_score*` and the layer / encoder calls run
 * the softmax over tokens of vit.py:137-139 in log2 units with the hardware exp2 and one reciprocal per dictionary column instead
 * of two IEEE divisions and a precise expf per logit (that phase is VALU-bound: 10 of 17 us at 197 tokens).  The parity modes
 * (fp32, f16x3) keep on = 0: their arithmetic is the reference's.  Process-wide, set together with the precision mode
 * (madtp_amd/runtime.py); the LDS-resident kernels (one workgroup per sample, and the column-split one for long sequences) have
 * the fast form, the global-memory fallback does not.  Returns
 * the previous value.  No reference counterpart. */
int madtp_set_score_fast(int on);

/* Range flag of the f16 formats (MADTP_F16S planes and MADTP_F16 operands): a producer kernel (madtp_split_f16, the LayerNorm /
 * GEMM / attention epilogues that emit f16) that meets a value outside the f16 range (|x| >= 65504 or NaN) sets a sticky flag in
 * pinned host memory instead of silently handing an infinity on (which the next GEMM would turn into NaNs).  Returns the flag
 * (0 = clean) after the work queued on `stream` so far has completed, and clears it when `reset` != 0.  The layer-level entry
 * points check it with their host read of k and return MADTP_E_RANGE. */
int madtp_range_status(int reset, void* stream);

/* f32 [rows, K] (row stride ld_src) -> f16-split activation planes [rows, 2K] f16 (row stride ld_dst f16 elements):
 * dst[r, c] = f16(x), dst[r, K + c] = f16((x - f16(x)) * 2^11).  Activations entering an F16S GEMM whose producer is not
 * one of the fused epilogues (attention output, image tokens handed to the text encoder).  K % 4 == 0. */
int madtp_split_f16(const float* src, int ld_src, void* dst, int ld_dst, int rows, int K, void* stream);

/* Weight preparation for F16S GEMMs: w f32 [n, K] (row stride ldw) -> planes [Q0 | Q1] f16 [n, 2K] of w * 2^s, with
 * inv_scale = 2^s chosen by the caller as a power of two such that max|w| * 2^s <= 2^14 (madtp_lin.w_scale = 2^-s). */
int madtp_split_f16_weight(const float* w, int ldw, void* dst, int n, int K, float inv_scale, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * Layer-level entry points.  One call enqueues the kernel sequence of half a transformer layer (the host-side cut is
 * where the reference reads k = max_b count back with `.item()`: vit.py:145 / med.py:373 / nlvr_encoder.py:432).
 * Weights are "prepared" Linears: w is [n padded to 128 rows, k] in the compute dtype, b is f32 (may be NULL).
 * Scratch comes from a caller-owned workspace of at least *_workspace() bytes (256-byte aligned base).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct madtp_lin {
    const void* w;  /* [n_pad, k] row-major, compute dtype ([n_pad, 2k] f16 planes for MADTP_F16S) */
    const float* b; /* [n] or NULL */
    int n, k;
    float w_scale;  /* accumulator scale of the prepared weight: 2^-s for F16S planes (w stored as w * 2^s), 1 otherwise */
} madtp_lin;

/* models/vit.py Block (:106-207): norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2; also clip/model.py
 * ResidualAttentionBlock (:174-261): ln_1, attn.in_proj, attn.out_proj, ln_2, mlp.c_fc, mlp.c_proj */
typedef struct madtp_vit_block_w {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float eps, scale; /* LayerNorm eps; attention scale head_dim^-0.5 */
    madtp_lin qkv, proj, fc1, fc2;
    int heads, dim, dtype; /* dtype: MADTP_F32 (parity mode), MADTP_F16S (fp32-accurate on the f16 MFMA) or MADTP_BF16 (fast mode) */
    int act;               /* MLP activation: MADTP_ACT_GELU_ERF (BLIP ViT) or MADTP_ACT_QUICK_GELU (CLIP) */
    const float* attn_mask; /* optional additive [>=N, >=N] attention mask (CLIP text tower: causal), row stride ld_attn_mask */
    int ld_attn_mask;
} madtp_vit_block_w;

size_t madtp_vit_block_workspace(int B, int N, int dim, int hidden, int heads, int dtype);

/* Block.forward up to the pruning decision (vit.py:186-190 + Reduce_token :125-145):
 * x_out = x + proj(attention(qkv(norm1(x)))); if temperature > 0 also score[B,N-1], threshold[B], count[B], kmax[1]. */
int madtp_vit_block_attn(const madtp_vit_block_w* w, const float* x, float* x_out, void* ws, size_t ws_bytes, int B, int N,
                         const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature, float* score,
                         float* threshold, int32_t* count, int32_t* kmax, void* stream);

/* Rest of Block.forward (vit.py:148-161,195-205): k > 0 -> select/gather/merge to [B,k+2,dim] (writes indices[B,k],
 * indices_sort[B,N-1]); then y = x' + fc2(GELU(fc1(norm2(x')))).  k == 0 -> no pruning. */
int madtp_vit_block_mlp(const madtp_vit_block_w* w, const float* x, float* y, void* ws, size_t ws_bytes, int B, int N, int k,
                        const float* score, int64_t* indices, int64_t* indices_sort, void* stream);

/* Block.forward (vit.py:184-205) in one call = madtp_vit_block_attn, host read of k (madtp_token_score_sync), pruning rule
 * vit.py:148-149, madtp_vit_block_mlp.  x_attn [B,N,dim] scratch/output of the attention half; y and indices sized for the
 * unpruned case ([B,N,dim], [B,N-1]); *k_out = max_b count (0 when temperature <= 0), *k_used = k actually applied: when
 * > 0, y holds [B,k_used+2,dim] and indices [B,k_used] (both contiguous from the start of their buffers). */
int madtp_vit_block(const madtp_vit_block_w* w, const float* x, float* x_attn, float* y, void* ws, size_t ws_bytes, int B, int N,
                    const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature, float* score,
                    float* threshold, int32_t* count, int64_t* indices, int64_t* indices_sort, int* k_out, int* k_used,
                    void* stream);

/* The same with the pruning rule of CLIP's ResidualAttentionBlock (clip/model.py:220-221): the layer is pruned unless
 * k <= max_keep or N-1-k <= 1 (the text tower passes max_keep = position of the EOT token + 2, :492; max_keep = 0 is
 * madtp_vit_block).  As in madtp_vit_block the host read of k overlaps the projection GEMM. */
int madtp_vit_block_keep(const madtp_vit_block_w* w, const float* x, float* x_attn, float* y, void* ws, size_t ws_bytes, int B, int N,
                    const float* token_attn, int ldt_row, int ldt_batch, int K, float temperature, float* score,
                    float* threshold, int32_t* count, int64_t* indices, int64_t* indices_sort, int max_keep, int* k_out, int* k_used,
                    void* stream);

/* Query_model.forward(return_token_att=True) (models/utils.py:147-183) over a contiguous token buffer x[B,N,dim]:
 * token_attn_full[B*N, 128] = x @ sd^T (exact-f32 MFMA; sd_w is f32 [128,dim], rows >= K zero); row b*N+1+t is patch t.
 * att_ft[B,K,dim] (+)= softmax_t(logits/sqrt(sd_dim)) @ x[:,1:]  (skipped when att_ft is NULL). */
int madtp_query_model(const float* x, const void* sd_w, const void* sd_hi, const void* sd_lo, int split_dtype, float sd_scale,
                      int K, float* token_attn_full, float* att_ft, float* stats_ws, int accumulate, float inv_sqrt_sd, int B,
                      int N, int dim, void* stream);
/* sd_hi/sd_lo != NULL selects madtp_align_logits for the logits (split_dtype / sd_scale as there); with split_dtype
 * MADTP_BF16 att_ft also runs its bf16 kernel (needs stats_ws = B*256 floats), with MADTP_F16S it stays on the exact-f32 one */

/* models/med.py BertLayer (:332-467) / models/nlvr_encoder.py BertLayer (:385-559) */
typedef struct madtp_bert_layer_w {
    madtp_lin qkv, attn_out;            /* attention.self.{query|key|value} fused; attention.output.dense */
    const float *ln_att_g, *ln_att_b;   /* attention.output.LayerNorm */
    int cross;                          /* 0 no crossattention module, 1 single (MED), 2 twin (NLVR self0/self1) */
    int variant_nlvr;                   /* mask-gather rule and cross-attention mask rule of nlvr_encoder.py */
    int has_merge;                      /* crossattention.output.merge_layer present (NLVR layers >= 6) */
    madtp_lin cq[2], ckv[2], cdense[2], merge;
    /* optional fused twin projections (fused_twin != 0): cq_fused = [q0;q1] ([2*dim, dim]); cdense_fused has K = 2*dim
     * and consumes [c0|c1]: either [W0|W1] with bias b0+b1 and epilogue scale 0.5 (average, layers < 6) or the
     * merge_layer folded in, Wm[:, :dim] W0 | Wm[:, dim:] W1 (fused_twin == 2, epilogue scale 1). */
    int fused_twin;
    madtp_lin cq_fused, cdense_fused;
    const float *ln_cross_g, *ln_cross_b;
    madtp_lin inter, out;               /* intermediate.dense, output.dense */
    const float *ln_out_g, *ln_out_b;
    float eps, scale;
    int heads, dim, dtype;
    /* decoder layers (BertModel.forward(is_decoder=True), med.py:752-786: extended mask = causal[L,L] * padding[B,L]): the
     * additive causal part, f32 [L, >= L] with row stride ld_self_mask_qk, shared by all samples and heads and added to the
     * self-attention scores next to the padding mask (madtp_attention_qk_mask); NULL for encoder layers. */
    const float* self_mask_qk;
    int ld_self_mask_qk;
} madtp_bert_layer_w;

size_t madtp_bert_layer_workspace(int B, int L, int Nk, int dim, int hidden, int heads, int dtype);

/* att = LayerNorm(attention.output.dense(self_attention(hidden)) + hidden)  (med.py:408-418) [+ score/threshold/count/kmax] */
int madtp_bert_layer_attn(const madtp_bert_layer_w* w, const float* hidden, const float* mask2d, float* att, void* ws,
                          size_t ws_bytes, int B, int L, int Nk, const float* token_attn, int ldt_row, int ldt_batch, int K,
                          float temperature, float* score, float* threshold, int32_t* count, int32_t* kmax, void* stream);

/* Rest of BertLayer.forward (med.py:422-462 / nlvr_encoder.py:519-554): prune att+mask (k > 0), optional
 * cross-attention to enc0/enc1 ([B*Nk,dim], compute dtype), FFN.  y is [B,L',dim], mask_out [B,L'] (L' = k+2 or L). */
int madtp_bert_layer_rest(const madtp_bert_layer_w* w, const float* att, const float* mask2d, float* y, float* mask_out,
                          void* ws, size_t ws_bytes, int B, int L, int k, const float* score, int64_t* indices,
                          int64_t* indices_sort, int cross_mode, const void* enc0, const void* enc1, int Nk,
                          const float* enc_mask0, const float* enc_mask1, void* stream);

/* BertLayer.forward in one call = madtp_bert_layer_attn, host read of k, pruning rule med.py:374-375,
 * madtp_bert_layer_rest.  Buffers sized for the unpruned case (y [B,L,dim], mask_out [B,L], indices [B,L-1]); *k_used > 0:
 * y is [B,k_used+2,dim], mask_out [B,k_used+2], indices [B,k_used]. */
int madtp_bert_layer(const madtp_bert_layer_w* w, const float* hidden, const float* mask2d, float* att, float* y,
                     float* mask_out, void* ws, size_t ws_bytes, int B, int L, int Nk, const float* token_attn, int ldt_row,
                     int ldt_batch, int K, float temperature, float* score, float* threshold, int32_t* count,
                     int64_t* indices, int64_t* indices_sort, int cross_mode, const void* enc0, const void* enc1,
                     const float* enc_mask0, const float* enc_mask1, const void* hidden_lp, void* y_lp, const void* kv_pre0,
                     const void* kv_pre1, const int32_t* kv_index, int kv_ld, int* k_out, int* k_used, void* stream);
/* hidden_lp (optional, bf16 mode): compute-dtype copy of `hidden` - skips the cast; y_lp (optional): receives the
 * compute-dtype copy of y, emitted by the output LayerNorm, to be passed as the next layer's hidden_lp.
 * kv_pre0 / kv_pre1 (optional): a cache of this layer's cross-attention [k|v] projections ([blocks*Nk, 2*dim], compute dtype,
 * = madtp_gemm of the encoder tokens with the layer's fused key|value weights) - the projection GEMM is skipped (enc0/enc1
 * may then be NULL) and sample b attends to block kv_index[b] (NULL: block b).  kv_ld: row stride of kv_pre in elements (0 =
 * 2*dim) - the rows may be column slices of ONE [blocks*Nk, layers*2*dim] projection of the encoder tokens with all layers'
 * key|value weights stacked (the NLVR text encoder does that: 2 GEMMs per forward instead of 24).  Retrieval re-ranking
 * projects every image once per layer instead of once per (query, candidate) pair. */

/* ------------------------------------------------------------------------------------------------------------
 * Encoder-level entry points: ONE call runs every layer of an encoder - the query model in front of
 * each layer (vit.py:297-303 / med.py:513-524 / nlvr_encoder.py:608-613) and the layer itself - so the host side between two
 * layers is a few lines of C instead of a return to Python (two FFI crossings, ~10 tensor allocations and ~40 us per layer:
 * at 64 samples the text encoder's kernels run 5-18 us each and the GPU idled 15-25 us per layer waiting for the host).
 * The per-layer host read of k = max_b count stays (the pruning rule is the reference's, vit.py:145-149); all buffers are
 * caller-owned and sized for the UNPRUNED sequence, results of a pruned layer sit contiguously at the start of them.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct madtp_query_w {        /* models/utils.py Query_model operands, see madtp_query_model */
    const void* sd_w;                 /* f32 [128, dim] dictionary (rows >= K zero) */
    const void* sd_hi; const void* sd_lo; int split_dtype; float sd_scale;   /* optional split planes (madtp_align_logits) */
    int K; float inv_sqrt_sd;
    float* att_ft;                    /* [B,K,dim] running sum of the layers' att_ft, or NULL (not computed here) */
    float* stats_ws;                  /* B*256 floats when att_ft runs its bf16 kernel, else NULL */
} madtp_query_w;

typedef struct madtp_layer_io {       /* buffers and results of one layer */
    float* logits;                    /* [B*N_in, 128] logits of the layer's query model (token_attn = rows 1.. of each sample) */
    float* x_attn;                    /* [B,N_in,dim]: ViT x + attn(norm1(x)); BERT attention_output */
    float* y;                         /* [B,N_in,dim] layer output ([B,n_out,dim] contiguous when pruned) */
    void* y_lp;                       /* BERT, compute dtype != f32: copy of y for the next layer (may be NULL) */
    float* mask_out;                  /* BERT: [B,N_in] compacted additive mask (written when pruned) */
    float* score; float* threshold; int32_t* count;       /* [B,N_in-1], [B], [B] (written when temperature > 0) */
    int64_t* indices; int64_t* indices_sort;              /* [B,N_in-1] each; indices holds [B,k_used] when pruned */
    int k_out, k_used, n_out;         /* RESULTS: max_b count, k applied (0 = not pruned), tokens per sample after the layer */
} madtp_layer_io;

/* VisionTransformer.forward's block loop (vit.py:292-305): for each layer the query model on x (skipped when q == NULL or
 * temperature <= 0 ... see below) and Block.forward.  layers[l] as for madtp_vit_block; x0 [B,N0,dim] f32; ws: workspace of
 * madtp_vit_block_workspace(B, N0, ...) bytes.  q == NULL: plain blocks (temperature ignored).  The output of the last layer
 * is io[n_layers-1].y with io[n_layers-1].n_out tokens per sample (the final LayerNorm is the caller's, vit.py:309). */
int madtp_vit_encoder(const madtp_vit_block_w* const* layers, int n_layers, const madtp_query_w* q, const float* x0,
                      madtp_layer_io* io, void* ws, size_t ws_bytes, int B, int N0, float temperature, void* stream);

/* The same loop WITHOUT the per-layer host read of k (device-side lengths).  token_score leaves each
 * layer's decision - k = max_b count, the k applied under vit.py:148-149, the next layer's token count - in a device-side record
 * and every later kernel (top-k select, gather + norm2, the GEMMs' M, the next layer's LayerNorm / alignment logits / attention N)
 * reads its size from there, with grids, key-tile instantiations and buffers sized for the unpruned sequence; the whole encoder
 * is enqueued ahead and the host reads the records ONCE at the end (io[l].k_out / k_used / n_out).  Results are bit-identical to
 * madtp_vit_encoder.  For the launch-bound regime: B * N0 < 4096 token rows and N0 <= 256 (MADTP_E_SHAPE otherwise); q and
 * temperature > 0 are required, q->att_ft must be NULL (the caller sums att_ft after the call, when the token counts are known);
 * dims_dev: (n_layers + 2) * 4 int32 of device scratch, dims_host: (n_layers + 1) * 4 int32 of host memory - or NULL (round 6):
 * enqueue only, no copy and no wait (dims_dev[4 l .. 4 l + 3] = {N_l, k, k applied, N_{l+1}} is the caller's to read later): the form a
 * stream capture into a hipGraph takes (tools/graph_replay_probe.py). */
int madtp_vit_encoder_async(const madtp_vit_block_w* const* layers, int n_layers, const madtp_query_w* q, const float* x0,
                            madtp_layer_io* io, void* ws, size_t ws_bytes, int B, int N0, float temperature, int32_t* dims_dev,
                            int32_t* dims_host, void* stream);

/* BertEncoder.forward's layer loop (med.py:509-571 / nlvr_encoder.py:600-660): query model on the hidden states, then
 * BertLayer.forward; the compacted mask and the compute-dtype copy of the output are handed from layer to layer.
 * hidden0 [B,L0,dim] f32, hidden0_lp optional compute-dtype copy, mask0 additive [B,L0] (may be NULL when temperature <= 0);
 * cross-attention operands as for madtp_bert_layer, kv_pre0 / kv_pre1: optional arrays of n_layers pointers. */
int madtp_bert_encoder(const madtp_bert_layer_w* const* layers, int n_layers, const madtp_query_w* q, const float* hidden0,
                       const void* hidden0_lp, const float* mask0, madtp_layer_io* io, void* ws, size_t ws_bytes, int B, int L0,
                       int Nk, float temperature, int cross_mode, const void* enc0, const void* enc1, const float* enc_mask0,
                       const float* enc_mask1, const void* const* kv_pre0, const void* const* kv_pre1, const int32_t* kv_index,
                       int kv_ld, void* stream);

/* The same loop WITHOUT the per-layer host read of k (device-side lengths for the text encoders -
 * models/med.py:369, models/nlvr_encoder.py:432 `topk_num = torch.max(idx.sum(dim=1)).item()`).  As in madtp_vit_encoder_async the
 * layer's decision stays in a device-side record and every later kernel reads its size from it - including the compaction of
 * the additive padding mask (med.py:388-390 / nlvr_encoder.py:451-452) and the query count of the cross-attention; grids and
 * buffers are the unpruned sequence's; ONE host read of the records at the end (io[l].k_out / k_used / n_out).  Linear +
 * LayerNorm pairs run without split-K (whose factor depends on a row count the host does not know), so the summation order
 * differs from madtp_bert_encoder's: f32-accurate, pinned against the reference fixtures.  Takes text mode, MED single
 * cross-attention and the NLVR twin cross-attention (fused or separate projections); B * L0 < 4096 rows, L0 <= 256, Nk <= 256
 * (MADTP_E_SHAPE otherwise); q, mask0 and temperature > 0 are required, q->att_ft must be NULL; io[l].mask_out is written by
 * every layer; dims_dev: (n_layers + 2) * 4 int32 of device scratch, dims_host: (n_layers + 1) * 4 int32 of host memory. */
int madtp_bert_encoder_async(const madtp_bert_layer_w* const* layers, int n_layers, const madtp_query_w* q, const float* hidden0,
                             const void* hidden0_lp, const float* mask0, madtp_layer_io* io, void* ws, size_t ws_bytes, int B, int L0,
                             int Nk, float temperature, int cross_mode, const void* enc0, const void* enc1, const float* enc_mask0,
                             const float* enc_mask1, const void* const* kv_pre0, const void* const* kv_pre1, const int32_t* kv_index,
                             int kv_ld, int32_t* dims_dev, int32_t* dims_host, void* stream);

/* Incremental decoding (models/med.py:1071-1094: prepare_inputs_for_generation feeds the last token only once past_key_values
 * exist; the reference caches every layer's self-attention key / value): ONE new token per row through all decoder layers.
 * layers: MED layers with single cross-attention (madtp_bert_layer_w.cross == 1); x [rows, dim] f32 = the embedded new tokens at
 * position t; kv_cache [n_layers][rows][Lmax][2 dim] in the attention dtype (bf16 / f16 in the fast modes, f32 otherwise): this
 * call appends position t of every row and attends to positions 0..t; the caller re-orders the rows between steps
 * (_reorder_cache :1091-1094); kv_pre[l] / kv_index / kv_ld / Nk: the cached cross-attention [k|v] of the encoder states as for
 * madtp_bert_encoder; group (ABI 28): `group` consecutive rows - the beams of one item - share an encoder [k|v] block and run the
 * cross-attention as one sequence of `group` queries (kv_index then holds rows / group entries; 1: one entry per row); y [rows, dim]
 * f32; ws: the larger of madtp_bert_layer_workspace(rows, 1, Nk, ...) and (rows / group, group, Nk, ...) bytes.  Lmax <= 256. */
int madtp_bert_decode_step(const madtp_bert_layer_w* const* layers, int n_layers, const float* x, void* kv_cache, int rows, int t,
                           int Lmax, const void* const* kv_pre, const int32_t* kv_index, int kv_ld, int Nk, int group, float* y,
                           void* ws, size_t ws_bytes, void* stream);
/* _reorder_cache (models/med.py:1091-1094) for that cache: dst[l, r, 0..t) = src[l, beam_src[r], 0..t) for the t positions filled so
 * far; src / dst: two [n_layers][rows][Lmax][row_bytes] buffers (src != dst), beam_src int64 [rows]; row_bytes % 16 == 0. */
int madtp_kv_cache_reorder(const void* src, void* dst, const int64_t* beam_src, int n_layers, int rows, int Lmax, int t, int row_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Answer ranking with the teacher-forced decoder (inference half): models/med.py BertLMHeadModel
 * :1036-1042 and models/blip_vqa.py rank_answer :166-172.  The decoder itself is madtp_bert_layer with self_mask_qk set, the
 * LM head (med.py:616-657) is madtp_gemm (GELU) + madtp_layernorm + madtp_gemm over the vocabulary.
 * ------------------------------------------------------------------------------------------------------------ */
/* loss[b] = sum over t < n_pred of the label-smoothed cross-entropy of row (b, t) of the prediction scores against
 * labels[b, t + 1] (next-token prediction: shifted scores / labels, med.py:1038-1039; CrossEntropyLoss(reduction='none',
 * label_smoothing) summed per sequence :1040-1042); labels == -100 contribute 0.  logits f32 [B*rows_per_seq, >= V] with row
 * stride ld (row b*rows_per_seq + t), labels int64 [B, ld_labels], loss f32 [B]. */
int madtp_lm_loss(const float* logits, int ld, int rows_per_seq, int n_pred, int V, const int64_t* labels, int ld_labels,
                  float label_smoothing, float* loss, int B, void* stream);
/* out[q, a] = softmax(logits[q, :V])[tok[a]]  (blip_vqa.py:170-171: F.softmax(logits, dim=1).index_select(1, answer_first_token));
 * logits f32 [Q, >= V] with row stride ld, tok int64 [A], out f32 [Q, A]. */
int madtp_token_prob(const float* logits, int ld, int V, const int64_t* tok, int A, float* out, int Q, void* stream);
/* Candidate selection of one beam-search step (transformers 4.15 generation_utils.py `beam_search`, the search behind
 * text_decoder.generate(num_beams=...) at models/blip_vqa.py:134-140 and models/blip.py:189-196):
 *   score[b, j * V + t] = log_softmax(logits[b * num_beams + j, :V])[t] + beam_scores[b * num_beams + j]   (t == suppress_token: -inf,
 *   the MinLengthLogitsProcessor's treatment of EOS below min_length; -1 = none)
 *   out_scores / out_index [B, n_top] = the n_top (= 2 * num_beams) best of the num_beams * V candidates of item b in descending
 *   order (ties: the lower flat index j * V + t first; fewer than n_top finite candidates: index -1, score -inf).
 * logits f32 [B * num_beams, >= V] with row stride ld (the LM head's scores at the last position), beam_scores f32
 * [B * num_beams]; num_beams <= 8, n_top <= 16. */
int madtp_beam_topk(const float* logits, int ld, int V, const float* beam_scores, int num_beams, int n_top, int suppress_token,
                    float* out_scores, int32_t* out_index, int B, void* stream);
/* The same with the library's RepetitionPenaltyLogitsProcessor in front (generate(repetition_penalty=...), models/blip.py:161,
 * 195): a token that already occurs in row (b, j) of prev_ids int64 [B * num_beams, ld_prev] (the beams' sequences so far,
 * prompt included, cur_len columns) has its log-probability lp replaced by lp < 0 ? lp * penalty : lp / penalty - beam_search
 * of transformers 4.15 runs the processors on the log-softmax scores - before the beam score is added. */
int madtp_beam_topk_penalty(const float* logits, int ld, int V, const float* beam_scores, int num_beams, int n_top,
                            int suppress_token, const int64_t* prev_ids, int ld_prev, int cur_len, float repetition_penalty,
                            float* out_scores, int32_t* out_index, int B, void* stream);
/* The hypothesis book-keeping of one beam-search step on the device (ABI 28; transformers 4.15 generation_beam_search.py
 * BeamSearchScorer.process + BeamHypotheses.add / is_done behind text_decoder.generate(num_beams=...), models/blip_vqa.py:134-140,
 * models/blip.py:189-196), fed by madtp_beam_topk's candidates [B, n_top]: per item, in rank order, an EOS candidate of rank <
 * num_beams closes a hypothesis of the cur_len tokens of its row (score = sum_logprobs / denom in double, denom = cur_len **
 * length_penalty computed by the caller; kept while among the num_beams best), the other candidates become the next beams until
 * num_beams are found.  ids_in / ids_out: int64 [B * num_beams, ld_ids] (the beams' sequences; ids_out gets the re-ordered rows +
 * the new token at column cur_len), beam_scores f32 [B * num_beams] (in place), beam_src int64 [B * num_beams] (the row each new
 * beam continues - the _reorder_cache index of models/med.py:1091-1094).  Hypothesis state per item, S = num_beams + 1 slots:
 * hyp_n [B] (count), hyp_order [B,S] (list order -> slot), hyp_score [B,S] (double), hyp_len [B,S], hyp_tok [B,S,ld_ids], worst [B]
 * (double, 1e9 when empty), done [B] (a finished item emits pad tokens, zero scores, source row 0, as the library does).  err [1] is
 * set when an item has fewer than num_beams open continuations among its candidates.  num_beams <= 8. */
int madtp_beam_update(const float* cand_scores, const int32_t* cand_index, int n_top, int V, const int64_t* ids_in, int64_t* ids_out,
                      int ld_ids, int cur_len, float* beam_scores, int64_t* beam_src, int32_t* hyp_n, int32_t* hyp_order,
                      double* hyp_score, int32_t* hyp_len, int64_t* hyp_tok, double* worst, int32_t* done, int32_t* err, double denom,
                      int num_beams, int eos_token, int pad_token, int early_stopping, int B, void* stream);
/* One step of nucleus sampling (models/blip.py:175-186: text_decoder.generate(do_sample=True, top_p=0.9, repetition_penalty=1.1);
 * transformers 4.15 `sample`): on the raw last-position scores of every row - RepetitionPenaltyLogitsProcessor over prev_ids
 * (s < 0 ? s * penalty : s / penalty for tokens already in the row; NULL / 1.0: none), EOS suppression below min_length
 * (suppress_token, -1 = none), TopKLogitsWarper(top_k; the library default config.top_k = 50), TopPLogitsWarper(top_p: a token
 * stays while the tokens ranked before it hold <= top_p of the softmax mass), softmax of the survivors and ONE draw per row as the
 * inverse CDF (survivors in descending order of score, ties by ascending index) at the caller's uniform number u[row] in [0, 1).
 * logits f32 [B, >= V] with row stride ld, prev_ids int64 [B, ld_prev] (cur_len columns), out_token int64 [B], out_prob (optional)
 * f32 [B] the drawn token's probability among the survivors.  V <= 36864, top_k <= 64. */
int madtp_sample_top_p(const float* logits, int ld, int V, const int64_t* prev_ids, int ld_prev, int cur_len, float repetition_penalty,
                       int suppress_token, int top_k, float top_p, const float* u, int64_t* out_token, float* out_prob, int B,
                       void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * Backward of the pruned ViT block (first half; csrc/backward.hip): the pieces of loss.backward()
 * through models/vit.py Block.forward (:183-207) that are not a GEMM.  fp32 arithmetic, fixed-order reductions.  The GEMMs
 * (dgrad = dY W, wgrad = dY^T X) are madtp_gemm (MADTP_F32) on operands transposed by madtp_transpose_pad.  Orchestrated by
 * madtp_amd/backward.py (torch.autograd.Function around Block.forward).
 * ------------------------------------------------------------------------------------------------------------ */
/* dst[c, r] = src[r, c] (r < R, c < C), zero elsewhere of dst [Cp, Rp] (row stride ld_dst >= Rp): GEMM operands of the
 * backward need K contiguous (nn.Linear layout) and padded (K % 32 == 0, 128-row weight padding). */
int madtp_transpose_pad(const float* src, int ld_src, int R, int C, float* dst, int ld_dst, int Rp, int Cp, void* stream);
/* The same transpose straight into the f16-split operand planes of the f16x3 backward's weight gradient (ABI 28): dst f16
 * [Cp, 2*Rp] with row c = the planes of src[:, c] - weight_format 0: activation planes [P0 | P1] (dY^T), 1: weight planes [Q0 | Q1]
 * at scale 1 (X^T) - zero beyond R / C.  Rp % 4 == 0.  One pass instead of madtp_transpose_pad + madtp_split_f16(_weight).
 * colsum_out (or NULL): f32 [C], the column sums of src over its R rows - the bias gradient of the same dY - from per-tile
 * partials in colsum_ws (ceil(Rp/64) * C floats), added in a fixed order. */
int madtp_transpose_split(const float* src, int ld_src, int R, int C, void* dst, int Rp, int Cp, int weight_format,
                          float* colsum_out, float* colsum_ws, void* stream);
/* Both f16-split operand forms of a weight for the f16x3 training step in one pass (ABI 28): w f32 [N, K] (row stride ldw) times
 * inv_scale = 2^s -> planes f16 [rows, 2K]: rows row0 .. row0+N-1 = [Q0 | Q1] (the forward GEMMs' weight operand; several weights of
 * a fused projection write their row blocks of one buffer) and planes_t f16 [>= K rows, 2*Ntp]: (k, col0+n) = Q0, (k, Ntp+col0+n) = Q1
 * of w[n, k] (dgrad's operand W^T).  Either may be NULL.  Padding is not written: the buffers are zero-initialised once by the
 * caller and reused across parameter versions.  K % 4 == 0, Ntp % 4 == 0, col0 % 4 == 0. */
int madtp_weight_planes(const float* w, int ldw, int N, int K, float inv_scale, void* planes, int row0, void* planes_t, int Ntp, int col0,
                        void* stream);
/* out[c] = sum_r dy[r, c]: bias gradients (nn.Linear, LayerNorm beta).  part_ws: P * N floats of scratch, P = 64 for M >= 4096,
 * 16 for M >= 256, else 1 (the row chunks summed in order; 64 * N always suffices). */
int madtp_colsum(const float* dy, int ld, int M, int N, float* out, float* part_ws, void* stream);
/* g = act(u) (g != NULL) and / or du = dg * act'(u) (du != NULL): Mlp's GELU (vit.py:34) and its derivative; n % 4 == 0. */
int madtp_act_fwd_bwd(const float* u, const float* dg, float* g, float* du, size_t n, int act, void* stream);
/* LayerNorm backward (vit.py:186,205 norm1 / norm2): dx = dLN(x)^T dy (+ add, the residual branch's gradient, may be NULL);
 * dgamma / dbeta (may be NULL).  ws: 2 * rows + 64 * dim floats. */
int madtp_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* add, float* dx, float* dgamma,
                        float* dbeta, float* ws, int rows, int dim, float eps, void* stream);
/* Backward of madtp_token_gather (vit.py:153-161, values only - `indices` carry no gradient): dy [B,k+2,dim] ->
 * dx [B,N,dim] (kept token: its row of dy; dropped token: merge_w * dy[b,k+1]; CLS: dy[b,0]) and
 * dw [B,N-1] = <dy[b,k+1], x[b,1+t]> for dropped tokens (0 for kept): the gradient of the merge weights. */
int madtp_token_gather_bwd(const float* dy, const float* x, const int32_t* dst_pos, const float* merge_w, float* dx, float* dw,
                           int B, int N, int k, int dim, void* stream);
/* Backward of the importance score as far as autograd follows it in the reference (vit.py:126-134, :95-101): from dw (above)
 * through w = I / (sum_dropped I + 1e-8), I = (self_attn_w + token_attn_w + cls_attn) / 3 to
 *   da [B,N]          gradient of a_j = sum_{i>=1} max_h P[b,h,i,j] (un-normalised; da[b,0] = 0)
 *   dp0 [B,H,N]       gradient of P[b,h,0,j]
 *   dnrm_scale [B,H,N] gradient of ||attn_out[b,h,j,:]|| divided by that norm (d attn_out += dnrm_scale * attn_out)
 *   dtoken_attn [B,N-1,K] dense gradient of the alignment logits (one non-zero per row: the row maximum)
 * Inputs as produced by madtp_attention / madtp_token_score / madtp_token_select in the forward. */
int madtp_token_score_bwd(const float* dw, const float* score, const int32_t* dst_pos, const float* merge_w,
                          const float* colsum_part, int n_row_tiles, const float* p0, const float* onorm,
                          const float* token_attn, int ldt_row, int ldt_batch, int K, float* da, float* dp0, float* dnrm_scale,
                          float* dtoken_attn, int B, int H, int N, void* stream);
/* Softmax-attention backward with recomputed probabilities (vit.py:81-91) plus the score terms above:
 *   dP[h,i,j] = dout_i . v_j + [i == 0, j >= 1] dp0[h,j] + [i >= 1, j >= 1, h == argmax_h' P[h',i,j]] da[j]
 *   dS = P (dP - rowsum(P dP)); dq = scale dS k; dk = scale dS^T q; dv = P^T dout      (dout += dnrm_scale * out first)
 * q/k/v/dq/dk/dv: f32, rows b*N+i, head h at columns [64h, 64h+64) of each base pointer (slices of the fused qkv buffer).
 * ws: madtp_attention_bwd_workspace(B,H,N) bytes (P and dS [B,H,N,N] f32 + the head arg-max [B,N,N]).  N <= 1024. */
/* Cross-attention backward (med.py:143-236 with encoder_hidden_states; nlvr_encoder.py:142-237): Nq text queries against Nk
 * encoder tokens, softmax probabilities recomputed, no score terms.  q [B*Nq, ldq], k / v [B*Nk, ldkv] (e.g. the halves of a fused
 * [k|v] projection), dout [B*Nq, ldo] -> dq [B*Nq, lddq], dk / dv [B*Nk, lddkv]; key_mask additive [B,Nk] or NULL; exact f32. */
size_t madtp_attention_bwd_cross_workspace(int B, int H, int Nq, int Nk);
int madtp_attention_bwd_cross(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* key_mask,
                              const float* dout, int ldo, float* dq, int lddq, float* dk, float* dv, int lddkv, void* ws,
                              size_t ws_bytes, int B, int H, int Nq, int Nk, float scale, float p_drop,
                              unsigned long long seed, unsigned long long site, void* stream);

/* Dropout / DropPath of the training forward (ABI 28; models/med.py:55,111,244,323: nn.Dropout at hidden_dropout_prob /
 * attention_probs_dropout_prob 0.1, configs/med_config.json:5,7; models/vit.py:114,186,205: DropPath): y = residual + x * keep /
 * (1 - p) (residual may be NULL), keep drawn per element (per_sample == 0) or once per run of per_sample elements (DropPath: one
 * draw per sample).  Masks are counter-based - Philox4x32-10 keyed by `seed`, counter (index / 4, site), word index & 3, keep = u >= p
 * with u = (word >> 8) 2^-24 - so the backward applies the same call to dY with the same (seed, site) and nothing is stored.
 * n % 4 == 0, per_sample % 4 == 0. */
int madtp_dropout(const float* x, const float* residual, float* y, size_t n, size_t per_sample, float p, unsigned long long seed,
                  unsigned long long site, void* stream);
/* Attention of the training forward with attention_probs dropout (med.py:202-222, nlvr_encoder.py:201-223): P = softmax(scale q k^T
 * + key_mask + mask_qk) written to P [B,H,Nq,Nk] f32, out = (P o mask / (1 - p_drop)) v with the mask of (seed, site) over P's
 * elements in memory order; colsum_part / p0 / onorm: madtp_attention's side outputs (from the undropped P and the dropped out, as
 * med.py:227-233 takes them) or NULL.  madtp_attention_bwd / _cross take the same (p_drop, seed, site).  Exact f32. */
int madtp_attention_train(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* key_mask, const float* mask_qk,
                          int ld_mqk, float* P, float* out, int ldo, float* colsum_part, float* p0, float* onorm, int B, int H, int Nq,
                          int Nk, float scale, float p_drop, unsigned long long seed, unsigned long long site, void* stream);

/* Backward of the query model's att_ft branch (models/utils.py:174-178: W = softmax over tokens of inner / sqrt(sd_dim), att_ft =
 * W q), the part of VisionTransformer.forward's second output (vit.py:297-303, consumed by the training drivers' alignment loss).
 * Given dA = d att_ft [B,K,D]: dinner[B,n,K] += W (q dA^T - sum_n W q dA^T) / sqrt(sd_dim), dq[B,n,D] += W^T dA.  inner, dinner: dense
 * [B,n,K] (dinner usually already holds the gradient that reaches the logits through token_attn); q, dq: dense [B,n,D]; ws: 2*B*K*n
 * floats (ABI 28: W and q dA^T, the latter a batched exact-f32 MFMA product).  n <= 1024, K <= 128, D <= 1024.  Exact f32, fixed
 * summation order. */
int madtp_att_ft_bwd(const float* inner, const float* q, const float* dA, float inv_sqrt_d, float* dinner, float* dq, float* ws,
                     int B, int n, int K, int D, void* stream);

/* The attention map itself, P[b,h,i,j] = softmax_j(scale q_i . k_j) as f32 [B,H,N,N] (vit.py:81-83 `self.save_attention_map(attn)`):
 * the forward never materialises it; Attention.get_attention_map() of the mirror recomputes it on demand from the layer's input
 * (q / k: f32 row views as in madtp_attention_bwd). */
int madtp_attention_probs(const float* q, const float* k, int ld, const float* key_mask, float* P, int B, int H, int N, float scale,
                          void* stream);  /* key_mask: additive [B,N] over the keys (BERT padding mask, med.py:197-199) or NULL */
/* The general form (ABI 29): Nq queries against Nk keys per sample (cross-attention med.py:158-162, or a self-attention whose keys
 * include a cache, med.py:164-168), an optional additive mask over the keys [B,Nk] and over (query, key) pairs [Nq, ld_mqk] (the
 * decoder's causal mask) -> P f32 [B,H,Nq,Nk]: what `output_attentions=True` adds to BertLayer.forward's tuple (med.py:221, 450-456). */
int madtp_attention_probs_x(const float* q, const float* k, int ldq, int ldk, const float* key_mask, const float* mask_qk, int ld_mqk,
                            float* P, int B, int H, int Nq, int Nk, float scale, void* stream);
size_t madtp_attention_bwd_workspace(int B, int H, int N);
int madtp_attention_bwd(const float* q, const float* k, const float* v, int ld, const float* key_mask, const float* mask_qk,
                        int ld_mqk, /* mask_qk: additive [N, ld_mqk] over (query, key) pairs - the decoder's causal mask - or NULL */
                        const float* dout, int ldo, const float* out,
                        int ldout, const float* dnrm_scale, const float* da, const float* dp0, float* dq, float* dk, float* dv,
                        int ldd, void* ws, size_t ws_bytes,
                        float* dp_out, /* [B,H,N,N] or NULL: the gradient of the attention probabilities (vit.py:189 register_hook) */
                        int B, int H, int N, float scale, float p_drop, unsigned long long seed,
                        unsigned long long site, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MADTP_HIP_H */
