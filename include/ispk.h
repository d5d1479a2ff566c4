/* ispk.h — C ABI of libispk.so: the MI355X (gfx950) kernels behind the isp-tts acoustic-model forward path.
 *
 * The reference (ilya16/isp-tts) has no FFI layer: its operator surface is Python nn.Modules plus one
 * numba-compiled function.  This header is the boundary a drop-in binds instead (ctypes stub: INTEGRATION.md);
 * every entry point names the reference interface it replaces (paths relative to the reference's `tts/`).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch/HIP types except the opaque stream handle.
 *   - All pointers are DEVICE pointers (caller-allocated, e.g. tensor.data_ptr()); row-major; strides in ELEMENTS.
 *   - No allocation, no ownership transfer, no host synchronisation: launches are asynchronous on `stream`
 *     (NULL = the legacy default stream) and are graph-capturable.
 *   - Return value: 0 ok; < 0 argument error (ISPK_E_*), message via ispk_last_error_string() (thread-local);
 *     > 0 a hipError_t from the launch.
 *   - Thread-safe and re-entrant: no mutable global state.
 *   - `lengths` are int64 on device, as the reference's collator produces them (data/collator.py:36,45).
 *   - `row_mask` is a byte per row (torch.bool storage): nonzero = valid.
 */
#ifndef ISPK_H
#define ISPK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ispk_stream_t; /* == hipStream_t */

#define ISPK_ABI_VERSION 2 /* 2: round 3 - entry points removed (fused variants, _amp attention pair) and added (split fp16, util, graph support) */

#define ISPK_E_NULL (-1)        /* required pointer is NULL */
#define ISPK_E_SHAPE (-2)       /* size out of the supported range */
#define ISPK_E_ALIGN (-3)       /* pointer / stride alignment */
#define ISPK_E_UNSUPPORTED (-4) /* combination not implemented */

int32_t ispk_abi_version(void);
const char* ispk_last_error_string(void);
/* Fills name[cap] with the device's gcnArchName; returns CU count (>0) or a negative/hip error. */
int32_t ispk_device_info(char* name, int32_t cap);

/* ---------------------------------------------------------------------------------------------------------------
 * Monotonic Alignment Search.
 * Replaces: modules/aligner/mas.py:29-35 `b_mas` (numba CPU) and modules/aligner/cuda_mas.py:11-46 `cuda_b_mas`
 * (numba CUDA), as dispatched by models/acoustic/modules/alignment.py:291-331.
 *   logits    [B][M_max][L_max] fp32, element strides (stride_b, stride_m, 1); NOT modified
 *   text_len  [B] int64 (reference `in_lens`),  1 <= text_len[b] <= L_max <= 512
 *   mel_len   [B] int64 (reference `out_lens`), 1 <= mel_len[b]  <= M_max <= 4096
 *   attn_hard [B][M_max][L_max] int16, contiguous: one-hot rows, zero outside (mel_len, text_len)  (fully written)
 *   dur       [B][L_max] int64 or NULL: column sums of attn_hard (alignment.py:275 `attn_hard.sum(dim=1)`) with the
 *             reference's fix-up applied (alignment.py:278-282: mel_len[b] - sum goes to column 0; zero for valid lengths)
 *   path      [B][M_max] int16 or NULL: chosen text index per mel row, -1 for rows >= mel_len
 * One wavefront runs the DP of one utterance; ties go to the diagonal predecessor exactly as mas.py:17.
 */
int32_t ispk_mas_f32(const float* logits, const int64_t* text_len, const int64_t* mel_len, int16_t* attn_hard,
                     int64_t* dur, int16_t* path, int32_t B, int32_t M_max, int32_t L_max, int64_t stride_b,
                     int64_t stride_m, ispk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * LayerNorm / AdaptiveLayerNorm (+ row mask).
 * Replaces: modules/transformer/normalization.py:20-27 (LayerNorm) and :37-61 (AdaptiveLayerNorm), plus the
 * `out * mask` of transformer.py:101-102 / :205-206 when row_mask is given.
 *   y[r][:] = mask[r] * ( scale * ((x[r][:] - mean) / sqrt(var + eps)) + shift ),   biased variance
 *   plain   : scale = gamma[D], shift = beta[D]                       (ada_scale == NULL)
 *   adaptive: scale = ada_scale[b][D], shift = ada_shift[b][D], b = r / rows_per_batch, row stride ada_stride
 *             (ada_stride = 0 broadcasts one condition row over the whole batch: the [1,1,C] case of :57)
 * D must be a multiple of 64 and <= 1024.  x_f32 in, y in the named dtype.
 */
int32_t ispk_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* ada_scale,
                           const float* ada_shift, int64_t ada_stride, int32_t rows_per_batch,
                           const uint8_t* row_mask, float* y, int64_t ldy, int32_t rows, int32_t D, float eps,
                           ispk_stream_t stream);
int32_t ispk_layernorm_f32_bf16(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                const float* ada_scale, const float* ada_shift, int64_t ada_stride,
                                int32_t rows_per_batch, const uint8_t* row_mask, uint16_t* y, int64_t ldy, int32_t rows,
                                int32_t D, float eps, ispk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Linear layers on MFMA:  C[i][j] = epilogue( sum_k A[i][k] * W[j][k] ),  A [M][K], W [N][K] (nn.Linear layout).
 * Replaces: nn.Linear call sites attention.py:105,111,168 (to_q / to_kv / to_out), feedforward.py:33-36
 * (Linear -> GELU(erf) -> Linear), transformer.py:170 (project_emb), model.py:167-168 (to_mel + transpose + mask),
 * and the residual / mask arithmetic of transformer.py:91,105,110 and attention.py:172 through `flags`.
 *
 *   v = acc (+ bias)                      bias indexed by column j, or by row i with ISPK_EP_BIAS_ROW
 *   v = gelu_erf(v) | silu(v)             ISPK_EP_GELU | ISPK_EP_SILU
 *   v = mask * v                          ISPK_EP_MASK_ACC   (mask BEFORE the residual add: attention.py:172)
 *   v = v + resid[i][j]                   resid != NULL (same indexing as C, leading stride ldr)
 *   v = mask * v                          ISPK_EP_MASK_OUT   (mask AFTER the residual add: transformer.py:110)
 *   mask indexed by row i, or by column j with ISPK_EP_MASK_COL
 *   store: C[i*ldc + j], or with cols_per_batch > 0:  C[(j / cpb) * batch_stride + i*ldc + (j % cpb)]
 *          (to_mel: A = weight [80][384], "W" = activations [B*M][384], cpb = M, ldc = M, batch_stride = 80*M
 *           -> mel[B][80][M] written with consecutive lanes along the mel-frame axis)
 *
 * Requirements: K % 8 == 0; lda, ldw % 4 == 0 (fp32) / % 8 (bf16); A, W 16-byte aligned.  lda may be SMALLER than K:
 * rows then overlap in memory (the sliding-window view that turns a padded channel-last Conv1d into a GEMM, see below).
 * _f32      : fp32 in, v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate), fp32 out.
 * _bf16     : bf16 in, v_mfma_f32_32x32x16_bf16 (fp32 accumulate), epilogue in fp32; C/resid dtype per flags.
 */
#define ISPK_EP_GELU 1u
#define ISPK_EP_SILU 2u
#define ISPK_EP_MASK_ACC 4u
#define ISPK_EP_MASK_OUT 8u
#define ISPK_EP_BIAS_ROW 16u
#define ISPK_EP_MASK_COL 32u
#define ISPK_EP_OUT_BF16 64u   /* _bf16 entry only: C is bf16 (default fp32) */
#define ISPK_EP_RESID_BF16 128u /* _bf16 entry only: resid is bf16 (default fp32) */
#define ISPK_EP_DUAL_GELU 1024u /* ispk_gemm_bf16_gelu_train only: second output = dropout(gelu(C)) */
#define ISPK_EP_GELU_BWD 2048u  /* ispk_gemm_bf16_gelu_bwd only: C = (A W^T) gelu'(u) [dropout] */
#define ISPK_EP_OUT_SPLIT 512u  /* _split_f16 entry only: C is a pair of fp16 planes (hi at C, lo c_plane elements behind) */
#define ISPK_EP_ROWS_T 256u     /* _bf16 entry, K = 256 / 384, fp32 C, no resid: the M rows are [batch][T] frames with
                                   T = cols_per_batch and C is stored transposed per batch,
                                   C[(i / T) * batch_stride + j * ldc + (i % T)]  (bias by column j, mask by row i): the
                                   Linear + transpose(1, 2) of model.py:167-168 with frame-contiguous 128-B stores */

/* Which block tile ispk_gemm_f32 will use for (M, N, K): TM*10 + TN, block = 64*TM x 64*TN (22 -> 128x128).  Lets a
 * profiler label a launch with the kernel instance rocprof will report. */
int32_t ispk_gemm_f32_tile(int32_t M, int32_t N, int32_t K);
int32_t ispk_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, const float* bias,
                      const float* resid, int64_t ldr, const uint8_t* mask, int32_t M, int32_t N, int32_t K,
                      uint32_t flags, int32_t cols_per_batch, int64_t batch_stride, ispk_stream_t stream);
/* C[z] = A[z] . W[z]^T for z < batch (fp32, no epilogue): element strides stride_a / stride_w / stride_c between batch items.
 * The alignment gradient of the length regulator, d A[b] = d out[b] x[b]^T (temporal_adaptor.py:419-421), as ONE launch. */
int32_t ispk_gemm_f32_batched(const float* A, int64_t lda, int64_t stride_a, const float* W, int64_t ldw, int64_t stride_w,
                              float* C, int64_t ldc, int64_t stride_c, int32_t batch, int32_t M, int32_t N, int32_t K,
                              ispk_stream_t stream);
/* The feed-forward block's first Linear of a TRAINING step under autocast (feedforward.py:33-35: Linear -> GELU -> Dropout)
 * with both tensors the backward needs from ONE launch: u = A W^T (bf16, the pre-activation) and a = dropout(gelu(u)) (bf16;
 * the mask of ispk_gelu_bf16 with the same dropout_p / seed: hash(seed, row * N + feature)) - and its mirror in the backward:
 * du = (dY W2) gelu'(u) [keep / (1 - p)] straight from the GEMM's epilogue (row mask: dY's padded rows).  K = 256 / 384,
 * N % 8 == 0, contiguous-row bf16 tensors, 16-byte aligned. */
int32_t ispk_gemm_bf16_gelu_train(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, uint16_t* u, int64_t ldu,
                                  uint16_t* a, int64_t ld_a, int32_t M, int32_t N, int32_t K, float dropout_p, uint64_t seed,
                                  ispk_stream_t stream);
int32_t ispk_gemm_bf16_gelu_bwd(const uint16_t* dY, int64_t lddy, const uint16_t* W2t, int64_t ldw, const uint16_t* u,
                                int64_t ldu, uint16_t* du, int64_t lddu, const uint8_t* row_mask, int32_t M, int32_t N,
                                int32_t K, float dropout_p, uint64_t seed, ispk_stream_t stream);
/* Kernel instance dispatched by this thread's last ispk_gemm_bf16 call (profiler labels): 1000+KC panel<KC>,
 * 2000+10*TN+WM wide<TN,WM>, 3000+10*TM+TN generic<TM,TN>. */
int32_t ispk_gemm_bf16_last_variant(void);
int32_t ispk_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, void* C, int64_t ldc,
                       const float* bias, const void* resid, int64_t ldr, const uint8_t* mask, int32_t M, int32_t N,
                       int32_t K, uint32_t flags, int32_t cols_per_batch, int64_t batch_stride, ispk_stream_t stream);
/* Few rows, long reduction (a rank's share under strong scaling: the text-side stacks at 8 - 16 utterances per GPU): the same
 * product with K cut into `ksplit` slices that run as separate workgroups into fp32 slabs of `workspace` (ksplit * M * N
 * floats), then one pass adds the slabs in slice order and applies ispk_gemm_bf16's epilogue.  ispk_gemm_bf16_splitk_plan
 * returns the ksplit to use for a shape (1: call ispk_gemm_bf16).  Same reference sites as ispk_gemm_bf16
 * (feedforward.py:36 at K = inner, alignment.py:69-83 convolutions as GEMMs). */
int32_t ispk_gemm_bf16_splitk_plan(int32_t M, int32_t N, int32_t K, uint32_t flags);
int32_t ispk_gemm_bf16_splitk(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, void* C, int64_t ldc,
                              const float* bias, const void* resid, int64_t ldr, const uint8_t* mask, int32_t M, int32_t N,
                              int32_t K, uint32_t flags, float* workspace, int32_t ksplit, ispk_stream_t stream);

/* The whole feed-forward block in one kernel (bf16 operands, fp32 accumulation):
 *   out[i][:] = [mask[i]] * ( resid[i][:] + gelu_erf( x[i][:]·W1ᵀ + bias1 )·W2ᵀ + bias2 )
 * Replaces: feedforward.py:33-40 (Linear -> GELU -> Linear; W1 [inner][dim], W2 [dim][inner] as nn.Linear stores them)
 * with the residual add and row mask of transformer.py:105-110.  The [rows][inner] hidden activations never reach HBM.
 * x bf16 [rows][dim]; resid / out fp32 [rows][dim]; dim 256 or 384; inner % 32 == 0; flags: ISPK_EP_MASK_OUT (mask after
 * the residual add) or ISPK_EP_MASK_ACC (before).
 * ldw2 == 0: W2 is the packed image written by ispk_ffn_pack_w2_bf16 (every inner chunk of W2 one contiguous block in
 * fragment order: full-line loads and straight 16-byte LDS stores; ~2 % faster than the row-major layout). */
int32_t ispk_ffn_bf16(const uint16_t* x, int64_t ldx, const uint16_t* W1, int64_t ldw1, const float* bias1,
                      const uint16_t* W2, int64_t ldw2, const float* bias2, const float* resid, int64_t ldr,
                      const uint8_t* mask, float* out, int64_t ldo, int32_t rows, int32_t dim, int32_t inner, uint32_t flags,
                      ispk_stream_t stream);

/* The pre-norm feed-forward block of a transformer layer in one kernel, LayerNorm included:
 *   out[i][:] = [mask[i]] * ( x[i][:] + gelu_erf( LN(x[i][:])·W1ᵀ )·W2ᵀ + bias2 ),   LN = (x - mean_i) * rstd_i * gamma + beta
 * Replaces: transformer.py:101-110 (feed_forward_norm -> feed_forward -> residual add -> mask) = normalization.py:20-27 +
 * feedforward.py:33-40: the separate LayerNorm launch, its bf16 copy of the rows and that copy's re-read disappear.
 * x fp32 [rows][dim] is both the LayerNorm input and the residual; a wave owns whole rows and computes their statistics
 * itself (two-pass fp32, fixed summation order).  The reference also multiplies LN(x) by the row mask (:102); with the
 * same mask applied to the output (flags: ISPK_EP_MASK_OUT / ISPK_EP_MASK_ACC) that product cannot reach any kept value
 * and is skipped.  W2 packed (ispk_ffn_pack_w2_bf16), no first-Linear bias.  row_stats (optional): float [rows][2] =
 * (mean, rstd with stats_eps) of the OUTPUT rows, for ispk_gemm_bf16_lnin to apply the next LayerNorm while it stages them. */
int32_t ispk_ffn_bf16_prenorm(const float* x, int64_t ldx, const float* norm_gamma, const float* norm_beta, float norm_eps,
                              const uint16_t* W1, int64_t ldw1, const uint16_t* W2_packed, const float* bias2,
                              const uint8_t* mask, float* out, int64_t ldo, int32_t rows, int32_t dim, int32_t inner,
                              uint32_t flags, float* row_stats, float stats_eps, ispk_stream_t stream);

/* Second-generation kernel for the same block (dim 384 only; what the module mirror calls by default):
 *   out[i][:] = [mask[i]] * ( x[i][:] + gelu_erf( LN(x[i][:])·W1ᵀ )·W2ᵀ )          transformer.py:101-110 as above
 * 128 rows per workgroup on EIGHT waves, two per SIMD: the two waves that share a SIMD own the same 32 rows and split both
 * products (K-halves of the first, output-feature halves of the second), so one wave's GELU runs beside the other's
 * MFMAs (csrc/ffn2.hip).  GELU by Abramowitz-Stegun 7.1.27 (|gelu error| <= 2.5e-4 |x|, below the bf16 rounding of the
 * value it feeds).  W1 bf16 [inner][384] contiguous; W2_chunks = ispk_ffn_chunk_w2_bf16(W2): [inner/32][384][32], hidden
 * units in natural order; no biases.  flags / mask / row_stats as ispk_ffn_bf16_prenorm. */
int32_t ispk_ffn_chunk_w2_bf16(const uint16_t* W2, int64_t ldw2, int32_t dim, int32_t inner, uint16_t* out,
                               ispk_stream_t stream);
/* The same block for SMALL batches (the 6,400-row text encoder: 50 row blocks leave most of the 256 CUs idle, and every
 * workgroup streams all 2.4 MB of weights): the inner dimension is split over `splits` workgroups per row block
 * (ispk_ffn_bf16_prenorm2_split: parts[s][i][:] = gelu_erf(LN(x[i][:]) W1_s^T) W2_s^T, raw fp32, split s = hidden units
 * [s * inner / splits, (s + 1) * inner / splits)), then ispk_ffn_combine_ln_f32 adds them in split order:
 *   y[i][:] = [mask[i]] * (x[i][:] + sum_s parts[s][i][:])                              transformer.py:105-110
 *   (x may be NULL: the residual is then inside parts[0], ispk_attn_out_ffn_split_bf16)
 *   ln_out[i][:] = LN(y[i][:]) [* mask[i] if ln_mask]  (optional: the norm that consumes y - next layer's transformer.py:79
 *   or the final :205-206; fp32 or bf16), two-pass statistics.  parts: [splits] blocks at part_stride floats, rows x 384 each. */
int32_t ispk_ffn_bf16_prenorm2_split(const float* x, int64_t ldx, const float* norm_gamma, const float* norm_beta,
                                     float norm_eps, const uint16_t* W1, const uint16_t* W2_chunks, float* parts,
                                     int64_t part_stride, int32_t splits, int32_t rows, int32_t dim, int32_t inner,
                                     ispk_stream_t stream);
int32_t ispk_ffn_combine_ln_f32(const float* x, int64_t ldx, const float* parts, int64_t part_stride, int32_t splits,
                                const uint8_t* mask, float* y, int64_t ldy, const float* ln_gamma, const float* ln_beta,
                                float ln_eps, int32_t ln_mask, void* ln_out, int64_t ld_ln, int32_t ln_bf16, int32_t rows,
                                int32_t dim, ispk_stream_t stream);
int32_t ispk_ffn_bf16_prenorm2(const float* x, int64_t ldx, const float* norm_gamma, const float* norm_beta, float norm_eps,
                               const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask, float* out, int64_t ldo,
                               int32_t rows, int32_t dim, int32_t inner, uint32_t flags, float* row_stats, float stats_eps,
                               ispk_stream_t stream);
/* Second half of a pre-norm TransformerLayer in ONE kernel: the attention block's output projection and the feed-forward block,
 *   x1[i][:]  = x[i][:] + [mask[i] if ISPK_EP_MASK_ACC] * (attn_out[i][:] Wo^T)            attention.py:172-173, transformer.py:91
 *   out[i][:] = [mask[i] if ISPK_EP_MASK_OUT] * (x1[i][:] + gelu_erf(LN(x1[i][:]) W1^T) W2^T)   transformer.py:97-110
 * Replaces: `to_out` (attention.py:69, :172) + the residual add (transformer.py:91) + feed_forward_norm + FeedForward + the
 * second residual add and mask (transformer.py:97-110) - i.e. ispk_gemm_bf16(to_out, residual) followed by
 * ispk_ffn_bf16_prenorm2 - for decoder-sized batches with heads * 64 = dim = 384.  x1 never reaches memory: the row block's
 * residual rows are loaded INTO the accumulators of the kernel's second product, the projection is twelve more steps of that
 * product on top of them (Wo_chunks = ispk_ffn_chunk_w2_bf16(Wo): [384/32][384][32]; attn_out bf16 [rows][384] rows are the
 * B operands, zeroed for masked rows), the LayerNorm is taken from the accumulators, and the feed-forward block accumulates
 * onto them - one read of x, one of attn_out, one write of out per row (per layer: 50 + 25 + 50 MB at 32,768 rows instead
 * of 125 MB for the projection + 170 MB measured for the feed-forward kernel).  Everything else as ispk_ffn_bf16_prenorm2
 * (W1, W2_chunks, row_stats = (mean, rstd) of the output rows for the next layer's q/kv GEMM). */
int32_t ispk_attn_out_ffn_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                               const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta, float norm_eps,
                               const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask, float* out, int64_t ldo,
                               int32_t rows, int32_t dim, int32_t inner, uint32_t flags, float* row_stats, float stats_eps,
                               ispk_stream_t stream);

/* The same kernel with the NEXT layer's attention_norm and fused [to_q; to_kv] projection as its epilogue:
 *   qkv[i][:] = bf16( LN_next(out[i][:]) ) [Wq; Wkv]^T        transformer.py:79-80, attention.py:63-64 of the next TransformerLayer
 * Replaces, in addition: that layer's ispk_gemm_bf16_lnin launch (its re-read of the fp32 rows and 393 KB of weights per
 * workgroup) - between two decoder layers the residual stream is written once and read once.  The finished rows are normalised
 * from the registers that store them (two-pass statistics, next_eps), kept as a bf16 tile in LDS, and multiplied with the weight
 * streamed as 24 k-step chunks: Wqkv_chunks = ispk_chunk_k16_bf16([Wq; Wkv]): [384/16][512][16].  qkv bf16 [rows][512]
 * (6 heads x 64 query features, then 64 key and 64 value features), ld_qkv elements between rows.  No row_stats. */
/* ispk_attn_out_ffn_bf16 for the LAST layer of a stack: the stack's final LayerNorm (transformer.py:205-206, row-masked if ln_mask)
 * of the finished rows as a second output from the same registers, ln_out bf16 (ln_bf16) or fp32 [rows][dim] at ld_ln - the separate
 * LayerNorm launch and its re-read of the rows disappear; `out` may be NULL when only the normalised rows are consumed (the decoder:
 * its output goes through `norm` into to_mel, model.py:168-171) and the fp32 rows are then not stored at all. */
int32_t ispk_attn_out_ffn_norm_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                    const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta, float norm_eps,
                                    const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask, float* out, int64_t ldo,
                                    int32_t rows, int32_t dim, int32_t inner, uint32_t flags, const float* final_gamma,
                                    const float* final_beta, float final_eps, int32_t ln_mask, void* ln_out, int64_t ld_ln,
                                    int32_t ln_bf16, ispk_stream_t stream);
/* The SMALL-batch form (text encoder) of the projection prologue: ispk_ffn_bf16_prenorm2_split whose every split first forms
 * x1 = x + [mask if ISPK_EP_MASK_ACC] * (attn_out Wo^T) in its accumulators (each needs LN(x1)); split 0's partial product keeps
 * x1, the others start from zero:  parts[0] = x1 + ffn_0(LN(x1)),  parts[s] = ffn_s(LN(x1)).  ispk_ffn_combine_ln_f32 then runs
 * with x = NULL (y = [mask] * sum_s parts[s]).  Replaces the to_out GEMM launch of an encoder layer (attention.py:172-173,
 * transformer.py:91). */
int32_t ispk_attn_out_ffn_split_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                     const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta, float norm_eps,
                                     const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask, uint32_t flags,
                                     float* parts, int64_t part_stride, int32_t splits, int32_t rows, int32_t dim, int32_t inner,
                                     ispk_stream_t stream);
int32_t ispk_chunk_k16_bf16(const uint16_t* W, int64_t ldw, int32_t N, int32_t K, uint16_t* out, ispk_stream_t stream);
int32_t ispk_attn_out_ffn_qkv_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                   const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta, float norm_eps,
                                   const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask, float* out, int64_t ldo,
                                   int32_t rows, int32_t dim, int32_t inner, uint32_t flags, const float* next_gamma,
                                   const float* next_beta, float next_eps, const uint16_t* Wqkv_chunks, uint16_t* qkv,
                                   int64_t ld_qkv, ispk_stream_t stream);

/* Linear whose input is LayerNorm(x), with the row statistics supplied by the kernel that produced x:
 *   C[i][n] = epilogue( sum_k bf16( (x[i][k] - mean_i) * rstd_i * ln_gamma[k] + ln_beta[k] ) * W[n][k] )
 * Replaces: normalization.py:20-27 + the Linear that follows it (transformer.py:79-80, attention.py:63-64: attention_norm
 * -> to_q / to_kv) on the bf16 path - the separate LayerNorm launch, its re-read of the fp32 residual stream and the
 * bf16 copy it writes disappear.  x fp32 [M][K] (the residual stream), row_stats float [M][2] = (mean, rstd) from
 * ispk_ffn_bf16_prenorm / ispk_ffn_bf16_prenorm2, or NULL: the kernel computes the statistics itself (its waves
 * own whole rows; two-pass fp32 with ln_eps, fixed summation order) - then any LayerNorm -> Linear pair qualifies
 * (transformer.py:79-80 of a stack's first layer, :101-105 + feedforward.py:33 on the unfused path).
 * K 256 or 384; flags / bias / resid / mask / C as ispk_gemm_bf16 (row-major outputs). */
int32_t ispk_gemm_bf16_lnin(const float* x, int64_t ldx, const float* row_stats, const float* ln_gamma, const float* ln_beta,
                            float ln_eps, const uint16_t* W, int64_t ldw, void* C, int64_t ldc, const float* bias, const void* resid,
                            int64_t ldr, const uint8_t* mask, int32_t M, int32_t N, int32_t K, uint32_t flags,
                            ispk_stream_t stream);

/* One-time weight staging for ispk_ffn_bf16: W2 [dim][inner] (nn.Linear layout, feedforward.py:27) ->
 * packed [inner/32][dim][32], each chunk's 32 hidden units in MFMA accumulator-fragment order.  packed holds
 * dim*inner bf16 elements; inner % 32 == 0. */
int32_t ispk_ffn_pack_w2_bf16(const uint16_t* W2, int64_t ldw2, int32_t dim, int32_t inner, uint16_t* packed,
                              ispk_stream_t stream);

/* Small / odd-shaped Linear (any K, N): one thread per output, fp32 FMA chain in k order.
 * Replaces the tiny nn.Linear sites: embeddings.py:149-153 (time MLP 65->32->32), normalization.py:43-51 (AdaLN
 * condition projections 32->D), temporal_adaptor.py:43,98 (linear_layer D->3), transformer.py:170 with
 * emb_dim 2 / the 3 flow channels of the 387-wide predictor input.
 *   out[i][j] = act( sum_k a[i][k]*w[j][k] + bias[j] ) + resid[i][j];  act: 0 none, ISPK_EP_GELU, ISPK_EP_SILU */
int32_t ispk_linear_small_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias,
                              const float* resid, int64_t ldr, float* out, int64_t ldo, int32_t M, int32_t N, int32_t K,
                              uint32_t act, ispk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * ALiBi-biased multi-query attention (one shared K/V head, head_dim 64).
 * Replaces: modules/transformer/attend.py:49-122 (`Attend.efficient_attn`: SDPA with a materialised
 * [B,H,N,N] fp32 bias) together with the bias construction of embeddings.py:51-72 and the key-mask assembly of
 * attention.py:128-152.  Nothing of size N*N touches HBM here.
 *   out[b][i][h*64 + d] = sum_j softmax_j( q[b][i][h]·k[b][j] / 8 - slopes[h] * |i - j| ) * v[b][j][d],
 *   j restricted to j < key_len[b] (masked keys get weight exactly 0, as the reference's min/2 fill does)
 *   q   [B][N][H*64] leading stride ldq;  k, v [B][N][64] leading stride ldkv (k and v may alias one
 *   [B][N][128] `to_kv` buffer: k = kv, v = kv + 64);  out [B][N][H*64] leading stride ldo
 *   slopes [H] fp32 = exp(learned_logslopes) (embeddings.py:81-82);  key_len [B] int64 or NULL (= N)
 *   1 <= H <= 8.  Rows i >= key_len[b] are computed like the reference (finite values; zeroed later by the
 *   caller's row mask, attention.py:172).
 */
int32_t ispk_alibi_mqa_attn_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                const float* slopes, const int64_t* key_len, float* out, int64_t ldo, int32_t B,
                                int32_t N, int32_t H, ispk_stream_t stream);
int32_t ispk_alibi_mqa_attn_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, const uint16_t* v, int64_t ldkv,
                                 const float* slopes, const int64_t* key_len, uint16_t* out, int64_t ldo, int32_t B,
                                 int32_t N, int32_t H, ispk_stream_t stream);
/* Same kernel with the work split chosen by the caller: `q_tiles_per_workgroup` 64-query tiles of one batch item share
 * one K/V fetch (only while the whole key range fits the LDS ring, N <= 512; else 1).  0 = automatic (what
 * ispk_alibi_mqa_attn_bf16 uses: as many as still leave >= 256 workgroups).  Results are identical bit for bit. */
int32_t ispk_alibi_mqa_attn_bf16_tiles(const uint16_t* q, int64_t ldq, const uint16_t* k, const uint16_t* v, int64_t ldkv,
                                       const float* slopes, const int64_t* key_len, uint16_t* out, int64_t ldo, int32_t B,
                                       int32_t N, int32_t H, int32_t q_tiles_per_workgroup, ispk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Aligner front-end (the ConvAttention that produces the MAS input), fp32, channel-last padded layout.
 * Replaces: models/acoustic/modules/alignment.py:69-83 (ConvBlock1D), :159-208 (ConvAttention.forward), :18-37
 * (batch_diagonal_prior) and modules/normalization.py:160-208 (_masked_norm "instance").
 *
 * Layout: activations are [B][T+4][C] — channel-last, two zero rows before and after every utterance.  A Conv1d with
 * kernel 5 / padding 2 over such a buffer is ONE ispk_gemm_f32 call over overlapping rows (lda = C, K = 5*C,
 * M = B*(T+4) - 4, weight [O][5*C] = conv.weight.permute(0,2,1)); output row b*(T+4)+t is frame t.
 *
 * Both kernels read fp32 and write fp32 (out_bf16 == 0), bf16 (1: the bf16 throughput path then runs the conv GEMMs on
 * ispk_gemm_bf16 with fp32 outputs, so statistics and scores stay fp32) or split fp16 planes (2: hi plane at `out`, lo plane
 * B*(T+4)*C elements behind - the operand format of ispk_gemm_split_f16, the parity-grade fast path).
 * ispk_pad_rows_f32        out[b][t+2][c] = t < len[b] ? x[b*sb + t*st + c*sc] : 0, pad rows zero
 *                          (x*mask of alignment.py:75 plus the layout change; strides in elements, so a channel-first
 *                           mel [B][C][T] is read with st = 1, sc = T).
 * ispk_masked_instnorm_f32 y: conv output [B][T+4][C] (row t = frame t); per (b, c) mean / biased variance over frames
 *                          t < len[b]; out[b][t+2][c] = t < len[b] ? (y - mean)/sqrt(var + eps)*weight[c] + bias[c] : 0,
 *                          pad rows zero — i.e. the norm, the affine AND the next block's `x * input_mask`.
 * ispk_aligner_scores_f32  q_enc [B][..][128] (frame rows at q_stride_b per utterance), k_enc likewise (text rows):
 *                          S = q·k / sqrt(128) clamped to fp32 max; attn_logits = log_softmax(S over all L columns)
 *                          + log(prior + 1e-6) with the diagonal prior evaluated analytically (gamma 0.1, rows normalised
 *                          with +1e-5, entries < 1e-4 zeroed, zero outside the lengths); attn_soft = softmax over the
 *                          keys < text_len of attn_logits, zero for keys >= text_len and frames >= mel_len.
 *                          Both outputs [B][M][L] fp32 contiguous.  L <= 320, attention_dim 128.
 * ispk_soft_average_f32    TemporalAverager soft branch for pitch and energy plus log1p(duration)
 *                          (models/acoustic/modules/temporal_adaptor.py:257-269, :446-449):
 *                          feats[b][l] = { log1p(dur[b][l]), mask*sum_m pitch[b][m] A[b][m][l] / (sum_m A + 1e-5), same for
 *                          energy }, feats [B][L][3].  duration may be NULL (column 0 is then written as 0): the pitch /
 *                          energy targets need only attn_soft, so they can be had before MAS has finished.
 */
int32_t ispk_pad_rows_f32(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_c, const int64_t* len,
                          void* out, int32_t out_bf16, int32_t B, int32_t T, int32_t C, ispk_stream_t stream);
int32_t ispk_masked_instnorm_f32(const float* y, const float* weight, const float* bias, const int64_t* len, void* out,
                                 int32_t out_bf16, int32_t B, int32_t T, int32_t C, float eps, ispk_stream_t stream);
int32_t ispk_aligner_scores_f32(const float* q_enc, int64_t q_stride_b, const float* k_enc, int64_t k_stride_b,
                                const int64_t* text_len, const int64_t* mel_len, float* attn_logits, float* attn_soft,
                                int32_t B, int32_t M, int32_t L, int32_t D, ispk_stream_t stream);
/* The same operator for the bf16 compute path (its q_enc / k_enc carry the convolutions' bf16 rounding): score products as
 * three bf16 MFMAs over hi / lo splits of the fp32 operands (~2^-16 relative), v_exp_f32 / v_log_f32 instead of libm. */
int32_t ispk_aligner_scores_fast_f32(const float* q_enc, int64_t q_stride_b, const float* k_enc, int64_t k_stride_b,
                                const int64_t* text_len, const int64_t* mel_len, float* attn_logits, float* attn_soft,
                                int32_t B, int32_t M, int32_t L, int32_t D, ispk_stream_t stream);
int32_t ispk_soft_average_f32(const float* attn_soft, const float* pitch, const float* energy, const int64_t* duration,
                              const int64_t* text_len, float* feats, int32_t B, int32_t M, int32_t L,
                              ispk_stream_t stream);

/* Flow-matching algebra of the adaptor's predictor, FlowTransformerTemporalModule.forward (temporal_adaptor.py:120-147): [B][L][C]
 * fp32 tensors (C = 3 flow channels), t [B].
 * ispk_flow_mix_f32     x_t = (1 - (1 - sigma) t_b) x0 + t_b x1 ;  flow = x1 - (1 - sigma) x0      (:123-126; each operation
 *                       rounded once, in the reference's order)
 * ispk_flow_finish_f32  pf = pred_raw * mask ;  pred = (x0 + pf) * mask (:145) ;  duration = max(exp(pred[..., 0]) - 1, 0)
 *                       (the adaptor's duration estimate, :221 of this repo's mirror / temporal_adaptor.py:276) ;
 *                       loss_ratio[b] = sum over valid (l, c) of (pf - flow)^2 / max(C * valid_l, 1e-5): masked_mean of the
 *                       MSE (:146, utils/functions.py:44-58) before its final mean over the batch; loss_mean[0] (or NULL)
 *                       = that mean, the flow loss itself.  mask uint8 [B][L]. */
/* infer (temporal_adaptor.py:140-170, :351-384):
 * ispk_flow_euler_f32      one Euler step of FlowTransformerTemporalModule.infer: out = x_t + velocity * dt (:166-168; product and
 *                          sum each rounded once), times the [B][L] row mask when given (the `* mask` after the last step, :170).
 *                          dt is a host value: the warped time grid (:150-156) depends only on (steps, step_factor).
 * ispk_infer_features_f32  FlowTemporalAdaptor.infer between predictor and embedding stack (:351-384), pred [B][L][3]:
 *                          duration[b][l] = max(duration_factor * (exp(pred[..,0]) - 1), 0), replaced by the duration target
 *                          (fp32 or int64, at most one given) wherever that is >= 0 (:355-362); features [B][L][2] =
 *                          { (pitch_target | pred[..,1]) * pitch_factor + pitch_delta, (energy_target | pred[..,2]) *
 *                          energy_factor + energy_delta } (:366-381) - the embedding stack's input. */
/* ispk_flow_head_f32    what ends FlowTransformerTemporalModule.forward behind the stack's last layer, in one call: the stack's
 *                       final LayerNorm with its row mask (transformer.py:205-206) on the raw rows y [B][L][256], linear_layer
 *                       (256 -> 3, temporal_adaptor.py:98, :131) and ispk_flow_finish_f32's algebra (:133-147) - pred, duration
 *                       estimate, per-utterance loss ratios and their mean (loss_mean may be NULL).  workspace: 2 * B *
 *                       ceil(L / 16) floats (per-block partial sums, added in block order: deterministic). */
int32_t ispk_flow_head_f32(const float* y, int64_t ldy, const float* norm_gamma, const float* norm_beta, float norm_eps,
                           const float* W, const float* bias, const float* flow, const float* x0, const uint8_t* mask,
                           float* pred, float* duration, float* loss_ratio, float* loss_mean, float* workspace, int32_t B,
                           int32_t L, int32_t D, int32_t C, ispk_stream_t stream);
int32_t ispk_flow_euler_f32(const float* x_t, const float* velocity, float dt, const uint8_t* mask, float* out, int32_t B,
                            int32_t L, int32_t C, ispk_stream_t stream);
int32_t ispk_infer_features_f32(const float* pred, const float* duration_target_f32, const int64_t* duration_target_i64,
                                const float* pitch_target, const float* energy_target, float duration_factor, float pitch_factor,
                                float pitch_delta, float energy_factor, float energy_delta, float* duration, float* features,
                                int32_t B, int32_t L, ispk_stream_t stream);
int32_t ispk_flow_mix_f32(const float* x0, const float* x1, const float* t, float sigma, float* x_t, float* flow, int32_t B,
                          int32_t L, int32_t C, ispk_stream_t stream);
int32_t ispk_flow_finish_f32(const float* pred_raw, const float* flow, const float* x0, const uint8_t* mask, float* pred,
                             float* duration, float* loss_ratio, float* loss_mean, int32_t B, int32_t L, int32_t C,
                             ispk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The steps between the transformer stacks, one launch each instead of ATen glue / a library bmm.
 *
 * ispk_embed_tokens_f32     models/acoustic/model.py:131-134: emb[b][l][:] = table[text[b][l]][:] (nn.Embedding lookup, row
 *                           0 = padding row) and mask[b][l] = l < text_len[b] (utils/functions.py:61-65; mask / text_len may
 *                           be NULL).  Ids outside [0, vocab) read row 0 (the host-side F.embedding would raise).
 * ispk_add_speaker_f32      models/acoustic/model.py:93-97, :205-207 (`infer` of a multi-speaker model): x[b][l][:] +=
 *                           table[speaker[b * id_stride]][:] in place on the encoder output, EVERY row of the utterance (the
 *                           reference adds before any re-masking); id_stride 1 = one id per utterance (the collator's
 *                           [B, 1] field), 0 = one id for the batch (the notebook's `torch.tensor([speaker])`).  Ids outside
 *                           [0, speakers) are clamped (the host-side F.embedding would raise).
 * ispk_time_embedding_f32   modules/transformer/embeddings.py:131-157 with `with_steps` (temporal_adaptor.py:87-89):
 *                           f = [t, sin(t * freq_scale * inv_freq), cos(..)] (1 + 2*half_dim values),
 *                           out[n][:] = W1 silu(W0 f + b0) + b1; w0 [emb_dim][1 + 2*half_dim], w1 [emb_dim][emb_dim];
 *                           freq_scale is the module's 1-element buffer (device pointer: no host read).
 * ispk_length_regulate_f32  models/acoustic/modules/temporal_adaptor.py:411-436 (LengthRegulator, soft branch):
 *                           out[b][y][:] = sum_t A[b][y][t] * x[b][t][:], dec_len[b] = (sum_t dur[b][t] + 0.5).long()
 *                           clamped to max_len when max_len >= 0, dec_mask[b][y] = y < dec_len[b] (or NULL).
 *                           A = `alignment` fp32 [B][M][L] (forward: the aligner's attn_soft) or, with alignment NULL, the
 *                           soft path of :468-478 generated on the fly from the fp32 durations (infer, :388-397):
 *                           P[t][y] = clamp(cum[t] - y, 0, 1) - clamp(cum[t-1] - y, 0, 1), cum = cumsum(dur) in index order,
 *                           masked by t < enc_len[b] (NULL: L) and y < dec_len[b].  Exactly one of dur_f32 / dur_i64
 *                           is given: fp32 [B][L], or int64 [B][dur_cols] (the MAS durations, dur_cols = L; they are only
 *                           summed, so any [B][dur_cols] array with the same row sums serves - the teacher-forced forward
 *                           passes mel_len as [B][1], which equals the sum of the MAS durations by construction, and so
 *                           does not wait for MAS).  x fp32 [B][L][D] rows at stride ldx, D 256 / 384;
 *                           exact fp32 products (v_mfma_f32_32x32x2_f32). */
int32_t ispk_embed_tokens_f32(const int64_t* text, const float* table, int64_t ld_table, int32_t vocab,
                              const int64_t* text_len, float* emb, uint8_t* mask, int32_t B, int32_t L, int32_t D,
                              ispk_stream_t stream);
int32_t ispk_add_speaker_f32(float* x, const float* table, int64_t ld_table, int32_t speakers, const int64_t* speaker,
                             int32_t id_stride, int32_t B, int32_t L, int32_t D, ispk_stream_t stream);
int32_t ispk_time_embedding_f32(const float* t, int32_t n, const float* inv_freq, const float* freq_scale, int32_t half_dim,
                                const float* w0, const float* b0, const float* w1, const float* b1, int32_t emb_dim,
                                float* out, ispk_stream_t stream);
int32_t ispk_length_regulate_f32(const float* alignment, const float* dur_f32, const int64_t* dur_i64, const int64_t* enc_len,
                                 const float* x, int64_t ldx, float* out, int64_t* dec_len, uint8_t* dec_mask, int32_t B,
                                 int32_t M, int32_t L, int32_t D, int32_t max_len, int32_t dur_cols, ispk_stream_t stream);
/* The same operator for the bf16 compute path: each fp32 operand value is split into two bf16 terms in registers and a product
 * is three bf16 MFMAs (hi hi + hi lo + lo hi, ~2^-16 relative error) instead of eight exact fp32 ones. */
int32_t ispk_length_regulate_split_bf16(const float* alignment, const float* dur_f32, const int64_t* dur_i64, const int64_t* enc_len,
                                 const float* x, int64_t ldx, float* out, int64_t* dec_len, uint8_t* dec_mask, int32_t B,
                                 int32_t M, int32_t L, int32_t D, int32_t max_len, int32_t dur_cols, ispk_stream_t stream);
/* ... and for the split-fp16 parity path: the same three products over fp16 terms (22 significant bits per operand: fp32-grade). */
int32_t ispk_length_regulate_split_f16(const float* alignment, const float* dur_f32, const int64_t* dur_i64, const int64_t* enc_len,
                                const float* x, int64_t ldx, float* out, int64_t* dec_len, uint8_t* dec_mask, int32_t B,
                                int32_t M, int32_t L, int32_t D, int32_t max_len, int32_t dur_cols, ispk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Training step (BASELINE config 5): fp32 kernels with the recipes' dropout; under AMP the Linear GEMMs and
 * their weight gradients take bf16 operands.  The reference has no backward code of its own (autograd of the modules above);
 * what these replace is cited per entry.
 *
 * ispk_transpose_f32           y[c][r] = x[r][c]: weights for dX = dY . W through the NT GEMM (ispk_gemm_f32 wants W^T rows).
 * ispk_gemm_tn_f32             C[N1][N2] (+)= sum_m mask[m] A[m][N1-slice] B[m][N2-slice]: dW = dY^T . X of every nn.Linear
 *                              (autograd of F.linear).  Rows are split into ranges whose partial products go to `workspace`
 *                              (>= N1*N2 floats; more = more ranges, up to 256) and are added in range order: deterministic.
 *                              row_mask uint8 [M] or NULL; accumulate != 0 adds to C.  N1, N2, lda, ldb multiples of 4,
 *                              A and B 16-byte aligned; ldb < N2 is allowed (overlapping rows: the windows of a padded
 *                              convolution input, for the convolution's weight gradient).
 * ispk_gemm_tn_bf16            the same product, operands rounded to bf16 in flight (autocast's weight gradient), fp32 sums.
 * ispk_alibi_mqa_attn_train_bf16 / ispk_alibi_mqa_attn_bwd_bf16   the attention pair below on bf16 tensors (the step under autocast).
 * ispk_layernorm_bwd_f32       backward of modules/transformer/normalization.py:20-31 followed by `* mask` (transformer.py:102):
 *                              dx (=) or (+=, add_to_dx) rstd (g - mean(g) - xhat mean(g xhat)), g = dy mask gamma;
 *                              dgamma = sum_rows dy mask xhat, dbeta = sum_rows dy mask (either may be NULL; both NULL needs no
 *                              workspace, else ceil(rows / 64) * 2 * dim floats).  dim 256 or 384; statistics are recomputed
 *                              from x (two-pass, as the forward kernel).
 * ispk_gelu_f32                a = gelu(u), exact erf (modules/layers.py:29), as a pass of its own: the training forward keeps
 *                              the pre-activation u for the backward (the inference GEMMs apply GELU in their epilogue).
 * ispk_gelu_bwd_f32            du = da * (Phi(u) + u phi(u)): exact-erf GELU (modules/layers.py:29), n % 4 == 0.
 *                              Dropout (feedforward.py:35, nn.Dropout after the activation; attend.py:118, SDPA dropout_p on the
 *                              attention probabilities): with dropout_p > 0 ispk_gelu_f32 returns gelu(u) keep / (1 - p),
 *                              ispk_gelu_bwd_f32 routes the gradient through the same mask, ispk_alibi_mqa_attn_train_f32 is the
 *                              attention forward with dropped probabilities (it also returns the rows' log-sum-exp, which the
 *                              backward takes as lse_in instead of recomputing it) and ispk_alibi_mqa_attn_bwd_f32 differentiates
 *                              through the same mask.  keep = hash(seed, element index) >= p 2^32, a pure function of the seed
 *                              (element index modulo 2^32: flat index of u; ((b H + h) N + query) N + key for attention), so nothing is
 *                              stored; the draw sequence differs from torch's Philox stream (same Bernoulli(1 - p) law).
 *                              ispk_dropout_mask_u8 writes keep for indices 0 .. n-1 (tests).
 * ispk_alibi_mqa_attn_bwd_f32  backward of ispk_alibi_mqa_attn_f32 (attend.py:49-122, embeddings.py:51-82, attention.py:128-152):
 *                              qkv / dqkv fp32 [B][N][H*64 + 128] = [Q heads | K | V] at row stride ld_qkv, o / d_o fp32
 *                              [B][N][H*64] at ld_o, slopes [H] = exp(learned_logslopes), key_len int64 [B] or NULL.
 *                              dQ per head; dK, dV summed over the H heads that share them; dlogslopes[h] = slope_h *
 *                              sum_ij dS_ij (-|i - j|) (or NULL).  The softmax statistics are recomputed (no state is
 *                              kept from the forward).  workspace >= 2*B*H*N + H*B*ceil(N/32) floats.
 * ispk_mel_loss_f32            models/acoustic/loss.py:22-35 (MelLoss, weight folded into grad_out by the caller):
 *                              ratio[b] = sum over (c, t < mel_len[b]) of (out - target)^2 / max(C * len_b, 1e-5)
 *                              (utils/functions.py:44-58), loss[0] = mean_b ratio[b]; grad (or NULL) = d loss / d mel_out *
 *                              grad_out, zero on padded frames.  mel fp32 [B][C][T].
 * ispk_aligner_scores_bwd_f32  backward of ispk_aligner_scores_f32 (alignment.py:187-208) from d attn_soft and / or d attn_logits
 *                              (either may be NULL): the gradient of the UNSCALED q . k products, as dS [B][M][ld_s] and its
 *                              transpose dSt [B][L][ld_t] (padding columns must arrive zeroed) - the operands of the two batched
 *                              products d q_enc = dS k_enc, d k_enc = dS^T q_enc.  The diagonal prior is recomputed.
 * ispk_masked_instnorm_bwd_f32 backward of ispk_masked_instnorm_f32 (modules/normalization.py:160-208): y / d_out / d_y are
 *                              [B][T+4][C] with row t = frame t; d_y is zero past each utterance's length; d_weight / d_bias
 *                              summed over utterances in order.  workspace >= 2 B C floats.
 * ispk_soft_average_bwd_f32    d attn_soft (=) or (+=) from d feats [B][L][3] of ispk_soft_average_f32 (temporal_adaptor.py:446-449;
 *                              column 0, the log1p duration, carries no gradient).  workspace >= 3 B L floats.
 * ispk_flow_loss_bwd_f32       d (flow loss) / d pred_raw of ispk_flow_finish_f32 (temporal_adaptor.py:145-146): go 2 m (raw m - flow) /
 *                              (max(C n_b, 1e-5) B), n_b = valid positions of utterance b.
 * ispk_adaln_bwd_f32           backward of AdaptiveLayerNorm (normalization.py:37-61) as ispk_layernorm_f32 applies it with per-
 *                              utterance scale / shift rows: dx (=) or (+=), dscale[b][:] = sum_rows dy mask xhat, dshift[b][:] =
 *                              sum_rows dy mask (rows of utterance b: rows_per_batch consecutive rows); dim 256 / 384.
 * ispk_time_embedding_bwd_f32  backward of ispk_time_embedding_f32 for its four parameters (the time value gets no gradient):
 *                              dw0 [emb_dim][1 + 2 half_dim], db0, dw1 [emb_dim][emb_dim], db1 from d_out [n][emb_dim].
 * ispk_attn_ctc_loss_f32       models/acoustic/loss.py:39-77 (AttentionCTCLoss, weight folded into grad_out): attn_logits fp32
 *                              [B][M][L] -> a blank class with logit blank_logprob in front, log-softmax over the L + 1 classes,
 *                              nn.CTCLoss(blank 0, zero_infinity, reduction "mean") against the targets 1 .. text_len[b] with
 *                              mel_len[b] frames: loss[0] = mean_b nll_b / max(text_len[b], 1) (inf -> 0); grad (or NULL) =
 *                              d loss / d attn_logits * grad_out (what autograd gives through the pad and the log-softmax).
 *                              workspace >= B*M + B + 2*B*M*S floats, S = 2L + 1 rounded up to a multiple of 64.
 * ispk_attn_bin_loss_f32       models/acoustic/loss.py:90-107 (AttentionBinarizationLoss, weight folded into grad_out):
 *                              loss[0] = -sum over the cells of the hard alignment of log(clamp(attn_soft, eps)) / loss[1],
 *                              loss[1] = the number of such cells; attn_hard = the int16 one-hot MAS output [B][M][L].  grad
 *                              (or NULL; must arrive zeroed) = d loss / d attn_soft * grad_out.  workspace: 2048 floats.
 * ispk_mel_grad_rows_f32       first step of the backward of `to_mel` (Linear + transpose + mask, model.py:167-168):
 *                              g[(b, t)][c] = mask[b][t] * dmel[b][c][t], frames as rows for the two GEMMs that follow
 *                              (d dec = g W through ispk_gemm_f32 on W^T, dW = g^T dec through ispk_gemm_tn_f32).
 * ispk_colsum_f32              out[c] = sum_r [mask[r]] x[r][c] (bias gradients), fixed summation order; workspace >= 256 * cols floats.
 * ispk_smallk_wgrad_f32        out[n][k] = sum_r g[r][n] x[r][k], k < K <= 8: weight gradient of a Linear with a handful of input
 *                              features (the adaptor's 2 -> 256 embedding projection, transformer.py:170); workspace >= 256 N K.
 * ispk_embedding_bwd_f32       backward of nn.Embedding (model.py:131): d_table[v][:] = sum of d_emb rows whose id is v, in row order;
 *                              row padding_idx gets zeros.
 * ispk_grad_sqnorm_f32         out[0] = sum g[i]^2 over a flat gradient arena (what clip_grad_norm_ needs,
 *                              experiments/optimizers.py:236-237); partial = 2048 floats (8 KB, 8-byte aligned) of scratch;
 *                              fp64 accumulation in a fixed order.
 * ispk_adamw_f32               one torch.optim.AdamW step (optimizers.py:72-74; amsgrad off) over flat arenas p, g, m, v of n
 *                              floats.  Elements [0, n_decay) are the weight-decay group of optimizers.py:15-20 (tensors
 *                              with >= 2 non-unit dimensions): they get `weight_decay` and, when grad_sqnorm (device, the
 *                              sum of squares of that group's UNSCALED gradients) is given, the clip coefficient
 *                              min(1, max_norm / (grad_scale * sqrt(sqnorm) + 1e-6)) - the reference clips group 0 only.
 *                              Every gradient is first multiplied by grad_scale (1 / world size after a summing
 *                              reduce-scatter).  step >= 1 is the 1-based step count of the bias corrections. */
int32_t ispk_transpose_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t rows, int32_t cols,
                           ispk_stream_t stream);
int32_t ispk_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M,
                         int32_t N1, int32_t N2, const uint8_t* row_mask, int32_t accumulate, float* workspace,
                         int64_t workspace_floats, ispk_stream_t stream);
/* The same product with the fp32 operands rounded to bf16 on their way to the MFMAs, fp32 accumulation: the weight gradient of a
 * Linear under autocast (recipes/default.yaml:56; torch's Linear backward multiplies bf16 copies of dY and X). */
int32_t ispk_gemm_tn_bf16(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M,
                          int32_t N1, int32_t N2, const uint8_t* row_mask, int32_t accumulate, float* workspace,
                          int64_t workspace_floats, ispk_stream_t stream);
/* ... and with operands that ARE bf16 in memory (the activations an AMP step keeps in bf16 as their producers wrote them: half
 * the operand bytes of a kernel that is bound by them).  lda, ldb multiples of 4, A and B 8-byte aligned. */
int32_t ispk_gemm_tn_b16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t M,
                         int32_t N1, int32_t N2, const uint8_t* row_mask, int32_t accumulate, float* workspace,
                         int64_t workspace_floats, ispk_stream_t stream);
/* The same product for `batch` independent pairs (A_b, B_b) at element strides stride_a / stride_b, C_b at stride_c - e.g. the
 * backward of the length regulator (temporal_adaptor.py:419-421: out_b = A_b x_b): d x_b = A_b^T d out_b per utterance.
 * row_mask (or NULL) is [batch][M]; workspace >= batch * N1 * N2 floats. */
int32_t ispk_gemm_tn_batched_f32(const float* A, int64_t lda, int64_t stride_a, const float* B, int64_t ldb, int64_t stride_b,
                                 float* C, int64_t ldc, int64_t stride_c, int32_t batch, int32_t M, int32_t N1, int32_t N2,
                                 const uint8_t* row_mask, int32_t accumulate, float* workspace, int64_t workspace_floats,
                                 ispk_stream_t stream);
int32_t ispk_layernorm_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma,
                               const uint8_t* row_mask, float* dx, int64_t lddx, int32_t add_to_dx, float* dgamma,
                               float* dbeta, float* workspace, int64_t workspace_floats, int64_t rows, int32_t dim,
                               float eps, ispk_stream_t stream);
/* ispk_layernorm_bwd_f32 that also writes dx as bf16 rows (an AMP step: the next dX GEMM and weight gradient read that copy) */
int32_t ispk_layernorm_bwd_dual_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma,
                                    const uint8_t* row_mask, float* dx, int64_t lddx, int32_t add_to_dx, float* dgamma,
                                    float* dbeta, float* workspace, int64_t workspace_floats, int64_t rows, int32_t dim,
                                    float eps, uint16_t* dx_bf16, int64_t ld_dx_bf16, ispk_stream_t stream);
int32_t ispk_gelu_f32(const float* u, float* a, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream);
int32_t ispk_gelu_bwd_f32(const float* da, const float* u, float* du, int64_t n, float dropout_p, uint64_t seed,
                          ispk_stream_t stream);
/* The AMP step's forms (recipes/default.yaml:56: under autocast the feed-forward activation and its gradient are bf16 GEMM
 * operands): a / da / du in bf16 as their consumers take them, the pre-activation u stays fp32; same arithmetic, same mask. */
int32_t ispk_gelu_f32_bf16(const float* u, uint16_t* a, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream);
int32_t ispk_gelu_bwd_bf16(const uint16_t* da, const float* u, uint16_t* du, int64_t n, float dropout_p, uint64_t seed,
                           ispk_stream_t stream);
/* ... and with the pre-activation u stored in bf16 too (under autocast the first Linear's output is bf16) */
int32_t ispk_gelu_bf16(const uint16_t* u, uint16_t* a, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream);
int32_t ispk_gelu_bwd_b16(const uint16_t* da, const uint16_t* u, uint16_t* du, int64_t n, float dropout_p, uint64_t seed,
                          ispk_stream_t stream);
int32_t ispk_dropout_mask_u8(uint8_t* out, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream);
int32_t ispk_alibi_mqa_attn_train_f32(const float* qkv, int64_t ld_qkv, const float* slopes, const int64_t* key_len, float* o,
                                      int64_t ld_o, float* lse, int32_t B, int32_t N, int32_t H, float dropout_p, uint64_t seed,
                                      ispk_stream_t stream);
int32_t ispk_alibi_mqa_attn_bwd_f32(const float* qkv, int64_t ld_qkv, const float* o, const float* d_o, int64_t ld_o,
                                    const float* slopes, const int64_t* key_len, float* dqkv, float* dlogslopes,
                                    float* workspace, int64_t workspace_floats, int32_t B, int32_t N, int32_t H,
                                    const float* lse_in, float dropout_p, uint64_t seed, ispk_stream_t stream);
/* The attention of a step under autocast (recipes/default.yaml:56: SDPA and its backward see bf16 q / k / v / dO and return
 * bf16): qkv / o / d_o / dqkv are bf16 (uint16 bit patterns) in the layouts of the fp32 pair, products on bf16 MFMAs with
 * fp32 accumulation and statistics, K / V (forward, dQ) and the heads' Q / dO tiles (dK / dV) staged once per workgroup in
 * LDS (csrc/attention_train.hip).  lse fp32 [B][H][N] is written by the forward and read by the backward: use them as a pair,
 * with the same dropout_p and seed.  workspace: B H N + 2 H B ceil(N / 64) floats.  H <= 6, ld_qkv and ld_o multiples of 8. */
int32_t ispk_alibi_mqa_attn_train_bf16(const uint16_t* qkv, int64_t ld_qkv, const float* slopes, const int64_t* key_len,
                                       uint16_t* o, int64_t ld_o, float* lse, int32_t B, int32_t N, int32_t H, float dropout_p,
                                       uint64_t seed, ispk_stream_t stream);
int32_t ispk_alibi_mqa_attn_bwd_bf16(const uint16_t* qkv, int64_t ld_qkv, const uint16_t* o, const uint16_t* d_o, int64_t ld_o,
                                     const float* slopes, const int64_t* key_len, const float* lse, uint16_t* dqkv,
                                     float* dlogslopes, float* workspace, int64_t workspace_floats, int32_t B, int32_t N,
                                     int32_t H, float dropout_p, uint64_t seed, ispk_stream_t stream);
int32_t ispk_mel_loss_f32(const float* mel_out, const float* mel_target, const int64_t* mel_len, float* ratio, float* loss,
                          float* grad, float grad_out, int32_t B, int32_t C, int32_t T, ispk_stream_t stream);
int32_t ispk_aligner_scores_bwd_f32(const float* attn_logits, const float* attn_soft, const float* d_soft, const float* d_logits,
                                    const int64_t* text_len, const int64_t* mel_len, float* dS, int64_t ld_s, float* dSt,
                                    int64_t ld_t, int32_t B, int32_t M, int32_t L, float scale, ispk_stream_t stream);
int32_t ispk_masked_instnorm_bwd_f32(const float* y, const float* d_out, const float* weight, const int64_t* lengths, float* d_y,
                                     float* d_weight, float* d_bias, float* workspace, int64_t workspace_floats, int32_t B,
                                     int32_t T, int32_t C, float eps, ispk_stream_t stream);
int32_t ispk_soft_average_bwd_f32(const float* attn_soft, const float* pitch, const float* energy, const float* d_feats,
                                  const int64_t* text_len, float* workspace, int64_t workspace_floats, float* d_attn,
                                  int32_t accumulate, int32_t B, int32_t M, int32_t L, ispk_stream_t stream);
int32_t ispk_flow_loss_bwd_f32(const float* pred_raw, const float* flow, const uint8_t* mask, float grad_out, float* d_raw, int32_t B,
                               int32_t L, int32_t C, ispk_stream_t stream);
int32_t ispk_adaln_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* scale, int64_t ld_scale,
                           const uint8_t* row_mask, float* dx, int64_t lddx, int32_t add_to_dx, float* dscale, float* dshift,
                           int64_t ld_out, int32_t B, int32_t rows_per_batch, int32_t dim, float eps, ispk_stream_t stream);
int32_t ispk_time_embedding_bwd_f32(const float* t, int32_t n, const float* inv_freq, const float* freq_scale, int32_t half_dim,
                                    const float* w0, const float* b0, const float* w1, int32_t emb_dim, const float* d_out,
                                    float* dw0, float* db0, float* dw1, float* db1, ispk_stream_t stream);
int32_t ispk_attn_ctc_loss_f32(const float* attn_logits, const int64_t* text_len, const int64_t* mel_len, float blank_logprob,
                               float* workspace, int64_t workspace_floats, float* loss, float* grad, float grad_out, int32_t B,
                               int32_t M, int32_t L, ispk_stream_t stream);
int32_t ispk_attn_bin_loss_f32(const float* attn_soft, const int16_t* attn_hard, float eps, float* workspace, float* loss,
                               float* grad, float grad_out, int32_t B, int32_t M, int32_t L, ispk_stream_t stream);
int32_t ispk_mel_grad_rows_f32(const float* dmel, const uint8_t* mask, float* g, int32_t B, int32_t C, int32_t T,
                               ispk_stream_t stream);
int32_t ispk_colsum_f32(const float* x, int64_t ldx, int64_t rows, int32_t cols, const uint8_t* row_mask, float* workspace,
                        int64_t workspace_floats, float* out, ispk_stream_t stream);
int32_t ispk_smallk_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t rows, int32_t N, int32_t K,
                              float* workspace, int64_t workspace_floats, float* out, ispk_stream_t stream);
int32_t ispk_embedding_bwd_f32(const int64_t* ids, const float* d_emb, int64_t rows, int32_t dim, int32_t vocab,
                               int32_t padding_idx, float* d_table, int64_t ld_table, ispk_stream_t stream);
int32_t ispk_grad_sqnorm_f32(const float* g, int64_t n, float* partial, float* out, ispk_stream_t stream);
int32_t ispk_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int32_t step, const float* grad_sqnorm, float max_norm,
                       float grad_scale, ispk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The fp32-grade fast path ("split fp16"): every fp32 operand value v is carried as two fp16 terms, hi = fp16(v) and
 * lo = fp16(v - hi) (22 significant bits), and a product is three fp16 MFMAs, hi hi + hi lo + lo hi, fp32 accumulation:
 * 5.3x the rate of the exact-fp32 MFMAs at fp32-grade results (mel L-inf vs the reference stays < 1e-4; csrc/split.hip).
 * Replaces, on the parity path, the same reference call sites as ispk_gemm_f32 / ispk_alibi_mqa_attn_f32: every nn.Linear
 * of the stacks (attention.py:105,111,168; feedforward.py:33-36; transformer.py:170; model.py:167-168), the aligner's Conv1d
 * as a GEMM (alignment.py:69-83) and Attend.efficient_attn (attend.py:49-122).
 *
 * "Split planes": a [rows][cols] matrix as two fp16 matrices of the same leading stride, the lo plane `*_plane` ELEMENTS
 * behind the hi plane.  Domain |v| <= 65504 (values beyond are clamped).
 * ispk_split_f16            x fp32 [rows][cols] (stride ldx) -> hi / lo planes (stride ldy); cols % 4 == 0.
 * ispk_gemm_split_f16       C = epilogue(A W^T) as ispk_gemm_f32, A [M][K] and W [N][K] as split planes (lda < K allowed: the
 *                           sliding-window Conv1d view).  Epilogue: bias by column, GELU (A&S 7.1.28 erf, |err| <= 3e-7) /
 *                           SILU, MASK_ACC, fp32 residual, MASK_OUT; C fp32 row-major, or ISPK_EP_OUT_SPLIT: C is a pair of
 *                           planes (c_plane; no residual), or ISPK_EP_ROWS_T (cols_per_batch = T, batch_stride): rows are
 *                           [batch][T] frames and C[b][n][t] is written (Linear + transpose(1, 2) of model.py:167-168; bias
 *                           + MASK_OUT only).  K % 8 == 0, N % 4 == 0.
 * ispk_gemm_split_f16_tile  which tile ispk_gemm_split_f16 uses for (M, N, K): TN * 100 + WM * 10 + RT = 64 TN features x 32 WM RT rows.
 * ispk_layernorm_f32_split  ispk_layernorm_f32 with the result written as split planes (y_hi, lo y_plane behind).
 * ispk_alibi_mqa_attn_split_f16   ispk_alibi_mqa_attn_f32 on split terms: q / k / v fp32 as there (K / V are split once per
 *                           workgroup while staged, Q and the probabilities in registers); out fp32 [B][N][H*64] (o_plane
 *                           == 0) or split planes (uint16 elements, lo o_plane behind) for the out-projection GEMM. */
int32_t ispk_split_f16(const float* x, int64_t ldx, uint16_t* hi, uint16_t* lo, int64_t ldy, int32_t rows, int32_t cols,
                       ispk_stream_t stream);
int32_t ispk_gemm_split_f16_tile(int32_t M, int32_t N, int32_t K);
int32_t ispk_gemm_split_f16(const uint16_t* A, int64_t lda, int64_t a_plane, const uint16_t* W, int64_t ldw, int64_t w_plane,
                            void* C, int64_t ldc, int64_t c_plane, const float* bias, const float* resid, int64_t ldr,
                            const uint8_t* mask, int32_t M, int32_t N, int32_t K, uint32_t flags, int32_t cols_per_batch,
                            int64_t batch_stride, ispk_stream_t stream);
int32_t ispk_layernorm_f32_split(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* ada_scale,
                                 const float* ada_shift, int64_t ada_stride, int32_t rows_per_batch, const uint8_t* row_mask,
                                 uint16_t* y_hi, int64_t ldy, int64_t y_plane, int32_t rows, int32_t D, float eps,
                                 ispk_stream_t stream);
int32_t ispk_alibi_mqa_attn_split_f16(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                      const float* slopes, const int64_t* key_len, void* out, int64_t ldo, int64_t o_plane,
                                      int32_t B, int32_t N, int32_t H, ispk_stream_t stream);

/* fp32 -> bf16 conversion (round-to-nearest-even) of a [rows][cols] matrix; used to stage weights/activations. */
int32_t ispk_cast_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int32_t rows, int32_t cols,
                           ispk_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Data movement of a training step (csrc/util.hip), so that a step issues no PyTorch kernel: what the reference does with
 * autograd's AccumulateGrad, torch.cat / .to(bfloat16) on weights, zero fills and scalar algebra on the losses
 * (tts/experiments/trainer.py:538-579, tts/models/acoustic/loss.py:140-182).
 *
 * ispk_segments_f32     up to any number of contiguous fp32 segments in one call (32 per launch): mode 0 dst = src (a
 *                       concatenation), 1 dst += src (gradient delivery into an optimizer arena), 2 dst = bf16(src).
 * ispk_fill_zero        bytes of zeros (4-byte aligned buffer).
 * ispk_scale_f32        x *= s_dev[0] * s_host (s_dev NULL: s_host only): a loss gradient times the incoming scalar gradient.
 * ispk_sum_scalars_f32  out[0] = sum_i weights[i] * terms[i][0] in index order (n <= 8; weights NULL = ones): the total loss.
 * ispk_exp_pad_f32      dst[i] = exp(src[i]) (i < n), 0 (n <= i < total): ALiBi slopes from learned_logslopes.
 * ispk_sqrt_scale_f32   dst[i] = sqrt(src[i]) * scale: the clipped group's gradient norm from its square.
 * ispk_copy2d_f32       rows x cols with leading strides.
 * ispk_permute021_f32   dst[a][c][b] = src[a][b][c]: Conv1d weights [O][C][k] <-> GEMM weights [O][k][C].
 * ispk_conv_weight_flip_f32   wf[c][(K-1-k) O + o] = w[o][c][k]: the GEMM weight of a convolution's input gradient. */
/* A captured (HIP-graph) training step freezes its launch arguments; two things must still change from replay to replay:
 *   ispk_set_dropout_seed_source   while the PROCESS has a source (a DEVICE address of one uint64), every dropout
 *                                  kernel launched from any of its threads - the GELU / attention forward and backward pairs,
 *                                  the mask export - folds that word into its seed when it RUNS; the host rewrites the word
 *                                  between replays.  NULL switches it off.  Process-wide on purpose: an autograd engine
 *                                  launches the backward kernels from a worker thread, and they must draw the forward's masks.
 *   ispk_adam_args_f32 / ispk_adamw_f32_dev   the AdamW step's scalar factors (bias corrections of step t, lr, decay) as a
 *                                  40-byte record: computed on the host exactly as ispk_adamw_f32 computes them, copied to the
 *                                  device by the caller, read by the kernel when it runs. */
typedef struct { float f[10]; } ispk_adam_args_t;
int32_t ispk_set_dropout_seed_source(const uint64_t* device_word);
int32_t ispk_adam_args_f32(float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                           float grad_scale, ispk_adam_args_t* out);
int32_t ispk_adamw_f32_dev(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay,
                           const ispk_adam_args_t* args_dev, const float* grad_sqnorm, ispk_stream_t stream);

/* ispk_stage_weights   kernel-ready images of fp32 [rows][cols] parameters, 16 per launch: flags 1 = transposed (dst[c][r]: the
 *                       W^T rows the NT GEMM wants for dX = dY W), 2 = bf16 output, 4 = exp of the values; dst rows ld_dst
 *                       elements apart (so that [to_q; to_kv] lands in one fused image).  What the reference does with
 *                       torch.cat / .t() / autocast's weight casts after every optimizer step. */
typedef struct { const float* src; void* dst; int32_t rows, cols; int64_t ld_dst; int32_t flags; } ispk_stage_t;
int32_t ispk_stage_weights(const ispk_stage_t* segs, int32_t nseg, ispk_stream_t stream);
typedef struct { const float* src; void* dst; int64_t n; int32_t mode; } ispk_segment_t;
int32_t ispk_segments_f32(const ispk_segment_t* segs, int32_t nseg, ispk_stream_t stream);
int32_t ispk_fill_zero(void* p, int64_t bytes, ispk_stream_t stream);
int32_t ispk_scale_f32(float* x, int64_t n, const float* s_dev, float s_host, ispk_stream_t stream);
int32_t ispk_sum_scalars_f32(const float* const* terms, const float* weights, int32_t n, float* out, ispk_stream_t stream);
int32_t ispk_exp_pad_f32(const float* src, float* dst, int32_t n, int32_t total, ispk_stream_t stream);
int32_t ispk_sqrt_scale_f32(const float* src, float* dst, int32_t n, float scale, ispk_stream_t stream);
int32_t ispk_copy2d_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int32_t rows, int32_t cols,
                        ispk_stream_t stream);
int32_t ispk_permute021_f32(const float* src, float* dst, int32_t A, int32_t B, int32_t C, ispk_stream_t stream);
int32_t ispk_conv_weight_flip_f32(const float* w, float* wf, int32_t O, int32_t C, int32_t K, ispk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ISPK_H */
