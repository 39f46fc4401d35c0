/* libttsmi - C ABI of the MI355X-native (gfx950 / CDNA4) ForwardTransformer TTS hot path.
 *
 * The reference (as-ideas/TransformerTTS, Python on TensorFlow 2 + librosa) has NO plugin / FFI
 * interface: its boundary is the Python API
 * model/models.py:ForwardTransformer.{call,train_step,val_step,predict} and
 * data/audio.py:Audio.mel_spectrogram.  This header is the C ABI inserted UNDER that Python API:
 * each entry point replaces the TensorFlow / NumPy op(s) the reference dispatches to at the cited
 * file:line (paths relative to the reference root).  The host-side mirror of the Python API lives
 * in transformertts_amd/ and binds these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 (TTSMI_OK) or a negative TTSMI_ERR_* code; the message is
 *     available from ttsmi_last_error() (thread-local).  No exception crosses the ABI.
 *   - no allocation, no synchronisation: every buffer is a caller-owned DEVICE pointer, scratch is
 *     passed as (ws, ws_bytes) with a *_ws_bytes() query, every launch is asynchronous on the
 *     explicit stream (a hipStream_t passed as void*).  All entry points are therefore
 *     hipGraph-capturable.
 *   - state: compute entry points keep NO mutable state between calls.  The exceptions are all
 *     measurement / diagnostics and never change a result: the thread-local strings behind
 *     ttsmi_last_error() and ttsmi_last_kernel(); the process-wide profiling callback of
 *     ttsmi_set_launch_observer() (set it before the threads that launch, clear it after); and the
 *     TTSMI_* environment knobs, each read once into a function-local static const.
 *   - layout: row-major, channels-last [B, T, C]; matrices are [rows, cols] with a leading
 *     dimension in ELEMENTS.  Keras kernel layouts are kept: Dense [in, out], Conv1D [k, in, out].
 *   - dtype: TTSMI_F32 = everything fp32 (exact-fp32 MFMA, the 1e-4 parity path);
 *            TTSMI_BF16 = GEMM/attention operands rounded to bf16, fp32 accumulate, fp32 storage.
 */
#ifndef TTSMI_H
#define TTSMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point's argument list or the ttsmi_dense_block layout changes.  History: 101 round 3's `denom`
 * argument of ttsmi_l1_losses_weighted and the res16 / relu_bits tail of ttsmi_dense_block; 102 round 4; 103 ttsmi_mel_nnls
 * added; 104 the shifted-rows taps of ttsmi_hgemm_wgrad_rows (the conv weight gradient, which the conv stacks of the host
 * mirror call); 105 round 5 - ttsmi_dense_chain_* and the chain tail of ttsmi_dense_block; 106 the backward chain
 * (ttsmi_dense_chain_bwd*, chain_bw, relu_bits_layout); 107 round 6 - ttsmi_dense_chain_bwd_nparts,
 * ttsmi_dense_block_bwd_chained, ttsmi_ft_train_step; the opt-in one-pass attention backward
 * (ttsmi_attention_bwd_fused*, attn_fused_ws) removed - it never beat the two kernels inside the step.  Bindings check it at load time (transformertts_amd/_lib.py) so that a stale build is
 * refused instead of being called with shifted arguments. */
#define TTSMI_VERSION 109

enum {
    TTSMI_OK = 0,
    TTSMI_ERR_INVALID_ARG = -1,
    TTSMI_ERR_LAUNCH = -2,
    TTSMI_ERR_UNSUPPORTED = -3,
    TTSMI_ERR_WORKSPACE = -4
};
/* TTSMI_BF16_IO (attention entry points only): as TTSMI_BF16, and every activation the entry point
 * touches - qkv, ctx, dctx, dqkv - is bf16 in HBM (written by / feeding ttsmi_hgemm_tn and
 * ttsmi_hgemm_wgrad_rows, which take bf16 operands); lse and the delta scratch stay fp32. */
/* TTSMI_BF16X3 (the fp32 GEMM entry points: ttsmi_linear_* / ttsmi_conv1d_*): fp32 tensors as TTSMI_F32, every product as THREE
 * bf16 MFMAs on hi / lo splits of both operands (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate): ~2^-16 per product instead
 * of bf16's 2^-8, at ~4 x the exact-fp32 kernels' rate - the GEMM family of the host mirror's precision='bf16x3'. */
enum { TTSMI_F32 = 0, TTSMI_BF16 = 1, TTSMI_BF16_IO = 2, TTSMI_BF16X3 = 3 };

typedef void* ttsmi_stream_t; /* hipStream_t */

int ttsmi_version(void);
const char* ttsmi_last_error(void);
/* Diagnostics: the name of the kernel variant the most recent size-routed entry point on THIS thread launched
 * (ttsmi_hgemm_tn, ttsmi_hgemm_ln_fwd/_bwd, ttsmi_hgemm_wgrad_rows, ttsmi_hgemm_k256_split, the bf16 attention entry
 * points), e.g. "rowgemm_dma_kernel<0, 128>"; "" before the first such call.  Lets a parity test assert that it
 * exercised the variant a given launch size selects.  The string is a literal owned by the library. */
const char* ttsmi_last_kernel(void);

/* ---------------------------------------------------------------------------------------------
 * Dense layers.  Replaces tf.keras.layers.Dense at model/layers.py:93-94,116-120,479 and
 * model/models.py:410,422 (forward) and the tape.gradient of them (model/models.py:480).
 * y[M,N] = act([x | x2][M,K] . w[K,N] + bias).  x2 != NULL splits the K axis: columns [0,K1) come
 * from x, [K1,K) from x2 - this is Dense(concat([q_in, ctx])) of model/layers.py:148-149 without
 * the concat buffer.  bias may be NULL.  relu != 0 applies max(.,0).
 * ------------------------------------------------------------------------------------------- */
int ttsmi_linear_fwd(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int K1,
                     const void* w, int64_t ldw, const float* bias, void* y, int64_t ldy,
                     int M, int N, int K, int relu, int dtype, ttsmi_stream_t stream);
/* dx[M,K] (+)= (dy[M,N] . w[K,N]^T) * (relu_src > 0)   (relu_src NULL = no mask; accumulate != 0 adds
 * into dx - the sum of a multiply-used activation's gradients without a separate add pass) */
int ttsmi_linear_dgrad(const void* dy, int64_t lddy, const void* w, int64_t ldw,
                       const void* relu_src, int64_t ld_relu, void* dx, int64_t lddx,
                       int M, int N, int K, int accumulate, int dtype, ttsmi_stream_t stream);
/* dw[K,N] = x[M,K]^T . dy[M,N]  (deterministic split over M through ws),  db[N] = colsum(dy)
 * (db may be NULL). */
size_t ttsmi_linear_wgrad_ws_bytes(int M, int N, int K);
int ttsmi_linear_wgrad(const void* x, int64_t ldx, const void* dy, int64_t lddy, float* dw,
                       int64_t lddw, float* db, int M, int N, int K, void* ws, size_t ws_bytes,
                       int dtype, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Conv1D(kernel k, stride 1, padding 'same', channels-last) as an implicit GEMM over the k taps.
 * Replaces tf.keras.layers.Conv1D at model/layers.py:19-26,56-63,498-505.
 * x [B,T,Cin], w [k,Cin,Cout], y [B,T,Cout]; zero padding (k-1)/2 left, k/2 right.
 * ------------------------------------------------------------------------------------------- */
int ttsmi_conv1d_fwd(const void* x, const void* w, const float* bias, void* y, int B, int T,
                     int Cin, int Cout, int k, int relu, int dtype, ttsmi_stream_t stream);
/* dx = conv_transpose(dy, w) * (relu_src > 0) */
int ttsmi_conv1d_dgrad(const void* dy, const void* w, const void* relu_src, void* dx, int B, int T,
                       int Cin, int Cout, int k, int dtype, ttsmi_stream_t stream);
size_t ttsmi_conv1d_wgrad_ws_bytes(int B, int T, int Cin, int Cout, int k);
int ttsmi_conv1d_wgrad(const void* x, const void* dy, float* dw, float* db, int B, int T, int Cin,
                       int Cout, int k, void* ws, size_t ws_bytes, int dtype,
                       ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused scaled-dot-product self-attention (flash style: the [B,H,T,T] logits never reach HBM).
 * Replaces model/layers.py:123-129 (split heads), :176-195 (ScaledDotProductAttention) and
 * :144-147 (merge heads).
 * qkv  [B*T, 3*H*dh]: the fused Wq|Wk|Wv projection output, columns [q | k | v], head h at
 *      columns h*dh..(h+1)*dh of each third (so no head transpose is ever materialised).
 * key_pad [B,T] uint8, 1 = padded key: logits += -1e9 there, exactly like the reference's
 *      additive mask (model/layers.py:186-187), including fp32 absorption of the logit.
 * klen [B] int32: 1 + index of the last unpadded key (T if every key is padded) - lets the kernel
 *      skip key tiles that are entirely padded (their softmax weight is exactly 0 in fp32).
 * ctx  [B*T, H*dh] merged-head context;  lse [B,H,T] log-sum-exp of each row (for backward).
 * p_drop/seed/site: inverted dropout on the attention weights (model/layers.py:192).
 * ------------------------------------------------------------------------------------------- */
int ttsmi_attention_fwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx,
                        float* lse, int B, int H, int T, int dh, float p_drop, uint64_t seed,
                        const int64_t* step_dev, uint32_t site, int dtype, ttsmi_stream_t stream);
/* dqkv [B*T, 3*H*dh] from dctx; delta_ws: B*H*T floats of scratch. */
size_t ttsmi_attention_bwd_ws_bytes(int B, int H, int T, int dh);
int ttsmi_attention_bwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen,
                        const void* ctx, const void* dctx, const float* lse, void* dqkv, int B,
                        int H, int T, int dh, float p_drop, uint64_t seed, const int64_t* step_dev,
                        uint32_t site, void* ws, size_t ws_bytes, int dtype,
                        ttsmi_stream_t stream);
/* Dropout on the attention weights with PRECOMPUTED keep bits (bf16 MFMA kernels, dh 32/64/192; dtype TTSMI_BF16_IO =
 * bf16 tensors, TTSMI_BF16 = fp32 tensors): evaluating the
 * counter-based hash inside the attention inner loops costs a third of the forward and is repeated twice by the
 * backward; ttsmi_attention_dropmask evaluates the SAME keep(seed, step, site, row, key) decisions once per layer
 * and step into a bit table (uint64 [B*H][T/32 query tiles][T/32 key blocks][16], layout in attention_bf16.hip), and
 * the *_masked entry points read bits instead.  Results equal ttsmi_attention_fwd/bwd with the same
 * p_drop / seed / step / site up to fp32 rounding order (model/layers.py:192). */
size_t ttsmi_attention_dropmask_bytes(int B, int H, int T);
int ttsmi_attention_dropmask(void* mask, int B, int H, int T, float p_drop, uint64_t seed,
                             const int64_t* step_dev, uint32_t site, ttsmi_stream_t stream);
/* the tables of n layers of one stack (same B, H, T, rate, seed, step; masks[i] gets site sites[i]) in one launch; the
 * same bits as n ttsmi_attention_dropmask calls */
int ttsmi_attention_dropmask_stack(void* const* masks, const uint32_t* sites, int n, int B, int H, int T, float p_drop,
                                   uint64_t seed, const int64_t* step_dev, ttsmi_stream_t stream);
int ttsmi_attention_fwd_masked(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx,
                               float* lse, int B, int H, int T, int dh, float p_drop, const void* dropmask, int dtype,
                               ttsmi_stream_t stream);
int ttsmi_attention_bwd_masked(const void* qkv, const uint8_t* key_pad, const int32_t* klen,
                               const void* ctx, const void* dctx, const float* lse, void* dqkv, int B,
                               int H, int T, int dh, float p_drop, const void* dropmask, void* ws, size_t ws_bytes,
                               int dtype, ttsmi_stream_t stream);
/* Inference forward (no dropout, TTSMI_BF16_IO tensors) for launches too small to fill the GPU - batch 1, a few heads:
 * the keys are split over extra workgroups, each split writes a normalised partial context + log-sum-exp into `ws`, and a
 * combine pass forms the result (model/layers.py:176-195 with training=False).  _ws_bytes returns 0 when B*H*T already
 * fills the GPU; the call then runs the plain forward and ignores ws.  Results equal ttsmi_attention_fwd up to the bf16
 * rounding of the partial contexts. */
size_t ttsmi_attention_fwd_splitkeys_ws_bytes(int B, int H, int T, int dh);
int ttsmi_attention_fwd_splitkeys(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx, float* lse,
                                  int B, int H, int T, int dh, void* ws, size_t ws_bytes, ttsmi_stream_t stream);
/* Materialise the (post-dropout) attention weights [B,H,T,T] the reference returns from every call
 * (model/layers.py:195,302-310) - only when the caller asks for them. */
int ttsmi_attention_weights(const void* qkv, const uint8_t* key_pad, const float* lse,
                            float* weights, int B, int H, int T, int dh, float p_drop,
                            uint64_t seed, const int64_t* step_dev, uint32_t site, int dtype,
                            ttsmi_stream_t stream);
/* The same with the dropout decisions read from the layer's keep-bit table (ttsmi_attention_dropmask; dtype must be
 * TTSMI_BF16_IO): the hash costs ~23 vector instructions per weight, which makes the kernel compute-bound with dropout on. */
int ttsmi_attention_weights_masked(const void* qkv, const uint8_t* key_pad, const float* lse, float* weights, int B, int H,
                                   int T, int dh, float p_drop, const void* dropmask, int dtype, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused [dropout ->] residual add -> LayerNormalization(eps) [-> + s*PE] [-> dropout] [-> row mask]
 * Replaces tf.keras.layers.LayerNormalization at model/layers.py:27,96,207,295,508 together with
 * the element-wise ops around it: residual adds (:40,:102,:211), `x += s*PE` + dropout (:300-301),
 * `* dense_mask` (:229-230,:262-264), predictor dropout (:515,:523).
 *   z = keep_in(x) + res;  n = (z - mean)/sqrt(var + eps);  y = n*gamma + beta
 *   y += pe_scale[0] * pe[(row % T), :]     if pe != NULL
 *   y  = keep_out(y)                         if p_out > 0
 *   y  = 0 on rows with row_pad[row] != 0    if row_pad != NULL
 * mean/rstd [M] are saved for backward.  res, pe, row_pad may be NULL.
 * y_bf16 (may be NULL): a bf16 copy of y written by the same pass - the operand the consumer GEMMs
 * read (TTSMI_BF16 path: halves their A-side traffic; the fp32 y stays the residual stream).
 * Dropout masks are a pure function of (seed + step_dev[0], site, element index): backward
 * regenerates them, nothing is stored.  step_dev (may be NULL) is a DEVICE int64 step counter so
 * that a captured hipGraph draws fresh masks on every replay.
 * ------------------------------------------------------------------------------------------- */
int ttsmi_add_layernorm_fwd(const float* x, const float* res, const float* gamma,
                            const float* beta, const float* pe, const float* pe_scale, int T,
                            const uint8_t* row_pad, float p_in, uint32_t site_in, float p_out,
                            uint32_t site_out, uint64_t seed, const int64_t* step_dev, float eps,
                            float* y, float* mean, float* rstd, int M, int C, uint16_t* y_bf16,
                            ttsmi_stream_t stream);
/* dx (grad wrt x), dres (grad wrt res; may alias dx when p_in == 0; NULL if no res),
 * dgamma/dbeta [C] (both NULL = deferred, see ttsmi_layernorm_param_reduce_batched), dpe_scale [1] (NULL if
 * no pe).  relu_in != 0 additionally multiplies dx by
 * (x > 0) - the backward of the ReLU that produced x (predictor conv->relu->LN, layers.py:513).
 * Masks are regenerated, the forward output is not needed.
 * dx_bf16 (may be NULL): dx stored as bf16; when given, dx itself may be NULL (in the dense blocks
 * dx only feeds dgrad/wgrad GEMMs, the fp32 gradient stream continues through dres). */
size_t ttsmi_add_layernorm_bwd_ws_bytes(int M, int C);
int ttsmi_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma,
                            const float* mean, const float* rstd, const float* pe,
                            const float* pe_scale, int T, const uint8_t* row_pad, float p_in,
                            uint32_t site_in, float p_out, uint32_t site_out, uint64_t seed,
                            const int64_t* step_dev, int relu_in, float* dx, float* dres, float* dgamma, float* dbeta,
                            float* dpe_scale, int M, int C, void* ws, size_t ws_bytes,
                            uint16_t* dx_bf16, ttsmi_stream_t stream);
/* Deferred parameter gradients: called with dgamma == dbeta == NULL, ttsmi_add_layernorm_bwd leaves its
 * per-workgroup partial sums in `ws` (which the caller then keeps alive) and this entry finishes any number
 * of such calls in ONE launch - a training step has ~30 LayerNormalization instances (layers.py:27,96,207,
 * 295,508), whose dgamma/dbeta reductions are too small to be worth a launch each on the critical path.
 * HOST arrays of n device pointers / shapes: ws[i] = the workspace of call i, M[i], C[i] = its shape,
 * dgamma[i], dbeta[i] [C[i]], dpe_scale[i] [1] or NULL.  Same stream as (or ordered after) the bwd calls. */
int ttsmi_layernorm_param_reduce_batched(const void* const* ws, float* const* dgamma, float* const* dbeta,
                                         float* const* dpe_scale, const int* M, const int* C, int n,
                                         ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Small element-wise / gather ops of ForwardTransformer.call (model/models.py:518-550)
 * ------------------------------------------------------------------------------------------- */
/* models.py:521 + transformer_utils.py:24-26: key_pad[b,t] = (tok == 0); klen as above. */
int ttsmi_token_pad_mask(const int32_t* tokens, uint8_t* key_pad, int32_t* klen, int B, int T,
                         ttsmi_stream_t stream);
/* models.py:541 + transformer_utils.py:29-32 restated on lengths: key_pad[b,t] = (t >= len[b]). */
int ttsmi_length_pad_mask(const int32_t* len, uint8_t* key_pad, int32_t* klen, int B, int T,
                          ttsmi_stream_t stream);
/* models.py:522 Embedding lookup and its scatter-add backward (deterministic). */
int ttsmi_embedding_fwd(const int32_t* tokens, const float* table, float* y, int M, int V, int C,
                        ttsmi_stream_t stream);
int ttsmi_embedding_bwd(const int32_t* tokens, const float* dy, float* dtable, int M, int V, int C,
                        ttsmi_stream_t stream);
/* models.py:527-531: y = x + relu(p[m]*w[c] + b[c])  (Dense(d, relu) on a width-1 input). */
int ttsmi_pitch_embed_fwd(const float* x, const float* p, const float* w, const float* b, float* y,
                          int M, int C, ttsmi_stream_t stream);
size_t ttsmi_pitch_embed_bwd_ws_bytes(int M, int C);
int ttsmi_pitch_embed_bwd(const float* dy, const float* p, const float* w, const float* b,
                          float* dp, float* dw, float* db, int M, int C, void* ws, size_t ws_bytes,
                          ttsmi_stream_t stream);
/* layers.py:479,484-485: y[m] = act(x[m,:] . w + b) * (1 - row_pad[m])  (Dense(1) head). */
int ttsmi_rowdot_fwd(const float* x, const float* w, const float* b, const uint8_t* row_pad,
                     float* y, int M, int C, int relu, ttsmi_stream_t stream);
size_t ttsmi_rowdot_bwd_ws_bytes(int M, int C);
int ttsmi_rowdot_bwd(const float* dy, const float* y, const float* x, const float* w,
                     const uint8_t* row_pad, float* dx, float* dw, float* db, int M, int C,
                     int relu, void* ws, size_t ws_bytes, ttsmi_stream_t stream);
/* layers.py:482: y = x * (1 - row_pad[m]) (in place allowed). */
int ttsmi_rowmask_mul(const float* x, const uint8_t* row_pad, float* y, int M, int C,
                      ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Length regulator.  Replaces Expand.call model/layers.py:549-565 (tile + ragged boolean mask).
 * ttsmi_lenreg_index: dims = int32(round_half_even(dur)) (negative -> 0), cum[b, 0..Tp] =
 *   exclusive prefix sum, len[b] = sum, idx[b, j] = phoneme covering output frame j or -1 for
 *   j >= min(len[b], cap).  BIT-EXACT integer contract.  dur is f32 or i32 (dur_is_int).
 * ------------------------------------------------------------------------------------------- */
int ttsmi_lenreg_index(const void* dur, int dur_is_int, int32_t* idx, int32_t* cum, int32_t* len,
                       int B, int Tp, int cap, ttsmi_stream_t stream);
/* y[b,j,:] = idx[b,j] >= 0 ? x[b, idx[b,j], :] : 0 */
int ttsmi_lenreg_fwd(const float* x, const int32_t* idx, float* y, int B, int Tp, int cap, int C,
                     ttsmi_stream_t stream);
/* dx[b,i,:] = sum_{j in [cum[b,i], min(cum[b,i+1], cap))} dy[b,j,:]  (contiguous segment sum) */
int ttsmi_lenreg_bwd(const float* dy, const int32_t* cum, float* dx, int B, int Tp, int cap, int C,
                     ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Loss + optimiser.
 * ttsmi_l1_loss: utils/losses.py:41-49 with mask=None (= plain mean |t - p| over n elements),
 *   weighted as utils/losses.py:63-70.  loss_out[0] = mean|t-p| (unweighted); if grad != NULL,
 *   grad[i] = coeff * sign(p - t) / n.  target is f32 or i32 (target_is_int).
 *   pred is [rows, cols] with row stride ld_pred; target dense [rows, cols].
 * ttsmi_adam_tf: tf.keras.optimizers.Adam(lr, 0.9, 0.98, 1e-9) of
 *   utils/training_config_manager.py:102-106, applied at model/models.py:481:
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + eps), with
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed ON DEVICE from lr_dev[0] and t = step_dev[0] (the
 *   1-based iteration, i.e. optimizer.iterations after the increment) so that a captured hipGraph
 *   stays valid while lr and t change.  bf16_copy (may be NULL) receives
 *   the updated parameters rounded to bf16 for the TTSMI_BF16 GEMM path.
 * ------------------------------------------------------------------------------------------- */
size_t ttsmi_l1_loss_ws_bytes(int64_t n);
int ttsmi_l1_loss(const float* pred, int64_t ld_pred, const void* target, int target_is_int,
                  int64_t rows, int64_t cols, float coeff, float* grad, int64_t ld_grad,
                  float* loss_out, void* ws, size_t ws_bytes, ttsmi_stream_t stream);
/* utils/losses.py:63-70 (weighted_sum_losses) over n_terms <= 8 L1 terms in one call: term t is ttsmi_l1_loss with
 * coeff[t] (grad[t] may be NULL), losses_out[t] = its unweighted mean, and
 * total_out[0] = ((0 + coeff[0] losses[0]) + coeff[1] losses[1]) + ... in that order.  The arrays of pointers and
 * sizes are HOST arrays; pred / target / grad entries, losses_out, total_out and ws are device pointers.
 * denom (HOST array, may be NULL): denom[t] > 0 replaces rows[t]*cols[t] as the divisor of term t's mean and of its
 * gradient - the element count of the GLOBAL padded batch when the batch is sharded over data-parallel ranks, so that the
 * SUM of the ranks' losses / gradients is the single-device mean of utils/losses.py:41-49 (0 / NULL = the term's own count). */
size_t ttsmi_l1_losses_weighted_ws_bytes(int n_terms);
int ttsmi_l1_losses_weighted(int n_terms, const float* const* pred, const int64_t* ld_pred, const void* const* target,
                             const int32_t* target_is_int, const int64_t* rows, const int64_t* cols, const float* coeff,
                             const int64_t* denom, float* const* grad, const int64_t* ld_grad, float* losses_out,
                             float* total_out, void* ws, size_t ws_bytes, ttsmi_stream_t stream);
int ttsmi_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                  const int64_t* step_dev, float b1, float b2, float eps, uint16_t* bf16_copy,
                  ttsmi_stream_t stream);
/* step_dev[0] += 1 (so that the optimiser step and the dropout stream advance inside a graph). */
int ttsmi_step_increment(int64_t* step_dev, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * wav -> STFT -> mel -> log.  Replaces librosa.stft + librosa.feature.melspectrogram(S=|D|) +
 * MelGAN/WaveRNN normalisation at data/audio.py:72-92,209-242 as ONE batched kernel.
 * wav: concatenated float32 clips, clip c = wav[clip_off[c] .. clip_off[c+1]);
 * out: concatenated [frames, n_mels] float32, clip c at rows frame_off[c] .. frame_off[c+1]),
 *      frames_c = 1 + len_c / hop (center=True, reflect padding n_fft/2).
 * n_fft: 1024 (MelGAN / LJSpeech config) or 2048 (WaveRNN config, config/data_config_wavernn.yaml:16-23).
 * window [n_fft] float32 (periodic Hann of win_length centred in n_fft - built by the caller);
 * mel filterbank in CSR-by-row form: for mel m, bins mel_lo[m] .. mel_lo[m]+mel_cnt[m]) with
 * weights mel_w[mel_ptr[m] ..].  normalizer: 0 = MelGAN log(clip(S, clip_min)),
 * 1 = WaveRNN dB normalisation (min_level_db = -100, max_norm = 4).
 * ------------------------------------------------------------------------------------------- */
int ttsmi_stft_logmel(const float* wav, const int64_t* clip_off, const int64_t* frame_off,
                      int n_clips, int64_t total_frames, int n_fft, int hop, const float* window,
                      int n_mels, const int32_t* mel_lo, const int32_t* mel_cnt,
                      const int32_t* mel_ptr, const float* mel_w, int normalizer, float clip_min,
                      float* out, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Griffin-Lim phase reconstruction: the iteration loop of librosa.core.griffinlim (0.7.1, momentum form) that
 * data/audio.py:94-110 (reconstruct_waveform) runs on the CPU, as a GPU iSTFT / STFT loop.  n_fft = 1024.
 * mag     [T][513] fp32 linear-magnitude spectrogram, FRAME-major (librosa's [513, T] transposed);
 * angles  [T][513][2] fp32 unit phases (re, im): the start phases on entry (the caller draws them - the reference's
 *         are unseeded random), the final phases on return;
 * window  [1024] fp32: get_window('hann', win_length, fftbins=True) centred in n_fft (analysis = synthesis window);
 * wss     [n_fft + hop (T - 1)] fp32: librosa.filters.window_sumsquare of that window (the overlap-add envelope);
 * wav     [hop (T - 1)] fp32: istft(mag * angles) after n_iter iterations (center = True: n_fft / 2 trimmed each side).
 * ------------------------------------------------------------------------------------------- */
size_t ttsmi_griffinlim_ws_bytes(int T);
int ttsmi_griffinlim(const float* mag, float* angles, const float* window, const float* wss, int T, int n_fft, int hop,
                     int n_iter, float momentum, float* wav, void* ws, size_t ws_bytes, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * mel -> linear magnitudes, the step in front of Griffin-Lim: librosa.feature.inverse.mel_to_stft (0.7.1 [3P]) as
 * data/audio.py:94-110 (reconstruct_waveform) calls it - per frame, minimise 1/2 ||B x - m||^2 subject to x >= 0, from
 * the clipped least-squares start point.  One wave64 per frame runs an accelerated projected gradient on chip
 * (csrc/nnls.hip); librosa runs L-BFGS-B on the host.  The minimiser is unique in B x, not in x.
 * mel       [T][n_mels] fp32 amplitudes (de-normalised), frame-major;
 * pinv_t    [n_mels][n_bins] fp32: pinv(B) transposed;
 * row_lo / row_cnt / row_ptr [n_mels], w [n_w]: the filterbank's rows as runs of non-zeros (the layout ttsmi_stft_logmel
 *           takes);
 * x         [T][n_bins] fp32, frame-major (what ttsmi_griffinlim takes as `mag`): the solution raised to inv_power
 *           (1 / power of the mel: 1 for the reference's amplitude mels);
 * inv_lipschitz = 1 / ||B||_2^2 (the step), n_iter projected-gradient steps (300 pass scipy's objective; 512 is the
 *           host mirror's default).
 * ------------------------------------------------------------------------------------------- */
int ttsmi_mel_nnls(const float* mel, const float* pinv_t, const int* row_lo, const int* row_cnt, const int* row_ptr,
                   const float* w, int n_w, float* x, int T, int n_mels, int n_bins, float inv_lipschitz, int n_iter,
                   float inv_power, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * TTSMI_BF16 GEMM path: bf16 operands (round to nearest even), fp32 accumulate on
 * v_mfma_f32_32x32x16_bf16, fp32 results.  Same reference ops as the fp32 family above
 * (Dense / Conv1D forward, dgrad, wgrad); the callers provide K-contiguous operands:
 *
 * ttsmi_hgemm_tn:  c[M,N] (+)= act( sum_k a[m,k] * b[n,k] + bias ) * (relu_src > 0)   (accumulate adds into c)
 *   a: fp32 [M,K] (a_is_f32, converted while staging; optional second K segment a2 (same element type
 *      as a; bf16 needs K1 %% 64 == 0) from column K1;
 *      optional Conv1D window: conv_taps > 1 reads the contiguous window of conv_taps*conv_C values
 *      starting conv_pad frames before row m of a [B*conv_T, conv_C] activation, zero outside the
 *      sequence) or bf16 [M,K].   b: bf16 [N,K] - W^T for forward, W as stored for dgrad
 *      (rows = k_in), conv weights pre-laid-out by ttsmi_cast_transpose_bf16 / ttsmi_conv_wdgrad_layout_bf16.
 * ttsmi_hgemm_wgrad: dw[kin,n] = xT[kin,rows] . dyT[n,rows]^T (+ db[n] = row sums of dyT), operands
 *   produced by ttsmi_cast_transpose_bf16 with leading dimension ldt (multiple of 8, zero tail).
 * ttsmi_cast_transpose_bf16: dst[j*C + c][r] = bf16(src[r + j - pad][c]) (0 outside the sequence
 *   of length T when taps > 1); taps == 1 is a plain cast-transpose.  dst rows have ld_dst >= R.
 * ------------------------------------------------------------------------------------------- */
/* flags of ttsmi_hgemm_tn */
enum { TTSMI_GEMM_RELU = 1, TTSMI_GEMM_ACCUMULATE = 2, TTSMI_GEMM_OUT_BF16 = 4, TTSMI_GEMM_MASK_BF16 = 8 };
int ttsmi_hgemm_tn(const void* a, int a_is_f32, int64_t lda, const void* a2, int64_t lda2, int K1,
                   const uint16_t* b, int64_t ldb, const float* bias, const float* relu_src,
                   int64_t ld_relu, void* c, int64_t ldc, int M, int N, int K, int flags,
                   int conv_taps, int conv_T, int conv_C, int conv_pad, ttsmi_stream_t stream);
/* One launch for a K = 256 projection whose output columns go two ways (the output-projection dgrad of a dense block:
 * d(h) += d_o . Wo_top^T in fp32 and d(ctx) = d_o . Wo_ctx^T in bf16 share the operand d_o - model/layers.py:148-150
 * differentiated):  c_acc[M, :n_acc] += a[M, 256] . bt[:n_acc, 256]^T;  c_bf16[M, N - n_acc] = a . bt[n_acc:, 256]^T.
 * a / bt bf16, n_acc a multiple of 128, 16-byte aligned operands and row pitches. */
int ttsmi_hgemm_k256_split(const void* a, int64_t lda, const uint16_t* bt, int64_t ldb, float* c_acc, int64_t ldc_acc,
                           int n_acc, void* c_bf16, int64_t ldc_bf16, int M, int N, ttsmi_stream_t stream);
size_t ttsmi_hgemm_wgrad_ws_bytes(int rows, int kin, int n);
int ttsmi_hgemm_wgrad(const uint16_t* xT, const uint16_t* dyT, int64_t ldt, float* dw, int64_t lddw,
                      float* db, int rows, int kin, int n, void* ws, size_t ws_bytes,
                      ttsmi_stream_t stream);
/* dw[kin,n] = x[rows,kin]^T . dy[rows,n] (+ db) straight from the row-major fp32 tensors (operands are
 * rounded to bf16 and transposed inside the kernel).  Conv1D wgrad: conv_taps > 1, x is [B*conv_T,
 * conv_C] with conv_C %% 128 == 0 and kin = conv_taps*conv_C (fp32 x, windows clipped at the sequence ends).
 * conv_taps > 1 with conv_T == 0: SHIFTED-ROWS taps for the zero-margin layout of a 'same' Conv1D (every sequence
 * carries its own zero rows, so no clipping): rows [j conv_C, (j + 1) conv_C) of dw = x[j : j + rows]^T . dy, all taps in
 * one launch; bf16 x and dy, ldx == conv_C, conv_pad == 0, conv_C %% 128 == 0, n %% 128 == 0, rows %% 32 == 0, and x
 * readable for conv_taps - 1 rows past `rows`. */
size_t ttsmi_hgemm_wgrad_rows_ws_bytes(int rows, int kin, int n);
int ttsmi_hgemm_wgrad_rows(const void* x, int x_is_bf16, int64_t ldx, const void* dy, int dy_is_bf16, int64_t lddy,
                           float* dw, int64_t lddw, float* db, int rows, int kin, int n, int conv_taps,
                           int conv_T, int conv_C, int conv_pad, void* ws, size_t ws_bytes,
                           ttsmi_stream_t stream);
int ttsmi_cast_transpose_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int R,
                              int C, int taps, int T, int pad, ttsmi_stream_t stream);
/* Batched cast-transpose (the per-step refresh of every GEMM weight's bf16 W^T in one launch).
 * desc_dev: DEVICE array of n_desc descriptors; descriptor i owns the 64x64 tiles
 * [tile_start, tile_start + tiles_r * ceil(C/64)), tiles_r = ceil(ld_dst/64), ordered by tile_start;
 * dst[c*ld_dst + r] = bf16(src[r*ld_src + c]) for r < R (0 for R <= r < ld_dst), c < C. */
typedef struct {
    const float* src;
    uint16_t* dst;
    int64_t ld_src, ld_dst;
    int32_t R, C, tile_start, tiles_r;
} ttsmi_transpose_desc;
int ttsmi_cast_transpose_bf16_batched(const ttsmi_transpose_desc* desc_dev, int n_desc, int total_tiles,
                                      ttsmi_stream_t stream);
/* Conv1D dgrad operand: dst[ci][jp*cout_ld + co] = bf16(w[k-1-jp][ci][co]) (0 for Cout <= co < cout_ld);
 * cout_ld > Cout pads an output-channel count that is not a multiple of 8 (predictor filters 226)
 * so that the dgrad runs as a bf16 implicit GEMM over a zero-padded dy. */
int ttsmi_conv_wdgrad_layout_bf16(const float* w, uint16_t* dst, int k, int Cin, int Cout, int cout_ld,
                                  ttsmi_stream_t stream);

/* Utility: fp32 -> bf16 (round to nearest even) for the TTSMI_BF16 weight copies. */
int ttsmi_cast_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Full-row bf16 GEMMs (N = 256 output columns = the model width of BASELINE.json configs[1]) with the
 * neighbouring LayerNormalization fused into the epilogue - csrc/rowgemm.hip.  One 64 x 256 workgroup tile
 * holds complete rows, so the pre-norm tensor never reaches HBM and the LayerNorm costs no launch.
 *
 * ttsmi_hgemm_ln_fwd:  z = keep_in(a . bt^T + bias) + res;  x^ = (z - mean) * rstd;  y = rowmask(x^ * gamma + beta)
 *   = Dense + dropout + residual add + LayerNormalization + dense_mask (model/layers.py:148-150,211,229 for the
 *   attention output projection, :100-102,230 for the second FFN layer).  a bf16 [M,K] (+ second K segment a2 from
 *   column K1, for concat([q_in, ctx])), bt = W^T bf16 [256,K].  Writes y fp32, y as bf16 (the next GEMM's operand),
 *   x^ as bf16 and rstd [M] for the backward (mean is not needed again).
 * ttsmi_hgemm_ln_bwd:  dy = dy_part + a . bt^T;  g = rowmask(dy);  t = g * gamma;
 *   dz = rstd * (t - mean(t) - x^ * mean(t * x^));  dx = keep_in(dz) as bf16, dres = dz fp32.
 *   = the dgrad GEMM whose result completes the gradient of a LayerNorm output, followed by that LayerNorm's backward.
 *   The parameter gradients dgamma = sum g * x^, dbeta = sum g leave as one partial row per workgroup in part_ws
 *   (ttsmi_hgemm_ln_bwd_nparts(M) rows; ttsmi_layernorm_partials_bytes) for ttsmi_layernorm_param_reduce_batched_nw.
 * ttsmi_layernorm_bwd_xhat: the same backward for an upstream gradient that is already complete (dy fp32 [M,256]);
 *   ttsmi_layernorm_bwd_xhat_nparts(M) partial rows.
 * ------------------------------------------------------------------------------------------- */
int ttsmi_hgemm_ln_fwd(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                       int64_t ldb, const float* bias, const float* res, const float* gamma, const float* beta,
                       const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                       float eps, float* y, uint16_t* y_bf16, uint16_t* xhat_bf16, float* rstd, int M, int N, int K,
                       ttsmi_stream_t stream);
size_t ttsmi_layernorm_partials_bytes(int nparts, int C);
int ttsmi_hgemm_ln_bwd_nparts(int M);
int ttsmi_hgemm_ln_bwd(const uint16_t* a, int64_t lda, const uint16_t* bt, int64_t ldb, const float* dy_part,
                       const uint16_t* xhat_bf16, const float* rstd, const float* gamma, const uint8_t* row_pad, float p_in,
                       uint32_t site_in, uint64_t seed, const int64_t* step_dev, uint16_t* dx_bf16, float* dres,
                       void* part_ws, size_t part_ws_bytes, int M, int N, int K, ttsmi_stream_t stream);
/* bf16-residual forms (see ttsmi_dense_block.res16).  _fwd_h: the residual is bf16 [M,256] and y (fp32) may be NULL.
 * _bwd_dual_h: dy_part is bf16; dres is bf16 [M,256] when dres_is_bf16 != 0, else fp32.  _bwd_xhat_h: dres is bf16. */
int ttsmi_hgemm_ln_fwd_h(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                         int64_t ldb, const float* bias, const uint16_t* res_bf16, const float* gamma, const float* beta,
                         const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                         float eps, float* y, uint16_t* y_bf16, uint16_t* xhat_bf16, float* rstd, int M, int N, int K,
                         ttsmi_stream_t stream);
int ttsmi_hgemm_ln_bwd_dual_h(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                              int64_t ldb, const uint16_t* bt2, int64_t ldb2, const uint16_t* dy_part_bf16,
                              const uint16_t* xhat_bf16, const float* rstd, const float* gamma, const uint8_t* row_pad, float p_in,
                              uint32_t site_in, uint64_t seed, const int64_t* step_dev, uint16_t* dx_bf16, void* dres,
                              int dres_is_bf16, void* part_ws, size_t part_ws_bytes, int M, int N, int K, ttsmi_stream_t stream);
int ttsmi_layernorm_bwd_xhat_h(const float* dy, const uint16_t* xhat_bf16, const float* rstd, const float* gamma,
                               const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                               uint16_t* dx_bf16, uint16_t* dres_bf16, void* part_ws, size_t part_ws_bytes, int M, int C,
                               ttsmi_stream_t stream);
/* The same with a second K segment that has its own operands: dy = dy_part + a . bt^T + a2 . bt2^T (a [M,K1], bt [256,K1];
 * a2 [M,K-K1], bt2 [256,K-K1]; K1 % 64 == 0).  The block backward uses it to complete the gradient of a block's input in
 * one pass over it: dqkv . Wqkv^T and d_o . Wo[:d]^T (the `q_in` half of Dense(concat([q_in, ctx])), model/layers.py:148-149)
 * arrive in the same accumulators instead of the second product being added to the fp32 tensor by a kernel of its own. */
int ttsmi_hgemm_ln_bwd_dual(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                            int64_t ldb, const uint16_t* bt2, int64_t ldb2, const float* dy_part, const uint16_t* xhat_bf16,
                            const float* rstd, const float* gamma, const uint8_t* row_pad, float p_in, uint32_t site_in,
                            uint64_t seed, const int64_t* step_dev, uint16_t* dx_bf16, float* dres, void* part_ws,
                            size_t part_ws_bytes, int M, int N, int K, ttsmi_stream_t stream);
int ttsmi_layernorm_bwd_xhat_nparts(int M);
int ttsmi_layernorm_bwd_xhat(const float* dy, const uint16_t* xhat_bf16, const float* rstd, const float* gamma,
                             const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                             uint16_t* dx_bf16, float* dres, void* part_ws, size_t part_ws_bytes, int M, int C,
                             ttsmi_stream_t stream);
/* number of partial rows ttsmi_add_layernorm_bwd leaves in its workspace for M rows */
int ttsmi_add_layernorm_bwd_nparts(int M);
/* ttsmi_layernorm_param_reduce_batched with the partial-row count of every item given explicitly */
int ttsmi_layernorm_param_reduce_batched_nw(const void* const* ws, float* const* dgamma, float* const* dbeta,
                                            float* const* dpe_scale, const int* nparts, const int* C, int n,
                                            ttsmi_stream_t stream);

/* The FFN's ReLU (model/layers.py:99) as a bit matrix for the backward, K = 256 GEMMs only (TTSMI_ERR_UNSUPPORTED when
 * ttsmi_hgemm_k256_eligible(M, N, 256) is 0).  bits: ttsmi_relu_bits_bytes(M, N) bytes (8-byte aligned), one bit per
 * element in the order the storing threads of the kernel hold a tile - an opaque hand-over between these two calls at the
 * SAME (M, N), not a row-major matrix.
 *   _relu_bits:   c = relu(a . bt^T + bias) as bf16, and bits = (c > 0)      (a [M,256], bt [N,256] bf16)
 *   _masked_bits: c = (a . bt^T) where the bit is set, 0 elsewhere, as bf16 (the ReLU' of the FFN2 dgrad) */
size_t ttsmi_relu_bits_bytes(int M, int N);
int ttsmi_hgemm_k256_relu_bits(const uint16_t* a, int64_t lda, const uint16_t* bt, int64_t ldb, const float* bias,
                               uint16_t* c, int64_t ldc, uint8_t* bits, int M, int N, ttsmi_stream_t stream);
int ttsmi_hgemm_k256_masked_bits(const uint16_t* a, int64_t lda, const uint16_t* bt, int64_t ldb, const uint8_t* bits,
                                 uint16_t* c, int64_t ldc, int M, int N, ttsmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * One SelfAttentionDenseBlock (model/layers.py:214-230: MultiHeadAttention + two res-norms + FFN) per call,
 * TTSMI_BF16 path: the forward enqueues its 8 launches, the backward its 9 main-stream launches and 5 weight
 * gradients (second stream) from C++.  Driving the same launches one by one from Python costs ~14 us of host
 * time each - with ~400 launches a step the HOST, not the GPU, bounded the train step (6.1 ms of enqueueing for
 * 5.0 ms of GPU work); a descriptor is filled once per (block, batch shape) and a step then costs two calls
 * per block.  Every pointer is a caller-owned device buffer with the element type named in the comment;
 * nothing is allocated, nothing synchronises.  M = B*T rows, d = H*dh model width, F = FFN width.
 * ------------------------------------------------------------------------------------------- */
typedef void* ttsmi_event_t; /* hipEvent_t */
typedef struct ttsmi_dense_block {
    int32_t B, H, T, d, F;
    float rate;                                   /* dropout rate of the block (0 = inference) */
    uint32_t site_attn, site_ln1, site_ln2;       /* dropout sites, in the order the per-layer path draws them */
    uint64_t seed;
    const int64_t* step_dev;                      /* device step counter of the dropout stream */
    const uint8_t* pad;                           /* [B,T] 1 = padded row / key */
    const int32_t* klen;                          /* [B] */
    const void* dropmask;                         /* keep bits of the attention dropout (ttsmi_attention_dropmask) or NULL */
    /* parameters: fp32 vectors, bf16 matrices in both operand layouts (forward W^T [N][K], dgrad W as stored [K][N]) */
    const float *bqkv, *bo, *ln1_g, *ln1_b, *b1, *b2, *ln2_g, *ln2_b;
    const uint16_t *wqkv_t, *wo_t, *w1_t, *w2_t;
    const uint16_t *wqkv_b, *wo_b, *w1_b, *w2_b;
    /* gradient sinks (fp32, Keras layouts: wqkv [d,3d], wo [2d,d], w1 [d,F], w2 [F,d]) */
    float *g_wqkv, *g_bqkv, *g_wo, *g_bo, *g_ln1_g, *g_ln1_b, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln2_g, *g_ln2_b;
    /* activations kept for the backward */
    uint16_t *qkv /*[M,3d]*/, *cx /*[M,d]*/, *a_bf /*[M,d]*/, *h1 /*[M,F]*/, *out_bf /*[M,d]*/;
    float *lse /*[B,H,T]*/, *o /*[M,d]*/, *a /*[M,d]*/, *f /*[M,d]*/, *out /*[M,d]*/;
    float *mean1, *rstd1, *mean2, *rstd2;         /* [M] */
    /* fuse_ln != 0 (needs d == 256): the two res-norms run in the epilogues of the o-projection / FFN2 GEMMs
     * (ttsmi_hgemm_ln_fwd) and res-norm 1's backward in the epilogue of the FFN1 dgrad (ttsmi_hgemm_ln_bwd); o, f,
     * mean1, mean2, ln_ws1, ln_ws2 are then unused and these are required instead: */
    int32_t fuse_ln;
    int32_t attn_split;                           /* != 0 and rate == 0: the forward may split the keys
                                                     (ttsmi_attention_fwd_splitkeys) with attn_ws as its scratch */
    uint16_t *xhat1, *xhat2;                      /* [M,d] normalised pre-activations kept for the backward */
    void *lnp_ws1, *lnp_ws2;                      /* parameter-gradient partial rows of res-norm 1 (ttsmi_hgemm_ln_bwd) /
                                                     res-norm 2 (ttsmi_layernorm_bwd_xhat) */
    uint64_t lnp_ws1_bytes, lnp_ws2_bytes;
    /* backward temporaries */
    uint16_t *df /*[M,d]*/, *dh1 /*[M,F]*/, *d_o /*[M,d]*/, *dctx /*[M,d]*/, *dqkv /*[M,3d]*/;
    float *da /*[M,d]*/, *dh /*[M,d]: the block's input gradient (output of the backward) */;
    void *attn_ws, *ln_ws1, *ln_ws2, *wgrad_ws;   /* ttsmi_attention_bwd_ws_bytes / add_layernorm_bwd_ws_bytes /
                                                     the largest ttsmi_hgemm_wgrad_rows_ws_bytes of the block */
    uint64_t attn_ws_bytes, ln_ws_bytes, wgrad_ws_bytes;
    ttsmi_stream_t main_stream, side_stream;      /* side_stream == NULL: weight gradients on the main stream */
    ttsmi_event_t ev[4];                          /* main -> side hand-offs of the four weight-gradient groups */
    /* Backward chaining of consecutive blocks of a stack (both fuse_ln): the gradient of THIS block's input is the
     * upstream gradient of res-norm 2 of the block below, and this block's last dgrad (dqkv . Wqkv^T) completes it - with
     * `below` set that GEMM runs as ttsmi_hgemm_ln_bwd on the lower block's x^2 / rstd2 / gamma2 and writes ITS df, da and
     * res-norm-2 parameter partials (lnp_ws2, ttsmi_hgemm_ln_bwd_nparts rows), so the input gradient never reaches HBM
     * and the lower block's own ttsmi_layernorm_bwd_xhat launch disappears.  The lower block is then run with
     * ln2_done != 0: its `dout` argument is ignored.  Only valid when this block is the sole consumer of the lower
     * block's output. */
    const struct ttsmi_dense_block* below;
    int32_t ln2_done;
    /* res16 != 0 (fuse_ln only): the residual stream between the fused kernels is bf16.  Forward: the residual adds read
     * h_bf / a_bf (the tensors the GEMMs read anyway), `a` is not written and `out` only with bit 1 set (a block whose fp32
     * output somebody reads: the last of a stack); `h` may be NULL.  Backward: `da` holds bf16 [M,d]; `dh` is bf16 when the
     * block is chained to a lower one (`below`, which then must be res16 too) and the fp32 result of the call otherwise.
     * bf16 rounding of a LayerNorm output / of its gradient once per kernel boundary: depth-12 parity as measured in
     * tests/test_config1_parity_gpu.py. */
    int32_t res16;
    /* optional (NULL = not used): ttsmi_relu_bits_bytes(B*T, F) bytes.  When set and both FFN GEMMs are launches of the
     * K = 256 weight-stationary kernel (fuse_ln, d == 256, ttsmi_hgemm_k256_eligible), the forward also leaves
     * (h1 > 0) as one bit per element here and the backward masks the FFN2 dgrad with it instead of re-reading h1. */
    void* relu_bits;
    /* round 5 - the row-local chain (csrc/chain.hip, ttsmi_dense_chain_fwd): with chain_w set (fuse_ln and res16 required,
     * d == 256, F % 64 == 0) the forward's o-projection + res-norm 1, FFN1, FFN2 + res-norm 2 - and, with `above` set, the
     * qkv projection of the NEXT block of the stack - run as ONE launch on the weight stream chain_w
     * (ttsmi_dense_chain_pack of this block's wo_t / w1_t / w2_t and above->wqkv_t, repacked whenever the weights change).
     * qkv_done != 0: this block's qkv was written by the chain of the block below (whose `above` is this descriptor) - its
     * own qkv projection is skipped.  The caller sets above / qkv_done in pairs and runs the lower block's forward first. */
    const void* chain_w;
    uint64_t chain_w_bytes;
    const struct ttsmi_dense_block* above;
    int32_t qkv_done;
    int32_t chain_pad_;
    /* the backward chain (ttsmi_dense_chain_bwd): with chain_bw set (chain_w required; the forward then writes relu_bits in the
     * backward chain's lane layout - the same ttsmi_relu_bits_bytes(B * T, F) bytes hold it) the FFN2 dgrad, the FFN1
     * dgrad + res-norm 1 backward and the dctx product of ttsmi_dense_block_bwd run as ONE launch on the weight stream
     * chain_bw (ttsmi_dense_chain_bwd_pack of w1_b / w2_b / wo_b). */
    const void* chain_bw;
    uint64_t chain_bw_bytes;
} ttsmi_dense_block;
/* ---------------------------------------------------------------------------------------------
 * Batch data parallelism for a binding WITHOUT a collective library of its own (one all-reduce of the
 * flat fp32 gradient buffer per step; the reference itself has no distributed code).  The Python host side of this
 * repository uses torch.distributed (backend "nccl" = RCCL) instead; these four are the same collective behind the C
 * ABI.  RCCL is resolved at first use with dlopen (the copy already loaded in the process, else librccl.so.1):
 * TTSMI_ERR_UNSUPPORTED when it is absent.  One process per GPU:
 *   rank 0: ttsmi_comm_unique_id(id) -> 128 bytes, shipped to the other ranks by the caller's own means;
 *   every rank: hipSetDevice(local_rank); ttsmi_comm_init_rank(&comm, nranks, id, rank);
 *   every step: ttsmi_allreduce_sum_f32(comm, flat_grad, n, stream)  - in place, asynchronous on `stream`
 *               (the global-count loss divisors of ttsmi_l1_losses_weighted make SUM the right reduction);
 *   ttsmi_comm_destroy(comm).
 * ------------------------------------------------------------------------------------------- */
int ttsmi_comm_unique_id(void* id128);
int ttsmi_comm_init_rank(void** comm, int nranks, const void* id128, int rank);
int ttsmi_comm_destroy(void* comm);
int ttsmi_allreduce_sum_f32(void* comm, float* buf, int64_t n, ttsmi_stream_t stream);

/* Measurement hook: when set, ttsmi_dense_block_fwd/_bwd announce every launch group they issue - phase 0 before, 1 after
 * it is enqueued - with the entry point's name, its algorithmic FLOPs and bytes and the stream it goes to, so a profiler
 * can bracket the launches with HIP events.  NULL (the default) disables it.  Process-wide, for measurement only. */
typedef void (*ttsmi_launch_observer)(int phase, const char* name, double flops, double bytes, ttsmi_stream_t stream);
int ttsmi_set_launch_observer(ttsmi_launch_observer cb);
/* Measurement only (the CU-partitioning A/B of the weight-gradient side stream): a HIP stream whose
 * kernels may only run on the CUs whose bits are set in mask[0 .. nwords) (hipExtStreamCreateWithCUMask; the caller owns
 * the stream), and a census - counts8[x] += the number of one-wave workgroups of an nblocks launch on `stream` that ran on
 * XCC x - that shows which CUs a mask really selects (tools/probe_cu_mask.py).  The product never calls either. */
int ttsmi_debug_stream_create_cu_mask(const uint32_t* mask, int nwords, ttsmi_stream_t* out);
int ttsmi_debug_xcc_census(int32_t* counts8, int nblocks, ttsmi_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * The row-local chain of a dense block (model/layers.py:148-150,211,229 -> :99-102,230 -> the next block's :116-118) as
 * one launch for d_model = 256 (csrc/chain.hip): per 128-row workgroup
 *     a = LN1(keep([h | ctx].Wo + bo) + h) * rowmask;  h1 = relu(a.W1 + b1);  out = LN2(keep(h1.W2 + b2) + a) * rowmask;
 *     qkv_next = out.Wqkv' + bqkv'                                   (qkv_next / bqkv_next NULL: no next block)
 * with the activations in registers between the products; only what the backward keeps is written (a_bf, xhat1, rstd1, h1,
 * xhat2, rstd2, out_bf, qkv_next; out32 = the fp32 block output, or NULL; relu_bits = (h1 > 0) in the layout
 * ttsmi_hgemm_k256_masked_bits reads for (M, F), or NULL).  Residuals are the bf16 tensors (h_bf, a_bf): the res16
 * arithmetic of ttsmi_dense_block.  Same results as ttsmi_hgemm_ln_fwd_h + ttsmi_hgemm_k256_relu_bits +
 * ttsmi_hgemm_ln_fwd_h + ttsmi_hgemm_tn up to fp32 summation order.
 *   wpack: ttsmi_dense_chain_pack(wo_t [256][512], w1_t [F][256], w2_t [256][F], wqkv_next_t [768][256] or NULL) - the bf16
 *          W^T matrices as one stream of MFMA fragments in the kernel's order; ttsmi_dense_chain_pack_bytes(F, with_qkv).
 *   dropout: keep(seed + *step_dev, site, row, column) as in ttsmi_hgemm_ln_fwd (site_ln1 / site_ln2).
 * ------------------------------------------------------------------------------------------- */
size_t ttsmi_dense_chain_pack_bytes(int F, int with_qkv);
int ttsmi_dense_chain_pack(const uint16_t* wo_t, const uint16_t* w1_t, const uint16_t* w2_t, const uint16_t* wqkv_next_t, int F,
                           void* out, size_t out_bytes, ttsmi_stream_t stream);
/* Several weight streams in one launch (a train step repacks every block's two streams after each optimiser update).
 * backward == 0: wo / w1 / w2 / wqkv_next are ttsmi_dense_chain_pack's wo_t / w1_t / w2_t / wqkv_next_t;
 * backward != 0: wo / w1 / w2 are ttsmi_dense_chain_bwd_pack's wo_b / w1_b / w2_b (wqkv_next unused).  Same bytes as the
 * single calls. */
typedef struct ttsmi_chain_pack_job {
    const uint16_t *wo, *w1, *w2, *wqkv_next;
    void* out;
    size_t out_bytes;
    int32_t F, backward;
} ttsmi_chain_pack_job;
int ttsmi_dense_chain_pack_batched(const ttsmi_chain_pack_job* jobs, int n, ttsmi_stream_t stream);
int ttsmi_dense_chain_supported(int M, int d, int F);
int ttsmi_dense_chain_fwd(const uint16_t* h_bf, const uint16_t* ctx, const void* wpack, size_t wpack_bytes, int M, int F,
                          const float* bo, const float* ln1_g, const float* ln1_b, const float* b1, const float* b2,
                          const float* ln2_g, const float* ln2_b, const float* bqkv_next, const uint8_t* row_pad, float p_drop,
                          uint64_t seed, const int64_t* step_dev, uint32_t site_ln1, uint32_t site_ln2, float eps, uint16_t* a_bf,
                          uint16_t* xhat1, float* rstd1, uint16_t* h1, void* relu_bits, int relu_bits_layout, uint16_t* out_bf,
                          uint16_t* xhat2, float* rstd2, float* out32, uint16_t* qkv_next, ttsmi_stream_t stream);
/* relu_bits_layout: 0 = the bit matrix ttsmi_hgemm_k256_masked_bits reads (ttsmi_relu_bits_bytes(M, F) bytes), 1 = the layout
 * ttsmi_dense_chain_bwd reads (16-bit word (row / 16, 64-feature chunk, lane) of the kernel's own lanes; M * F / 8 bytes
 * rounded up to whole 16-row tiles).
 *
 * The BACKWARD of the same block between its two res-norms as one launch (csrc/chain16b.h):
 *     dh1 = (df . W2^T) * [h1 > 0];  g = da + dh1 . W1^T;  (d_o, dres) = LN1'(g) with x^1, rstd1, gamma1 (ttsmi_hgemm_ln_bwd's
 *     arithmetic: d_o = keep(dz), dres = dz, one partial row of dgamma / dbeta per 128-row workgroup in part_ws:
 *     ttsmi_layernorm_partials_bytes(cdiv(M, 128), 256));  dctx = d_o . Wo[256:512]^T
 * df / da: bf16 [M,256] (the res-norm 2 backward's outputs); relu_bits_lane: the forward chain's relu_bits with
 * relu_bits_layout = 1; wpack: ttsmi_dense_chain_bwd_pack(w1_b [256][F], w2_b [F][256], wo_b [512][256]) - the weights AS
 * STORED (bf16); dres: bf16 [M,256] when dres_is_bf16, else fp32.  Replaces ttsmi_hgemm_k256_masked_bits +
 * ttsmi_hgemm_ln_bwd_dual_h + the dctx GEMM of ttsmi_dense_block_bwd (same results up to fp32 summation order). */
size_t ttsmi_dense_chain_bwd_pack_bytes(int F);
int ttsmi_dense_chain_bwd_supported(int M, int d, int F);
/* rows of dgamma / dbeta partials ttsmi_dense_chain_bwd leaves in part_ws (one per 128-row workgroup): the `nw` of the
 * reduction that follows (ttsmi_layernorm_param_reduce_batched_nw) */
int ttsmi_dense_chain_bwd_nparts(int M);
int ttsmi_dense_chain_bwd_pack(const uint16_t* w1_b, const uint16_t* w2_b, const uint16_t* wo_b, int F, void* out, size_t out_bytes,
                               ttsmi_stream_t stream);
int ttsmi_dense_chain_bwd(const uint16_t* df, const uint16_t* da, const uint16_t* xhat1, const float* rstd1, const float* ln1_g,
                          const uint8_t* row_pad, const void* relu_bits_lane, const void* wpack, size_t wpack_bytes, int M, int F,
                          float p_drop, uint64_t seed, const int64_t* step_dev, uint32_t site_ln1, uint16_t* dh1, uint16_t* d_o,
                          void* dres, int dres_is_bf16, uint16_t* dctx, void* part_ws, size_t part_ws_bytes, ttsmi_stream_t stream);
/* h [M,d] fp32 block input, h_bf its bf16 copy.  Writes desc->out / out_bf (+ the kept activations). */
int ttsmi_dense_block_fwd(const ttsmi_dense_block* desc, const float* h, const uint16_t* h_bf);
/* dout [M,d] fp32 gradient of the block output.  Writes desc->dh, the parameter gradients, and leaves the two
 * LayerNorm parameter-gradient partials in ln_ws1 / ln_ws2 for ttsmi_layernorm_param_reduce_batched.
 * The caller joins side_stream before it reads the weight gradients. */
int ttsmi_dense_block_bwd(const ttsmi_dense_block* desc, const float* h, const uint16_t* h_bf, const float* dout);
/* 1 when ttsmi_dense_block_bwd(desc) will take the backward chain (chain_bw, chain_w, fuse_ln, res16, relu_bits set, the
 * shape supported, TTSMI_WGRAD_EVENTS at its default): lnp_ws1 then holds ttsmi_dense_chain_bwd_nparts(B * T) partial rows,
 * otherwise ttsmi_hgemm_ln_bwd_nparts(B * T).  ttsmi_dense_block_fwd asks the same predicate for the ReLU bit layout. */
int ttsmi_dense_block_bwd_chained(const ttsmi_dense_block* desc);
/* A whole STACK of n consecutive dense blocks (SelfAttentionBlocks.call's loop over its dense blocks, model/layers.py:
 * 303-306) from one call: block i reads block i - 1's out / out_bf, the backward walks n - 1 .. 0 and hands block i + 1's
 * dh to block i as its dout (ignored by a chained block, see `below`).  Exactly the launches, in exactly the order, of n
 * ttsmi_dense_block_fwd / _bwd calls - minus their host round trips: with ~12 k rows per batch (the reference's bucket
 * sizes) a train step is bound by the host's issue rate, not by the GPU (bench.py --workload lj-dist).  h / h_bf: the
 * stack's input (block 0's); dout: the gradient of block n - 1's output.  blocks[i]->out / out_bf / dh must be the
 * buffers the neighbouring descriptors expect (every block of one shape B, T, d). */
int ttsmi_dense_stack_fwd(const ttsmi_dense_block* const* blocks, int n, const float* h, const uint16_t* h_bf);
int ttsmi_dense_stack_bwd(const ttsmi_dense_block* const* blocks, int n, const float* h, const uint16_t* h_bf, const float* dout);

/* ---------------------------------------------------------------------------------------------
 * One teacher-forced TRAIN STEP of ForwardTransformer from ONE descriptor (round 6): the reference compiles its step
 * once (model/models.py:442-451, tf.function + input signature) and its loop (train_tts.py:149-160) only feeds batches;
 * here the whole launch sequence of `_train_step` (models.py:464-482) - masks, embedding, encoder stack, the two
 * StatPredictors on a side stream, pitch embedding, Expand, decoder stack, mel projection, the three L1 losses, the
 * hand-ordered backward, the LayerNorm parameter reductions, TF-form Adam and the bf16 shadow refresh - is issued from
 * C++ by three calls (phases), with no autograd engine and no interpreter between the ~290 launches.  Every buffer is
 * caller-owned and persistent (sized for the largest batch seen), the dense blocks are the ttsmi_dense_block descriptors
 * of ttsmi_dense_stack_fwd / _bwd, events are caller-created hipEvent_t.  Same launches, same arguments, same streams as
 * the per-layer host path (transformertts_amd/ops.py): results are bit-identical to it.
 *   phase 0: keep-bit tables + chain weight streams (side stream), forward, losses, backward down to the decoder's entry
 *            LayerNorm;   [a data-parallel host launches the decoder-half all-reduce here]
 *   phase 1: Expand backward .. embedding backward, LayerNorm parameter reductions, stream joins;
 *   phase 2: step counter, Adam, bf16 shadows.
 * ------------------------------------------------------------------------------------------- */
#define TTSMI_FT_MAX_PRED_LAYERS 8
#define TTSMI_FT_MAX_BLOCKS 32
typedef struct {
    /* Conv1D(k, 'same') + ReLU -> LayerNorm -> dropout: one layer of a StatPredictor (model/layers.py:510-515) */
    int32_t k, Cin, Cout, Cout_pad;            /* Cout_pad = Cout rounded up to 8 (row pitch of dc, column count of w_d) */
    uint32_t site;                             /* dropout site of the layer's output */
    int32_t pad_;
    const uint16_t* w_t;                       /* bf16 [Cout][k Cin]: forward operand */
    const uint16_t* w_d;                       /* bf16 [Cin][k Cout_pad]: dgrad operand (ttsmi_conv_wdgrad_layout_bf16) */
    const float* bias;
    const float* ln_g; const float* ln_b;
    float* g_w; float* g_b; float* g_ln_g; float* g_ln_b;      /* gradient sinks */
    float* c;                                  /* [rows, Cout] relu(conv) */
    float* n;                                  /* [rows, Cout] LayerNorm output */
    float* mean; float* rstd;                  /* [rows] */
    float* dc;                                 /* [rows, Cout] gradient of c (ReLU' applied) */
    float* dc_pad;                             /* [rows, Cout_pad] zero-padded copy when Cout_pad != Cout, else unused */
    float* dx;                                 /* [rows, Cin] gradient of the layer's input */
    void* ln_ws; uint64_t ln_ws_bytes;         /* ttsmi_add_layernorm_bwd workspace (kept until the batched reduce) */
    uint16_t* xT; uint16_t* dyT;               /* cast-transpose scratch of the weight gradient when Cin % 128 or Cout % 4 != 0 */
    void* wg_ws; uint64_t wg_ws_bytes;         /* weight-gradient workspace */
} ttsmi_ft_pred_layer;

typedef struct {
    int32_t n_layers, relu_head;
    ttsmi_ft_pred_layer layer[TTSMI_FT_MAX_PRED_LAYERS];
    const float* lin_w; const float* lin_b; float* g_lin_w; float* g_lin_b;     /* Dense(1) head */
    float* hm;                                 /* [rows, d] masked encoder output (layers.py:482) */
    float* y;                                  /* [rows] prediction */
    float* dn;                                 /* [rows, C_last] gradient of the last LayerNorm output */
    float* dbranch;                            /* [rows, d] gradient wrt the encoder output (row-masked) */
    void* rd_ws; uint64_t rd_ws_bytes;         /* ttsmi_rowdot_bwd workspace */
} ttsmi_ft_predictor;

typedef struct {
    int32_t B, Tp, Tm, d, V, n_mel, n_enc, n_dec;
    float rate, prate;                         /* dropout_rate, predictors_dropout */
    uint64_t seed; const int64_t* step_dev;
    uint32_t site_enc_ln, site_dec_ln;
    ttsmi_stream_t main_stream, side_stream, wgrad_stream;
    void* ev[12];                              /* hipEvent_t, caller-created (cross-stream hand-offs, one use each per step) */
    /* the batch (device) */
    const int32_t* tokens; const float* tgt_mel; const int32_t* tgt_dur; const float* tgt_pitch;
    /* parameters outside the blocks and their gradient sinks */
    const float* emb; float* g_emb;
    const float* enc_ln_g; const float* enc_ln_b; const float* enc_ps; float* g_enc_ln_g; float* g_enc_ln_b; float* g_enc_ps;
    const float* dec_ln_g; const float* dec_ln_b; const float* dec_ps; float* g_dec_ln_g; float* g_dec_ln_b; float* g_dec_ps;
    const float* pe_enc; const float* pe_dec;  /* sinusoid tables [max_pos, d] */
    const float* pit_w; const float* pit_b; float* g_pit_w; float* g_pit_b;
    const uint16_t* out_wt; const uint16_t* out_wb; const float* out_b; float* g_out_w; float* g_out_b;
    const ttsmi_dense_block* enc[TTSMI_FT_MAX_BLOCKS];
    const ttsmi_dense_block* dec[TTSMI_FT_MAX_BLOCKS];
    ttsmi_ft_predictor dur, pit;
    /* persistent activations / gradients */
    uint8_t* pad_e; int32_t* klen_e; uint8_t* pad_d; int32_t* klen_d;
    float* x_emb; float* h0; uint16_t* h0_bf; float* mean0; float* rstd0;      /* encoder entry */
    float* hp;                                 /* [B Tp, d] encoder output + pitch embedding */
    int32_t* idx; int32_t* cum; int32_t* lens; /* Expand tables */
    float* x_dec; float* h1; uint16_t* h1_bf; float* mean1; float* rstd1;      /* decoder entry */
    float* mel;                                /* [B Tm, n_mel] */
    float* loss_out;                           /* [4]: mel, duration, pitch, total */
    float* g_mel; float* g_dur; float* g_pit;  /* loss gradients */
    void* loss_ws; uint64_t loss_ws_bytes;
    float loss_w[3]; int32_t pad2_;
    int64_t loss_denom[3];                     /* 0 = the term's own element count */
    float* d_dec_out;                          /* [B Tm, d] */
    float* d_x_dec;                            /* [B Tm, d] */
    float* d_hp;                               /* [B Tp, d] */
    float* d_branch;                           /* [B Tp, d] sum of the predictors' gradients */
    float* d_enc_out;                          /* [B Tp, d] */
    float* d_x_emb;                            /* [B Tp, d] */
    void* ln_ws0; void* ln_ws1; uint64_t ln_ws0_bytes, ln_ws1_bytes;            /* entry LayerNorm backward workspaces */
    void* pit_ws; uint64_t pit_ws_bytes;
    void* wgrad_ws; uint64_t wgrad_ws_bytes;   /* the weight-gradient stream's workspace (shared with the blocks) */
    /* phase 2 */
    float* p_flat; float* g_flat; float* m_flat; float* v_flat; int64_t n_flat;
    const float* lr_dev; int64_t* step_rw;
    float beta1, beta2, eps;
    int32_t pack_now;                          /* phase 0 packs the chain kernels' weight streams itself (else: the previous phase 2 did) */
    uint16_t* flat_bf16;
    const void* tr_desc; int32_t tr_n, tr_tiles;
    int32_t n_conv_wd;
    int32_t pack_ahead;                        /* phase 2 packs the NEXT step's weight streams on the side stream */
    const float* conv_w[2 * TTSMI_FT_MAX_PRED_LAYERS]; uint16_t* conv_wd[2 * TTSMI_FT_MAX_PRED_LAYERS];
    int32_t conv_k[2 * TTSMI_FT_MAX_PRED_LAYERS], conv_cin[2 * TTSMI_FT_MAX_PRED_LAYERS], conv_cout[2 * TTSMI_FT_MAX_PRED_LAYERS];
} ttsmi_ft_step;
/* phase: 0, 1, 2 as above.  Returns TTSMI_OK or the first failing entry point's code (ttsmi_last_error()). */
int ttsmi_ft_train_step(const ttsmi_ft_step* step, int phase);
/* out = a + b (fp32, n elements): the sum of a multiply-used tensor's gradients (the encoder output feeds the pitch
 * embedding and both predictors) */
int ttsmi_add2_f32(const float* a, const float* b, float* out, int64_t n, ttsmi_stream_t stream);
/* dst[m, 0 .. C) = src[m, 0 .. C), dst[m, C .. Cp) = 0: the zero-padded gradient a Conv1D dgrad with Cout % 8 != 0 reads */
int ttsmi_pad_cols_f32(const float* src, int C, float* dst, int Cp, int M, ttsmi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TTSMI_H */
