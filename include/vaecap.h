/* libvaecap C ABI -- MI355X (gfx950) kernels for the CVAE / AG-CVAE captioning trainer.
 *
 * The reference (yiyang92/vae_captioning) has NO native/FFI interface: its hot path is a
 * TensorFlow-1 static graph.  Each entry point below therefore names the reference GRAPH
 * CALL SITE (file:line under /root/reference) whose forward and/or tf.gradients-derived
 * backward it replaces; INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns 0 on success, else a
 *     hipError_t value or a VC_E* code (message: vc_last_error()).  No exceptions.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it.
 *   - All tensor pointers are DEVICE pointers owned by the caller; the library never
 *     allocates or frees caller-visible memory.  Scratch comes from caller workspaces.
 *   - fp32 everywhere (tf.float32 graph); token ids / lengths are int32.
 *   - Sequences are TIME-MAJOR [T, N, ...] (the reference feeds [N, T]; the transpose is
 *     host-side index plumbing).
 */
#ifndef VAECAP_H
#define VAECAP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (binders gate on vc_abi_version(); a mismatch is a LAYOUT break, not just a symbol break):
 *   4  precision and step-kernel choice are PER CALL: vc_gemm_f32 takes VC_GEMM_BF16X3 in `flags`; vc_lstm_seq_fwd_f32 / _bwd_f32 /
 *      _bwd_data_f32 / _bwd_weights_f32 gained a trailing `int flags` (VC_LSTM_BF16X3, VC_LSTM_KERNELS(k)): a v3 binder would pass one
 *      argument too few.  vc_gemm_set_precision / vc_lstm_set_mode remain as DEPRECATED process-wide defaults for calls that pass no
 *      choice of their own (the product path never calls them: two trainers of different precision coexist in one process).  Gone: the
 *      direct split-bf16 forward / data-gradient convolutions (vc_conv3x3_bx_* except the weight gradient, vc_conv3x3_bx2_*), which no
 *      product path called.  New: the F(4x4,3x3) convolution with a pre-transformed input (vc_conv3x3_wino4v_*), the decode-round
 *      entries vc_softmax_topk_rows_f32 / vc_beam_gather_f32 / vc_eos_track_i32 / vc_beam_init.
 *   3  the 3x3-convolution family (vc_conv3x3_wino_*, vc_conv3x3_wino4_*, vc_conv3x3_wino_wgrad_*, vc_conv1_fwd* / vc_conv1_wgrad*,
 *      vc_maxpool2x2_bwd_bits_f32) takes and returns activations in the C4 layout [B][C/4][H][W][4] (v2: NHWC) and the pool routing
 *      codes / ReLU mask bits follow it; the vc_conv3x3_patch_*, vc_conv3x3_pack_f32, *_packed_f32 and wgrad_patch_* entries of v2 are
 *      gone.  A v2 binder would link and compute wrong numbers: it must refuse to run against a v3 library.  New in 3: the split-bf16
 *      products (vc_gemm_bf16x3_*), vc_vgg_preprocess_u8.
 *   2  round-3 surface (NHWC convolutions).  */
#define VC_ABI_VERSION 4
int vc_abi_version(void);
const char* vc_last_error(void);
/* 0 if a gfx950 device is usable from this process, else an error code. */
int vc_device_check(int device);
/* Tracing: named roctx ranges (`rocprofv3 --marker-trace --kernel-trace`); libroctx64 is bound at run time, without it push / pop are
 * no-ops.  trainer.Trainer brackets the phases of a step (VGG16 forward, caption forward / backward, VGG16 backward with its gradient
 * buckets, optimisers) when VC_TRACE=1. */
int vc_trace_available(void);
int vc_trace_push(const char* name);
int vc_trace_pop(void);

/* ------------------------------------------------------------------------------------
 * GEMM  C[M,N] = op(A)[M,K] . op(B)[K,N] (+ bias[N]) ; flags below.   fp32 MFMA.
 *   ta = 0: A stored [M,K] row-major (lda);  ta = 1: A stored [K,M] (computes A^T.B)
 *   tb = 0: B stored [K,N] row-major (ldb);  tb = 1: B stored [N,K] (computes A.B^T)
 * replaces tf.matmul / tf.layers.dense: main.py:94,108; vae_model/encoder.py:60-65,78-81,
 * 94-97; vae_model/decoder.py:111,127-129; utils/image_embeddings.py:223,234 and their
 * gradients (ops/optimizers.py:13,51).
 * Small outputs with long K are split along K into `ws` (vc_gemm_workspace_bytes) and
 * reduced in a fixed order (deterministic).
 * ---------------------------------------------------------------------------------- */
#define VC_GEMM_RELU 1
#define VC_GEMM_ACCUMULATE 2
#define VC_GEMM_BF16X3 4 /* this call on the bf16 matrix pipe with split operands (see vc_gemm_bf16x3_f32) */
size_t vc_gemm_workspace_bytes(int M, int N, int K);
int vc_gemm_f32(void* stream, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B,
                long ldb, float* C, long ldc, const float* bias, int flags, float* ws, size_t ws_bytes);
/* The same product on the bf16 matrix pipe ("bf16x3"): the f32 operands are split into (hi, lo) bf16 pairs while they are staged
 * into the LDS, three v_mfma_f32_32x32x16_bf16 products (hi.hi + hi.lo + lo.hi) accumulate in f32.  Same arguments, tile plan,
 * workspace and summation structure as vc_gemm_f32; |error| <= ~1e-5 of sum |a.b| (tests/test_gpu_bf16x3.py) instead of ~1e-7.
 * NOT the reference's arithmetic (tf.float32 matmul, main.py / vae_model/decoder.py:126-129): an opt-in mode, reported separately.
 * Per call: vc_gemm_f32 with VC_GEMM_BF16X3 in `flags` == this entry.  DEPRECATED: vc_gemm_set_precision(1) makes every vc_gemm_f32 /
 * vc_lstm_seq_* call of the process that carries no flag of its own take this path (0, the default = f32 MFMA); process-wide state,
 * not thread-safe, kept for ABI-3 callers only. */
int vc_gemm_bf16x3_f32(void* stream, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B,
                       long ldb, float* C, long ldc, const float* bias, int flags, float* ws, size_t ws_bytes);
int vc_gemm_set_precision(int mode);
int vc_gemm_get_precision(void);

/* ------------------------------------------------------------------------------------
 * Embedding lookup and its gradient.   tf.nn.embedding_lookup, vae_model/encoder.py:31-36,
 * vae_model/decoder.py:77-83 (pinned to /cpu:0 in the reference; on-device here).
 *   gather      out[r, :] = table[ids[r], :]
 *   scatter_add dtable[ids[r], :] += dX[r, :]      (IndexedSlices -> dense, TF-sem.)
 *   mark_rows   touched[ids[i]] = 1                 (rows a sparse Momentum update touches)
 * ---------------------------------------------------------------------------------- */
int vc_embedding_gather_f32(void* stream, const float* table, const int32_t* ids, long rows, int E, int vocab, float* out);
int vc_embedding_scatter_add_f32(void* stream, float* dtable, const int32_t* ids, long rows, int E, int vocab, const float* dX);
/* Deterministic form of scatter_add: out row v = sum of dX[order[j]] for j in seg_start[v] ..
 * seg_start[v+1] (nrows+1 entries), summed in that order; order = stable argsort of the ids (NULL =
 * identity).  Writes EVERY output row (zeros for empty segments), so repeated runs are
 * bit-identical.  Hot tokens are handled by calling it twice (sub-segments, then partial rows). */
int vc_embedding_grad_sorted_f32(void* stream, float* dtable, const int32_t* order, const int32_t* seg_start, int E,
                                 int nrows, const float* dX);
/* Inverted index of the token ids for the call pair above, built on device (stable counting sort, integer work, no
 * atomics): order [R] = stable argsort of clip(ids, 0, vocab-1); seg1 [max_subsegments + 1] = boundaries into `order` of
 * sub-segments of <= chunk positions that never cross an id boundary (entries past the real count are R: empty
 * sub-segments, so the first-level call may always run over max_subsegments rows -- no host read-back); seg2 [vocab + 1] =
 * for every id the range of its sub-segments.  ws: int32 scratch of vc_embedding_index_workspace_bytes. */
size_t vc_embedding_index_workspace_bytes(long R, int vocab);
size_t vc_embedding_index_max_subsegments(long R, int vocab, int chunk);
int vc_embedding_index_max_vocab(void); /* larger vocabularies: build the index on the host (engine.embedding_grad_index) */
int vc_embedding_grad_index(void* stream, const int32_t* ids, long R, int vocab, int chunk, int32_t* order, int32_t* seg1,
                            int32_t* seg2, int32_t* ws, size_t ws_bytes);
int vc_mark_rows_f32(void* stream, float* touched, const int32_t* ids, long n, int vocab);

/* ------------------------------------------------------------------------------------
 * LSTM.  utils/rnn_model.py:23-51 (make_rnn_cell), stepped at vae_model/encoder.py:46-55 and
 * vae_model/decoder.py:100-121.  W = [E+H, 4H] (rows 0..E-1 multiply x), gate order i,j,f,o,
 * forget_bias 1.0 (TF-sem.).  lens_eff[n]: step t of row n is active iff t < lens_eff[n]
 * (tf.nn.dynamic_rnn sequence_length semantics: state carried through when inactive).
 * H must be a multiple of 32.
 *   step_fwd: gact [N,4H] holds x.Wx+b on entry and the gate activations (i,j,f,o) on exit.
 *   step_bwd: one step of back-propagation through time (see lstm.hip for the recurrences).
 *   seq_fwd / seq_bwd: native loops over the T steps plus the batched input-projection and
 *   weight-gradient GEMMs.  cs / hs are [T+1,N,H] with index 0 = initial state.
 * ---------------------------------------------------------------------------------- */
int vc_lstm_step_fwd_f32(void* stream, int N, int H, int t, const float* h_prev, const float* c_prev, const float* Wh,
                         float* gact, const int32_t* lens_eff, float* c_out, float* h_out);
int vc_lstm_step_bwd_f32(void* stream, int N, int H, int t, int first, const float* dG_next, const float* Wh,
                         const int32_t* lens_eff, const float* dh_ext, float* dH_run, float* dC_run, const float* act,
                         const float* c_prev, const float* c_cur, float* dG);
/* Sequence drivers: `flags` of the vc_lstm_seq_* calls = VC_LSTM_KERNELS(k) | VC_LSTM_BF16X3.
 *   VC_LSTM_BF16X3   the call's products (input projection, recurrence, data / weight gradients) in the split-bf16 arithmetic of
 *                    vc_gemm_bf16x3_f32 (an opt-in mode, NOT the reference's tf.float32 LSTMCell)
 *   VC_LSTM_KERNELS(k), k = 0..3: which step kernels run (0 in the low bits = the default, auto); the numbering is vc_lstm_set_mode's.
 * vc_lstm_set_mode is the DEPRECATED process-wide default for calls whose flags choose nothing (identical arithmetic up to fp32 summation order and, in the
 * recurrence kernels, sigmoid / tanh built from v_exp_f32 + v_rcp_f32, |error| < 3e-7):
 * 2 (default) = auto: the register-operand recurrence step kernels (one workgroup per CU, four waves split K, Wh slice packed
 * in MFMA-operand order, 16x16x4 tiles, gate math in the epilogue; above 400 rows the forward uses eight waves on 16-unit slices)
 * when H == 512, otherwise 1; 1 = recurrent GEMM via vc_gemm_f32 (split-K) + element-wise gate kernels; 3 = recurrence kernels wherever
 * supported; 0 = the round-1 fused step kernels. */
#define VC_LSTM_BF16X3 0x10
#define VC_LSTM_KERNELS(k) ((k) + 1)
int vc_lstm_set_mode(int split);
/* Single steps on the recurrence kernel for callers that advance one token at a time with fixed weights (generation: greedy /
 * sampling / beam search, vae_model/decoder.py:145-320): pack Wh [H,4H] once into whp (2 * H * 4H floats: the operand orders of
 * both step kernels, chosen by N), then step with the same arguments as vc_lstm_step_fwd_f32.  H == 512 only: ask vc_lstm_step_packed_supported. */
int vc_lstm_step_packed_supported(int N, int H);
int vc_lstm_pack_wh_f32(void* stream, int H, const float* Wh, float* whp);
int vc_lstm_step_fwd_packed_f32(void* stream, int N, int H, int t, const float* h_prev, const float* c_prev, const float* whp,
                                float* gact, const int32_t* lens_eff, float* c_out, float* h_out);
size_t vc_lstm_seq_workspace_bytes(int T, int N, int E, int H);
int vc_lstm_seq_fwd_f32(void* stream, int T, int N, int E, int H, const float* X, const float* W, const float* b,
                        const int32_t* lens_eff, float* act, float* cs, float* hs, float* ws, size_t ws_bytes, int flags);
int vc_lstm_seq_bwd_f32(void* stream, int T, int N, int E, int H, const float* X, const float* W, const int32_t* lens_eff,
                        const float* act, const float* cs, const float* hs, const float* dhs_ext, float* dH_run,
                        float* dC_run, float* dG, float* dX, float* dW, float* db, float* ws, size_t ws_bytes, int flags);
/* The backward pass in two calls (vc_lstm_seq_bwd_f32 = the first followed by the second on one stream): the recurrence with
 * dX = dG.Wx^T, which is all the gradient chain below the LSTM waits for, and the weight gradients dWx = X^T.dG, dWh = hs[0:T]^T.dG,
 * db = colsum(dG) from the dG the first call left -- tf.gradients has no consumer for them before the optimiser
 * (ops/optimizers.py:13-16), so a caller may enqueue the second call on another stream with its own workspace. */
int vc_lstm_seq_bwd_data_f32(void* stream, int T, int N, int E, int H, const float* W, const int32_t* lens_eff, const float* act,
                             const float* cs, const float* dhs_ext, float* dH_run, float* dC_run, float* dG, float* dX, float* ws,
                             size_t ws_bytes, int flags);
int vc_lstm_seq_bwd_weights_f32(void* stream, int T, int N, int E, int H, const float* X, const float* hs, const float* dG, float* dW,
                                float* db, float* ws, size_t ws_bytes, int flags);

/* ------------------------------------------------------------------------------------
 * Masked sparse softmax cross-entropy, main.py:152-158.  In place: `logits` [rows, V] (ld)
 * is overwritten by d(loss)/d(logits) = (softmax - onehot) * (label != 0) * gscale / den[0]
 * when write_grad; row_loss[r] = (logsumexp - logit[label]) * (label != 0).
 * den is a DEVICE scalar (number of non-PAD labels, global over data-parallel ranks).
 * softmax_rows: gen-mode probabilities (vae_model/decoder.py:140,142).
 * ---------------------------------------------------------------------------------- */
int vc_softmax_xent_f32(void* stream, float* logits, const int32_t* labels, long rows, int V, long ld, const float* den,
                        float gscale, float* row_loss, int write_grad);
int vc_softmax_rows_f32(void* stream, const float* x, long rows, int V, long ld, float* y, long ldy);
int vc_argmax_rows_f32(void* stream, const float* x, long rows, int cols, long ld, int32_t* out);
/* The first k entries of each row under a STABLE descending sort (ties -> lower index first): the
 * candidate expansion of beam search, vae_model/decoder.py:273-276. */
int vc_topk_rows_f32(void* stream, const float* x, long rows, int cols, long ld, int k, float* out_val, int32_t* out_idx);
/* vc_softmax_rows_f32 followed by vc_topk_rows_f32 (k <= 8) in ONE read of the logits and no [rows, V] probability buffer: out_p / out_idx
 * [rows, k] equal the two-call sequence bit for bit (same expressions, same summation order, same tie rule).  One beam-search round,
 * vae_model/decoder.py:248-276. */
int vc_softmax_topk_rows_f32(void* stream, const float* logits, long rows, int V, long ld, int k, float* out_p, int32_t* out_idx);
int vc_fill_f32(void* stream, float* x, long n, float value);
/* tf.multinomial(logits / temperature, 1) (vae_model/decoder.py:137-138) by inverse CDF with injected
 * uniforms u[rows] in [0,1): out[r] = first index whose cumulative softmax exceeds u[r]. */
int vc_multinomial_rows_f32(void* stream, const float* logits, long rows, int V, long ld, float temperature,
                            const float* u, int32_t* out);
/* Uniform [0,1) floats from the Philox stream (same counter convention as the other vc_philox_* calls). */
int vc_philox_uniform_f32(void* stream, float* out, long n, uint64_t seed, uint64_t offset, const int32_t* step);

/* ------------------------------------------------------------------------------------
 * Latent variable.  vae_model/encoder.py:59-109, main.py:118-145.
 *   latent_sample  z[s,n,l] = mean[n,l] + std[n,l]*eps[s,n,l]   (zs.Normal n_samples=S)
 *   kl_rows        mode 0: Normal/GMM KL per row (main.py:120-124,131-135); mode 1: AG
 *                  (main.py:140-145; mu_p = c_i.cluster_means [N,L])
 *   latent_bwd     dmean, dstd (or dlogstd = dstd*std when out_logstd) from dz [S,N,L] plus
 *                  ann[0]*kl_scale * dKL;  ann = device scalar (may be null = 1)
 *   heads_mix_*    GMM pick (idx != NULL, encoder.py:87-88) / AG mixture (c_i, :105-107) over
 *                  heads [N, 2*K*L] = [means | log-stds]
 * ---------------------------------------------------------------------------------- */
int vc_latent_sample_f32(void* stream, int S, int N, int L, const float* mean, const float* std_, const float* eps, float* z);
/* Data-parallel forms (the Q1 reshape mixes rows of the GLOBAL batch): this rank's z_rnn rows are the
 * flat range q in [q0, q0+nq) of the global [S, Ng, L] sample tensor, q = s*Ng + n; mean_g / std_g are the
 * all-gathered [Ng, L] statistics; sums_mixed returns the [Ng, L] partial sums to reduce-scatter.
 * vc_latent_bwd_f32 with S = 0 then adds the KL gradient to sums already stored in dmean / dstd. */
int vc_latent_sample_mixed_f32(void* stream, int Ng, int L, long q0, long nq, const float* mean_g, const float* std_g,
                               const float* eps, float* z);
int vc_latent_sums_mixed_f32(void* stream, int Ng, int L, long q0, long nq, const float* dz, const float* eps,
                             float* dmean_part, float* dstd_part);
int vc_kl_rows_f32(void* stream, int N, int L, int mode, const float* mean, const float* std_, const float* mu_p, float* row_kl);
int vc_latent_bwd_f32(void* stream, int S, int N, int L, int mode, int out_logstd, const float* dz, const float* eps,
                      const float* mean, const float* std_, const float* mu_p, const float* ann, float kl_scale,
                      float* dmean, float* dstd);
int vc_heads_mix_fwd_f32(void* stream, int N, int K, int L, const float* heads, const float* c_i, const int32_t* idx,
                         float* mean, float* std_);
int vc_heads_mix_bwd_f32(void* stream, int N, int K, int L, const float* heads, const float* c_i, const int32_t* idx,
                         const float* dmean, const float* dstd, float* dheads);
/* Step scalars, main.py:156-177: out4 = {rec_loss = ce_num/ce_den + reg_scale*reg_sumsq, mean KL =
 * kl_sum*inv_n, lower_bound = rec + ann*KL/10, ann}; every pointer is a device scalar (reg_sumsq,
 * kl_sum, ann may be NULL). */
int vc_loss_finalize_f32(void* stream, const float* ce_num, const float* ce_den, const float* reg_sumsq, float reg_scale,
                         const float* kl_sum, float inv_n, const float* ann, float* out4);

/* ------------------------------------------------------------------------------------
 * Small data-movement ops.
 *   colsum       out[c] (+)= sum_r x[r,c]                (bias gradients)
 *   dropout      y = x*mask/keep                          (tf.nn.dropout, decoder.py:85-87, rnn_model.py:45-46)
 *   relu_bwd     dx = dy*(y>0) [*mask/keep]               (ReluGrad [+ dropout grad], image_embeddings.py:224-237)
 *   tile_rows    y[i*nc+j,:] = x[i,:]                     (main.py:84-89);  segment_sum_rows = its gradient
 * ---------------------------------------------------------------------------------- */
size_t vc_colsum_workspace_bytes(long rows, int cols);
int vc_colsum_f32(void* stream, const float* x, long rows, int cols, long ld, float* out, int accumulate, float* ws, size_t ws_bytes);
int vc_dropout_f32(void* stream, const float* x, const float* mask, float keep, long n, float* y);
int vc_relu_bwd_f32(void* stream, const float* dy, const float* y, const float* mask, float keep, long n, float* dx);
int vc_tile_rows_f32(void* stream, const float* x, long B, int nc, int E, float* y);
int vc_segment_sum_rows_f32(void* stream, const float* y, long B, int nc, int E, float* x, int accumulate);
int vc_exp_f32(void* stream, const float* x, long n, float* y);
int vc_axpy_f32(void* stream, float a, const float* x, long n, float* y);
int vc_reduce_sum_f32(void* stream, const float* x, long n, float scale, float* out, int accumulate);
int vc_count_nonzero_i32(void* stream, const int32_t* x, long n, float* out);

/* ------------------------------------------------------------------------------------
 * Optimisers, ops/optimizers.py.
 *   sumsq_partial + clip_finalize = tf.clip_by_global_norm (:15-16): partial must hold
 *     vc_sumsq_blocks() floats per call; finalize sums n_partial of them in a fixed order and
 *     writes out[0] = global norm, out[1] = clip * min(1/norm, 1/clip).
 *   step_update: device-resident global_step, Adam lr_t, annealing coefficient
 *     (main.py:163-170), staircase lr decay (:24-31).  scalars[0..4], see optim.hip.
 *   adam / sgd / momentum: apply_gradients (:33-46, :68-81) over flat buffers; lr, scale are
 *     DEVICE scalars (scale may be NULL = 1); l2 adds l2*p to the gradient (main.py:69-74).
 * ---------------------------------------------------------------------------------- */
int vc_sumsq_blocks(void);
int vc_sumsq_partial_f32(void* stream, const float* x, long n, float* partial);
int vc_clip_finalize_f32(void* stream, const float* partial, int n_partial, float clip, float* out_norm_scale);
int vc_step_update(void* stream, int32_t* step, float* scalars, float lr, float cnn_lr, float beta1, float beta2,
                   float ann_param, int ann_on, int decay_steps);
int vc_adam_f32(void* stream, float* p, const float* g, float* m, float* v, long n, const float* lr_t, const float* scale,
                float beta1, float beta2, float eps, float l2);
/* The same update that also leaves sum(p_new^2) per workgroup in sumsq_partial[0 .. vc_adam_blocks(n)) (summed by vc_reduce_sum_f32):
 * the L2 regulariser's loss term of the NEXT step (main.py:69-74) without another pass over the 134 M VGG16 parameters. */
int vc_adam_blocks(long n);
int vc_adam_sumsq_f32(void* stream, float* p, const float* g, float* m, float* v, long n, const float* lr_t, const float* scale,
                      float beta1, float beta2, float eps, float l2, float* sumsq_partial);
int vc_sgd_f32(void* stream, float* p, const float* g, long n, const float* lr, const float* scale, float l2);
int vc_momentum_f32(void* stream, float* p, const float* g, float* accum, long n, const float* lr, const float* scale,
                    float momentum, float l2, const float* row_mask, int E);

/* ------------------------------------------------------------------------------------
 * Philox4x32-10 counter-based RNG (replaces TF's random_normal / dropout streams; the
 * streams cannot match TF's, parity tests inject noise instead).  Element i comes from
 * counter (i/4, offset) word i%4 under key (seed lo, seed hi + step); `step` (device int, may be
 * NULL) lives in the KEY so that graph replays draw fresh numbers and stream ids kept in
 * offset >> 32 never alias across steps.
 * ---------------------------------------------------------------------------------- */
int vc_philox_u32(void* stream, uint32_t* out, long n, uint64_t seed, uint64_t offset, const int32_t* step);
int vc_philox_normal_f32(void* stream, float* out, long n, uint64_t seed, uint64_t offset, const int32_t* step);
int vc_philox_bernoulli_f32(void* stream, float* out, long n, float keep, uint64_t seed, uint64_t offset, const int32_t* step);

/* ------------------------------------------------------------------------------------
 * VGG16 feature extractor, utils/image_embeddings.py:26-238.  NHWC activations, HWIO kernels,
 * channel counts powers of two >= 4 (conv1_1: RGB zero-padded to 4 channels).
 *   conv3x3_fwd    y = [relu](conv2d(x, w, stride 1, SAME) + bias)        (:40-44 ...)
 *   conv3x3_dgrad  dx = conv2d_backprop_input(dy, w) [* (relu_src > 0)]
 *   conv3x3_wgrad  dw (+)= conv2d_backprop_filter(x, dy)   (split-K, deterministic reduce); db != NULL also
 *                  returns the bias gradient db (+)= sum_pixels dy, summed from the dy tiles already staged in LDS
 *   maxpool2x2     tf.nn.max_pool 2x2/2 (:59-63 ...); bwd optionally fused with ReluGrad of x
 *   vgg_preprocess images [B,H,W,3] (0..255 RGB) - mean -> NHWC4            (:31-34)
 *   pad_dim        dst[o][c][i] = c < c_src ? src[o][c][i] : 0  (pad / strip a middle dim)
 * ---------------------------------------------------------------------------------- */
/* fwd / dgrad workspace (optional; NULL or too small -> single launch): lets the library run the last partial round of
 * tiles as a K-split second launch instead of a ragged round (csrc/conv.hip, launch_rounds); 0 when not needed. */
size_t vc_conv3x3_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout);
size_t vc_conv3x3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int vc_conv3x3_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                       const float* bias, float* y, int relu, float* ws, size_t ws_bytes);
int vc_conv3x3_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* w,
                         const float* relu_src, float* dx, float* ws, size_t ws_bytes);
size_t vc_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int vc_conv3x3_wgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy, float* dw,
                         float* db, int accumulate, float* ws, size_t ws_bytes);
int vc_maxpool2x2_fwd_f32(void* stream, int B, int H, int W, int C, const float* x, float* y);
int vc_maxpool2x2_bwd_f32(void* stream, int B, int H, int W, int C, const float* x, const float* dy, float* dx, int relu_grad);
int vc_vgg_preprocess_f32(void* stream, const float* images, int B, int H, int W, float* out_nhwc4);
/* The same from uint8 RGB pixels [B,H,W,3] -- what the reference's HDF5 file holds (preprocess.py:27-28) and feeds (utils/image_utils.py:8;
 * the placeholder's float32 cast, utils/image_embeddings.py:31-34, is this kernel): a quarter of the host -> device bytes of the
 * float form (9.6 MB instead of 38.5 MB per 64 images), bit-identical output.  B*H*W % 4 == 0. */
int vc_vgg_preprocess_u8(void* stream, const void* images_u8, int B, int H, int W, float* out_nhwc4);
int vc_pad_dim_f32(void* stream, const float* src, long outer, int c_src, int c_dst, int inner, float* dst);

/* THE C4 ACTIVATION LAYOUT.  Every Winograd entry below (vc_conv3x3_wino_*, vc_conv3x3_wino4_*, vc_conv3x3_wino_wgrad_*) and conv1_1's
 * output / output gradient take their activation tensors channel-blocked: [B][C/4][H][W][4] -- per image C/4 planes, a plane = H x W
 * pixels, a pixel = 16 bytes = four consecutive channels (element (b, y, x, c) at float index (((b * C/4 + c/4) * H + y) * W + x) * 4 + c%4).
 * Why: the kernels gather 16-byte pieces (a pixel's four channels of one Winograd phase) for the pixels of a halo patch; in NHWC the
 * pixels of a patch row are 4 C bytes apart, one wave load touches 64 cache lines and the L1's tag rate bounds the kernel (measured:
 * HISTORY.md section 4g); in C4 a patch row is 288 consecutive bytes.  Images stay contiguous (image ranges, data-parallel shards and the
 * > 2 GiB cuts are slices of the leading dimension as before); C = 4 (conv1_1's zero-padded RGB input) is the same memory in both
 * layouts.  The reference's NHWC order (utils/image_embeddings.py:222: pool5 flattened as (h, w, c)) is restored at the fc1 boundary.
 * vc_maxpool2x2_fwd_f32 / vc_maxpool2x2_bwd_f32 work on C4 tensors when called with (B * C/4, H, W, 4). */
int vc_nhwc_to_c4_f32(void* stream, int B, int H, int W, int C, const float* nhwc, float* c4);
int vc_c4_to_nhwc_f32(void* stream, int B, int H, int W, int C, const float* c4, float* nhwc);

/* Winograd F(2x2, 3x3) forward / data gradient (csrc/conv_wino.hip): the same convolution in fp32 with 2.25x fewer multiplications.
 * Input transform, sixteen position products on MFMA (16x16x4 tiles) and output transform in one kernel; a wave owns up to 16 tiles
 * (2x2 output pixels each) x 32 output columns x all sixteen positions, two workgroups share a CU.  The weights are transformed and packed once per optimiser step
 * (wp: 16 * Cin * Cout floats; transpose 0 = forward, 1 = flipped taps + transposed channels for the data gradient).  Results agree
 * with conv3x3_fwd / conv3x3_dgrad to fp32 rounding of a different summation (tests/test_gpu_conv_wino.py: same fp64 oracle, same
 * tolerance class).  x, y, ypool, dy, dx, relu_src: C4 layout.  ypool != NULL also writes max_pool2x2(y) (a pooling window is one Winograd tile).  Shapes: H, W even, gathered
 * channels % 16 == 0, output channels % 32 == 0; ask vc_conv3x3_wino_supported. */
int vc_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout, int dgrad);
/* The kernels address a launch's tensors with 32-bit offsets: calls on more than 2 GiB per tensor are cut into launches over image
 * ranges inside the library; 1 if [B,H,W,max(Cin,Cout)] floats fit one launch (the mask-bit variants below require it). */
int vc_conv3x3_wino_single_launch_supported(int B, int H, int W, int Cin, int Cout);
int vc_conv3x3_wino_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp);
int vc_conv3x3_wino_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                            const float* bias, float* y, float* ypool, int relu);
int vc_conv3x3_wino_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                              const float* relu_src, float* dx);

/* MaxPoolGrad routing codes: a pooled forward (bias + ReLU applied) can also leave, per pooled element, four bits = the position of the
 * first maximum of its 2x2 window (row-major) | 4 if that maximum is > 0 -- vc_conv3x3_wino_pool_words(B, H, W, Cout) 32-bit words; a
 * 16-bit half-word = the four channels of one pooled pixel of one channel quad, layout [B][Cout/4][H/2][W/2] half-words --, and
 * vc_maxpool2x2_bwd_bits_f32 (dy, dx in the C4 layout) routes the pooled gradient with them: bit-identical to vc_maxpool2x2_bwd_f32(x = y,
 * dy, dx, relu_grad = 1) on the same planes without re-reading the pre-pool activation (H, W = the PRE-pool size in both calls). */
size_t vc_conv3x3_wino_pool_words(int B, int H, int W, int C);
int vc_conv3x3_wino_fwd_pool_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                 const float* bias, float* y, float* ypool, uint32_t* pool_bits);
int vc_maxpool2x2_bwd_bits_f32(void* stream, int B, int H, int W, int C, const uint32_t* pool_bits, const float* dy, float* dx);

/* ReLU mask as bits: the forward of a layer can leave (y > 0) of every lane's 2x2 pixels x 8 columns as 32 bits (vc_conv3x3_wino_mask_words
 * (B, H, W, Cout) 32-bit words), and the data gradient of the NEXT 3x3 layer -- whose output has the same shape, hence the
 * same tiles and lanes -- reads those bits (one 4-byte load per lane) instead of relu_src (eight 16-byte loads + packing per lane:
 * 6-17 % of a data-gradient call).  Results are bit-identical to vc_conv3x3_wino_dgrad_f32 with relu_src = that y. */
size_t vc_conv3x3_wino_mask_words(int B, int H, int W, int C);
int vc_conv3x3_wino_fwd_mask_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                 const float* bias, float* y, int relu, uint32_t* mask_out);
int vc_conv3x3_wino_dgrad_bits_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                   const uint32_t* mask_bits, float* dx);

/* Winograd F(4x4,3x3) (csrc/conv_wino4.hip): the same convolution with 36 positions per 4 x 4 output tile -- 2.25 multiplications per
 * output where F(2x2,3x3) spends 4 -- for layers whose gathered channels are a multiple of 8 and produced channels a multiple of 32.
 * Every entry mirrors its vc_conv3x3_wino_* namesake argument for argument (same tensors, same routing-code format of the pooled
 * forward); what differs: the packed weights (36 * Cin * Cout floats, vc_conv3x3_wino4_pack_f32), the ReLU mask bits (64 per lane,
 * vc_conv3x3_wino4_mask_words words; bits written by this family's forward are read by this family's data gradient only), the
 * rounding (~1e-5 of the tensor maximum against ~1e-6).  vc_conv3x3_wino4_preferred: 1 where this kernel is the faster of the two
 * Winograd forms (the 224-, 112-, 28- and 14-wide layers of VGG16; the callers keep one family per block of layers between two
 * pools, since the mask bits pass from a layer's forward to the next layer's data gradient). */
int vc_conv3x3_wino4_supported(int B, int H, int W, int Cin, int Cout, int dgrad);
int vc_conv3x3_wino4_preferred(int B, int H, int W, int Cin, int Cout);
int vc_conv3x3_wino4_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp);
int vc_conv3x3_wino4_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                             float* y, float* ypool, int relu);
int vc_conv3x3_wino4_fwd_pool_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                  const float* bias, float* y, float* ypool, uint32_t* pool_bits);
size_t vc_conv3x3_wino4_mask_words(int B, int H, int W, int C);
int vc_conv3x3_wino4_fwd_mask_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                  const float* bias, float* y, int relu, uint32_t* mask_out);
int vc_conv3x3_wino4_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                               const float* relu_src, float* dx);
int vc_conv3x3_wino4_dgrad_bits_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                    const uint32_t* mask_bits, float* dx);
/* The F(4x4, 3x3) forward / data gradient with the input transformed ONCE per layer and pass (csrc/conv_wino4.hip, MODE 2): a first
 * kernel writes V = B^T d B of every (4 x 4 tile, gathered channel) into the workspace `vws` in the B-operand order of the MFMAs
 * (2.25 x the activation's bytes, HBM-bound), the main kernel then streams its B operands from V straight into registers -- no patch
 * staging, no patch LDS traffic and no transform arithmetic beside the MFMAs, where the fused kernel re-transforms every patch for
 * every 32-output-channel tile (16 x on the 512-channel layers).  Same packed weights (vc_conv3x3_wino4_pack_f32), same arithmetic in
 * the same order -- outputs, ReLU mask bits, pooled outputs and routing codes are BIT-IDENTICAL to the vc_conv3x3_wino4_* entry of the
 * same name, argument for argument + (vws, vws_bytes).  Shapes: those of vc_conv3x3_wino4_* whose launch puts the same tile in the same
 * lane (the linear-tile layers -- VGG16's 56- and 28-wide -- and 13..16 x 13..16 images), one launch (< 2 GiB): ask
 * vc_conv3x3_wino4v_supported.  vc_conv3x3_wino4v_workspace_bytes(B, H, W, C): C = the GATHERED channels (Cin forward, Cout data
 * gradient).  utils/image_embeddings.py:96-212 (conv3_1 .. conv5_3) and tf.gradients of them. */
int vc_conv3x3_wino4v_supported(int B, int H, int W, int Cin, int Cout, int dgrad);
int vc_conv3x3_wino4v_preferred(int B, int H, int W, int Cin, int Cout, int dgrad); /* 1: faster than the fused form for this launch (measured rule) */
size_t vc_conv3x3_wino4v_workspace_bytes(int B, int H, int W, int C);
int vc_conv3x3_wino4v_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                              float* y, float* ypool, int relu, float* vws, size_t vws_bytes);
int vc_conv3x3_wino4v_fwd_pool_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                                   float* y, float* ypool, uint32_t* pool_bits, float* vws, size_t vws_bytes);
int vc_conv3x3_wino4v_fwd_mask_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                                   float* y, int relu, uint32_t* mask_out, float* vws, size_t vws_bytes);
int vc_conv3x3_wino4v_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt, const float* relu_src,
                                float* dx, float* vws, size_t vws_bytes);
int vc_conv3x3_wino4v_dgrad_bits_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                     const uint32_t* mask_bits, float* dx, float* vws, size_t vws_bytes);
/* Winograd F(3x3, 2x2) weight gradient (csrc/conv_wino_wgrad.hip): both operands transformed in registers, the contraction runs over the
 * 2x2-pixel tiles; raw position sums per K split in the workspace, a reduce kernel sums the splits in fixed order and applies the
 * output transform.  Same contract as conv3x3_wgrad (db != NULL also returns the bias gradient, accumulate adds to dw / db); the
 * workspace is REQUIRED.  x, dy: C4 layout; dw: HWIO.  Shapes: H, W even, H >= 4, Cin % 64 == 0, Cout % 64 == 0; ask vc_conv3x3_wino_wgrad_supported. */
int vc_conv3x3_wino_wgrad_supported(int B, int H, int W, int Cin, int Cout);
size_t vc_conv3x3_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int vc_conv3x3_wino_wgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy, float* dw,
                              float* db, int accumulate, float* ws, size_t ws_bytes);

/* DIRECT 3x3 weight gradient on the bf16 matrix pipe (csrc/conv_wgrad_bx.hip; backward of tf.nn.conv2d w.r.t. the filter,
 * utils/image_embeddings.py:36-212, in the split-bf16 arithmetic of vc_gemm_bf16x3_f32 -- the opt-in mode of Trainer(precision="bf16x3")).
 * The contraction runs over the pixels; both operands are split in registers.  Same contract as vc_conv3x3_wino_wgrad_f32 (db != NULL also
 * returns the bias gradient -- summed in f32 --, accumulate adds to dw / db, the workspace is REQUIRED, results bit-reproducible).
 * x, dy: C4 layout, f32; dw: HWIO.  Shapes: Cin % 64 == 0, Cout % 64 == 0, any H, W >= 1; ask vc_conv3x3_bx_wgrad_supported. */
int vc_conv3x3_bx_wgrad_supported(int B, int H, int W, int Cin, int Cout);
size_t vc_conv3x3_bx_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int vc_conv3x3_bx_wgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy, float* dw,
                            float* db, int accumulate, float* ws, size_t ws_bytes);

/* conv1_1 (utils/image_embeddings.py:36-48: 3 -> 64 channels), csrc/conv_first.hip: the layer is HBM-bound (it writes / re-reads
 * the 64-channel activation, 822 MB at 64 images, for 0.6 % of the multiply-adds), so it has its own kernels: the forward makes
 * the 64 output channels the M dimension of the MFMA so that every lane stores 16-byte vectors of consecutive channels straight
 * from its accumulators (32 lanes = 32 consecutive pixels of one C4 plane); the weight gradient contracts 27 patch rows + one row of
 * ones (= the bias gradient) against dy and reduces per-workgroup partials in a fixed order.  y, dy: C4 layout ([B][16][H][W][4]);
 * x4: [B,H,W,4] from vc_vgg_preprocess_f32 (fourth channel ignored);
 * w / dw: [3,3,3,64] HWIO (no padding to 4 channels); W % 32 == 0 (ask vc_conv1_supported, else use vc_conv3x3_*_f32 on the
 * zero-padded weights).  There is no data gradient: the images are not trained. */
int vc_conv1_supported(int B, int H, int W);
size_t vc_conv1_wgrad_workspace_bytes(void);
int vc_conv1_fwd_f32(void* stream, int B, int H, int W, const float* x4, const float* w, const float* bias, float* y, int relu);
/* The forward with ReLU, also leaving (y > 0) as bits for the F(4x4,3x3) data gradient of the next layer (conv1_2): mask_out =
 * vc_conv3x3_wino4_mask_words(B, H, W, 64) words, to be passed to vc_conv3x3_wino4_dgrad_bits_f32(B, H, W, 64, Cout, ...); bit-identical
 * to that entry's float-mask form on y.  One launch (B images within 2 GiB), H % 16 == 0, W % 32 == 0. */
int vc_conv1_fwd_mask_f32(void* stream, int B, int H, int W, const float* x4, const float* w, const float* bias, float* y, uint32_t* mask_out);
int vc_conv1_wgrad_f32(void* stream, int B, int H, int W, const float* x4, const float* dy, float* dw, float* db, int accumulate,
                       float* ws, size_t ws_bytes);

/* ------------------------------------------------------------------------------------
 * Beam-search bookkeeping after one decoder step, on device, one wave per image: the loop body of
 * vae_model/decoder.py:254-293 with utils/top_n.py's TopN (heapq min-heap keyed by score; ties resolved by
 * heapq's sift order, reproduced exactly).  Rows are [B, beam]: row b*beam + i is the i-th live beam of
 * image b in heap-array order.
 *   in : top_p / top_i [B*beam, beam]  the beam_size most probable words of every row, descending, stable
 *        pcount [B] live beams; p_score / p_logprob (double) / p_len [B, beam]; sent_cur [B, beam, Lmax]
 *   out: the same for the new live beams (sent_next), the complete-caption heap c_* (captions in the pool
 *        c_sent [B, beam+1, Lmax], c_slot -> pool row, c_free = free-row bit mask, initially 2^(beam+1)-1),
 *        and for the next step parent [B*beam] (row whose LSTM state each new beam continues) and tok [B*beam].
 *   score of a completed caption = logprob / len^len_norm_f (len_norm_f <= 0: logprob); words with p < 1e-12 skipped.
 * ---------------------------------------------------------------------------------- */
int vc_beam_update(void* stream, int B, int beam, int Lmax, int eos, double len_norm_f, const float* top_p,
                   const int32_t* top_i, int32_t* pcount, int32_t* ccount, double* p_score, double* p_logprob,
                   int32_t* p_len, const int32_t* sent_cur, int32_t* sent_next, double* c_score, double* c_logprob,
                   int32_t* c_len, int32_t* c_slot, int32_t* c_free, int32_t* c_sent, int32_t* parent, int32_t* tok);

/* The state vc_beam_update starts from, in ONE launch (vae_model/decoder.py:238-247: partial = [Beam([bos], state, 0.0, 0.0)], complete
 * empty): pcount = 1, ccount = 0, p_score = p_logprob = 0, p_len = 1, every sent_cur token = bos, sent_next / c_* / c_sent = 0,
 * c_free = 2^(beam+1)-1, parent[r] = r, tok[r] = bos, and the images' LSTM state expanded to the beam rows:
 * c_out[r] = c_in[r / beam], h_out[r] = h_in[r / beam] ([B, H] -> [B*beam, H]).  Same buffers and shapes as vc_beam_update. */
int vc_beam_init(void* stream, int B, int beam, int Lmax, int bos, int H, const float* c_in, const float* h_in, float* c_out, float* h_out,
                 int32_t* pcount, int32_t* ccount, double* p_score, double* p_logprob, int32_t* p_len, int32_t* sent_cur,
                 int32_t* sent_next, double* c_score, double* c_logprob, int32_t* c_len, int32_t* c_slot, int32_t* c_free,
                 int32_t* c_sent, int32_t* parent, int32_t* tok);

/* One beam-search round's row moves in ONE launch (vae_model/decoder.py:254-262): cg[r] = c[parent[r]], hg[r] = h[parent[r]] ([rows, H]
 * each) and, when xproj != NULL, gact[r] = xproj[tok[r]] ([vocab, G] -> [rows, G]: a word's LSTM input projection E.Wx + b looked up from a
 * table built once per generation call instead of multiplied every round).  H, G multiples of 4, 16-byte aligned pointers. */
int vc_beam_gather_f32(void* stream, const float* c, const float* h, const int32_t* parent, int rows, int H, float* cg, float* hg,
                       const float* xproj, const int32_t* tok, int vocab, int G, float* gact);

/* Stop-word bookkeeping of greedy / sampled decoding (vae_model/decoder.py:186-194), on device: done[b] |= (tok[b] == eos);
 * pending[0] = number of rows that have not emitted eos yet (a float, like the other device scalars). */
int vc_eos_track_i32(void* stream, const int32_t* tok, int B, int eos, int32_t* done, float* pending);

/* ------------------------------------------------------------------------------------
 * Host-side helper (the only entry point that takes HOST pointers): CRC-32C (Castagnoli) of a byte
 * range, continuing from *crc_inout (start with 0).  Used by the TensorFlow V2 checkpoint
 * ("tensor bundle") reader / writer for the per-tensor and per-block checksums
 * (main.py:189-190,286-288 tf.train.Saver; gen_caption.py:113-115 saver.restore).
 * ---------------------------------------------------------------------------------- */
int vc_host_crc32c(const void* data, size_t nbytes, uint32_t* crc_inout);

/* ------------------------------------------------------------------------------------
 * Collectives of the data-parallel step (RCCL over xGMI; csrc/comm.hip).  The reference is single-GPU
 * (utils/parameters.py:163-164 picks one device): these have no counterpart there.  One process per GPU; rank 0 calls
 * vc_comm_unique_id and hands the 128 bytes to every rank out of band (a file, the launcher's store, MPI ...); every rank then calls
 * vc_comm_init_rank (collective: returns when all `world` ranks have joined).  RCCL is bound at run time (the process's
 * already-loaded librccl.so.1 if there is one -- PyTorch bundles its own --, else the ROCm install; VC_RCCL_LIB overrides), so a
 * single-GPU user needs none.  All calls are stream-ordered on `stream` and in place / out of place as declared; any RCCL failure
 * (incl. an asynchronous error an earlier collective left on the communicator) is a non-zero return with the RCCL message in
 * vc_last_error(), and a destroyed / aborted communicator is refused instead of dereferenced.
 *   vc_allreduce_sum_f32      buf[n] <- sum over ranks, in place: the gradient buffer (or one bucket of it), the CE denominator,
 *                             the loss scalars (trainer.py, engine.py: fw_loss)
 *   vc_allgather_f32          out[world * n_per_rank] <- rank-ordered concatenation of in[n_per_rank]: mean / std of the global batch
 *                             before the Q1-mixed latent sample (vae_model/decoder.py:109-110 reshapes across rows)
 *   vc_reducescatter_sum_f32  out[n_per_rank] <- this rank's slice of the sum of in[world * n_per_rank]: the gradients of that mix */
int vc_comm_available(void);                                   /* 1: an RCCL library can be loaded */
int vc_comm_unique_id(void* id128);                            /* rank 0: 128 bytes to distribute */
int vc_comm_init_rank(int world, int rank, const void* id128, int device, void** comm_out);
int vc_comm_info(void* comm, int* world, int* rank, int* rccl_version);
int vc_comm_destroy(void* comm);
int vc_comm_abort(void* comm);
int vc_allreduce_sum_f32(void* comm, void* stream, float* buf, size_t n);
int vc_allgather_f32(void* comm, void* stream, const float* in, float* out, size_t n_per_rank);
int vc_reducescatter_sum_f32(void* comm, void* stream, const float* in, float* out, size_t n_per_rank);

#ifdef __cplusplus
}
#endif
#endif
