/* libcrnn_mi355x.so -- C ABI of the MI355X-native (gfx950) CRNN-OCR hot path.
 *
 * The reference (gasparian/CRNN-OCR-lite) has no native code / FFI: its hot path is the Keras graph built
 * by utils.py:58-96 and driven by Model.fit_generator (train.py:201) / predict_generator (predict.py:166),
 * plus K.ctc_decode (utils.py:353).  This header is the seam a maintainer binds (ctypes) instead of the
 * Keras/TensorFlow calls; each entry point names the reference code it replaces.
 *
 * Conventions: every function returns 0 on success, a positive hipError_t or a negative library code
 * (-2 bad argument, -3 unsupported configuration).  No C++ exception crosses the ABI.  All tensor arguments
 * are caller-owned DEVICE pointers (fp32 unless noted), NHWC / row-major contiguous unless a leading
 * dimension is given.  Every launch is asynchronous on `stream`; nothing allocates or synchronises.
 * One host thread per GPU process; no global mutable state: the library reads no environment variable (schedule A/B switches are
 * crnn_config.flags) and keeps nothing between calls except caches of device properties -- per-device "dynamic LDS size raised" latches
 * (atomic bit masks over device ordinals, so a second device in the same process gets its own hipFuncSetAttribute) and the thread-local
 * events of the side-stream schedules (crnn_backward_*_ex).
 */
#ifndef CRNN_MI355X_H
#define CRNN_MI355X_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* crnn_stream_t; /* hipStream_t (torch.cuda.current_stream().cuda_stream) */

/* ---- model description: CRNN(num_classes, max_string_len, shape=(imgh,imgw,1), time_dense_size, GRU, n_units)
 *      (utils.py:34-41) + per-process batch size ---- */
typedef struct {
  int batch;        /* images per step on this GPU */
  int imgh, imgw;   /* 100 x 32: axis-1 = text width/time axis (utils.py:60) */
  int num_classes;  /* len(lexicon)+1 = 38; blank = num_classes-1 */
  int max_len;      /* max_string_len (labels row length) */
  int tds;          /* time_dense_size */
  int units;        /* n_units (multiple of 64) */
  int gru;          /* 0 = LSTM (utils.py:78-79), 1 = GRU (utils.py:81-82) */
  int stn;          /* 1 = spatial transformer enabled (utils.py:62) */
  int dropout;      /* 1 = Dropout(.1/.4/.2) active in train mode (utils.py:56,75,83); 0 = off (parity runs) */
  int mfma_bf16;    /* 0 = fp32 MFMA everywhere (parity mode); 1 = conv-stack / dense / RNN-input GEMMs multiply in bf16
                       (operands rounded while staged into LDS, fp32 accumulate, fp32 tensors in HBM); 2 = as 1 and the
                       conv-stack activations / gradients are stored as bf16 in HBM (statistics, RNN, CTC, optimizer fp32) */
  int flags;        /* bit set of CRNN_FLAG_* (0 = the default schedule); A/B switches: bit-identical results except where a flag says otherwise */
} crnn_config;
#define CRNN_FLAG_BLOCK1_KERNELS 16384 /* training: block 1's single-channel stage as its stand-alone kernels (depthwise conv, column statistics, BatchNorm-1 apply,
                                         * outer product; backward: weight gradient, data gradient and BatchNorm-1 statistics as three passes) instead of
                                         * crnn_dwconv3x3_c1_fwd / crnn_pw1_bn_fwd / crnn_pw1_bn_bwd; the same values with BatchNorm-1's statistics summed in another order */
#define CRNN_FLAG_LOC_NET_KERNELS 8192  /* spatial transformer: the localisation net as its stand-alone kernels (five launches forward, about ten backward) instead of
                                         * crnn_loc_net_fwd / crnn_loc_net_bwd (one workgroup per sample); forward bit-identical, the localisation net's weight
                                         * gradients the same sums in another order */
#define CRNN_FLAG_NO_BN2_DW_FUSION 4096 /* fp32-storage training, where the two fusions below are the DEFAULT schedule since round 4 (fp32 forms of the same
                                         * kernels; half the elements per byte leave them bandwidth-bound: -0.6 ms of 14.8 per step at batch 256): block outputs
                                         * materialised, BatchNorm-2's backward statistics as a pass of their own.  Forward / data gradients bit-identical,
                                         * BatchNorm-2 gradients the same sums in another order */
#define CRNN_FLAG_BN2_STATS_FUSION 2048 /* bf16 tensors: opt-in (fp32 tensors: the default, see CRNN_FLAG_NO_BN2_DW_FUSION; the statistics in the DX waves) -- with the fusion below: statistics pass of the block outputs' BatchNorm-2 backward inside the next block's
                                         * depthwise-stage backward (crnn_dwconv3x3_bwd_stream_pro with bn2_stat_partials: a twelfth wave) instead of a kernel of
                                         * its own (crnn_bn_bwd_ex); same data gradients bit for bit, BatchNorm-2 gradients / coefficients the same sums in another
                                         * order.  Measured neutral (the pass it removes costs 0.28 ms, the depthwise-stage kernels grow by 0.24): not the default */
#define CRNN_FLAG_BN2_DW_FUSION 1024  /* bf16-storage training: opt-in (fp32 storage: the default schedule, CRNN_FLAG_NO_BN2_DW_FUSION switches it off): the output x = Dropout(ReLU6(BatchNorm-2(q))) of the un-pooled blocks 1, 2, 4, 6 is not
                                         * materialised (crnn_bn_act_pool_drop_ex); the NEXT block's depthwise row-stream kernels apply it to q in LDS
                                         * (crnn_dwconv3x3_fwd_stream_pro, forward; crnn_dwconv3x3_bwd_stream_pro re-forms it in backward; the dropout decisions
                                         * as keep bytes, crnn_dropout_keep_bytes_batch on the side stream).  Bit-identical.  4 of the 8 BatchNorm-apply launches
                                         * and one write + two read passes of those tensors less per step -- but the re-forming is VALU work on a few waves of a
                                         * bandwidth kernel: measured -1.5 % ... +2.5 % step time depending on the box's clocks: not the default */
#define CRNN_FLAG_DW_TILE_KERNEL 32    /* bf16-storage modes: depthwise 3x3 forward and fused depthwise-stage backward on the halo-tile kernels
                                        * (crnn_dwconv3x3_fwd_ex, crnn_dwconv3x3_bwd_fused) instead of the row-stream kernels (crnn_dwconv3x3_fwd_stream,
                                        * crnn_dwconv3x3_bwd_stream); tensors bit-identical, BatchNorm statistics / depthwise weight gradients to
                                        * summation round-off.  fp32 storage (round 4): the round-3 schedule -- halo-tile forward, three-kernel depthwise-stage
                                        * backward, every BatchNorm-2 pass on its own -- instead of the fp32 forms of the row-stream kernels */
#define CRNN_FLAG_NO_DW_BWD_FUSION 16 /* bf16-storage training: depthwise-stage backward as three kernels (BatchNorm backward pass 2, depthwise weight
                                         gradient, depthwise data gradient) instead of crnn_dwconv3x3_bwd_fused / crnn_dwconv3x3_bwd_stream[_ex] (either
                                         storage type; also switches the BatchNorm-2 fusions off); same data gradients bit for bit */
#define CRNN_FLAG_NO_DW_BN_FUSION 8   /* bf16-storage training: materialise a = ReLU6(BN(d)) in a pass of its own instead of applying it while the
                                         pointwise GEMMs stage their operand; bit-identical */
#define CRNN_FLAG_GEMM_TILE_KERNELS 2 /* every pointwise conv of the conv stack (forward, data gradient, weight gradient) on the tile-per-workgroup
                                         GEMM (crnn_gemm_bf16_ex / crnn_pwconv_bnrelu6_*) instead of the streaming kernels (crnn_pwconv_bnrelu6_fwd_wres,
                                         crnn_gemm_wres_bf16, crnn_pwconv_bnrelu6_wgrad_stream).  Same products and data gradients bit for bit; the
                                         BatchNorm-2 statistics and the weight gradients are the same sums in another order (fp32 round-off).  In the parity mode it
                                         also returns the recurrent layers' and dense1's gradients from the round-6 plane streams (crnn_gemm_tn_planes_stream,
                                         crnn_gemm_nt_f32x2_stream) to the tile kernel: same planes and products, another summation order */
#define CRNN_FLAG_THREE_PLANE_BACKWARD 65536 /* parity mode: the backward GEMMs (weight and data gradients of the conv stack, the dense layers and the RNN projections)
                                         with three bf16 planes per operand (fp32-accurate, as the forward) instead of two (crnn_gemm_f32x2*: 16 significant
                                         bits per factor, gradients within 1e-5 of these, half the MFMA work -- the default since round 4) */
#define CRNN_FLAG_TWO_PLANE_FORWARD 131072 /* opt-in, parity mode: the conv stack's forward pointwise GEMMs with two planes too (logits move by ~1e-5: inside
                                         north_star's 1e-3, outside "fp32-accurate") */
#define CRNN_FLAG_NO_GRADIENT_PLANES 262144 /* parity mode, two-plane backward: keep BatchNorm-2's input gradient of every block an fp32 tensor that both pointwise
                                         GEMMs split while staging (rounds 4-5) instead of two bf16 planes written by the BatchNorm backward and read by
                                         crnn_gemm_pres_bnstats / crnn_pwconv_bnrelu6_wgrad_planes_stream_gp (round 6: same words and products; the data gradient
                                         bit-identical, the weight gradient and the statistics the same sums in another order) */
#define CRNN_FLAG_NO_POOL_ARGMAX_Q 524288 /* the pooled blocks (3, 5): the statistics pass of BatchNorm-2's backward re-reads the whole pre-BatchNorm tensor and finds each
                                         window's arg-max again (rounds 1-5) instead of reading the value the forward saved per window (round 6: bit-identical sums) */
#define CRNN_FLAG_WEIGHT_PLANES 32768    /* opt-in, parity mode: the pointwise GEMMs read bf16 planes of their weights split once per step (crnn_split3_planes +
                                         the *_pl entry points) instead of splitting the fp32 weights in every tile that stages them; bit-identical.
                                         Measured (profiles/r04_x3_planes_bench.txt): forward -2 %, data gradient +12 % -- three 8-byte loads per item
                                         cost the staging waves more than the split arithmetic they save; step time unchanged.  Not the default */
#define CRNN_FLAG_DEFERRED_SUMS 512      /* opt-in: second stage of every streaming weight gradient batched at the end of its backward stage
                                         (crnn_wgrad_sum_batch: 2 launches instead of 13 per step) instead of right after its first stage;
                                         bit-identical; measured neutral (6.583 vs 6.584 ms: the second stages are bandwidth, not launch latency) */
#define CRNN_FLAG_F32_MFMA_GEMMS 256    /* parity mode (mfma_bf16 = 0): conv-stack / dense / RNN-projection GEMMs on v_mfma_f32_32x32x2_f32 (crnn_gemm_f32:
                                         an fmaf chain bit for bit) instead of three-plane bf16 products (crnn_gemm_f32x3: fp32-level accuracy, six bf16
                                         MFMAs per k-step, 1.2-1.7x faster per GEMM); results agree to fp32 round-off */
#define CRNN_FLAG_NO_BN_STATS_FUSION 128 /* bf16-storage training: statistics pass of the depthwise BatchNorm's backward as a kernel of its own
                                         (crnn_bn_bwd_ex) instead of inside the data-gradient GEMM (crnn_gemm_wres_bf16_bnstats); same data gradients
                                         bit for bit, the BatchNorm-1 gradients / coefficients are the same sums in another order */
#define CRNN_FLAG_RNN_LINEAR_CLUSTERS 64 /* persistent recurrences: cluster members = consecutive workgroup ids (dealt over all XCDs, write-through
                                         exchange stores) instead of the XCD-local map (plain stores inside a verified one-XCD cluster); bit-identical */
#define CRNN_FLAG_RNN_STEP_KERNELS 1  /* LSTM / GRU recurrences as one (two) launch(es) per timestep (crnn_lstm_*_ex, crnn_gru_*_ex) instead of
                                         the persistent one-launch-per-layer kernels (crnn_lstm_*_persist, crnn_gru_*_persist); bit-identical */

/* ---- parameter / statistics layout (Keras weight order) -------------------------------------- */
int  crnn_num_params(const crnn_config* cfg);                 /* number of trainable tensors */
long crnn_params_total(const crnn_config* cfg);               /* floats in the flat buffer (16-B padded tensors) */
int  crnn_param_info(const crnn_config* cfg, int idx, char* name, int name_cap, long* offset, long* size,
                     int* ndim, int* dims /* [4] */);
int  crnn_bn_total(const crnn_config* cfg);                   /* channels over all 14 BatchNorm layers (3969) */
int  crnn_bn_info(const crnn_config* cfg, int idx, char* name, int name_cap, int* offset, int* channels, long* count);
int  crnn_time_steps(const crnn_config* cfg);                 /* T = (imgh+4)/2 (utils.py:72) */

/* ---- workspace (activations kept for the backward pass + scratch) ------------------------------------------ */
size_t crnn_workspace_bytes(const crnn_config* cfg);
/* named view into the workspace (float offset, element count) -- for parity tests / debugging */
int  crnn_ws_tensor(const crnn_config* cfg, const char* name, long* offset, long* count);
/* 1 when a training forward does not materialise the output "x<block>" of conv block `block` (1..7): the next block's depthwise row-stream kernels
 * form it from "q<block>" in LDS (CRNN_FLAG_BN2_DW_FUSION set and the shape rules hold); else 0 */
int  crnn_block_output_fused(const crnn_config* cfg, int block);
/* same + storage type of the tensor (0 = fp32, 1 = bf16; offset stays in floats, count in elements) */
int  crnn_ws_tensor_info(const crnn_config* cfg, const char* name, long* offset, long* count, int* dtype);


/* ---- whole-path drivers ------------------------------------------------------------------------------------- */
/* Forward of the predictor sub-model (utils.py:308-312; Model.predict_generator, predict.py:166):
 * x [B,imgh,imgw,1] -> y_pred [B,T,num_classes] softmax.  train=1: batch statistics + dropout(seed)
 * (learning_phase=1), activations kept in ws for crnn_backward; train=0: moving statistics, no dropout. */
int crnn_forward(const crnn_config* cfg, const float* params, const float* bn_mean, const float* bn_var,
                 const float* x, float* ws, size_t ws_bytes, float* y_pred, int train, uint64_t seed,
                 crnn_stream_t stream);
/* the same with a side stream (NULL: none) for work nothing at the head of the forward waits for -- the dropout keep bytes of the training forward
 * (crnn_dropout_keep_bytes_batch) next to the spatial transformer's small kernels; joined inside, same results */
int crnn_forward_ex(const crnn_config* cfg, const float* params, const float* bn_mean, const float* bn_var,
                 const float* x, float* ws, size_t ws_bytes, float* y_pred, int train, uint64_t seed,
                 crnn_stream_t stream, crnn_stream_t aux_stream);
/* CTC loss (utils.py:98-103) + full backward of the graph after a train=1 crnn_forward on the same ws.
 * labels [B,max_len] int32 (blank-padded), input_length/label_length [B] int32 (Readf batch contract,
 * utils.py:485-500).  loss [B] = per-sample CTC cost (the model's 'ctc' output); grads (flat, same layout as
 * params) = d mean(loss) / d params  (model.compile(loss={'ctc': y_pred}), train.py:192). */
int crnn_backward(const crnn_config* cfg, const float* params, float* grads, const float* x, const int* labels,
                  const int* input_length, const int* label_length, float* ws, size_t ws_bytes, float* loss,
                  uint64_t seed, crnn_stream_t stream);
/* The same backward in two stages, for data-parallel hosts that overlap the gradient exchange with compute:
 * crnn_backward_top = CTC, dense2, recurrent layers, dense1 (zeroes grads first; fills grads[crnn_grad_split_offset..]),
 * crnn_backward_bottom = conv stack + spatial transformer (fills grads[0..crnn_grad_split_offset)).  Calling top then
 * bottom on one stream == crnn_backward. */
long crnn_grad_split_offset(const crnn_config* cfg);
int crnn_backward_top(const crnn_config* cfg, const float* params, float* grads, const int* labels, const int* input_length,
                      const int* label_length, float* ws, size_t ws_bytes, float* loss, uint64_t seed, crnn_stream_t stream);
int crnn_backward_bottom(const crnn_config* cfg, const float* params, float* grads, const float* x, float* ws, size_t ws_bytes,
                         uint64_t seed, crnn_stream_t stream);
/* Same results, with a caller-owned second stream (aux_stream, NULL = serial): the weight-gradient GEMMs of dense2 and of the
 * upper recurrent layer run there while the BPTT chains (2 x T dependent launches of a few microseconds, GPU almost
 * idle) run on `stream`; events fork / join the two inside the call, everything is complete on `stream` order-wise when
 * the call's last kernel is.  Bit-identical to the serial schedule. */
int crnn_backward_top_ex(const crnn_config* cfg, const float* params, float* grads, const int* labels, const int* input_length,
                         const int* label_length, float* ws, size_t ws_bytes, float* loss, uint64_t seed, crnn_stream_t stream,
                         crnn_stream_t aux_stream);
/* crnn_backward_bottom with a side stream: the pointwise weight-gradient GEMM of every block runs there, next to the block's
 * data-gradient GEMM, BatchNorm statistics pass and fused depthwise stage on `stream` (the gradient buffers rotate over three
 * allocations; the main stream waits for a GEMM before the buffer it reads is rewritten).  Bit-identical to the serial schedule. */
int crnn_backward_bottom_ex(const crnn_config* cfg, const float* params, float* grads, const float* x, float* ws, size_t ws_bytes,
                            uint64_t seed, crnn_stream_t stream, crnn_stream_t aux_stream);
int crnn_backward_ex(const crnn_config* cfg, const float* params, float* grads, const float* x, const int* labels,
                     const int* input_length, const int* label_length, float* ws, size_t ws_bytes, float* loss, uint64_t seed,
                     crnn_stream_t stream, crnn_stream_t aux_stream);
/* BatchNorm moving-average update from the batch statistics left in ws by a train=1 forward (momentum .99) */
int crnn_bn_update(const crnn_config* cfg, float* bn_mean, float* bn_var, float* ws, size_t ws_bytes,
                   crnn_stream_t stream);

/* One whole single-GPU train step on one stream, what Keras' train_on_batch does for this model (train.py:201-209):
 * crnn_forward(train=1) -> crnn_backward -> crnn_global_norm(clipnorm) -> crnn_adam_step -> crnn_bn_update.
 * m, v: Adam moments (same layout as params); lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) is computed by the caller
 * (Keras 2.2.2 form, t = 1-based iteration); norm_scratch >= 4096 bytes, norm_out 2 floats (global norm, clip factor).
 * Data-parallel hosts call the stages themselves (crnn_backward_top / _bottom around their all-reduce). */
int crnn_train_step_adam(const crnn_config* cfg, float* params, float* grads, float* m, float* v, float* bn_mean, float* bn_var,
                         const float* x, const int* labels, const int* input_length, const int* label_length, float* ws,
                         size_t ws_bytes, float* y_pred, float* loss, void* norm_scratch, float* norm_out, float lr_t, float beta1,
                         float beta2, float eps, float clipnorm, uint64_t seed, crnn_stream_t stream);

/* ---- optimizers (keras.optimizers.Adam / SGD with clipnorm, train.py:187-190) ------------------------------ */
/* norm_out[0] = global L2 norm of g, norm_out[1] = clip multiplier; scratch >= 4096 bytes */
int crnn_global_norm(const float* g, long n, float clipnorm, void* scratch, float* norm_out, crnn_stream_t stream);
int crnn_adam_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                   float eps, const float* norm_out, crnn_stream_t stream);
int crnn_sgd_step(float* p, const float* g, float* vel, long n, float lr, float momentum, int nesterov,
                  const float* norm_out, crnn_stream_t stream);
int crnn_scale(float* x, long n, float s, crnn_stream_t stream);
/* y (bf16) = round-to-nearest-even(x); n % 4 == 0 */
int crnn_convert_f32_to_bf16(const float* x, void* y, long n, crnn_stream_t stream);

/* ---- decoding (DecodeCTCPred.decode -> K.ctc_decode, utils.py:347-357) ------------------------------------- */
/* y [B,T,C] softmax; out [B,T] int32 padded with -1; out_len [B]; input_len may be NULL (= T) */
int crnn_ctc_greedy_decode(const float* y, const int* input_len, int* out, int* out_len, int B, int T, int C,
                           crnn_stream_t stream);
/* tf.nn.ctc_beam_search_decoder(beam_width <= 64, top_paths=1, merge_repeated); scores [B] = log-score of the
 * best beam (sum of max-shifted log-probs, as TF r1.8 accumulates it).  State lives in LDS: no workspace. */
int crnn_ctc_beam_decode(const float* y, const int* input_len, int* out, int* out_len, float* scores, int B, int T,
                         int C, int beam_width, int merge_repeated, crnn_stream_t stream);

/* ---- individual operators (unit-tested one by one; the drivers above chain them) --------------------------- */
/* mode 0: C=A[M,K]*B[K,N]; 1: C=A[M,K]*Bt[N,K]^T; 2: C=At[K,M]^T*B[K,N].  bias[N]|NULL, act 0|1(relu),
 * accumulate: C+=, permP: out_row=(m%P)*(M/P)+m/P (0=off), scratch: split-reduction partials (may be NULL) */
int crnn_gemm_f32(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                  const float* bias, int act, int accumulate, int permP, float* scratch, size_t scratch_bytes,
                  crnn_stream_t stream);
/* The same contract with fp32-accurate products from three bf16 planes per operand: x = hi + mid + lo (bf16 each, |x - sum| <= 2^-27 |x|), a
 * product keeps hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi (dropped terms <= 2^-26 of it), every partial product is exact in the MFMA's fp32
 * accumulator: six v_mfma_f32_32x32x16_bf16 per k-step instead of eight four-times slower v_mfma_f32_32x32x2_f32.  Equal to crnn_gemm_f32 to
 * fp32 round-off, not bit for bit.  The parity mode's GEMMs (CRNN_FLAG_F32_MFMA_GEMMS selects crnn_gemm_f32 instead). */
int crnn_gemm_f32x3(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                    const float* bias, int act, int accumulate, int permP, float* scratch, size_t scratch_bytes, crnn_stream_t stream);
/* same contract, products in bf16 on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 operands/result in HBM) */
int crnn_gemm_bf16(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                   const float* bias, int act, int accumulate, int permP, float* scratch, size_t scratch_bytes,
                   crnn_stream_t stream);
/* storage-typed variants (dt*: 0 = fp32, 1 = bf16 tensors in HBM; arithmetic stays fp32 / bf16-MFMA with fp32 accumulate) */
int crnn_gemm_bf16_ex(int mode, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                      const float* bias, int act, int accumulate, int permP, float* scratch, size_t scratch_bytes, int dtA,
                      int dtB, int dtC, crnn_stream_t stream);
/* Pointwise 1x1 convolution (reference utils.py:49, Conv2D(1x1, no bias)) as one GEMM over the pixels:
 * q[M][N] = a[M][K] * w[K][N] (w_transposed = 1: w is given as W^T [N][K]).
 * Training: stat_partials (may be NULL) receives [crnn_pwconv_stat_rows(M)][2][N] per-tile column sums / sums of
 * squares of q AS STORED (i.e. after rounding to dt_q): the batch statistics of the BatchNorm that follows
 * (utils.py:50), produced by the GEMM epilogue instead of a separate pass over q.  Deterministic.
 * Inference: out_bnstate (may be NULL; [mean|var|scale|shift] from crnn_bn_infer_state) folds that BatchNorm and the
 * ReLU6 after it (utils.py:50-51) into the epilogue: q = ReLU6(product * scale + shift).  Not both at once.
 * bf16_products: 0 = fp32 MFMA (all dt_* must be 0), 1 = bf16 MFMA products with fp32 accumulation, 2 = fp32 tensors with fp32-accurate
 * three-plane bf16 products (crnn_gemm_f32x3; all dt_* must be 0). */
int crnn_pwconv_stat_rows(long M);
int crnn_pwconv_fwd(const void* a, const void* w, void* q, long M, int N, int K, float* stat_partials, const float* out_bnstate,
                    int bf16_products, int dt_a, int dt_w, int dt_q, int w_transposed, crnn_stream_t stream);
/* Producer-fused form for training with bf16 conv-stack tensors: `d` is the depthwise output BEFORE its BatchNorm
 * (utils.py:44) and in_bnstate = [mean|var|scale|shift] of that BatchNorm (crnn_bn_finalize*); the GEMM applies
 * a = ReLU6(d * scale + shift) (utils.py:45-46), rounded to bf16 exactly as crnn_bn_act_pool_drop_ex would store it,
 * while it stages the operand -- the activated tensor `a` is never materialised (forward and weight gradient):
 *   fwd  : q[M][N] = a[M][K] * w     (+ stat_partials as for crnn_pwconv_fwd; w bf16, [K][N] or W^T [N][K])
 *   wgrad: dw[K][N] (fp32) = a^T * g[M][N]   (g bf16; scratch for the split reduction as for crnn_gemm_bf16_ex)
 * d, w, g are bf16; K % 8 == 0, K <= 512.  Results are bit-identical to the two-pass path. */
int crnn_pwconv_bnrelu6_fwd(const void* d, const float* in_bnstate, const void* w, void* q, long M, int N, int K,
                            float* stat_partials, int dt_q, int w_transposed, crnn_stream_t stream);
int crnn_pwconv_bnrelu6_wgrad(const void* d, const float* in_bnstate, const void* g, float* dw, long M, int N, int K,
                              float* scratch, size_t scratch_bytes, crnn_stream_t stream);
/* Parity mode (fp32 tensors, three-plane products, crnn_gemm_f32x3): the same two GEMMs fed by the PRE-BatchNorm depthwise output d [M][K] fp32; the
 * staging waves apply ReLU6(d * scale[ch] + shift[ch]) (crnn_bn_act_pool_drop_ex's arithmetic, bit for bit) before the plane split, so the activated
 * tensor is never written.  w [K][N] fp32, K <= 512; results equal the unfused sequence bit for bit; -3 outside the kernel's shape rules. */
int crnn_pwconv_bnrelu6_fwd_f32x3(const float* d, const float* in_bnstate, const float* w, float* q, long M, int N, int K, float* stat_partials,
                                  crnn_stream_t stream);
/* ... with the weights as planes (crnn_split3_planes of w [K][N]; null: the entry point above).  Whole tiles only (-3 otherwise). */
int crnn_pwconv_bnrelu6_fwd_f32x3_pl(const float* d, const float* in_bnstate, const float* w, const void* w_planes, long w_plane_stride, float* q, long M,
                                     int N, int K, float* stat_partials, crnn_stream_t stream);
int crnn_pwconv_bnrelu6_wgrad_f32x3(const float* d, const float* in_bnstate, const float* g, float* dw, long M, int N, int K, float* scratch,
                                    size_t scratch_bytes, crnn_stream_t stream);
/* Round 6: the same forward product and the data gradient with the WEIGHTS' PLANES RESIDENT IN REGISTERS (gemm_wres3.hip; replaces Keras' Conv2D(1x1)
 * forward / backward of reference utils.py:48-49 in the parity mode): a workgroup owns a slice of 128 output channels (64 at K = 512, where the reduction
 * runs as two halves on two waves each), splits its slice of w into planes once, and the pixel rows stream through it once -- IO waves load fp32 rows,
 * apply BatchNorm-1 + ReLU6 (forward), split into planes into an LDS ring; the result leaves through LDS staging tiles whose drain also takes the
 * statistics.  planes = 3 | 2.  Results: K <= 256 bit-identical to crnn_pwconv_bnrelu6_fwd_f32x3 / _f32x2 and crnn_gemm_f32x3_bnstats / _f32x2_bnstats;
 * K = 512 equal up to the order of ONE addition (the two half-reduction chains are added at the end).  Statistics: the same sums in another order,
 * [crnn_gemm_wres3_stat_rows(M, N, K)][2][N], every element written.  Shapes (crnn_gemm_wres3_supported, -3 otherwise): K in {64, 128, 256, 512}
 * (data gradient: 256 | 512), N a multiple of 128 (64 at K = 512) up to 1024, M a multiple of 64 (32 at K >= 256), 16-byte aligned tensors. */
int crnn_gemm_wres3_supported(long M, int N, int K);
int crnn_gemm_wres3_stat_rows(long M, int N, int K);
int crnn_pwconv_bnrelu6_fwd_wres3(const float* d, const float* in_bnstate, const float* w, float* q, long M, int N, int K, int planes,
                                  float* stat_partials, crnn_stream_t stream);
/* da[M][N] = dq[M][K] . w[N][K]^T (w = the convolution's kernel [N input channels][K output channels]) + BatchNorm-1 backward statistics:
 * stat_partials = partial sums of gy and gy * xhat, gy = da where 0 < d * scale + shift < 6; d [M][N], bnstate = [mean|var|scale|shift] x N
 * (as crnn_gemm_f32x3_bnstats; feed crnn_bn_bwd_finalize_folded). */
int crnn_gemm_wres3_bnstats(const float* dq, const float* w, float* da, long M, int N, int K, int planes, const float* d, const float* bnstate,
                            float* stat_partials, crnn_stream_t stream);
/* ... and the weight gradient dw[K][N] = ReLU6(d * scale + shift)^T [K][M] . g[M][N] of the same convolution with TWO bf16 planes per operand (the parity
 * mode's default backward precision) on a pixel stream (gemm_wgrad3.hip): a workgroup keeps one 128 x 128 tile of dw in registers over a contiguous range of
 * 32-pixel chunks, IO waves load the fp32 rows, apply BatchNorm-1 + ReLU6 (in_bnstate; NULL: dw = d^T . g), split both operands into planes and feed an LDS
 * ring; fp32 partial tiles in `scratch` ([ranges][K][N], crnn_pwconv_wgrad_planes_stream_scratch_bytes), fixed-order second stage.  Same planes and products as
 * crnn_pwconv_bnrelu6_wgrad_f32x2; other reduction ranges: equal to fp32 summation round-off.  Shapes (-3 otherwise): M % 32 == 0, K and N multiples of 128 up to
 * 1024 with at most 32 tiles, 16-byte aligned tensors.  d [M][K], g [M][N] fp32. */
int crnn_pwconv_wgrad_planes_stream_supported(long M, int N, int K);
size_t crnn_pwconv_wgrad_planes_stream_scratch_bytes(long M, int N, int K);
int crnn_pwconv_bnrelu6_wgrad_planes_stream(const float* d, const float* in_bnstate, const float* g, float* dw, long M, int N, int K, float* scratch,
                                            size_t scratch_bytes, crnn_stream_t stream);
/* ... with g given as its two bf16 planes (hi plane [M][N], the mid plane g_plane_stride elements behind it: crnn_bn_bwd_planes_ex's output); bit-identical. */
int crnn_pwconv_bnrelu6_wgrad_planes_stream_gp(const float* d, const float* in_bnstate, const void* g_planes, long g_plane_stride, float* dw, long M, int N, int K,
                                               float* scratch, size_t scratch_bytes, crnn_stream_t stream);
/* The planes stream without a transform and with leading dimensions (round 6): C[M][N] (row stride ldc) = A^T . B over the K rows of A [K][lda] and B [K][ldb], fp32,
 * two bf16 planes per operand (crnn_gemm_f32x2's precision, another summation order) -- the parity mode's recurrent weight gradients (utils.py:77-82 backwards).
 * Supported (else -3): crnn_pwconv_wgrad_planes_stream_supported(K, N, M), leading dimensions % 4 == 0, 16-byte aligned pointers;
 * scratch: crnn_pwconv_wgrad_planes_stream_scratch_bytes(K, N, M) */
int crnn_gemm_tn_planes_stream(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                               size_t scratch_bytes, crnn_stream_t stream);
/* Input gradient of a Bidirectional recurrent layer's input projections in the parity mode (round 6; utils.py:77-82 backwards): Y[M][N] (row stride ldy) =
 * A0[M][K] . W0[N][K]^T (+ A1 . W1^T when A1 != NULL), everything fp32, two bf16 planes per operand (crnn_gemm_f32x2's precision, another summation order), one
 * workgroup per 64-row stripe and 128-column slab over the whole reduction of both pairs.  Supported (else -3): M % 64 == 0, N % 128 == 0, K % 64 == 0,
 * leading dimensions % 4 == 0, 16-byte aligned pointers */
int crnn_gemm_nt_f32x2_stream(const float* A0, const float* W0, const float* A1, const float* W1, float* Y, int M, int N, int K, int lda, int ldw, int ldy,
                              crnn_stream_t stream);
/* The data gradient from PRE-SPLIT planes (round 6, gemm_pres.hip): da[M][N] = dq[M][K] . w[N][K]^T with dq given as bf16 planes
 * (plane pl of dq[m][k] at dq_planes[pl * plane_stride + m * K + k]: the words of crnn_split3_planes; planes = 2 | 3) -- written once by the kernel that produces
 * dq (crnn_bn_bwd_planes_ex) instead of being split by every slice of every GEMM that reads it.  Four waves per workgroup, one per SIMD, the planes of a
 * 128-channel slice of w resident in up to 512 registers, the pixel planes and the rows of d by LDS-DMA.  stat_partials
 * [crnn_gemm_pres_stat_rows][2][N]: partial sums of gy and gy * xhat (as crnn_gemm_wres3_bnstats; feed crnn_bn_bwd_finalize_folded).  a_planes (may be NULL):
 * a_count (2 | 3) planes of a = ReLU6(d * scale + shift) [a_count][M][N] at element stride a_stride, the other operand of the same convolution's weight
 * gradient (crnn_pwconv_wgrad_planes).  Same planes and products as crnn_gemm_f32x2_bnstats / crnn_gemm_f32x3_bnstats, three accumulator chains: fp32 summation
 * order only.  Shapes (-3 otherwise): K in {256, 512} (three planes: 256), N a multiple of 128 up to 1024, M % 32 == 0, 16-byte aligned tensors. */
int crnn_gemm_pres_supported(long M, int N, int K, int planes);
int crnn_gemm_pres_stat_rows(long M, int N, int K, int planes);
int crnn_gemm_pres_bnstats(const void* dq_planes, long plane_stride, const float* w, float* da, long M, int N, int K, int planes, const float* d,
                           const float* bnstate, float* stat_partials, void* a_planes, long a_stride, int a_count, crnn_stream_t stream);
/* The same pointwise conv for ONE input channel (block 1: Conv2D(64, 1x1) on the single-channel depthwise output,
 * utils.py:64 / 49): an outer product q[m][c] = a[m] * w[c], its data gradient da[m] = dq[m] . w and weight gradient
 * dw[c] = sum_m a[m] dq[m][c].  a / da fp32; q / dq fp32 (dt_q 0) or bf16 (1); N a power of two, 8 <= N <= 256.
 * stat_partials as for crnn_pwconv_fwd; scratch: crnn_colreduce_chunks(M) * N floats; da may be NULL. */
int crnn_pw1_fwd(const float* a, const float* w, void* q, long M, int N, float* stat_partials, int dt_q, crnn_stream_t stream);
/* inference form: y = ReLU6((a (x) w) * scale + shift), out_bnstate = [mean|var|scale|shift] of the BatchNorm after the convolution */
int crnn_pw1_fwd_folded(const float* a, const float* w, void* y, long M, int N, const float* out_bnstate, int dt_y, crnn_stream_t stream);
int crnn_pw1_bwd(const float* a, const float* w, const void* dq, float* da, float* dw, float* scratch, long M, int N, int dt_q,
                 crnn_stream_t stream);
/* Round 4: block 1's single-channel stage with its BatchNorm-1 folded into the neighbouring kernels (d, da fp32 [M]; in_bnstate = [mean|var|scale|shift] of the
 * one-channel BatchNorm).  crnn_dwconv3x3_c1_fwd: out = dwconv3x3(x, k[9]) on [B][H][W] + [crnn_dwconv_c1_stat_rows][2] partial sums / sums of squares of out.
 * crnn_pw1_bn_fwd: q [M][N] = relu6(fma(d, scale, shift)) (x) w -- crnn_bn_act_pool_drop_ex + crnn_pw1_fwd without the activated tensor, the same bits.
 * crnn_pw1_bn_bwd: ONE pass over dq -> dw [N] and da [M] (crnn_pw1_bwd's bits on the re-formed activation) and bn_stat_partials [crnn_pw1_bn_bwd_rows(M)][2] =
 * partial sums of gy and gy * xhat, gy = da where 0 < BN(d) < 6, for crnn_bn_bwd_finalize (C = 1); scratch: crnn_pw1_bn_bwd_rows(M) * N floats. */
int crnn_dwconv_c1_stat_rows(int B, int H, int W);
int crnn_dwconv3x3_c1_fwd(const float* x, const float* k, float* out, float* stat_partials, int B, int H, int W, crnn_stream_t stream);
int crnn_pw1_bn_fwd(const float* d, const float* in_bnstate, const float* w, void* q, long M, int N, float* stat_partials, int dt_q, crnn_stream_t stream);
/* crnn_dwconv3x3_c1_bwd: the one-channel depthwise stage backwards in ONE kernel = crnn_bn_bwd_apply_ex (coef from crnn_bn_bwd_finalize) + crnn_dwconv3x3_wgrad_ex +
 * crnn_dwconv3x3_fwd_ex(flip = 1); dx [B][H][W] (NULL: not wanted) bit-identical, dk [9] to the order of its partial sums; scratch: crnn_dwconv_c1_bwd_rows * 9 floats */
int crnn_dwconv_c1_bwd_rows(int B, int H, int W);
int crnn_dwconv3x3_c1_bwd(const float* d, const float* da, const float* bnstate, const float* coef, const float* x, const float* k, float* dx, float* dk,
                          float* scratch, int B, int H, int W, crnn_stream_t stream);
int crnn_pw1_bn_bwd_rows(long M);
int crnn_pw1_bn_bwd(const float* d, const float* in_bnstate, const float* w, const void* dq, float* da, float* dw, float* scratch, float* bn_stat_partials,
                    long M, int N, int dt_q, crnn_stream_t stream);
/* Depthwise 3x3 with the inference BatchNorm + ReLU6 after it (utils.py:44-46) folded into the epilogue:
 * out = ReLU6(dwconv3x3(x, k) * scale + shift); C must be a multiple of 32 (fp32 storage) / 64 (bf16 storage). */
int crnn_dwconv3x3_bn_relu6_fwd(const void* x, const float* k, const float* bnstate, void* out, int B, int H, int W, int C,
                                int dtype, crnn_stream_t stream);
/* Depthwise 3x3 of a bf16 NHWC map as a row stream (dwconv_stream.hip): a loader wave keeps whole rows (W*C*2 contiguous bytes) in flight
 * into an LDS ring with global_load_lds, the compute waves own a 16-byte column each and add every arriving row's taps to three running output
 * rows (nothing is held or re-read).  stat_partials != NULL: [crnn_dwconv_fwd_stream_rows][2][C] BatchNorm partial sums of the fp32 results;
 * bnstate != NULL ([mean|var|scale|shift]): out = ReLU6(conv * scale + shift) (inference), no statistics; flip = 1: the data gradient.
 * Results bit-identical to crnn_dwconv3x3_fwd_ex / crnn_dwconv3x3_bn_relu6_fwd (reference utils.py:44-46).  _supported: CRNN_OK when the shape
 * is taken -- a row, cut into the smallest number of channel ranges (whole groups of 8 channels) that fit 576 sixteen-byte columns, with a whole number of
 * row bands side by side, fills five to nine compute waves (257..576 columns): every block of the CRNN at image width 32 (576), 48 (416), 64 (544) --, else
 * CRNN_ERR_UNSUPPORTED (the caller runs the halo-tile kernel). */
int crnn_dwconv_fwd_stream_supported(int B, int H, int W, int C);
int crnn_dwconv_fwd_stream_rows(int B, int H, int W, int C);
int crnn_dwconv3x3_fwd_stream(const void* x, const float* k, void* out, float* stat_partials, const float* bnstate, int B, int H, int W, int C,
                              int flip, crnn_stream_t stream);
/* out_order 1 (inference form only: bnstate != NULL; H, W even): the rows of `out` are in 2x2-window-major order -- pixel (y, x) of an image is its row
 * ((y/2) (W/2) + x/2) 4 + (y&1) 2 + (x&1) -- so that crnn_pwconv_fwd_wres_folded_pool(pool_rows = 4) can pool in its epilogue.  0: NHWC. */
int crnn_dwconv3x3_fwd_stream_ex(const void* x, const float* k, void* out, float* stat_partials, const float* bnstate, int B, int H, int W, int C,
                                 int flip, int out_order, crnn_stream_t stream);
/* Prologue form (training, round 4): `q` is the PREVIOUS block's pointwise output and pro_bnstate its BatchNorm-2 state [mean|var|scale|shift]; the
 * kernel convolves x = Dropout(ReLU6(q * scale + shift)) (utils.py:48-56), formed in LDS by two transform waves one row ahead of the compute
 * waves -- the block output x never exists in HBM.  rate > 0: `keep` = crnn_dropout_keep_bytes(B*H*W*C/8 groups, rate, seed, layer) of the dropout
 * site crnn_bn_act_pool_drop_ex would use (one byte per 16-byte chunk); rate == 0: keep may be NULL.  out / stat_partials bit-identical to
 * crnn_bn_act_pool_drop_ex(q -> x, no pooling) + crnn_dwconv3x3_fwd_stream(x).  _supported: the stream shape rule, 128 % (C/8) == 0 and fewer than
 * 2^32 dropout groups (B*H*W*C/8). */
int crnn_dwconv_fwd_stream_pro_supported(int B, int H, int W, int C);
int crnn_dwconv3x3_fwd_stream_pro(const void* q, const float* pro_bnstate, float rate, const void* keep, const float* k, void* out,
                                  float* stat_partials, int B, int H, int W, int C, crnn_stream_t stream);
/* n (<= 8) independent matrix transposes in one launch: out[i] [C_i][R_i] = in[i]^T, in[i] = src + in_off[i] (fp32
 * elements), out[i] = dst + out_off[i] (elements of dt_out: 0 fp32 | 1 bf16).  src / dst are device pointers; the four
 * descriptor arrays (in_off, out_off, R, C) are HOST arrays of n entries, copied into the kernel arguments. */
int crnn_transpose_batch(const float* src, void* dst, int n, const long* in_off, const long* out_off, const int* R, const int* C,
                         int dt_out, crnn_stream_t stream);
int crnn_dwconv3x3_fwd_ex(const void* x, const float* k, void* out, float* stat_partials, int B, int H, int W, int C, int flip,
                          int dtype, crnn_stream_t stream);
int crnn_dwconv3x3_wgrad_ex(const void* x, const void* g, float* dk, float* scratch, int B, int H, int W, int C, int dtype,
                            crnn_stream_t stream);
int crnn_colreduce_ex(const void* x, float* partials, long M, int C, int ld, int nv, int dtype, crnn_stream_t stream);
int crnn_bn_act_pool_drop_ex(const void* x, const float* bnstate, void* y, int B, int H, int W, int C, int ph, int pw, float rate,
                             uint64_t seed, uint32_t layer, int dt_in, int dt_out, crnn_stream_t stream);
int crnn_bn_bwd_ex(const void* x, const void* g, const float* bnstate, const float* gamma, void* dx, float* dgamma, float* dbeta,
                   float* scratch_partials, float* coef, int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed,
                   uint32_t layer, int dtype, crnn_stream_t stream);
/* DepthwiseConv2D 3x3 'same' (utils.py:44): k [9][C]; flip=1 = data gradient; stat_partials [tiles][2][C] */
int crnn_dwconv_num_tiles(int B, int H, int W);
/* Backward of one depthwise stage (DepthwiseConv2D(3x3) -> BatchNormalization -> ReLU(6.), utils.py:44-46) in one kernel, bf16 storage:
 * from d (depthwise output), da = dL/d ReLU6(BN(d)), the BatchNorm state and the coefficients coef = [mean(gy) | mean(gy*xhat)] that
 * crnn_bn_bwd_ex(..., dx = NULL, ...) leaves (first pass + finalize only), it forms the BatchNorm-input gradient in its halo-tile fill
 * (never written to HBM) and produces dx = dL/d(depthwise input) and dk [9][C] = dL/d(depthwise kernel): 4 tensor passes instead of
 * the 7 of crnn_bn_bwd_ex pass 2 + crnn_dwconv3x3_wgrad_ex + crnn_dwconv3x3_fwd_ex(flip).  dx is bit-identical to that sequence.
 * scratch: crnn_dwconv_bwd_fused_rows(B, H, W, C) * 9 * C floats.  Supported: C % 64 == 0, any W (column tiles of <= 24 pixels) (else -3). */
int crnn_dwconv_bwd_fused_supported(int H, int W, int C);
int crnn_dwconv_bwd_fused_rows(int B, int H, int W, int C);
int crnn_dwconv3x3_bwd_fused(const void* d, const void* da, const float* bnstate, const float* coef, const void* xin, const float* k, void* dx,
                             float* dk, float* scratch, int B, int H, int W, int C, crnn_stream_t stream);
/* The same stage as a row stream (dwconv_bwd_stream.hip): a loader wave brings one row each of d, da and xin per step into an LDS ring
 * (global_load_lds); five "DK" waves form dd of the arriving row and keep the 72 weight-gradient sums of their 16-byte column, five "DX"
 * waves run the three running output rows of the data gradient on the dd row the others left in LDS.  Same contract and arithmetic as
 * crnn_dwconv3x3_bwd_fused (dx bit-identical, dk to the order of its partial sums); scratch: crnn_dwconv_bwd_stream_rows() * 9 * C floats.
 * _supported: CRNN_OK when W * C / 8 sixteen-byte columns split over whole channel octets into workgroups of 129..320 columns (every block
 * of the CRNN), else CRNN_ERR_UNSUPPORTED. */
int crnn_dwconv_bwd_stream_supported(int B, int H, int W, int C);
int crnn_dwconv_bwd_stream_rows(int B, int H, int W, int C);
int crnn_dwconv3x3_bwd_stream(const void* d, const void* da, const float* bnstate, const float* coef, const void* xin, const float* k, void* dx,
                              float* dk, float* scratch, int B, int H, int W, int C, crnn_stream_t stream);
/* The same stage by storage type (dtype: CRNN_BF16 = the three entry points above; CRNN_F32, round 4: four channels per lane, W * C / 4 columns
 * in workgroups of 129..320): the parity mode's crnn_bn_bwd_apply_ex + crnn_dwconv3x3_wgrad_ex + crnn_dwconv3x3_fwd_ex(flip = 1) in one pass over
 * d, da, xin -- 4 tensor passes instead of 7; dx bit-identical to that sequence, dk to the order of its partial sums. */
int crnn_dwconv_bwd_stream_supported_ex(int B, int H, int W, int C, int dtype);
int crnn_dwconv_bwd_stream_rows_ex(int B, int H, int W, int C, int dtype);
int crnn_dwconv3x3_bwd_stream_ex(const void* d, const void* da, const float* bnstate, const float* coef, const void* xin, const float* k, void* dx,
                                 float* dk, float* scratch, int B, int H, int W, int C, int dtype, crnn_stream_t stream);
/* Prologue form: `q` (in place of xin) is the previous block's pointwise output; x = Dropout(ReLU6(q * scale + shift)) is re-formed in LDS by a
 * DX waves one row ahead (the forward did not keep it: crnn_dwconv3x3_fwd_stream_pro; same `keep` bytes).  dx / dk bit-identical to
 * crnn_dwconv3x3_bwd_stream on the materialised x.  bn2_stat_partials != NULL: the DX waves also take the statistics pass of the producer's BatchNorm-2
 * backward -- [crnn_dwconv_bwd_stream_rows][2][C] partial sums (sum gy | sum gy * xhat), gy = dx through the dropout mask and the ReLU6 gate of q: what
 * crnn_bn_bwd_ex(q, dx, pro_bnstate, ...) would read q and dx again for; finish with crnn_bn_bwd_finalize + crnn_bn_bwd_apply_ex. */
int crnn_dwconv_bwd_stream_pro_supported(int B, int H, int W, int C);
int crnn_dwconv3x3_bwd_stream_pro(const void* d, const void* da, const float* bnstate, const float* coef, const void* q, const float* pro_bnstate,
                                  float rate, const void* keep, const float* k, void* dx, float* dk, float* scratch, float* bn2_stat_partials,
                                  int B, int H, int W, int C, crnn_stream_t stream);
/* Round 4: the row-stream kernels on fp32 maps (the parity mode; dtype CRNN_F32, CRNN_BF16 = the entry points above).  Rows of 18 KiB run as two
 * channel ranges of 9 KiB (one workgroup each), four channels per lane; the keep bytes (still one per 8 elements) are read as nibbles.
 * crnn_dwconv3x3_fwd_stream_dt: training form (statistics, no folded BatchNorm); out bit-identical to crnn_dwconv3x3_fwd_ex on fp32 tensors.
 * The prologue forms: bit-identical to crnn_bn_act_pool_drop_ex(q -> x) + the plain forms on the materialised x, as for bf16. */
int crnn_dwconv_fwd_stream_supported_ex(int B, int H, int W, int C, int dtype);
int crnn_dwconv_fwd_stream_rows_ex(int B, int H, int W, int C, int dtype);
int crnn_dwconv3x3_fwd_stream_dt(const void* x, const float* k, void* out, float* stat_partials, int B, int H, int W, int C, int flip, int dtype,
                                 crnn_stream_t stream);
int crnn_dwconv_fwd_stream_pro_supported_ex(int B, int H, int W, int C, int dtype);
int crnn_dwconv3x3_fwd_stream_pro_ex(const void* q, const float* pro_bnstate, float rate, const void* keep, const float* k, void* out,
                                     float* stat_partials, int B, int H, int W, int C, int dtype, crnn_stream_t stream);
int crnn_dwconv_bwd_stream_pro_supported_ex(int B, int H, int W, int C, int dtype);
int crnn_dwconv3x3_bwd_stream_pro_ex(const void* d, const void* da, const float* bnstate, const float* coef, const void* q, const float* pro_bnstate,
                                     float rate, const void* keep, const float* k, void* dx, float* dk, float* scratch, float* bn2_stat_partials,
                                     int B, int H, int W, int C, int dtype, crnn_stream_t stream);
int crnn_dwconv3x3_fwd(const float* x, const float* k, float* out, float* stat_partials, int B, int H, int W, int C,
                       int flip, crnn_stream_t stream);
int crnn_dwconv3x3_wgrad(const float* x, const float* g, float* dk, float* scratch, int B, int H, int W, int C,
                         crnn_stream_t stream);
/* column reductions of a [M][C] matrix (sum / sum+sumsq) and their second stage */
int crnn_colreduce_chunks(long M);
int crnn_colreduce(const float* x, float* partials, long M, int C, int ld, int nv, crnn_stream_t stream);
int crnn_partials_sum(const float* partials, int nparts, int n, float* out, float scale, crnn_stream_t stream);
/* BatchNormalization(axis=-1, eps 1e-3) (utils.py:45,48): bnstate = [mean|var|scale|shift] */
int crnn_bn_finalize(const float* partials, int nparts, int C, long n, const float* gamma, const float* beta,
                     float* bnstate, crnn_stream_t stream);
/* same result for long partial lists (e.g. one row per GEMM tile): folds the rows into 32 chunk sums first;
 * scratch: 32 * 2 * C floats (NULL or nparts <= 1024: plain crnn_bn_finalize) */
int crnn_bn_finalize_folded(const float* partials, int nparts, int C, long n, const float* gamma, const float* beta,
                            float* bnstate, float* scratch, crnn_stream_t stream);
int crnn_bn_infer_state(const float* mmean, const float* mvar, const float* gamma, const float* beta, int C,
                        float* bnstate, crnn_stream_t stream);
/* ... for n <= CRNN_BN_INFER_BATCH_MAX layers in one launch: host arrays of n device pointers (moving mean / variance, gamma, beta, bnstate out) and
 * channel counts.  Same values as n crnn_bn_infer_state calls. */
#define CRNN_BN_INFER_BATCH_MAX 16
int crnn_bn_infer_state_batch(int n, const float* const* mmean, const float* const* mvar, const float* const* gamma, const float* const* beta,
                              const int* C, float* const* bnstate, crnn_stream_t stream);
int crnn_bn_act(const float* x, const float* bnstate, float* y, long M, int C, crnn_stream_t stream);
/* y = Dropout(MaxPool(ReLU6(BN(x))))  (utils.py:45-56) */
int crnn_bn_act_pool_drop(const float* x, const float* bnstate, float* y, int B, int H, int W, int C, int ph, int pw,
                          float rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
int crnn_bn_bwd_chunks(long M);
int crnn_bn_bwd(const float* x, const float* g, const float* bnstate, const float* gamma, float* dx, float* dgamma,
                float* dbeta, float* scratch_partials, float* coef, int B, int H, int W, int C, int ph, int pw,
                float rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
int crnn_add(const float* a, const float* b, float* o, long n, crnn_stream_t stream);
int crnn_dropout(const float* x, float* y, long rows, int C, int ldx, int ldy, float rate, uint64_t seed,
                 uint32_t layer, crnn_stream_t stream);
/* crnn_dropout that also writes the keep bytes of the site (the table of crnn_dropout_keep_bytes below: one byte per 8 consecutive elements of the compact
 * [rows][C] index space) -- round 5: dense2's one-pass backward reads them (crnn_dense_bwd_small).  C % 8 == 0, ldx / ldy % 4 == 0, 16-byte aligned x / y;
 * else CRNN_ERR_UNSUPPORTED.  keep: rows * C / 8 bytes. */
int crnn_dropout_keep(const float* x, float* y, void* keep, long rows, int C, int ldx, int ldy, float rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
int crnn_dropout_mask(float* m, long n, float rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
/* keep bits of the same dropout site, one byte per group of 8 consecutive elements (bit e: element 8 g + e is kept; rate <= 0: 0xFF) -- the form the
 * prologue row-stream depthwise kernels read (crnn_dwconv3x3_fwd_stream_pro / _bwd_stream_pro).  out: 4-byte aligned, (ngroups + 3) / 4 * 4 bytes written */
int crnn_dropout_keep_bytes(void* out, long ngroups, float rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
/* the same for n <= CRNN_KEEP_BATCH_MAX dropout sites of one step in one launch (host arrays of n entries) */
#define CRNN_KEEP_BATCH_MAX 8
int crnn_dropout_keep_bytes_batch(int n, void* const* out, const long* ngroups, const uint32_t* layer, float rate, uint64_t seed, crnn_stream_t stream);
int crnn_relu_bwd(const float* y, const float* g, float* go, long rows, int C, float scale, int permP,
                  crnn_stream_t stream);
/* the same with a bf16 copy of the result (go_bf16, may be NULL; round to nearest even) -- round 5: dense1's data gradient runs on the weights-resident GEMM */
int crnn_relu_bwd_ex(const float* y, const float* g, float* go, void* go_bf16, long rows, int C, float scale, int permP, crnn_stream_t stream);
/* spatial transformer pieces (utils.py:116-258) */
int crnn_maxpool_fwd(const float* x, float* y, int B, int H, int W, int C, int ph, int pw, crnn_stream_t stream);
int crnn_maxpool_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int ph, int pw,
                     crnn_stream_t stream);
int crnn_im2col(const float* x, float* col, int B, int H, int W, int C, int K, crnn_stream_t stream);
int crnn_col2im(const float* dcol, float* dx, int B, int H, int W, int C, int K, crnn_stream_t stream);
int crnn_sampler_fwd(const float* img, const float* theta, float* out, int B, int H, int W, int pad,
                     crnn_stream_t stream);
int crnn_sampler_bwd(const float* img, const float* theta, const float* gout, float* dtheta, int B, int H, int W,
                     int pad, crnn_stream_t stream);
int crnn_pad_copy(const float* img, float* out, int B, int H, int W, int pad, crnn_stream_t stream);
/* Localisation net of the spatial transformer (utils.py:248-256) as direct kernels -- weights in LDS, no im2col:
 * 5x5 'valid' convolution with 20 filters, NHWC fp32, Cin = 1 (first conv) or 20 (second), kernel [5][5][Cin][20];
 * forward (+bias, linear), weight/bias gradient (scratch: crnn_loc_conv_wgrad_chunks(B,H,W) * (25*Cin*20 + 20) floats, plus one
 * more row when db != dk + 25*Cin*20), data gradient (Cin = 20 only; the first conv's input needs none). */
int crnn_loc_conv_fwd(const float* x, const float* k, const float* bias, float* y, int B, int H, int W, int Cin, crnn_stream_t stream);
int crnn_loc_conv_wgrad_chunks(int B, int H, int W);
int crnn_loc_conv_wgrad(const float* x, const float* gy, float* dk, float* db, float* scratch, int B, int H, int W, int Cin,
                        crnn_stream_t stream);
int crnn_loc_conv_dgrad(const float* gy, const float* k, float* dx, int B, int H, int W, crnn_stream_t stream);
/* its two dense layers fused: fc1 = relu(flat W1 + b1) [F -> 50], theta = fc1 W2 + b2 [50 -> 6]; and their backward:
 * dfc1 / dflat per image, dW1 [F][50], db1, dW2 [50][6], db2 summed over the batch in image order (deterministic) */
int crnn_loc_fc_fwd(const float* flat, const float* w1, const float* b1, const float* w2, const float* b2, float* fc1, float* theta,
                    int B, int F, crnn_stream_t stream);
int crnn_loc_fc_bwd(const float* flat, const float* fc1, const float* dtheta, const float* w1, const float* w2, float* dfc1,
                    float* dflat, float* dw1, float* db1, float* dw2, float* db2, int B, int F, crnn_stream_t stream);
/* Round 4: the localisation net of a sample in ONE workgroup (utils.py:248-256).  crnn_loc_net_fwd = crnn_maxpool_fwd + crnn_loc_conv_fwd + crnn_maxpool_fwd +
 * crnn_loc_conv_fwd + crnn_loc_fc_fwd in one launch, every output (pool1, c1, pool2, flat, fc1, theta) bit-identical; crnn_loc_net_bwd = the whole backward from
 * dtheta (crnn_loc_fc_bwd + crnn_loc_conv_wgrad + crnn_loc_conv_dgrad + crnn_maxpool_bwd + crnn_loc_conv_wgrad) in two launches: one workgroup per sample for the data
 * path and the sample's convolution weight-gradient terms (scratch: crnn_loc_net_bwd_scratch(B) floats), then one launch summing the terms in sample order and forming
 * the dense layers' weight gradients -- the same sums in another (fixed) order.  _supported: whole pooling windows over the first convolution's map (even Ho1, Wo1)
 * and a sample's maps within LDS (images up to about 250 x 32); else CRNN_ERR_UNSUPPORTED and the caller runs the stand-alone kernels. */
int crnn_loc_net_fused_supported(int H0, int W0);
int crnn_loc_net_fwd(const float* x, const float* k1, const float* bc1, const float* k2, const float* bc2, const float* w1, const float* b1, const float* w2,
                     const float* b2, float* pool1, float* c1, float* pool2, float* flat, float* fc1, float* theta, int B, int H0, int W0, crnn_stream_t stream);
long crnn_loc_net_bwd_scratch(int B);
int crnn_loc_net_bwd(const float* dtheta, const float* flat, const float* fc1, const float* pool1, const float* c1, const float* pool2, const float* w1,
                     const float* w2, const float* k2, float* dfc1, float* scratch, float* dk1, float* dbc1, float* dk2, float* dbc2, float* dw1, float* db1,
                     float* dw2, float* db2, int B, int H0, int W0, crnn_stream_t stream);
/* Bidirectional LSTM recurrence (utils.py:78-79), time-major */
int crnn_lstm_fwd(const float* xw0, const float* xw1, const float* ut0, const float* ut1, float* h0, float* h1, int ldh,
                  float* c0, float* c1, float* g0, float* g1, int T, int B, int u, crnn_stream_t stream);
int crnn_lstm_bwd(const float* u0, const float* u1, const float* c0, const float* c1, const float* g0, const float* g1,
                  const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* dc0, float* dc1, int T,
                  int B, int u, crnn_stream_t stream);
/* same, dt_u = storage of the recurrent weights ut* / u* (0 = fp32; 1 = bf16: the per-step products then run on
 * v_mfma_f32_16x16x32_bf16 with the fp32 state rounded to bf16 as it is packed, fp32 accumulation; needs u % 128 == 0) */
int crnn_lstm_fwd_ex(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1, int ldh,
                     float* c0, float* c1, float* g0, float* g1, int T, int B, int u, int dt_u, crnn_stream_t stream);
int crnn_lstm_bwd_ex(const void* u0, const void* u1, const float* c0, const float* c1, const float* g0, const float* g1,
                     const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* dc0, float* dc1, int T,
                     int B, int u, int dt_u, crnn_stream_t stream);
/* Persistent LDS-DMA GEMM for the pointwise-convolution products (utils.py:49): Y[M][N] = X[M][K] . W[N][K]^T, all three bf16
 * (row strides K, K, N), fp32 accumulate on v_mfma_f32_32x32x16_bf16.  One persistent workgroup per CU: two loader waves stream
 * 64-k chunks of pixel rows (4-slot LDS ring) and weight rows (2 slots) with global_load_lds, four MFMA waves consume them and
 * store 16 bytes per lane straight from the accumulators (weights as the MFMA A operand, v_permlane32_swap).  Requires
 * K % 64 == 0, N % 128 == 0, 16-byte aligned pointers; else -3 (use crnn_gemm_bf16_ex). */
int crnn_gemm_nt_bf16(const void* X, const void* W, void* Y, int M, int N, int K, crnn_stream_t stream);
/* The same product with the weights RESIDENT IN REGISTERS (gemm_wres.hip): a workgroup owns 128 output channels, each of its four MFMA
 * waves holds 32 channels x K of W as MFMA operand fragments for the whole launch, only the pixel rows stream (LDS-DMA ring of 8 x 16 KiB
 * per CU), the N/128 channel slices of a pixel stripe run on one XCD so HBM sees the stripe once.  Bit-identical to crnn_gemm_nt_bf16.
 * Supported (else -3): N % 128 == 0, N <= 1024 (K <= 128: N <= 8192 -- round 5, dense1's data gradient: more slices than an XCD has CUs, one workgroup per
 * slice and stripe lane), K in {64, 128, 256, 512}, 16-byte aligned pointers. */
int crnn_gemm_wres_supported(int N, int K);
int crnn_gemm_wres_bf16(const void* X, const void* W, void* Y, int M, int N, int K, crnn_stream_t stream);
/* The same product as the data gradient da = dq . W^T of a depthwise-separable block (utils.py:45-49 backwards), with the statistics pass of
 * the backward of the BatchNorm in front of the pointwise convolution taken by the kernel's storer waves from the staged result and the
 * matching rows of d (the BatchNorm's input, bf16 [M][N]; bnstate = [mean | var | scale | shift] x N): stat_partials
 * [crnn_gemm_wres_bnstats_rows(M, N, K)][2][N] = partial sums of gy and gy * xhat, gy = da where 0 < d * scale + shift < 6.  Saves the
 * stand-alone pass's second read of da and d.  Y bit-identical to crnn_gemm_wres_bf16; the statistics are crnn_bn_bwd_ex's sums in
 * another order.  Shapes: M % 128 == 0, K in {256, 512}, N % 128 == 0 (-3 otherwise).  crnn_bn_bwd_finalize turns the partials into
 * dgamma, dbeta and coef = [mean(gy) | mean(gy * xhat)] (count = M). */
int crnn_gemm_wres_bnstats_supported(long M, int N, int K);
int crnn_gemm_wres_bnstats_rows(long M, int N, int K);
int crnn_gemm_wres_bf16_bnstats(const void* X, const void* W, void* Y, long M, int N, int K, const void* d, const float* bnstate,
                                float* stat_partials, crnn_stream_t stream);
int crnn_bn_bwd_finalize(const float* partials, int nparts, int C, long count, float* dgamma, float* dbeta, float* coef, crnn_stream_t stream);
/* ... for long partial lists (folded into 32 chunk rows first; scratch: 32 * 2 * C floats, may be NULL for nparts <= 1024) */
int crnn_bn_bwd_finalize_folded(const float* partials, int nparts, int C, long count, float* dgamma, float* dbeta, float* coef, float* scratch,
                                crnn_stream_t stream);
/* crnn_bn_bwd_ex / crnn_bn_bwd_apply_ex for fp32 tensors with dx written as bf16 PLANES (planes = 2 | 3; plane pl of dx[i] at dx_planes[pl * plane_stride + i]:
 * the words crnn_split3_planes forms from the fp32 dx those entry points write) -- the operand format of crnn_gemm_pres_bnstats and
 * crnn_pwconv_bnrelu6_wgrad_planes_stream_gp, split once where dx is produced instead of once per tile where it is read.  C % 4 == 0, plane_stride % 4 == 0. */
int crnn_bn_bwd_planes_ex(const float* x, const float* g, const float* bnstate, const float* gamma, void* dx_planes, long plane_stride, int planes,
                          float* dgamma, float* dbeta, float* scratch_partials, float* coef, int B, int H, int W, int C, int ph, int pw, float rate,
                          uint64_t seed, uint32_t layer, crnn_stream_t stream);
int crnn_bn_bwd_apply_planes_ex(const float* x, const float* g, const float* bnstate, const float* coef, void* dx_planes, long plane_stride, int planes,
                                int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
/* The pooled blocks' shortcut (round 6): crnn_bn_act_pool_drop_qmax_ex = crnn_bn_act_pool_drop_ex that also writes qmax [B][H/ph][W/pw][C] (storage dt_in), x at the
 * FIRST maximum of x * scale + shift over each 2 x 2 / 1 x 2 pool window; crnn_bn_bwd_qmax_ex = crnn_bn_bwd_ex / crnn_bn_bwd_planes_ex (dx as a tensor of `dtype`, or --
 * fp32, dx NULL -- as bf16 planes) whose statistics pass reads that one value per window instead of the window: the arg-max carries all of the window's gradient,
 * so the sums are the same, bit for bit, from a quarter (half) of the bytes.  qmax NULL: the entry points above.  -3 for other windows. */
int crnn_bn_act_pool_drop_qmax_ex(const void* x, const float* bnstate, void* y, void* qmax, int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed,
                                  uint32_t layer, int dt_in, int dt_out, crnn_stream_t stream);
int crnn_bn_bwd_qmax_ex(const void* x, const void* qmax, const void* g, const float* bnstate, const float* gamma, void* dx, void* dx_planes, long plane_stride, int planes,
                        float* dgamma, float* dbeta, float* scratch_partials, float* coef, int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed,
                        uint32_t layer, int dtype, crnn_stream_t stream);
/* Pass 2 of crnn_bn_bwd_ex alone: dx from coef = [mean(gy) | mean(gy * xhat)] of a statistics pass that ran elsewhere (crnn_bn_bwd_finalize*). */
int crnn_bn_bwd_apply_ex(const void* x, const void* g, const float* bnstate, const float* coef, void* dx, int B, int H, int W, int C, int ph, int pw,
                         float rate, uint64_t seed, uint32_t layer, int dtype, crnn_stream_t stream);
/* Parity mode: the data gradient da [M][N] = dq [M][K] . W [N][K]^T of a block's pointwise conv as three-plane products (crnn_gemm_f32x3 mode 1, bit for
 * bit) whose epilogue also takes the statistics pass of the BatchNorm in front of the conv: stat_partials [M / 128][2][N] = per-tile column sums of gy and
 * gy * xhat, gy = da where 0 < d * scale + shift < 6 (d [M][N] fp32 = the BatchNorm's input, bnstate = [mean|var|scale|shift] x N).  Whole tiles only
 * (M % 128 == 0, N = 64 or a multiple of 128, K % 64 == 0, 16-byte aligned pointers); -3 otherwise (run crnn_bn_bwd_ex). */
int crnn_gemm_f32x3_bnstats_supported(long M, int N, int K);
int crnn_gemm_f32x3_bnstats_rows(long M);
int crnn_gemm_f32x3_bnstats(const float* dq, const float* W, float* da, long M, int N, int K, const float* d, const float* bnstate, float* stat_partials,
                            crnn_stream_t stream);
/* Two-plane forms (f32x2) of crnn_gemm_f32x3_bnstats, crnn_pwconv_bnrelu6_fwd_f32x3 and crnn_pwconv_bnrelu6_wgrad_f32x3: every operand split into TWO bf16
 * planes (hi = bf16(x), mid = bf16(x - hi): 16 significant bits) and the three products hi*hi + hi*mid + mid*hi accumulated in fp32 -- relative error of
 * a product <= 3 * 2^-18 (fp32: 2^-24, TF32: 2^-11), half the MFMA work and two thirds of the LDS traffic of the three-plane forms.  Same arguments, shapes
 * and return codes.  The parity-mode step uses them for its BACKWARD GEMMs (CRNN_FLAG_THREE_PLANE_BACKWARD: three planes there too);
 * its forward keeps three planes. */
int crnn_gemm_f32x2(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* bias, int act,
                    int accumulate, int permP, float* scratch, size_t scratch_bytes, crnn_stream_t stream);   /* crnn_gemm_f32x3's contract */
int crnn_gemm_f32x2_bnstats(const float* dq, const float* W, float* da, long M, int N, int K, const float* d, const float* bnstate, float* stat_partials,
                            crnn_stream_t stream);
int crnn_pwconv_bnrelu6_fwd_f32x2(const float* d, const float* in_bnstate, const float* w, float* q, long M, int N, int K, float* stat_partials,
                                  crnn_stream_t stream);
int crnn_pwconv_bnrelu6_wgrad_f32x2(const float* d, const float* in_bnstate, const float* g, float* dw, long M, int N, int K, float* scratch,
                                    size_t scratch_bytes, crnn_stream_t stream);
/* The three bf16 planes of n fp32 values, formed once instead of by every GEMM tile that stages them: plane pl of x[i] at planes[pl * plane_stride + i]
 * (bf16 words; n % 4 == 0, plane_stride % 4 == 0 and >= n, x 16-byte and planes 8-byte aligned) -- the very words the three-plane kernel's staging waves
 * form, so the *_pl entry points below return what their fp32-operand forms return, bit for bit.  With CRNN_FLAG_WEIGHT_PLANES the parity-mode step splits the
 * pointwise-conv weights this way at the start of the forward pass. */
int crnn_split3_planes(const float* x, void* planes, long n, long plane_stride, crnn_stream_t stream);
/* crnn_gemm_f32x3_bnstats with operands given as planes (null planes: split from the fp32 operand while staging).  W_planes: planes of W [N][K];
 * dq_planes: planes of dq [M][K] (only together with W_planes).  The fp32 pointers stay required (alignment rules, fallback). */
int crnn_gemm_f32x3_bnstats_pl(const float* dq, const void* dq_planes, long dq_plane_stride, const float* W, const void* W_planes, long W_plane_stride,
                               float* da, long M, int N, int K, const float* d, const float* bnstate, float* stat_partials, crnn_stream_t stream);
/* Inference forward of a pointwise convolution on the same kernel with the BatchNorm + ReLU6 that follows folded into the MFMA waves'
 * epilogue: y[M][N] (bf16) = ReLU6((a . wT^T) * scale[n] + shift[n]), out_bnstate = [mean|var|scale|shift] (crnn_bn_infer_state).
 * Bit-identical to crnn_pwconv_fwd(..., out_bnstate, ...) on bf16 tensors.  Same shape rules as crnn_gemm_wres_bf16. */
int crnn_pwconv_fwd_wres_folded(const void* a, const void* wT, void* y, long M, int N, int K, const float* out_bnstate, crnn_stream_t stream);
/* ... and with the MaxPooling2D after the block's ReLU6 (utils.py:52-54) in the epilogue: y [M / pool_rows][N] = max over every group of pool_rows
 * consecutive rows of ReLU6((a . wT^T) * scale + shift); the un-pooled map never reaches HBM.  pool_rows = 2: MaxPooling2D((1,2)) on an NHWC map of
 * even width; pool_rows = 4: MaxPooling2D((2,2)) when the rows of `a` are in 2x2-window-major order (crnn_dwconv3x3_fwd_stream_ex, out_order 1).
 * Equal to max-pooling crnn_pwconv_fwd_wres_folded's output.  (K, pool_rows) = (128, 4) | (256, 2), M % pool_rows == 0; -3 otherwise. */
int crnn_pwconv_fwd_wres_folded_pool_supported(long M, int N, int K, int pool_rows);
int crnn_pwconv_fwd_wres_folded_pool(const void* a, const void* wT, void* y, long M, int N, int K, const float* out_bnstate, int pool_rows,
                                     crnn_stream_t stream);
/* The forward pointwise convolution of a training block on the same weights-resident core, fed by the PRE-BatchNorm depthwise output:
 * q[M][N] (bf16) = ReLU6(BN(d))[M][K] . wT[N][K]^T, in_bnstate = [mean|var|scale|shift] of that BatchNorm.  Four IO waves per workgroup
 * load the pixel stages into registers three stages ahead, apply the BatchNorm + ReLU6 (bit for bit the arithmetic of
 * crnn_pwconv_bnrelu6_fwd) on the way into the LDS ring, drain the finished bf16 stripes and accumulate their column sums / sums of
 * squares over the whole launch: stat_partials (may be NULL) = [crnn_pwconv_fwd_wres_rows(M, N, K)][2][N], every element written.
 * q is bit-identical to crnn_pwconv_bnrelu6_fwd(w_transposed = 1, bf16 q); the statistics are the same sums in a different order.
 * Supported (else -3): M % 128 == 0, N % 128 == 0, N <= 1024, K in {64, 128, 256, 512}. */
/* The weight gradient of the same convolution as a pixel stream (gemm_wgrad.hip): dw[K][N] (fp32) = ReLU6(BN(d))^T . g, d and g bf16.
 * A workgroup keeps one 128 x 128 output tile in its MFMA waves' registers over a contiguous range of 64-pixel chunks; its IO waves
 * load the d / g rows three chunks ahead, apply the BatchNorm + ReLU6 (the arithmetic of crnn_pwconv_bnrelu6_wgrad bit for bit) and
 * write both operands k-major into an LDS ring; the tiles of a range share an XCD; partial tiles [ranges][K][N] go to scratch
 * (crnn_pwconv_wgrad_stream_scratch_bytes) and a fixed-order second stage sums them.  Deterministic; agrees with
 * crnn_pwconv_bnrelu6_wgrad to fp32 summation round-off (other range boundaries).  Supported (else -3): M % 64 == 0, K % 128 == 0,
 * N % 128 == 0, K, N <= 1024. */
int crnn_pwconv_wgrad_stream_supported(long M, int N, int K);
size_t crnn_pwconv_wgrad_stream_scratch_bytes(long M, int N, int K);
int crnn_pwconv_bnrelu6_wgrad_stream(const void* d, const float* in_bnstate, const void* g, float* dw, long M, int N, int K, float* scratch,
                                     size_t scratch_bytes, crnn_stream_t stream);
/* The same stream for fp32 operands (rounded to bf16 on the way in, as crnn_gemm_bf16_ex mode 2 does): C[M][N] (fp32, row stride ldc) =
 * A^T . B with A [K][lda >= M], B [K][ldb >= N] fp32 and the reduction over the K rows -- the weight gradients of the recurrent layers
 * (dW = X^T dZ, dU = H^T dZ over T*B rows).  Supported (else -3): M, N multiples of 128 up to 1024, K % 64 == 0, leading dimensions
 * multiples of 4, 16-byte aligned pointers; scratch: crnn_pwconv_wgrad_stream_scratch_bytes(K, N, M). */
int crnn_gemm_tn_stream(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                        size_t scratch_bytes, crnn_stream_t stream);
/* The same stream for bf16 operands without a transform (round 5): C[M][N] (fp32, row stride ldc) = A^T . B with A [K][lda >= M], B [K][ldb >= N] bf16
 * and the reduction over the K rows -- dense1's weight gradient dW1 [feat][tds] = x7^T . gbm (utils.py:73-74 backwards; M = feat = 4608 is 36 tiles of
 * 128 features: more tiles than an XCD has CUs, so the rows split into cus / 36 ranges).  Supported (else -3): M % 128 == 0 up to 8192, N % 128 == 0 up
 * to 1024, K % 64 == 0, lda / ldb multiples of 8, ldc of 4, 16-byte aligned pointers. */
int crnn_gemm_tn_bf16_stream_supported(int M, int N, long K);
size_t crnn_gemm_tn_bf16_stream_scratch_bytes(int M, int N, long K);
int crnn_gemm_tn_bf16_stream(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                             size_t scratch_bytes, crnn_stream_t stream);
/* dense1's forward (round 5; utils.py:72-75): Y[perm(m)][N] (fp32, row stride N) = Dropout(ReLU(X[M][K] . WT[N][K]^T + bias)) -- X bf16 (row stride lda), WT the
 * bf16 W^T copy (row stride ldw), one workgroup per 64-row stripe over the whole reduction (the recurrent layers' stripe stream with a bf16 operand);
 * relu 0 | 1; permP: rows written at (m % permP) * (M / permP) + m / permP (0: in place; batch-major rows to time-major as crnn_gemm_f32's permP);
 * drop_rate > 0: the multipliers crnn_dropout applies to the compact [M][N] output for (seed, layer).  Supported (else -3): M % 64 == 0, N = 128 | 256,
 * K % 64 == 0, lda / ldw multiples of 8, 16-byte aligned pointers.
 * Summation order (round 6): with drop_rate == 0 (inference) every 64-row stripe sums its 64-k chunks in ascending order -- a row's result does not depend on
 * its position in the batch or on the batch size.  With drop_rate > 0 (training) stripe i starts its walk at chunk (3 i) mod (K / 64) (memory-channel skew):
 * results differ between stripes at fp32 round-off level and are deterministic run to run only.  crnn_gemm_nt_f32_stream (the recurrent layers' input
 * gradients, training only) always rotates. */
int crnn_dense_fwd_stream_supported(long M, int N, long K);
int crnn_dense_fwd_stream(const void* X, const void* WT, const float* bias, float* Y, long M, int N, long K, int lda, int ldw, int relu, int permP,
                          float drop_rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
/* Deferred second stages.  The two streaming weight-gradient entries above are stage 1 (partial tiles into `scratch`) + stage 2 (a fixed-order
 * sum into the gradient, ~5 us of dependent launch each, 13 per train step).  The *_defer forms run stage 1 only and describe stage 2 in
 * *job; crnn_wgrad_sum_batch runs up to CRNN_SUM_BATCH_MAX of them in ONE launch -- the same sums in the same order, bit-identical
 * gradients.  Each deferred call needs its own scratch until the batch has run (crnn_pwconv_wgrad_stream_scratch_bytes). */
#define CRNN_SUM_BATCH_MAX 16
typedef struct { const float* partials; float* out; long total; int nsplit, N, ldc; } crnn_sum_job;
int crnn_pwconv_bnrelu6_wgrad_stream_defer(const void* d, const float* in_bnstate, const void* g, float* dw, long M, int N, int K,
                                           float* scratch, size_t scratch_bytes, crnn_sum_job* job, crnn_stream_t stream);
int crnn_gemm_tn_stream_defer(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                              size_t scratch_bytes, crnn_sum_job* job, crnn_stream_t stream);
int crnn_wgrad_sum_batch(const crnn_sum_job* jobs, int n, crnn_stream_t stream);
/* Input gradient of a Bidirectional layer's input projections in one streaming launch: Y[M][N] (fp32, row stride ldy) = A0[M][K] . W0[N][K]^T
 * (+ A1 . W1^T when A1 != NULL), A fp32 (rounded to bf16 on the way in), W bf16; one workgroup per 64-row stripe keeps its result in the
 * MFMA waves' registers over the whole reduction.  Supported (else -3): M % 64 == 0, N in {128, 256}, K % 64 == 0, leading dimensions
 * multiples of 8, 16-byte aligned pointers. */
int crnn_gemm_nt_f32_stream(const float* A0, const void* W0, const float* A1, const void* W1, float* Y, int M, int N, int K, int lda, int ldw,
                            int ldy, crnn_stream_t stream);
/* The same kernel over column slabs of a wider result, with an optional bias added to the finished sums: Y[M][N] = A0 . W0[N][K]^T (+ A1 . W1^T)
 * (+ bias[N]); N % 128 == 0.  The recurrent layers' input projections xw = X W + b from the bf16 W^T copies. */
int crnn_gemm_nt_f32_stream_bias(const float* A0, const void* W0, const float* A1, const void* W1, float* Y, const float* bias, int M, int N, int K,
                                 int lda, int ldw, int ldy, crnn_stream_t stream);
/* Both directions' input projections of a Bidirectional recurrent layer in ONE launch (reference utils.py:77-82, x W + b hoisted out of the recurrence;
 * gemm_wgrad.hip, round 5): Yf = X . Wf^T + bias_f, Yb = X . Wb^T + bias_b with fp32 X [M][K] (row stride lda), bf16 W [N][K] (row stride ldw: the W^T copies the
 * forward keeps), fp32 Y [M][N] (row stride ldy).  Persistent workgroups keep a 256-column weight slab in LDS and walk the 64-row stripes.  Bit-identical to two
 * crnn_gemm_nt_f32_stream_bias calls.  _supported: M % 64 == 0, N % 256 == 0, K in {64, 128, 192, 256}; else CRNN_ERR_UNSUPPORTED. */
int crnn_rnn_input_proj_supported(int M, int N, int K);
int crnn_rnn_input_proj(const float* X, const void* Wf, const void* Wb, const float* bias_f, const float* bias_b, float* Yf, float* Yb, int M, int N, int K,
                        int lda, int ldw, int ldy, crnn_stream_t stream);
int crnn_pwconv_fwd_wres_supported(long M, int N, int K);
int crnn_pwconv_fwd_wres_rows(long M, int N, int K);
int crnn_pwconv_bnrelu6_fwd_wres(const void* d, const float* in_bnstate, const void* wT, void* q, long M, int N, int K, float* stat_partials,
                                 crnn_stream_t stream);

/* Persistent recurrences: ONE launch per Bidirectional(LSTM) layer (utils.py:77-82) instead of T dependent step launches.
 * A cluster of u/16 workgroups runs the chain of one 16- or 32-row batch tile of one direction; each workgroup keeps its
 * 256x64 slice of the recurrent weights in registers (MFMA B fragments) and the cell state / cell-gradient carry in
 * registers for all T steps; per step the cluster all-gathers h_t (forward) / dz_t (backward) through `xbuf` with
 * write-through stores and L1-bypassing polled loads (the data is its own ready flag) and stages it through LDS as the next
 * step's MFMA A operand.  Bit-identical to crnn_lstm_*_ex.  `xbuf`: caller-owned scratch of crnn_lstm_persist_xbuf_bytes()
 * bytes, 16-byte aligned.  Status: the unsigned at byte 16 of xbuf is 0xFFFFFFFF after a clean kernel launch, anything else means a
 * bounded wait gave up (the cluster was not co-resident: results invalid); the unsigned at byte 0 is a STICKY counter of give-ups
 * that no launch resets -- the caller zeroes it once after allocating xbuf and compares it with the last value it saw (the engine
 * does that wherever it synchronises with the host anyway and raises; inside the workspace this is the tensor "rnnx").
 * mt = batch rows per workgroup / 16 (1 | 2), uw = 16-unit groups per workgroup (1 | 2 | 4: 256 / 512 / 1024 threads, the
 * cluster has u/(16 uw) members); 0 = automatic; uw | CRNN_RNN_XCD_LOCAL: the members of a cluster are the workgroup ids congruent
 * modulo 8 (observed: one XCD) instead of consecutive ids; each cluster then checks HW_REG_XCC_ID of all its members once per launch and,
 * only if they agree, exchanges with plain L2-resident stores instead of write-through ones -- same results for any placement.  crnn_lstm_persist_supported: 0 if (u, dt_u) has a kernel
 * (fp32: u in {64,128,256}; bf16: u in {128,256,512}), else -3 -- use the step kernels then. */
#define CRNN_RNN_XCD_LOCAL 0x100
#define CRNN_RNN_DEBUG_DROP_MEMBER 0x200   /* crnn_lstm_fwd_persist only, tests: the last workgroup of the grid is not launched, so its cluster loses a
                                              member for good -- the others wait their 2 s, give up and report it (status word, sticky counter) */
size_t crnn_lstm_persist_xbuf_bytes(int T, int B, int u, int dt_u);
int crnn_lstm_persist_supported(int u, int dt_u);
/* zero the sticky give-up counter at the head of an exchange buffer: once after allocation, for callers that do not zero-fill it */
int crnn_rnn_status_reset(void* xbuf, crnn_stream_t stream);
int crnn_lstm_fwd_persist(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1, int ldh,
                          float* c0, float* c1, float* g0, float* g1, int T, int B, int u, int dt_u, void* xbuf, size_t xbuf_bytes,
                          int mt, int uw, crnn_stream_t stream);
int crnn_lstm_bwd_persist(const void* u0, const void* u1, const float* c0, const float* c1, const float* g0, const float* g1,
                          const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, int T, int B, int u, int dt_u,
                          void* xbuf, size_t xbuf_bytes, int mt, int uw, crnn_stream_t stream);
/* crnn_lstm_bwd_persist that also leaves the layer's bias-gradient partials (round 4): db_partials0 / 1 [crnn_rnn_db_rows(B)][4u] = the column sums of dz0 / dz1 over time
 * for every 16-row batch tile -- finish with crnn_partials_sum over the rows (a fixed order); the stand-alone column reduction reads dz once more for the same sums */
int crnn_rnn_db_rows(int B);
int crnn_lstm_bwd_persist_db(const void* u0, const void* u1, const float* c0, const float* c1, const float* g0, const float* g1, const float* dout0,
                             const float* dout1, int ldo, float* dz0, float* dz1, float* db_partials0, float* db_partials1, int T, int B, int u, int dt_u,
                             void* xbuf, size_t xbuf_bytes, int mt_req, int uw_req, crnn_stream_t stream);
/* Persistent Bidirectional(GRU) recurrences (utils.py:80-82, the cell train.py:119 really builds): ONE launch per layer and pass instead of
 * 2 T step launches, the cluster / sentinel-ring design of crnn_lstm_*_persist with two all-gathers per step (h_{t-1}, then r * h_{t-1}:
 * the candidate's recurrent product needs r of every unit; backward: [dz|dr]_{t+1}, then dhh_t).  Bit-identical to crnn_gru_*_ex.
 * xbuf: crnn_lstm_persist_xbuf_bytes(T, B, u, dt_u) bytes, same status words; flags: 0 or CRNN_RNN_XCD_LOCAL.
 * crnn_gru_persist_supported: 0 if (u, dt_u) has a kernel (fp32: u in {64,128,256}; bf16: u in {128,256,512}), else -3. */
int crnn_gru_persist_supported(int u, int dt_u);
int crnn_gru_fwd_persist(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1, int ldh,
                         float* g0, float* g1, float* rh0, float* rh1, int T, int B, int u, int dt_u, void* xbuf, size_t xbuf_bytes,
                         int flags, crnn_stream_t stream);
int crnn_gru_bwd_persist(const void* u0, const void* u1, const float* h0, const float* h1, int ldh, const float* g0, const float* g1,
                         const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, int T, int B, int u, int dt_u,
                         void* xbuf, size_t xbuf_bytes, int flags, crnn_stream_t stream);
/* ... and with the layer's bias-gradient partials, db_partials0 / 1 [crnn_rnn_db_rows(B)][3u] (crnn_lstm_bwd_persist_db's contract) */
int crnn_gru_bwd_persist_db(const void* u0, const void* u1, const float* h0, const float* h1, int ldh, const float* g0, const float* g1, const float* dout0,
                            const float* dout1, int ldo, float* dz0, float* dz1, float* db_partials0, float* db_partials1, int T, int B, int u, int dt_u,
                            void* xbuf, size_t xbuf_bytes, int flags, crnn_stream_t stream);
/* Bidirectional GRU recurrence (utils.py:81-82; reset_after=False), time-major; gates = z,r,hh; rh = r*h_prev */
int crnn_gru_fwd(const float* xw0, const float* xw1, const float* ut0, const float* ut1, float* h0, float* h1, int ldh,
                 float* g0, float* g1, float* rh0, float* rh1, int T, int B, int u, crnn_stream_t stream);
int crnn_gru_bwd(const float* u0, const float* u1, const float* h0, const float* h1, int ldh, const float* g0,
                 const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* dh0,
                 float* dh1, float* dhp0, float* dhp1, int T, int B, int u, crnn_stream_t stream);
/* dt_u as for crnn_lstm_*_ex (1 = bf16 recurrent weights, products on the bf16 MFMA; u % 128 == 0) */
int crnn_gru_fwd_ex(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1, int ldh,
                    float* g0, float* g1, float* rh0, float* rh1, int T, int B, int u, int dt_u, crnn_stream_t stream);
int crnn_gru_bwd_ex(const void* u0, const void* u1, const float* h0, const float* h1, int ldh, const float* g0,
                    const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* dh0,
                    float* dh1, float* dhp0, float* dhp1, int T, int B, int u, int dt_u, crnn_stream_t stream);
int crnn_transpose(const float* in, float* out, int R, int C, crnn_stream_t stream);
int crnn_transpose_ex(const float* in, void* out, int R, int C, int dt_out, crnn_stream_t stream);   /* dt_out 1: bf16 result */
/* softmax + CTC (utils.py:86, 98-103) */
int crnn_softmax_rows(const float* z, float* p, long rows, int C, crnn_stream_t stream);
/* dense2's epilogue in one pass (round 5): logits = z[:, :C] + bias (z [rows][ldz]: the raw products of a GEMM over a padded weight matrix), written in permuted
 * row order out_row = (m % permP) * (rows / permP) + m / permP (permP = 0: none; time-major rows back to batch-major as crnn_gemm_f32's permP), together with their row
 * softmax in p1 and, when p2 != NULL, in p2 as well (reference utils.py:85-86).  Same arithmetic as crnn_softmax_rows.  C <= 64. */
int crnn_softmax_rows_perm(const float* z, int ldz, const float* bias, float* logits, float* p1, float* p2, long rows, int C, int permP, crnn_stream_t stream);
/* dense2's backward in one pass (round 5; dense.hip; reference utils.py:82-86 under the CTC loss of utils.py:98-103): for x [M][K] (the layer's
 * dropped-out input, leading dimension ldx), dy [M][C] (compact) and W [K][C],
 *   dW[k][c] = sum_m x[m][k] dy[m][c],  db[c] = sum_m dy[m][c],  dx[m][k] = (sum_c dy[m][c] W[k][c]) * (dropout multiplier of element m * K + k of site `layer`)
 * in exact fp32, weights and weight-gradient accumulators in registers (thread = two input features), rows in ascending order per workgroup, the
 * per-workgroup partial gradients (scratch) summed in workgroup order.  db must be dW + K * C (the two gradients are one span of the gradient
 * buffer).  drop_rate 0: no dropout; keep: the site's keep bytes (crnn_dropout_keep / crnn_dropout_keep_bytes; M * K / 8 bytes, 4-byte aligned) or NULL
 * (the kernel's loader wave evaluates the decisions itself: same results, 1.6 us per 8-row step slower).  x rows contiguous (ldx == K).
 * _supported: K % 128 == 0, 128 <= K <= 512, C <= 40; else CRNN_ERR_UNSUPPORTED (the caller runs the GEMMs). */
int crnn_dense_bwd_small_supported(long M, int K, int C);
size_t crnn_dense_bwd_small_scratch_bytes(long M, int K, int C);
int crnn_dense_bwd_small(const float* x, const float* dy, const float* W, float* dx, float* dW, float* db, float* scratch, size_t scratch_bytes,
                         long M, int K, int C, int ldx, int lddx, const void* keep, float drop_rate, uint64_t seed, uint32_t layer, crnn_stream_t stream);
int crnn_ctc_loss_grad(const float* y, const int* labels, const int* input_len, const int* label_len, float* loss,
                       float* dlogits, int B, int T, int C, int Lmax, int skip, float grad_scale, crnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
