/* rsrgan.h -- C ABI of librsrgan_hip.so: RSRGAN's sequence-level GAN training step
 * (LSTMP generator + LSTMP discriminator, LSGAN losses, SGD(D)/Adam(G)) as
 * hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference (wangkenpu/rsrgan, Python 2.7 + TensorFlow 1.4) has no FFI; the
 * de-facto boundary is the object surface that
 *   scripts/train_gan_rnn_placeholder.py:train_one_iteration (:48-133),
 *   eval_one_iteration (:136-201) and decode (:204-302)
 * touch on models/gan_rnn_placeholder.py:GAN_RNN (:62-298).  Every entry point
 * below names the reference interface it replaces.  The reference-side binding
 * (a ctypes stub a maintainer would drop into models/gan_rnn_placeholder.py) is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch types.  Every `const float*` / `float*` data pointer is a
 *     DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()) and is
 *     only borrowed for the duration of the call's stream work, exactly like a
 *     TF feed_dict entry (gan_rnn_placeholder.py:94-104).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     work is enqueued asynchronously on it; nothing synchronises the device.
 *   - every function returns 0 on success or a negative rsrgan_status; nothing
 *     throws across the ABI; rsrgan_last_error() returns a thread-local string.
 *   - one handle per process/GPU, single caller thread
 *     (train_gan_rnn_placeholder.py:463-478: all sess.run calls come from the
 *     main thread).
 *   - all arithmetic is IEEE fp32 (tf.float32 placeholders, :94-104).
 */
#ifndef RSRGAN_H_
#define RSRGAN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rsrgan_status {
  RSRGAN_OK = 0,
  RSRGAN_ERR_INVALID = -1,     /* bad argument / unsupported configuration (ValueError in the reference, :131-132) */
  RSRGAN_ERR_HIP = -2,         /* a HIP runtime call failed */
  RSRGAN_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
  RSRGAN_ERR_STATE = -4        /* call sequence error (e.g. apply without backward) */
} rsrgan_status;

/* args.g_type (gan_rnn_placeholder.py:125-132) */
enum { RSRGAN_G_LSTM = 0, RSRGAN_G_RES_LSTM_L = 1, RSRGAN_G_RES_LSTM_BASE = 2,
       RSRGAN_G_DNN = 3, /* models/gan.py:109-110 + models/dnn.py: frame-level FC generator */
       RSRGAN_G_RCED = 4 /* models/rced.py: frame-level 9 x conv2d + FC generator (dnn_trainer.py:98-99), batch_norm=False */ };
/* self.discriminator (gan_rnn_placeholder.py:117; models/gan.py:104) */
enum { RSRGAN_D_LSTM = 0, RSRGAN_D_DNN = 1 /* models/discriminator_dnn.py */ };
/* which network a call addresses */
enum { RSRGAN_NET_G = 0, RSRGAN_NET_D = 1 };

/* Mutable scalars that the reference changes with sess.run(tf.assign(...))
 * between steps (train_gan_rnn_placeholder.py:63-64,460-461,531-533;
 * gan_rnn_placeholder.py:112-123). */
typedef enum rsrgan_scalar {
  RSRGAN_G_LEARNING_RATE = 0,
  RSRGAN_D_LEARNING_RATE = 1,
  RSRGAN_MSE_LAMBDA = 2,
  RSRGAN_D_REAL = 3,
  RSRGAN_D_FAKE = 4,
  RSRGAN_L2_SCALE = 5,
  RSRGAN_CLIP_NORM = 6,
  RSRGAN_ADAM_STEP = 7,       /* Adam's t (beta powers) of the generator, for checkpoint/resume */
  RSRGAN_ADAM_STEP_D = 8,     /* Adam's t of the discriminator (frame-level GAN only: models/gan.py:125) */
  RSRGAN_SCALAR_COUNT_
} rsrgan_scalar;

/* Construction arguments: the fields GAN_RNN.__init__ reads from `args`
 * (gan_rnn_placeholder.py:65-137) plus the layer sizes the reference
 * hard-codes (models/lstm.py:43-45, models/res_lstm_l.py:43-45,
 * models/discriminator_lstm.py:26-28) made runtime parameters. */
typedef struct rsrgan_cfg {
  int32_t batch_size;      /* per-GPU B (:96) */
  int32_t max_frames;      /* capacity for the padded time axis T */
  int32_t input_dim;       /* input_dim*(left_context+1+right_context) (:96-98) */
  int32_t output_dim;      /* 40 */
  int32_t g_type;          /* RSRGAN_G_* */
  int32_t g_layers;        /* 3 (lstm) / 4 (res_lstm_*) */
  int32_t g_cells;         /* 760 */
  int32_t g_proj;          /* 280 (lstm) / 257 (res_lstm_*) */
  int32_t d_type;          /* RSRGAN_D_LSTM */
  int32_t d_layers;        /* 2 */
  int32_t d_cells;         /* 256 */
  int32_t d_proj;          /* 40 */
  float   l2_scale;        /* args.l2_scale (:91) */
  float   clip_norm;       /* self.max_grad_norm = 15 (:71) */
  float   adam_beta1;      /* 0.9   (tf.train.AdamOptimizer defaults, :147) */
  float   adam_beta2;      /* 0.999 */
  float   adam_eps;        /* 1e-8  */
  float   ema_decay;       /* MOVING_AVERAGE_DECAY 0.9999 (:70); 0 disables the shadow copy */
  float   lrelu_alpha;     /* utils/ops.py:120 (0.3) */
  float   forget_bias;     /* LSTMCell(forget_bias=1.0) (models/lstm.py:94) */
  int32_t cross_validation;/* 1 = the cross_validation=True twin: no L2 term (:253) */
  int32_t flags;           /* RSRGAN_FLAG_* */
  /* frame-level GAN (models/gan.py:158-175): D sees concat(inputs[:, d_joint_off : +d_joint_dim], labels|G(x));
   * d_joint_dim = 0 feeds D the 40-dim target only, as gan_rnn_placeholder.py:207-208 does */
  int32_t d_joint_off;
  int32_t d_joint_dim;
  /* R-CED generator (models/rced.py:36-52): the fed frame is reshaped to [g_splice, input_dim / g_splice, 1]
   * (g_splice = left_context + 1 + right_context); 9 conv2d layers 12,16,20,24,32,24,20,16,12 x [g_splice, 13..7..13] */
  int32_t g_splice;
} rsrgan_cfg;

enum {
  RSRGAN_FLAG_WAVEFRONT = 1,   /* run the stacked LSTMs as one (layer,t) wavefront (default schedule when set) */
  RSRGAN_FLAG_GRAPH = 2,       /* replay the (static, per T) launch sequences of the wavefront schedule as hipGraphs: 1.6 us per
                                  dependent kernel on the GPU vs 3.1-4.6 us host-bound per eager launch (measured, round 2) */
  RSRGAN_FLAG_NO_SPLITK_B = 8, /* backward phase B as one launch of 32x16 tiles (round-1 first form) instead of split-K + reduce */
  RSRGAN_FLAG_SUPERVISED = 16, /* generator-only trainer (models/rnn_trainer.py:66-156, models/dnn_trainer.py:64-148):
                                  g_loss = mse_lambda*g_mse + g_l2, no discriminator pass; rsrgan_d_step is an error */
  RSRGAN_FLAG_OVERLAP = 4,     /* weight-gradient GEMMs on a side stream, chunked over time, concurrent with the backward
                                  wave (measured SLOWER on MI355X: 12.77 vs 12.20 ms/step; off by default) */
  RSRGAN_FLAG_BATCH_NORM = 32  /* args.batch_norm (run_gan_dnn.sh:134, run_dnn.sh:134): the hidden fully_connected layers of the
                                  frame-level generator (models/dnn.py:56-61) and of discriminator_dnn (:36-41) are
                                  relu(batch_norm(x.W, is_training = !cross_validation, scale=True, renorm=True)) without biases;
                                  the variable table gains <scope>/BatchNorm/{beta,gamma,moving_mean,moving_variance,renorm_mean,
                                  renorm_mean_weight,renorm_stddev,renorm_stddev_weight}.  Frame-level nets only. */
};

typedef struct rsrgan_handle_s* rsrgan_handle;

/* fills *cfg with the reference's hard-coded sizes for `g_type`. */
int rsrgan_default_cfg(int32_t g_type, rsrgan_cfg* cfg);

/* GAN_RNN(sess, args, devices, cross_validation, infer) (gan_rnn_placeholder.py:65-137).
 * Allocates parameters (xavier-uniform / zero biases from `seed`, as
 * xavier_initializer()/zeros_initializer(), models/lstm.py:86-87), optimizer
 * state and all activation stashes for (batch_size, max_frames). */
int rsrgan_create(const rsrgan_cfg* cfg, uint64_t seed, rsrgan_handle* out);
int rsrgan_destroy(rsrgan_handle h);
const char* rsrgan_last_error(void);

/* sess.run(tf.assign(model.<scalar>, v)) (train_gan_rnn_placeholder.py:63-64,460-461,531-533) */
int rsrgan_set_scalar(rsrgan_handle h, int32_t which, double v);
int rsrgan_get_scalar(rsrgan_handle h, int32_t which, double* v);

/* Variable table == tf.trainable_variables() split by the g_/d_ prefix
 * (gan_rnn_placeholder.py:301-317), in graph-construction order. */
int rsrgan_num_tensors(rsrgan_handle h, int32_t net);
int rsrgan_tensor_info(rsrgan_handle h, int32_t net, int32_t idx,
                       char* name, int32_t name_cap,
                       int32_t* rows, int32_t* cols, int64_t* dense_offset);
/* number of floats of the DENSE (TF-shaped, unpadded) flat parameter vector */
int64_t rsrgan_param_count(rsrgan_handle h, int32_t net);

/* tf.train.Saver save/restore payload (gan_rnn_placeholder.py:26-60) and parity
 * injection: dense flat vectors in variable-table order, DEVICE pointers.
 * `what`: 0 = variables, 1 = Adam m, 2 = Adam v (G only), 3 = EMA shadow. */
int rsrgan_get_params(rsrgan_handle h, int32_t net, int32_t what, float* dense, void* stream);
int rsrgan_set_params(rsrgan_handle h, int32_t net, int32_t what, const float* dense, void* stream);
/* last computed (tower-local or all-reduced) gradients, dense, for tests */
int rsrgan_get_grads(rsrgan_handle h, int32_t net, float* dense, void* stream);

/* sess.run(model.g_outputs, {inputs, lengths}) (train_gan_rnn_placeholder.py:282-285)
 *   x [B,T,Din] batch-major, lengths int32 [B], y [B,T,Dout]. */
int rsrgan_forward_g(rsrgan_handle h, const float* x, const int32_t* lengths, int32_t T,
                     float* y, void* stream);

/* sess.run([model.d_opt, model.d_rl_losses, model.d_fk_losses, model.d_losses], feed)
 * (train_gan_rnn_placeholder.py:77-82).  labels [B,T,Dout].  noise_real/noise_fake
 * are the two gaussian_noise_layer draws ([B,Dout], broadcast over T,
 * utils/ops.py:19-30) or NULL for disc_noise_std == 0.  out_losses: DEVICE
 * float[3] = {d_rl, d_fk, d_loss}.  For the frame-level GAN (g_type RSRGAN_G_DNN: models/gan.py,
 * scripts/train_gan_dnn.py) the same entry points are used with T = 1, x [N,1,Din*(L+1+R)], labels
 * [N,1,Dout]; lengths may be NULL there.  train=0 gives the eval fetch
 * (train_gan_rnn_placeholder.py:154-160): losses only, no update.
 * Every buffer is read in `stream` order -- unless the process runs with RSRGAN_DPIPE=1, by which the CALLER guarantees that
 * `labels` and `lengths` (=2: `noise_real` too) are complete when the call is made: they are then read on a side stream, possibly
 * before earlier work on `stream` has finished, so that D(real) of this call can run beside the previous call's tail
 * (DESIGN.md 6-R5 (13)).  Results do not depend on it. */
int rsrgan_d_step(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths,
                  int32_t T, const float* noise_real, const float* noise_fake,
                  float* out_losses, int32_t train, void* stream);

/* sess.run([model.g_opt, model.g_adv_losses, model.g_mse_losses, model.g_l2_losses,
 *           model.g_losses], feed) (train_gan_rnn_placeholder.py:94-101).
 * out_losses: DEVICE float[4] = {g_adv, g_mse, g_l2, g_loss}.
 * reuse_g_forward=1: the generator forward of the immediately preceding
 * rsrgan_d_step / rsrgan_d_backward on the SAME batch is still valid (G did not
 * change in between) and is not recomputed. */
int rsrgan_g_step(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths,
                  int32_t T, const float* noise_fake, float* out_losses,
                  int32_t train, int32_t reuse_g_forward, void* stream);

/* Data-parallel split of the two steps (gan_rnn_placeholder.py:164-184):
 *   *_backward  = per-tower compute_gradients (:169,:173) into the gradient
 *                 buffer, losses as above;
 *   caller      = average_gradients over towers (utils/ops.py:343-376) as an
 *                 RCCL all-reduce(avg) of rsrgan_grad_buffer();
 *   *_apply     = clip_by_norm per tensor (:178-182) then apply_gradients
 *                 (:183-184) and the EMA (:185-186). */
int rsrgan_d_backward(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths,
                      int32_t T, const float* noise_real, const float* noise_fake,
                      float* out_losses, void* stream);
int rsrgan_g_backward(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths,
                      int32_t T, const float* noise_fake, float* out_losses,
                      int32_t reuse_g_forward, void* stream);
int rsrgan_apply(rsrgan_handle h, int32_t net, void* stream);
/* device pointer + float count of the (padded) flat gradient buffer of `net`;
 * padding entries are always zero, so it can be all-reduced as one message. */
int rsrgan_grad_buffer(rsrgan_handle h, int32_t net, float** ptr, int64_t* count);
/* Gradient buckets (average_gradients iterates per variable, utils/ops.py:356-375): the flat buffer as
 * contiguous float ranges [offset, offset+count) listed in the order the backward pass completes them (generator, merged
 * wavefront backward: output FC, input FC, LSTM layer 0..L-1; otherwise one bucket = the whole buffer).  The ranges tile
 * the buffer exactly.  rsrgan_grad_bucket_wait makes `stream` wait (hipStreamWaitEvent) until bucket i of the most recent
 * *_backward on this handle is final, so the caller can all-reduce bucket i on a communication stream while the weight
 * gradients of the later buckets are still being computed. */
int rsrgan_grad_bucket_count(rsrgan_handle h, int32_t net);
int rsrgan_grad_bucket_info(rsrgan_handle h, int32_t net, int32_t i, int64_t* offset, int64_t* count);
int rsrgan_grad_bucket_wait(rsrgan_handle h, int32_t net, int32_t i, void* stream);

/* Live timing of the dominant kernel (k_fwd_gates: LSTMCell gates + cell update of every (layer, t) job of a wavefront
 * diagonal) for bench.py's roofline object: between rsrgan_profile_begin and rsrgan_profile_read every launch of that
 * kernel is bracketed by HIP events on the stream it runs on; read returns the launch count, the summed event time and
 * the summed algorithmic FLOPs (2*N*(I+P)*4H per job; the layer-0 x-part is excluded when it was batched into a GEMM). */
int rsrgan_profile_begin(rsrgan_handle h);
int rsrgan_profile_read(rsrgan_handle h, int32_t* launches, double* total_us, double* alg_flops);
/* the same window for another kernel class: kind 0 = k_fwd_gates (as above), kind 1 = k_glstm_fwd, the persistent launch that runs the
 * generator's whole forward recurrence (csrc/gpersist.hip; algorithmic FLOP = every layer's input product -- layer 0's included: it runs
 * inside the launch since round 4 --, recurrent product and projection, over the CALLER's rows: padding rows of a row-padded model do
 * not count), kind 2 = k_glstm_bwd, its BPTT (state-gradient product, dh = dm . W_p^T, the input-gradient product
 * above layer 0; in the G-run the launch is k_glstm_bwd_dt, which also carries the discriminator's BPTT in its trailing form,
 * csrc/dpersist_dev.h: that half's state- and input-gradient products, dh = dm . W_p^T and dy . W_out^T are counted too), kind 3 =
 * k_glstm_fwd_dt, the forward launch with D(G(x)) trailing inside it (the D-run under RSRGAN_DPIPE, every G-run that recomputes the
 * forward): the NUMBER of launches only (total_us and alg_flops come back 0).  Call before rsrgan_profile_read (which closes the window). */
int rsrgan_profile_read_kind(rsrgan_handle h, int32_t kind, int32_t* launches, double* total_us, double* alg_flops);

/* Health of the persistent recurrence kernels (csrc/dpersist.hip, csrc/gpersist.hip): synchronises the handle's stream and returns in
 * *code 0, or 1 + the first workgroup whose bounded wait for another workgroup's partials expired -- of a discriminator launch as is,
 * of a generator launch (k_glstm_fwd / k_glstm_bwd; in k_glstm_bwd_dt each half reports as its own kind) with 0x10000 added -- and clears the (sticky) device word.  A failed launch has
 * already poisoned its step's losses with NaN.  A failure means the launch's workgroups were not all resident at once (CUs taken
 * away after rsrgan_create, which asks the device how many it can hold: csrc/gpersist.hip resident_probe): the handle re-arms its
 * hand-off rings and takes the launch-per-phase path for that recurrence from then on.  Nothing in the reference corresponds to this. */
int rsrgan_device_status(rsrgan_handle h, int32_t* code);

/* tf.nn.dropout(h, keep_prob) after every hidden ReLU of the frame-level nets (models/dnn.py:86,99,116-121 and
 * models/discriminator_dnn.py:68,81,100-105; `--keep_prob` of scripts/train_gan_dnn.py).  0 < keep_prob <= 1.  As in the
 * reference it only acts in training runs with l2_scale > 0 (dnn.py:67-71 resets keep_prob to 1.0 otherwise).  `seed` selects
 * the mask stream (give every rank its own); masks change with every training run.
 * On the sequence model it is tf.contrib.rnn.DropoutWrapper(cell, output_keep_prob=keep_prob) around every generator layer
 * (models/lstm.py:99-102, models/res_lstm_l.py:96-99; the discriminator has none): the output of (layer, t) that feeds the layer
 * above / the output FC / the residual sum is dropped, the carried state is not; is_training only, no l2_scale condition
 * (lstm.py:71-72).  RSRGAN_ERR_INVALID for generator layers without a projection. */
int rsrgan_set_dropout(rsrgan_handle h, float keep_prob, uint64_t seed);

/* launches of the recurrence kernels (gates / projection / backward A, B, B-reduce) the host issued since rsrgan_profile_begin: with the
 * floor of a dependent launch (rsrgan_op_launch_floor) this is the serial-recurrence latency bound SURVEY 8d asks bench.py to report */
int rsrgan_profile_launches(rsrgan_handle h, int64_t* n);
/* microseconds per launch of a replayed hipGraph of n dependent 256-workgroup launches: mode 0 = empty kernels (the kernel boundary),
 * mode 1 = each reads 1 KB per wave of what its predecessor wrote and stores it back (one dependent operand round trip) */
int rsrgan_op_launch_floor(int32_t n, int32_t mode, double* us_per_launch, void* stream);

/* ---- low-level operator entry points (unit parity tests + micro-benchmarks) ----
 * C[M,N] = op(A)*op(B) (+bias) with fp32 MFMA.  a_kcontig: A is [M,K] row-major
 * (else stored [K,M]); b_kcontig: B is stored [N,K] (else [K,N] row-major).
 * All leading dimensions must be multiples of 4 floats, pointers 16-byte aligned.
 * act: 0 none, 1 leaky-relu(alpha).  accumulate: C += result. */
int rsrgan_op_gemm(const float* A, int32_t lda, int32_t a_kcontig,
                   const float* B, int32_t ldb, int32_t b_kcontig,
                   float* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                   const float* bias, int32_t act, float alpha, int32_t accumulate, void* stream);

/* ---- SEGAN-style conv G/D (models/segan.py:SEGAN with generator.py:AEGenerator, discriminator.py:discriminator, utils/bnorm.py:VBN;
 * BASELINE.json configs[4]).  The reference's trainer cannot run as shipped (segan.py:136 calls an undefined variables_on_gpu0(),
 * scripts/train_segan.py:20 imports a missing module); the graph it would build is fully specified and is what these entry
 * points compute.  One run = sess.run([model.d_opt, model.d_losses[0]]) / sess.run([model.g_opt, model.g_losses[0]])
 * (scripts/train_segan.py:32-52).  x [B, input_len], labels [B, output_dim]; the random draws of a run are INPUTS: z
 * [B, len(code), depth_last] (generator.py:201-205) and one gaussian_noise_layer draw [B, input_len + output_dim] per
 * discriminator call -- reference ("dummy") pass, real, fake (discriminator.py:74; NULL = std 0). */
typedef struct rsrgan_segan_cfg {
  int32_t batch_size;      /* args.batch_size (also the VBN mixing weight 1 / (B + 1), bnorm.py:37) */
  int32_t input_len;       /* input_dim * (left_context + 1 + right_context) (segan.py:96-100) */
  int32_t output_dim;      /* units of the generator's last dense layer (generator.py:283-287) */
  int32_t n_layers;        /* len(g_enc_depths) = len(d_num_fmaps) = 11 (segan.py:89-91) */
  int32_t g_depths[16];    /* 16,32,32,64,64,128,128,256,256,512,1024; multiples of 16 */
  int32_t d_depths[16];
  int32_t g_kwidth;        /* 20 (generator.py:151) */
  int32_t d_kwidth;        /* 31 (discriminator.py:79,88) */
  int32_t g_prelu;         /* args.g_nl == 'prelu' (run_segan.sh:120); 0 = leakyrelu */
  float   lrelu_alpha;     /* utils/ops.py:120 (0.3) */
  float   vbn_eps;         /* bnorm.py:17 (1e-5) */
  float   rms_decay;       /* tf.train.RMSPropOptimizer defaults (segan.py:123-124): 0.9 */
  float   rms_eps;         /* 1e-10 */
} rsrgan_segan_cfg;
typedef struct rsrgan_segan_handle_s* rsrgan_segan_handle;
enum { RSRGAN_SEGAN_G_LR = 0, RSRGAN_SEGAN_D_LR = 1, RSRGAN_SEGAN_L1_LAMBDA = 2 };   /* segan.py:110-111,106 */
int rsrgan_segan_default_cfg(rsrgan_segan_cfg* cfg);
int rsrgan_segan_create(const rsrgan_segan_cfg* cfg, uint64_t seed, rsrgan_segan_handle* out);
int rsrgan_segan_destroy(rsrgan_segan_handle h);
int rsrgan_segan_set_scalar(rsrgan_segan_handle h, int32_t which, double v);
/* variable table = tf.trainable_variables() split by the g_/d_ prefix (segan.py:269-283), graph-construction order */
int rsrgan_segan_num_tensors(rsrgan_segan_handle h, int32_t net);
int rsrgan_segan_tensor_info(rsrgan_segan_handle h, int32_t net, int32_t idx, char* name, int32_t name_cap, int32_t* rows, int32_t* cols,
                             int64_t* dense_offset);
int64_t rsrgan_segan_param_count(rsrgan_segan_handle h, int32_t net);
/* what: 0 = variables, 1 = the RMSProp "rms" slots, 2 = last gradients (tower-local or all-reduced); dense flat DEVICE vectors */
int rsrgan_segan_get_params(rsrgan_segan_handle h, int32_t net, int32_t what, float* dense, void* stream);
int rsrgan_segan_set_params(rsrgan_segan_handle h, int32_t net, int32_t what, const float* dense, void* stream);
/* G(x) [B, output_dim] (model.Gs, segan.py:194-197) */
int rsrgan_segan_forward_g(rsrgan_segan_handle h, const float* x, const float* z, float* y, void* stream);
/* per-tower compute_gradients (segan.py:139-146): losses DEVICE float[3] = {d_rl, d_fk, d_loss} / {g_adv, g_l1, g_loss}; train = 0: losses only */
int rsrgan_segan_d_backward(rsrgan_segan_handle h, const float* x, const float* labels, const float* z, const float* noise_ref,
                            const float* noise_real, const float* noise_fake, float* out_losses, int32_t train, void* stream);
int rsrgan_segan_g_backward(rsrgan_segan_handle h, const float* x, const float* labels, const float* z, const float* noise_ref,
                            const float* noise_fake, float* out_losses, int32_t train, void* stream);
/* average_gradients (segan.py:148-149) = the caller's RCCL all-reduce(avg) of this buffer; then apply_gradients (:150-151) */
int rsrgan_segan_grad_buffer(rsrgan_segan_handle h, int32_t net, float** ptr, int64_t* count);
int rsrgan_segan_apply(rsrgan_segan_handle h, int32_t net, void* stream);

int rsrgan_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RSRGAN_H_ */
