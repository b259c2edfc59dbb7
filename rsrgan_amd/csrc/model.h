// model.h -- host-side state of one GAN_RNN replica on one MI355X: parameter tables in the
// reference's variable order, padded device buffers, activation stashes and the launch
// schedule of D-step / G-step (models/gan_rnn_placeholder.py:139-298).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <functional>
#include <vector>

#include "../../include/rsrgan.h"
#include "kernels.h"

namespace rsr {

inline int pad4(int x) { return (x + 3) & ~3; }

struct TensorDesc {
  std::string name;
  int rows, cols, ld;       // [rows][ld] in the padded flat buffer; 1-D tensors have rows == 1
  int64_t off;              // float offset in the padded flat buffer (multiple of 64)
  int64_t dense_off;        // float offset in the dense (TF-shaped) flat vector
  bool l2;                  // takes the L2 term: "bias" not in name (gan_rnn_placeholder.py:254)
  bool trainable = true;    // false: batch-norm statistics (not in tf.trainable_variables(): their gradient stays zero)
  bool is_vector;
  int xavier_fan_out = 0;   // > 0: fan_out of the xavier limit (conv: receptive field x Cout) instead of cols
  float bias_init = 0.f;    // constant initial value of a bias vector
};

struct ParamSet {
  std::vector<TensorDesc> t;
  int64_t padded = 0, dense = 0;
  float *w = nullptr, *g = nullptr, *m = nullptr, *v = nullptr, *ema = nullptr;
  ChunkTable ct{};
  float* partial = nullptr;      // [n_chunks] sum of squares per chunk
  int add(const std::string& name, int rows, int cols, bool is_vector);
  float* W(int i) const { return w + t[i].off; }
  float* Gd(int i) const { return g + t[i].off; }
};

struct LstmLayer {               // one tf.contrib.rnn.LSTMCell(H, use_peepholes, num_proj): P = output/recurrent width
  bool has_proj = true;          // num_proj=None: m = h, P == H, no projection kernel (tWp = -1)
  int I, H, P, ldI, ldP, ldH;
  int tK, tb, twf, twi, two, tWp;        // indices into the ParamSet
  float *KxT = nullptr, *KhT = nullptr, *WpT = nullptr;   // k-contiguous transposed copies (forward: layer-0 GEMMs, folded kernels)
  // fragment-tiled copies the step kernels stream with contiguous 1 KB wave-loads (kernels.h SwizzleJob), refreshed with the above
  float *Wg_full = nullptr, *Wg_h = nullptr;              // gates: [x | m] . K and m . K[I:] alone (x-part batched)
  float *WpT_sw = nullptr, *Wp_sw = nullptr;              // projection forward / backward phase A
  float *Kb_full = nullptr, *Kb_rec = nullptr;            // backward phase B: rows [0, I+P) and rows [I, I+P) of K
};

struct ConvLayer {               // tf.contrib.layers.conv2d([S, fw], SAME) of models/rced.py: weights [S*fw*Cin][Cout] (= [S, fw, Cin, Cout])
  int fw, Cin, Cout, K, ldK, ldCin, ldCout, tW, tb;
  // normalizer_fn=batch_norm (rced.py:67-72): no biases (tb = -1); moments per output channel over the M = N*S*W positions
  bool bn = false;
  int tbn[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  float* pre = nullptr;          // [M][ldCout] conv output before the normaliser
  float* stat = nullptr;         // [BN_STAT_ROWS][ldCout]
};

struct FcLayer {                 // contrib.layers.fully_connected: weights [in][out], biases [out]
  int in, out, ld_in, ld_out, tW, tb;
  // normalizer_fn=batch_norm(scale=True, renorm=True) (dnn.py:56-61): no biases (tb = -1), <scope>/BatchNorm/* instead
  bool bn = false;
  int tbn[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // beta gamma moving_mean moving_variance renorm_mean renorm_mean_weight renorm_stddev renorm_stddev_weight
  float* pre = nullptr;          // [rows][ld_out] x.W before the normaliser (kept for the backward pass)
  float* stat = nullptr;         // [2 calls][BN_STAT_ROWS][ld_out]: the real | fake discriminator calls keep their own moments
};

struct LstmStash {               // everything one dynamic_rnn keeps for BPTT, time-major
  float *gates = nullptr;        // [T][N][4H]   zx -> gate activations -> dz (in place)
  float *c = nullptr;            // [T+1][N][H]  c[0] = 0
  float *h = nullptr;            // [T][N][ldH]
  float *mst = nullptr;          // [T+1][N][ldP] carried m state, mst[0] = 0
  float *out = nullptr;          // [T][N][ldP]  masked output
  float *dmt = nullptr;          // [T][N][ldP]  total dm per step
  float *dc = nullptr;           // [N][H]   carried during BPTT
  float *dmst = nullptr;         // [N][ldP] carried during BPTT
};

// One dynamic_rnn layer applied to N rows of a (possibly row-stacked) time-major buffer set.
// A chain = the layers of one stack, bottom first; independent chains can share launches.
struct LayerRun {
  const ParamSet* ps = nullptr;
  const LstmLayer* L = nullptr;
  LstmStash* S = nullptr;
  const float* in = nullptr;      // [T][Ns][ldI] layer input (row 0 of the stacked buffer)
  int N = 0, Ns = 0, row0 = 0;    // rows processed, rows per time step in every buffer, first row
  const int* len = nullptr;       // lengths of the N rows
  bool zx_batched = false;        // x-part precomputed for all t by one GEMM (needs Ns == N, row0 == 0)
  const float* res_in = nullptr;  // forward: res_out[t] = out[t] + res_in[t]
  float* res_out = nullptr;
  // backward
  const float* dout = nullptr;    // [T][Ns][ldP] gradient of the masked outputs
  float* din = nullptr;           // [T][Ns][ldI] gradient w.r.t. the layer input (nullptr = not needed)
  bool din_accumulate = false;
  bool want_wgrads = false;
  DropSpec drop{};                // DropoutWrapper on this layer's output (tag = the layer's; the jobs add t); ctr == nullptr: off
};
typedef std::vector<LayerRun> Chain;

// A per-time-step fully_connected stage riding a wavefront (wave mode only).
//  forward : y[t] = in[t] . W + b  -> y (stride ldy) and, with noise, -> out2 rows of a stacked buffer
//  backward: dst[t] (=|+=) src[t] . W^T
struct FcStage {
  int offset = 0;                 // diagonal at which time step 0 (forward) / T-1 (backward) runs
  int N = 0, K = 0, D = 0;        // rows, input width, output width
  const float* in = nullptr; int ld_in = 0;
  const float* WT = nullptr;      // forward: [D][ld_in] transposed weights ; backward: W [K'][ldk] TF layout
  const float* bias = nullptr;
  const float* noise = nullptr;
  float* y = nullptr; int ldy = 0;
  float* out2 = nullptr; int ld2 = 0, Ns2 = 0, row02 = 0;
  bool accumulate = false;
};

struct Model {
  rsrgan_cfg cfg{};
  int B = 0, Tmax = 0, Din = 0, Dout = 0, ldDin = 0, ldDout = 0;
  // B = rows per frame of every internal buffer; Bt = the caller's batch_size.  They differ when the batch is PADDED up to a multiple
  // of the persistent generator kernels' 32-row group (run_gan_rnn_placeholder.sh:126 ships batch_size=8, decode feeds 1): the
  // padding rows have length 0 -- dynamic_rnn's masking makes them inert (zero outputs, zero gradients) -- and the loss kernels leave
  // them out of every mean (pad_Bp() / Bt).  RSRGAN_PAD_ROWS=0: off.
  int Bt = 0;
  int pad_Bp() const { return Bt != B ? B : 0; }
  int gR = 0, dR = 0;            // output width of the generator / discriminator LSTM stacks (num_proj, or cells when None)
  ParamSet G, D;
  std::vector<LstmLayer> gl, dl;
  int g_fc_in_w = -1, g_fc_in_b = -1, g_fc_out_w = -1, g_fc_out_b = -1, d_fc_w = -1, d_fc_b = -1;

  // generator activations (N = B rows per frame)
  float *x_tm = nullptr, *lab_tm = nullptr, *g_h0 = nullptr, *y_tm = nullptr;
  std::vector<float*> g_ins;      // g_ins[l] = input of layer l, g_ins[L] = input of the output FC
  std::vector<float*> g_res;      // res_lstm_l: out_l + in_l buffers (g_ins[l+1] aliases these)
  std::vector<LstmStash> g_st;
  float *g_dA = nullptr, *g_dB = nullptr, *g_dC = nullptr;   // ping-pong gradient buffers [T*B][max ld]
  // discriminator activations (N = 2B rows per frame)
  float *xd = nullptr, *logits = nullptr, *dlogits = nullptr;
  std::vector<LstmStash> d_st;
  float *d_dA = nullptr, *d_dB = nullptr, *last_dx0 = nullptr;
  int *len_dev = nullptr;        // [2B]: lengths duplicated for the real|fake stacked batch
  float *zeros = nullptr;        // 256 B of zeros
  // ---- folded small-cell recurrence (the discriminator running alone): m_{t-1}.Kh = h_{t-1}.(Wp.Kh) and, above layer 0,
  // x_t.Kx = h^{below}_t.(Wp^{below}.Kx), so with the folded kernels Kf = [Kx' ; Wp.Kh] the cell's recurrent state is h itself
  // (a num_proj=None cell): ONE launch per time step instead of gates + projection.  Same math re-associated (fp32, ~1e-7);
  // the masked outputs out_t = h_t.Wp of the top layer follow as one time-batched GEMM (h_t is stored 0 on masked rows).
  std::vector<LstmLayer> dl_fold;                         // folded views of dl (has_proj = false, I' = H below, P' = H)
  std::vector<float*> dl_fold_K;                          // Kf [(I' + H)][4H], refreshed with the other weight copies
  std::vector<LstmStash> d_fold_st;                       // gates / c / h alias d_st; mst = carried h [T+1][2B][ldH]
  bool fold_env = true;                                   // RSRGAN_DFOLD=0: off
  bool fold_forward(Chain& ch, int T, hipStream_t s);     // false: not applicable -> caller falls back
  void refresh_fold(hipStream_t s);
  // ---- persistent recurrence (dpersist.hip): the same chain as ONE launch; RSRGAN_DPERSIST=0: off (bits below)
  unsigned long long* dp_gran = nullptr;
  unsigned* dp_ctl = nullptr;                             // kernels.h DP_CTL_*
  size_t dp_gran_bytes = 0;
  int dp_env = 3;                                         // RSRGAN_DPERSIST: bit 0 the forward launch, bit 1 the backward launch
  bool persist_forward(Chain& ch, int T, hipStream_t s);  // false: not applicable -> caller falls back to fold_forward
  bool persist_backward(Chain& ch, int T, hipStream_t s); // BPTT of the chain + its weight gradients; false: not applicable
  // the G-run's discriminator BPTT in its trailing form (dpersist_dev.h): fills dt_args for the k_glstm_bwd_dt launch that
  // persist_backward_g makes next (gp_trail_next): dy += layer 0's input gradient, dtop = dy . W_out^T step by step
  bool persist_backward_trail(Chain& ch, int T, hipStream_t s, float* dy, int ld_dy, float* dtop, int ld_dtop, bool check_only = false);
  bool persist_forward_g_trail(Chain& ch, int T, hipStream_t s, const float* nf, bool check_only = false);      // k_glstm_fwd_dt: the generator's forward recurrence + D(G(x)) behind it, one launch
  // RSRGAN_DPIPE=1 (the caller guarantees that labels, lengths and noise_real of rsrgan_d_step are complete when the call is made): the
  // D-run's D(real) -- which depends on nothing of the generator -- leaves the chain: staged and run on the side stream as soon as the
  // previous run no longer needs the discriminator's stash (ev_dfree), i.e. beside the previous G-run's weight-gradient GEMMs when the
  // host runs ahead; the D-run itself is k_glstm_fwd_dt with D(G(x)) trailing + the stacked BPTT.
  bool dpipe = false;
  bool dfree_inside = false;                              // this call recorded ev_dfree itself (behind the fused backward launch)
  bool dfree_current = false;                             // ev_dfree has been recorded behind the last run that touched the discriminator's stash / input rows
  int gp_phase = 0;                                       // persist_backward_g: 1 the launch only, 2 what follows it (the G-run split in two graph segments around ev_dfree)
  hipEvent_t ev_dfree = nullptr, ev_real = nullptr;
  unsigned long long* dp_gran2 = nullptr;                 // granules / control block of the D(real) launch (it may overlap another discriminator launch's epilogue)
  unsigned* dp_ctl2 = nullptr;
  bool persist_forward_real(int T, hipStream_t q, bool check_only = false);
  bool trail_fits = false;                                // both launches resident at once (resident_probe at init)
  bool gp_trail_next = false;                             // the next generator BPTT launch is k_glstm_bwd_dt (dt_args)
  int trail_mode = 1;                                     // RSRGAN_TRAIL: 0 off
  DPersistArgs dt_args{};                                 // (mode 1) the discriminator half of the next k_glstm_bwd_dt launch
  // ---- persistent GENERATOR recurrence (gpersist.hip): the forward pass of the generator's stack as ONE launch (weights resident
  // in VGPRs / LDS for all T steps); RSRGAN_GPERSIST bit 0.  Needs B % 32 == 0, projected cells, no residual sums, no dropout.
  unsigned long long *gp_gran1 = nullptr, *gp_gran2 = nullptr, *gp_gran3 = nullptr;
  unsigned* gp_ctl = nullptr;
  size_t gp_gran2_bytes = 0;
  int gp_env = 3;                                         // RSRGAN_GPERSIST: bit 0 the forward launch, bit 1 the backward launch (0: the launch-per-phase wavefront)
  int gp_Tcap = 0;                                        // the rings are sized for min(max_frames, GP_TMAX) steps; longer batches take the launch path
  // the discriminator's weight gradients inside its stand-alone BPTT launch (dpersist.hip dp_dw_body; RSRGAN_DW_INKERNEL=0: the GEMM /
  // column-sum launches behind it): per-(layer, tile) partial sums, progress words, the tensors' offsets inside a record (device)
  float* dw_ws = nullptr; unsigned* dw_flag = nullptr; long long* dw_src = nullptr; size_t dw_stride = 0;
  bool d_partial_fresh = false;                           // k_dw_reduce has just left D.partial (the clip's sums of squares): the inlined update skips k_sumsq
  int dp_max_grid = 0;                                    // largest discriminator launch the device proved it can hold (resident_probe)
  void persist_disable(int which);                        // after a reported failure: 0 = discriminator, 1 = generator launches off for this handle
  bool gp_fwd_on() const { return gp_gran1 && (gp_env & 1); }
  bool gpersist_shape(GPersistArgs& a, int T) const;      // sizes + plan only (no buffers)
  int gp_np_nt = 0;                                       // ... its gate tiles per workgroup, fixed at init by the resident probe (0: not decided)
  bool gp_noproj = false;                                 // the generator's cells are unprojected (num_proj=None): the single-hop form (k_glstm_np_fwd; forward only)
  bool gpersist_args(GPersistArgs& a, int T) const;       // false: not applicable
  void gpersist_rearm();                                  // the "not written" pattern in every ring slot (after allocation, after a failed launch)
  bool persist_forward_g(int T, hipStream_t s);           // layer 0's x-part batched first; fills the complete stash of every layer
  typedef std::function<void(hipStream_t)> StreamFn;
  bool persist_backward_g(Chain& ch, int T, hipStream_t s, bool check_only = false, const StreamFn& pre = nullptr, const StreamFn& post = nullptr);   // BPTT of the generator chain (k_glstm_bwd), layer 0's input gradient as a GEMM, the weight gradients unless deferred
  // fully-connected stacks: models/dnn.py generator and models/discriminator_dnn.py discriminator
  std::vector<FcLayer> gfc, dfc;
  std::vector<float*> g_act, d_act;        // act[l] = input of FC layer l, act[L] = output of the stack
  float *fc_dA = nullptr, *fc_dB = nullptr, *joint = nullptr, *dy_buf = nullptr;
  int ldJ = 0;
  int *adam_t_dev_d = nullptr;
  bool g_dnn() const { return cfg.g_type == RSRGAN_G_DNN || cfg.g_type == RSRGAN_G_RCED; }   // frame-level generator
  bool g_rced() const { return cfg.g_type == RSRGAN_G_RCED; }
  // R-CED generator (dnn.cpp): rc_act[l] = input of conv layer l as [M][ldCin] positions x channels, rc_act[L] = its output
  std::vector<ConvLayer> gconv;
  FcLayer rc_fc{};
  std::vector<float*> rc_act;
  float *rc_col = nullptr, *rc_dcol = nullptr, *rc_dA = nullptr, *rc_dB = nullptr;
  std::vector<float*> rc_cols;     // per-layer patch matrices kept from the forward pass (rc_keep_cols)
  std::vector<float*> rc_ft_fwd, rc_ft_bwd;   // prepared filters of the implicit-GEMM conv (conv.hip); nullptr = patch-matrix path
  std::vector<char> rc_wgrad_implicit;        // per layer: weight gradient by k_conv_wgrad
  float* rc_wg_ws = nullptr;                  // its partial tiles
  float* rc_x4 = nullptr;                     // layer 0's single-channel input as [positions][4] when it takes the implicit path
  bool rc_implicit = true;         // RSRGAN_RCED_IMPLICIT=0: patch-matrix GEMMs everywhere (the first correct path, kept for A/B)
  bool rc_keep_cols = false;
  size_t scratch_floats = 0;
  int rcS = 0, rcW = 0;
  void g_frame_forward(int rows, hipStream_t s);                        // DNN or R-CED generator on `rows` frames
  void g_frame_backward(int rows, float* dy, hipStream_t s);            // parameter gradients from d(output)
  void rced_forward(int rows, hipStream_t s);
  void rced_backward(int rows, float* dy, hipStream_t s);
  bool d_dnn() const { return cfg.d_type == RSRGAN_D_DNN; }
  bool d_adam() const { return g_dnn(); }     // models/gan.py:125 (Adam) vs gan_rnn_placeholder.py:144 (SGD)
  // `calls` = how many batch-norm calls the `rows` rows are (1, or 2 = the discriminator's real | fake halves, each with its own
  // batch moments); row0 = first row of act[] / pre to work on (the fake half alone in the G-run's backward pass)
  void fc_forward(const ParamSet& ps, const std::vector<FcLayer>& L, const std::vector<float*>& act, int rows, hipStream_t s, int calls = 1,
                  int call0 = 0);
  // tf.nn.dropout after every hidden ReLU of the frame-level nets (dnn.py:86,99, discriminator_dnn.py:68,81).  The reference resets
  // keep_prob to 1.0 unless l2_scale > 0 and is_training (dnn.py:67-71, discriminator_dnn.py:47-51): so does drop_training().
  // Masks are a counter-based hash of (seed, training run, net, layer, call, element): every sess.run draws fresh ones.
  float keep_prob = 1.f;
  uint64_t drop_seed = 0, drop_run = 0;
  bool drop_training() const {
    return g_dnn() && keep_prob < 1.f && !cfg.cross_validation && !bn_eval_call && scal[RSRGAN_L2_SCALE] > 0.0;
  }
  uint64_t drop_key(int net, int layer, int call) const;
  // sequence generators: tf.contrib.rnn.DropoutWrapper(cell, output_keep_prob) on every layer (models/lstm.py:99-102,
  // res_lstm_l.py:96-99; the discriminator has none), is_training only -- no l2_scale condition on this path (lstm.py:71-72)
  unsigned long long* drop_ctr = nullptr;       // device: index of the training run (kernels.h DropSpec)
  bool seq_drop_on() const { return !g_dnn() && keep_prob < 1.f && !cfg.cross_validation && !bn_eval_call; }
  unsigned drop_thr() const { return (unsigned)((double)keep_prob * 16777216.0); }
  float* fc_backward(const ParamSet& ps, const std::vector<FcLayer>& L, const std::vector<float*>& act, int rows, float* dtop,
                     bool want_wgrads, bool want_din, hipStream_t s, int calls = 1, int row0 = 0, int call0 = 0);
  bool bn_on() const { return (cfg.flags & RSRGAN_FLAG_BATCH_NORM) != 0; }
  // is_training (dnn.py:49-50): false on the cross_validation twin -- a model built with cross_validation=1, or a d/g run without
  // gradients on the training model (those ARE the twin's fetches on the shared variables: train_gan_dnn.py:182-215)
  bool bn_eval_call = false;
  bool bn_training() const { return bn_on() && !cfg.cross_validation && !bn_eval_call; }
  BnVars bn_vars(const ParamSet& ps, const FcLayer& F) const;
  BnVars bn_vars(const ParamSet& ps, const int (&tbn)[8]) const;
  void bn_commit_stack(const ParamSet& ps, const std::vector<FcLayer>& L, int times0, int times1, BnCommitList& cl);
  float* bn_sums = nullptr;      // [2][max ld_out] work space of launch_bn_backward
  void d_dnn_forward_loss(int T, int Nd, int n_real, bool want_grads, float* loss3, hipStream_t s, int calls = 1, int call0 = 0);
  void bn_commit_run(bool with_d, hipStream_t s);
  int dnn_d_backward(const float* x, const float* labels, int T, float* out_losses, bool want_grads, hipStream_t s);
  int dnn_g_backward(const float* x, const float* labels, int T, float* out_losses, bool want_grads, bool reuse, hipStream_t s);

  float *dyn = nullptr;          // device scalars (see kernels.hip DYN_*)
  int *adam_t_dev = nullptr;
  float *losses = nullptr;       // [8]: d_rl d_fk d_loss | g_adv g_mse g_l2 g_loss | tmp
  float *tmp3 = nullptr;
  float *scratch = nullptr;      // colsum / mse partials
  double scal[RSRGAN_SCALAR_COUNT_] = {0};
  bool g_fwd_valid = false;
  int cur_T = 0;
  bool d_grads_ready = false, g_grads_ready = false;
  std::vector<void*> allocs;

  int init(const rsrgan_cfg& c, uint64_t seed);
  void destroy();

  // steps
  int prepare_batch(const float* x, const float* labels, const int32_t* lengths, int T, hipStream_t s, const float** nr = nullptr,
                    const float** nf = nullptr, hipStream_t early = nullptr);      // nr / nf: callers' noise, staged in the same launch (in/out: the staged copy)
  void g_forward(int T, hipStream_t s, Chain* extra = nullptr);
  void g_forward_head(int T, hipStream_t s);
  void g_forward_tail(int T, hipStream_t s);
  void d_logits(int N, int T, hipStream_t s);
  void d_backward_pass(int N, int T, bool want_wgrads, bool need_dx0, const float* dlog, hipStream_t s, bool head_done = false);
  bool d_head(int N, int T, int n_real, const float* t_real, const float* t_fake, float* loss3, bool want_grads, bool want_wgrads, hipStream_t s);
  void g_backward_pass(int T, float* dy, hipStream_t s);
  int d_backward(const float* x, const float* labels, const int32_t* lengths, int T, const float* nr, const float* nf,
                 float* out_losses, bool want_grads, hipStream_t s);
  int g_backward(const float* x, const float* labels, const int32_t* lengths, int T, const float* nf,
                 float* out_losses, bool want_grads, bool reuse, hipStream_t s);
  int apply(int net, hipStream_t s);
  void apply_body(int net, hipStream_t s);     // the update's launches (clip, optimizer, EMA, weight copies)
  int apply_inlined = 0;                       // bit 0 (G) / bit 1 (D): the backward segment of this call already contains them (fused steps)
  void refresh_transposes(int net, hipStream_t s);
  void refresh_swizzles(int net, hipStream_t s);
  bool lazy_sw = false;            // the fragment-tiled copies are rebuilt where they are read (rnn_forward / rnn_backward), not after every update

  // building blocks: run chains layer-by-layer (v1) or as one fused (layer,t) wavefront
  void rnn_forward(std::vector<Chain>& chains, int T, hipStream_t s, const std::vector<int>* offsets = nullptr,
                   const std::vector<FcStage>* fcs = nullptr);
  void rnn_backward(std::vector<Chain>& chains, int T, hipStream_t s, const std::vector<int>* offsets = nullptr,
                    const std::vector<FcStage>* fcs = nullptr);
  void layer_wgrads(const LayerRun& R, int T, hipStream_t s);
  void chain_wgrads(Chain& ch, int T, hipStream_t s, const StreamFn& between, const StreamFn& pre = nullptr, const StreamFn& post = nullptr);   // all layers of a finished BPTT, on two streams
  void layer_wgrads_gemms(const LayerRun& R, int t0, int t1, bool accumulate, hipStream_t s, bool do_dK = true, bool do_dWp = true);
  int trail_nrt() const;
  bool batch_wgrads(Chain& ch, int T, hipStream_t s, bool dK_too, bool* dK_done, bool check_only = false);      // all layers of one shape: one launch per kind
  void layer_wgrads_colsums(const LayerRun& R, int T, hipStream_t s, float* scr);
  // side stream: weight-gradient GEMMs (MFMA-bound) overlap the byte-bound backward wave
  hipStream_t side = nullptr;
  // gradient buckets: contiguous ranges of the flat gradient buffer in the order the backward completes them,
  // each with an event recorded on the compute stream when its range is final -> the caller all-reduces bucket i while the
  // weight-gradient GEMMs of bucket i+1.. are still running.
  struct GradBucket { int64_t off = 0, count = 0; hipEvent_t ev = nullptr; bool marked = false; };
  std::vector<GradBucket> gbk[2];
  bool fused_apply = false;        // set by rsrgan_g_step around g_backward: the update follows in the same call (no all-reduce in between)
  bool defer_wgrads = false;       // rnn_backward leaves the weight-gradient launches of want_wgrads runs to its caller
  int build_buckets();
  void mark_bucket(int net, int i, hipStream_t s);
  void finish_buckets(int net, hipStream_t s);     // record every bucket not marked by this backward, reset the marks
  // live per-kernel timing for bench.py's roofline object: when prof_on, every k_fwd_gates launch is bracketed by HIP events on
  // the stream it runs on and its algorithmic FLOPs (2*N*(I+P)*4H per job) are summed; read back by rsrgan_profile_read.
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  int prof_n = 0;
  double prof_flops = 0.0;
  // ... and the same for the persistent generator launch (k_glstm_fwd: one launch per forward pass): rsrgan_profile_read_kind(h, 1, ...)
  std::vector<hipEvent_t> prof_gp_ev;
  int prof_gp_n = 0;
  double prof_gp_flops = 0.0;
  // ... and for the persistent generator BPTT (k_glstm_bwd: one launch per G-run): rsrgan_profile_read_kind(h, 2, ...)
  std::vector<hipEvent_t> prof_gb_ev;
  int prof_gb_n = 0;
  int prof_fdt_n = 0;               // fused forward launches (k_glstm_fwd_dt) since profile_begin
  double prof_gb_flops = 0.0;
  void gates_launch(const FwdGateJobs& gj, int blocks, int kb, hipStream_t s);
  int gates_blocks(int H, int N) const;
  int proj_blocks(int P, int N) const;
  int bwd_a_blocks(int H, int N) const;
  void run_bwd_b_splitk(BwdBJobs& bj, hipStream_t s);
  hipEvent_t ev_pool[16] = {};
  int ev_next = 0;
  float* gemm_ws2 = nullptr;
  // kernel gradients of layers whose input width is no multiple of 4 (res_lstm_l: 257): the stacked product [x (ld columns: the padding
  // is zero) | m]^T dZ lands here, [ldI + P][4H] per layer, and its two row blocks are copied into dK behind it (chain_wgrads)
  float* dk_tmp = nullptr;
  size_t dk_tmp_per = 0;
  float* scratch2 = nullptr;
  bool overlap() const { return side != nullptr && (cfg.flags & RSRGAN_FLAG_OVERLAP) != 0; }
  Chain g_chain(int T);
  Chain d_chain(int N, int Ns, int row0);
  void gemm(const float* A, int lda, bool a_kc, const float* B, int ldb, bool b_kc, float* C, int ldc, int M, int N,
            int K, const float* bias, int act, float alpha, bool accumulate, hipStream_t s);
  bool supervised() const { return (cfg.flags & RSRGAN_FLAG_SUPERVISED) != 0; }
  bool wavefront() const { return (cfg.flags & RSRGAN_FLAG_WAVEFRONT) != 0; }
  float* g_fc_out_wT = nullptr;   // [Dout][ldP] transposed copy of the output FC weights (per-step FC stage)
  float* bwdb_ws = nullptr;        // split-K partial tiles of backward phase B
  size_t bwdb_ws_floats = 0;
  bool bwd_b_splitk_ok(const BwdBJobs& jobs) const;
  float* gemm_ws = nullptr;
  size_t gemm_ws_floats = 0;

  // ---- hipGraph replay (RSRGAN_FLAG_GRAPH): for a given T the launch sequence of a step is static (device pointers of the
  // model's own buffers, job tables by value, scalars read from device memory), so each segment is run eagerly once, captured
  // on its second use and replayed afterwards: ~1.6 us per dependent kernel on the GPU instead of a host-bound 3-4.6 us per
  // eager launch (profiles/r2_ubench_launch_l2_barrier.txt).  Caller-owned pointers never enter a graph: inputs are packed
  // into the model's buffers before, losses are copied out after.  The legacy null stream cannot be captured: work handed
  // to it runs on an internal stream, ordered against the caller's stream by events (enter / leave).
  struct GraphSlot { int uses = 0; hipGraphExec_t exec = nullptr; };
  std::unordered_map<uint64_t, GraphSlot> graphs;
  hipStream_t main_s = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  hipEvent_t ev_last = nullptr; bool ev_last_set = false; // recorded behind every call on the stream it worked on (rsrgan_device_status waits for it)
  float *noise_r_buf = nullptr, *noise_f_buf = nullptr;      // staged gaussian_noise_layer draws [B][Dout]
  bool graphs_on() const { return (cfg.flags & RSRGAN_FLAG_GRAPH) != 0 && wavefront() && !overlap() && !prof_on && !g_dnn() && graphs_env; }
  bool graphs_env = true;
  template <class F> void run_seg(uint64_t key, hipStream_t s, F&& body);
  void drop_graphs();
  hipStream_t enter(hipStream_t caller);
  void leave(hipStream_t caller, hipStream_t work);
  const float* stage_noise(const float* src, float* buf, hipStream_t s);

  template <typename T> T* alloc(size_t n);
};

void set_error(const char* fmt, ...);
const char* get_error();

}  // namespace rsr
