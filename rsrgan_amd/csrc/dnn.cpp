// dnn.cpp -- fully-connected stacks: the frame-level DNN-GAN (models/gan.py:GAN, generator
// models/dnn.py:DNN, discriminator models/discriminator_dnn.py) and discriminator_dnn as the D of the
// sequence model.  Everything here is time-batched fp32-MFMA GEMM (gemm.hip) plus small epilogue kernels:
//   forward  h_{l+1} = relu(h_l . W_l + b_l)            (last layer linear)              dnn.py:79-110
//   backward dW_l = h_l^T . d ; db_l = colsum(d) ; d <- (d . W_l^T) * [h_l > 0]
#include "model.h"

namespace rsr {

#define HIPC(expr)                                                                   \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RSRGAN_ERR_HIP;                                                         \
    }                                                                                \
  } while (0)

static const float kDClipLo = -0.5f, kDClipHi = 1.5f;     // discriminator_dnn.py:93 tf.clip_by_value(y, -0.5, 1.5)

BnVars Model::bn_vars(const ParamSet& ps, const int (&tbn)[8]) const {
  BnVars v;
  v.beta = ps.W(tbn[0]); v.gamma = ps.W(tbn[1]); v.mm = ps.W(tbn[2]); v.mv = ps.W(tbn[3]);
  v.rm = ps.W(tbn[4]); v.rmw = ps.W(tbn[5]); v.rs = ps.W(tbn[6]); v.rsw = ps.W(tbn[7]);
  return v;
}
BnVars Model::bn_vars(const ParamSet& ps, const FcLayer& F) const { return bn_vars(ps, F.tbn); }

static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
uint64_t Model::drop_key(int net, int layer, int call) const {
  return splitmix64(splitmix64(drop_seed ^ (drop_run * 0xD1342543DE82EF95ull)) + (uint64_t)((net << 16) | (layer << 8) | call));
}

void Model::fc_forward(const ParamSet& ps, const std::vector<FcLayer>& L, const std::vector<float*>& act, int rows, hipStream_t s, int calls,
                       int call0) {
  const bool drop = drop_training();
  for (size_t l = 0; l < L.size(); ++l) {
    const FcLayer& F = L[l];
    if (!F.bn) {
      gemm(act[l], F.ld_in, true, ps.W(F.tW), F.ld_out, false, act[l + 1], F.ld_out, rows, F.out, F.in,
           ps.W(F.tb), l + 1 < L.size() ? 2 : 0, 0.f, false, s);
    } else {
      // relu(batch_norm(x.W)): the product once over all rows, the normaliser once per call (each call has its own batch moments)
      gemm(act[l], F.ld_in, true, ps.W(F.tW), F.ld_out, false, F.pre, F.ld_out, rows, F.out, F.in, nullptr, 0, 0.f, false, s);
      launch_bn_forward(F.pre, F.ld_out, act[l + 1], F.ld_out, rows / calls, F.out, bn_vars(ps, F), F.stat, F.ld_out, bn_training(), true,
                        scratch, scratch_floats, s, calls);
    }
    if (drop && l + 1 < L.size()) {          // h = dropout(h, keep_prob): one draw per call (D(real) and D(fake) are two ops)
      const int rpc = rows / calls;
      for (int k = 0; k < calls; ++k)
        launch_dropout_fwd(act[l + 1] + (size_t)k * rpc * F.ld_out, (size_t)rpc, F.out, F.ld_out, drop_key(&ps == &D ? 1 : 0, (int)l, call0 + k),
                           drop_thr(), keep_prob, s);
    }
  }
}

// the UPDATE_OPS of every batch-norm layer of the stack: call 0 `times0` times, then call 1 `times1` times (oracle/bn_renorm.py)
void Model::bn_commit_stack(const ParamSet& ps, const std::vector<FcLayer>& L, int times0, int times1, BnCommitList& cl) {
  for (const FcLayer& F : L)
    if (F.bn && cl.n < 24) cl.e[cl.n++] = BnCommit{bn_vars(ps, F), F.stat, F.out, F.ld_out, times0, times1};
}

// dtop: gradient w.r.t. the stack's (linear) output, [rows][ld_out of the last layer].  Returns the gradient w.r.t.
// the stack's input (in fc_dA or fc_dB, leading dimension = ld_in of layer 0) when want_din, else nullptr.
// With batch norm the rows are `calls` calls of rows/calls rows (statistics slots call0 ..), starting at row0 of act[] / pre.
float* Model::fc_backward(const ParamSet& ps, const std::vector<FcLayer>& L, const std::vector<float*>& act, int rows, float* dtop,
                          bool want_wgrads, bool want_din, hipStream_t s, int calls, int row0, int call0) {
  float* d = dtop;
  float* bufs[2] = {fc_dA, fc_dB};
  int nb = 0;
  for (int l = (int)L.size() - 1; l >= 0; --l) {
    const FcLayer& F = L[l];
    const float* a_in = act[l] + (size_t)row0 * F.ld_in;
    if (l + 1 < (int)L.size() && drop_training())          // through dropout and the ReLU under it: act[l + 1] is the dropped output
      launch_dropout_bwd(act[l + 1] + (size_t)row0 * F.ld_out, d, (size_t)rows, F.out, F.ld_out, keep_prob, s);
    if (F.bn) {                            // d (w.r.t. the ReLU's output) -> gradient w.r.t. x.W, per call; dbeta / dgamma summed over the calls
      const size_t ra = (size_t)row0 * F.ld_out;
      launch_bn_backward(d, F.ld_out, act[l + 1] + ra, F.ld_out, F.pre + ra, F.ld_out, rows / calls, F.out,
                         F.stat + (size_t)call0 * BN_STAT_ROWS * F.ld_out, F.ld_out, want_wgrads ? ps.Gd(F.tbn[0]) : nullptr,
                         want_wgrads ? ps.Gd(F.tbn[1]) : nullptr, false, true, bn_sums, scratch, scratch_floats, s, calls);
    } else if (l + 1 < (int)L.size()) {
      launch_lrelu_bwd(act[l + 1] + (size_t)row0 * F.ld_out, d, (size_t)rows, F.out, F.ld_out, 0.f, s);   // relu': d *= [h > 0]
    }
    if (want_wgrads) {
      gemm(a_in, F.ld_in, false, d, F.ld_out, false, ps.Gd(F.tW), F.ld_out, F.in, F.out, rows, nullptr, 0, 0.f, false, s);
      if (!F.bn) launch_colsum(d, F.ld_out, nullptr, 0, ps.Gd(F.tb), rows, F.out, scratch, s);
    }
    if (l > 0 || want_din) {
      float* dn = bufs[nb]; nb ^= 1;
      if (dn == d) { dn = bufs[nb]; nb ^= 1; }
      gemm(d, F.ld_out, true, ps.W(F.tW), F.ld_out, true, dn, F.ld_in, rows, F.in, F.out, nullptr, 0, 0.f, false, s);
      d = dn;
    } else {
      d = nullptr;
    }
  }
  return d;
}

// ---- R-CED generator (models/rced.py): every conv2d is patch matrix -> GEMM (+bias, ReLU); the patch matrix of a layer is
// rebuilt in the backward pass instead of being kept (it is S*fw times the size of the activation it comes from) ----
void Model::rced_forward(int rows, hipStream_t s) {
  const size_t M = (size_t)rows * rcS * rcW;
  for (size_t l = 0; l < gconv.size(); ++l) {
    const ConvLayer& L = gconv[l];
    if (rc_ft_fwd[l]) {                          // implicit GEMM: no patch matrix
      const float* src = rc_act[l];
      int ldc = L.ldCin;
      if (l == 0) { launch_expand_c4(x_tm, ldDin, rcS * rcW, rc_x4, (size_t)rows, s); src = rc_x4; ldc = 4; }
      launch_conv_fwd(src, ldc, L.Cin, rc_ft_fwd[l], L.bn ? nullptr : G.W(L.tb), !L.bn, L.bn ? L.pre : rc_act[l + 1], L.ldCout, L.Cout, rows,
                      rcS, rcW, L.fw, s);
    } else {
      float* col = rc_keep_cols ? rc_cols[l] : rc_col;
      launch_im2col(rc_act[l], l == 0 ? (size_t)ldDin : (size_t)rcS * rcW * L.ldCin, L.ldCin, L.Cin, rcS, rcW, rcS, L.fw, col, L.ldK, M, s);
      gemm(col, L.ldK, true, G.W(L.tW), L.ldCout, false, L.bn ? L.pre : rc_act[l + 1], L.ldCout, (int)M, L.Cout, L.K, L.bn ? nullptr : G.W(L.tb),
           L.bn ? 0 : 2, 0.f, false, s);
    }
    // relu(batch_norm(conv)): moments per channel over the M positions of the batch (rced.py:97-99)
    if (L.bn)
      launch_bn_forward(L.pre, L.ldCout, rc_act[l + 1], L.ldCout, (int)M, L.Cout, bn_vars(G, L.tbn), L.stat, L.ldCout, bn_training(), true,
                        scratch, scratch_floats, s);
  }
  // reshape [rows, S*W*C] (rced.py:110: contiguous because C % 4 == 0) -> linear FC
  gemm(rc_act[gconv.size()], rc_fc.ld_in, true, G.W(rc_fc.tW), ldDout, false, y_tm, ldDout, rows, Dout, rc_fc.in, G.W(rc_fc.tb), 0, 0.f, false, s);
}

void Model::rced_backward(int rows, float* dy, hipStream_t s) {
  const size_t M = (size_t)rows * rcS * rcW;
  const int Lc = (int)gconv.size();
  gemm(rc_act[Lc], rc_fc.ld_in, false, dy, ldDout, false, G.Gd(rc_fc.tW), ldDout, rc_fc.in, Dout, rows, nullptr, 0, 0.f, false, s);
  launch_colsum(dy, ldDout, nullptr, 0, G.Gd(rc_fc.tb), rows, Dout, scratch, s);
  float* d = rc_dA;
  float* other = rc_dB;
  gemm(dy, ldDout, true, G.W(rc_fc.tW), ldDout, true, d, rc_fc.ld_in, rows, rc_fc.in, Dout, nullptr, 0, 0.f, false, s);   // = [M][Cout_last]
  bool d_masked = false;                     // relu' of this layer already applied by the data-gradient kernel of the layer above
  for (int l = Lc - 1; l >= 0; --l) {
    const ConvLayer& L = gconv[l];
    if (L.bn)
      launch_bn_backward(d, L.ldCout, rc_act[l + 1], L.ldCout, L.pre, L.ldCout, (int)M, L.Cout, L.stat, L.ldCout, G.Gd(L.tbn[0]), G.Gd(L.tbn[1]),
                         false, true, bn_sums, scratch, scratch_floats, s);
    else if (!d_masked)
      launch_lrelu_bwd(rc_act[l + 1], d, M, L.Cout, L.ldCout, 0.f, s);                    // relu': d *= [a > 0]
    d_masked = false;
    bool db_done = false;                    // bias gradient already summed by the weight-gradient kernel
    if (rc_wgrad_implicit[l] && rc_wg_ws) {
      launch_conv_wgrad(l == 0 ? rc_x4 : rc_act[l], l == 0 ? 4 : L.ldCin, L.Cin, d, L.ldCout, L.Cout, G.Gd(L.tW), L.ldCout, rc_wg_ws, rows,
                        rcS, rcW, L.fw, s, L.bn ? nullptr : G.Gd(L.tb));
      db_done = !L.bn;
    } else {
      float* col = rc_keep_cols ? rc_cols[l] : rc_col;          // kept from the forward pass of the same batch, or rebuilt
      if (!rc_keep_cols)
        launch_im2col(rc_act[l], l == 0 ? (size_t)ldDin : (size_t)rcS * rcW * L.ldCin, L.ldCin, L.Cin, rcS, rcW, rcS, L.fw, col, L.ldK, M, s);
      gemm(col, L.ldK, false, d, L.ldCout, false, G.Gd(L.tW), L.ldCout, L.K, L.Cout, (int)M, nullptr, 0, 0.f, false, s);
    }
    if (!L.bn && !db_done) launch_colsum_tall(d, L.ldCout, G.Gd(L.tb), (int)M, L.Cout, scratch, scratch_floats, s);
    if (l > 0) {
      if (rc_ft_bwd[l]) {                        // d(in) = conv_SAME(d, flipped filter): the same implicit-GEMM kernel
        // (the layer below's relu' rides the epilogue: its activations rc_act[l] share the layout of d(in))
        d_masked = !gconv[l - 1].bn;
        launch_conv_fwd(d, L.ldCout, L.Cout, rc_ft_bwd[l], nullptr, false, other, L.ldCin, L.Cin, rows, rcS, rcW, L.fw, s,
                        d_masked ? rc_act[l] : nullptr);
      } else {
        gemm(d, L.ldCout, true, G.W(L.tW), L.ldCout, true, rc_dcol, L.ldK, (int)M, L.K, L.Cout, nullptr, 0, 0.f, false, s);
        launch_col2im(rc_dcol, L.ldK, L.Cin, rcS, rcW, rcS, L.fw, other, L.ldCin, M, s);
      }
      std::swap(d, other);
    }
  }
}

void Model::g_frame_forward(int rows, hipStream_t s) {
  if (g_rced()) rced_forward(rows, s);
  else fc_forward(G, gfc, g_act, rows, s);
}
void Model::g_frame_backward(int rows, float* dy, hipStream_t s) {
  if (g_rced()) rced_backward(rows, dy, s);
  else fc_backward(G, gfc, g_act, rows, dy, true, false, s);
}

// D(.) on d_act[0] ([T][Nd] rows), clipped LSGAN losses, dlogits
// every training sess.run executes all batch-norm update ops of the graph (gan.py:139-146): the generator's call twice, the
// discriminator's real-joint call twice (dummy + real, gan.py:162-181) and its fake-joint call once -- oracle/bn_renorm.py
void Model::bn_commit_run(bool with_d, hipStream_t s) {
  BnCommitList cl; cl.n = 0;
  bn_commit_stack(G, gfc, 2, 0, cl);
  for (const ConvLayer& L : gconv)
    if (L.bn && cl.n < 24) cl.e[cl.n++] = BnCommit{bn_vars(G, L.tbn), L.stat, L.Cout, L.ldCout, 2, 0};
  if (with_d) bn_commit_stack(D, dfc, 2, 1, cl);
  launch_bn_commit_many(cl, s);
}

void Model::d_dnn_forward_loss(int T, int Nd, int n_real, bool want_grads, float* loss3, hipStream_t s, int calls, int call0) {
  fc_forward(D, dfc, d_act, T * Nd, s, calls, call0);
  launch_lsgan(logits, 4, want_grads ? dlogits : nullptr, T, Nd, n_real, dyn + DYN_D_REAL,
               n_real > 0 ? dyn + DYN_D_FAKE : dyn + DYN_D_REAL, loss3, s, true, kDClipLo, kDClipHi);
}

// ---- frame-level GAN: sess.run([model.d_opt, ...]) of scripts/train_gan_dnn.py on [N, Din*(L+1+R)] frames ----
int Model::dnn_d_backward(const float* x, const float* labels, int T, float* out_losses, bool want_grads, hipStream_t s) {
  const int R = T * B;
  if (!x || !labels) { set_error("null input pointer"); return RSRGAN_ERR_INVALID; }
  if (T <= 0 || T > Tmax) { set_error("T=%d outside (0, max_frames=%d]", T, Tmax); return RSRGAN_ERR_INVALID; }
  bn_eval_call = !want_grads;
  if (want_grads) ++drop_run;
  launch_pack_tm(x, x_tm, B, T, Din, ldDin, s);
  launch_pack_tm(labels, lab_tm, B, T, Dout, ldDout, s);
  cur_T = T;
  g_frame_forward(R, s);                                 // g = self.generator(inputs, ...)  gan.py:171
  g_fwd_valid = true;
  // d_rl_joint = concat(d_inputs, labels) ; d_fk_joint = concat(d_inputs, g)   gan.py:173-174
  launch_build_joint(x_tm, ldDin, cfg.d_joint_off, cfg.d_joint_dim, lab_tm, ldDout, Dout, joint, ldJ, 0, R, s);
  launch_build_joint(x_tm, ldDin, cfg.d_joint_off, cfg.d_joint_dim, y_tm, ldDout, Dout, joint, ldJ, R, R, s);
  d_dnn_forward_loss(1, 2 * R, R, want_grads, losses, s, 2);
  if (want_grads) {
    fc_backward(D, dfc, d_act, 2 * R, dlogits, true, false, s, 2);
    if (bn_training()) bn_commit_run(true, s);
    { finish_buckets(RSRGAN_NET_D, s); d_grads_ready = true; }
  }
  if (out_losses) launch_copy_f(losses, out_losses, 3, s);
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

int Model::dnn_g_backward(const float* x, const float* labels, int T, float* out_losses, bool want_grads, bool reuse, hipStream_t s) {
  const int R = T * B;
  bn_eval_call = !want_grads;
  if (bn_on()) reuse = false;       // the D-run's update ops changed the generator's renorm state: its forward differs now
  if (want_grads) ++drop_run;
  if (drop_training()) reuse = false;   // a new sess.run draws new masks
  if (reuse) {
    if (!g_fwd_valid || T != cur_T) { set_error("reuse_g_forward without a valid generator forward"); return RSRGAN_ERR_STATE; }
  } else {
    if (!x || !labels) { set_error("null input pointer"); return RSRGAN_ERR_INVALID; }
    if (T <= 0 || T > Tmax) { set_error("T=%d outside (0, max_frames=%d]", T, Tmax); return RSRGAN_ERR_INVALID; }
    launch_pack_tm(x, x_tm, B, T, Din, ldDin, s);
    launch_pack_tm(labels, lab_tm, B, T, Dout, ldDout, s);
    cur_T = T;
    g_frame_forward(R, s);
    g_fwd_valid = true;
  }
  const bool sup = supervised();           // DNNTrainer (models/dnn_trainer.py:139-148): g_loss = g_mse + g_l2, no discriminator
  if (sup) {
    HIPC(hipMemsetAsync(losses + 3, 0, sizeof(float), s));
  } else if (bn_training() && want_grads) {
    // with batch norm the real-joint call runs too (only its update ops matter here): rows [0, R) real, [R, 2R) fake as in the D-run
    launch_build_joint(x_tm, ldDin, cfg.d_joint_off, cfg.d_joint_dim, lab_tm, ldDout, Dout, joint, ldJ, 0, R, s);
    launch_build_joint(x_tm, ldDin, cfg.d_joint_off, cfg.d_joint_dim, y_tm, ldDout, Dout, joint, ldJ, R, R, s);
    fc_forward(D, dfc, d_act, 2 * R, s, 2);
    launch_lsgan(logits + (size_t)R * 4, 4, dlogits + (size_t)R * 4, 1, R, 0, dyn + DYN_D_REAL, dyn + DYN_D_REAL, tmp3, s, true, kDClipLo, kDClipHi);
    launch_copy_f(tmp3 + 1, losses + 3, 1, s);
  } else {
    launch_build_joint(x_tm, ldDin, cfg.d_joint_off, cfg.d_joint_dim, y_tm, ldDout, Dout, joint, ldJ, 0, R, s);
    d_dnn_forward_loss(1, R, 0, want_grads, tmp3, s, 1, 1);      // g_adv = mean((D(fake) - 1)^2)  gan.py:202 (call 1 = the fake joint)
    launch_copy_f(tmp3 + 1, losses + 3, 1, s);
  }
  const bool l2_on = !cfg.cross_validation && scal[RSRGAN_L2_SCALE] > 0.0;
  if (want_grads) {
    if (!sup) {
      float* dj = bn_training() ? fc_backward(D, dfc, d_act, R, dlogits + (size_t)R * 4, false, true, s, 1, R, 1)   // the fake half
                                : fc_backward(D, dfc, d_act, R, dlogits, false, true, s);                           // d g_adv / d joint
      launch_slice_cols(dj, ldJ, cfg.d_joint_dim, dy_buf, ldDout, R, Dout, s);            // ... / d g
    }
    launch_mse(y_tm, lab_tm, ldDout, dy_buf, R, Dout, dyn + DYN_LAMBDA, !sup, losses + 4, scratch, s);
    g_frame_backward(R, dy_buf, s);
    if (l2_on) {
      launch_l2(G.w, G.g, G.ct, dyn + DYN_L2, G.partial, s);
      launch_l2_total(G.partial, G.ct.n_chunks, dyn + DYN_L2, losses + 5, s);
    }
    if (bn_training()) bn_commit_run(!sup, s);
    { finish_buckets(RSRGAN_NET_G, s); g_grads_ready = true; }
  } else {
    launch_mse(y_tm, lab_tm, ldDout, nullptr, R, Dout, dyn + DYN_LAMBDA, false, losses + 4, scratch, s);
  }
  if (!(want_grads && l2_on)) HIPC(hipMemsetAsync(losses + 5, 0, sizeof(float), s));
  launch_g_total(losses + 3, dyn + DYN_LAMBDA, s);
  if (out_losses) launch_copy_f(losses + 3, out_losses, 4, s);
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

}  // namespace rsr
