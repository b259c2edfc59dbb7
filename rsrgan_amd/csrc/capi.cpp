// capi.cpp -- extern "C" entry points declared in include/rsrgan.h.
#include <cstring>
#include <exception>
#include <new>
#include <vector>

#include "model.h"

using namespace rsr;

struct rsrgan_handle_s { Model m; };

#define CHECK_H(h)                                                    \
  if (!(h)) { set_error("null handle"); return RSRGAN_ERR_INVALID; }

// Nothing may throw across the C ABI (include/rsrgan.h): every entry point that can allocate host memory runs its body
// through guard(), which turns std::bad_alloc / any other exception into RSRGAN_ERR_INVALID with a message.
template <class F>
static int guard(const char* what, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    set_error("%s: out of host memory", what);
  } catch (const std::exception& e) {
    set_error("%s: %s", what, e.what());
  } catch (...) {
    set_error("%s: unknown C++ exception", what);
  }
  return RSRGAN_ERR_INVALID;
}
// Work handed to the legacy null stream runs on the model's own stream (hipGraph capture needs a real stream), ordered
// against the caller's stream by events on both sides.
struct StreamScope {
  Model& m; hipStream_t caller, work;
  StreamScope(Model& m_, void* s) : m(m_), caller((hipStream_t)s), work(m_.enter((hipStream_t)s)) {}
  ~StreamScope() {
    // (RSRGAN_DPIPE) whatever this call did to the discriminator's weights, stash or input rows is complete behind this point -- unless
    // the call has recorded the event itself, earlier (rsrgan_g_step / rsrgan_g_backward behind the fused backward launch)
    if (m.dpipe && !m.dfree_inside) { (void)hipEventRecord(m.ev_dfree, work); m.dfree_current = true; }
    m.dfree_inside = false;
    // rsrgan_device_status waits for THIS point of the caller's stream through an event of the handle's own: the stream itself may be
    // gone by then (a raw copy of a destroyed stream is a dangling handle)
    if (m.ev_last && hipEventRecord(m.ev_last, work) == hipSuccess) m.ev_last_set = true;
    m.leave(caller, work);
  }
};

extern "C" {

const char* rsrgan_last_error(void) { return get_error(); }
int rsrgan_version(void) { return 100; }

int rsrgan_default_cfg(int32_t g_type, rsrgan_cfg* c) {
  if (!c) { set_error("null cfg"); return RSRGAN_ERR_INVALID; }
  std::memset(c, 0, sizeof(*c));
  c->batch_size = 8; c->max_frames = 100; c->input_dim = 257; c->output_dim = 40;
  c->g_type = g_type;
  if (g_type == RSRGAN_G_LSTM) { c->g_layers = 3; c->g_cells = 760; c->g_proj = 280; }           // models/lstm.py:43-45
  else if (g_type == RSRGAN_G_RES_LSTM_L || g_type == RSRGAN_G_RES_LSTM_BASE) { c->g_layers = 4; c->g_cells = 760; c->g_proj = 257; }  // models/res_lstm_l.py:43-45
  else if (g_type == RSRGAN_G_DNN) { c->g_layers = 4; c->g_cells = 1024; c->g_proj = 0; }        // models/dnn.py:34-35 (1+3 hidden layers)
  else if (g_type == RSRGAN_G_RCED) { c->g_layers = 9; c->g_cells = 32; c->g_proj = 0; c->g_splice = 11; }   // models/rced.py:92-93 (fixed filter table)
  else { set_error("Unrecognized G type %d", g_type); return RSRGAN_ERR_INVALID; }
  c->d_type = RSRGAN_D_LSTM; c->d_layers = 2; c->d_cells = 256; c->d_proj = 40;                   // models/discriminator_lstm.py:26-28
  c->l2_scale = 0.f; c->clip_norm = 15.f;
  if (g_type == RSRGAN_G_DNN || g_type == RSRGAN_G_RCED) {        // models/gan.py: discriminator_dnn on concat(centre frame, target), Adam/Adam, no clipping
    c->input_dim = 257 * 11; c->d_type = RSRGAN_D_DNN; c->d_layers = 4; c->d_cells = 1024; c->d_proj = 0;
    c->d_joint_off = 257 * 5; c->d_joint_dim = 257; c->clip_norm = 0.f; c->batch_size = 1024; c->max_frames = 1;
  } c->adam_beta1 = 0.9f; c->adam_beta2 = 0.999f; c->adam_eps = 1e-8f;
  c->ema_decay = 0.9999f; c->lrelu_alpha = 0.3f; c->forget_bias = 1.0f; c->cross_validation = 0; c->flags = 0;
  return RSRGAN_OK;
}

int rsrgan_create(const rsrgan_cfg* cfg, uint64_t seed, rsrgan_handle* out) {
  if (!cfg || !out) { set_error("null argument"); return RSRGAN_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device visible: librsrgan_hip needs an MI355X (gfx950); there is no CPU fallback");
    return RSRGAN_ERR_NO_DEVICE;
  }
  return guard("rsrgan_create", [&]() -> int {
    rsrgan_handle h = new (std::nothrow) rsrgan_handle_s();
    if (!h) { set_error("out of host memory"); return RSRGAN_ERR_INVALID; }
    int rc = RSRGAN_ERR_INVALID;
    try { rc = h->m.init(*cfg, seed); } catch (...) { h->m.destroy(); delete h; throw; }
    if (rc != RSRGAN_OK) { h->m.destroy(); delete h; return rc; }
    *out = h;
    return RSRGAN_OK;
  });
}

int rsrgan_destroy(rsrgan_handle h) {
  CHECK_H(h);
  return guard("rsrgan_destroy", [&]() -> int {
    (void)hipDeviceSynchronize();
    h->m.destroy();
    delete h;
    return RSRGAN_OK;
  });
}

int rsrgan_set_scalar(rsrgan_handle h, int32_t which, double v) {
  CHECK_H(h);
  Model& m = h->m;
  // the copies below are synchronous on the null stream; the step kernels that read these device scalars may still be queued on
  // ANY stream the caller passed to the step entry points (the Python host runs them on a non-blocking pool stream the null stream
  // does not order with), so drain the device first (scalars change once per iteration: train_gan_rnn_placeholder.py:63-64,525-533)
  if (hipDeviceSynchronize() != hipSuccess) { set_error("hipDeviceSynchronize failed"); return RSRGAN_ERR_HIP; }
  int idx = -1;
  switch (which) {
    case RSRGAN_G_LEARNING_RATE: idx = DYN_G_LR; break;
    case RSRGAN_D_LEARNING_RATE: idx = DYN_D_LR; break;
    case RSRGAN_MSE_LAMBDA: idx = DYN_LAMBDA; break;
    case RSRGAN_D_REAL: idx = DYN_D_REAL; break;
    case RSRGAN_D_FAKE: idx = DYN_D_FAKE; break;
    case RSRGAN_L2_SCALE: idx = DYN_L2; break;
    case RSRGAN_CLIP_NORM: idx = DYN_CLIP; break;
    case RSRGAN_ADAM_STEP:
    case RSRGAN_ADAM_STEP_D: {
      const int t = (int)v;
      int* dst = which == RSRGAN_ADAM_STEP ? m.adam_t_dev : m.adam_t_dev_d;
      if (hipMemcpy(dst, &t, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return RSRGAN_ERR_HIP; }
      m.scal[which] = t;
      return RSRGAN_OK;
    }
    default: set_error("unknown scalar %d", which); return RSRGAN_ERR_INVALID;
  }
  const float f = (float)v;      // the reference keeps these as tf.float32 variables
  if (hipMemcpy(m.dyn + idx, &f, sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return RSRGAN_ERR_HIP; }
  m.scal[which] = v;
  return RSRGAN_OK;
}

int rsrgan_get_scalar(rsrgan_handle h, int32_t which, double* v) {
  CHECK_H(h);
  if (which < 0 || which >= RSRGAN_SCALAR_COUNT_ || !v) { set_error("unknown scalar %d", which); return RSRGAN_ERR_INVALID; }
  *v = h->m.scal[which];
  return RSRGAN_OK;
}

static ParamSet* pset(rsrgan_handle h, int net) {
  if (net == RSRGAN_NET_G) return &h->m.G;
  if (net == RSRGAN_NET_D) return &h->m.D;
  return nullptr;
}

int rsrgan_num_tensors(rsrgan_handle h, int32_t net) {
  CHECK_H(h);
  ParamSet* p = pset(h, net);
  if (!p) { set_error("bad net"); return RSRGAN_ERR_INVALID; }
  return (int)p->t.size();
}

int rsrgan_tensor_info(rsrgan_handle h, int32_t net, int32_t idx, char* name, int32_t cap, int32_t* rows, int32_t* cols, int64_t* dense_offset) {
  CHECK_H(h);
  ParamSet* p = pset(h, net);
  if (!p || idx < 0 || idx >= (int)p->t.size()) { set_error("bad net/index"); return RSRGAN_ERR_INVALID; }
  const TensorDesc& t = p->t[idx];
  if (name && cap > 0) { std::strncpy(name, t.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (rows) *rows = t.is_vector ? t.cols : t.rows;
  if (cols) *cols = t.is_vector ? 0 : t.cols;       // cols == 0 marks a 1-D variable
  if (dense_offset) *dense_offset = t.dense_off;
  return RSRGAN_OK;
}

int64_t rsrgan_param_count(rsrgan_handle h, int32_t net) {
  if (!h) return RSRGAN_ERR_INVALID;
  ParamSet* p = pset(h, net);
  return p ? p->dense : (int64_t)RSRGAN_ERR_INVALID;
}

static float* which_buf(ParamSet* p, int what) {
  switch (what) {
    case 0: return p->w;
    case 1: return p->m;
    case 2: return p->v;
    case 3: return p->ema;
    case 4: return p->g;
  }
  return nullptr;
}

static int copy_params(rsrgan_handle h, int net, int what, float* dense, bool to_padded, void* stream) {
  CHECK_H(h);
  ParamSet* p = pset(h, net);
  if (!p || !dense) { set_error("bad net / null pointer"); return RSRGAN_ERR_INVALID; }
  float* buf = which_buf(p, what);
  if (!buf) { set_error("buffer %d not present for net %d", what, net); return RSRGAN_ERR_INVALID; }
  StreamScope sc(h->m, stream);
  hipStream_t s = sc.work;
  for (const TensorDesc& t : p->t) launch_pad_copy(dense + t.dense_off, buf + t.off, t.rows, t.cols, t.ld, to_padded, s);
  if (to_padded && what == 0) {
    h->m.refresh_transposes(net, s);
    if (net == RSRGAN_NET_G) h->m.g_fwd_valid = false;
  }
  if (hipGetLastError() != hipSuccess) { set_error("kernel launch failed in copy_params"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}

int rsrgan_get_params(rsrgan_handle h, int32_t net, int32_t what, float* dense, void* stream) {
  if (what < 0 || what > 3) { set_error("bad what"); return RSRGAN_ERR_INVALID; }
  return copy_params(h, net, what, dense, false, stream);
}
int rsrgan_set_params(rsrgan_handle h, int32_t net, int32_t what, const float* dense, void* stream) {
  if (what < 0 || what > 3) { set_error("bad what"); return RSRGAN_ERR_INVALID; }
  return copy_params(h, net, what, const_cast<float*>(dense), true, stream);
}
int rsrgan_get_grads(rsrgan_handle h, int32_t net, float* dense, void* stream) {
  return copy_params(h, net, 4, dense, false, stream);
}

int rsrgan_forward_g(rsrgan_handle h, const float* x, const int32_t* lengths, int32_t T, float* y, void* stream) {
  CHECK_H(h);
  return guard("rsrgan_forward_g", [&]() -> int {
    Model& m = h->m;
    if (!y) { set_error("null output"); return RSRGAN_ERR_INVALID; }
    StreamScope sc(m, stream);
    hipStream_t s = sc.work;
    int rc = m.prepare_batch(x, nullptr, lengths, T, s);
    if (rc) return rc;
    m.bn_eval_call = false;     // the graph of THIS model: is_training unless it was built with cross_validation
    m.g_forward(T, s);
    m.g_fwd_valid = false;      // labels were not packed: the stash is not a valid training forward
    launch_unpack_bm(m.y_tm, m.ldDout, y, m.B, T, m.Dout, s, m.Bt);      // (the caller's Bt rows of a padded model)
    if (hipGetLastError() != hipSuccess) { set_error("kernel launch failed in forward_g"); return RSRGAN_ERR_HIP; }
    return RSRGAN_OK;
  });
}

int rsrgan_d_backward(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                      const float* nr, const float* nf, float* out_losses, void* stream) {
  CHECK_H(h);
  return guard("rsrgan_d_backward", [&]() -> int {
    StreamScope sc(h->m, stream);
    return h->m.d_backward(x, labels, lengths, T, nr, nf, out_losses, true, sc.work);
  });
}
int rsrgan_g_backward(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                      const float* nf, float* out_losses, int32_t reuse, void* stream) {
  CHECK_H(h);
  return guard("rsrgan_g_backward", [&]() -> int {
    StreamScope sc(h->m, stream);
    return h->m.g_backward(x, labels, lengths, T, nf, out_losses, true, reuse != 0, sc.work);
  });
}
int rsrgan_apply(rsrgan_handle h, int32_t net, void* stream) {
  CHECK_H(h);
  return guard("rsrgan_apply", [&]() -> int {
    StreamScope sc(h->m, stream);
    return h->m.apply(net, sc.work);
  });
}

int rsrgan_d_step(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                  const float* nr, const float* nf, float* out_losses, int32_t train, void* stream) {
  CHECK_H(h);
  return guard("rsrgan_d_step", [&]() -> int {
    StreamScope sc(h->m, stream);
    struct Fused { Model& m; Fused(Model& m_, bool on) : m(m_) { m.fused_apply = on; } ~Fused() { m.fused_apply = false; m.apply_inlined = 0; } } fused(h->m, train != 0);
    int rc = h->m.d_backward(x, labels, lengths, T, nr, nf, out_losses, train != 0, sc.work);
    if (rc || !train) return rc;
    return h->m.apply(RSRGAN_NET_D, sc.work);
  });
}
int rsrgan_g_step(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                  const float* nf, float* out_losses, int32_t train, int32_t reuse, void* stream) {
  CHECK_H(h);
  return guard("rsrgan_g_step", [&]() -> int {
    StreamScope sc(h->m, stream);
    struct Fused { Model& m; Fused(Model& m_, bool on) : m(m_) { m.fused_apply = on; } ~Fused() { m.fused_apply = false; m.apply_inlined = 0; } } fused(h->m, train != 0);
    int rc = h->m.g_backward(x, labels, lengths, T, nf, out_losses, train != 0, reuse != 0, sc.work);
    if (rc || !train) return rc;
    return h->m.apply(RSRGAN_NET_G, sc.work);
  });
}

int rsrgan_grad_buffer(rsrgan_handle h, int32_t net, float** ptr, int64_t* count) {
  CHECK_H(h);
  ParamSet* p = pset(h, net);
  if (!p || !ptr || !count) { set_error("bad argument"); return RSRGAN_ERR_INVALID; }
  *ptr = p->g;
  *count = p->padded;
  return RSRGAN_OK;
}

int rsrgan_grad_bucket_count(rsrgan_handle h, int32_t net) {
  if (!h || (net != RSRGAN_NET_G && net != RSRGAN_NET_D)) return 0;
  return (int)h->m.gbk[net].size();
}
int rsrgan_grad_bucket_info(rsrgan_handle h, int32_t net, int32_t i, int64_t* offset, int64_t* count) {
  CHECK_H(h);
  Model* m = &h->m;
  if ((net != RSRGAN_NET_G && net != RSRGAN_NET_D) || i < 0 || i >= (int)m->gbk[net].size() || !offset || !count) {
    set_error("bad bucket index"); return RSRGAN_ERR_INVALID;
  }
  *offset = m->gbk[net][i].off;
  *count = m->gbk[net][i].count;
  return RSRGAN_OK;
}
int rsrgan_grad_bucket_wait(rsrgan_handle h, int32_t net, int32_t i, void* stream) {
  CHECK_H(h);
  Model* m = &h->m;
  if ((net != RSRGAN_NET_G && net != RSRGAN_NET_D) || i < 0 || i >= (int)m->gbk[net].size()) { set_error("bad bucket index"); return RSRGAN_ERR_INVALID; }
  if (hipStreamWaitEvent((hipStream_t)stream, m->gbk[net][i].ev, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}

int rsrgan_profile_begin(rsrgan_handle h) {
  CHECK_H(h);
  h->m.prof_on = true; h->m.prof_n = 0; h->m.prof_flops = 0.0; h->m.prof_gp_n = 0; h->m.prof_gp_flops = 0.0; h->m.prof_gb_n = 0; h->m.prof_gb_flops = 0.0; h->m.prof_fdt_n = 0;
  g_chain_launches = 0;
  return RSRGAN_OK;
}
int rsrgan_set_dropout(rsrgan_handle h, float keep_prob, uint64_t seed) {
  CHECK_H(h);
  if (!(keep_prob > 0.f && keep_prob <= 1.f)) { set_error("keep_prob=%g outside (0, 1]", (double)keep_prob); return RSRGAN_ERR_INVALID; }
  Model& m = h->m;
  if (keep_prob < 1.f && !m.g_dnn()) {
    for (const LstmLayer& L : m.gl)
      if (!L.has_proj) { set_error("DropoutWrapper is built for generator layers with a projection (num_proj) only"); return RSRGAN_ERR_INVALID; }
  }
  if (hipDeviceSynchronize() != hipSuccess) { set_error("hipDeviceSynchronize failed"); return RSRGAN_ERR_HIP; }
  m.keep_prob = keep_prob; m.drop_seed = seed; m.drop_run = 0;
  if (m.drop_ctr && hipMemset(m.drop_ctr, 0, 16) != hipSuccess) { set_error("hipMemset failed"); return RSRGAN_ERR_HIP; }
  m.drop_graphs();                                  // captured launch sequences carry the jobs' DropSpec
  m.g_fwd_valid = false;
  return RSRGAN_OK;
}

int rsrgan_device_status(rsrgan_handle h, int32_t* code) {
  CHECK_H(h);
  if (!code) { set_error("null output pointer"); return RSRGAN_ERR_INVALID; }
  Model& m = h->m;
  *code = 0;
  // the streams this handle has worked on (not the whole device: other handles and other tenants are none of this call's business;
  // see also gpersist.hip k_arm for what a device-wide synchronisation + blocking copy did to replayed fill nodes)
  if (m.ev_last_set && hipEventSynchronize(m.ev_last) != hipSuccess) { set_error("hipEventSynchronize failed"); return RSRGAN_ERR_HIP; }
  for (hipStream_t q : {m.main_s, m.side})
    if (q && hipStreamSynchronize(q) != hipSuccess) { set_error("hipStreamSynchronize failed"); return RSRGAN_ERR_HIP; }
  if (!m.dp_ctl && !m.gp_ctl) return RSRGAN_OK;
  // the control blocks of the persistent recurrences (dpersist.hip, gpersist.hip): the first failure wins; a generator failure is
  // reported as 0x10000 + workgroup
  unsigned* blocks[3] = {m.dp_ctl, m.gp_ctl, m.dp_ctl2};      // (dp_ctl2: the D(real) launches of RSRGAN_DPIPE; reported like the other discriminator launches)
  for (int k_ = 0; k_ < 3; ++k_) {
    const int k = k_ == 2 ? 0 : k_;
    if (!blocks[k_]) continue;
    unsigned ctl[DP_CTL_WORDS];
    if (hipMemcpy(ctl, blocks[k_], sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess) { set_error("hipMemcpy failed"); return RSRGAN_ERR_HIP; }
    if (*code == 0 && ctl[DP_CTL_ERR] != 0) *code = (int32_t)(ctl[DP_CTL_ERR] + (k ? 0x10000u : 0u));
    if (ctl[DP_CTL_ERR] != 0 || ctl[DP_CTL_DONE] != 0) {          // (an aborted launch can leave the arrival count behind)
      const unsigned z[2] = {0u, 0u};
      if (hipMemcpy(blocks[k_] + DP_CTL_DONE, z, sizeof(z), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return RSRGAN_ERR_HIP; }
      if (k == 1) m.gpersist_rearm();                               // (an aborted generator launch leaves ring slots written: arm them again)
      if (ctl[DP_CTL_ERR] != 0) m.persist_disable(k);                // (its workgroups were not all resident: this handle stops trying)
    }
  }
  return RSRGAN_OK;
}
int rsrgan_profile_launches(rsrgan_handle h, int64_t* n) {
  CHECK_H(h);
  if (!n) { set_error("null output pointer"); return RSRGAN_ERR_INVALID; }
  *n = g_chain_launches;
  return RSRGAN_OK;
}
int rsrgan_profile_read(rsrgan_handle h, int32_t* launches, double* total_us, double* alg_flops) {
  CHECK_H(h);
  Model& m = h->m;
  if (!launches || !total_us || !alg_flops) { set_error("null output pointer"); return RSRGAN_ERR_INVALID; }
  m.prof_on = false;
  double us = 0.0;
  for (int i = 0; i < m.prof_n; ++i) {
    if (hipEventSynchronize(m.prof_ev[2 * i + 1]) != hipSuccess) { set_error("hipEventSynchronize failed"); return RSRGAN_ERR_HIP; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, m.prof_ev[2 * i], m.prof_ev[2 * i + 1]) != hipSuccess) { set_error("hipEventElapsedTime failed"); return RSRGAN_ERR_HIP; }
    us += 1e3 * ms;
  }
  *launches = m.prof_n; *total_us = us; *alg_flops = m.prof_flops;
  return RSRGAN_OK;
}
int rsrgan_profile_read_kind(rsrgan_handle h, int32_t kind, int32_t* launches, double* total_us, double* alg_flops) {
  CHECK_H(h);
  if (kind == 0) return rsrgan_profile_read(h, launches, total_us, alg_flops);
  Model& m = h->m;
  if (kind == 3 && launches && total_us && alg_flops) {            // k_glstm_fwd_dt launches since profile_begin: a count only (not bracketed)
    *launches = m.prof_fdt_n; *total_us = 0.0; *alg_flops = 0.0;
    return RSRGAN_OK;
  }
  if ((kind != 1 && kind != 2) || !launches || !total_us || !alg_flops) { set_error("profile_read_kind: bad argument"); return RSRGAN_ERR_INVALID; }
  std::vector<hipEvent_t>& ev = kind == 1 ? m.prof_gp_ev : m.prof_gb_ev;
  const int n = kind == 1 ? m.prof_gp_n : m.prof_gb_n;
  double us = 0.0;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(ev[2 * i + 1]) != hipSuccess) { set_error("hipEventSynchronize failed"); return RSRGAN_ERR_HIP; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) { set_error("hipEventElapsedTime failed"); return RSRGAN_ERR_HIP; }
    us += 1e3 * ms;
  }
  *launches = n; *total_us = us; *alg_flops = kind == 1 ? m.prof_gp_flops : m.prof_gb_flops;
  return RSRGAN_OK;
}

int rsrgan_op_launch_floor(int32_t n, int32_t mode, double* us_per_launch, void* stream) {
  if (n <= 0 || !us_per_launch) { set_error("op_launch_floor: bad argument"); return RSRGAN_ERR_INVALID; }
  // captured once and replayed, like the step's segments (an eager chain is host-bound at 3-5 us per launch)
  static float* buf = nullptr;
  if (!buf && hipMalloc((void**)&buf, 2 * 65536 * 4 * sizeof(float)) != hipSuccess) { set_error("hipMalloc failed"); return RSRGAN_ERR_HIP; }
  hipStream_t s = nullptr;
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = RSRGAN_ERR_HIP;
  do {
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
    if (hipMemsetAsync(buf, 0, 2 * 65536 * 4 * sizeof(float), s) != hipSuccess) break;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) break;
    launch_floor_chain(buf, buf + 65536 * 4, n, mode, s);
    if (hipStreamEndCapture(s, &g) != hipSuccess) break;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) break;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
    if (hipGraphLaunch(ge, s) != hipSuccess) break;                 // warm-up replay
    if (hipEventRecord(e0, s) != hipSuccess || hipGraphLaunch(ge, s) != hipSuccess || hipEventRecord(e1, s) != hipSuccess) break;
    if (hipEventSynchronize(e1) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
    *us_per_launch = 1e3 * ms / n;
    rc = RSRGAN_OK;
  } while (0);
  if (rc != RSRGAN_OK) set_error("op_launch_floor: HIP call failed");
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (ge) (void)hipGraphExecDestroy(ge);
  if (g) (void)hipGraphDestroy(g);
  if (s) (void)hipStreamDestroy(s);
  (void)stream;
  return rc;
}

int rsrgan_op_gemm(const float* A, int32_t lda, int32_t a_kc, const float* B, int32_t ldb, int32_t b_kc, float* C, int32_t ldc,
                   int32_t M, int32_t N, int32_t K, const float* bias, int32_t act, float alpha, int32_t accumulate, void* stream) {
  if (!A || !B || !C || (lda & 3) || (ldb & 3)) { set_error("op_gemm: null pointer or leading dimension not a multiple of 4"); return RSRGAN_ERR_INVALID; }
  // unit-test entry: a private split-K workspace so the same code path as the model is exercised
  static float* ws = nullptr;
  static const size_t ws_floats = (size_t)16 << 20;
  if (!ws && hipMalloc((void**)&ws, ws_floats * sizeof(float)) != hipSuccess) ws = nullptr;
  launch_gemm(A, lda, a_kc != 0, B, ldb, b_kc != 0, C, ldc, M, N, K, bias, act, alpha, accumulate != 0, (hipStream_t)stream,
              ws, ws ? ws_floats : 0);
  if (hipGetLastError() != hipSuccess) { set_error("op_gemm launch failed"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}

}  // extern "C"
