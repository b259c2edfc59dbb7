// capi.cpp -- extern "C" entry points declared in include/rsrgan.h.
#include <cstring>
#include <new>
#include <vector>

#include "model.h"

using namespace rsr;

struct rsrgan_handle_s { Model m; };

#define CHECK_H(h)                                                    \
  if (!(h)) { set_error("null handle"); return RSRGAN_ERR_INVALID; }

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
  rsrgan_handle h = new (std::nothrow) rsrgan_handle_s();
  if (!h) { set_error("out of host memory"); return RSRGAN_ERR_INVALID; }
  int rc = h->m.init(*cfg, seed);
  if (rc != RSRGAN_OK) { h->m.destroy(); delete h; return rc; }
  *out = h;
  return RSRGAN_OK;
}

int rsrgan_destroy(rsrgan_handle h) {
  CHECK_H(h);
  hipDeviceSynchronize();
  h->m.destroy();
  delete h;
  return RSRGAN_OK;
}

int rsrgan_set_scalar(rsrgan_handle h, int32_t which, double v) {
  CHECK_H(h);
  Model& m = h->m;
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
  hipStream_t s = (hipStream_t)stream;
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
  Model& m = h->m;
  hipStream_t s = (hipStream_t)stream;
  if (!y) { set_error("null output"); return RSRGAN_ERR_INVALID; }
  int rc = m.prepare_batch(x, nullptr, lengths, T, s);
  if (rc) return rc;
  m.g_forward(T, s);
  m.g_fwd_valid = false;      // labels were not packed: the stash is not a valid training forward
  launch_unpack_bm(m.y_tm, m.ldDout, y, m.B, T, m.Dout, s);
  if (hipGetLastError() != hipSuccess) { set_error("kernel launch failed in forward_g"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}

int rsrgan_d_backward(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                      const float* nr, const float* nf, float* out_losses, void* stream) {
  CHECK_H(h);
  return h->m.d_backward(x, labels, lengths, T, nr, nf, out_losses, true, (hipStream_t)stream);
}
int rsrgan_g_backward(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                      const float* nf, float* out_losses, int32_t reuse, void* stream) {
  CHECK_H(h);
  return h->m.g_backward(x, labels, lengths, T, nf, out_losses, true, reuse != 0, (hipStream_t)stream);
}
int rsrgan_apply(rsrgan_handle h, int32_t net, void* stream) {
  CHECK_H(h);
  return h->m.apply(net, (hipStream_t)stream);
}

int rsrgan_d_step(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                  const float* nr, const float* nf, float* out_losses, int32_t train, void* stream) {
  CHECK_H(h);
  int rc = h->m.d_backward(x, labels, lengths, T, nr, nf, out_losses, train != 0, (hipStream_t)stream);
  if (rc || !train) return rc;
  return h->m.apply(RSRGAN_NET_D, (hipStream_t)stream);
}
int rsrgan_g_step(rsrgan_handle h, const float* x, const float* labels, const int32_t* lengths, int32_t T,
                  const float* nf, float* out_losses, int32_t train, int32_t reuse, void* stream) {
  CHECK_H(h);
  int rc = h->m.g_backward(x, labels, lengths, T, nf, out_losses, train != 0, reuse != 0, (hipStream_t)stream);
  if (rc || !train) return rc;
  return h->m.apply(RSRGAN_NET_G, (hipStream_t)stream);
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
  h->m.prof_on = true; h->m.prof_n = 0; h->m.prof_flops = 0.0;
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

// Micro-benchmark of one wavefront launch (within-probe A/B of kernel variants; not part of the
// drop-in surface).  kind 3 = backward phase B with `layers` jobs of (N rows, H cells, I inputs, P
// proj) on random data; returns the mean microseconds per launch over `reps` launches.
int rsrgan_microbench(int32_t kind, int32_t variant, int32_t N, int32_t H, int32_t I, int32_t P, int32_t layers,
                      int32_t reps, float* out_us) {
  if (kind == 5 && out_us) {     // flag exchange in groups: N = workgroups, H = group size, I = floats written per member, reps = iterations
    const int rc = flagx_microbench(variant, N, reps, H, I, out_us);
    if (rc == -2) { set_error("microbench: flag spin limit hit"); return RSRGAN_ERR_STATE; }
    if (rc) { set_error("microbench: hipMalloc"); return RSRGAN_ERR_HIP; }
    return RSRGAN_OK;
  }
  if (kind == 4 && out_us) {     // grid-barrier cost: N = workgroups, reps = barriers per launch, I = floats written per WG, P = floats read per WG
    const int rc = gridbar_microbench(variant, N, reps, I, P, out_us);
    if (rc == -2) { set_error("microbench: grid barrier spin limit hit (workgroups not co-resident?)"); return RSRGAN_ERR_STATE; }
    if (rc) { set_error("microbench: hipMalloc"); return RSRGAN_ERR_HIP; }
    return RSRGAN_OK;
  }
  if (kind != 3 || layers < 1 || layers > MAXJ || !out_us) { set_error("microbench: unsupported"); return RSRGAN_ERR_INVALID; }
  const int H4 = 4 * H, ldI = pad4(I), ldP = pad4(P);
  std::vector<void*> bufs;
  auto dal = [&](size_t n, float v) { float* p = nullptr; if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) return (float*)nullptr; launch_fill(p, n, v, nullptr); bufs.push_back(p); return p; };
  int* len = nullptr;
  if (hipMalloc((void**)&len, N * sizeof(int)) != hipSuccess) { set_error("hipMalloc"); return RSRGAN_ERR_HIP; }
  std::vector<int> hl(N, 1 << 20);
  (void)hipMemcpy(len, hl.data(), N * sizeof(int), hipMemcpyHostToDevice);
  BwdBJobs bj{};
  int bb = 0;
  for (int l = 0; l < layers; ++l) {
    BwdBJob& b = bj.j[bj.n++];
    b.dz = dal((size_t)N * H4, 0.01f * (l + 1)); b.K = dal((size_t)(I + P) * H4, 0.003f);
    b.dx = dal((size_t)N * ldI, 0.f); b.dmst = dal((size_t)N * ldP, 0.f); b.len = len;
    b.I = I; b.n_begin = 0; b.n_end = I + P; b.lddx = ldI; b.ldm = ldP; b.t = 0; b.N = N; b.H4 = H4; b.dx_accumulate = 0;
    b.nblk_c = (I + P + 15) / 16; b.blk_base = bb; bb += job_blocks(b.nblk_c, N);
    if (!b.dz || !b.K || !b.dx || !b.dmst) { set_error("hipMalloc"); return RSRGAN_ERR_HIP; }
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_bwd_b_variant(bj, bb, variant, nullptr);
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps; ++i) launch_bwd_b_variant(bj, bb, variant, nullptr);
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *out_us = ms * 1000.f / reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  for (void* p : bufs) (void)hipFree(p);
  (void)hipFree(len);
  if (hipGetLastError() != hipSuccess) { set_error("microbench launch failed"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}

}  // extern "C"
