// segan_capi.cpp -- extern "C" entry points of the SEGAN-style conv G/D (include/rsrgan.h, rsrgan_segan_*).
#include <cstring>
#include <exception>
#include <new>

#include "segan.h"

using namespace rsr;

struct rsrgan_segan_handle_s { SeganModel m; };

#define CHECK_S(h) \
  if (!(h)) { set_error("null handle"); return RSRGAN_ERR_INVALID; }

template <class F>
static int sguard(const char* what, F&& body) {
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
static ParamSet* spset(rsrgan_segan_handle h, int net) { return net == RSRGAN_NET_G ? &h->m.G : (net == RSRGAN_NET_D ? &h->m.D : nullptr); }

extern "C" {

int rsrgan_segan_default_cfg(rsrgan_segan_cfg* c) {
  if (!c) { set_error("null cfg"); return RSRGAN_ERR_INVALID; }
  std::memset(c, 0, sizeof(*c));
  static const int depths[11] = {16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024};     // models/segan.py:89,91
  c->batch_size = 32; c->input_len = 257 * 11; c->output_dim = 40; c->n_layers = 11;     // run_segan.sh:99-102
  for (int i = 0; i < 11; ++i) { c->g_depths[i] = depths[i]; c->d_depths[i] = depths[i]; }
  c->g_kwidth = 20; c->d_kwidth = 31; c->g_prelu = 1;
  c->lrelu_alpha = 0.3f; c->vbn_eps = 1e-5f; c->rms_decay = 0.9f; c->rms_eps = 1e-10f;
  return RSRGAN_OK;
}

int rsrgan_segan_create(const rsrgan_segan_cfg* cfg, uint64_t seed, rsrgan_segan_handle* out) {
  if (!cfg || !out) { set_error("null argument"); return RSRGAN_ERR_INVALID; }
  *out = nullptr;
  return sguard("rsrgan_segan_create", [&]() -> int {
    rsrgan_segan_handle h = new rsrgan_segan_handle_s();
    const int rc = h->m.init(*cfg, seed);
    if (rc != RSRGAN_OK) { h->m.destroy(); delete h; return rc; }
    *out = h;
    return RSRGAN_OK;
  });
}
int rsrgan_segan_destroy(rsrgan_segan_handle h) {
  CHECK_S(h);
  (void)hipDeviceSynchronize();
  h->m.destroy();
  delete h;
  return RSRGAN_OK;
}
int rsrgan_segan_set_scalar(rsrgan_segan_handle h, int32_t which, double v) {
  CHECK_S(h);
  if (which < 0 || which > RSRGAN_SEGAN_L1_LAMBDA) { set_error("bad scalar %d", which); return RSRGAN_ERR_INVALID; }
  h->m.scal[which] = v;
  const float f = (float)v;
  (void)hipDeviceSynchronize();                          // (sess.run(tf.assign(...)) is a synchronous run of its own)
  if (hipMemcpy(h->m.dyn + which, &f, sizeof f, hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}
int rsrgan_segan_num_tensors(rsrgan_segan_handle h, int32_t net) {
  if (!h) return RSRGAN_ERR_INVALID;
  ParamSet* p = spset(h, net);
  return p ? (int)p->t.size() : (int)RSRGAN_ERR_INVALID;
}
int rsrgan_segan_tensor_info(rsrgan_segan_handle h, int32_t net, int32_t idx, char* name, int32_t cap, int32_t* rows, int32_t* cols, int64_t* dense_offset) {
  CHECK_S(h);
  ParamSet* p = spset(h, net);
  if (!p || idx < 0 || idx >= (int)p->t.size()) { set_error("bad net/index"); return RSRGAN_ERR_INVALID; }
  const TensorDesc& t = p->t[idx];
  if (name && cap > 0) { std::strncpy(name, t.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (rows) *rows = t.is_vector ? t.cols : t.rows;
  if (cols) *cols = t.is_vector ? 0 : t.cols;
  if (dense_offset) *dense_offset = t.dense_off;
  return RSRGAN_OK;
}
int64_t rsrgan_segan_param_count(rsrgan_segan_handle h, int32_t net) {
  if (!h) return RSRGAN_ERR_INVALID;
  ParamSet* p = spset(h, net);
  return p ? p->dense : (int64_t)RSRGAN_ERR_INVALID;
}
static int scopy(rsrgan_segan_handle h, int net, int what, float* dense, bool to_padded, void* stream) {
  CHECK_S(h);
  ParamSet* p = spset(h, net);
  if (!p || !dense || what < 0 || what > 2) { set_error("bad net / what / null pointer"); return RSRGAN_ERR_INVALID; }
  float* buf = what == 0 ? p->w : (what == 1 ? p->v : p->g);
  hipStream_t s = (hipStream_t)stream;
  for (const TensorDesc& t : p->t) launch_pad_copy(dense + t.dense_off, buf + t.off, t.rows, t.cols, t.ld, to_padded, s);
  if (to_padded && what == 0) h->m.refresh_weights(net, s);
  if (hipGetLastError() != hipSuccess) { set_error("kernel launch failed in copy"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}
int rsrgan_segan_get_params(rsrgan_segan_handle h, int32_t net, int32_t what, float* dense, void* stream) { return scopy(h, net, what, dense, false, stream); }
int rsrgan_segan_set_params(rsrgan_segan_handle h, int32_t net, int32_t what, const float* dense, void* stream) {
  return scopy(h, net, what, const_cast<float*>(dense), true, stream);
}
int rsrgan_segan_forward_g(rsrgan_segan_handle h, const float* x, const float* z, float* y, void* stream) {
  CHECK_S(h);
  return sguard("rsrgan_segan_forward_g", [&]() -> int {
    if (!x || !z || !y) { set_error("null pointer"); return RSRGAN_ERR_INVALID; }
    SeganModel& m = h->m;
    m.g_forward(x, z, (hipStream_t)stream);
    launch_copy_cols(m.Gy, pad4(m.U), 0, y, m.U, 0, m.U, m.B, false, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) { set_error("kernel launch failed in segan forward"); return RSRGAN_ERR_HIP; }
    return RSRGAN_OK;
  });
}
int rsrgan_segan_d_backward(rsrgan_segan_handle h, const float* x, const float* labels, const float* z, const float* n_ref, const float* n_real,
                            const float* n_fake, float* out_losses, int32_t train, void* stream) {
  CHECK_S(h);
  return sguard("rsrgan_segan_d_backward", [&]() -> int { return h->m.d_run(x, labels, z, n_ref, n_real, n_fake, out_losses, train != 0, (hipStream_t)stream); });
}
int rsrgan_segan_g_backward(rsrgan_segan_handle h, const float* x, const float* labels, const float* z, const float* n_ref, const float* n_fake,
                            float* out_losses, int32_t train, void* stream) {
  CHECK_S(h);
  return sguard("rsrgan_segan_g_backward", [&]() -> int { return h->m.g_run(x, labels, z, n_ref, n_fake, out_losses, train != 0, (hipStream_t)stream); });
}
int rsrgan_segan_grad_buffer(rsrgan_segan_handle h, int32_t net, float** ptr, int64_t* count) {
  CHECK_S(h);
  ParamSet* p = spset(h, net);
  if (!p || !ptr || !count) { set_error("bad net / null pointer"); return RSRGAN_ERR_INVALID; }
  *ptr = p->g; *count = p->padded;
  return RSRGAN_OK;
}
int rsrgan_segan_apply(rsrgan_segan_handle h, int32_t net, void* stream) {
  CHECK_S(h);
  if (net != RSRGAN_NET_G && net != RSRGAN_NET_D) { set_error("bad net"); return RSRGAN_ERR_INVALID; }
  return sguard("rsrgan_segan_apply", [&]() -> int { return h->m.apply(net, (hipStream_t)stream); });
}

}  // extern "C"
