// model.cpp -- parameter tables, buffers and the launch schedule of the GAN step.
// Reference graph: models/gan_rnn_placeholder.py:139-298 (build_model / build_model_single_gpu),
// generator models/lstm.py:41-129 | models/res_lstm_l.py:41-199, discriminator
// models/discriminator_lstm.py:24-110.
#include "model.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>

namespace rsr {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

#define HIPC(expr)                                                                   \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RSRGAN_ERR_HIP;                                                         \
    }                                                                                \
  } while (0)

int ParamSet::add(const std::string& name, int rows, int cols, bool is_vector) {
  TensorDesc d;
  d.name = name;
  d.rows = rows; d.cols = cols; d.ld = pad4(cols);
  d.off = padded; d.dense_off = dense;
  d.l2 = name.find("bias") == std::string::npos;
  d.is_vector = is_vector;
  padded += ((int64_t)rows * d.ld + 63) / 64 * 64;
  dense += (int64_t)rows * cols;
  t.push_back(d);
  return (int)t.size() - 1;
}

template <typename T>
T* Model::alloc(size_t n) {
  void* p = nullptr;
  if (n == 0) n = 1;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemset(p, 0, n * sizeof(T));
  allocs.push_back(p);
  return reinterpret_cast<T*>(p);
}

static void add_lstm(ParamSet& ps, std::vector<LstmLayer>& out, const std::string& prefix, int I, int H, int P) {
  LstmLayer L;
  L.I = I; L.H = H; L.P = P; L.ldI = pad4(I); L.ldP = pad4(P); L.ldH = pad4(H);
  L.tK = ps.add(prefix + "/kernel", I + P, 4 * H, false);
  L.tb = ps.add(prefix + "/bias", 1, 4 * H, true);
  L.twf = ps.add(prefix + "/w_f_diag", 1, H, true);
  L.twi = ps.add(prefix + "/w_i_diag", 1, H, true);
  L.two = ps.add(prefix + "/w_o_diag", 1, H, true);
  L.tWp = ps.add(prefix + "/projection/kernel", H, P, false);
  out.push_back(L);
}

static int build_chunks(Model& M, ParamSet& ps) {
  constexpr int CH = 4096;
  std::vector<int> tensor, off, len, tfirst, tcount, tl2;
  for (size_t i = 0; i < ps.t.size(); ++i) {
    const int64_t n = (int64_t)ps.t[i].rows * ps.t[i].ld;
    tfirst.push_back((int)tensor.size());
    int cnt = 0;
    for (int64_t o = 0; o < n; o += CH) {
      tensor.push_back((int)i);
      off.push_back((int)(ps.t[i].off + o));
      len.push_back((int)std::min<int64_t>(CH, n - o));
      ++cnt;
    }
    tcount.push_back(cnt);
    tl2.push_back(ps.t[i].l2 ? 1 : 0);
  }
  auto up = [&](const std::vector<int>& v) -> int* {
    int* d = M.alloc<int>(v.size());
    if (d) hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
    return d;
  };
  ps.ct.tensor = up(tensor); ps.ct.off = up(off); ps.ct.len = up(len);
  ps.ct.t_first = up(tfirst); ps.ct.t_count = up(tcount); ps.ct.t_l2 = up(tl2);
  ps.ct.n_chunks = (int)tensor.size();
  ps.ct.n_tensors = (int)ps.t.size();
  ps.partial = M.alloc<float>(tensor.size());
  return (ps.ct.tensor && ps.partial) ? 0 : -1;
}

static void alloc_stash(Model& M, LstmStash& S, const LstmLayer& L, int N, int T) {
  S.gates = M.alloc<float>((size_t)T * N * 4 * L.H);
  S.c = M.alloc<float>((size_t)(T + 1) * N * L.H);
  S.h = M.alloc<float>((size_t)T * N * L.ldH);
  S.mst = M.alloc<float>((size_t)(T + 1) * N * L.ldP);
  S.out = M.alloc<float>((size_t)T * N * L.ldP);
  S.dmt = M.alloc<float>((size_t)T * N * L.ldP);
  S.dc = M.alloc<float>((size_t)N * L.H);
  S.dmst = M.alloc<float>((size_t)N * L.ldP);
}

int Model::init(const rsrgan_cfg& c, uint64_t seed) {
  cfg = c;
  B = c.batch_size; Tmax = c.max_frames; Din = c.input_dim; Dout = c.output_dim;
  ldDin = pad4(Din); ldDout = pad4(Dout);
  if (B <= 0 || Tmax <= 0 || Din <= 0 || Dout <= 0 || c.g_layers <= 0 || c.d_layers <= 0 || c.g_cells <= 0 ||
      c.d_cells <= 0 || c.g_layers > MAXJ || c.d_layers > MAXJ) {
    set_error("invalid sizes in rsrgan_cfg");
    return RSRGAN_ERR_INVALID;
  }
  if (c.g_proj <= 0 || c.d_proj <= 0) {
    set_error("num_proj=None (proj <= 0) is not supported yet");
    return RSRGAN_ERR_INVALID;
  }
  if (c.d_type != RSRGAN_D_LSTM) { set_error("Unrecognized D type %d", c.d_type); return RSRGAN_ERR_INVALID; }
  const int P = c.g_proj, H = c.g_cells;
  // ---- variable tables in graph-construction order (gan_rnn_placeholder.py:301-317) ----
  if (c.g_type == RSRGAN_G_LSTM) {                                       // models/lstm.py:82-124
    g_fc_in_w = G.add("g_model/fully_connected/weights", Din, P, false);
    g_fc_in_b = G.add("g_model/fully_connected/biases", 1, P, true);
    for (int l = 0; l < c.g_layers; ++l)
      add_lstm(G, gl, "g_model/rnn/multi_rnn_cell/cell_" + std::to_string(l) + "/lstm_cell", P, H, P);
    g_fc_out_w = G.add("g_model/fully_connected_1/weights", P, Dout, false);
    g_fc_out_b = G.add("g_model/fully_connected_1/biases", 1, Dout, true);
  } else if (c.g_type == RSRGAN_G_RES_LSTM_L || c.g_type == RSRGAN_G_RES_LSTM_BASE) {   // models/res_lstm_l.py:101-194
    if (c.g_type == RSRGAN_G_RES_LSTM_L && P != Din) {
      set_error("res_lstm_l needs g_proj == input_dim (models/res_lstm_l.py:111)");
      return RSRGAN_ERR_INVALID;
    }
    int in = Din;
    for (int l = 0; l < c.g_layers; ++l) {
      add_lstm(G, gl, "g_model/lstm_cell_" + std::to_string(l + 1) + "/rnn/lstm_cell", in, H, P);
      in = P;
    }
    g_fc_out_w = G.add("g_model/forward_out/fully_connected/weights", P, Dout, false);
    g_fc_out_b = G.add("g_model/forward_out/fully_connected/biases", 1, Dout, true);
  } else {
    set_error("Unrecognized G type %d", c.g_type);                       // gan_rnn_placeholder.py:131-132
    return RSRGAN_ERR_INVALID;
  }
  {                                                                      // models/discriminator_lstm.py:70-104
    int in = Dout;
    for (int l = 0; l < c.d_layers; ++l) {
      add_lstm(D, dl, "d_model/rnn/multi_rnn_cell/cell_" + std::to_string(l) + "/lstm_cell", in, c.d_cells, c.d_proj);
      in = c.d_proj;
    }
    d_fc_w = D.add("d_model/fully_connected/weights", c.d_proj, 1, false);
    d_fc_b = D.add("d_model/fully_connected/biases", 1, 1, true);
  }
  // ---- device buffers ----
  const bool ema_on = c.ema_decay > 0.f;
  G.w = alloc<float>(G.padded); G.g = alloc<float>(G.padded); G.m = alloc<float>(G.padded); G.v = alloc<float>(G.padded);
  G.ema = ema_on ? alloc<float>(G.padded) : nullptr;
  D.w = alloc<float>(D.padded); D.g = alloc<float>(D.padded);
  D.ema = ema_on ? alloc<float>(D.padded) : nullptr;
  if (!G.w || !G.g || !G.m || !G.v || !D.w || !D.g) { set_error("hipMalloc failed (parameters)"); return RSRGAN_ERR_HIP; }
  if (build_chunks(*this, G) || build_chunks(*this, D)) { set_error("hipMalloc failed (chunk tables)"); return RSRGAN_ERR_HIP; }
  for (auto* layers : {&gl, &dl})
    for (auto& L : *layers) {
      L.KxT = alloc<float>((size_t)4 * L.H * L.ldI);
      L.KhT = alloc<float>((size_t)4 * L.H * L.ldP);
      L.WpT = alloc<float>((size_t)L.P * L.ldH);
    }
  const size_t TB = (size_t)Tmax * B;
  x_tm = alloc<float>(TB * ldDin); lab_tm = alloc<float>(TB * ldDout); y_tm = alloc<float>(TB * ldDout);
  const int ldP = pad4(P);
  g_st.resize(gl.size());
  for (size_t l = 0; l < gl.size(); ++l) alloc_stash(*this, g_st[l], gl[l], B, Tmax);
  g_ins.resize(gl.size() + 1);
  if (c.g_type == RSRGAN_G_LSTM) {
    g_h0 = alloc<float>(TB * ldP);
    g_ins[0] = g_h0;
    for (size_t l = 0; l < gl.size(); ++l) g_ins[l + 1] = g_st[l].out;
  } else {
    g_ins[0] = x_tm;
    for (size_t l = 0; l < gl.size(); ++l) {
      if (c.g_type == RSRGAN_G_RES_LSTM_L) {
        g_res.push_back(alloc<float>(TB * ldP));
        g_ins[l + 1] = g_res.back();
      } else {
        g_ins[l + 1] = g_st[l].out;
      }
    }
  }
  const int gmaxld = std::max(ldP, ldDin);
  g_dA = alloc<float>(TB * gmaxld); g_dB = alloc<float>(TB * gmaxld);
  const size_t TB2 = TB * 2;
  const int ldPd = pad4(c.d_proj);
  xd = alloc<float>(TB2 * ldDout); logits = alloc<float>(TB2 * 4); dlogits = alloc<float>(TB2 * 4);
  d_st.resize(dl.size());
  for (size_t l = 0; l < dl.size(); ++l) alloc_stash(*this, d_st[l], dl[l], 2 * B, Tmax);
  const int dmaxld = std::max(ldPd, ldDout);
  d_dA = alloc<float>(TB2 * dmaxld); d_dB = alloc<float>(TB2 * dmaxld);
  len_dev = alloc<int>(2 * B);
  dyn = alloc<float>(DYN_COUNT); adam_t_dev = alloc<int>(1);
  losses = alloc<float>(8); tmp3 = alloc<float>(4);
  size_t maxcols = 4 * (size_t)std::max(c.g_cells, c.d_cells);
  maxcols = std::max(maxcols, (size_t)std::max(ldP, ldDin));
  scratch = alloc<float>(std::max<size_t>(64 * maxcols, 1024));
  if (!scratch || !d_dB || !g_dB || !xd) { set_error("hipMalloc failed (activations)"); return RSRGAN_ERR_HIP; }

  // ---- initial values: xavier_initializer() uniform / zeros (models/lstm.py:86-87,93) ----
  std::mt19937_64 rng(seed);
  for (ParamSet* ps : {&G, &D}) {
    std::vector<float> host((size_t)ps->padded, 0.f);
    for (auto& t : ps->t) {
      if (!t.l2) continue;                                   // biases stay zero
      const double fan_in = t.is_vector ? t.cols : t.rows, fan_out = t.cols;
      const double lim = std::sqrt(6.0 / (fan_in + fan_out));
      std::uniform_real_distribution<double> u(-lim, lim);
      for (int r = 0; r < t.rows; ++r)
        for (int cc = 0; cc < t.cols; ++cc) host[(size_t)t.off + (size_t)r * t.ld + cc] = (float)u(rng);
    }
    HIPC(hipMemcpy(ps->w, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    if (ps->ema) HIPC(hipMemcpy(ps->ema, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  scal[RSRGAN_G_LEARNING_RATE] = 8e-5; scal[RSRGAN_D_LEARNING_RATE] = 1e-3; scal[RSRGAN_MSE_LAMBDA] = 10.0;
  scal[RSRGAN_D_REAL] = 1.0; scal[RSRGAN_D_FAKE] = 0.0; scal[RSRGAN_L2_SCALE] = c.l2_scale;
  scal[RSRGAN_CLIP_NORM] = c.clip_norm; scal[RSRGAN_ADAM_STEP] = 0;
  float hdyn[DYN_COUNT] = {0};
  hdyn[DYN_G_LR] = 8e-5f; hdyn[DYN_D_LR] = 1e-3f; hdyn[DYN_LAMBDA] = 10.f; hdyn[DYN_D_REAL] = 1.f; hdyn[DYN_D_FAKE] = 0.f;
  hdyn[DYN_L2] = c.l2_scale; hdyn[DYN_CLIP] = c.clip_norm; hdyn[DYN_B1] = c.adam_beta1; hdyn[DYN_B2] = c.adam_beta2;
  hdyn[DYN_EPS] = c.adam_eps; hdyn[DYN_EMA] = c.ema_decay;
  HIPC(hipMemcpy(dyn, hdyn, sizeof(hdyn), hipMemcpyHostToDevice));
  refresh_transposes(RSRGAN_NET_G, nullptr);
  refresh_transposes(RSRGAN_NET_D, nullptr);
  HIPC(hipDeviceSynchronize());
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

void Model::destroy() {
  for (void* p : allocs) (void)hipFree(p);
  allocs.clear();
}

void Model::refresh_transposes(int net, hipStream_t s) {
  const ParamSet& ps = net == RSRGAN_NET_G ? G : D;
  auto& layers = net == RSRGAN_NET_G ? gl : dl;
  for (auto& L : layers) {
    const float* K = ps.W(L.tK);
    const int H4 = 4 * L.H;
    launch_transpose(K, H4, L.KxT, L.ldI, L.I, H4, s);                       // [I][4H] -> [4H][ldI]
    launch_transpose(K + (size_t)L.I * H4, H4, L.KhT, L.ldP, L.P, H4, s);    // [P][4H] -> [4H][ldP]
    launch_transpose(ps.W(L.tWp), L.ldP, L.WpT, L.ldH, L.H, L.P, s);         // [H][ldP] -> [P][ldH]
  }
}

// ------------------------------------------------------------------------------------------
// one dynamic_rnn(LSTMCell) over [T][N][I]  (schedule v1: layer by layer, 2 launches per step)
// ------------------------------------------------------------------------------------------
void Model::lstm_forward(const ParamSet& ps, const LstmLayer& L, LstmStash& S, const float* in, int N, int T,
                         const float* res_in, float* res_out, hipStream_t s) {
  const int H = L.H, H4 = 4 * H;
  // x-part of every step in one time-batched GEMM: zx = in . K[0:I] + bias
  launch_gemm(in, L.ldI, true, ps.W(L.tK), H4, false, S.gates, H4, T * N, H4, L.I, ps.W(L.tb), 0, 0.f, false, s);
  const int nbr = (N + 15) / 16;
  for (int t = 0; t < T; ++t) {
    FwdGateJobs gj{};
    gj.n = 1; gj.forget_bias = cfg.forget_bias;
    FwdGateJob& a = gj.j[0];
    a.x = nullptr; a.KxT = L.KxT; a.ldx = L.ldI;
    a.m = S.mst + (size_t)t * N * L.ldP; a.KhT = L.KhT; a.ldm = L.ldP;
    a.zx = S.gates + (size_t)t * N * H4; a.bias = ps.W(L.tb);
    a.wf = ps.W(L.twf); a.wi = ps.W(L.twi); a.wo = ps.W(L.two);
    a.c_prev = S.c + (size_t)t * N * H; a.c_out = S.c + (size_t)(t + 1) * N * H;
    a.gates = S.gates + (size_t)t * N * H4;
    a.h = S.h + (size_t)t * N * L.ldH; a.ldh = L.ldH;
    a.len = len_dev; a.t = t; a.N = N; a.H = H;
    a.nblk_c = (H + 15) / 16; a.blk_base = 0;
    launch_fwd_gates(gj, a.nblk_c * nbr, s);

    FwdProjJobs pj{};
    pj.n = 1;
    FwdProjJob& p = pj.j[0];
    p.h = a.h; p.WpT = L.WpT; p.ldh = L.ldH;
    p.m_prev = a.m; p.m_out = S.mst + (size_t)(t + 1) * N * L.ldP; p.out = S.out + (size_t)t * N * L.ldP;
    p.res_in = res_in ? res_in + (size_t)t * N * L.ldP : nullptr;
    p.res_out = res_out ? res_out + (size_t)t * N * L.ldP : nullptr;
    p.len = len_dev; p.ldm = L.ldP; p.P = L.P; p.t = t; p.N = N;
    p.nblk_c = (L.P + 15) / 16; p.blk_base = 0;
    launch_fwd_proj(pj, p.nblk_c * nbr, s);
  }
}

void Model::lstm_backward(const ParamSet& ps, const LstmLayer& L, LstmStash& S, const float* in, int N, int T,
                          const float* dout, float* din, bool din_accumulate, bool want_wgrads, hipStream_t s) {
  const int H = L.H, H4 = 4 * H;
  const int nbr = (N + 15) / 16;
  (void)hipMemsetAsync(S.dc, 0, (size_t)N * H * sizeof(float), s);
  (void)hipMemsetAsync(S.dmst, 0, (size_t)N * L.ldP * sizeof(float), s);
  for (int t = T - 1; t >= 0; --t) {
    BwdAJobs aj{};
    aj.n = 1;
    BwdAJob& a = aj.j[0];
    a.dout = dout + (size_t)t * N * L.ldP; a.dmst = S.dmst; a.Wp = ps.W(L.tWp);
    a.dmt = S.dmt + (size_t)t * N * L.ldP;
    a.gates = S.gates + (size_t)t * N * H4;
    a.c_prev = S.c + (size_t)t * N * H; a.c_cur = S.c + (size_t)(t + 1) * N * H;
    a.wf = ps.W(L.twf); a.wi = ps.W(L.twi); a.wo = ps.W(L.two);
    a.dc = S.dc; a.len = len_dev; a.ldm = L.ldP; a.P = L.P; a.t = t; a.N = N; a.H = H;
    a.nblk_c = (H + 15) / 16; a.blk_base = 0;
    launch_bwd_a(aj, a.nblk_c * nbr, s);

    BwdBJobs bj{};
    bj.n = 1;
    BwdBJob& b = bj.j[0];
    b.dz = a.gates; b.K = ps.W(L.tK); b.dx = nullptr; b.dmst = S.dmst; b.len = len_dev;
    b.I = L.I; b.n_begin = L.I; b.n_end = L.I + L.P; b.lddx = L.ldI; b.ldm = L.ldP; b.t = t; b.N = N; b.H4 = H4;
    b.dx_accumulate = 0;
    b.nblk_c = (L.P + 15) / 16; b.blk_base = 0;
    launch_bwd_b(bj, b.nblk_c * nbr, s);
  }
  const int R = T * N;
  if (want_wgrads) {
    float* dK = ps.Gd(L.tK);
    // dK[0:I] = in^T . dZ ; dK[I:I+P] = m_{t-1}^T . dZ ; dWp = h^T . dm
    launch_gemm(in, L.ldI, false, S.gates, H4, false, dK, H4, L.I, H4, R, nullptr, 0, 0.f, false, s);
    launch_gemm(S.mst, L.ldP, false, S.gates, H4, false, dK + (size_t)L.I * H4, H4, L.P, H4, R, nullptr, 0, 0.f, false, s);
    launch_gemm(S.h, L.ldH, false, S.dmt, L.ldP, false, ps.Gd(L.tWp), L.ldP, H, L.P, R, nullptr, 0, 0.f, false, s);
    launch_colsum(S.gates, H4, nullptr, 0, ps.Gd(L.tb), R, H4, scratch, s);
    // peepholes: dw_i = sum dai*c_{t-1}; dw_f = sum daf*c_{t-1}; dw_o = sum dao*c_t
    launch_colsum(S.gates, H4, S.c, H, ps.Gd(L.twi), R, H, scratch, s);
    launch_colsum(S.gates + 2 * H, H4, S.c, H, ps.Gd(L.twf), R, H, scratch, s);
    launch_colsum(S.gates + 3 * H, H4, S.c + (size_t)N * H, H, ps.Gd(L.two), R, H, scratch, s);
  }
  if (din)   // din (+)= dZ . K[0:I]^T
    launch_gemm(S.gates, H4, true, ps.W(L.tK), H4, true, din, L.ldI, R, L.I, H4, nullptr, 0, 0.f, din_accumulate, s);
}

// ------------------------------------------------------------------------------------------
int Model::prepare_batch(const float* x, const float* labels, const int32_t* lengths, int T, hipStream_t s) {
  if (T <= 0 || T > Tmax) { set_error("T=%d outside (0, max_frames=%d]", T, Tmax); return RSRGAN_ERR_INVALID; }
  if (!x || !lengths) { set_error("null input pointer"); return RSRGAN_ERR_INVALID; }
  launch_pack_tm(x, x_tm, B, T, Din, ldDin, s);
  if (labels) launch_pack_tm(labels, lab_tm, B, T, Dout, ldDout, s);
  HIPC(hipMemcpyAsync(len_dev, lengths, B * sizeof(int), hipMemcpyDeviceToDevice, s));
  HIPC(hipMemcpyAsync(len_dev + B, lengths, B * sizeof(int), hipMemcpyDeviceToDevice, s));
  cur_T = T;
  g_fwd_valid = false;
  return RSRGAN_OK;
}

void Model::g_forward(int T, hipStream_t s) {
  const int R = T * B;
  const int P = cfg.g_proj, ldP = pad4(P);
  if (cfg.g_type == RSRGAN_G_LSTM) {
    // h = leakyrelu(x.W + b)  (models/lstm.py:82-87)
    launch_gemm(x_tm, ldDin, true, G.W(g_fc_in_w), ldP, false, g_h0, ldP, R, P, Din, G.W(g_fc_in_b), 1, cfg.lrelu_alpha, false, s);
    for (size_t l = 0; l < gl.size(); ++l) lstm_forward(G, gl[l], g_st[l], g_ins[l], B, T, nullptr, nullptr, s);
  } else {
    const bool res = cfg.g_type == RSRGAN_G_RES_LSTM_L;
    for (size_t l = 0; l < gl.size(); ++l)      // inputs_{l+1} = outputs_l + inputs_l (models/res_lstm_l.py:111,121,131,190)
      lstm_forward(G, gl[l], g_st[l], g_ins[l], B, T, res ? g_ins[l] : nullptr, res ? g_res[l] : nullptr, s);
  }
  // y = outputs.W + b (models/lstm.py:121-124)
  launch_gemm(g_ins[gl.size()], ldP, true, G.W(g_fc_out_w), ldDout, false, y_tm, ldDout, R, Dout, P, G.W(g_fc_out_b), 0, 0.f, false, s);
  g_fwd_valid = true;
}

void Model::d_forward(int N, int T, hipStream_t s) {
  const float* in = xd;
  for (size_t l = 0; l < dl.size(); ++l) {
    lstm_forward(D, dl[l], d_st[l], in, N, T, nullptr, nullptr, s);
    in = d_st[l].out;
  }
  const int ldPd = pad4(cfg.d_proj);
  launch_gemm(in, ldPd, true, D.W(d_fc_w), 4, false, logits, 4, T * N, 1, cfg.d_proj, D.W(d_fc_b), 0, 0.f, false, s);
}

// leaves in last_dx0 (d_dA or d_dB) the gradient w.r.t. the discriminator input when need_dx0
void Model::d_backward_pass(int N, int T, bool want_wgrads, bool need_dx0, const float* dlog, hipStream_t s) {
  const int R = T * N;
  const int ldPd = pad4(cfg.d_proj);
  const size_t Ld = dl.size();
  const float* top = d_st[Ld - 1].out;
  if (want_wgrads) {
    launch_gemm(top, ldPd, false, dlog, 4, false, D.Gd(d_fc_w), 4, cfg.d_proj, 1, R, nullptr, 0, 0.f, false, s);
    launch_colsum(dlog, 4, nullptr, 0, D.Gd(d_fc_b), R, 1, scratch, s);
  }
  float* cur = d_dB;
  float* other = d_dA;
  // d(outputs) = dlogits . W^T
  launch_gemm(dlog, 4, true, D.W(d_fc_w), 4, true, cur, ldPd, R, cfg.d_proj, 1, nullptr, 0, 0.f, false, s);
  for (int l = (int)Ld - 1; l >= 0; --l) {
    const float* in = l == 0 ? xd : d_st[l - 1].out;
    const bool need = l > 0 || need_dx0;
    lstm_backward(D, dl[l], d_st[l], in, N, T, cur, need ? other : nullptr, false, want_wgrads, s);
    std::swap(cur, other);
  }
  last_dx0 = cur;
}

void Model::g_backward_pass(int T, float* dy, hipStream_t s) {
  const int R = T * B;
  const int P = cfg.g_proj, ldP = pad4(P);
  const size_t Lg = gl.size();
  // output FC: dW = in^T . dy ; db = colsum(dy) ; d(in) = dy . W^T
  launch_gemm(g_ins[Lg], ldP, false, dy, ldDout, false, G.Gd(g_fc_out_w), ldDout, P, Dout, R, nullptr, 0, 0.f, false, s);
  launch_colsum(dy, ldDout, nullptr, 0, G.Gd(g_fc_out_b), R, Dout, scratch, s);
  float* cur = g_dA;
  float* other = g_dB;
  launch_gemm(dy, ldDout, true, G.W(g_fc_out_w), ldDout, true, cur, ldP, R, P, Dout, nullptr, 0, 0.f, false, s);
  if (cfg.g_type == RSRGAN_G_LSTM) {
    for (int l = (int)Lg - 1; l >= 0; --l) {
      lstm_backward(G, gl[l], g_st[l], g_ins[l], B, T, cur, other, false, true, s);
      std::swap(cur, other);
    }
    // through leakyrelu and the input FC (models/lstm.py:82-87)
    launch_lrelu_bwd(g_h0, cur, (size_t)R, P, ldP, cfg.lrelu_alpha, s);
    launch_gemm(x_tm, ldDin, false, cur, ldP, false, G.Gd(g_fc_in_w), ldP, Din, P, R, nullptr, 0, 0.f, false, s);
    launch_colsum(cur, ldP, nullptr, 0, G.Gd(g_fc_in_b), R, P, scratch, s);
  } else if (cfg.g_type == RSRGAN_G_RES_LSTM_L) {
    // d(inputs_l) = dx_l + d(inputs_{l+1}): accumulate in place
    for (int l = (int)Lg - 1; l >= 0; --l)
      lstm_backward(G, gl[l], g_st[l], g_ins[l], B, T, cur, l > 0 ? cur : nullptr, true, true, s);
  } else {
    for (int l = (int)Lg - 1; l >= 0; --l) {
      lstm_backward(G, gl[l], g_st[l], g_ins[l], B, T, cur, l > 0 ? other : nullptr, false, true, s);
      std::swap(cur, other);
    }
  }
}

int Model::d_backward(const float* x, const float* labels, const int32_t* lengths, int T, const float* nr, const float* nf,
                      float* out_losses, bool want_grads, hipStream_t s) {
  if (!labels) { set_error("labels required"); return RSRGAN_ERR_INVALID; }
  int rc = prepare_batch(x, labels, lengths, T, s);
  if (rc) return rc;
  g_forward(T, s);
  launch_build_d_input(lab_tm, y_tm, nr, nf, xd, B, T, Dout, ldDout, true, s);
  d_forward(2 * B, T, s);
  launch_lsgan(logits, 4, want_grads ? dlogits : nullptr, T, 2 * B, B, dyn + DYN_D_REAL, dyn + DYN_D_FAKE, losses, s);
  if (want_grads) {
    d_backward_pass(2 * B, T, true, false, dlogits, s);
    d_grads_ready = true;
  }
  if (out_losses) launch_copy_f(losses, out_losses, 3, s);
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

int Model::g_backward(const float* x, const float* labels, const int32_t* lengths, int T, const float* nf,
                      float* out_losses, bool want_grads, bool reuse, hipStream_t s) {
  if (!labels) { set_error("labels required"); return RSRGAN_ERR_INVALID; }
  if (reuse) {
    if (!g_fwd_valid || T != cur_T) { set_error("reuse_g_forward without a valid generator forward"); return RSRGAN_ERR_STATE; }
  } else {
    int rc = prepare_batch(x, labels, lengths, T, s);
    if (rc) return rc;
    g_forward(T, s);
  }
  launch_build_d_input(lab_tm, y_tm, nullptr, nf, xd, B, T, Dout, ldDout, false, s);
  d_forward(B, T, s);
  // g_adv = mean((D(G(x)) - d_real)^2)  (gan_rnn_placeholder.py:246): all rows "fake", target d_real
  launch_lsgan(logits, 4, want_grads ? dlogits : nullptr, T, B, 0, dyn + DYN_D_REAL, dyn + DYN_D_REAL, tmp3, s);
  launch_copy_f(tmp3 + 1, losses + 3, 1, s);
  const bool l2_on = !cfg.cross_validation && scal[RSRGAN_L2_SCALE] > 0.0;
  if (want_grads) {
    d_backward_pass(B, T, false, true, dlogits, s);
    float* dy = last_dx0;                      // d g_adv / d y
    launch_mse(y_tm, lab_tm, ldDout, dy, T * B, Dout, dyn + DYN_LAMBDA, true, losses + 4, scratch, s);
    g_backward_pass(T, dy, s);
    if (l2_on) {
      launch_l2(G.w, G.g, G.ct, dyn + DYN_L2, G.partial, s);
      launch_l2_total(G.partial, G.ct.n_chunks, dyn + DYN_L2, losses + 5, s);
    }
    g_grads_ready = true;
  } else {
    launch_mse(y_tm, lab_tm, ldDout, nullptr, T * B, Dout, dyn + DYN_LAMBDA, false, losses + 4, scratch, s);
  }
  if (!(want_grads && l2_on)) HIPC(hipMemsetAsync(losses + 5, 0, sizeof(float), s));
  launch_g_total(losses + 3, dyn + DYN_LAMBDA, s);
  if (out_losses) launch_copy_f(losses + 3, out_losses, 4, s);
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

int Model::apply(int net, hipStream_t s) {
  if (net == RSRGAN_NET_D) {
    if (!d_grads_ready) { set_error("apply(D) without gradients"); return RSRGAN_ERR_STATE; }
    launch_sumsq(D.g, D.ct, D.partial, s);
    launch_apply_sgd(D.w, D.g, D.ema, D.ct, D.partial, dyn, s);
    refresh_transposes(RSRGAN_NET_D, s);
    d_grads_ready = false;
  } else if (net == RSRGAN_NET_G) {
    if (!g_grads_ready) { set_error("apply(G) without gradients"); return RSRGAN_ERR_STATE; }
    launch_sumsq(G.g, G.ct, G.partial, s);
    launch_adam_tick(dyn, adam_t_dev, (double)cfg.adam_beta1, (double)cfg.adam_beta2, s);
    scal[RSRGAN_ADAM_STEP] += 1;
    launch_apply_adam(G.w, G.g, G.m, G.v, G.ema, G.ct, G.partial, dyn, s);
    refresh_transposes(RSRGAN_NET_G, s);
    g_grads_ready = false;
    g_fwd_valid = false;
  } else {
    set_error("bad net %d", net);
    return RSRGAN_ERR_INVALID;
  }
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

}  // namespace rsr
