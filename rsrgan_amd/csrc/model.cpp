// model.cpp -- parameter tables, buffers and the launch schedule of the GAN step.
// Reference graph: models/gan_rnn_placeholder.py:139-298 (build_model / build_model_single_gpu),
// generator models/lstm.py:41-129 | models/res_lstm_l.py:41-199, discriminator
// models/discriminator_lstm.py:24-110.
#include "model.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
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

// ------------------------------------------------------------------------------------------ hipGraph segments
static inline uint64_t seg_key(int seg, int T, unsigned bits) { return ((uint64_t)seg << 48) | ((uint64_t)(unsigned)T << 16) | (bits & 0xffffu); }
enum { SEG_D = 1, SEG_G_MAIN = 2, SEG_G_FCIN = 3, SEG_G_LAYER0 = 4 /* .. + MAXJ */, SEG_G_L2 = 20, SEG_G_TAIL = 21, SEG_APPLY_D = 22, SEG_APPLY_G = 23, SEG_G_BATCH = 24 };

template <class F>
void Model::run_seg(uint64_t key, hipStream_t s, F&& body) {
  if (!graphs_on() || s == nullptr) { body(); return; }
  if (graphs.size() > 96) drop_graphs();                   // many distinct T (length-bucketed training): start over
  GraphSlot& g = graphs[key];
  if (g.exec) { if (hipGraphLaunch(g.exec, s) != hipSuccess) { (void)hipGetLastError(); g.exec = nullptr; g.uses = -1; body(); } return; }
  if (g.uses < 0) { body(); return; }                       // capture failed once: eager from then on
  if (g.uses++ == 0) { body(); return; }                    // first use: eager (also runs every lazy one-time setup)
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); g.uses = -1; body(); return; }
  body();
  hipGraph_t gr = nullptr;
  if (hipStreamEndCapture(s, &gr) != hipSuccess || !gr) {
    (void)hipGetLastError(); g.uses = -1;
    if (getenv("RSRGAN_GRAPH_DEBUG")) fprintf(stderr, "rsrgan: capture of segment %llx failed, eager from here on\n", (unsigned long long)key);
    body(); return;
  }
  hipGraphExec_t ex = nullptr;
  if (hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) != hipSuccess || !ex) { (void)hipGetLastError(); (void)hipGraphDestroy(gr); g.uses = -1; body(); return; }
  (void)hipGraphDestroy(gr);
  g.exec = ex;
  if (hipGraphLaunch(g.exec, s) != hipSuccess) { (void)hipGetLastError(); (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; g.uses = -1; body(); }
}
void Model::drop_graphs() {
  for (auto& kv : graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  graphs.clear();
}
hipStream_t Model::enter(hipStream_t caller) {
  if (caller != nullptr || !main_s) return caller;
  (void)hipEventRecord(ev_in, caller);
  (void)hipStreamWaitEvent(main_s, ev_in, 0);
  return main_s;
}
void Model::leave(hipStream_t caller, hipStream_t work) {
  if (work == caller) return;
  (void)hipEventRecord(ev_out, work);
  (void)hipStreamWaitEvent(caller, ev_out, 0);
}
const float* Model::stage_noise(const float* src, float* buf, hipStream_t s) {
  if (!src) return nullptr;
  (void)hipMemcpyAsync(buf, src, (size_t)Bt * Dout * sizeof(float), hipMemcpyDeviceToDevice, s);
  return buf;
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

static void add_lstm(ParamSet& ps, std::vector<LstmLayer>& out, const std::string& prefix, int I, int H, int proj) {
  LstmLayer L;
  const int P = proj > 0 ? proj : H;
  L.has_proj = proj > 0;
  L.I = I; L.H = H; L.P = P; L.ldI = pad4(I); L.ldP = pad4(P); L.ldH = pad4(H);
  L.tK = ps.add(prefix + "/kernel", I + P, 4 * H, false);
  L.tb = ps.add(prefix + "/bias", 1, 4 * H, true);
  L.twf = ps.add(prefix + "/w_f_diag", 1, H, true);
  L.twi = ps.add(prefix + "/w_i_diag", 1, H, true);
  L.two = ps.add(prefix + "/w_o_diag", 1, H, true);
  L.tWp = L.has_proj ? ps.add(prefix + "/projection/kernel", H, P, false) : -1;
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
    if (d && hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) d = nullptr;
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
  B = Bt = c.batch_size; Tmax = c.max_frames; Din = c.input_dim; Dout = c.output_dim;
  ldDin = pad4(Din); ldDout = pad4(Dout);
  if (B <= 0 || Tmax <= 0 || Din <= 0 || Dout <= 0 || c.g_layers <= 0 || c.d_layers <= 0 || c.g_cells <= 0 ||
      c.d_cells <= 0 || (c.g_layers > MAXJ && !g_dnn()) || c.d_layers > MAXJ) {
    set_error("invalid sizes in rsrgan_cfg");
    return RSRGAN_ERR_INVALID;
  }
  if (c.d_type != RSRGAN_D_LSTM && c.d_type != RSRGAN_D_DNN) { set_error("Unrecognized D type %d", c.d_type); return RSRGAN_ERR_INVALID; }
  if (g_dnn() && !d_dnn()) { set_error("the frame-level generator (dnn) needs discriminator_dnn (models/gan.py:104)"); return RSRGAN_ERR_INVALID; }
  if (!g_dnn() && (c.g_proj < 0 || c.g_proj > 384)) { set_error("generator num_proj must be in [0, 384] (0 = num_proj=None)"); return RSRGAN_ERR_INVALID; }
  if (!d_dnn() && (c.d_proj < 0 || c.d_proj > 384)) { set_error("discriminator num_proj must be in [0, 384] (0 = num_proj=None)"); return RSRGAN_ERR_INVALID; }
  if (!g_dnn() && d_dnn() && c.d_joint_dim != 0) { set_error("discriminator_dnn on the sequence model is fed the target only (d_joint_dim must be 0)"); return RSRGAN_ERR_INVALID; }
  gR = c.g_proj > 0 ? c.g_proj : c.g_cells;
  dR = c.d_proj > 0 ? c.d_proj : c.d_cells;
  if (!g_dnn() && 2 * pad4(gR) > 1024) { set_error("generator layer input+state wider than 1024 floats is not supported by k_fwd_gates"); return RSRGAN_ERR_INVALID; }
  if (!d_dnn() && pad4(dR) + std::max(pad4(dR), pad4(c.output_dim)) > 1024) { set_error("discriminator layer too wide for k_fwd_gates"); return RSRGAN_ERR_INVALID; }
  if (d_dnn() && (c.d_joint_dim < 0 || c.d_joint_off < 0 || c.d_joint_off + c.d_joint_dim > Din)) { set_error("bad d_joint slice"); return RSRGAN_ERR_INVALID; }
  const int P = gR, H = c.g_cells;
  auto fc_name = [](const char* net, int i) { return std::string(net) + "/fully_connected" + (i == 0 ? "" : "_" + std::to_string(i)); };
  // <scope>/BatchNorm/* in normalization.py's build order (contrib layers create no biases under a normalizer_fn)
  auto add_bn = [&](ParamSet& ps, const std::string& nm, int o, int (&tbn)[8]) {
    static const char* kBn[8] = {"beta", "gamma", "moving_mean", "moving_variance", "renorm_mean", "renorm_mean_weight", "renorm_stddev",
                                 "renorm_stddev_weight"};
    for (int k = 0; k < 8; ++k) {
      const bool scalar = k == 5 || k == 7;
      tbn[k] = ps.add(nm + "/BatchNorm/" + kBn[k], 1, scalar ? 1 : o, true);
      ps.t[tbn[k]].l2 = false;                                     // constant-initialised, never regularised
      ps.t[tbn[k]].bias_init = (k == 1 || k == 3) ? 1.f : 0.f;     // gamma = 1, moving_variance = 1
      ps.t[tbn[k]].trainable = k < 2;
    }
  };
  auto add_fc = [&](ParamSet& ps, std::vector<FcLayer>& out, const std::string& nm, int in, int o, bool bn = false) {
    FcLayer F; F.in = in; F.out = o; F.ld_in = pad4(in); F.ld_out = pad4(o);
    F.tW = ps.add(nm + "/weights", in, o, false);
    if (bn) {
      F.bn = true; F.tb = -1;
      add_bn(ps, nm, o, F.tbn);
    } else {
      F.tb = ps.add(nm + "/biases", 1, o, true);
    }
    out.push_back(F);
  };
  if (bn_on() && !g_dnn()) {
    set_error("RSRGAN_FLAG_BATCH_NORM is built for the frame-level generators (dnn, rced) + discriminator_dnn only");
    return RSRGAN_ERR_INVALID;
  }
  if (bn_on()) {      // the update ops of a run are ONE launch over a fixed table (kernels.h BnCommitList): refuse what it cannot hold
    const int nbn = (c.g_type == RSRGAN_G_RCED ? 9 : c.g_layers) + (c.d_type == RSRGAN_D_DNN ? c.d_layers : 0);
    if (nbn > 24) { set_error("batch norm: %d normalised layers, at most 24 are supported", nbn); return RSRGAN_ERR_INVALID; }
  }
  // ---- variable tables in graph-construction order (gan_rnn_placeholder.py:301-317) ----
  if (c.g_type == RSRGAN_G_RCED) {                                       // models/rced.py:90-116
    static const int kNum[9] = {12, 16, 20, 24, 32, 24, 20, 16, 12}, kWidth[9] = {13, 11, 9, 7, 7, 7, 9, 11, 13};
    if (c.g_splice <= 0 || Din % c.g_splice) { set_error("R-CED: input_dim must be g_splice x frame width"); return RSRGAN_ERR_INVALID; }
    if (c.g_layers < 1 || c.g_layers > 9) { set_error("R-CED: g_layers must be in [1, 9]"); return RSRGAN_ERR_INVALID; }
    rcS = c.g_splice; rcW = Din / c.g_splice;
    int cin = 1;
    for (int l = 0; l < c.g_layers; ++l) {
      // the reference's table; g_cells < 32 scales it down for tests (every width stays a multiple of 4, col2im needs that)
      const int co = c.g_cells >= 32 ? kNum[l] : std::max(4, kNum[l] * c.g_cells / 32 / 4 * 4);
      ConvLayer L; L.fw = kWidth[l]; L.Cin = cin; L.Cout = co; L.K = rcS * L.fw * cin; L.ldK = pad4(L.K);
      L.ldCin = l == 0 ? 1 : pad4(cin); L.ldCout = pad4(co);
      const std::string nm = std::string("g_model/Conv") + (l == 0 ? "" : "_" + std::to_string(l));
      L.tW = G.add(nm + "/weights", L.K, co, false);
      if (bn_on()) { L.bn = true; L.tb = -1; add_bn(G, nm, co, L.tbn); }
      else L.tb = G.add(nm + "/biases", 1, co, true);
      G.t[L.tW].xavier_fan_out = rcS * L.fw * co;           // xavier for conv: receptive field x channels on both sides
      gconv.push_back(L);
      cin = co;
    }
    const int flat = rcS * rcW * cin;
    rc_fc.in = flat; rc_fc.out = Dout; rc_fc.ld_in = flat; rc_fc.ld_out = ldDout;
    rc_fc.tW = G.add("g_model/fully_connected/weights", flat, Dout, false);
    rc_fc.tb = G.add("g_model/fully_connected/biases", 1, Dout, true);
    G.t[rc_fc.tb].bias_init = 0.1f;
  } else if (c.g_type == RSRGAN_G_DNN) {                                 // models/dnn.py:79-110: (1+3) x [FC units, ReLU], FC -> Dout
    int in = Din;
    for (int l = 0; l < c.g_layers; ++l) { add_fc(G, gfc, fc_name("g_model", l), in, c.g_cells, bn_on()); in = c.g_cells; }
    add_fc(G, gfc, fc_name("g_model", c.g_layers), in, Dout);
  } else if (c.g_type == RSRGAN_G_LSTM) {                                // models/lstm.py:82-124
    g_fc_in_w = G.add("g_model/fully_connected/weights", Din, P, false);
    g_fc_in_b = G.add("g_model/fully_connected/biases", 1, P, true);
    for (int l = 0; l < c.g_layers; ++l)
      add_lstm(G, gl, "g_model/rnn/multi_rnn_cell/cell_" + std::to_string(l) + "/lstm_cell", P, H, c.g_proj);
    g_fc_out_w = G.add("g_model/fully_connected_1/weights", P, Dout, false);
    g_fc_out_b = G.add("g_model/fully_connected_1/biases", 1, Dout, true);
  } else if (c.g_type == RSRGAN_G_RES_LSTM_L || c.g_type == RSRGAN_G_RES_LSTM_BASE) {   // models/res_lstm_l.py:101-194
    if (c.g_type == RSRGAN_G_RES_LSTM_L && P != Din) {
      set_error("res_lstm_l needs g_proj == input_dim (models/res_lstm_l.py:111)");
      return RSRGAN_ERR_INVALID;
    }
    int in = Din;
    for (int l = 0; l < c.g_layers; ++l) {
      add_lstm(G, gl, "g_model/lstm_cell_" + std::to_string(l + 1) + "/rnn/lstm_cell", in, H, c.g_proj);
      in = P;
    }
    g_fc_out_w = G.add("g_model/forward_out/fully_connected/weights", P, Dout, false);
    g_fc_out_b = G.add("g_model/forward_out/fully_connected/biases", 1, Dout, true);
  } else {
    set_error("Unrecognized G type %d", c.g_type);                       // gan_rnn_placeholder.py:131-132
    return RSRGAN_ERR_INVALID;
  }
  if (d_dnn()) {                                                         // models/discriminator_dnn.py:61-92
    int in = c.d_joint_dim + Dout;
    for (int l = 0; l < c.d_layers; ++l) { add_fc(D, dfc, fc_name("d_model", l), in, c.d_cells, bn_on()); in = c.d_cells; }
    add_fc(D, dfc, fc_name("d_model", c.d_layers), in, 1);
  } else {                                                               // models/discriminator_lstm.py:70-104
    int in = Dout;
    for (int l = 0; l < c.d_layers; ++l) {
      add_lstm(D, dl, "d_model/rnn/multi_rnn_cell/cell_" + std::to_string(l) + "/lstm_cell", in, c.d_cells, c.d_proj);
      in = dR;
    }
    d_fc_w = D.add("d_model/fully_connected/weights", dR, 1, false);
    d_fc_b = D.add("d_model/fully_connected/biases", 1, 1, true);
  }
  // ---- row padding for the persistent generator recurrences (model.h Bt) ----
  if (const char* e = getenv("RSRGAN_GPERSIST")) gp_env = atoi(e);        // bit 0: the generator's forward recurrence, bit 1: its BPTT
  {
    static const bool pad_env = [] { const char* e = getenv("RSRGAN_PAD_ROWS"); return !e || atoi(e) != 0; }();
    const int Bp = (Bt + GP_ROWS - 1) / GP_ROWS * GP_ROWS;
    if (pad_env && gp_env && Bp != Bt && !g_dnn() && !d_dnn() && wavefront()) {
      B = Bp;
      GPersistArgs ga{};
      if (!gpersist_shape(ga, std::min(Tmax, (int)GP_TMAX))) B = Bt;     // (only where the padded batch does take the persistent path)
    }
  }
  // ---- device buffers ----
  const bool ema_on = c.ema_decay > 0.f;
  G.w = alloc<float>(G.padded); G.g = alloc<float>(G.padded); G.m = alloc<float>(G.padded); G.v = alloc<float>(G.padded);
  G.ema = ema_on ? alloc<float>(G.padded) : nullptr;
  D.w = alloc<float>(D.padded); D.g = alloc<float>(D.padded);
  if (d_adam()) { D.m = alloc<float>(D.padded); D.v = alloc<float>(D.padded); }
  D.ema = ema_on ? alloc<float>(D.padded) : nullptr;
  if (!G.w || !G.g || !G.m || !G.v || !D.w || !D.g) { set_error("hipMalloc failed (parameters)"); return RSRGAN_ERR_HIP; }
  if (build_chunks(*this, G) || build_chunks(*this, D)) { set_error("hipMalloc failed (chunk tables)"); return RSRGAN_ERR_HIP; }
  for (auto* layers : {&gl, &dl})
    for (auto& L : *layers) {
      L.KxT = alloc<float>((size_t)4 * L.H * L.ldI);
      L.KhT = alloc<float>((size_t)4 * L.H * L.ldP);
      L.WpT = L.has_proj ? alloc<float>((size_t)L.P * L.ldH) : nullptr;
      const int ncb = (L.H + 15) / 16, kbI = (L.ldI + L.ldP + 15) / 16, kbP = (L.ldP + 15) / 16, kbH = (L.ldH + 15) / 16, kb4 = (4 * L.H + 15) / 16;
      L.Wg_full = alloc<float>(swizzle_floats(4 * ncb, kbI));
      L.Wg_h = alloc<float>(swizzle_floats(4 * ncb, kbP));
      L.WpT_sw = L.has_proj ? alloc<float>(swizzle_floats((L.P + 15) / 16, kbH)) : nullptr;
      L.Wp_sw = L.has_proj ? alloc<float>(swizzle_floats(ncb, kbP)) : nullptr;
      L.Kb_full = alloc<float>(swizzle_floats((L.I + L.P + 15) / 16, kb4));
      L.Kb_rec = alloc<float>(swizzle_floats((L.P + 15) / 16, kb4));
    }
  const size_t TB = (size_t)Tmax * B;
  x_tm = alloc<float>(TB * ldDin); lab_tm = alloc<float>(TB * ldDout); y_tm = alloc<float>(TB * ldDout);
  const int ldP = g_dnn() ? 4 : pad4(P);
  g_st.resize(gl.size());
  for (size_t l = 0; l < gl.size(); ++l) alloc_stash(*this, g_st[l], gl[l], B, Tmax);
  g_ins.resize(gl.size() + 1);
  if (g_rced()) {
    const size_t M = TB * rcS * rcW;
    int maxK = 4, maxC = 4;
    rc_act.push_back(x_tm);
    for (auto& L : gconv) {
      rc_act.push_back(alloc<float>(M * L.ldCout));
      if (L.bn) { L.pre = alloc<float>(M * L.ldCout); L.stat = alloc<float>((size_t)BN_STAT_ROWS * L.ldCout); }
      maxK = std::max(maxK, L.ldK); maxC = std::max(maxC, L.ldCout);
    }
    // implicit-GEMM convolution (conv.hip) for every layer it covers: forward and data gradient never build a patch matrix
    if (const char* e = getenv("RSRGAN_RCED_IMPLICIT")) rc_implicit = atoi(e) != 0;
    rc_ft_fwd.assign(gconv.size(), nullptr); rc_ft_bwd.assign(gconv.size(), nullptr);
    rc_wgrad_implicit.assign(gconv.size(), 0);
    bool any_implicit = false;
    size_t wg_ws = 0;
    for (size_t l = 0; l < gconv.size() && rc_implicit; ++l) {
      const ConvLayer& L = gconv[l];
      if (conv_fwd_supported(L.Cin, L.Cout, rcS, rcW, L.fw)) { rc_ft_fwd[l] = alloc<float>(conv_prep_floats(rcS, L.fw, L.Cin)); any_implicit = true; }
      if (l > 0 && conv_fwd_supported(L.Cout, L.Cin, rcS, rcW, L.fw)) rc_ft_bwd[l] = alloc<float>(conv_prep_floats(rcS, L.fw, L.Cout));
      if (conv_wgrad_supported(L.Cin, L.Cout, rcS, rcW, L.fw)) {
        rc_wgrad_implicit[l] = 1;
        wg_ws = std::max(wg_ws, conv_wgrad_ws_floats(L.Cin, (int)TB, rcS, rcW, L.fw));
      }
    }
    if (wg_ws) rc_wg_ws = alloc<float>(wg_ws);
    if (rc_ft_fwd[0] && rc_wgrad_implicit[0]) rc_x4 = alloc<float>(M * 4);
    else { rc_ft_fwd[0] = nullptr; rc_wgrad_implicit[0] = 0; }          // layer 0 is implicit only as a whole
    // patch matrices: kept per layer from the forward pass when they fit a 96 GB budget (288 GB HBM3E), else one shared
    // buffer that the backward pass refills
    size_t keep = 0;
    for (auto& L : gconv) keep += M * L.ldK;
    rc_keep_cols = !any_implicit && keep * sizeof(float) <= ((size_t)96 << 30) && !cfg.cross_validation;
    if (rc_keep_cols) {
      for (auto& L : gconv) { rc_cols.push_back(alloc<float>(M * L.ldK)); if (!rc_cols.back()) rc_keep_cols = false; }
      if (!rc_keep_cols) rc_cols.clear();
    }
    // shared patch-matrix buffers only as wide as the layers that still take the patch-matrix path need them
    int colK = 4, dcolK = 4;
    for (size_t l = 0; l < gconv.size(); ++l) {
      if (!rc_ft_fwd[l] || !rc_wgrad_implicit[l]) colK = std::max(colK, gconv[l].ldK);
      if (l > 0 && !rc_ft_bwd[l]) dcolK = std::max(dcolK, gconv[l].ldK);
    }
    (void)maxK;
    rc_col = rc_keep_cols ? rc_cols[0] : alloc<float>(M * colK);
    rc_dcol = alloc<float>(M * dcolK);
    rc_dA = alloc<float>(M * maxC); rc_dB = alloc<float>(M * maxC);
    if (!rc_col || !rc_dcol || !rc_dA || !rc_dB) { set_error("hipMalloc failed (R-CED buffers: %zu positions)", M); return RSRGAN_ERR_HIP; }
  } else if (g_dnn()) {
    g_act.push_back(x_tm);
    for (size_t l = 0; l + 1 < gfc.size(); ++l) g_act.push_back(alloc<float>(TB * gfc[l].ld_out));
    g_act.push_back(y_tm);
  } else if (c.g_type == RSRGAN_G_LSTM) {
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
  const int gmaxld = std::max(std::max(ldP, ldDin), ldDout);      // (g_dB / g_dC also carry dy [T*B][ldDout]: an output wider than the generator's
                                                                  //  layers overflowed them -- found by __graft_entry__.smoke()'s second configuration)
  g_dA = alloc<float>(TB * gmaxld); g_dB = alloc<float>(TB * gmaxld); g_dC = alloc<float>(TB * gmaxld);
  const size_t TB2 = TB * 2;
  const int ldPd = d_dnn() ? 4 : pad4(dR);
  xd = alloc<float>(TB2 * ldDout); logits = alloc<float>(TB2 * 4); dlogits = alloc<float>(TB2 * 4);
  if (d_dnn()) {
    ldJ = pad4(c.d_joint_dim + Dout);
    if (c.d_joint_dim > 0) joint = alloc<float>(TB2 * ldJ);
    d_act.push_back(c.d_joint_dim > 0 ? joint : xd);
    for (size_t l = 0; l + 1 < dfc.size(); ++l) d_act.push_back(alloc<float>(TB2 * dfc[l].ld_out));
    d_act.push_back(logits);
    dy_buf = alloc<float>(TB * ldDout);
  }
  if (g_dnn() || d_dnn()) {
    int mx = std::max(ldJ, ldDin);
    for (auto& F : gfc) mx = std::max(mx, std::max(F.ld_in, F.ld_out));
    for (auto& F : dfc) mx = std::max(mx, std::max(F.ld_in, F.ld_out));
    fc_dA = alloc<float>(TB2 * mx); fc_dB = alloc<float>(TB2 * mx);
    if (bn_on()) {
      bn_sums = alloc<float>((size_t)2 * mx);
      for (auto& F : gfc) if (F.bn) { F.pre = alloc<float>(TB * F.ld_out); F.stat = alloc<float>((size_t)2 * BN_STAT_ROWS * F.ld_out); }
      for (auto& F : dfc) if (F.bn) { F.pre = alloc<float>(TB2 * F.ld_out); F.stat = alloc<float>((size_t)2 * BN_STAT_ROWS * F.ld_out); }
    }
  }
  adam_t_dev_d = alloc<int>(1);
  d_st.resize(dl.size());
  for (size_t l = 0; l < dl.size(); ++l) alloc_stash(*this, d_st[l], dl[l], 2 * B, Tmax);
  if (const char* e = getenv("RSRGAN_DFOLD")) fold_env = atoi(e) != 0;
  {   // folded views of the discriminator's cells (model.h): only for stacks of projected cells
    bool ok = fold_env && !dl.empty();
    for (auto& L : dl) ok = ok && L.has_proj && L.H <= 288 && (L.ldH & 3) == 0;
    if (ok) {
      dl_fold.resize(dl.size()); dl_fold_K.resize(dl.size()); d_fold_st.resize(dl.size());
      for (size_t l = 0; l < dl.size(); ++l) {
        LstmLayer F = dl[l];
        F.has_proj = false; F.tWp = -1;
        F.I = l == 0 ? dl[0].I : dl[l - 1].H; F.ldI = l == 0 ? dl[0].ldI : dl[l - 1].ldH;
        F.P = dl[l].H; F.ldP = dl[l].ldH;
        F.KxT = F.KhT = F.WpT = nullptr; F.WpT_sw = F.Wp_sw = F.Kb_full = F.Kb_rec = nullptr;
        const int ncb = (F.H + 15) / 16;
        F.Wg_full = alloc<float>(swizzle_floats(4 * ncb, (F.ldI + F.ldP + 15) / 16));
        F.Wg_h = alloc<float>(swizzle_floats(4 * ncb, (F.ldP + 15) / 16));
        dl_fold_K[l] = alloc<float>((size_t)(F.I + F.H) * 4 * F.H);
        dl_fold[l] = F;
        LstmStash S = d_st[l];
        S.mst = alloc<float>((size_t)(Tmax + 1) * 2 * B * F.ldP);
        S.out = nullptr; S.dmt = nullptr; S.dmst = nullptr;
        d_fold_st[l] = S;
      }
    }
  }
  if (const char* e = getenv("RSRGAN_DPERSIST")) dp_env = atoi(e);        // bit 0: forward, bit 1: backward
  if (dp_env && !dl.empty() && dl.size() <= (size_t)DP_MAXL && B % 16 == 0) {
    // every workgroup of a launch waits for the others: ask the device whether the stacked (2B rows) or at least the single (B rows)
    // launch is resident at once -- a CU mask, a partition or a neighbour can take CUs away without multiProcessorCount knowing
    const int nl_ = (int)dl.size();
    for (int rows : {2 * B, B}) {
      const int g_ = dpersist_grid(nl_, rows);
      if (g_ <= 128 && resident_probe(g_, 512, dpersist_lds_bytes())) { dp_max_grid = g_; break; }
    }
  }
  if (dp_max_grid > 0) {
    dp_gran_bytes = dpersist_granule_bytes((int)dl.size(), 2 * B, Tmax);
    dp_gran = (unsigned long long*)alloc<float>(dp_gran_bytes / sizeof(float));
    dp_ctl = (unsigned*)alloc<float>(16);
    const unsigned ctl0[DP_CTL_WORDS] = {1u, 0u, 0u, 0u};
    HIPC(hipMemcpy(dp_ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice));
    // the weight gradients inside the D-run's BPTT launch (the stacked call over 2B rows): shapes only here, pointers per call
    static const bool dw_env = [] { const char* e = getenv("RSRGAN_DW_INKERNEL"); return !e || atoi(e) != 0; }();
    DPersistArgs pa{};
    pa.nl = (int)dl.size(); pa.N = 2 * B; pa.H = dl[0].H;
    for (size_t l = 0; l < dl.size(); ++l) { pa.L[l].I = dl[l].I; pa.L[l].P = dl[l].P; pa.L[l].ldP = dl[l].ldP; pa.L[l].ldI = dl[l].ldI; }
    bool shapes = dw_env && (dp_env & 2) && dpersist_dw_supported(pa) && !d_adam();
    for (auto& L : dl) shapes = shapes && L.has_proj && L.H == dl[0].H && L.ldH == dl[0].ldH;
    if (shapes && dpersist_grid(pa.nl, pa.N) == dp_max_grid) {
      const int nt = pa.N / 16;
      dw_stride = dpersist_dw_stride(pa);
      dw_ws = alloc<float>((size_t)pa.nl * nt * dw_stride);
      dw_flag = (unsigned*)alloc<float>(dpersist_dw_flag_bytes(pa.nl, pa.N, Tmax) / sizeof(float));
      std::vector<long long> src(D.t.size(), -1);
      const int IP = dl[0].I + dl[0].P, H4 = 4 * dl[0].H, H_ = dl[0].H;
      for (size_t l = 0; l < dl.size(); ++l) {
        const long long base = (long long)l * nt * (long long)dw_stride;
        src[dl[l].tK] = base; src[dl[l].tb] = base + (long long)IP * H4;
        const long long pp = base + (long long)(IP + 1) * H4;
        src[dl[l].twi] = pp; src[dl[l].twf] = pp + H_; src[dl[l].two] = pp + 2 * H_; src[dl[l].tWp] = pp + 3 * H_;
      }
      dw_src = (long long*)alloc<float>(src.size() * 2);
      if (dw_ws && dw_flag && dw_src) HIPC(hipMemcpy(dw_src, src.data(), src.size() * sizeof(long long), hipMemcpyHostToDevice));
      else dw_ws = nullptr;
    }
  }
  if (gp_env && !g_dnn()) {
    GPersistArgs ga{};
    gp_Tcap = std::min(Tmax, (int)GP_TMAX);
    gp_noproj = !gl.empty() && !gl[0].has_proj;
    if (gp_noproj) {
      // 8 cells per workgroup (all weights in registers) needs twice the workgroups of 16: B = 64 with two layers is the whole
      // device -- ask it; if it cannot hold them, try the 16-cell form
      bool fits = false;
      for (int nt : {2, 4}) {
        gp_np_nt = nt;
        if (gpersist_shape(ga, gp_Tcap) && resident_probe(gpersist_grid(ga), GP_THREADS, gpersist_np_lds_bytes())) { fits = true; break; }
      }
      if (!fits) gp_np_nt = 0;
      if (fits) {
        gp_gran2_bytes = gpersist_np_gran2_bytes(ga);
        gp_gran2 = (unsigned long long*)alloc<float>(gp_gran2_bytes / sizeof(float));
        gp_ctl = (unsigned*)alloc<float>(16);
        // the BPTT form exists for 8 cells per workgroup only: its two rings (state gradient, input gradient between layers)
        static const bool npb_env = [] { const char* e = getenv("RSRGAN_GP_NP_BWD"); return !e || atoi(e) != 0; }();
        if (gp_np_nt == 2 && (gp_env & 2) && npb_env) {
          gp_gran1 = (unsigned long long*)alloc<float>(gpersist_np_gran1_bytes(ga) / sizeof(float));
          gp_gran3 = (unsigned long long*)alloc<float>(gpersist_np_gran3_bytes(ga) / sizeof(float));
          if (!gp_gran1 || !gp_gran3) { gp_gran1 = gp_gran3 = nullptr; }
        }
        if (!gp_gran1) gp_gran1 = gp_gran2;                       // (forward only: the pointer just says "the forward path is on")
        if (gp_gran2 && gp_ctl) {
          const unsigned ctl0[DP_CTL_WORDS] = {1u, 0u, 0u, 0u};
          HIPC(hipMemcpy(gp_ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice));
          gpersist_rearm();
        } else { gp_gran1 = gp_gran2 = gp_gran3 = nullptr; gp_ctl = nullptr; }
      }
    } else if (gpersist_args(ga, gp_Tcap) && resident_probe(gpersist_grid(ga), GP_THREADS, gpersist_lds_bytes())) {
      gp_gran1 = (unsigned long long*)alloc<float>(gpersist_gran1_bytes(ga) / sizeof(float));
      gp_gran2_bytes = gpersist_gran2_bytes(ga);
      gp_gran2 = (unsigned long long*)alloc<float>(gp_gran2_bytes / sizeof(float));
      gp_ctl = (unsigned*)alloc<float>(16);
      if (gp_env & 2) gp_gran3 = (unsigned long long*)alloc<float>(gpersist_gran3_bytes(ga) / sizeof(float));
      if (gp_gran1 && gp_gran2 && gp_ctl) {
        const unsigned ctl0[DP_CTL_WORDS] = {1u, 0u, 0u, 0u};
        HIPC(hipMemcpy(gp_ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice));
        gpersist_rearm();
      } else { gp_gran1 = gp_gran2 = nullptr; gp_ctl = nullptr; }
    }
  }
  {
    // the trailing discriminator BPTT beside k_glstm_bwd (the G-run): both launches' workgroups resident at once -- ask the device
    if (const char* e = getenv("RSRGAN_TRAIL")) trail_mode = atoi(e);
    GPersistArgs ga{};
    if (trail_mode && dp_gran && (dp_env & 2) && gp_gran1 && gp_gran3 && !gp_noproj && (gp_env & 2) && B % 32 == 0 && gpersist_args(ga, gp_Tcap))
      trail_fits = resident_probe(gpersist_grid(ga) + ((dpersist_trail_grid((int)dl.size(), B) + 7) & ~7), GP_THREADS,
                                  std::max(gpersist_lds_bytes(), dpersist_trail_lds_bytes()));
  }
  {
    static const bool lazy_env = [] { const char* e = getenv("RSRGAN_LAZY_SWIZZLE"); return !e || atoi(e) != 0; }();
    lazy_sw = lazy_env && wavefront() && gp_gran1 && gp_gran3 && (gp_env & 3) == 3 && dp_gran && (dp_env & 3) == 3;
  }
  drop_ctr = (unsigned long long*)alloc<float>(4);
  HIPC(hipMemset(drop_ctr, 0, 16));
  const int dmaxld = std::max(ldPd, ldDout);
  d_dA = alloc<float>(TB2 * dmaxld); d_dB = alloc<float>(TB2 * dmaxld);
  len_dev = alloc<int>(2 * B);
  zeros = alloc<float>(64);
  noise_r_buf = alloc<float>((size_t)B * Dout); noise_f_buf = alloc<float>((size_t)B * Dout);
  if (const char* e = getenv("RSRGAN_GRAPHS")) graphs_env = atoi(e) != 0;
  if (hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking) != hipSuccess) main_s = nullptr;
  if (main_s && (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
                 hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess)) { (void)hipStreamDestroy(main_s); main_s = nullptr; }
  dyn = alloc<float>(DYN_COUNT); adam_t_dev = alloc<int>(1);
  losses = alloc<float>(8); tmp3 = alloc<float>(4);
  size_t maxcols = 7 * (size_t)std::max(c.g_cells, c.d_cells);
  maxcols = std::max(maxcols, (size_t)Din + 4);
  maxcols = std::max(maxcols, (size_t)std::max(ldP, ldDin));
  scratch_floats = std::max<size_t>(4 * 64 * maxcols, 16384);            // (x 4: the column sums of up to four layers in one launch)
  scratch = alloc<float>(scratch_floats);
  scratch2 = alloc<float>(std::max<size_t>(4 * 64 * maxcols, 1024));
  g_fc_out_wT = g_dnn() ? nullptr : alloc<float>((size_t)Dout * ldP);
  if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) side = nullptr;
  if (hipEventCreateWithFlags(&ev_last, hipEventDisableTiming) != hipSuccess) ev_last = nullptr;
  {
    const char* e = getenv("RSRGAN_DPIPE");
    if (e && atoi(e) != 0 && trail_fits && side && !d_dnn() && dp_max_grid >= dpersist_grid((int)dl.size(), B)) {
      const size_t gb = dpersist_granule_bytes((int)dl.size(), B, Tmax) / 2;      // (a forward launch: one edge per layer)
      dp_gran2 = (unsigned long long*)alloc<float>(gb / sizeof(float));
      dp_ctl2 = (unsigned*)alloc<float>(16);
      if (dp_gran2 && dp_ctl2 && hipEventCreateWithFlags(&ev_dfree, hipEventDisableTiming) == hipSuccess &&
          hipEventCreateWithFlags(&ev_real, hipEventDisableTiming) == hipSuccess) {
        const unsigned ctl0[DP_CTL_WORDS] = {1u, 0u, 0u, 0u};
        HIPC(hipMemcpy(dp_ctl2, ctl0, sizeof(ctl0), hipMemcpyHostToDevice));
        dpipe = true;
      }
    }
  }
  if (side) {
    for (auto& e : ev_pool)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { side = nullptr; break; }
  }
  gemm_ws_floats = (size_t)32 << 20;          // 128 MiB of split-K partial tiles / stream-K pieces (two 192 x 256 pieces per worker: 100 MB)
  gemm_ws = alloc<float>(gemm_ws_floats);
  if (!gemm_ws) gemm_ws_floats = 0;
  bwdb_ws_floats = (size_t)8 << 20;
  bwdb_ws = alloc<float>(bwdb_ws_floats);
  gemm_ws2 = alloc<float>(gemm_ws_floats ? gemm_ws_floats : 1);
  static const bool dk_pad = [] { const char* e = getenv("RSRGAN_DK_PAD"); return !e || atoi(e) != 0; }();
  if (dk_pad && !gl.empty() && gl[0].has_proj && gl[0].I % 4 != 0 && gl[0].ldI % 4 == 0 && gl.size() <= (size_t)GEMM_MAXB) {
    dk_tmp_per = (size_t)(gl[0].ldI + gl[0].P) * 4 * gl[0].H;
    dk_tmp = alloc<float>(dk_tmp_per * gl.size());
    if (!dk_tmp) dk_tmp_per = 0;
  }
  if (!gemm_ws2) side = nullptr;
  if (!side) dpipe = false;
  if (!scratch || !d_dB || !g_dB || !xd) { set_error("hipMalloc failed (activations)"); return RSRGAN_ERR_HIP; }

  // ---- initial values: xavier_initializer() uniform / zeros (models/lstm.py:86-87,93) ----
  std::mt19937_64 rng(seed);
  for (ParamSet* ps : {&G, &D}) {
    std::vector<float> host((size_t)ps->padded, 0.f);
    for (auto& t : ps->t) {
      if (!t.l2) {                                           // biases: zero, except R-CED's output FC (rced.py:116: 0.1)
        if (t.bias_init != 0.f)
          for (int cc = 0; cc < t.cols; ++cc) host[(size_t)t.off + cc] = t.bias_init;
        continue;
      }
      const double fan_in = t.is_vector ? t.cols : t.rows, fan_out = t.xavier_fan_out > 0 ? t.xavier_fan_out : t.cols;
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
  { int rc = build_buckets(); if (rc) return rc; }
  HIPC(hipDeviceSynchronize());
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

void Model::destroy() {
  drop_graphs();
  if (main_s) {
    (void)hipStreamSynchronize(main_s);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    (void)hipStreamDestroy(main_s);
    main_s = nullptr;
  }
  if (side) {
    (void)hipStreamSynchronize(side);
    for (auto& e : ev_pool) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(side);
    side = nullptr;
  }
  if (ev_last) { (void)hipEventDestroy(ev_last); ev_last = nullptr; ev_last_set = false; }
  if (ev_dfree) { (void)hipEventDestroy(ev_dfree); ev_dfree = nullptr; }
  if (ev_real) { (void)hipEventDestroy(ev_real); ev_real = nullptr; }
  dpipe = false;
  for (auto& v : gbk) { for (auto& b : v) if (b.ev) (void)hipEventDestroy(b.ev); v.clear(); }
  for (auto& e : prof_ev) if (e) (void)hipEventDestroy(e);
  prof_ev.clear();
  for (auto& e : prof_gp_ev) if (e) (void)hipEventDestroy(e);
  prof_gp_ev.clear();
  for (auto& e : prof_gb_ev) if (e) (void)hipEventDestroy(e);
  prof_gb_ev.clear();
  for (void* p : allocs) (void)hipFree(p);
  allocs.clear();
}

int Model::build_buckets() {
  auto range = [](const ParamSet& ps, int lo, int hi, GradBucket& b) {      // tensors lo..hi inclusive
    b.off = ps.t[lo].off;
    b.count = (hi + 1 < (int)ps.t.size() ? ps.t[hi + 1].off : ps.padded) - b.off;
  };
  auto whole = [](const ParamSet& ps, std::vector<GradBucket>& v) { GradBucket b; b.off = 0; b.count = ps.padded; v.assign(1, b); };
  whole(D, gbk[RSRGAN_NET_D]);
  whole(G, gbk[RSRGAN_NET_G]);
  if (!g_dnn() && wavefront()) {
    // completion order of the merged generator backward: output FC, input FC (g_type lstm), then the LSTM layers
    std::vector<GradBucket> v;
    bool ok = true;
    GradBucket b;
    range(G, g_fc_out_w, g_fc_out_b, b); ok = ok && g_fc_out_b == g_fc_out_w + 1; v.push_back(b);
    if (cfg.g_type == RSRGAN_G_LSTM) { range(G, g_fc_in_w, g_fc_in_b, b); ok = ok && g_fc_in_b == g_fc_in_w + 1; v.push_back(b); }
    for (auto& L : gl) {
      const int lo = L.tK, hi = L.has_proj ? L.tWp : L.two;
      ok = ok && L.tb > lo && L.tb < hi && L.twf > lo && L.twi > lo && L.two <= hi && hi - lo == (L.has_proj ? 5 : 4);
      range(G, lo, hi, b); v.push_back(b);
    }
    int64_t covered = 0;
    for (auto& x : v) covered += x.count;
    if (ok && covered == G.padded) gbk[RSRGAN_NET_G] = v;      // else: keep the single whole-buffer bucket
  }
  for (auto& v : gbk)
    for (auto& b : v)
      if (hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return RSRGAN_ERR_HIP; }
  return RSRGAN_OK;
}
void Model::mark_bucket(int net, int i, hipStream_t s) {
  GradBucket& b = gbk[net][i];
  (void)hipEventRecord(b.ev, s);
  b.marked = true;
}
void Model::finish_buckets(int net, hipStream_t s) {
  for (auto& b : gbk[net]) {
    if (!b.marked) (void)hipEventRecord(b.ev, s);
    b.marked = false;
  }
}

void Model::refresh_transposes(int net, hipStream_t s) {
  const ParamSet& ps = net == RSRGAN_NET_G ? G : D;
  auto& layers = net == RSRGAN_NET_G ? gl : dl;
  TransposeList tl{};
  auto add = [&](const float* src, int lds_, float* dst, int ldd, int R, int C) {
    if (tl.n == 16) { launch_transpose_many(tl, s); tl.n = 0; }
    tl.j[tl.n++] = TransposeJob{src, dst, lds_, ldd, R, C, 0};
  };
  for (auto& L : layers) {
    const float* K = ps.W(L.tK);
    const int H4 = 4 * L.H;
    add(K, H4, L.KxT, L.ldI, L.I, H4);                       // [I][4H] -> [4H][ldI]
    add(K + (size_t)L.I * H4, H4, L.KhT, L.ldP, L.P, H4);    // [P][4H] -> [4H][ldP]
    if (L.has_proj) add(ps.W(L.tWp), L.ldP, L.WpT, L.ldH, L.H, L.P);         // [H][ldP] -> [P][ldH]
  }
  if (net == RSRGAN_NET_G && g_fc_out_wT && g_fc_out_w >= 0)
    add(G.W(g_fc_out_w), ldDout, g_fc_out_wT, pad4(gR), gR, Dout);   // [P][ldDout] -> [Dout][ldP]
  launch_transpose_many(tl, s);
  if (!lazy_sw) refresh_swizzles(net, s);
  // (the folded discriminator kernels are derived where they are used, fold_forward: with the persistent discriminator launch on,
  //  that is never -- three GEMMs, a copy and a swizzle launch, 60 us per step, used to follow every discriminator update)
  if (net == RSRGAN_NET_G)
    for (size_t l = 0; l < gconv.size(); ++l) {                       // R-CED: re-arranged filters of the implicit-GEMM conv
      const ConvLayer& L = gconv[l];
      if (rc_ft_fwd[l]) launch_conv_prep(G.W(L.tW), L.ldCout, rcS, L.fw, L.Cin, L.Cout, false, rc_ft_fwd[l], s);
      if (rc_ft_bwd[l]) launch_conv_prep(G.W(L.tW), L.ldCout, rcS, L.fw, L.Cin, L.Cout, true, rc_ft_bwd[l], s);
    }
}

// The fragment-tiled weight copies of the launch-per-phase recurrence kernels (k_fwd_gates, k_fwd_proj, k_bwd_a2, k_bwd_bp).  With
// both generator and discriminator recurrences on their persistent launches (lazy_sw) nothing reads them in a training step: they are
// then rebuilt at the top of rnn_forward / rnn_backward -- the only readers -- instead of after every optimizer step (46 us for the
// generator's 23 MB, 11 us for the discriminator).
void Model::refresh_swizzles(int net, hipStream_t s) {
  const ParamSet& ps = net == RSRGAN_NET_G ? G : D;
  auto& layers = net == RSRGAN_NET_G ? gl : dl;
  {
    SwizzleList sl{};
    auto addz = [&](const SwizzleJob& j) {
      if (!j.dst) return;
      if (sl.n == 24) { launch_swizzle_many(sl, s); sl.n = 0; }
      sl.j[sl.n++] = j;
    };
    for (auto& L : layers) {
      const float* K = ps.W(L.tK);
      const int H4 = 4 * L.H, ncb = (L.H + 15) / 16, kbI = (L.ldI + L.ldP + 15) / 16, kbP = (L.ldP + 15) / 16, kbH = (L.ldH + 15) / 16, kb4 = (H4 + 15) / 16;
      //            src  dst        ld     gates H    C          c0   K1  I    P    kx     nct                 nkb
      addz(SwizzleJob{K, L.Wg_full, H4,    4,    L.H, 0,         0,   0,  L.I, L.P, L.ldI, 4 * ncb,            kbI, 0});
      addz(SwizzleJob{K, L.Wg_h,    H4,    4,    L.H, 0,         0,   0,  L.I, L.P, 0,     4 * ncb,            kbP, 0});
      addz(SwizzleJob{K, L.Kb_full, H4,    0,    0,   L.I + L.P, 0,   H4, 0,   0,   0,     (L.I + L.P + 15) / 16, kb4, 0});
      addz(SwizzleJob{K, L.Kb_rec,  H4,    0,    0,   L.P,       L.I, H4, 0,   0,   0,     (L.P + 15) / 16,    kb4, 0});
      if (L.has_proj) {
        const float* Wp = ps.W(L.tWp);                   // [H][ldP]
        addz(SwizzleJob{Wp, L.WpT_sw, L.ldP, 1,  L.P, 0,         0,   0,  L.H, 0,   16 * kbH, (L.P + 15) / 16, kbH, 0});     // M[p][k] = Wp[k][p]
        addz(SwizzleJob{Wp, L.Wp_sw,  L.ldP, 0,  0,   L.H,       0,   L.P, 0,  0,   0,     ncb,                kbP, 0});     // M[cell][k] = Wp[cell][k]
      }
    }
    launch_swizzle_many(sl, s);
  }
}

// ------------------------------------------------------------------------------------------
// stacks of dynamic_rnn(LSTMCell) layers.  Two schedules over the same kernels:
//   layer-sequential: per layer one time-batched GEMM for the x-part, then 2 launches per step;
//   wavefront (RSRGAN_FLAG_WAVEFRONT): launch d carries every (layer l, t = d - l) job of every
//   co-scheduled chain, so a 3-layer stack over T steps costs 2*(T+2) launches instead of 6*T.
// ------------------------------------------------------------------------------------------
static inline int kb16(int ld) { return (ld + 15) >> 4; }

bool Model::bwd_b_splitk_ok(const BwdBJobs& jobs) const {
  if (!bwdb_ws || (cfg.flags & RSRGAN_FLAG_NO_SPLITK_B)) return false;
  BwdBJobs tmp = jobs;
  const size_t need = bwd_b_plan(tmp, nullptr);
  if (need > bwdb_ws_floats) return false;
  int blocks = 0;
  for (int i = 0; i < tmp.n; ++i) {
    if ((tmp.j[i].kpg + 1) / 2 > 12) return false;       // k_bwd_bp holds <= 12 k-blocks of weights per wave
    blocks += tmp.j[i].KG * tmp.j[i].ncg * tmp.j[i].nrg;
  }
  return blocks >= 96;          // small launches (the discriminator alone): one 32x16-tile launch is faster (9.3 vs 10.4 us)
}

void Model::gemm(const float* A, int lda, bool a_kc, const float* B_, int ldb, bool b_kc, float* C, int ldc, int M, int N,
                 int K, const float* bias, int act, float alpha, bool accumulate, hipStream_t s) {
  launch_gemm(A, lda, a_kc, B_, ldb, b_kc, C, ldc, M, N, K, bias, act, alpha, accumulate, s,
              (side && s == side) ? gemm_ws2 : gemm_ws, gemm_ws_floats);
}

static void fill_gate(FwdGateJob& a, const LayerRun& R, int t, bool zx) {
  const LstmLayer& L = *R.L; const LstmStash& S = *R.S; const ParamSet& ps = *R.ps;
  const int H = L.H, H4 = 4 * H;
  const size_t r = (size_t)t * R.Ns + R.row0, rn = (size_t)(t + 1) * R.Ns + R.row0;
  a.x = zx ? nullptr : R.in + r * L.ldI;
  a.KxT = L.KxT; a.ldx = L.ldI; a.Wsw = zx ? L.Wg_h : L.Wg_full;
  a.m = S.mst + r * L.ldP; a.KhT = L.KhT; a.ldm = L.ldP;
  a.zx = zx ? S.gates + r * H4 : nullptr; a.bias = ps.W(L.tb);
  a.wf = ps.W(L.twf); a.wi = ps.W(L.twi); a.wo = ps.W(L.two);
  a.c_prev = S.c + r * H; a.c_out = S.c + rn * H;
  a.gates = S.gates + r * H4;
  a.h = S.h + r * L.ldH; a.ldh = L.ldH;
  a.len = R.len; a.t = t; a.N = R.N; a.H = H;
  a.nblk_c = (H + 15) / 16;
  if (L.has_proj) { a.np_m_out = nullptr; a.np_out = nullptr; a.np_res_in = nullptr; a.np_res_out = nullptr; }
  else {     // num_proj=None: m = h; the gates epilogue also does the dynamic_rnn masking and the residual add
    a.np_m_out = S.mst + rn * L.ldP; a.np_out = S.out ? S.out + r * L.ldP : nullptr;
    a.np_res_in = R.res_in ? R.res_in + r * L.ldP : nullptr;
    a.np_res_out = R.res_out ? R.res_out + r * L.ldP : nullptr;
  }
}
static void fill_proj(FwdProjJob& p, const LayerRun& R, int t) {
  const LstmLayer& L = *R.L; const LstmStash& S = *R.S;
  const size_t r = (size_t)t * R.Ns + R.row0, rn = (size_t)(t + 1) * R.Ns + R.row0;
  p.h = S.h + r * L.ldH; p.WpT = L.WpT; p.WpT_sw = L.WpT_sw; p.ldh = L.ldH;
  p.m_prev = S.mst + r * L.ldP; p.m_out = S.mst + rn * L.ldP; p.out = S.out + r * L.ldP;
  p.res_in = R.res_in ? R.res_in + r * L.ldP : nullptr;
  p.res_out = R.res_out ? R.res_out + r * L.ldP : nullptr;
  p.len = R.len; p.bias = nullptr; p.noise = nullptr;
  p.ldm = L.ldP; p.ldo = L.ldP; p.P = L.P; p.t = t; p.N = R.N;
  p.nblk_c = (L.P + 15) / 16;
  p.drop = R.drop; p.drop.tag += (unsigned long long)t;
}
static void fill_fc_fwd(FwdProjJob& p, const FcStage& F, int t) {
  p.h = F.in + (size_t)t * F.N * F.ld_in; p.WpT = F.WT; p.WpT_sw = nullptr; p.ldh = F.ld_in;
  p.m_prev = nullptr; p.m_out = F.y + (size_t)t * F.N * F.ldy; p.ldm = F.ldy;
  p.out = F.out2 + ((size_t)t * F.Ns2 + F.row02) * F.ld2; p.ldo = F.ld2;
  p.res_in = nullptr; p.res_out = nullptr; p.len = nullptr; p.bias = F.bias; p.noise = F.noise;
  p.P = F.D; p.t = t; p.N = F.N;
  p.nblk_c = (F.D + 15) / 16;
  p.drop = DropSpec{};
}
static void fill_fc_bwd(BwdBJob& b, const FcStage& F, int t) {   // y[t] (=|+=) in[t] . W^T, W = [D rows][ld_in]
  b.dz = F.in + (size_t)t * F.N * F.ld_in; b.K = F.WT; b.Ksw = nullptr; b.H4 = F.ld_in;
  b.dx = F.y + (size_t)t * F.N * F.ldy; b.lddx = F.ldy; b.dmst = nullptr; b.len = nullptr;
  b.I = F.D; b.n_begin = 0; b.n_end = F.D; b.ldm = 0; b.t = t; b.N = F.N;
  b.dx_accumulate = F.accumulate ? 1 : 0;
  b.nblk_c = (F.D + 15) / 16;
}
static void fill_bwd_a(BwdAJob& a, const LayerRun& R, int t) {
  const LstmLayer& L = *R.L; const LstmStash& S = *R.S; const ParamSet& ps = *R.ps;
  const int H = L.H, H4 = 4 * H;
  const size_t r = (size_t)t * R.Ns + R.row0, rn = (size_t)(t + 1) * R.Ns + R.row0;
  a.dout = R.dout ? R.dout + r * L.ldP : nullptr;
  a.dmst = S.dmst + (size_t)R.row0 * L.ldP; a.Wp = L.has_proj ? ps.W(L.tWp) : nullptr; a.Wp_sw = L.has_proj ? L.Wp_sw : nullptr;
  a.dmt = S.dmt + r * L.ldP;
  a.gates = S.gates + r * H4;
  a.c_prev = S.c + r * H; a.c_cur = S.c + rn * H;
  a.wf = ps.W(L.twf); a.wi = ps.W(L.twi); a.wo = ps.W(L.two);
  a.dc = S.dc + (size_t)R.row0 * H; a.len = R.len; a.ldm = L.ldP; a.P = L.P; a.t = t; a.N = R.N; a.H = H;
  a.nblk_c = (H + bwd_a_cells() - 1) / bwd_a_cells();
  a.drop = R.drop; a.drop.tag += (unsigned long long)t;
}
static void fill_bwd_b(BwdBJob& b, const LayerRun& R, int t, bool with_dx) {
  const LstmLayer& L = *R.L; const LstmStash& S = *R.S; const ParamSet& ps = *R.ps;
  const int H4 = 4 * L.H;
  const size_t r = (size_t)t * R.Ns + R.row0;
  b.dz = S.gates + r * H4; b.K = ps.W(L.tK); b.Ksw = with_dx ? L.Kb_full : L.Kb_rec;
  b.dx = with_dx ? R.din + r * L.ldI : nullptr;
  b.dmst = S.dmst + (size_t)R.row0 * L.ldP; b.len = R.len;
  b.I = L.I; b.n_begin = with_dx ? 0 : L.I; b.n_end = L.I + L.P; b.lddx = L.ldI; b.ldm = L.ldP; b.t = t; b.N = R.N; b.H4 = H4;
  b.dx_accumulate = R.din_accumulate ? 1 : 0;
  b.nblk_c = (b.n_end - b.n_begin + 15) / 16;
}

int Model::gates_blocks(int H, int N) const { return job_blocks((H + 15) / 16, N, fwd_gates_rows()); }
int Model::proj_blocks(int P, int N) const { return job_blocks((P + 15) / 16, N); }
int Model::bwd_a_blocks(int H, int N) const { return job_blocks((H + bwd_a_cells() - 1) / bwd_a_cells(), N); }
static void run_gates(const FwdGateJobs& gj, int blocks, int kb, hipStream_t s) { launch_fwd_gates(gj, blocks, kb, s); }
static void run_proj(const FwdProjJobs& pj, int blocks, int kb, hipStream_t s) { launch_fwd_proj(pj, blocks, kb, s); }
static void run_bwd_a(const BwdAJobs& aj, int blocks, int kb, hipStream_t s) { launch_bwd_a(aj, blocks, kb, s); }
void Model::run_bwd_b_splitk(BwdBJobs& bj, hipStream_t s) { bwd_b_plan(bj, bwdb_ws); launch_bwd_b_splitk(bj, s); }

void Model::gates_launch(const FwdGateJobs& gj, int blocks, int kb, hipStream_t s) {
  if (!prof_on) { run_gates(gj, blocks, kb, s); return; }
  if ((size_t)(2 * prof_n + 2) > prof_ev.size()) {
    const size_t old = prof_ev.size();
    prof_ev.resize(old + 128, nullptr);
    for (size_t i = old; i < prof_ev.size(); ++i) (void)hipEventCreate(&prof_ev[i]);
  }
  for (int i = 0; i < gj.n; ++i) {
    const FwdGateJob& J = gj.j[i];
    prof_flops += 2.0 * J.N * ((J.x ? J.ldx : 0) + J.ldm) * 4.0 * J.H;
  }
  (void)hipEventRecord(prof_ev[2 * prof_n], s);
  run_gates(gj, blocks, kb, s);
  (void)hipEventRecord(prof_ev[2 * prof_n + 1], s);
  ++prof_n;
}

void Model::rnn_forward(std::vector<Chain>& chains, int T, hipStream_t s, const std::vector<int>* offsets,
                        const std::vector<FcStage>* fcs) {
  if (lazy_sw) { refresh_swizzles(RSRGAN_NET_G, s); refresh_swizzles(RSRGAN_NET_D, s); }      // (from the current variables, every time)
  // zero initial state (cell.zero_state, models/lstm.py:107): slot 0 of c / m for the rows of each run
  {
    ZeroList zl{};
    for (auto& ch : chains)
      for (auto& R : ch) {
        if (zl.n + 2 > 32) { launch_zero_many(zl, s); zl.n = 0; }
        zl.p[zl.n] = R.S->c + (size_t)R.row0 * R.L->H; zl.len[zl.n++] = (unsigned)((size_t)R.N * R.L->H);
        zl.p[zl.n] = R.S->mst + (size_t)R.row0 * R.L->ldP; zl.len[zl.n++] = (unsigned)((size_t)R.N * R.L->ldP);
      }
    launch_zero_many(zl, s);
  }
  auto zx_gemm = [&](const LayerRun& R) {     // x-part of every step: zx = in . K[0:I] + bias, batched over T*N frames
    const int H4 = 4 * R.L->H;
    gemm(R.in, R.L->ldI, true, R.ps->W(R.L->tK), H4, false, R.S->gates, H4, T * R.N, H4, R.L->I, R.ps->W(R.L->tb), 0, 0.f, false, s);
  };
  if (!wavefront()) {
    for (auto& ch : chains)
      for (auto& R : ch) {
        const bool zx = R.Ns == R.N && R.row0 == 0;          // rows contiguous over time -> batch the x-part
        if (zx) zx_gemm(R);
        for (int t = 0; t < T; ++t) {
          FwdGateJobs gj{}; gj.n = 1; gj.forget_bias = cfg.forget_bias;
          fill_gate(gj.j[0], R, t, zx);
          gates_launch(gj, gates_blocks(R.L->H, R.N), (zx ? 0 : kb16(R.L->ldI)) + kb16(R.L->ldP), s);
          if (R.L->has_proj) {
            FwdProjJobs pj{}; pj.n = 1;
            fill_proj(pj.j[0], R, t);
            run_proj(pj, proj_blocks(R.L->P, R.N), kb16(R.L->ldH), s);
          }
        }
      }
    return;
  }
  int last = 0;      // last diagonal with work
  for (size_t c = 0; c < chains.size(); ++c) {
    const int off = offsets ? (*offsets)[c] : 0;
    last = std::max(last, off + (int)chains[c].size() - 1 + T - 1);
    for (auto& R : chains[c])
      if (R.zx_batched) zx_gemm(R);
  }
  if (fcs)
    for (auto& F : *fcs) last = std::max(last, F.offset + T - 1);
  for (int d = 0; d <= last; ++d) {
    FwdGateJobs gj{}; gj.forget_bias = cfg.forget_bias;
    FwdProjJobs pj{};
    int gb = 0, pb = 0, gk = 0, pk = 0;
    // a diagonal's jobs only depend on earlier diagonals, so they may be split over several launches
    auto flush_g = [&]() { if (gj.n) gates_launch(gj, gb, gk, s); gj.n = 0; gb = gk = 0; };
    auto flush_p = [&]() { if (pj.n) run_proj(pj, pb, pk, s); pj.n = 0; pb = pk = 0; };
    for (size_t c = 0; c < chains.size(); ++c) {
      Chain& ch = chains[c];
      const int off = offsets ? (*offsets)[c] : 0;
      for (int l = (int)ch.size() - 1; l >= 0; --l) {     // upper layers first: their K is twice layer 0's (heavy blocks dispatch first)
        const int t = d - off - l;
        if (t < 0 || t >= T) continue;
        const LayerRun& R = ch[l];
        if (gj.n == MAXJ) flush_g();
        FwdGateJob& a = gj.j[gj.n++]; fill_gate(a, R, t, R.zx_batched); gb += gates_blocks(R.L->H, R.N);
        gk = std::max(gk, (R.zx_batched ? 0 : kb16(R.L->ldI)) + kb16(R.L->ldP));
      }
    }
    flush_g();
    for (size_t c = 0; c < chains.size(); ++c) {
      Chain& ch = chains[c];
      const int off = offsets ? (*offsets)[c] : 0;
      for (int l = (int)ch.size() - 1; l >= 0; --l) {
        const int t = d - off - l;
        if (t < 0 || t >= T) continue;
        const LayerRun& R = ch[l];
        if (!R.L->has_proj) continue;
        if (pj.n == MAXJ) flush_p();
        FwdProjJob& p = pj.j[pj.n++]; fill_proj(p, R, t); pb += proj_blocks(R.L->P, R.N);
        pk = std::max(pk, kb16(R.L->ldH));
      }
    }
    if (fcs)
      for (auto& F : *fcs) {
        const int t = d - F.offset;
        if (t < 0 || t >= T) continue;
        if (pj.n == MAXJ) flush_p();
        FwdProjJob& p = pj.j[pj.n++]; fill_fc_fwd(p, F, t); pb += proj_blocks(F.D, F.N);
        pk = std::max(pk, kb16(F.ld_in));
      }
    flush_p();
  }
}

// Kf_l = [ Kx' ; Wp_l . Kh_l ] with Kx' = K_0[0:I] (layer 0) or Wp_{l-1} . K_l[0:I_l] (I_l = P_{l-1}), then its fragment-tiled copies
void Model::refresh_fold(hipStream_t s) {
  SwizzleList sl{};
  for (size_t l = 0; l < dl.size(); ++l) {
    const LstmLayer& L = dl[l]; const LstmLayer& F = dl_fold[l];
    const int H4 = 4 * L.H;
    const float* K = D.W(L.tK);
    float* Kf = dl_fold_K[l];
    if (l == 0) (void)hipMemcpyAsync(Kf, K, (size_t)L.I * H4 * sizeof(float), hipMemcpyDeviceToDevice, s);
    else gemm(D.W(dl[l - 1].tWp), dl[l - 1].ldP, true, K, H4, false, Kf, H4, dl[l - 1].H, H4, L.I, nullptr, 0, 0.f, false, s);
    gemm(D.W(L.tWp), L.ldP, true, K + (size_t)L.I * H4, H4, false, Kf + (size_t)F.I * H4, H4, L.H, H4, L.P, nullptr, 0, 0.f, false, s);
    const int ncb = (F.H + 15) / 16;
    sl.j[sl.n++] = SwizzleJob{Kf, F.Wg_full, H4, 4, F.H, 0, 0, 0, F.I, F.P, F.ldI, 4 * ncb, (F.ldI + F.ldP + 15) / 16, 0};
    sl.j[sl.n++] = SwizzleJob{Kf, F.Wg_h, H4, 4, F.H, 0, 0, 0, F.I, F.P, 0, 4 * ncb, (F.ldP + 15) / 16, 0};
  }
  launch_swizzle_many(sl, s);
}

// The forward recurrence of a discriminator chain running alone, one launch per time step (folded cells, model.h), then the
// masked outputs of every layer as time-batched GEMMs.  The carried projection state mst is NOT produced (only the weight
// gradients read it, and this path serves the runs that do not train the discriminator).
bool Model::fold_forward(Chain& ch, int T, hipStream_t s) {
  if (dl_fold.empty() || !wavefront() || ch.size() != dl_fold.size()) return false;
  Chain fch;
  for (size_t l = 0; l < ch.size(); ++l) {
    const LayerRun& R = ch[l];
    if (R.L != &dl[l] || R.S != &d_st[l] || R.res_in || R.res_out || R.zx_batched || R.want_wgrads || R.row0 != 0 || R.Ns != R.N) return false;
    LayerRun Rf = R;
    Rf.L = &dl_fold[l]; Rf.S = &d_fold_st[l];
    if (l > 0) Rf.in = d_st[l - 1].h;                  // the masked h of the layer below (0 where t >= len)
    fch.push_back(Rf);
  }
  refresh_fold(s);                                   // from the current variables, every time: nothing can go stale, capture or not
  std::vector<Chain> chains{fch};
  const bool prof_was = prof_on;
  prof_on = false;                                   // bench.py's dominant-kernel timing covers the merged forward wave's launches only
  rnn_forward(chains, T, s);                         // (these launches execute re-associated products: K is not the algorithmic K)
  prof_on = prof_was;
  for (size_t l = 0; l < ch.size(); ++l) {           // out_t = h_t . Wp  (h_t = 0 on masked rows: dynamic_rnn's zero output)
    const LstmLayer& L = dl[l];
    gemm(d_st[l].h, L.ldH, true, D.W(L.tWp), L.ldP, false, d_st[l].out, L.ldP, T * ch[l].N, L.P, L.H, nullptr, 0, 0.f, false, s);
  }
  return true;
}

// The same chain as ONE persistent launch (dpersist.hip): layer 0's x-part batched over time first, everything else in the kernel.
// Produces the complete stash (gates, c, h, mst, out), unlike fold_forward.
bool Model::persist_forward(Chain& ch, int T, hipStream_t s) {
  if (!dp_gran || !(dp_env & 1) || !wavefront() || ch.size() != dl.size()) return false;
  DPersistArgs a{};
  a.nl = (int)ch.size(); a.N = ch[0].N; a.T = T; a.H = dl[0].H; a.len = ch[0].len;
  a.gran = dp_gran; a.ctl = dp_ctl; a.forget_bias = cfg.forget_bias;
  for (size_t l = 0; l < ch.size(); ++l) {
    const LayerRun& R = ch[l]; const LstmLayer& L = dl[l]; const LstmStash& S = d_st[l];
    if (R.L != &L || R.S != &S || R.res_in || R.res_out || R.row0 != 0 || R.Ns != R.N || R.N != a.N || !L.has_proj || L.H != a.H) return false;
    if (l > 0 && R.in != d_st[l - 1].out) return false;
    DPersistLayer& D_ = a.L[l];
    D_.K = D.W(L.tK); D_.bias = D.W(L.tb); D_.wi = D.W(L.twi); D_.wf = D.W(L.twf); D_.wo = D.W(L.two); D_.Wp = D.W(L.tWp);
    D_.gates = S.gates; D_.c = S.c; D_.h = S.h; D_.mst = S.mst; D_.out = S.out;
    D_.I = L.I; D_.P = L.P; D_.ldP = L.ldP; D_.ldH = L.ldH; D_.ldI = L.ldI;
    D_.in = l == 0 ? R.in : nullptr;                   // (layer 0's input product runs inside the launch as well)
  }
  if (!dpersist_supported(a) || dpersist_grid(a.nl, a.N) > dp_max_grid || dpersist_granule_bytes(a.nl, a.N, a.T) > dp_gran_bytes) return false;
  static const bool fwd_t_env = [] { const char* e = getenv("RSRGAN_DFWD_T"); return e && atoi(e) != 0; }();
  if (fwd_t_env && a.N % 32 == 0) launch_dlstm_fwd_t(a, s); else launch_dlstm_fwd(a, s);
  return true;
}

// The generator's stack as ONE persistent launch (gpersist.hip).  Same stash as the wavefront launches leave (gates, c, h, mst, out of
// every layer), so the backward pass does not know which forward ran.
// the discriminator's halves of the fused launches: the second tile of the (only) tile pair is padding (DPersistArgs::nrt; RSRGAN_DP_NRT=0: it runs)
int Model::trail_nrt() const {
  static const bool on = [] {
    const char* e = getenv("RSRGAN_DP_NRT"); const char* g = getenv("RSRGAN_GP_NRT");
    return (!e || atoi(e) != 0) && (!g || atoi(g) != 0);             // (only beside a generator that drops the tile as well)
  }();
  // (and only when every recurrence of both nets runs persistent: see gpersist_shape's note on the padding rows of the stash)
  return on && B == 32 && Bt <= 16 && (gp_env & 3) == 3 && (dp_env & 3) == 3 ? 1 : 0;
}
bool Model::gpersist_shape(GPersistArgs& a, int T) const {              // (sizes only: usable before any buffer exists)
  static const bool res_env = [] { const char* e = getenv("RSRGAN_GP_RES"); return !e || atoi(e) != 0; }();
  const bool res = cfg.g_type == RSRGAN_G_RES_LSTM_L && res_env;          // the running residual sum rides the hand-offs (gpersist.hip RES)
  // (res_lstm_base, models/res_lstm_base.py:101-139: the same stack of projected cells fed the input frames directly, no sums)
  if (!gp_env || gl.empty() || gl.size() > (size_t)GP_MAXL || (cfg.g_type != RSRGAN_G_LSTM && cfg.g_type != RSRGAN_G_RES_LSTM_BASE && !res)) return false;
  a = GPersistArgs{};
  a.nl = (int)gl.size(); a.N = B; a.T = T; a.H = gl[0].H; a.res = res ? 1 : 0;
  bool noproj = !gl[0].has_proj;
  // ring slots tagged with the parity of the ring pass instead of re-armed with sentinels (gpersist.hip gp_store_t); RSRGAN_GP_TAGS=0: the sentinel form
  static const bool tags_env = [] { const char* e = getenv("RSRGAN_GP_TAGS"); return !e || atoi(e) != 0; }();
  a.tags = tags_env && !noproj ? 1 : 0;
  for (size_t l = 0; l < gl.size(); ++l) {
    const LstmLayer& L = gl[l];
    if (L.has_proj == noproj || L.H != a.H) return false;
    GPersistLayer& G_ = a.L[l];
    G_.I = L.I; G_.P = L.P; G_.ldI = L.ldI; G_.ldP = L.ldP; G_.ldH = L.ldH;
  }
  // num_proj=None (BASELINE.json's 2 x 512 generator): the single-hop form, forward only (gpersist.hip np_fwd_body)
  static const bool np_env = [] { const char* e = getenv("RSRGAN_GP_NOPROJ"); return !e || atoi(e) != 0; }();
  if (noproj) return np_env && !res && (gp_env & 1) && gpersist_np_plan(a, gp_np_nt);
  if (!gpersist_plan(a)) return false;
  // a padded model whose real rows fit one 16-row tile (the shipped batch_size = 8, decode's single utterance): the other tile of the
  // row group holds padding rows only -- length 0 in every batch, zeros in every stash since the allocation -- and does not run
  // (GPersistArgs::nrt).  RSRGAN_GP_NRT=0: both tiles run.
  static const bool nrt_env = [] { const char* e = getenv("RSRGAN_GP_NRT"); return !e || atoi(e) != 0; }();
  // (only when BOTH recurrences run persistent: a launch-path forward writes gate activations into the padding rows of the stash,
  // which a one-lane BPTT would never turn into dz = 0 -- the weight-gradient products would read them as dZ)
  if (nrt_env && B == 32 && Bt <= 16 && (gp_env & 3) == 3) a.nrt = 1;
  // the forward launch's off-chain work (the X waves' products, the stash stores) behind the lane's publication instead of beside it
  // (GPersistArgs::sched; pays with two row groups on the fabric: 13.5 -> 13.0 us per step at 64 rows, nothing at 32).  RSRGAN_GP_SCHED=0..3
  static const int sched_env = [] { const char* e = getenv("RSRGAN_GP_SCHED"); return e ? atoi(e) : -1; }();
  a.sched = sched_env >= 0 ? sched_env : (B >= 64 ? 3 : 0);
  return true;
}
bool Model::gpersist_args(GPersistArgs& a, int T) const {
  if (!gpersist_shape(a, T)) return false;
  a.len = len_dev;
  a.gran1 = gp_gran1; a.gran2 = gp_gran2; a.gran3 = gp_gran3; a.ctl = gp_ctl; a.forget_bias = cfg.forget_bias;
  for (size_t l = 0; l < gl.size(); ++l) {
    const LstmLayer& L = gl[l]; const LstmStash& S = g_st[l];
    GPersistLayer& G_ = a.L[l];
    G_.KxT = L.KxT; G_.KhT = L.KhT; G_.bias = G.W(L.tb); G_.wi = G.W(L.twi); G_.wf = G.W(L.twf); G_.wo = G.W(L.two); G_.Wp = L.has_proj ? G.W(L.tWp) : nullptr;
    G_.gates = S.gates; G_.c = S.c; G_.h = S.h; G_.mst = S.mst; G_.out = S.out; G_.dmt = S.dmt;
    G_.res_out = a.res ? g_res[l] : nullptr;
  }
  return true;
}

// A persistent launch reported a failed bounded wait (rsrgan_device_status): its workgroups were not all resident -- somebody else's
// work on the device, CUs taken away after rsrgan_create.  Another attempt would spin into the same time-out on every step: this
// handle takes the launch-per-phase path from now on (the captured graphs hold the persistent launches: dropped).
void Model::persist_disable(int which) {
  if (which == 1) gp_env = 0; else dp_env = 0;
  lazy_sw = false;
  trail_fits = false;
  dpipe = false;
  drop_graphs();
  refresh_swizzles(RSRGAN_NET_G, nullptr);
  refresh_swizzles(RSRGAN_NET_D, nullptr);
  // (on the null stream, which orders nothing against the non-blocking streams the next call works on: rsrgan_device_status, the only
  // caller, is a synchronising call anyway)
  (void)hipStreamSynchronize(nullptr);
}

void Model::gpersist_rearm() {
  GPersistArgs a{};
  if (!gp_gran1 || !gpersist_args(a, gp_Tcap)) return;
  if (gp_noproj) {                                                       // (the unprojected form: only its BPTT has rings)
    if (gp_gran3) { gpersist_np_arm(a, 0); (void)hipDeviceSynchronize(); }
    return;
  }
  gpersist_arm(a, 0);
  (void)hipDeviceSynchronize();
}

// A stack planned as one launch per row group (GPersistArgs::ngl: res_lstm_l at 64 rows does not fit the device at once): the launches
// one after the other; `fused` (k_glstm_fwd_dt / k_glstm_bwd_dt with the discriminator's half over ALL rows) rides the launch of group
// `fused_at` -- the LAST forward launch (the FC workgroups read the earlier groups' chunks, which stay in their slots), the FIRST
// backward launch (the later groups' launches find d(outputs) complete and do not poll).
static void glstm_fwd_groups(GPersistArgs a, const DPersistArgs* d, hipStream_t s) {
  const int ngr = a.N / GP_ROWS;
  if (!a.ngl) { if (d) launch_glstm_fwd_dt(a, *d, s); else launch_glstm_fwd(a, s); return; }
  for (int g = 0; g < ngr; ++g) {
    a.grp0 = g;
    if (d && g == ngr - 1) launch_glstm_fwd_dt(a, *d, s); else launch_glstm_fwd(a, s);
  }
}
static void glstm_bwd_groups(GPersistArgs a, const DPersistArgs* d, hipStream_t s) {
  const int ngr = a.N / GP_ROWS;
  if (!a.ngl) { if (d) launch_glstm_bwd_dt(a, *d, s); else launch_glstm_bwd(a, s); return; }
  for (int g = 0; g < ngr; ++g) {
    a.grp0 = g;
    if (d && g == 0) { a.dout_trail = 1; launch_glstm_bwd_dt(a, *d, s); } else { a.dout_trail = 0; launch_glstm_bwd(a, s); }
  }
}

bool Model::persist_forward_g(int T, hipStream_t s) {
  if (!gp_fwd_on() || !wavefront() || seq_drop_on()) return false;
  GPersistArgs a{};
  if (!gpersist_args(a, T) || (gp_noproj ? gpersist_np_gran2_bytes(a) : gpersist_gran2_bytes(a)) > gp_gran2_bytes) return false;
  a.L[0].in = g_ins[0];                                // (layer 0's input product runs inside the launch as well)
  if (gp_noproj) {                                     // num_proj=None: the single-hop form (no event bracket: bench.py's dominant-kernel timing is the projected form's)
    for (size_t l = 0; l < gl.size(); ++l) a.L[l].Wp = nullptr;
    launch_glstm_np_fwd(a, s);
    return true;
  }
  if (prof_on) {
    if ((size_t)(2 * prof_gp_n + 2) > prof_gp_ev.size()) {
      const size_t old = prof_gp_ev.size();
      prof_gp_ev.resize(old + 8, nullptr);
      for (size_t i = old; i < prof_gp_ev.size(); ++i) (void)hipEventCreate(&prof_gp_ev[i]);
    }
    // algorithmic FLOP of the launch: every layer's input and recurrent product and its projection
    for (size_t l = 0; l < gl.size(); ++l)
      prof_gp_flops += 2.0 * Bt * T * ((double)(gl[l].I + gl[l].P) * 4.0 * gl[l].H + (double)gl[l].H * gl[l].P);
    (void)hipEventRecord(prof_gp_ev[2 * prof_gp_n], s);
    glstm_fwd_groups(a, nullptr, s);
    (void)hipEventRecord(prof_gp_ev[2 * prof_gp_n + 1], s);
    ++prof_gp_n;
    return true;
  }
  glstm_fwd_groups(a, nullptr, s);
  return true;
}

// The generator's forward recurrence with D(G(x)) a few steps behind it as ONE launch (gpersist.hip k_glstm_fwd_dt): the G-run of a
// schedule that recomputes the generator's forward (gen_updates > 1: run_gan_rnn_placeholder.sh:130): y, the discriminator's input rows
// (y + noise) and both stashes are complete behind it.  False: not applicable (the caller runs the launches one after the other).
// D(real) of the D-run as a launch of its own over rows [0, B) of the stacked stash (RSRGAN_DPIPE), with granules and control block of its own
bool Model::persist_forward_real(int T, hipStream_t q, bool check_only) {
  if (!dpipe || !dp_gran2 || !(dp_env & 1) || !wavefront()) return false;
  Chain ch = d_chain(B, 2 * B, 0);
  DPersistArgs a{};
  a.nl = (int)ch.size(); a.N = B; a.T = T; a.H = dl[0].H; a.len = ch[0].len; a.Ns = 2 * B; a.row0 = 0;
  a.gran = dp_gran2; a.ctl = dp_ctl2; a.forget_bias = cfg.forget_bias;
  for (size_t l = 0; l < ch.size(); ++l) {
    const LstmLayer& L = dl[l]; const LstmStash& S = d_st[l];
    if (!L.has_proj || L.H != a.H) return false;
    DPersistLayer& D_ = a.L[l];
    D_.K = D.W(L.tK); D_.bias = D.W(L.tb); D_.wi = D.W(L.twi); D_.wf = D.W(L.twf); D_.wo = D.W(L.two); D_.Wp = D.W(L.tWp);
    D_.gates = S.gates; D_.c = S.c; D_.h = S.h; D_.mst = S.mst; D_.out = S.out;
    D_.I = L.I; D_.P = L.P; D_.ldP = L.ldP; D_.ldH = L.ldH; D_.ldI = L.ldI;
    D_.in = l == 0 ? xd : nullptr;
  }
  if (!dpersist_supported(a) || dpersist_grid(a.nl, a.N) > dp_max_grid || T > Tmax) return false;
  if (check_only) return true;
  launch_dlstm_fwd(a, q);
  return true;
}

bool Model::persist_forward_g_trail(Chain& ch, int T, hipStream_t s, const float* nf, bool check_only) {
  static const bool env = [] { const char* e = getenv("RSRGAN_TRAIL_FWD"); return !e || atoi(e) != 0; }();
  if (!env || !trail_fits || !gp_fwd_on() || gp_noproj || !wavefront() || seq_drop_on() || !dp_gran || !(dp_env & 1) || ch.size() != dl.size()) return false;
  GPersistArgs a{};
  if (!gpersist_args(a, T) || gpersist_gran2_bytes(a) > gp_gran2_bytes) return false;
  a.L[0].in = g_ins[0];
  a.fwd_trail = 1;
  DPersistArgs d{};
  d.nl = (int)ch.size(); d.N = ch[0].N; d.T = T; d.H = dl[0].H; d.len = ch[0].len;
  d.gran = dp_gran; d.ctl = dp_ctl; d.forget_bias = cfg.forget_bias;
  for (size_t l = 0; l < ch.size(); ++l) {
    const LayerRun& R = ch[l]; const LstmLayer& L = dl[l]; const LstmStash& S = d_st[l];
    // (rows [row0, row0 + N) of a stash Ns rows tall: the D-run's D(G(x)) writes the second half of the stacked stash)
    if (R.L != &L || R.S != &S || R.res_in || R.res_out || R.Ns != ch[0].Ns || R.row0 != ch[0].row0 || R.N != d.N || !L.has_proj || L.H != d.H) return false;
    if (l > 0 && R.in != d_st[l - 1].out) return false;
    DPersistLayer& D_ = d.L[l];
    D_.K = D.W(L.tK); D_.bias = D.W(L.tb); D_.wi = D.W(L.twi); D_.wf = D.W(L.twf); D_.wo = D.W(L.two); D_.Wp = D.W(L.tWp);
    D_.gates = S.gates; D_.c = S.c; D_.h = S.h; D_.mst = S.mst; D_.out = S.out;
    D_.I = L.I; D_.P = L.P; D_.ldP = L.ldP; D_.ldH = L.ldH; D_.ldI = L.ldI;
    D_.in = l == 0 ? R.in : nullptr;
  }
  d.Ns = ch[0].Ns; d.row0 = ch[0].row0;
  if (ch[0].in != xd || d.N != B || B % 32 != 0 || dl[0].I != Dout || Dout % 4 != 0 || !dpersist_supported(d)) return false;
  if (dpersist_granule_bytes(d.nl + 1, d.N, d.T) / 2 > dp_gran_bytes) return false;      // (one edge per layer and one for layer 0's input)
  d.dy = y_tm; d.ld_dy = ldDout; d.fc_w = G.W(g_fc_out_w); d.ld_fcw = ldDout; d.fc_P = gR; d.fc_b = G.W(g_fc_out_b);
  d.noise = nf; d.dtop = xd; d.ld_dtop = ldDout; d.xd_Ns = ch[0].Ns; d.xd_row0 = ch[0].row0;
  d.nrt = trail_nrt();
  if (check_only) return true;
  if (prof_on) ++prof_fdt_n;
  glstm_fwd_groups(a, &d, s);
  g_fwd_valid = true;
  return true;
}

// BPTT through the generator's stack as ONE persistent launch (gpersist.hip k_glstm_bwd): dz over the gate activations of every
// layer's stash, dm per step in dmt.  Layer 0's input gradient (the input FC's d(h0)) is one GEMM over the dz stash afterwards.
bool Model::persist_backward_g(Chain& ch, int T, hipStream_t s, bool check_only, const StreamFn& pre, const StreamFn& post) {
  if (!gp_gran1 || !gp_gran3 || !(gp_env & 2) || !wavefront() || seq_drop_on() || ch.size() != gl.size()) return false;
  GPersistArgs a{};
  if (!gpersist_args(a, T) || (gp_noproj ? gpersist_np_gran2_bytes(a) : gpersist_gran2_bytes(a)) > gp_gran2_bytes) return false;
  if (gp_noproj) {
    // num_proj=None (gpersist.hip k_glstm_np_bwd): dz over the gate activations of every layer's stash; no input gradient for layer 0
    if (a.NT != 2 || a.res) return false;
    for (size_t l = 0; l < ch.size(); ++l) {
      const LayerRun& R = ch[l];
      if (R.L != &gl[l] || R.S != &g_st[l] || R.res_in || R.res_out || R.row0 != 0 || R.Ns != R.N || R.N != a.N || R.len != a.len) return false;
      if (l > 0 && (R.din_accumulate || ch[l].din != ch[l - 1].dout)) return false;
    }
    if (ch[0].din) return false;
    a.dout_top = ch.back().dout; a.ld_dout = gl.back().ldP;
    if (!a.dout_top) return false;
    if (check_only) return true;
    launch_glstm_np_bwd(a, s);
    if (!defer_wgrads) chain_wgrads(ch, T, s, nullptr, pre, post);
    else { if (pre) pre(s); if (post) post(s); }
    return true;
  }
  for (size_t l = 0; l < ch.size(); ++l) {
    const LayerRun& R = ch[l];
    if (R.L != &gl[l] || R.S != &g_st[l] || R.row0 != 0 || R.Ns != R.N || R.N != a.N || R.len != a.len) return false;
    if (!a.res) {
      if (R.res_in || R.res_out) return false;
      if (l > 0 && (R.din_accumulate || ch[l].din != ch[l - 1].dout)) return false;
    } else {
      // res_lstm_l: the callers keep d(inputs_l) = dx_l + d(inputs_{l+1}) accumulated in ONE buffer (dout == din, accumulate); inside the
      // launch that sum travels from reducer to reducer, the buffer is only read as the top layer's d(outputs)
      if (R.res_in != g_ins[l] || R.res_out != g_res[l] || R.dout != ch.back().dout) return false;
      if (l > 0 ? (!R.din_accumulate || R.din != R.dout) : R.din != nullptr) return false;
    }
  }
  if (!a.res && ch[0].din && ch[0].din_accumulate) return false;
  a.dout_top = ch.back().dout; a.ld_dout = gl.back().ldP;
  if (!a.dout_top) return false;
  if (check_only) return true;
  // d(inputs of layer 0) = dZ_0 . K_x^T: a time-batched GEMM behind the launch (126 us).  RSRGAN_GP_DIN0=1 runs it inside the launch
  // (layer 0's X waves publish partials to ring 0 of gran3, its reducers sum them a step late): built, bit-stable, parity-green --
  // and 0.2 ms per step SLOWER (k_glstm_bwd 16.7 -> 19.4 us per step: a third more hand-off traffic and MFMA bursts on the layer
  // every other layer waits for), so it stays off.
  static const bool din0_env = [] { const char* e = getenv("RSRGAN_GP_DIN0"); return e && atoi(e) != 0; }();
  const bool din_inside = ch[0].din && din0_env && (gl[0].I + 15) / 16 <= (gl[0].P + 15) / 16 && gl[0].ldI % 4 == 0;
  if (din_inside) { a.din0 = ch[0].din; a.ld_din0 = gl[0].ldI; }
  a.dout_trail = gp_trail_next ? 1 : 0;
  if (gp_trail_next && getenv("RSRGAN_TRAIL_DBG")) a.dout_trail = atoi(getenv("RSRGAN_TRAIL_DBG"));
  if (gp_phase == 2) {
  } else if (prof_on) {
    if ((size_t)(2 * prof_gb_n + 2) > prof_gb_ev.size()) {
      const size_t old = prof_gb_ev.size();
      prof_gb_ev.resize(old + 8, nullptr);
      for (size_t i = old; i < prof_gb_ev.size(); ++i) (void)hipEventCreate(&prof_gb_ev[i]);
    }
    // algorithmic FLOP of the launch: every layer's state-gradient product and dh = dm . W_p^T, the input-gradient product (layer 0's
    // only when it runs inside the launch)
    for (size_t l = 0; l < gl.size(); ++l)
      prof_gb_flops += 2.0 * Bt * T * ((double)((l || din_inside ? gl[l].I : 0) + gl[l].P) * 4.0 * gl[l].H + (double)gl[l].H * gl[l].P);
    if (gp_trail_next) {      // k_glstm_bwd_dt: + the discriminator's data gradient (every layer's state- and input-gradient product, dh = dm . W_p^T) and dy . W_out^T
      for (size_t l = 0; l < dl.size(); ++l)
        prof_gb_flops += 2.0 * Bt * T * ((double)(dl[l].I + dl[l].P) * 4.0 * dl[l].H + (double)dl[l].H * dl[l].P);
      prof_gb_flops += 2.0 * Bt * T * (double)Dout * gR;
    }
    (void)hipEventRecord(prof_gb_ev[2 * prof_gb_n], s);
    glstm_bwd_groups(a, gp_trail_next ? &dt_args : nullptr, s);
    (void)hipEventRecord(prof_gb_ev[2 * prof_gb_n + 1], s);
    ++prof_gb_n;
  } else
    glstm_bwd_groups(a, gp_trail_next ? &dt_args : nullptr, s);
  if (gp_phase == 1) return true;
  auto din0 = [&](hipStream_t q) {
    if (ch[0].din && !din_inside) {
      const LayerRun& R = ch[0];
      const int H4 = 4 * R.L->H;
      gemm(R.S->gates, H4, true, R.ps->W(R.L->tK), H4, true, R.din, R.L->ldI, T * R.N, R.L->I, H4, nullptr, 0, 0.f, false, q);
    }
  };
  if (!defer_wgrads) chain_wgrads(ch, T, s, din0, pre, post);
  else { if (pre) pre(s); din0(s); if (post) post(s); }
  return true;
}

// BPTT through a discriminator chain running alone as ONE persistent launch (dpersist.hip k_dlstm_bwd), then the weight-gradient
// GEMMs over the dz it leaves in the stash.  No input gradient for layer 0 (the D-run does not need one).
bool Model::persist_backward(Chain& ch, int T, hipStream_t s) {
  if (!dp_gran || !(dp_env & 2) || !wavefront() || ch.size() != dl.size() || ch[0].din) return false;
  DPersistArgs a{};
  a.nl = (int)ch.size(); a.N = ch[0].N; a.T = T; a.H = dl[0].H; a.len = ch[0].len;
  a.gran = dp_gran; a.ctl = dp_ctl; a.forget_bias = cfg.forget_bias;
  for (size_t l = 0; l < ch.size(); ++l) {
    const LayerRun& R = ch[l]; const LstmLayer& L = dl[l]; const LstmStash& S = d_st[l];
    if (R.L != &L || R.S != &S || R.res_in || R.res_out || R.row0 != 0 || R.Ns != R.N || R.N != a.N || !L.has_proj || L.H != a.H) return false;
    if (l > 0 && R.in != d_st[l - 1].out) return false;
    DPersistLayer& D_ = a.L[l];
    D_.K = D.W(L.tK); D_.bias = D.W(L.tb); D_.wi = D.W(L.twi); D_.wf = D.W(L.twf); D_.wo = D.W(L.two); D_.Wp = D.W(L.tWp);
    D_.gates = S.gates; D_.c = S.c; D_.h = S.h; D_.mst = S.mst; D_.out = S.out; D_.dmt = S.dmt;
    D_.I = L.I; D_.P = L.P; D_.ldP = L.ldP; D_.ldH = L.ldH; D_.ldI = L.ldI; D_.in = R.in;
  }
  a.dout_top = ch.back().dout; a.ld_dout = dl.back().ldP;
  if (!a.dout_top || !dpersist_supported(a) || dpersist_grid(a.nl, a.N) > dp_max_grid || dpersist_granule_bytes(a.nl, a.N, a.T) > dp_gran_bytes) return false;
  // the weight gradients ride the launch (dp_dw_body) when every layer wants them, for the stacked call the records were sized for
  bool dw = dw_ws && !defer_wgrads && a.N == 2 * B && T <= Tmax;
  for (auto& R : ch) dw = dw && R.want_wgrads && R.in && R.ps == &D;
  if (dw) { a.dw_ws = dw_ws; a.dw_flag = dw_flag; a.dw_stride = dw_stride; }
  launch_dlstm_bwd(a, s);
  if (dw) {
    launch_dw_reduce(dw_ws, dw_stride, a.N / 16, dw_src, D.g, D.ct, D.partial, s);
    d_partial_fresh = true;
    return true;
  }
  if (!defer_wgrads) {           // (on one stream: the discriminator's sequences are short launches, two streams cost them 0.04 ms)
    bool dK_done = false;
    const bool rest = batch_wgrads(ch, T, s, true, &dK_done);
    for (auto& R : ch)
      if (R.want_wgrads) {
        if (!dK_done || !rest) layer_wgrads_gemms(R, 0, T, false, s, !dK_done, !rest);
        if (!rest) layer_wgrads_colsums(R, T, s, (side && s == side) ? scratch2 : scratch);
      }
  }
  return true;
}

bool Model::persist_backward_trail(Chain& ch, int T, hipStream_t s, float* dy, int ld_dy, float* dtop, int ld_dtop, bool check_only) {
  if (!trail_fits || !dp_gran || !(dp_env & 2) || !wavefront() || ch.size() != dl.size() || ch[0].din) return false;
  DPersistArgs a{};
  a.nl = (int)ch.size(); a.N = ch[0].N; a.T = T; a.H = dl[0].H; a.len = ch[0].len;
  a.gran = dp_gran; a.ctl = dp_ctl; a.forget_bias = cfg.forget_bias;
  for (size_t l = 0; l < ch.size(); ++l) {
    const LayerRun& R = ch[l]; const LstmLayer& L = dl[l]; const LstmStash& S = d_st[l];
    if (R.L != &L || R.S != &S || R.res_in || R.res_out || R.row0 != 0 || R.Ns != R.N || R.N != a.N || !L.has_proj || L.H != a.H || R.want_wgrads) return false;
    if (l > 0 && R.in != d_st[l - 1].out) return false;
    DPersistLayer& D_ = a.L[l];
    D_.K = D.W(L.tK); D_.bias = D.W(L.tb); D_.wi = D.W(L.twi); D_.wf = D.W(L.twf); D_.wo = D.W(L.two); D_.Wp = D.W(L.tWp);
    D_.gates = S.gates; D_.c = S.c; D_.h = S.h; D_.mst = S.mst; D_.out = S.out; D_.dmt = S.dmt;
    D_.I = L.I; D_.P = L.P; D_.ldP = L.ldP; D_.ldH = L.ldH; D_.ldI = L.ldI;
  }
  a.dout_top = ch.back().dout; a.ld_dout = dl.back().ldP;
  a.dy = dy; a.ld_dy = ld_dy; a.fc_w = G.W(g_fc_out_w); a.ld_fcw = ldDout; a.fc_P = gR; a.dtop = dtop; a.ld_dtop = ld_dtop;
  if (!a.dout_top || a.N != B || dl[0].I != Dout || !dpersist_trail_supported(a) || dpersist_granule_bytes(a.nl, a.N, a.T) > dp_gran_bytes) return false;
  if (check_only) return true;
  a.nrt = trail_nrt();
  dt_args = a;                                           // (persist_backward_g launches k_glstm_bwd_dt with it)
  return true;
}

void Model::layer_wgrads_gemms(const LayerRun& R, int t0, int t1, bool accumulate, hipStream_t s, bool do_dK, bool do_dWp) {
  const LstmLayer& L = *R.L; const LstmStash& S = *R.S; const ParamSet& ps = *R.ps;
  const int H = L.H, H4 = 4 * H, Rws = (t1 - t0) * R.N;      // needs Ns == N (rows contiguous over time)
  const size_t r0 = (size_t)t0 * R.N;
  float* dK = ps.Gd(L.tK);
  // dK[0:I] (+)= in^T . dZ ; dK[I:I+P] (+)= m_{t-1}^T . dZ ; dWp (+)= h^T . dm   over frames [t0, t1)
  if (!do_dK) {
  } else if (L.I % 4 == 0) {        // one GEMM over the stacked operand [x_t | m_{t-1}] (two source stashes, one output tensor)
    launch_gemm2(R.in + r0 * L.ldI, L.ldI, S.mst + r0 * L.ldP, L.ldP, L.I, false, S.gates + r0 * H4, H4, false, dK, H4,
                 L.I + L.P, H4, Rws, nullptr, 0, 0.f, accumulate, s, (side && s == side) ? gemm_ws2 : gemm_ws, gemm_ws_floats);
  } else {
    gemm(R.in + r0 * L.ldI, L.ldI, false, S.gates + r0 * H4, H4, false, dK, H4, L.I, H4, Rws, nullptr, 0, 0.f, accumulate, s);
    gemm(S.mst + r0 * L.ldP, L.ldP, false, S.gates + r0 * H4, H4, false, dK + (size_t)L.I * H4, H4, L.P, H4, Rws, nullptr, 0, 0.f, accumulate, s);
  }
  if (L.has_proj && do_dWp)
    gemm(S.h + r0 * L.ldH, L.ldH, false, S.dmt + r0 * L.ldP, L.ldP, false, ps.Gd(L.tWp), L.ldP, H, L.P, Rws, nullptr, 0, 0.f, accumulate, s);
}

// The layers of a stack have ONE shape more often than not (generator: 3 x (280 + 280 -> 760 / p280), discriminator: 2 x (40 + 40 -> 256 /
// p40)): their projection gradients h^T dm, their bias / peephole column sums and -- where the product is small enough for k_gemm16 --
// their kernel gradients [x | m]^T dZ run as one launch per kind for all layers (blockIdx.z = layer) instead of one per layer: these
// launches are latency-bound (the discriminator's six per layer left the chip idle for 0.26 ms per D-run).  Returns whether dWp and
// the column sums are done (*dK_done: the kernel gradients as well); the caller runs per layer what is not.
bool Model::batch_wgrads(Chain& ch, int T, hipStream_t s, bool dK_too, bool* dK_done, bool check_only) {
  static const bool on = [] { const char* e = getenv("RSRGAN_WGRAD_BATCH"); return !e || atoi(e) != 0; }();
  *dK_done = false;
  std::vector<const LayerRun*> rs;
  for (auto& R : ch) if (R.want_wgrads) rs.push_back(&R);
  if (!on || rs.size() < 2 || rs.size() > (size_t)GEMM16_MAXB) return false;
  const LstmLayer& L0 = *rs[0]->L;
  for (auto* R : rs) {
    const LstmLayer& L = *R->L;
    if (L.I != L0.I || L.P != L0.P || L.H != L0.H || L.ldI != L0.ldI || L.ldP != L0.ldP || L.ldH != L0.ldH || !L.has_proj ||
        R->N != rs[0]->N || R->Ns != R->N || R->row0 != 0)
      return false;
    // (an input width that is no multiple of 4 -- res_lstm_l's 257: the kernel gradient is the caller's, over the zero-padded ld columns
    //  into dk_tmp; the projection gradients and the column sums below do not care)
    if (L.I % 4 != 0 && (dK_too || !dk_tmp || R->ps != &G)) return false;
  }
  const int H = L0.H, H4 = 4 * H, Rws = T * rs[0]->N;
  if ((size_t)rs.size() * 64 * 7 * H > scratch_floats) return false;
  float* ws = (side && s == side) ? gemm_ws2 : gemm_ws;
  float* scr = (side && s == side) ? scratch2 : scratch;
  // (the rule of launch_gemm_mapped: only products with little work per tile run on k_gemm16)
  auto small = [&](int M, int N, int K) { const double outs = (double)M * N; return !(K >= 256 && outs >= 4.0e6) && !(K >= 2048 && outs >= 1.5e6); };
  if (check_only) { *dK_done = dK_too && small(L0.I + L0.P, H4, Rws); return small(H, L0.P, Rws); }      // (host-only: what a call would do)
  if (dK_too && small(L0.I + L0.P, H4, Rws)) {
    Gemm16Batch bt{}; bt.n = (int)rs.size();
    for (int p = 0; p < bt.n; ++p) { bt.A[p] = rs[p]->in; bt.A2[p] = rs[p]->S->mst; bt.B[p] = rs[p]->S->gates; bt.C[p] = rs[p]->ps->Gd(rs[p]->L->tK); }
    launch_gemm16_batch(bt, L0.ldI, L0.ldP, L0.I, H4, H4, L0.I + L0.P, H4, Rws, false, s, ws, gemm_ws_floats);
    *dK_done = true;
  }
  if (!small(H, L0.P, Rws)) return false;                     // (never for the nets of this repository: the caller runs the rest per layer)
  {
    Gemm16Batch bt{}; bt.n = (int)rs.size();
    for (int p = 0; p < bt.n; ++p) { bt.A[p] = rs[p]->S->h; bt.A2[p] = nullptr; bt.B[p] = rs[p]->S->dmt; bt.C[p] = rs[p]->ps->Gd(rs[p]->L->tWp); }
    launch_gemm16_batch(bt, L0.ldH, 0, 0, L0.ldP, L0.ldP, H, L0.P, Rws, false, s, ws, gemm_ws_floats);
  }
  {
    ColsumsBatch cb{}; cb.n = (int)rs.size();
    for (int p = 0; p < cb.n; ++p) {
      const LayerRun& R = *rs[p]; const LstmLayer& L = *R.L; const ParamSet& ps = *R.ps;
      cb.dz[p] = R.S->gates; cb.cprev[p] = R.S->c; cb.ccur[p] = R.S->c + (size_t)R.N * H;
      cb.db[p] = ps.Gd(L.tb); cb.dwi[p] = ps.Gd(L.twi); cb.dwf[p] = ps.Gd(L.twf); cb.dwo[p] = ps.Gd(L.two);
    }
    launch_lstm_colsums_batch(cb, Rws, H, scr, s);
  }
  return true;
}
void Model::layer_wgrads_colsums(const LayerRun& R, int T, hipStream_t s, float* scr) {
  const LstmLayer& L = *R.L; const LstmStash& S = *R.S; const ParamSet& ps = *R.ps;
  const int H = L.H, H4 = 4 * H, Rws = T * R.N;
  // bias: colsum(dZ); peepholes: dw_i = sum dai*c_{t-1}; dw_f = sum daf*c_{t-1}; dw_o = sum dao*c_t -- one pass over dZ
  (void)H4;
  launch_lstm_colsums(S.gates, S.c, S.c + (size_t)R.N * H, ps.Gd(L.tb), ps.Gd(L.twi), ps.Gd(L.twf), ps.Gd(L.two), Rws, H, scr, s);
}
void Model::layer_wgrads(const LayerRun& R, int T, hipStream_t s) {
  layer_wgrads_gemms(R, 0, T, false, s);
  layer_wgrads_colsums(R, T, s, (side && s == side) ? scratch2 : scratch);
}
// The weight gradients of a chain whose BPTT is complete (the persistent launches leave every layer's dz at once): the layers'
// launch sequences (dK GEMM + fix-up, dWp GEMM + reduce, the two column-sum kernels) do not depend on each other, and only the dK
// GEMM fills the chip -- the upper layers' sequences ride the side stream beside the lowest layer's (and `between`, the caller's
// launches that only need the BPTT), joined before returning.  RSRGAN_WGRAD_STREAMS=1: everything on s.
// `pre` / `post` (optional): launches of the caller that only need the BPTT's inputs / only `between`'s result (the output and input
// FCs' parameter gradients: short launches that used to follow the dK GEMMs): they ride the side stream too, `post` behind an event
// recorded on s right after `between`.
void Model::chain_wgrads(Chain& ch, int T, hipStream_t s, const StreamFn& between, const StreamFn& pre, const StreamFn& post) {
  static const int streams = [] { const char* e = getenv("RSRGAN_WGRAD_STREAMS"); return e ? atoi(e) : 2; }();
  int nw = 0;
  for (auto& R : ch) nw += R.want_wgrads ? 1 : 0;
  if (!side || streams < 2 || nw < 2) {
    if (pre) pre(s);
    if (between) between(s);
    if (post) post(s);
    for (auto& R : ch)
      if (R.want_wgrads) layer_wgrads(R, T, s);
    return;
  }
  // RSRGAN_DIN0_SIDE=1: `between` (the input gradient of layer 0 and what hangs on it) rides the side stream as well, the chip-filling
  // kernel-gradient GEMMs start at once on s
  static const bool between_side = [] { const char* e = getenv("RSRGAN_DIN0_SIDE"); return e && atoi(e) != 0; }();
  hipEvent_t ev = ev_pool[ev_next++ & 15];
  (void)hipEventRecord(ev, s);
  (void)hipStreamWaitEvent(side, ev, 0);
  if (pre) pre(side);
  auto after_between = [&]() {
    if (!post) return;
    hipEvent_t e2 = ev_pool[ev_next++ & 15];
    (void)hipEventRecord(e2, s);
    (void)hipStreamWaitEvent(side, e2, 0);
    post(side);
  };
  bool dK_done = false;
  if (between_side && between && post) {
    between(side); post(side);                                  // (first on the side stream: what hangs on it is the longest chain there)
    const bool rest = batch_wgrads(ch, T, side, false, &dK_done);
    std::vector<const LayerRun*> rs;
    for (auto& R : ch) if (R.want_wgrads) rs.push_back(&R);
    bool batched = false;
    if (rest && rs.size() >= 2 && rs.size() <= (size_t)GEMM_MAXB) {
      const float *A_[GEMM_MAXB], *A2_[GEMM_MAXB], *B_[GEMM_MAXB]; float* C_[GEMM_MAXB];
      const LstmLayer& L0 = *rs[0]->L;
      const bool padI = L0.I % 4 != 0;                     // (x over its ld columns into dk_tmp, rows copied behind: see model.h dk_tmp)
      const int Ie = padI ? L0.ldI : L0.I;
      for (size_t i = 0; i < rs.size(); ++i) { A_[i] = rs[i]->in; A2_[i] = rs[i]->S->mst; B_[i] = rs[i]->S->gates; C_[i] = padI ? dk_tmp + i * dk_tmp_per : rs[i]->ps->Gd(rs[i]->L->tK); }
      batched = launch_gemm_batch((int)rs.size(), A_, L0.ldI, A2_, L0.ldP, Ie, B_, 4 * L0.H, C_, 4 * L0.H, Ie + L0.P, 4 * L0.H, T * rs[0]->N, false, s, gemm_ws, gemm_ws_floats);
      if (batched && padI)
        for (size_t i = 0; i < rs.size(); ++i) {
          float* dK_ = rs[i]->ps->Gd(rs[i]->L->tK);
          launch_copy_f(C_[i], dK_, L0.I * 4 * L0.H, s);
          launch_copy_f(C_[i] + (size_t)Ie * 4 * L0.H, dK_ + (size_t)L0.I * 4 * L0.H, L0.P * 4 * L0.H, s);
        }
    }
    for (auto& R : ch)
      if (R.want_wgrads) {
        if (!batched || !rest) layer_wgrads_gemms(R, 0, T, false, s, !batched, !rest);
        if (!rest) layer_wgrads_colsums(R, T, s, scratch);
      }
  } else if (batch_wgrads(ch, T, side, false, &dK_done)) {
    // every layer's dWp + column sums as three launches beside the chip-filling dK GEMMs, which stay on s one after the other
    if (between) between(s);
    after_between();
    // the layers' kernel gradients [x | m]^T dZ: same shapes (batch_wgrads checked that) -> one stream-K launch over all of them
    std::vector<const LayerRun*> rs;
    for (auto& R : ch) if (R.want_wgrads) rs.push_back(&R);
    bool batched = false;
    if (rs.size() >= 2 && rs.size() <= (size_t)GEMM_MAXB) {
      const float *A_[GEMM_MAXB], *A2_[GEMM_MAXB], *B_[GEMM_MAXB]; float* C_[GEMM_MAXB];
      const LstmLayer& L0 = *rs[0]->L;
      const bool padI = L0.I % 4 != 0;                     // (x over its ld columns into dk_tmp, rows copied behind: see model.h dk_tmp)
      const int Ie = padI ? L0.ldI : L0.I;
      for (size_t i = 0; i < rs.size(); ++i) { A_[i] = rs[i]->in; A2_[i] = rs[i]->S->mst; B_[i] = rs[i]->S->gates; C_[i] = padI ? dk_tmp + i * dk_tmp_per : rs[i]->ps->Gd(rs[i]->L->tK); }
      batched = launch_gemm_batch((int)rs.size(), A_, L0.ldI, A2_, L0.ldP, Ie, B_, 4 * L0.H, C_, 4 * L0.H, Ie + L0.P, 4 * L0.H, T * rs[0]->N, false, s,
                                  (side && s == side) ? gemm_ws2 : gemm_ws, gemm_ws_floats);
      if (batched && padI)
        for (size_t i = 0; i < rs.size(); ++i) {
          float* dK_ = rs[i]->ps->Gd(rs[i]->L->tK);
          launch_copy_f(C_[i], dK_, L0.I * 4 * L0.H, s);
          launch_copy_f(C_[i] + (size_t)Ie * 4 * L0.H, dK_ + (size_t)L0.I * 4 * L0.H, L0.P * 4 * L0.H, s);
        }
    }
    if (!batched)
      for (auto& R : ch)
        if (R.want_wgrads) layer_wgrads_gemms(R, 0, T, false, s, true, false);
  } else {
    bool first = true;
    for (auto& R : ch) {
      if (!R.want_wgrads) continue;
      if (first) { first = false; continue; }                  // (the lowest layer with weight gradients stays on s)
      layer_wgrads(R, T, side);
    }
    if (between) between(s);
    after_between();
    for (auto& R : ch)
      if (R.want_wgrads) { layer_wgrads(R, T, s); break; }
  }
  ev = ev_pool[ev_next++ & 15];
  (void)hipEventRecord(ev, side);
  (void)hipStreamWaitEvent(s, ev, 0);                       // join: the optimizer needs every gradient
}

void Model::rnn_backward(std::vector<Chain>& chains, int T, hipStream_t s, const std::vector<int>* offsets,
                         const std::vector<FcStage>* fcs) {
  if (lazy_sw) { refresh_swizzles(RSRGAN_NET_G, s); refresh_swizzles(RSRGAN_NET_D, s); }
  {
    ZeroList zl{};
    for (auto& ch : chains)
      for (auto& R : ch) {
        if (zl.n + 2 > 32) { launch_zero_many(zl, s); zl.n = 0; }
        zl.p[zl.n] = R.S->dc + (size_t)R.row0 * R.L->H; zl.len[zl.n++] = (unsigned)((size_t)R.N * R.L->H);
        zl.p[zl.n] = R.S->dmst + (size_t)R.row0 * R.L->ldP; zl.len[zl.n++] = (unsigned)((size_t)R.N * R.L->ldP);
      }
    launch_zero_many(zl, s);
  }
  if (!wavefront()) {
    for (auto& ch : chains)
      for (int l = (int)ch.size() - 1; l >= 0; --l) {
        const LayerRun& R = ch[l];
        for (int t = T - 1; t >= 0; --t) {
          BwdAJobs aj{}; aj.n = 1; fill_bwd_a(aj.j[0], R, t);
          run_bwd_a(aj, bwd_a_blocks(R.L->H, R.N), kb16(R.L->ldP), s);
          BwdBJobs bj{}; bj.n = 1; fill_bwd_b(bj.j[0], R, t, false);
          if (bwd_b_splitk_ok(bj)) run_bwd_b_splitk(bj, s);
          else launch_bwd_b(bj, job_blocks(bj.j[0].nblk_c, R.N, kb16(4 * R.L->H) <= 64 ? 16 : 32), kb16(4 * R.L->H), s);
        }
        if (R.want_wgrads) layer_wgrads(R, T, s);
        if (R.din) {   // din (+)= dZ . K[0:I]^T, batched over time
          const int H4 = 4 * R.L->H;
          gemm(R.S->gates, H4, true, R.ps->W(R.L->tK), H4, true, R.din, R.L->ldI, T * R.N, R.L->I, H4, nullptr, 0, 0.f,
               R.din_accumulate, s);
        }
      }
    return;
  }
  int last = 0;
  for (size_t c = 0; c < chains.size(); ++c)
    last = std::max(last, (offsets ? (*offsets)[c] : 0) + (int)chains[c].size() - 1 + T - 1);
  if (fcs)
    for (auto& F : *fcs) last = std::max(last, F.offset + T - 1);
  // Weight-gradient GEMMs ride a side stream in time chunks: a chunk [tb, te) of a chain is final once
  // its bottom layer has passed time tb, i.e. after diagonal (T-1-tb) + off + Lc - 1.
  const int nchunk = (overlap() && T >= 16) ? 4 : 1;
  bool any_w = false;
  for (auto& ch : chains) for (auto& R : ch) any_w |= R.want_wgrads;
  const bool ovl = overlap() && any_w;
  std::vector<int> chunk_done(chains.size(), 0);            // chunks already handed to the side stream
  auto chunk_lo = [&](int c) { return T - ((c + 1) * T) / nchunk; };   // chunk c (c=0 = latest frames): [lo, hi)
  auto chunk_hi = [&](int c) { return T - (c * T) / nchunk; };
  for (int d = 0; d <= last; ++d) {
    BwdAJobs aj{}; BwdBJobs bj{};
   
    int ab = 0, bb = 0, ak = 0, bk = 0;
    auto flush_a = [&]() { if (aj.n) run_bwd_a(aj, ab, ak, s); aj.n = 0; ab = ak = 0; };
    auto flush_b = [&]() {
      if (bj.n) {
        if (bwd_b_splitk_ok(bj)) run_bwd_b_splitk(bj, s);
        else {
          launch_bwd_b(bj, bb, bk, s);                 // (small-K launches use 16-row tiles; the launcher places the jobs)
        }
      }
      bj.n = 0; bb = bk = 0;
    };
    for (size_t c = 0; c < chains.size(); ++c) {
      Chain& ch = chains[c];
      const int Lc = (int)ch.size(), off = offsets ? (*offsets)[c] : 0;
      for (int l = Lc - 1; l >= 0; --l) {
        const int t = T - 1 - (d - off - (Lc - 1 - l));
        if (t < 0 || t >= T) continue;
        if (aj.n == MAXJ) flush_a();
        BwdAJob& a = aj.j[aj.n++]; fill_bwd_a(a, ch[l], t); ab += bwd_a_blocks(ch[l].L->H, ch[l].N);
        ak = std::max(ak, kb16(ch[l].L->ldP));
      }
    }
    flush_a();
    for (size_t c = 0; c < chains.size(); ++c) {
      Chain& ch = chains[c];
      const int Lc = (int)ch.size(), off = offsets ? (*offsets)[c] : 0;
      for (int l = Lc - 1; l >= 0; --l) {
        const int t = T - 1 - (d - off - (Lc - 1 - l));
        if (t < 0 || t >= T) continue;
        const LayerRun& R = ch[l];
        if (bj.n == MAXJ) flush_b();
        BwdBJob& b = bj.j[bj.n++]; fill_bwd_b(b, R, t, R.din != nullptr); bb += job_blocks(b.nblk_c, R.N);
        bk = std::max(bk, kb16(4 * R.L->H));
      }
    }
    if (fcs)
      for (auto& F : *fcs) {
        const int t = T - 1 - (d - F.offset);
        if (t < 0 || t >= T) continue;
        if (bj.n == MAXJ) flush_b();
        BwdBJob& b = bj.j[bj.n++]; fill_fc_bwd(b, F, t); bb += job_blocks(b.nblk_c, F.N);
        bk = std::max(bk, kb16(F.ld_in));
      }
    flush_b();
    if (ovl) {
      for (size_t c = 0; c < chains.size(); ++c) {
        Chain& ch = chains[c];
        if (!ch[0].want_wgrads) continue;
        const int off = offsets ? (*offsets)[c] : 0, Lc = (int)ch.size();
        while (chunk_done[c] < nchunk && d >= (T - 1 - chunk_lo(chunk_done[c])) + off + Lc - 1) {
          const int cc = chunk_done[c]++;
          hipEvent_t ev = ev_pool[ev_next++ & 15];
          (void)hipEventRecord(ev, s);
          (void)hipStreamWaitEvent(side, ev, 0);
          for (auto& R : ch)
            if (R.want_wgrads) layer_wgrads_gemms(R, chunk_lo(cc), chunk_hi(cc), cc > 0, side);
        }
      }
    }
  }
  if (ovl) {
    for (auto& ch : chains)
      for (auto& R : ch)
        if (R.want_wgrads) layer_wgrads_colsums(R, T, side, scratch2);
    hipEvent_t ev = ev_pool[ev_next++ & 15];
    (void)hipEventRecord(ev, side);
    (void)hipStreamWaitEvent(s, ev, 0);                     // join: the optimizer needs every gradient
  } else if (!defer_wgrads) {
    for (auto& ch : chains)
      for (auto& R : ch)
        if (R.want_wgrads) layer_wgrads(R, T, s);
  }
}

// ------------------------------------------------------------------------------------------
int Model::prepare_batch(const float* x, const float* labels, const int32_t* lengths, int T, hipStream_t s, const float** nr,
                         const float** nf, hipStream_t early) {
  if (T <= 0 || T > Tmax) { set_error("T=%d outside (0, max_frames=%d]", T, Tmax); return RSRGAN_ERR_INVALID; }
  if (!x || (!lengths && !g_dnn())) { set_error("null input pointer"); return RSRGAN_ERR_INVALID; }
  // one launch: both packs, the lengths (twice: the stacked discriminator batch reads rows [B, 2B) as well) and the callers' noise
  // (a padded model, Bt < B: the caller's Bt rows land in rows [0, Bt) of every frame; rows [Bt, B) of the packs, their lengths and
  // their noise stay at the zeros of the allocation)
  StageJobs j{}; j.B = B; j.T = T; j.Bt = Bt;
  j.pack[0] = StagePack{x, x_tm, Din, ldDin};
  if (labels) j.pack[1] = StagePack{labels, lab_tm, Dout, ldDout};
  if (lengths) { j.copy[0] = StageCopy{lengths, len_dev, Bt}; j.copy[1] = StageCopy{lengths, len_dev + B, Bt}; }
  if (nr && *nr) { j.copy[2] = StageCopy{*nr, noise_r_buf, Bt * Dout}; *nr = noise_r_buf; }
  if (nf && *nf) { j.copy[3] = StageCopy{*nf, noise_f_buf, Bt * Dout}; *nf = noise_f_buf; }
  if (early) {
    // (RSRGAN_DPIPE) what D(real) needs -- the labels, the lengths, its noise -- goes ahead on the side stream, behind the last use of
    // the buffers it overwrites (ev_dfree); the input frames and D(G(x))'s noise stay in stream order
    StageJobs e{}; e.B = B; e.T = T; e.Bt = Bt;
    e.pack[1] = j.pack[1]; e.copy[0] = j.copy[0]; e.copy[1] = j.copy[1]; e.copy[2] = j.copy[2];
    j.pack[1] = StagePack{}; j.copy[0] = StageCopy{}; j.copy[1] = StageCopy{}; j.copy[2] = StageCopy{};
    (void)hipStreamWaitEvent(early, ev_dfree, 0);
    launch_stage_inputs(e, early);
  }
  launch_stage_inputs(j, s);
  cur_T = T;
  g_fwd_valid = false;
  return RSRGAN_OK;
}

Chain Model::g_chain(int T) {
  (void)T;
  Chain ch;
  const bool res = cfg.g_type == RSRGAN_G_RES_LSTM_L;
  for (size_t l = 0; l < gl.size(); ++l) {
    LayerRun R;
    R.ps = &G; R.L = &gl[l]; R.S = &g_st[l]; R.in = g_ins[l];
    R.N = B; R.Ns = B; R.row0 = 0; R.len = len_dev;
    R.zx_batched = (l == 0);                 // layer 0's input exists for all t before the wave starts
    if (res) { R.res_in = g_ins[l]; R.res_out = g_res[l]; }   // inputs_{l+1} = outputs_l + inputs_l (res_lstm_l.py:111,121,131,190)
    if (seq_drop_on()) R.drop = DropSpec{drop_ctr, drop_seed, (1ull << 40) | ((unsigned long long)l << 20), drop_thr(), keep_prob};
    ch.push_back(R);
  }
  return ch;
}

Chain Model::d_chain(int N, int Ns, int row0) {
  Chain ch;
  for (size_t l = 0; l < dl.size(); ++l) {
    LayerRun R;
    R.ps = &D; R.L = &dl[l]; R.S = &d_st[l]; R.in = l == 0 ? xd : d_st[l - 1].out;
    R.N = N; R.Ns = Ns; R.row0 = row0; R.len = len_dev + row0;
    R.zx_batched = false;
    ch.push_back(R);
  }
  return ch;
}

void Model::g_forward_head(int T, hipStream_t s) {
  if (cfg.g_type == RSRGAN_G_LSTM) {   // h = leakyrelu(x.W + b)  (models/lstm.py:82-87)
    const int P = gR, ldP = pad4(P);
    gemm(x_tm, ldDin, true, G.W(g_fc_in_w), ldP, false, g_h0, ldP, T * B, P, Din, G.W(g_fc_in_b), 1, cfg.lrelu_alpha, false, s);
  }
}
void Model::g_forward_tail(int T, hipStream_t s) {   // y = outputs.W + b (models/lstm.py:121-124)
  const int P = gR, ldP = pad4(P);
  gemm(g_ins[gl.size()], ldP, true, G.W(g_fc_out_w), ldDout, false, y_tm, ldDout, T * B, Dout, P, G.W(g_fc_out_b), 0, 0.f, false, s);
  g_fwd_valid = true;
}
void Model::g_forward(int T, hipStream_t s, Chain* extra) {
  if (g_dnn()) { bn_eval_call = false; g_frame_forward(T * B, s); g_fwd_valid = true; return; }
  g_forward_head(T, s);
  if (!extra && persist_forward_g(T, s)) { g_forward_tail(T, s); return; }
  std::vector<Chain> chains;
  chains.push_back(g_chain(T));
  if (extra) chains.push_back(*extra);
  rnn_forward(chains, T, s);
  g_forward_tail(T, s);
}

void Model::d_logits(int N, int T, hipStream_t s) {
  const int ldPd = pad4(dR);
  gemm(d_st[dl.size() - 1].out, ldPd, true, D.W(d_fc_w), 4, false, logits, 4, T * N, 1, dR, D.W(d_fc_b), 0, 0.f, false, s);
}

// discriminator_lstm's head in one pass (kernels.hip k_dhead1 / 2): logits and the LSGAN terms of N rows per frame (the first n_real
// of them against *t_real), and with want_grads dlogits and d(outputs) in d_dB, with want_wgrads the output FC's gradients too.
// False: not applicable (the caller runs d_logits + launch_lsgan + the GEMMs of d_backward_pass).
bool Model::d_head(int N, int T, int n_real, const float* t_real, const float* t_fake, float* loss3, bool want_grads, bool want_wgrads,
                   hipStream_t s) {
  static const bool on = [] { const char* e = getenv("RSRGAN_DHEAD"); return !e || atoi(e) != 0; }();
  const int ldPd = pad4(dR), nb = (T * N + 63) / 64;
  if (!on || d_dnn() || dl.empty() || dR % 4 != 0 || dR + 3 > 64 ||       // (k_dhead2 sums 64 quantities: 3 + dR)
      (size_t)nb * (DH_MAXR + 3) > scratch_floats) return false;
  DHeadArgs a{};
  a.top = d_st[dl.size() - 1].out; a.ldt = ldPd; a.w = D.W(d_fc_w); a.ldw = 4; a.b = D.W(d_fc_b);
  a.logits = logits; a.ldl = 4; a.dlogits = dlogits; a.dout = d_dB; a.ldo = ldPd; a.gw = D.Gd(d_fc_w); a.gb = D.Gd(d_fc_b);
  a.T = T; a.Nd = N; a.n_real = n_real; a.dR = dR; a.t_real = t_real; a.t_fake = t_fake; a.loss3 = loss3; a.part = scratch;
  a.want_grads = want_grads ? 1 : 0; a.want_wgrads = want_wgrads ? 1 : 0;
  a.Bp = pad_Bp(); a.Bt = Bt;
  launch_dhead(a, s);
  return true;
}

// leaves in last_dx0 (d_dA or d_dB) the gradient w.r.t. the discriminator input when need_dx0
void Model::d_backward_pass(int N, int T, bool want_wgrads, bool need_dx0, const float* dlog, hipStream_t s, bool head_done) {
  const int R = T * N;
  const int ldPd = pad4(dR);
  const size_t Ld = dl.size();
  const float* top = d_st[Ld - 1].out;
  if (want_wgrads && !head_done) {
    gemm(top, ldPd, false, dlog, 4, false, D.Gd(d_fc_w), 4, dR, 1, R, nullptr, 0, 0.f, false, s);
    launch_colsum(dlog, 4, nullptr, 0, D.Gd(d_fc_b), R, 1, scratch, s);
  }
  float* cur = d_dB;
  float* other = d_dA;
  // d(outputs) = dlogits . W^T  (d_head has left it in d_dB)
  if (!head_done) gemm(dlog, 4, true, D.W(d_fc_w), 4, true, cur, ldPd, R, dR, 1, nullptr, 0, 0.f, false, s);
  std::vector<Chain> chains(1, d_chain(N, N, 0));
  Chain& ch = chains[0];
  for (int l = (int)Ld - 1; l >= 0; --l) {
    ch[l].dout = cur;
    ch[l].din = (l > 0 || need_dx0) ? other : nullptr;
    ch[l].din_accumulate = false;
    ch[l].want_wgrads = want_wgrads;
    std::swap(cur, other);
  }
  if (!persist_backward(ch, T, s)) rnn_backward(chains, T, s);
  last_dx0 = cur;
}

void Model::g_backward_pass(int T, float* dy, hipStream_t s) {
  const int R = T * B;
  const int P = gR, ldP = pad4(P);
  const size_t Lg = gl.size();
  // output FC: dW = in^T . dy ; db = colsum(dy) ; d(in) = dy . W^T
  gemm(g_ins[Lg], ldP, false, dy, ldDout, false, G.Gd(g_fc_out_w), ldDout, P, Dout, R, nullptr, 0, 0.f, false, s);
  launch_colsum(dy, ldDout, nullptr, 0, G.Gd(g_fc_out_b), R, Dout, scratch, s);
  float* cur = g_dA;
  float* other = g_dB;
  gemm(dy, ldDout, true, G.W(g_fc_out_w), ldDout, true, cur, ldP, R, P, Dout, nullptr, 0, 0.f, false, s);
  std::vector<Chain> chains(1, g_chain(T));
  Chain& ch = chains[0];
  for (auto& r : ch) r.want_wgrads = true;
  if (cfg.g_type == RSRGAN_G_LSTM) {
    for (int l = (int)Lg - 1; l >= 0; --l) {
      ch[l].dout = cur; ch[l].din = other; ch[l].din_accumulate = false;
      std::swap(cur, other);
    }
    if (!persist_backward_g(ch, T, s)) rnn_backward(chains, T, s);
    // through leakyrelu and the input FC (models/lstm.py:82-87)
    launch_lrelu_bwd(g_h0, cur, (size_t)R, P, ldP, cfg.lrelu_alpha, s);
    gemm(x_tm, ldDin, false, cur, ldP, false, G.Gd(g_fc_in_w), ldP, Din, P, R, nullptr, 0, 0.f, false, s);
    launch_colsum(cur, ldP, nullptr, 0, G.Gd(g_fc_in_b), R, P, scratch, s);
  } else if (cfg.g_type == RSRGAN_G_RES_LSTM_L) {
    // d(inputs_l) = dx_l + d(inputs_{l+1}): accumulate in place in one buffer
    for (int l = (int)Lg - 1; l >= 0; --l) {
      ch[l].dout = cur; ch[l].din = l > 0 ? cur : nullptr; ch[l].din_accumulate = true;
    }
    if (!persist_backward_g(ch, T, s)) rnn_backward(chains, T, s);
  } else {
    for (int l = (int)Lg - 1; l >= 0; --l) {
      ch[l].dout = cur; ch[l].din = l > 0 ? other : nullptr; ch[l].din_accumulate = false;
      std::swap(cur, other);
    }
    if (!persist_backward_g(ch, T, s)) rnn_backward(chains, T, s);
  }
}

int Model::d_backward(const float* x, const float* labels, const int32_t* lengths, int T, const float* nr, const float* nf,
                      float* out_losses, bool want_grads, hipStream_t s) {
  if (!labels) { set_error("labels required"); return RSRGAN_ERR_INVALID; }
  if (supervised()) { set_error("RSRGAN_FLAG_SUPERVISED: the trainer graph has no discriminator step"); return RSRGAN_ERR_STATE; }
  if (g_dnn()) return dnn_d_backward(x, labels, T, out_losses, want_grads, s);
  if (d_dnn()) { nr = nullptr; nf = nullptr; }    // discriminator_dnn.py:58: the noise layer is commented out
  bn_eval_call = !want_grads;                      // (is_training of this fetch: the DropoutWrapper masks; read by seq_drop_on() below)
  dfree_current = false;                           // this run writes the discriminator's stash and input rows
  // RSRGAN_DPIPE: D(real) on the side stream, ahead of this call's place in the stream; the run itself is k_glstm_fwd_dt (D(G(x)) trailing)
  bool dsplit = false;
  // (RSRGAN_DPIPE=1 covers labels and lengths; a noise_real tensor drawn on the caller's stream right before the call is covered by =2 only)
  static const int dpipe_level = [] { const char* e = getenv("RSRGAN_DPIPE"); return e ? atoi(e) : 0; }();
  if (dpipe && (!nr || dpipe_level >= 2) && wavefront() && gp_fwd_on() && !d_dnn() && !seq_drop_on() && T > 0 && T <= Tmax) {
    Chain dchk = d_chain(B, 2 * B, B);
    dsplit = persist_forward_real(T, side, true) && persist_forward_g_trail(dchk, T, s, nf, true);
  }
  int rc = prepare_batch(x, labels, lengths, T, s, &nr, &nf, dsplit ? side : nullptr);      // (stages the noise too: caller pointers never enter a graph)
  if (rc) return rc;
  if (dsplit) {
    // discriminator input rows [0,B) = labels + noise_real, D(real) over them; this stream waits for it in front of its own launches
    // (the fused forward launch and the D(real) launch do not fit the device together)
    launch_add_noise_rows(lab_tm, nr, xd, B, T, Dout, ldDout, 2 * B, 0, side);
    (void)persist_forward_real(T, side);
    (void)hipEventRecord(ev_real, side);
    (void)hipStreamWaitEvent(s, ev_real, 0);
  }
  if (seq_drop_on()) launch_drop_tick(drop_ctr, s);     // a new training run: new masks (read from device memory: graph-safe)
  // rsrgan_d_step: the update follows in the same call -- its launches close this segment (one graph: no launch boundary in front
  // of the clip / SGD / weight-copy kernels); RSRGAN_FUSED_SEG=0 keeps them in a segment of their own
  static const bool fused_seg_d = [] { const char* e = getenv("RSRGAN_FUSED_SEG"); return !e || atoi(e) != 0; }();
  const bool inl = fused_apply && fused_seg_d && want_grads && !d_dnn() && graphs_on();
  const unsigned kbits = (want_grads ? 1u : 0u) | (nr ? 2u : 0u) | (nf ? 4u : 0u) | (inl ? 8u : 0u) | (dsplit ? 16u : 0u);
  run_seg(seg_key(SEG_D, T, kbits), s, [&]() {
  // discriminator input rows [0,B) = labels + noise_real (gan_rnn_placeholder.py:207,212; utils/ops.py:19-30)
  if (!dsplit) launch_add_noise_rows(lab_tm, nr, xd, B, T, Dout, ldDout, 2 * B, 0, s);
  bool g_done = false;
  if (dsplit) {
    g_forward_head(T, s);
    Chain dch = d_chain(B, 2 * B, B);
    g_done = persist_forward_g_trail(dch, T, s, nf);
  }
  if (!g_done && wavefront() && gp_fwd_on()) {
    // the generator as ONE persistent launch, then both discriminator calls stacked (N = 2B) as another
    g_forward_head(T, s);
    g_done = persist_forward_g(T, s);
    if (g_done) {
      g_forward_tail(T, s);
      launch_add_noise_rows(y_tm, nf, xd, B, T, Dout, ldDout, 2 * B, B, s);
      if (!d_dnn()) {
        std::vector<Chain> chains(1, d_chain(2 * B, 2 * B, 0));
        if (!persist_forward(chains[0], T, s)) rnn_forward(chains, T, s);
      }
    }
  }
  if (g_done) {
  } else if (wavefront()) {
    // ONE wave: G's layers | D(real) (independent of G) | per-step output FC -> y_t, xd fake rows |
    // D(fake) two diagonals behind G's top layer
    if (!gp_fwd_on()) g_forward_head(T, s);
    const int Lg = (int)gl.size(), ldP = pad4(gR);
    std::vector<Chain> chains{g_chain(T)};
    std::vector<int> offs{0};
    if (!d_dnn()) {
      chains.push_back(d_chain(B, 2 * B, 0)); offs.push_back(0);
      chains.push_back(d_chain(B, 2 * B, B)); offs.push_back(Lg + 1);
    }
    FcStage F;
    F.offset = Lg; F.N = B; F.K = gR; F.D = Dout;
    F.in = g_ins[Lg]; F.ld_in = ldP; F.WT = g_fc_out_wT; F.bias = G.W(g_fc_out_b); F.noise = nf;
    F.y = y_tm; F.ldy = ldDout; F.out2 = xd; F.ld2 = ldDout; F.Ns2 = 2 * B; F.row02 = B;
    std::vector<FcStage> fcs{F};
    rnn_forward(chains, T, s, &offs, &fcs);
  } else {
    g_forward(T, s);
    launch_add_noise_rows(y_tm, nf, xd, B, T, Dout, ldDout, 2 * B, B, s);
    if (!d_dnn()) {
      std::vector<Chain> chains(1, d_chain(2 * B, 2 * B, 0));
      rnn_forward(chains, T, s);
    }
  }
  if (d_dnn()) {
    d_dnn_forward_loss(T, 2 * B, B, want_grads, losses, s);
    if (want_grads) fc_backward(D, dfc, d_act, T * 2 * B, dlogits, true, false, s);
  } else {
    const bool head = d_head(2 * B, T, B, dyn + DYN_D_REAL, dyn + DYN_D_FAKE, losses, want_grads, want_grads, s);
    if (!head) {
      d_logits(2 * B, T, s);
      launch_lsgan(logits, 4, want_grads ? dlogits : nullptr, T, 2 * B, B, dyn + DYN_D_REAL, dyn + DYN_D_FAKE, losses, s, false, 0.f, 0.f, pad_Bp(), Bt);
    }
    if (want_grads) d_backward_pass(2 * B, T, true, false, dlogits, s, head);
  }
  if (inl) apply_body(RSRGAN_NET_D, s);
  });
  d_partial_fresh = false;        // (a later rsrgan_apply -- after the caller's all-reduce -- squares the gradients it finds)
  g_fwd_valid = true;
  if (inl) apply_inlined |= 2;
  if (want_grads) { finish_buckets(RSRGAN_NET_D, s); d_grads_ready = true; }
  if (out_losses) launch_copy_f(losses, out_losses, 3, s);
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

int Model::g_backward(const float* x, const float* labels, const int32_t* lengths, int T, const float* nf,
                      float* out_losses, bool want_grads, bool reuse, hipStream_t s) {
  if (!labels) { set_error("labels required"); return RSRGAN_ERR_INVALID; }
  if (g_dnn()) return dnn_g_backward(x, labels, T, out_losses, want_grads, reuse, s);
  bn_eval_call = !want_grads;                      // (is_training of this fetch)
  dfree_current = false;                           // this run reads and writes the discriminator's stash
  if (seq_drop_on()) { reuse = false; launch_drop_tick(drop_ctr, s); }     // a new sess.run: new DropoutWrapper masks, a new forward
  if (supervised()) {
    // RNNTrainer (models/rnn_trainer.py:131-156): g_loss = 0.5*Dout*mse(G(x), labels) + l2; no discriminator in the graph
    int rc = prepare_batch(x, labels, lengths, T, s);
    if (rc) return rc;
    g_forward(T, s);
    const bool l2s = !cfg.cross_validation && scal[RSRGAN_L2_SCALE] > 0.0;
    HIPC(hipMemsetAsync(losses + 3, 0, sizeof(float), s));
    if (want_grads) {
      float* dy = g_dC;                            // g_backward_pass ping-pongs g_dA / g_dB
      launch_mse(y_tm, lab_tm, ldDout, dy, T * B, Dout, dyn + DYN_LAMBDA, false, losses + 4, scratch, s, pad_Bp(), Bt);
      g_backward_pass(T, dy, s);
      if (l2s) {
        launch_l2(G.w, G.g, G.ct, dyn + DYN_L2, G.partial, s);
        launch_l2_total(G.partial, G.ct.n_chunks, dyn + DYN_L2, losses + 5, s);
      }
      { finish_buckets(RSRGAN_NET_G, s); g_grads_ready = true; }
    } else {
      launch_mse(y_tm, lab_tm, ldDout, nullptr, T * B, Dout, dyn + DYN_LAMBDA, false, losses + 4, scratch, s, pad_Bp(), Bt);
    }
    if (!(want_grads && l2s)) HIPC(hipMemsetAsync(losses + 5, 0, sizeof(float), s));
    launch_g_total(losses + 3, dyn + DYN_LAMBDA, s);
    if (out_losses) launch_copy_f(losses + 3, out_losses, 4, s);
    HIPC(hipGetLastError());
    return RSRGAN_OK;
  }
  if (d_dnn()) nf = nullptr;
  if (reuse) {
    if (!g_fwd_valid || T != cur_T) { set_error("reuse_g_forward without a valid generator forward"); return RSRGAN_ERR_STATE; }
  } else {
    int rc = prepare_batch(x, labels, lengths, T, s);
    if (rc) return rc;
  }
  nf = stage_noise(nf, noise_f_buf, s);
  const bool l2_on = !cfg.cross_validation && scal[RSRGAN_L2_SCALE] > 0.0;
  const bool wave_bwd = want_grads && !d_dnn() && wavefront();
  // (per-layer weight-gradient segments exist for the caller's bucketed all-reduce; rsrgan_g_step applies the update itself, so
  // nothing can run between the backward pass and the optimizer: one segment, no graph boundaries between the layers)
  static const bool fused_seg = [] { const char* e = getenv("RSRGAN_FUSED_SEG"); return !e || atoi(e) != 0; }();
  const bool bucketed = wave_bwd && gbk[RSRGAN_NET_G].size() > 1 && !overlap() && !(fused_apply && fused_seg);
  const int R = T * B, Ld = (int)dl.size(), Lg = (int)gl.size();
  const int ldPd = pad4(dR), P = gR, ldP = pad4(P);
  float* dy = g_dB;                               // [T*B][ldDout] (wavefront backward)
  float* bufA = g_dA; float* bufB = g_dC;         // G-side gradient ping-pong of the wavefront backward (dy itself lives in g_dB)
  std::vector<Chain> bw_chains;                   // filled by the main segment's body; the bucketed weight-gradient segments read
                                                  // only pointers that are the same on every call, so they rebuild it themselves
  auto build_bw_chains = [&]() {
    Chain dch = d_chain(B, B, 0);
    float* cur = d_dB; float* other = d_dA;
    for (int l = Ld - 1; l >= 0; --l) {
      dch[l].dout = cur; dch[l].want_wgrads = false;
      if (l == 0) { dch[l].din = dy; dch[l].din_accumulate = true; }
      else { dch[l].din = other; dch[l].din_accumulate = false; std::swap(cur, other); }
    }
    Chain gch = g_chain(T);
    bufA = g_dA; bufB = g_dC;
    for (int l = Lg - 1; l >= 0; --l) {
      gch[l].want_wgrads = true;
      if (cfg.g_type == RSRGAN_G_RES_LSTM_L) {     // d(inputs_l) = dx_l + d(inputs_{l+1}): one buffer, accumulated in place
        gch[l].dout = g_dA; gch[l].din = l > 0 ? g_dA : nullptr; gch[l].din_accumulate = true;
      } else {
        const bool need = cfg.g_type == RSRGAN_G_LSTM || l > 0;      // lstm: d(h0) feeds the input FC
        gch[l].dout = bufA; gch[l].din = need ? bufB : nullptr; gch[l].din_accumulate = false;
        std::swap(bufA, bufB);
      }
    }
    bw_chains = std::vector<Chain>{dch, gch};
  };
  if (wave_bwd) build_bw_chains();                // host-only bookkeeping (bufA after the loop = d(h0))
  // Without gradient buckets (rsrgan_g_step) and with the generator's BPTT as a persistent launch, the two FCs' parameter gradients --
  // eight short launches that used to FOLLOW the dK GEMMs -- ride the side stream beside them.  Decided here, on the host: a
  // replayed segment does not run its body.
  static const bool fc_side_env = [] {
    const char* e = getenv("RSRGAN_WGRAD_STREAMS"); const char* f = getenv("RSRGAN_FC_SIDE");
    return (!e || atoi(e) >= 2) && (!f || atoi(f) != 0);
  }();
  const bool fcs_inside = wave_bwd && !bucketed && side && fc_side_env && cfg.g_type == RSRGAN_G_LSTM && dl[0].ldI == ldDout &&
                          persist_backward_g(bw_chains[1], T, s, true);
  // rsrgan_g_step with nothing left between the backward segment and the update (no input-FC segment, no buckets): the loss
  // tail and the update's launches close the main segment -- ONE graph per G-run instead of three
  const bool inl = fused_apply && fused_seg && wave_bwd && !bucketed && graphs_on() && !(cfg.g_type == RSRGAN_G_LSTM && !fcs_inside);
  const unsigned kbits = (want_grads ? 1u : 0u) | (nf ? 2u : 0u) | (reuse ? 4u : 0u) | (l2_on ? 8u : 0u) | (bucketed ? 16u : 0u) | (inl ? 32u : 0u);
  auto tail_body = [&]() {
    if (want_grads && l2_on) {
      launch_l2(G.w, G.g, G.ct, dyn + DYN_L2, G.partial, s);
      launch_l2_total(G.partial, G.ct.n_chunks, dyn + DYN_L2, losses + 5, s);
    } else {
      (void)hipMemsetAsync(losses + 5, 0, sizeof(float), s);
    }
    launch_g_total(losses + 3, dyn + DYN_LAMBDA, s);
  };

  // the discriminator's BPTT in its trailing form beside the generator's (persist_backward_trail): decided here, on the host
  bool trail_plan = false;
  if (wave_bwd && want_grads && !d_dnn() && dl[0].ldI == ldDout && persist_backward_g(bw_chains[1], T, s, true)) {
    Chain dc0 = bw_chains[0];
    dc0[0].din = nullptr;
    trail_plan = persist_backward_trail(dc0, T, s, dy, ldDout, g_dA, ldP, true);
  }
  // RSRGAN_DPIPE: the next D-run's D(real) may start as soon as the fused backward launch has finished -- the event is recorded between
  // two graph segments: the launch closes the first, the weight gradients (and the inlined update) are the second
  const bool gsplit = dpipe && trail_plan;
  auto bwd_rest = [&]() {
    const int ldP_ = pad4(gR);
    StreamFn pre, post;
    if (fcs_inside) {
      pre = [&](hipStream_t q) {
        gemm(g_ins[Lg], ldP_, false, dy, ldDout, false, G.Gd(g_fc_out_w), ldDout, gR, Dout, R, nullptr, 0, 0.f, false, q);
        launch_colsum(dy, ldDout, nullptr, 0, G.Gd(g_fc_out_b), R, Dout, (side && q == side) ? scratch2 : scratch, q);
      };
      post = [&](hipStream_t q) {
        launch_lrelu_bwd(g_h0, bufA, (size_t)R, gR, ldP_, cfg.lrelu_alpha, q);
        gemm(x_tm, ldDin, false, bufA, ldP_, false, G.Gd(g_fc_in_w), ldP_, Din, gR, R, nullptr, 0, 0.f, false, q);
        launch_colsum(bufA, ldP_, nullptr, 0, G.Gd(g_fc_in_b), R, gR, (side && q == side) ? scratch2 : scratch, q);
      };
    }
    // the next D-run's D(real) (32 workgroups that own their CUs) runs beside these launches when the host is ahead: the chip-filling
    // GEMMs leave it room (a persistent stream-K launch on all 256 CUs would wait for it with 32 of its workers, and it for them)
    static const int pipe_w = [] { const char* e = getenv("RSRGAN_DPIPE_W"); const int v = e ? atoi(e) : 224; return v >= 64 && v <= 256 ? v & ~7 : 224; }();
    const int saved_w = g_gemm_workers;
    g_gemm_workers = std::min(saved_w, pipe_w);
    defer_wgrads = bucketed;
    gp_phase = 2; persist_backward_g(bw_chains[1], T, s, false, pre, post); gp_phase = 0;
    defer_wgrads = false;
    if (!fcs_inside) {
      gemm(g_ins[Lg], ldP_, false, dy, ldDout, false, G.Gd(g_fc_out_w), ldDout, gR, Dout, R, nullptr, 0, 0.f, false, s);
      launch_colsum(dy, ldDout, nullptr, 0, G.Gd(g_fc_out_b), R, Dout, scratch, s);
    }
    g_gemm_workers = saved_w;
    if (inl) { tail_body(); apply_body(RSRGAN_NET_G, s); }
  };
  run_seg(seg_key(SEG_G_MAIN, T, kbits | (trail_plan ? 64u : 0u) | (gsplit ? 128u : 0u)), s, [&]() {
  // (trailing form) the top layer's gradient buffer armed with the all-ones pattern HERE, at the head of the run: k_glstm_bwd polls it,
  // and a fill in front of the fork would be the node both launches depend on (a fill node as the fork point serialized them)
  if (trail_plan) gpersist_arm_bytes(g_dA, (size_t)T * B * ldP * sizeof(float), s);
  bool g_done = false;
  if (!reuse && !wavefront()) g_forward(T, s);
  bool d_trailed = false;      // D(G(x)) ran inside the generator's forward launch (k_glstm_fwd_dt): y, the discriminator's input rows and its stash are there
  if (!reuse && wavefront() && gp_fwd_on()) {
    g_forward_head(T, s);
    if (!d_dnn()) {
      Chain dch = d_chain(B, B, 0);
      g_done = d_trailed = persist_forward_g_trail(dch, T, s, nf);
    }
    if (!g_done) {
      g_done = persist_forward_g(T, s);
      if (g_done) g_forward_tail(T, s);
    }
  }
  if (d_trailed) {
  } else if (!reuse && wavefront() && !g_done) {
    // ONE forward wave: G's layers | per-step output FC (-> y_t and D's input rows, noise added) | D's layers
    if (!gp_fwd_on()) g_forward_head(T, s);
    std::vector<Chain> chains{g_chain(T)};
    std::vector<int> offs{0};
    if (!d_dnn()) { chains.push_back(d_chain(B, B, 0)); offs.push_back(Lg + 1); }
    FcStage F;
    F.offset = Lg; F.N = B; F.K = gR; F.D = Dout;
    F.in = g_ins[Lg]; F.ld_in = pad4(gR); F.WT = g_fc_out_wT; F.bias = G.W(g_fc_out_b); F.noise = nf;
    F.y = y_tm; F.ldy = ldDout; F.out2 = xd; F.ld2 = ldDout; F.Ns2 = B; F.row02 = 0;
    std::vector<FcStage> fcs{F};
    rnn_forward(chains, T, s, &offs, &fcs);
  } else {
    // D(G(x)) on y + noise (utils/ops.py:19-30).  Without a noise draw (init_disc_noise_std = 0, the shipped recipe) the discriminator
    // reads y where it lies -- the same [T][B][ld] layout as its input rows --: one launch less in front of the recurrence
    const bool direct = !nf && !d_dnn() && dl[0].ldI == ldDout;
    if (!direct) launch_add_noise_rows(y_tm, nf, xd, B, T, Dout, ldDout, B, 0, s);
    if (!d_dnn()) {
      std::vector<Chain> chains(1, d_chain(B, B, 0));
      if (direct) chains[0][0].in = y_tm;
      if (!persist_forward(chains[0], T, s) && !fold_forward(chains[0], T, s)) rnn_forward(chains, T, s);
    }
  }
  // g_adv = mean((D(G(x)) - d_real)^2)  (gan_rnn_placeholder.py:246): all rows "fake", target d_real
  bool g_head = false;
  if (d_dnn()) {
    d_dnn_forward_loss(T, B, 0, want_grads, tmp3, s);
  } else {
    g_head = d_head(B, T, 0, dyn + DYN_D_REAL, dyn + DYN_D_REAL, tmp3, want_grads, false, s);
    if (!g_head) {
      d_logits(B, T, s);
      launch_lsgan(logits, 4, want_grads ? dlogits : nullptr, T, B, 0, dyn + DYN_D_REAL, dyn + DYN_D_REAL, tmp3, s, false, 0.f, 0.f, pad_Bp(), Bt);
    }
  }
  launch_copy_f(tmp3 + 1, losses + 3, 1, s);
  if (want_grads && d_dnn()) {
    // discriminator_dnn: data gradient through the FC stack (time-batched GEMMs), then the generator's BPTT wave
    float* dyd = fc_backward(D, dfc, d_act, T * B, dlogits, false, true, s);      // [T*B][ldDout] (d_joint_dim == 0)
    launch_mse(y_tm, lab_tm, ldDout, dyd, T * B, Dout, dyn + DYN_LAMBDA, true, losses + 4, scratch, s, pad_Bp(), Bt);
    g_backward_pass(T, dyd, s);
  } else if (wave_bwd) {
    // ONE backward wave: D's layers (data gradient only) | per-step output-FC backward | G's layers.
    // dy[t] = lambda*(y-lab)/(B*T) (written first) + d g_adv/d y[t] (accumulated by D layer 0's phase B)
    if (!g_head) gemm(dlogits, 4, true, D.W(d_fc_w), 4, true, d_dB, ldPd, R, dR, 1, nullptr, 0, 0.f, false, s);      // d(D outputs) = dlogits . W^T
    launch_mse(y_tm, lab_tm, ldDout, dy, R, Dout, dyn + DYN_LAMBDA, false, losses + 4, scratch, s, pad_Bp(), Bt);
    FcStage F;                                     // d(ins[L])[t] = dy[t] . W_out^T
    F.offset = Ld; F.N = B; F.K = Dout; F.D = P;
    F.in = dy; F.ld_in = ldDout; F.WT = G.W(g_fc_out_w); F.y = g_dA; F.ldy = ldP; F.accumulate = false;
    std::vector<int> offs{0, Ld + 1};
    std::vector<FcStage> fcs{F};
    defer_wgrads = bucketed;
    if (dl[0].ldI == ldDout && persist_backward_g(bw_chains[1], T, s, true)) {
      // the generator's BPTT as ONE persistent launch: the discriminator's data gradient first and alone (its own persistent launch,
      // layer 0's input gradient as a GEMM on top of the mse term in dy), then the output FC's data gradient as one GEMM
      std::vector<Chain> dc1(1, bw_chains[0]);
      dc1[0][0].din = nullptr;
      // round 5: the discriminator's BPTT in its trailing form INSIDE the generator's launch (k_glstm_bwd_dt; the generator's top layer
      // polls its gradient step by step): dy and g_dA are completed there
      const bool trailed = trail_plan && persist_backward_trail(dc1[0], T, s, dy, ldDout, g_dA, ldP);
      if (!trailed) {
        if (!persist_backward(dc1[0], T, s)) rnn_backward(dc1, T, s);
        const int H4d = 4 * dl[0].H;
        gemm(d_st[0].gates, H4d, true, D.W(dl[0].tK), H4d, true, dy, ldDout, R, dl[0].I, H4d, nullptr, 0, 0.f, true, s);
        gemm(dy, ldDout, true, G.W(g_fc_out_w), ldDout, true, g_dA, ldP, R, P, Dout, nullptr, 0, 0.f, false, s);
      }
      gp_trail_next = trailed;
      // (fcs_inside: the output FC's parameter gradients need dy, complete here; the input FC's the d(h0) GEMM)
      StreamFn pre, post;
      if (fcs_inside) {
        pre = [&](hipStream_t q) {
          gemm(g_ins[Lg], ldP, false, dy, ldDout, false, G.Gd(g_fc_out_w), ldDout, P, Dout, R, nullptr, 0, 0.f, false, q);
          launch_colsum(dy, ldDout, nullptr, 0, G.Gd(g_fc_out_b), R, Dout, (side && q == side) ? scratch2 : scratch, q);
        };
        post = [&](hipStream_t q) {        // through leakyrelu and the input FC (models/lstm.py:82-87); bufA holds d(h0)
          launch_lrelu_bwd(g_h0, bufA, (size_t)R, P, ldP, cfg.lrelu_alpha, q);
          gemm(x_tm, ldDin, false, bufA, ldP, false, G.Gd(g_fc_in_w), ldP, Din, P, R, nullptr, 0, 0.f, false, q);
          launch_colsum(bufA, ldP, nullptr, 0, G.Gd(g_fc_in_b), R, P, (side && q == side) ? scratch2 : scratch, q);
        };
      }
      if (gsplit) {      // (RSRGAN_DPIPE) the launch only: ev_dfree is recorded behind it, what follows is a segment of its own (bwd_rest)
        gp_phase = 1; persist_backward_g(bw_chains[1], T, s, false, pre, post); gp_phase = 0;
        gp_trail_next = false;
        return;
      }
      persist_backward_g(bw_chains[1], T, s, false, pre, post);
      gp_trail_next = false;
    } else {
      rnn_backward(bw_chains, T, s, &offs, &fcs);
    }
    defer_wgrads = false;
    // output FC parameter gradients (batched over time, dy is complete now)
    if (!fcs_inside) {
      gemm(g_ins[Lg], ldP, false, dy, ldDout, false, G.Gd(g_fc_out_w), ldDout, P, Dout, R, nullptr, 0, 0.f, false, s);
      launch_colsum(dy, ldDout, nullptr, 0, G.Gd(g_fc_out_b), R, Dout, scratch, s);
    }
  } else if (want_grads) {
    d_backward_pass(B, T, false, true, dlogits, s, g_head);
    float* dyd = last_dx0;                        // d g_adv / d y
    launch_mse(y_tm, lab_tm, ldDout, dyd, T * B, Dout, dyn + DYN_LAMBDA, true, losses + 4, scratch, s, pad_Bp(), Bt);
    g_backward_pass(T, dyd, s);
  } else {
    launch_mse(y_tm, lab_tm, ldDout, nullptr, T * B, Dout, dyn + DYN_LAMBDA, false, losses + 4, scratch, s, pad_Bp(), Bt);
  }
  if (inl) { tail_body(); apply_body(RSRGAN_NET_G, s); }
  });
  if (gsplit) {
    defer_wgrads = false;
    (void)hipEventRecord(ev_dfree, s);
    dfree_inside = true; dfree_current = true;
    run_seg(seg_key(SEG_G_MAIN, T, kbits | 64u | 256u), s, bwd_rest);
  }
  if (inl) apply_inlined |= 1;
  if (!reuse) g_fwd_valid = true;
  if (wave_bwd) {
    int bi = 0;
    if (bucketed) mark_bucket(RSRGAN_NET_G, bi++, s);
    if (cfg.g_type == RSRGAN_G_LSTM && !fcs_inside) {
      run_seg(seg_key(SEG_G_FCIN, T, kbits), s, [&]() {
        // through leakyrelu and the input FC (models/lstm.py:82-87); bufA holds d(h0)
        launch_lrelu_bwd(g_h0, bufA, (size_t)R, P, ldP, cfg.lrelu_alpha, s);
        gemm(x_tm, ldDin, false, bufA, ldP, false, G.Gd(g_fc_in_w), ldP, Din, P, R, nullptr, 0, 0.f, false, s);
        launch_colsum(bufA, ldP, nullptr, 0, G.Gd(g_fc_in_b), R, P, scratch, s);
      });
      if (bucketed) mark_bucket(RSRGAN_NET_G, bi++, s);
    }
    if (bucketed) {        // LSTM weight gradients layer by layer: each completes one bucket (all-reduced while the next runs)
      // (same-shaped layers: every layer's dWp and column sums first, as one launch per kind -- decided on the host, a replayed
      //  segment does not run its body -- then a layer's bucket is complete behind its dK GEMM)
      bool dkd = false;
      const bool gb = batch_wgrads(bw_chains[1], T, s, false, &dkd, true);
      if (gb) run_seg(seg_key(SEG_G_BATCH, T, kbits), s, [&]() { bool d_ = false; (void)batch_wgrads(bw_chains[1], T, s, false, &d_); });
      int li = 0;
      for (auto& Rr : bw_chains[1]) {
        run_seg(seg_key(SEG_G_LAYER0 + li, T, kbits), s, [&]() { if (gb) layer_wgrads_gemms(Rr, 0, T, false, s, true, false); else layer_wgrads(Rr, T, s); });
        mark_bucket(RSRGAN_NET_G, bi++, s);
        ++li;
      }
    }
  }
  if (!inl) run_seg(seg_key(SEG_G_TAIL, T, kbits), s, tail_body);
  if (want_grads) {
    if (l2_on) for (auto& bk : gbk[RSRGAN_NET_G]) bk.marked = false;      // the L2 term touches every tensor: all buckets final only now
    finish_buckets(RSRGAN_NET_G, s);
    g_grads_ready = true;
  }
  if (out_losses) launch_copy_f(losses + 3, out_losses, 4, s);
  HIPC(hipGetLastError());
  return RSRGAN_OK;
}

// per-tensor clip_by_norm + the optimizer + EMA + the weight copies (gan_rnn_placeholder.py:177-189): the launches of one update
void Model::apply_body(int net, hipStream_t s) {
  if (net == RSRGAN_NET_D) {
    if (!d_partial_fresh) launch_sumsq(D.g, D.ct, D.partial, s);      // (k_dw_reduce of the same segment has left the sums of squares)
    d_partial_fresh = false;
    if (d_adam()) {                              // models/gan.py:125: d_opt = AdamOptimizer(d_learning_rate)
      launch_adam_tick(dyn, adam_t_dev_d, (double)cfg.adam_beta1, (double)cfg.adam_beta2, s, DYN_D_LR, DYN_ADAM_LRT_D);
      launch_apply_adam(D.w, D.g, D.m, D.v, D.ema, D.ct, D.partial, dyn, s, DYN_ADAM_LRT_D);
    } else {
      launch_apply_sgd(D.w, D.g, D.ema, D.ct, D.partial, dyn, s);
    }
    refresh_transposes(RSRGAN_NET_D, s);
  } else {
    launch_sumsq(G.g, G.ct, G.partial, s);
    launch_adam_tick(dyn, adam_t_dev, (double)cfg.adam_beta1, (double)cfg.adam_beta2, s);
    launch_apply_adam(G.w, G.g, G.m, G.v, G.ema, G.ct, G.partial, dyn, s);
    refresh_transposes(RSRGAN_NET_G, s);
  }
}

int Model::apply(int net, hipStream_t s) {
  if (net == RSRGAN_NET_D) {
    if (!d_grads_ready) { set_error("apply(D) without gradients"); return RSRGAN_ERR_STATE; }
    // (apply_inlined: rsrgan_d_step's backward segment already ran / replayed these launches at its end -- one graph, no boundary)
    if (!(apply_inlined & 2)) run_seg(seg_key(SEG_APPLY_D, 0, 0), s, [&]() { apply_body(RSRGAN_NET_D, s); });
    apply_inlined &= ~2;
    if (d_adam()) scal[RSRGAN_ADAM_STEP_D] += 1;
    d_grads_ready = false;
  } else if (net == RSRGAN_NET_G) {
    if (!g_grads_ready) { set_error("apply(G) without gradients"); return RSRGAN_ERR_STATE; }
    if (!(apply_inlined & 1)) run_seg(seg_key(SEG_APPLY_G, 0, 0), s, [&]() { apply_body(RSRGAN_NET_G, s); });
    // (RSRGAN_DPIPE: the generator's update touches nothing the next D(real) reads or writes, so the event stays where the G-run's
    // backward launch -- or the previous call's end -- recorded it; a run that has used the discriminator's stash since and has NOT
    // recorded it leaves the record to the end of this call)
    if (dfree_current) dfree_inside = true;
    apply_inlined &= ~1;
    scal[RSRGAN_ADAM_STEP] += 1;
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
