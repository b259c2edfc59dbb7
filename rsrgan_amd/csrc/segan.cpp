// segan.cpp -- the SEGAN-style conv G/D training runs (models/segan.py:SEGAN.build_model_single_gpu :155-236 with
// generator.py:AEGenerator :112-295, discriminator.py:discriminator :20-95, utils/bnorm.py:VBN :11-69; BASELINE.json configs[4]).
//
// Everything is channels-last [batch x position][channels] fp32.  A stride-2 SAME convolution (utils/ops.py:downconv) is the
// GEMM of an overlapping-window VIEW of its zero-padded input with the filter tensor as it stands ([kwidth, 1, Cin, Cout] =
// [kwidth*Cin][Cout]): no patch matrix exists.  conv2d_transpose (utils/ops.py:deconv) and the data gradient of a downconv
// are the same operation; an output position i receives the taps dk = i + pl - 2o, i.e. only taps of ITS parity, so the
// positions of each parity class are a stride-1 window GEMM over the source (ceil(k/2) or floor(k/2) taps) with a flipped,
// transposed filter prepared once per optimizer step -- again no patch matrix and no scatter.  The weight gradients are the
// transposed views (k-major operand walked through the same row map).  The single-channel ends of both nets are direct kernels.
#include "segan.h"

#include <cmath>
#include <cstring>
#include <random>

namespace rsr {

#define HIPS(expr)                                                                   \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RSRGAN_ERR_HIP;                                                         \
    }                                                                                \
  } while (0)

template <typename T> T* SeganModel::alloc(size_t n_) {
  void* p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(n_, 1) * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemset(p, 0, std::max<size_t>(n_, 1) * sizeof(T));
  allocs.push_back(p);
  return (T*)p;
}
void SeganModel::destroy() {
  for (void* p : allocs) (void)hipFree(p);
  allocs.clear();
}

static std::vector<int> lengths(int L, int n) {
  std::vector<int> v{L};
  for (int i = 0; i < n; ++i) v.push_back((v.back() + 1) / 2);
  return v;
}
struct TGeom { int pl, ne[2], i0[2], q0[2], Q[2], pf, pb; };
static TGeom tgeom(int Ls, int Lt, int k) {
  TGeom g{};
  g.pl = same_pad(Lt, k).pl;
  g.pf = 0; g.pb = 0;
  for (int e = 0; e < 2; ++e) {
    g.ne[e] = (k - e + 1) / 2;
    g.i0[e] = (((e - g.pl) % 2) + 2) % 2;
    g.q0[e] = (g.i0[e] + g.pl - e) / 2;
    g.Q[e] = Lt > g.i0[e] ? (Lt - g.i0[e] + 1) / 2 : 0;
    if (g.Q[e] > 0 && g.ne[e] > 0) {
      g.pf = std::max(g.pf, g.ne[e] - 1 - g.q0[e]);
      g.pb = std::max(g.pb, g.q0[e] + g.Q[e] - 1 - (Ls - 1));
    }
  }
  return g;
}

int SeganModel::init(const rsrgan_segan_cfg& c, uint64_t seed) {
  cfg = c;
  B = c.batch_size; Lx = c.input_len; U = c.output_dim; n = c.n_layers; Lj = Lx + U;
  if (B <= 0 || Lx <= 0 || U <= 0 || n < 2 || n > 16 || c.g_kwidth < 2 || c.d_kwidth < 2) { set_error("invalid sizes in rsrgan_segan_cfg"); return RSRGAN_ERR_INVALID; }
  for (int i = 0; i < n; ++i)
    if (c.g_depths[i] <= 0 || c.d_depths[i] <= 0 || (c.g_depths[i] & 15) || (c.d_depths[i] & 15)) {
      set_error("conv depths must be positive multiples of 16"); return RSRGAN_ERR_INVALID;
    }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device"); return RSRGAN_ERR_NO_DEVICE; }
  const std::vector<int> Le = lengths(Lx, n), Dl = lengths(Lj, n);
  Ld = Dl[n];
  const int gk = c.g_kwidth, dk = c.d_kwidth;
  char nm[128];
  // ---- variable tables, graph-construction order (oracle/segan_oracle.py g_param_specs / d_param_specs)
  enc.resize(n); dec.resize(n); blk.resize(n);
  for (int i = 0; i < n; ++i) {
    SgLayer& L = enc[i];
    L.Cin = i ? c.g_depths[i - 1] : 1; L.Cout = c.g_depths[i]; L.Lin = Le[i]; L.Lout = Le[i + 1]; L.k = gk;
    snprintf(nm, sizeof nm, "g_ae/enc_%d/W", i); L.tW = G.add(nm, gk * L.Cin, L.Cout, false);
    snprintf(nm, sizeof nm, "g_ae/enc_%d/b", i); L.tb = G.add(nm, 1, L.Cout, true);
    if (c.g_prelu) { snprintf(nm, sizeof nm, "g_ae/enc_prelu_%d/alpha", i); L.ta = G.add(nm, 1, L.Cout, true); }
  }
  int cin = 2 * c.g_depths[n - 1];
  for (int j = 0; j < n; ++j) {
    SgLayer& L = dec[j];
    L.Cin = cin; L.Cout = j < n - 1 ? c.g_depths[n - 2 - j] : 1; L.Lin = Le[n - j]; L.Lout = Le[n - 1 - j]; L.k = gk;
    snprintf(nm, sizeof nm, "g_ae/dec_%d/W", j); L.tW = G.add(nm, gk * L.Cout, L.Cin, false);
    snprintf(nm, sizeof nm, "g_ae/dec_%d/b", j); L.tb = G.add(nm, 1, L.Cout, true);
    if (j < n - 1) {
      if (c.g_prelu) { snprintf(nm, sizeof nm, "g_ae/dec_prelu_%d/alpha", j); L.ta = G.add(nm, 1, L.Cout, true); }
      cin = 2 * L.Cout;
    }
  }
  t_dense_w = G.add("g_ae/dense/kernel", Lx, U, false);
  t_dense_b = G.add("g_ae/dense/bias", 1, U, true);
  for (int i = 0; i < n; ++i) {
    SgLayer& L = blk[i];
    L.Cin = i ? c.d_depths[i - 1] : 1; L.Cout = c.d_depths[i]; L.Lin = Dl[i]; L.Lout = Dl[i + 1]; L.k = dk;
    snprintf(nm, sizeof nm, "d_model/d_block_%d/downconv/W", i); L.tW = D.add(nm, dk * L.Cin, L.Cout, false);
    snprintf(nm, sizeof nm, "d_model/d_block_%d/downconv/b", i); L.tb = D.add(nm, 1, L.Cout, true);
    snprintf(nm, sizeof nm, "d_model/d_block_%d/d_vbn_%d/gamma", i, i); L.tg = D.add(nm, 1, L.Cout, true);
    snprintf(nm, sizeof nm, "d_model/d_block_%d/d_vbn_%d/beta", i, i); L.tbeta = D.add(nm, 1, L.Cout, true);
  }
  t_lc = D.add("d_model/logits_conv/W", 1, dk * c.d_depths[n - 1], true);
  t_fcw = D.add("d_model/fully_connected/weights", Ld, 1, false);
  t_fcb = D.add("d_model/fully_connected/biases", 1, 1, true);
  for (ParamSet* ps : {&G, &D}) {
    ps->w = alloc<float>(ps->padded); ps->g = alloc<float>(ps->padded); ps->v = alloc<float>(ps->padded);
    if (!ps->w || !ps->g || !ps->v) { set_error("hipMalloc failed (parameters)"); return RSRGAN_ERR_HIP; }
    // truncated_normal(0.02) filters, zero biases / alpha / beta, gamma ~ N(1, 0.02), xavier dense / FC; rms slots = 1
    std::vector<float> h(ps->padded, 0.f), ones(ps->padded, 1.f);
    std::mt19937_64 rng(seed + (ps == &D ? 7919 : 0));
    std::normal_distribution<float> nd(0.f, 0.02f);
    for (const TensorDesc& t : ps->t) {
      const bool filt = t.name.size() > 2 && t.name.compare(t.name.size() - 2, 2, "/W") == 0;
      const bool gam = t.name.find("gamma") != std::string::npos;
      const bool dense = t.name.find("kernel") != std::string::npos || t.name.find("weights") != std::string::npos;
      const float lim = dense ? std::sqrt(6.f / (t.rows + t.cols)) : 0.f;
      std::uniform_real_distribution<float> ud(-lim, lim);
      for (int r = 0; r < t.rows; ++r)
        for (int cc = 0; cc < t.cols; ++cc) {
          float v = 0.f;
          if (filt) { do v = nd(rng); while (std::fabs(v) > 0.04f); }
          else if (gam) v = 1.f + nd(rng);
          else if (dense) v = ud(rng);
          h[t.off + (size_t)r * t.ld + cc] = v;
        }
    }
    HIPS(hipMemcpy(ps->w, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPS(hipMemcpy(ps->v, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  // ---- parity-class filters of the transposed convolutions
  auto prep_alloc = [&](SgLayer& L, int na, int nb) {      // Wt_e [(ne * na)][ld(nb)]
    for (int e = 0; e < 2; ++e) { L.ne[e] = (L.k - e + 1) / 2; L.Wt[e] = alloc<float>((size_t)L.ne[e] * na * pad4(nb)); }
  };
  for (int i = 1; i < n; ++i) { prep_alloc(enc[i], enc[i].Cout, enc[i].Cin); prep_alloc(blk[i], blk[i].Cout, blk[i].Cin); }
  for (int j = 0; j < n - 1; ++j) prep_alloc(dec[j], dec[j].Cin, dec[j].Cout);
  // ---- activations
  z_e.resize(n); a_e.resize(n); xd.resize(n); zd.resize(n);
  size_t gmax = (size_t)B * pad4(Lx);
  for (int i = 0; i < n; ++i) {
    const size_t e = (size_t)B * enc[i].Lout * enc[i].Cout;
    z_e[i] = alloc<float>(e); a_e[i] = alloc<float>(e);
    xd[i] = alloc<float>((size_t)B * dec[i].Lin * dec[i].Cin);
    zd[i] = i < n - 1 ? alloc<float>((size_t)B * dec[i].Lout * dec[i].Cout) : nullptr;
    gmax = std::max({gmax, e, (size_t)B * dec[i].Lin * dec[i].Cin, (size_t)B * dec[i].Lout * dec[i].Cout});
  }
  wave = alloc<float>((size_t)B * pad4(Lx)); Gy = alloc<float>((size_t)B * pad4(U));
  dh_.resize(n); dy_.resize(n); coef.resize(n);
  for (int i = 0; i < n; ++i) {
    const size_t e = (size_t)3 * B * blk[i].Lout * blk[i].Cout;
    dh_[i] = alloc<float>(e); dy_[i] = alloc<float>(e); coef[i] = alloc<float>((size_t)3 * 8 * blk[i].Cout);
    gmax = std::max(gmax, e);
  }
  joint = alloc<float>((size_t)3 * B * Lj); conv_out = alloc<float>((size_t)3 * B * Ld);
  logits = alloc<float>(3 * B + 4); dlogits = alloc<float>(3 * B + 4); djoint = alloc<float>((size_t)B * Lj);
  // ---- scratch: padded copies (window views), parity outputs, gradient ping-pong
  size_t pmax = 0, tmax = 0;
  auto use_conv = [&](int Bn, int L, int C, int k) { const SameGeom g = same_pad(L, k); pmax = std::max(pmax, (size_t)Bn * (g.pl + L + g.pr) * C); };
  auto use_tconv = [&](int Bn, int Ls, int Cs, int Lt, int Ct, int k) {
    const TGeom g = tgeom(Ls, Lt, k);
    pmax = std::max(pmax, (size_t)Bn * (g.pf + Ls + g.pb) * Cs);
    tmax = std::max(tmax, (size_t)Bn * std::max(g.Q[0], g.Q[1]) * Ct);
  };
  for (int i = 1; i < n; ++i) {
    use_conv(B, enc[i].Lin, enc[i].Cin, gk); use_tconv(B, enc[i].Lout, enc[i].Cout, enc[i].Lin, enc[i].Cin, gk);
    use_conv(3 * B, blk[i].Lin, blk[i].Cin, dk); use_tconv(3 * B, blk[i].Lout, blk[i].Cout, blk[i].Lin, blk[i].Cin, dk);
  }
  for (int j = 0; j < n - 1; ++j) { use_tconv(B, dec[j].Lin, dec[j].Cin, dec[j].Lout, dec[j].Cout, gk); use_conv(B, dec[j].Lout, dec[j].Cout, gk); }
  pad_floats = pmax + 64; t_floats = tmax + 64; g_floats = gmax + 64;
  pad = alloc<float>(pad_floats); t0 = alloc<float>(t_floats); t1 = alloc<float>(t_floats);
  gA = alloc<float>(g_floats); gB = alloc<float>(g_floats); gC = alloc<float>(g_floats);
  sums = alloc<float>(3 * 2 * 2048);
  red_floats = (size_t)4 << 20; red = alloc<float>(red_floats);
  gemm_ws_floats = (size_t)16 << 20; gemm_ws = alloc<float>(gemm_ws_floats);
  dyn = alloc<float>(8); losses = alloc<float>(8);
  if (!pad || !t0 || !t1 || !gA || !gB || !gC || !red || !gemm_ws || !dyn || !losses || !djoint) { set_error("hipMalloc failed (activations)"); return RSRGAN_ERR_HIP; }
  scal[RSRGAN_SEGAN_G_LR] = 1e-3; scal[RSRGAN_SEGAN_D_LR] = 1e-3; scal[RSRGAN_SEGAN_L1_LAMBDA] = 100.0;   // run_segan.sh:103-104,117
  const float d3[3] = {1e-3f, 1e-3f, 100.f};
  HIPS(hipMemcpy(dyn, d3, sizeof d3, hipMemcpyHostToDevice));
  refresh_weights(RSRGAN_NET_G, nullptr);
  refresh_weights(RSRGAN_NET_D, nullptr);
  HIPS(hipDeviceSynchronize());
  return RSRGAN_OK;
}

void SeganModel::refresh_weights(int net, hipStream_t s) {
  PrepTconvList pl{};
  auto add = [&](const float* W, int ldw, int nb, int na, int e, int ne, float* dst, int ldd) {
    if (pl.n == 44) { launch_prep_tconv_many(pl, s); pl.n = 0; }
    pl.j[pl.n++] = PrepTconvJob{W, dst, ldw, nb, na, e, ne, ldd};
  };
  auto prep_down = [&](const ParamSet& ps, SgLayer& L) {   // data gradient of a downconv: Wt[(rr, co)][ci] = W[dk][ci][co]
    for (int e = 0; e < 2; ++e) add(ps.W(L.tW), ps.t[L.tW].ld, L.Cin, L.Cout, e, L.ne[e], L.Wt[e], pad4(L.Cin));
  };
  if (net == RSRGAN_NET_G) {
    for (int i = 1; i < n; ++i) prep_down(G, enc[i]);
    for (int j = 0; j < n - 1; ++j)                        // deconv forward: Wt[(rr, ci)][co] = W[dk][co][ci]
      for (int e = 0; e < 2; ++e) add(G.W(dec[j].tW), G.t[dec[j].tW].ld, dec[j].Cout, dec[j].Cin, e, dec[j].ne[e], dec[j].Wt[e], pad4(dec[j].Cout));
  } else {
    for (int i = 1; i < n; ++i) prep_down(D, blk[i]);
  }
  launch_prep_tconv_many(pl, s);
}

// ---- primitives
void SeganModel::conv2_fwd(const float* X, int Bn, int L, int Cin, int k, const float* W, int ldw, const float* bias, int Cout, float* Z, hipStream_t s) {
  const SameGeom g = same_pad(L, k);
  launch_pad_rows(X, pad, Bn, L, Cin, g.pl, g.pr, s);
  const GemmRowMap map{g.out, (long long)(g.pl + L + g.pr) * Cin, 2LL * Cin};
  launch_gemm_mapped(pad, 0, map, nullptr, 0, 0, true, W, ldw, false, Z, pad4(Cout), Bn * g.out, Cout, k * Cin, bias, 0, 0.f, false, s, gemm_ws, gemm_ws_floats);
}
void SeganModel::conv2_wgrad(const float* X, int Bn, int L, int Cin, int k, const float* dZ, int ldz, int Cout, float* dW, int ldw, hipStream_t s) {
  const SameGeom g = same_pad(L, k);
  launch_pad_rows(X, pad, Bn, L, Cin, g.pl, g.pr, s);
  const GemmRowMap map{g.out, (long long)(g.pl + L + g.pr) * Cin, 2LL * Cin};
  launch_gemm_mapped(pad, 0, map, nullptr, 0, 0, false, dZ, ldz, false, dW, ldw, k * Cin, Cout, Bn * g.out, nullptr, 0, 0.f, false, s, gemm_ws, gemm_ws_floats);
}
void SeganModel::tconv2(const float* S, int Bn, int Ls, int Cs, int Lt, int k, float* const Wt[2], const int ne[2], int Ct, const float* bias, float* T,
                        hipStream_t s) {
  const TGeom g = tgeom(Ls, Lt, k);
  launch_pad_rows(S, pad, Bn, Ls, Cs, g.pf, g.pb, s);
  float* outs[2] = {t0, t1};
  for (int e = 0; e < 2; ++e) {
    if (g.Q[e] <= 0) continue;
    const GemmRowMap map{g.Q[e], (long long)(g.pf + Ls + g.pb) * Cs, (long long)Cs};
    const float* A = pad + (size_t)(g.pf + g.q0[e] - (ne[e] - 1)) * Cs;
    launch_gemm_mapped(A, 0, map, nullptr, 0, 0, true, Wt[e], pad4(Ct), false, outs[e], pad4(Ct), Bn * g.Q[e], Ct, ne[e] * Cs, nullptr, 0, 0.f, false, s,
                       gemm_ws, gemm_ws_floats);
  }
  launch_interleave(t0, t1, std::max(g.Q[0], 1), std::max(g.Q[1], 1), g.i0[0], g.i0[1], g.pl, bias, T, Bn, Lt, Ct, s);
}

// ---- generator forward (generator.py:112-295)
void SeganModel::g_forward(const float* x, const float* z, hipStream_t s) {
  const float leak = cfg.lrelu_alpha;
  const int Cl = enc[n - 1].Cout;
  launch_copy_cols(z, Cl, 0, xd[0], 2 * Cl, 0, Cl, (size_t)B * enc[n - 1].Lout, false, s);      // h = concat([z, code], 2)  :205
  for (int i = 0; i < n; ++i) {
    const SgLayer& L = enc[i];
    if (i == 0) launch_conv1_fwd(x, Lx, B, Lx, L.k, G.W(L.tW), G.t[L.tW].ld, G.W(L.tb), L.Cout, z_e[0], L.Cout, s);
    else conv2_fwd(a_e[i - 1], B, L.Lin, L.Cin, L.k, G.W(L.tW), G.t[L.tW].ld, G.W(L.tb), L.Cout, z_e[i], s);
    const float* al = L.ta >= 0 ? G.W(L.ta) : nullptr;
    const size_t rows = (size_t)B * L.Lout;
    if (i < n - 1) launch_act_fwd(z_e[i], L.Cout, al, leak, a_e[i], L.Cout, 0, rows, s);
    else launch_act_fwd(z_e[i], L.Cout, al, leak, xd[0], 2 * Cl, Cl, rows, s);
  }
  for (int j = 0; j < n - 1; ++j) {
    const SgLayer& L = dec[j];
    tconv2(xd[j], B, L.Lin, L.Cin, L.Lout, L.k, L.Wt, L.ne, L.Cout, G.W(L.tb), zd[j], s);
    const size_t rows = (size_t)B * L.Lout;
    launch_act_fwd(zd[j], L.Cout, L.ta >= 0 ? G.W(L.ta) : nullptr, leak, xd[j + 1], 2 * L.Cout, 0, rows, s);
    launch_copy_cols(z_e[n - 2 - j], L.Cout, 0, xd[j + 1], 2 * L.Cout, L.Cout, L.Cout, rows, false, s);      // concat([h, skip], 2)  :272
  }
  const SgLayer& L = dec[n - 1];
  launch_tconv1(xd[n - 1], L.Cin, B, L.Lin, L.Cin, Lx, L.k, G.W(L.tW), G.t[L.tW].ld, G.W(L.tb), wave, pad4(Lx), s);
  launch_gemm(wave, pad4(Lx), true, G.W(t_dense_w), pad4(U), false, Gy, pad4(U), B, U, Lx, G.W(t_dense_b), 0, 0.f, false, s, gemm_ws, gemm_ws_floats);
  g_fwd_valid = true;
}

// ---- discriminator on the P passes stacked in `joint` (pass 0 = the reference pass)
void SeganModel::d_forward(int P, hipStream_t s) {
  const int Bn = P * B;
  for (int i = 0; i < n; ++i) {
    const SgLayer& L = blk[i];
    if (i == 0) launch_conv1_fwd(joint, Lj, Bn, Lj, L.k, D.W(L.tW), D.t[L.tW].ld, D.W(L.tb), L.Cout, dh_[0], L.Cout, s);
    else conv2_fwd(dy_[i - 1], Bn, L.Lin, L.Cin, L.k, D.W(L.tW), D.t[L.tW].ld, D.W(L.tb), L.Cout, dh_[i], s);
    const size_t rows_per = (size_t)B * L.Lout;
    launch_colred(2, dh_[i], L.Cout, 0, nullptr, 0, L.Cout, rows_per, P, nullptr, 0, 0.f, sums, L.Cout, false, red, red_floats, s);
    launch_vbn_coef(sums, L.Cout, P, L.Cout, rows_per, B, cfg.vbn_eps, D.W(L.tg), D.W(L.tbeta), nullptr, coef[i], L.Cout, s);
    launch_vbn_apply(dh_[i], L.Cout, rows_per, P, coef[i], L.Cout, cfg.lrelu_alpha, dy_[i], s);
  }
  launch_dhead_fwd(dy_[n - 1], Bn, Ld, blk[n - 1].Cout, cfg.d_kwidth, D.W(t_lc), D.W(t_fcw), D.t[t_fcw].ld, D.W(t_fcb), conv_out, logits, s);
}

// backward through the passes [p0, P) (dlogits given); wgrads: parameter gradients (D-run); need_dx: d loss / d joint of those passes -> djoint
void SeganModel::d_backward_pass(int P, int p0, bool wgrads, bool need_dx, hipStream_t s) {
  const int np = P - p0, nB = np * B;
  const int Cl = blk[n - 1].Cout;
  float* d = gA;
  float* other = gB;
  launch_dhead_bwd(dlogits + (size_t)p0 * B, nB, Ld, Cl, cfg.d_kwidth, dy_[n - 1] + (size_t)p0 * B * Ld * Cl, conv_out + (size_t)p0 * B * Ld, D.W(t_lc), D.W(t_fcw),
                   D.t[t_fcw].ld, wgrads ? D.Gd(t_lc) : nullptr, wgrads ? D.Gd(t_fcw) : nullptr, wgrads ? D.Gd(t_fcb) : nullptr, d, s);
  for (int i = n - 1; i >= 0; --i) {
    const SgLayer& L = blk[i];
    const int C = L.Cout;
    const size_t rows_per = (size_t)B * L.Lout, off = (size_t)p0 * rows_per * C;
    float* cf = coef[i] + (size_t)p0 * 8 * C;
    launch_colred(3, dh_[i] + off, C, 0, d, C, C, rows_per, np, cf, C, cfg.lrelu_alpha, sums, C, false, red, red_floats, s);
    launch_vbn_bwd_coef(sums, C, np, p0 == 0 ? 1 : 0, C, rows_per, B, D.W(L.tg), cf, C, wgrads ? D.Gd(L.tg) : nullptr, wgrads ? D.Gd(L.tbeta) : nullptr, false, s);
    launch_vbn_bwd_apply(dh_[i] + off, d, C, rows_per, np, cf, C, cfg.lrelu_alpha, other, s);           // other = d loss / d (conv output)
    if (wgrads) {
      launch_colred(0, other, C, 0, nullptr, 0, C, rows_per * np, 1, nullptr, 0, 0.f, D.Gd(L.tb), C, false, red, red_floats, s);
      if (i == 0) launch_conv1_wgrad(joint + (size_t)p0 * B * Lj, Lj, nB, Lj, L.k, other, C, C, D.Gd(L.tW), D.t[L.tW].ld, red, red_floats, s);
      else conv2_wgrad(dy_[i - 1] + (size_t)p0 * B * L.Lin * L.Cin, nB, L.Lin, L.Cin, L.k, other, C, C, D.Gd(L.tW), D.t[L.tW].ld, s);
    }
    if (i > 0) { tconv2(other, nB, L.Lout, C, L.Lin, L.k, L.Wt, L.ne, L.Cin, nullptr, d, s); }
    else if (need_dx) launch_tconv1(other, C, nB, L.Lout, C, Lj, L.k, D.W(L.tW), D.t[L.tW].ld, nullptr, djoint, Lj, s);
  }
}

// d g_loss / d G is in gC [B][ld(U)]: back through the dense head, the decoder and the encoder
void SeganModel::g_backward_pass(hipStream_t s) {
  const float leak = cfg.lrelu_alpha;
  const int ldU = pad4(U), ldL = pad4(Lx);
  float* dGy = gC;
  launch_gemm(wave, ldL, false, dGy, ldU, false, G.Gd(t_dense_w), ldU, Lx, U, B, nullptr, 0, 0.f, false, s, gemm_ws, gemm_ws_floats);
  launch_colred(0, dGy, ldU, 0, nullptr, 0, U, B, 1, nullptr, 0, 0.f, G.Gd(t_dense_b), ldU, false, red, red_floats, s);
  float* dwave = gA;                                       // [B][ld(Lx)]
  launch_gemm(dGy, ldU, true, G.W(t_dense_w), ldU, true, dwave, ldL, B, Lx, U, nullptr, 0, 0.f, false, s, gemm_ws, gemm_ws_floats);
  // last deconv (one output channel): dX = strided conv of dwave with the filter as [k][Cin]; dW[dk][ci] = view(dwave)^T . X
  {
    const SgLayer& L = dec[n - 1];
    launch_sum_all(dwave, B, Lx, ldL, G.Gd(L.tb), red, s);
    launch_conv1_wgrad(dwave, ldL, B, Lx, L.k, xd[n - 1], L.Cin, L.Cin, G.Gd(L.tW), G.t[L.tW].ld, red, red_floats, s);
    launch_conv1_fwd(dwave, ldL, B, Lx, L.k, G.W(L.tW), G.t[L.tW].ld, nullptr, L.Cin, gB, L.Cin, s);
  }
  float* d = gB;                                           // d loss / d xd[j + 1], [B * Lout_j][2 Cout_j]
  float* other = gA;
  for (int j = n - 2; j >= 0; --j) {
    const SgLayer& L = dec[j];
    const int C = L.Cout;
    const size_t rows = (size_t)B * L.Lout;
    const float* al = L.ta >= 0 ? G.W(L.ta) : nullptr;
    // the skip half of the gradient waits in xd[j + 1]'s own skip columns (its forward content, the copy of z_e, is no longer needed:
    // the weight gradient of dec_{j+1} has been taken)
    launch_copy_cols(d, 2 * C, C, xd[j + 1], 2 * C, C, C, rows, false, s);
    if (al) launch_colred(1, d, 2 * C, 0, zd[j], C, C, rows, 1, nullptr, 0, 0.f, G.Gd(L.ta), C, false, red, red_floats, s);
    launch_act_bwd(d, 2 * C, 0, zd[j], C, al, leak, nullptr, other, rows, s);                   // other = d loss / d zd[j]
    launch_colred(0, other, C, 0, nullptr, 0, C, rows, 1, nullptr, 0, 0.f, G.Gd(L.tb), C, false, red, red_floats, s);
    conv2_wgrad(other, B, L.Lout, C, L.k, xd[j], L.Cin, L.Cin, G.Gd(L.tW), G.t[L.tW].ld, s);    // dW[(dk, co)][ci] = view(dzd)^T . X
    conv2_fwd(other, B, L.Lout, C, L.k, G.W(L.tW), G.t[L.tW].ld, nullptr, L.Cin, d, s);          // d loss / d xd[j]: the strided conv of dzd
  }
  // d = d loss / d xd[0] [B * L_code][2 Cl]: columns [Cl, 2 Cl) belong to prelu(enc_{n-1})
  const float* dy = d;
  int ldy = 2 * enc[n - 1].Cout, coff = enc[n - 1].Cout;
  for (int i = n - 1; i >= 0; --i) {
    const SgLayer& L = enc[i];
    const int C = L.Cout;
    const size_t rows = (size_t)B * L.Lout;
    const float* al = L.ta >= 0 ? G.W(L.ta) : nullptr;
    if (al) launch_colred(1, dy, ldy, coff, z_e[i], C, C, rows, 1, nullptr, 0, 0.f, G.Gd(L.ta), C, false, red, red_floats, s);
    float* dz = (dy == gA) ? gB : gA;
    launch_act_bwd(dy, ldy, coff, z_e[i], C, al, leak, nullptr, dz, rows, s);
    if (i < n - 1) launch_copy_cols(xd[n - 1 - i], 2 * C, C, dz, C, 0, C, rows, true, s);        // + the skip path (decoder j = n-2-i)
    launch_colred(0, dz, C, 0, nullptr, 0, C, rows, 1, nullptr, 0, 0.f, G.Gd(L.tb), C, false, red, red_floats, s);
    if (i == 0) { launch_conv1_wgrad(xin, Lx, B, Lx, L.k, dz, C, C, G.Gd(L.tW), G.t[L.tW].ld, red, red_floats, s); break; }
    conv2_wgrad(a_e[i - 1], B, L.Lin, L.Cin, L.k, dz, C, C, G.Gd(L.tW), G.t[L.tW].ld, s);
    float* dx = (dz == gA) ? gB : gA;
    tconv2(dz, B, L.Lout, C, L.Lin, L.k, L.Wt, L.ne, L.Cin, nullptr, dx, s);
    dy = dx; ldy = L.Cin; coff = 0;
  }
}

static const float* stage_in(const float* p) { return p; }

// sess.run([model.d_opt, model.d_losses[0]]) (scripts/train_segan.py:32-36): the gradients of d_loss w.r.t. the d_ variables
int SeganModel::d_run(const float* x, const float* labels, const float* z, const float* n_ref, const float* n_real, const float* n_fake, float* out_losses,
                      bool want_grads, hipStream_t s) {
  if (!x || !labels || !z) { set_error("null input pointer"); return RSRGAN_ERR_INVALID; }
  xin = const_cast<float*>(stage_in(x)); lab = const_cast<float*>(labels);
  g_forward(x, z, s);
  // gC <- G as a packed [B][U] tail for the fake joint
  launch_copy_cols(Gy, pad4(U), 0, gC, U, 0, U, B, false, s);
  launch_build_joint1(x, Lx, labels, U, n_ref, joint, B, s);                                    // the dummy pass  segan.py:186-188
  launch_build_joint1(x, Lx, labels, U, n_real, joint + (size_t)B * Lj, B, s);                  // D_rl_joint      :204
  launch_build_joint1(x, Lx, gC, U, n_fake, joint + (size_t)2 * B * Lj, B, s);                  // D_fk_joint      :205
  d_forward(3, s);
  launch_segan_lsgan(logits, B, 0, 2, 3, want_grads ? dlogits : nullptr, losses, s);
  if (want_grads) { d_backward_pass(3, 0, true, false, s); d_grads_ready = true; }
  if (out_losses) launch_copy_f(losses, out_losses, 3, s);
  HIPS(hipGetLastError());
  return RSRGAN_OK;
}

// sess.run([model.g_opt, model.g_losses[0]]) (:40-44): g_loss = g_adv + l1_lambda * mean|G - labels| w.r.t. the g_ variables
int SeganModel::g_run(const float* x, const float* labels, const float* z, const float* n_ref, const float* n_fake, float* out_losses, bool want_grads,
                      hipStream_t s) {
  if (!x || !labels || !z) { set_error("null input pointer"); return RSRGAN_ERR_INVALID; }
  xin = const_cast<float*>(stage_in(x)); lab = const_cast<float*>(labels);
  g_forward(x, z, s);
  launch_copy_cols(Gy, pad4(U), 0, gC, U, 0, U, B, false, s);
  launch_build_joint1(x, Lx, labels, U, n_ref, joint, B, s);
  launch_build_joint1(x, Lx, gC, U, n_fake, joint + (size_t)B * Lj, B, s);
  d_forward(2, s);                                                                              // reference pass + fake
  launch_segan_lsgan(logits, B, 1, 1, 2, want_grads ? dlogits : nullptr, losses + 3, s);
  if (want_grads) {
    d_backward_pass(2, 1, false, true, s);                                                      // d g_adv / d joint_fake -> djoint
    launch_copy_cols(djoint, Lj, Lx, gC, pad4(U), 0, U, B, false, s);                           // ... / d G
  }
  // the L1 term on packed [B*U] views: Gy has leading dimension ld(U); U % 4 == 0 keeps it packed, otherwise go through gA
  if (pad4(U) == U) {
    launch_segan_l1(Gy, labels, B * U, dyn + RSRGAN_SEGAN_L1_LAMBDA, want_grads ? gC : nullptr, true, losses + 3, s);
  } else {
    launch_copy_cols(Gy, pad4(U), 0, gA, U, 0, U, B, false, s);
    if (want_grads) launch_copy_cols(gC, pad4(U), 0, gB, U, 0, U, B, false, s);
    launch_segan_l1(gA, labels, B * U, dyn + RSRGAN_SEGAN_L1_LAMBDA, want_grads ? gB : nullptr, true, losses + 3, s);
    if (want_grads) launch_copy_cols(gB, U, 0, gC, pad4(U), 0, U, B, false, s);
  }
  if (want_grads) { g_backward_pass(s); g_grads_ready = true; }
  if (out_losses) launch_copy_f(losses + 3, out_losses, 3, s);
  HIPS(hipGetLastError());
  return RSRGAN_OK;
}

int SeganModel::apply(int net, hipStream_t s) {
  ParamSet& ps = net == RSRGAN_NET_G ? G : D;
  bool& ready = net == RSRGAN_NET_G ? g_grads_ready : d_grads_ready;
  if (!ready) { set_error("apply without a backward pass"); return RSRGAN_ERR_STATE; }
  ready = false;
  launch_rmsprop(ps.w, ps.g, ps.v, dyn + (net == RSRGAN_NET_G ? RSRGAN_SEGAN_G_LR : RSRGAN_SEGAN_D_LR), cfg.rms_decay, cfg.rms_eps, (size_t)ps.padded, s);
  refresh_weights(net, s);
  HIPS(hipGetLastError());
  return RSRGAN_OK;
}

}  // namespace rsr
