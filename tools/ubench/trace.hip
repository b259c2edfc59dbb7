// trace.hip -- stand-alone timeline of the product's step kernels on ONE wavefront diagonal (not part of the product library).
// Compiles kernels.hip with RSR_TRACE: thread 0 of every workgroup stamps s_memtime at the phase boundaries.  For each kernel:
// launch span, start skew, workgroups per CU, and the mean duration of every phase per job class (heavy / light / D).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 trace.hip -o trace      Run: ./trace
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)

#define RSR_TRACE 1
#include "../../rsrgan_amd/csrc/kernels.hip"

using namespace rsr;

static std::vector<void*> g_bufs;
static float* dal(size_t n, float v) {
  float* p; CK(hipMalloc(&p, n * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = v * ((float)((i * 2654435761u >> 20) & 255) / 128.f - 1.f);
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  g_bufs.push_back(p);
  return p;
}

static void clear_trace(hipStream_t s) {
  void* p = nullptr;
  CK(hipGetSymbolAddress(&p, HIP_SYMBOL(rsr::g_trace)));
  CK(hipMemsetAsync(p, 0, sizeof(unsigned long long) * 8192 * 16, s));
  CK(hipStreamSynchronize(s));
}
constexpr int TRACE_BLOCKS = 2048;      // the launchers size their own grids: read back this many trace records (unused ones are zero)
struct Cls { const char* name; const void* id; };      // id = the job pointer the kernels stamp into trace slot 12

static void report(const char* kernel, int blocks, const std::vector<Cls>& cls, const char* const* phase_names, const int* phase_idx, int nph, float us_graph) {
  std::vector<unsigned long long> t((size_t)blocks * 16);
  CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(rsr::g_trace), t.size() * 8));
  unsigned long long rt0 = ~0ull, rt1 = 0;
  int ran = 0;
  for (int b = 0; b < blocks; ++b) {
    const unsigned long long* r = &t[(size_t)b * 16];
    if (r[9] == 0) continue;          // returned early (padding block)
    ++ran; rt0 = std::min(rt0, r[0]); rt1 = std::max(rt1, r[13]);
  }
  // s_memtime ticks per us from the longest block
  double tick_per_us = 100.0;
  {
    double best = 0;
    for (int b = 0; b < blocks; ++b) {
      const unsigned long long* r = &t[(size_t)b * 16];
      if (r[9] == 0 || r[13] <= r[0]) continue;
      const double us = (double)(r[13] - r[0]) / 100.0;
      if (us > best) { best = us; tick_per_us = (double)(r[9] - r[1]) / us; }
    }
  }
  printf("== %s: %d blocks (%d do work), %.2f us per launch in a graph; traced launch: first start -> last end %.2f us, s_memtime %.0f ticks/us\n", kernel, blocks, ran,
         us_graph, (double)(rt1 - rt0) / 100.0, tick_per_us);
  // workgroups per CU (xcc, se, sh, cu from HW_ID)
  std::map<unsigned, int> per_cu;
  for (int b = 0; b < blocks; ++b) {
    const unsigned long long* r = &t[(size_t)b * 16];
    if (r[9] == 0) continue;
    const unsigned hw = (unsigned)r[14], xcc = (unsigned)r[15] & 15;
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  {   // which blocks share a CU?  (b, b + 256) pairs = the dispatcher fills CUs in block order, one workgroup per CU per round
    std::map<unsigned, std::vector<int>> ids;
    for (int b = 0; b < blocks; ++b) {
      const unsigned long long* r = &t[(size_t)b * 16];
      if (r[9] == 0) continue;
      const unsigned hw = (unsigned)r[14], xcc = (unsigned)r[15] & 15;
      ids[(xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)].push_back(b);
    }
    int pairs = 0, p256 = 0;
    for (auto& kv : ids) if (kv.second.size() == 2) { ++pairs; p256 += std::abs(kv.second[0] - kv.second[1]) == 256; }
    if (pairs) {
      printf("   CUs with two blocks: %d, of which ids differ by exactly 256: %d ; examples:", pairs, p256);
      int shown = 0;
      for (auto& kv : ids) if (kv.second.size() == 2 && shown++ < 6) printf(" (%d,%d)", kv.second[0], kv.second[1]);
      printf("\n");
    }
  }
  int hist[8] = {0};
  for (auto& kv : per_cu) hist[std::min(kv.second, 7)]++;
  printf("   CUs used %zu ; CUs with 1/2/3/4+ working blocks: %d / %d / %d / %d\n", per_cu.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
  for (const Cls& c : cls) {
    double start_min = 1e30, start_max = 0, end_max = 0, dur = 0;
    std::vector<double> ph(nph, 0.0);
    int n = 0;
    for (int b = 0; b < blocks; ++b) {
      const unsigned long long* r = &t[(size_t)b * 16];
      if (r[9] == 0 || r[12] != (unsigned long long)(size_t)c.id) continue;
      ++n;
      const double st = (double)(r[0] - rt0) / 100.0, en = (double)(r[13] - rt0) / 100.0;
      start_min = std::min(start_min, st); start_max = std::max(start_max, st); end_max = std::max(end_max, en);
      dur += en - st;
      unsigned long long prev = r[1];
      for (int p = 0; p < nph; ++p) {
        const unsigned long long cur = r[phase_idx[p]];
        if (cur) { ph[p] += (double)(cur - prev) / tick_per_us; prev = cur; }
      }
    }
    if (!n) continue;
    printf("   %-22s %3d blocks: start %.2f..%.2f us, last end %.2f us, mean block %.2f us |", c.name, n, start_min, start_max, end_max, dur / n);
    for (int p = 0; p < nph; ++p) printf(" %s %.2f", phase_names[p], ph[p] / n);
    printf("\n");
  }
}

template <typename F>
static float time_graph(hipStream_t s, int per_graph, int replays, F f) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < per_graph; ++i) f();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1000.f / (per_graph * replays);
}

int main(int argc, char** argv) {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (const char* e = getenv("RSRGAN_BWD_A_FORM")) set_bwd_a_form(atoi(e));
  const int N = 64, H = 760, P = 280, HD = 256, PD = 40;
  int* len; CK(hipMalloc(&len, 256 * 4)); { std::vector<int> h(256, 1000); CK(hipMemcpy(len, h.data(), 256 * 4, hipMemcpyHostToDevice)); }

  // tiled weights (any content: timing only)
  auto tiles = [&](int nct, int nkb) { return dal(swizzle_floats(nct, nkb), 0.05f); };

  // ------------------------------------------------------------------ forward gates: 3 G layers (layer 0: x-part batched) + 4 D jobs
  {
    FwdGateJobs gj{}; gj.forget_bias = 1.f;
    int base = 0;
    std::vector<Cls> cls;
    auto mk = [&](int n, int h, int ldx, int ldm, bool zx, const char* name) {
      FwdGateJob& a = gj.j[gj.n++];
      a = FwdGateJob{};
      const int ncb = (h + 15) / 16, nkb = ((zx ? 0 : ldx) + ldm + 15) / 16;
      a.x = zx ? nullptr : dal((size_t)n * ldx, 1.f); a.ldx = ldx; a.m = dal((size_t)n * ldm, 1.f); a.ldm = ldm;
      a.Wsw = tiles(4 * ncb, nkb);
      a.zx = zx ? dal((size_t)n * 4 * h, 0.1f) : nullptr; a.bias = dal(4 * h, 0.1f);
      a.wf = dal(h, 0.1f); a.wi = dal(h, 0.1f); a.wo = dal(h, 0.1f);
      a.c_prev = dal((size_t)n * h, 0.5f); a.c_out = dal((size_t)n * h, 0.f); a.gates = dal((size_t)n * 4 * h, 0.f);
      a.ldh = (h + 3) & ~3; a.h = dal((size_t)n * a.ldh, 0.f); a.len = len; a.t = 0; a.N = n; a.H = h; a.nblk_c = ncb;
      cls.push_back(Cls{name, a.gates});
      base += job_blocks(ncb, n, 32);
    };
    mk(N, H, P, P, false, "G layer 2 (K=560)"); mk(N, H, P, P, false, "G layer 1 (K=560)"); mk(N, H, P, P, true, "G layer 0 (K=280)");
    for (int i = 0; i < 4; ++i) mk(N, HD, PD, PD, false, "D job (K=80)");
    const int kb = (P + P + 15) / 16;
    for (int i = 0; i < 20; ++i) launch_fwd_gates(gj, base, kb, s);
    CK(hipStreamSynchronize(s));
    const float us = time_graph(s, 50, 4, [&] { launch_fwd_gates(gj, base, kb, s); });
    clear_trace(s); launch_fwd_gates(gj, base, kb, s); CK(hipStreamSynchronize(s));
    const char* names[] = {"lookup", "issueW", "issueDMA", "issueEpi", "loads+bar", "mfma", "bar", "zs+bar", "epilogue"};
    const int idx[] = {2, 10, 11, 3, 4, 5, 6, 7, 9};
    report("k_fwd_gates (3 G layers + 4 D jobs)", TRACE_BLOCKS, cls, names, idx, 9, us);
  }
  // ------------------------------------------------------------------ forward gates: the folded discriminator alone (2 num_proj=None jobs, K = 296 / 512)
  {
    FwdGateJobs gj{}; gj.forget_bias = 1.f;
    int base = 0;
    std::vector<Cls> cls;
    auto mk = [&](int n, int h, int ldx, const char* name) {
      FwdGateJob& a = gj.j[gj.n++];
      a = FwdGateJob{};
      const int ldm = (h + 3) & ~3, ncb = (h + 15) / 16, nkb = (ldx + ldm + 15) / 16;
      a.x = dal((size_t)n * ldx, 1.f); a.ldx = ldx; a.m = dal((size_t)n * ldm, 1.f); a.ldm = ldm;
      a.Wsw = tiles(4 * ncb, nkb);
      a.zx = nullptr; a.bias = dal(4 * h, 0.1f);
      a.wf = dal(h, 0.1f); a.wi = dal(h, 0.1f); a.wo = dal(h, 0.1f);
      a.c_prev = dal((size_t)n * h, 0.5f); a.c_out = dal((size_t)n * h, 0.f); a.gates = dal((size_t)n * 4 * h, 0.f);
      a.ldh = ldm; a.h = dal((size_t)n * a.ldh, 0.f); a.len = len; a.t = 0; a.N = n; a.H = h; a.nblk_c = ncb;
      a.np_m_out = dal((size_t)n * ldm, 0.f);
      cls.push_back(Cls{name, a.gates});
      base += job_blocks(ncb, n, 32);
    };
    mk(N, HD, HD, "folded D layer 1 (K=512)"); mk(N, HD, PD, "folded D layer 0 (K=296)");
    const int kb = (HD + HD + 15) / 16;
    for (int i = 0; i < 20; ++i) launch_fwd_gates(gj, base, kb, s);
    CK(hipStreamSynchronize(s));
    const float us = time_graph(s, 50, 4, [&] { launch_fwd_gates(gj, base, kb, s); });
    clear_trace(s); launch_fwd_gates(gj, base, kb, s); CK(hipStreamSynchronize(s));
    const char* names[] = {"lookup", "issueW", "issueDMA", "issueEpi", "loads+bar", "mfma", "bar", "zs+bar", "epilogue"};
    const int idx[] = {2, 10, 11, 3, 4, 5, 6, 7, 9};
    report("k_fwd_gates (folded discriminator alone)", TRACE_BLOCKS, cls, names, idx, 9, us);
  }
  // ------------------------------------------------------------------ forward projection: 3 G layers + 4 D jobs + output FC
  {
    FwdProjJobs pj{};
    int base = 0;
    std::vector<Cls> cls;
    auto mk = [&](int n, int h, int p, const char* name) {
      FwdProjJob& a = pj.j[pj.n++];
      a = FwdProjJob{};
      const int ldh = (h + 3) & ~3, ldp = (p + 3) & ~3;
      a.h = dal((size_t)n * ldh, 0.5f); a.WpT = dal((size_t)p * ldh, 0.05f); a.WpT_sw = tiles((p + 15) / 16, (ldh + 15) / 16);
      a.m_prev = dal((size_t)n * ldp, 0.1f); a.m_out = dal((size_t)n * ldp, 0.f); a.out = dal((size_t)n * ldp, 0.f);
      a.len = len; a.ldh = ldh; a.ldm = ldp; a.ldo = ldp; a.P = p; a.t = 0; a.N = n; a.nblk_c = (p + 15) / 16;
      cls.push_back(Cls{name, a.h});
      base += job_blocks(a.nblk_c, n, 32);
    };
    mk(N, H, P, "G layer (760->280)"); mk(N, H, P, "G layer (760->280)"); mk(N, H, P, "G layer (760->280)");
    for (int i = 0; i < 4; ++i) mk(N, HD, PD, "D job (256->40)");
    const int kb = (H + 15) / 16;
    for (int i = 0; i < 20; ++i) launch_fwd_proj(pj, base, kb, s);
    CK(hipStreamSynchronize(s));
    const float us = time_graph(s, 50, 4, [&] { launch_fwd_proj(pj, base, kb, s); });
    clear_trace(s); launch_fwd_proj(pj, base, kb, s); CK(hipStreamSynchronize(s));
    const char* names[] = {"lookup+setup", "loads+mfma", "zs+bar", "epilogue"};
    const int idx[] = {2, 5, 7, 9};
    report("k_fwd_proj (3 G layers + 4 D jobs)", TRACE_BLOCKS, cls, names, idx, 4, us);
  }
  // ------------------------------------------------------------------ backward A: 3 G layers + 2 D jobs
  {
    BwdAJobs aj{};
    int base = 0;
    std::vector<Cls> cls;
    auto mk = [&](int n, int h, int p, const char* name) {
      BwdAJob& a = aj.j[aj.n++];
      a = BwdAJob{};
      const int ldp = (p + 3) & ~3;
      a.dout = dal((size_t)n * ldp, 0.1f); a.dmst = dal((size_t)n * ldp, 0.1f); a.Wp = dal((size_t)h * ldp, 0.05f);
      a.Wp_sw = tiles((h + 15) / 16, (ldp + 15) / 16);
      a.dmt = dal((size_t)n * ldp, 0.f); a.gates = dal((size_t)n * 4 * h, 0.3f);
      a.c_prev = dal((size_t)n * h, 0.5f); a.c_cur = dal((size_t)n * h, 0.5f);
      a.wf = dal(h, 0.1f); a.wi = dal(h, 0.1f); a.wo = dal(h, 0.1f); a.dc = dal((size_t)n * h, 0.1f);
      a.len = len; a.ldm = ldp; a.P = p; a.t = 0; a.N = n; a.H = h; a.nblk_c = (h + bwd_a_cells() - 1) / bwd_a_cells();
      cls.push_back(Cls{name, a.gates});
      base += job_blocks(a.nblk_c, n, 32);
    };
    mk(N, H, P, "G layer (280->760)"); mk(N, H, P, "G layer (280->760)"); mk(N, H, P, "G layer (280->760)");
    mk(N, HD, PD, "D job (40->256)"); mk(N, HD, PD, "D job (40->256)");
    for (int i = 0; i < 20; ++i) launch_bwd_a(aj, base, (P + 15) / 16, s);
    CK(hipStreamSynchronize(s));
    const float us = time_graph(s, 50, 4, [&] { launch_bwd_a(aj, base, (P + 15) / 16, s); });
    clear_trace(s); launch_bwd_a(aj, base, (P + 15) / 16, s); CK(hipStreamSynchronize(s));
    if (bwd_a_cells() == 32) {
      const char* names[] = {"lookup", "dm loads issued", "stage+issue W,epi", "bar", "mfma", "epilogue"};
      const int idx[] = {2, 10, 3, 4, 5, 9};
      report("k_bwd_a2 (3 G layers + 2 D jobs)", TRACE_BLOCKS, cls, names, idx, 6, us);
    } else {
      const char* names[] = {"lookup", "issueOps", "issueEpi", "loads land+mfma", "zs+bar", "epilogue"};
      const int idx[] = {2, 10, 3, 5, 7, 9};
      report("k_bwd_a (3 G layers + 2 D jobs)", TRACE_BLOCKS, cls, names, idx, 6, us);
    }
  }
  // ------------------------------------------------------------------ backward B split-K: 3 G layers (layer 0: recurrent rows only)
  {
    BwdBJobs bj{};
    std::vector<Cls> cls;
    auto mk = [&](int n, int h, int i_, int p, bool with_dx) {
      BwdBJob& b = bj.j[bj.n++];
      b = BwdBJob{};
      const int H4 = 4 * h, ldi = (i_ + 3) & ~3, ldp = (p + 3) & ~3;
      b.dz = dal((size_t)n * H4, 0.1f); b.K = dal((size_t)(i_ + p) * H4, 0.05f);
      b.I = i_; b.n_begin = with_dx ? 0 : i_; b.n_end = i_ + p;
      b.Ksw = tiles((b.n_end - b.n_begin + 15) / 16, (H4 + 15) / 16);
      b.dx = with_dx ? dal((size_t)n * ldi, 0.f) : nullptr; b.dmst = dal((size_t)n * ldp, 0.f); b.len = len;
      b.lddx = ldi; b.ldm = ldp; b.t = 0; b.N = n; b.H4 = H4; b.nblk_c = (b.n_end - b.n_begin + 15) / 16;
    };
    mk(N, H, P, P, true); mk(N, H, P, P, true); mk(N, H, P, P, false);
    float* ws = dal((size_t)8 * 3 * N * 560 + 1024, 0.f);
    bwd_b_plan(bj, ws);
    int bp = 8;
    for (int i = 0; i < bj.n; ++i) {
      const BwdBJob& b = bj.j[i];
      cls.push_back(Cls{i < 2 ? "G layer (560 cols)" : "G layer 0 (280 cols)", b.dz});
      bp = std::max(bp, 8 * b.plp.se);
      printf("bwd_bp plan job %d: KG %d kpg %d ncg %d nrg %d  place x0 %d nx %d rounds %d..%d nb %d\n", i, b.KG, b.kpg, b.ncg, b.nrg, b.plp.x0, b.plp.nx, b.plp.sb, b.plp.se, b.plp.nb);
    }
    for (int i = 0; i < 20; ++i) launch_bwd_b_splitk(bj, s);
    CK(hipStreamSynchronize(s));
    const float us = time_graph(s, 50, 4, [&] { launch_bwd_b_splitk(bj, s); });
    launch_bwd_b_splitk(bj, s); CK(hipStreamSynchronize(s));
    // the trace buffer now holds the reduce launch's blocks on top: trace the split-K kernel alone
    clear_trace(s);
    hipLaunchKernelGGL(k_bwd_bp, dim3(bp), dim3(512), std::max((size_t)64 * bp_sa4(bj.j[0].kpg) * 16 + 8192, (size_t)84 * 1024), s, bj);
    CK(hipStreamSynchronize(s));
    const char* names[] = {"lookup+issue", "loads+bar", "mfma", "bar", "zs+partials"};
    const int idx[] = {3, 4, 5, 6, 9};
    report("k_bwd_bp (+k_bwd_b_red in the timed pair)", TRACE_BLOCKS, cls, names, idx, 5, us);
  }
  for (void* p : g_bufs) (void)hipFree(p);
  return 0;
}
