// dpersist_trace.hip -- stand-alone phase timeline of the persistent discriminator recurrence (csrc/dpersist.hip, compiled here with
// DP_TRACE); not part of the product library.  Synthetic weights at the reference's sizes (2 x LSTMCell(256, num_proj=40), N rows,
// T steps); prints the launch time per step and the mean duration of every phase of a step per layer.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 dpersist_trace.hip -o dpersist_trace      Run: ./dpersist_trace [N] [T] [bwd]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
namespace rsr { long long g_chain_launches = 0; }
#ifndef DP_NOTRACE                 // -DDP_NOTRACE: the product's code (the launch time only)
#define DP_TRACE 1
#endif
#include "../../rsrgan_amd/csrc/dpersist.hip"
using namespace rsr;

static float* dal(size_t n, float v) {
  float* p; CK(hipMalloc(&p, n * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = v * ((float)((i * 2654435761u >> 20) & 255) / 128.f - 1.f);
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 100, H = 256, P = 40, I0 = 40;          // (layer 0 reads the discriminator input: 40-dim MFCC rows)
  const bool bwd = argc > 3;
  DPersistArgs a{};
  a.nl = 2; a.N = N; a.T = T; a.H = H; a.forget_bias = 1.f;
  std::vector<int> len(N, T);
  int* dlen; CK(hipMalloc(&dlen, N * 4)); CK(hipMemcpy(dlen, len.data(), N * 4, hipMemcpyHostToDevice));
  a.len = dlen;
  for (int l = 0; l < 2; ++l) {
    DPersistLayer& L = a.L[l];
    L.I = l == 0 ? I0 : P; L.P = P; L.ldP = P; L.ldH = H; L.ldI = L.I;
    if (l == 0) L.in = dal((size_t)T * N * I0, 0.3f);
    L.K = dal((size_t)(L.I + P) * 4 * H, 0.05f); L.bias = dal(4 * H, 0.1f); L.wi = dal(H, 0.1f); L.wf = dal(H, 0.1f); L.wo = dal(H, 0.1f);
    L.Wp = dal((size_t)H * P, 0.05f);
    L.gates = dal((size_t)T * N * 4 * H, 0.5f); L.c = dal((size_t)(T + 1) * N * H, 0.f); L.h = dal((size_t)T * N * H, 0.f);
    L.mst = dal((size_t)(T + 1) * N * P, 0.f); L.out = dal((size_t)T * N * P, 0.f); L.dmt = dal((size_t)T * N * P, 0.f);
  }
  a.dout_top = dal((size_t)T * N * P, 0.1f); a.ld_dout = P;
  const size_t gb = dpersist_granule_bytes(2, N, T);
  CK(hipMalloc(&a.gran, gb)); CK(hipMalloc(&a.ctl, 64));
  CK(hipMemset(a.gran, 0, gb));
  { const unsigned c0[4] = {1u, 0u, 0u, 0u}; CK(hipMemcpy(a.ctl, c0, 16, hipMemcpyHostToDevice)); }
  if (!dpersist_supported(a)) { printf("unsupported shape\n"); return 1; }
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, s));
    if (bwd) launch_dlstm_bwd(a, s); else launch_dlstm_fwd(a, s);      // (the product's launchers: grid and block size are theirs to choose)
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  unsigned ctl[4]; CK(hipMemcpy(ctl, a.ctl, 16, hipMemcpyDeviceToHost)); const unsigned err = ctl[2];
  printf("%s N=%d T=%d: %.1f us per launch = %.2f us per step (err word %u, generation %u)\n", bwd ? "k_dlstm_bwd" : "k_dlstm_fwd", N, T, best * 1e3f, best * 1e3f / T, err, ctl[0]);
  static unsigned tr[64][128][12];
  CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(rsr::g_dp_trace), sizeof(tr)));
  // stamps of a step: 0 top, 1 after barrier A, 2 after the gate MFMAs, 3 after the cell, 4 after barrier B, 7 after the publish,
  // 5 after the run-ahead x-part, 6 after the stash stores
  const int seq[8] = {0, 1, 2, 3, 4, 7, 5, 6};
  const char* namesf[7] = {"wait at barrier A (m(t-1) handed over)", "sum of the partials + gate MFMAs", "cell + h -> LDS", "barrier B",
                           "(projection + publish: on the gather waves)", "run ahead: x-part MFMAs of step t+1", "stash stores (gates, c, h)"};
  // backward stamps: 0 top, 1 after barrier A, 2 after dm + dh MFMAs, 3 after the cell gradients, 4 after the dm_state MFMAs, 5 after
  // barrier B, 6 after the dx MFMAs
  const char* namesb[7] = {"wait at barrier A (dm_state handed over)", "sum of the partials, dm, dh MFMAs", "cell gradients + dz -> LDS",
                           "dm_state partial MFMAs + LDS", "barrier B", "dx partial MFMAs (layers above 0)", "-"};
  const int seqb[8] = {0, 1, 2, 3, 4, 5, 6, 6};
  const char** names = bwd ? namesb : namesf;
  const int nb = 2 * (N / 16) * DP_NQ, ncl = 2 * (N / 16);
  for (int l = 0; l < 2; ++l) {
    double ph[7] = {0, 0, 0, 0, 0, 0, 0}, per = 0; long cnt = 0;
    for (int b = 0; b < nb && b < 64; ++b) {
      if ((b % ncl) / (N / 16) != l) continue;
      for (int t = 10; t < T - 1 && t < 127; ++t) {
        const int* sq = bwd ? seqb : seq;
        for (int i = 0; i < 7; ++i) ph[i] += (double)(unsigned)(tr[b][t][sq[i + 1]] - tr[b][t][sq[i]]);
        per += bwd ? (double)(unsigned)(tr[b][t][0] - tr[b][t + 1][0]) : (double)(unsigned)(tr[b][t + 1][0] - tr[b][t][0]); ++cnt;
      }
    }
    printf("layer %d, shader-clock cycles (s_memtime), mean over workgroups and steps 10..T-2: period %.0f\n", l, per / cnt);
    for (int i = 0; i < 7; ++i) printf("   %-42s %6.0f\n", names[i], ph[i] / cnt);
    // gather wave 0 (stamps 8..11: loop top, sweep done, after barrier B, after the publish)
    double gp[4] = {0, 0, 0, 0};
    for (int b = 0; b < nb && b < 64; ++b) {
      if ((b % ncl) / (N / 16) != l) continue;
      for (int t = 10; t < T - 1 && t < 127; ++t) {
        gp[0] += (double)(unsigned)(tr[b][t][9] - tr[b][t][8]);        // sweep
        gp[1] += (double)(unsigned)(tr[b][t][10] - tr[b][t][9]);       // A .. B (idle)
        gp[2] += (double)(unsigned)(tr[b][t][11] - tr[b][t][10]);      // projection + publish
        gp[3] += bwd ? (double)(unsigned)(tr[b][t][9] - tr[b][t + 1][11]) : (double)(unsigned)(tr[b][t + 1][9] - tr[b][t][11]);   // publish -> next sweep complete
      }
    }
    printf("   gather wave 0: sweep %.0f | barriers A..B %.0f | projection MFMAs + granule stores %.0f | publish(t) -> m(t) swept %.0f\n",
           gp[0] / cnt, gp[1] / cnt, gp[2] / cnt, gp[3] / cnt);
  }
  return 0;
}
