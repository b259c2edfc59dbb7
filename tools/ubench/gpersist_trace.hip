// gpersist_trace.hip -- stand-alone phase timeline of the persistent generator recurrence (csrc/gpersist.hip, compiled here with
// GP_TRACE); not part of the product library.  Synthetic weights at the reference's sizes (3 x LSTMCell(760, num_proj=280), N rows,
// T steps); prints the launch time per step and the mean duration of every phase of a step per layer and wave role.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gpersist_trace.hip -o gpersist_trace      Run: ./gpersist_trace [N] [T] [layers] [b = the backward kernel] [x = with layer 0's input gradient inside it]
// -DGP_COUNT: also count the first full reads of the sweeps that fail (atomics: the timeline of such a build is not representative);
// -DGP_ABL=mask: timing ablations (1: no x sweeps, 2: hop-1 sentinels only, 4: hop-2 sentinels only)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
namespace rsr { long long g_chain_launches = 0; int dpersist_trail_grid(int nl, int N) { return nl * (N / 32) * 4 + N / 16; } }      // (dpersist.hip is not linked here)
#ifndef GP_NOTRACE                 // -DGP_NOTRACE: the product's code (no stamps: the launch time only; the stamps cost the backward kernel 18 spilled registers)
#define GP_TRACE 1
#endif
#include "../../rsrgan_amd/csrc/gpersist.hip"
using namespace rsr;

static float* dal(size_t n, float v) {
  float* p; CK(hipMalloc(&p, n * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = v * ((float)((i * 2654435761u >> 20) & 255) / 128.f - 1.f);
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 100, nl = argc > 3 ? atoi(argv[3]) : 3, H = 760, P = 280;
  const bool bwd = argc > 4 && argv[4][0] == 'b';                   // k_glstm_bwd instead (same shapes, synthetic stash)
  GPersistArgs a{};
  a.nl = nl; a.N = N; a.T = T; a.H = H; a.forget_bias = 1.f;
  std::vector<int> len(N, T);
  int* dlen; CK(hipMalloc(&dlen, N * 4)); CK(hipMemcpy(dlen, len.data(), N * 4, hipMemcpyHostToDevice));
  a.len = dlen;
  for (int l = 0; l < nl; ++l) {
    GPersistLayer& L = a.L[l];
    L.I = P; L.P = P; L.ldI = P; L.ldP = P; L.ldH = H;
    L.KxT = dal((size_t)4 * H * P, 0.03f); L.KhT = dal((size_t)4 * H * P, 0.03f); L.bias = dal(4 * H, 0.1f); L.wi = dal(H, 0.1f); L.wf = dal(H, 0.1f); L.wo = dal(H, 0.1f);
    L.Wp = dal((size_t)H * P, 0.03f);
    L.gates = dal((size_t)T * N * 4 * H, 0.5f); L.c = dal((size_t)(T + 1) * N * H, 0.f); L.h = dal((size_t)T * N * H, 0.f);
    L.mst = dal((size_t)(T + 1) * N * P, 0.f); L.out = dal((size_t)T * N * P, 0.f);
    if (l == 0) L.in = dal((size_t)T * N * P, 0.3f);
    if (bwd) L.dmt = dal((size_t)T * N * P, 0.f);
  }
  if (bwd) { a.dout_top = dal((size_t)T * N * P, 0.01f); a.ld_dout = P; }
  if (!gpersist_plan(a)) { printf("unsupported shape\n"); return 1; }
  if (const char* e = getenv("GP_TAGS")) a.tags = atoi(e);
  if (const char* e = getenv("GP_SCHED")) a.sched = atoi(e);
  if (const char* e = getenv("GP_NRT")) {                           // live row tiles per group (GPersistArgs::nrt): the rows of the dead tile get length 0
    a.nrt = atoi(e);
    if (a.nrt == 1) { for (int i = 0; i < N; ++i) if ((i & 31) >= 16) len[i] = 0; CK(hipMemcpy(dlen, len.data(), N * 4, hipMemcpyHostToDevice)); }
  }
  const size_t g1 = gpersist_gran1_bytes(a), g2 = gpersist_gran2_bytes(a);
  CK(hipMalloc(&a.gran1, g1)); CK(hipMalloc(&a.gran2, g2)); CK(hipMalloc(&a.ctl, 64));
  if (bwd) {
    const size_t g3 = gpersist_gran3_bytes(a);
    CK(hipMalloc(&a.gran3, g3));
    if (argc > 5 && argv[5][0] == 'x') { a.din0 = dal((size_t)T * N * P, 0.f); a.ld_din0 = P; }      // layer 0's input gradient inside the launch (the product: RSRGAN_GP_DIN0=1)
  }
  gpersist_arm(a, 0); CK(hipDeviceSynchronize());
  { const unsigned c0[4] = {1u, 0u, 0u, 0u}; CK(hipMemcpy(a.ctl, c0, 16, hipMemcpyHostToDevice)); }
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
#ifdef GP_NOREARM
    gpersist_arm(a, s);
#endif
    CK(hipEventRecord(e0, s));
    if (bwd) launch_glstm_bwd(a, s); else launch_glstm_fwd(a, s);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  if (const char* ls = getenv("GP_LOOP_SECONDS")) {                 // keep launching for that long (memory-controller activity is sampled from outside: tools/hbm_phase.sh)
    const double secs = atof(ls);
    hipEvent_t l0, l1; CK(hipEventCreate(&l0)); CK(hipEventCreate(&l1));
    CK(hipEventRecord(l0, s));
    long n = 0; float ms = 0.f;
    do {
      for (int i = 0; i < 20; ++i) { if (bwd) launch_glstm_bwd(a, s); else launch_glstm_fwd(a, s); }
      n += 20;
      CK(hipEventRecord(l1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, l0, l1));
    } while (ms < secs * 1e3);
    printf("loop: %ld launches in %.1f ms = %.1f us per launch\n", n, ms, ms * 1e3 / n);
  }
  unsigned ctl[4]; CK(hipMemcpy(ctl, a.ctl, 16, hipMemcpyDeviceToHost));
  printf(bwd ? "k_glstm_bwd N=%d T=%d layers=%d (NT=%d NC=%d, %zu + %zu MB of granule slots): %.1f us per launch = %.2f us per step (err word %u, generation %u)\n" : "k_glstm_fwd N=%d T=%d layers=%d (NT=%d NC=%d, %zu + %zu MB of granule slots): %.1f us per launch = %.2f us per step (err word %u, generation %u)\n",
         N, T, nl, a.NT, a.NC, g1 >> 20, g2 >> 20, best * 1e3f, best * 1e3f / T, ctl[2], ctl[0]);
#ifdef GP_COUNT
  { unsigned cn[8]; CK(hipMemcpyFromSymbol(cn, HIP_SYMBOL(rsr::g_gp_cnt), sizeof(cn)));
    printf("first full reads (5 launches): through the caches %u, of them failed %u; write-through %u, failed %u\n", cn[0], cn[1], cn[2], cn[3]); }
#endif
#ifdef GP_NOTRACE
  return 0;
#else
  static unsigned tr[256][24][24];
  CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(rsr::g_gp_trace), sizeof(tr)));
  if (const char* dump = getenv("GP_DUMP")) {                      // raw stamps [block][step][stamp] for an offline look at the skew between workgroups
    FILE* f = fopen(dump, "wb");
    if (f) { fwrite(tr, 1, sizeof(tr), f); fclose(f); }
  }
  // R wave 0, tile r (stamps 6 r + ..): 0 top, 1 x-part there, 2 m(t-1) there, 3 recurrent MFMAs + tiles written, 4 all partials there, 5 cell done
  // G wave 0 (tile 0): 12 top, 13 all cells done, 14 projection + publish issued, 15 hop 1 swept, 16 half chunk published, 17 hop 2 swept
  // X wave 0: 18 / 20 sweep start (tile 0 / 1), 19 / 21 sweep done; 22, 23: prologue
  const char* rnf[5] = {"wait: x-part of the step (X waves)", "wait: m(t-1) gathered (hop 2)", "recurrent MFMAs + tiles -> LDS",
                       "wait: the other R waves' partials (+ stash gone)", "cell"};
  // backward: R wave 0: 0 top, 1 dm(t) there, 2 dh MFMAs + partials written, 3 all partials there (and the X waves have taken dz(t+1)), 4 cell done,
  // 5 state-gradient product + publish issued (then: dz to the stash, prefetch of step t-1); G wave 0: 12 top, 13 input-gradient partials
  // summed, 14 state-gradient partials summed, 15 half chunk published, 16 hop 2 swept, 17 dm in LDS; X wave 0: 18 / 20 top, 19 / 21 dz there
  const char* rnb[5] = {"wait: dm(t) gathered (hop 2)", "dh = dm . W_p^T MFMAs + partials -> LDS", "wait: the other R waves' partials (+ X waves took dz)",
                        "cell gradient", "state-gradient product + publish (after all cells)"};
  const char** rn = bwd ? rnb : rnf;
  const int ngr = N / 32, xpg = 8 / ngr, nblk = 8 * ((nl * a.NC + xpg - 1) / xpg);
  for (int l = 0; l < nl; ++l) {
    double ph[2][5] = {{0}}, per = 0, gx[6] = {0}, xs[2] = {0}, pro = 0, lag = 0; long cnt = 0, nb_ = 0;
    for (int b = 0; b < nblk && b < 256; ++b) {
      const int xcd = b & 7, slot = b >> 3, idx = slot * xpg + (xcd % xpg);
      if (idx >= nl * a.NC || idx / a.NC != l || idx % a.NC >= 2 * ((P + 15) / 16)) continue;     // (reducers only: the others skip stamps 15, 16)
      pro += (double)(unsigned)(tr[b][0][22] - tr[b][0][23]); ++nb_;
      for (int t = 10; t < T - 1 && t < 23; ++t) {
        for (int r = 0; r < 2; ++r)
          for (int i = 0; i < 5; ++i) ph[r][i] += (double)(unsigned)(tr[b][t][6 * r + i + 1] - tr[b][t][6 * r + i]);
        per += (double)(unsigned)(tr[b][t + 1][0] - tr[b][t][0]);
        lag += (double)(unsigned)(tr[b][t][6] - tr[b][t][0]);
        gx[0] += (double)(unsigned)(tr[b][t][13] - tr[b][t][12]); gx[1] += (double)(unsigned)(tr[b][t][14] - tr[b][t][13]);
        gx[2] += (double)(unsigned)(tr[b][t][15] - tr[b][t][14]); gx[3] += (double)(unsigned)(tr[b][t][16] - tr[b][t][15]);
        gx[4] += (double)(unsigned)(tr[b][t][17] - tr[b][t][16]); gx[5] += (double)(unsigned)(tr[b][t + 1][2] - tr[b][t][5]);
        xs[0] += (double)(unsigned)(tr[b][t][19] - tr[b][t][18]); xs[1] += (double)(unsigned)(tr[b][t][21] - tr[b][t][20]);
        ++cnt;
      }
    }
    if (!cnt) continue;
    printf("layer %d, shader-clock cycles (s_memtime), mean over reducer workgroups and steps 10..%d: period %.0f (tile 1 enters %.0f behind tile 0)\n", l, T - 2 < 22 ? T - 2 : 22, per / cnt, lag / cnt);
    for (int i = 0; i < 5; ++i) printf("   R0 %-52s tile 0 %6.0f   tile 1 %6.0f\n", rn[i], ph[0][i] / cnt, ph[1][i] / cnt);
    if (bwd) {
      printf("   G0 (tile 0) input-gradient partials (layer above) %.0f | state-gradient partials (hop 1) %.0f | reduce + publish half chunk %.0f | hop-2 sweep %.0f | dm -> LDS %.0f\n",
             gx[0] / cnt, gx[1] / cnt, gx[2] / cnt, gx[3] / cnt, gx[4] / cnt);
      if (l > 0) printf("   X0 back-pressure poll + wait for dz, tile 0 / 1: %.0f / %.0f\n", xs[0] / cnt, xs[1] / cnt);
    } else {
    printf("   G0 (tile 0) wait for the cells %.0f | projection + publish (issue) %.0f | hop-1 sweep %.0f | reduce + publish half chunk %.0f | hop-2 sweep %.0f\n",
           gx[0] / cnt, gx[1] / cnt, gx[2] / cnt, gx[3] / cnt, gx[4] / cnt);
    printf("   tile 0: cell(t) done on R0 -> m(t) in LDS (the whole hand-off): %.0f\n", gx[5] / cnt);
    if (l > 0) printf("   X0 sweep of x(t), tile 0 / 1: %.0f / %.0f\n", xs[0] / cnt, xs[1] / cnt);
    }
    printf("   prologue (kernel entry -> R0 enters step 0): %.0f cycles\n", pro / nb_);
  }
  return 0;
#endif
}
