// dlstm.hip -- persistent, register-resident LSTMP recurrence for SMALL cells (the discriminator, models/discriminator_lstm.py:70-91:
// 2 x LSTMCell(256, use_peepholes, num_proj=40) under tf.nn.dynamic_rnn; cell math as models/BNLSTMCell.py:176-217 minus BN).
//
// Why: when the discriminator runs alone (G-step: D(G(x)) with the freshly updated D; D-step: D's BPTT) a time step is two (three)
// dependent kernel launches of ~5 us each that do ~10 MFLOP -- pure launch/latency chains, 2.2 ms of a 10.2 ms step
// (profiles/r1_final_*).  The rows of a batch are independent, and the discriminator's weights are tiny (328 KB of gate kernel +
// 40 KB of projection per layer), so ONE workgroup per (16-row tile, layer) can keep its layer's weights in VGPRs for the WHOLE
// sequence (16 waves x 64 lanes x ~100 VGPRs), its cell state c in VGPRs and its m state in LDS, and walk all T steps without
// talking to any other workgroup of the same layer.  Layer l+1 of the same row tile runs one step behind layer l in another
// workgroup; the only hand-off is layer l's masked output row block (16 x P floats) through global memory with write-through
// (sc1) stores, a drained flag store and sc1 loads on the consumer (cdna_hip_programming.md Guideline 16 R1; checked word by
// word in tools/ubench t3).  No grid barrier, no co-residency requirement beyond <= 32 workgroups on a 256-CU chip, every spin
// bounded.  A step is MFMA-bound on its CU: 16 rows x 80 x 1024 -> 1280 v_mfma_f32_16x16x4_f32 = 4.3 us.
//
// Ownership: wave w owns cells [16w, 16w+16) with all four gates: its 4 accumulators hold i, j, f, o of (rows 4q..4q+3, cell l&15)
// in the MFMA C layout, so the whole cell update is in registers.  h goes through a wave-private LDS tile to become the A operand
// of the wave's share of the projection (K split over the waves = over the cells they own); the <= 16 partial [16 x P] tiles are
// summed in fixed wave order (deterministic), masked (dynamic_rnn: t >= len keeps the state, outputs 0) and become m_t.
#include <algorithm>

#include "kernels.h"

namespace rsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DL_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned DL_SPIN_LIMIT = 1u << 22;

// sigma(x) = 1 / (1 + 2^(-x log2 e)) and tanh(x) = 2 sigma(2x) - 1 on the hardware exp2 / rcp (1 ulp each): absolute error ~1e-7,
// i.e. fp32 round-off; a cell update is 5 of these per element and this kernel does them for a whole layer on ONE CU
__device__ __forceinline__ float dl_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float dl_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f; }

// ---------------------------------------------------------------------------------------------------------------- forward
// LDS: abuf [16][SA] = [x_t | m_{t-1}] ; hbuf [nw][16][20] ; part [nw][16][DL_PS]
constexpr int DL_HS = 20, DL_PS = 49;

// tools/ubench compiles this file with DL_PROF: wave 0 of block 0 accumulates shader cycles per phase of the step into
// g_dl_prof[]; the product build has no such code
#ifdef DL_PROF
__device__ unsigned long long g_dl_prof[16];
#define DL_STAMP(i) do { if (blockIdx.x == DL_PROF_BLOCK && w == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_[i] += t_ - last_; last_ = t_; } } while (0)
#else
#define DL_STAMP(i) do { } while (0)
#endif

// CG cell groups (16 cells each) per wave: CG = 2 -> 8 waves cover H <= 256 with 2 waves per SIMD (<= 256 VGPRs each: the
// 4 x CG x KBMAX weight fragments alone are 160 at K = 80)
template <int KBMAX, int CG>
__global__ __launch_bounds__(512) void k_dl_fwd(const DlFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nrt = (a.N + 15) / 16;
  const int l = blockIdx.x / nrt, rt = blockIdx.x - l * nrt;
  const DlLayer& Ly = a.layer[l];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6, ncg = nw * CG;
  const int H = Ly.H, P = Ly.P, ldI = Ly.ldI, ldP = Ly.ldP, ldH = Ly.ldH, H4 = 4 * H;
  const int K = ldI + ldP, nkb = (K + 15) >> 4, SA = K + 4;
  const int PP = (P + 15) & ~15, npt = PP >> 4;
  const int Ns = a.Ns, N = a.N, T = a.T;
  const int r0 = rt * 16;
  float* abuf = smem;
  float* hbuf = abuf + 16 * SA;                            // [ncg][16][DL_HS]
  float* part = hbuf + ncg * 16 * DL_HS;                   // [nw][16][DL_PS]
  float* wpl = part + nw * 16 * DL_PS;                     // [ncg][3][64 lanes] float4: projection fragments of every cell group
  float* cst = wpl + ncg * 3 * 64 * 4;                     // [ncg*16 cells][8]: bias i,j,f,o, w_i, w_f, w_o
  int* lenl = reinterpret_cast<int*>(cst + ncg * 16 * 8);            // [16] lengths of the tile's rows (0 beyond N)
  int* bail_s = lenl + 16;                                           // all LDS is dynamic (16-byte aligned base, Guideline 17)

  // ---- weights of this wave's cells -> registers (once per sequence)
  int cell[CG]; bool cok[CG];
  f32x4 wk[CG][4][KBMAX];
#pragma unroll
  for (int cgi = 0; cgi < CG; ++cgi) {
    const int cg = w * CG + cgi;
    cell[cgi] = 16 * cg + lr;
    cok[cgi] = cell[cgi] < H;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < KBMAX; ++kb) {
        const int k = 16 * kb + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (cok[cgi] && kb < nkb && k < K) {
          const size_t gcol = (size_t)g * H + cell[cgi];
          const float* src = k < ldI ? Ly.KxT + gcol * ldI + k : Ly.KhT + gcol * ldP + (k - ldI);
          v = *reinterpret_cast<const f32x4*>(src);
        }
        wk[cgi][g][kb] = v;
      }
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {                       // projection fragment B[k = own cell][p] -> LDS, lane-linear
      const int p = 16 * pt + lr, k = 16 * cg + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pt < npt && p < P && k < ldH) v = *reinterpret_cast<const f32x4*>(Ly.WpT + (size_t)p * ldH + k);
      *reinterpret_cast<f32x4*>(wpl + ((cg * 3 + pt) * 64 + lane) * 4) = v;
    }
    if (q == 0) {
      float* cs = cst + (cg * 16 + lr) * 8;
#pragma unroll
      for (int g = 0; g < 4; ++g) cs[g] = cok[cgi] ? Ly.bias[g * H + cell[cgi]] : 0.f;
      cs[4] = cok[cgi] ? Ly.wi[cell[cgi]] : 0.f; cs[5] = cok[cgi] ? Ly.wf[cell[cgi]] : 0.f; cs[6] = cok[cgi] ? Ly.wo[cell[cgi]] : 0.f; cs[7] = 0.f;
    }
  }
  int rlen[4];                                             // lengths of this lane's 4 rows (0 beyond N: never live, never stored)
  bool rok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int row = r0 + 4 * q + r; rok[r] = row < N; rlen[r] = rok[r] ? a.len[row] : 0; }
  float c[CG][4];
#pragma unroll
  for (int cgi = 0; cgi < CG; ++cgi)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[cgi][r] = 0.f;

  // ---- zero initial state (cell.zero_state): LDS m part, stash slot 0
  for (int e = tid; e < 16 * SA; e += blockDim.x) abuf[e] = 0.f;
  for (int e = tid; e < 16 * H; e += blockDim.x) { const int row = r0 + e / H; if (row < N) Ly.c[(size_t)row * H + e % H] = 0.f; }
  for (int e = tid; e < 16 * ldP; e += blockDim.x) { const int row = r0 + e / ldP; if (row < N) Ly.mst[(size_t)row * ldP + e % ldP] = 0.f; }
  if (tid == 0) *bail_s = 0;
  if (tid < 16) lenl[tid] = r0 + tid < N ? a.len[r0 + tid] : 0;
  __syncthreads();

  const unsigned* in_flag = l > 0 ? a.flags + (size_t)(l - 1) * nrt + rt : nullptr;
  unsigned* my_flag = a.flags + (size_t)l * nrt + rt;
  const bool publish = l + 1 < a.L;

  // x_t -> abuf[:, 0:ldI].  Layer 0 reads the caller's input (known for all t): the loads of step t+1 are issued at the top of
  // step t and land in LDS at its end.  Layer l > 0 reads layer l-1's masked output written inside this launch: it stays TWO
  // steps behind (flag >= t+2 at the top of step t), so its loads of step t+1 are issued early as well.
  const int xl4 = ldI >> 2;
  const int xe = tid, xrow = xe / xl4, xk = (xe - xrow * xl4) * 4;          // one float4 per thread (16 * ldI / 4 <= blockDim checked on the host)
  const bool xmine = xe < 16 * xl4 && r0 + xrow < N;
  auto wait_in = [&](int t_needed) {                  // layer l-1 has finished step t_needed
    if (tid == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(in_flag, DL_RLX) < (unsigned)(t_needed + 1)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > DL_SPIN_LIMIT) { __hip_atomic_store(a.err, 1u, DL_RLX); *bail_s = 1; break; }      // never expected: fail loudly below
      }
    }
    __syncthreads();
  };
  auto issue_x = [&](int t) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (xmine) {
      const float* src = Ly.in + ((size_t)t * Ns + r0 + xrow) * ldI + xk;
      if (in_flag) {                                   // sc1 loads: written in this launch by another workgroup
        v.x = __hip_atomic_load(src, DL_RLX); v.y = __hip_atomic_load(src + 1, DL_RLX);
        v.z = __hip_atomic_load(src + 2, DL_RLX); v.w = __hip_atomic_load(src + 3, DL_RLX);
      } else {
        v = *reinterpret_cast<const f32x4*>(src);
      }
    }
    return v;
  };
  auto commit_x = [&](const f32x4& v) { if (xe < 16 * xl4) *reinterpret_cast<f32x4*>(abuf + xrow * SA + xk) = v; };
  if (in_flag) wait_in(0);
  commit_x(issue_x(0));
  __syncthreads();

#ifdef DL_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < T; ++t) {
    DL_STAMP(7);
    f32x4 xnext = {0.f, 0.f, 0.f, 0.f};
    if (t + 1 < T) {
      if (in_flag) wait_in(t + 1);
      xnext = issue_x(t + 1);
    }
    DL_STAMP(0);
    // ---- gates: z = [x_t | m_{t-1}] . [Kx ; Kh]
    f32x4 acc[CG][4];
#pragma unroll
    for (int cgi = 0; cgi < CG; ++cgi)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[cgi][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KBMAX; ++kb) {
      if (kb < nkb) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(abuf + lr * SA + min(16 * kb + 4 * q, K - 4));      // k past K meets zero weights
#pragma unroll
        for (int cgi = 0; cgi < CG; ++cgi)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[cgi][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, wk[cgi][g][kb].x, acc[cgi][g], 0, 0, 0);
#pragma unroll
        for (int cgi = 0; cgi < CG; ++cgi)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[cgi][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, wk[cgi][g][kb].y, acc[cgi][g], 0, 0, 0);
#pragma unroll
        for (int cgi = 0; cgi < CG; ++cgi)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[cgi][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, wk[cgi][g][kb].z, acc[cgi][g], 0, 0, 0);
#pragma unroll
        for (int cgi = 0; cgi < CG; ++cgi)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[cgi][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, wk[cgi][g][kb].w, acc[cgi][g], 0, 0, 0);
      }
    }
    DL_STAMP(1);
    // ---- cell update in registers (gate order i, j, f, o); stash for BPTT; h -> wave-private LDS tiles
    // (measured, tools/ubench t6: this VALU block does not overlap the matrix pipe -- both waves of a SIMD are in the same phase;
    //  hipcc does not interleave it with the next group's MFMAs even inside one scheduling region with sched_group_barrier)
#pragma unroll
    for (int cgi = 0; cgi < CG; ++cgi) {
      const int cg = w * CG + cgi;
      const f32x4 cb = *reinterpret_cast<const f32x4*>(cst + (cg * 16 + lr) * 8);          // bias i, j, f, o
      const f32x4 cw = *reinterpret_cast<const f32x4*>(cst + (cg * 16 + lr) * 8 + 4);      // w_i, w_f, w_o
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 4 * q + r;
        const size_t ri = (size_t)t * Ns + row;
        const bool live = t < rlen[r];
        const float cp = c[cgi][r];
        float gi = dl_sigmoid(acc[cgi][0][r] + cb.x + cw.x * cp);
        float gf = dl_sigmoid(acc[cgi][2][r] + cb.z + a.forget_bias + cw.y * cp);
        float gj = dl_tanh(acc[cgi][1][r] + cb.y);
        const float cn = gf * cp + gi * gj;
        float go = dl_sigmoid(acc[cgi][3][r] + cb.w + cw.z * cn);
        float hh = go * dl_tanh(cn);
        c[cgi][r] = live ? cn : cp;
        gi = live ? gi : 0.f; gj = live ? gj : 0.f; gf = live ? gf : 0.f; go = live ? go : 0.f; hh = live ? hh : 0.f;
        if (rok[r] && cok[cgi]) {
          float* gp = Ly.gates + ri * H4 + cell[cgi];
          gp[0] = gi; gp[H] = gj; gp[2 * H] = gf; gp[3 * H] = go;
          Ly.c[(ri + Ns) * H + cell[cgi]] = c[cgi][r];
          Ly.h[ri * ldH + cell[cgi]] = hh;
        }
        hbuf[(cg * 16 + 4 * q + r) * DL_HS + lr] = cok[cgi] ? hh : 0.f;
      }
    }
    DL_STAMP(2);
    // ---- this wave's share of the projection (K = its cells); the tiles are wave-private: no barrier
    f32x4 ap[3];
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) ap[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cgi = 0; cgi < CG; ++cgi) {
      const f32x4 hf = *reinterpret_cast<const f32x4*>(hbuf + ((w * CG + cgi) * 16 + lr) * DL_HS + 4 * q);
#pragma unroll
      for (int pt = 0; pt < 3; ++pt) {
        if (pt < npt) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wpl + (((w * CG + cgi) * 3 + pt) * 64 + lane) * 4);
          ap[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hf.x, wv.x, ap[pt], 0, 0, 0);
          ap[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hf.y, wv.y, ap[pt], 0, 0, 0);
          ap[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hf.z, wv.z, ap[pt], 0, 0, 0);
          ap[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hf.w, wv.w, ap[pt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
      if (pt < npt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(w * 16 + 4 * q + r) * DL_PS + 16 * pt + lr] = ap[pt][r];
    DL_STAMP(3);
    __syncthreads();                                        // partials complete; everyone is done reading abuf
    DL_STAMP(4);
    // ---- m_t = sum of the partials in wave order, dynamic_rnn masking; next step's x
    for (int e = tid; e < 16 * PP; e += blockDim.x) {
      const int row = e / PP, p = e - row * PP;
      if (p >= P) continue;
      float v = 0.f;
      for (int ww = 0; ww < nw; ++ww) v += part[(ww * 16 + row) * DL_PS + p];
      const int grow = r0 + row;
      if (grow >= N) continue;
      const bool live = t < lenl[row];
      const float mprev = abuf[row * SA + ldI + p];
      const float mo = live ? v : mprev;
      abuf[row * SA + ldI + p] = mo;
      const size_t ri = (size_t)t * Ns + grow;
      Ly.mst[(ri + Ns) * ldP + p] = mo;
      const float o = live ? v : 0.f;
      if (publish) __hip_atomic_store(Ly.out + ri * ldP + p, o, DL_RLX);      // write-through: read by layer l+1 in this launch
      else Ly.out[ri * ldP + p] = o;
    }
    if (t + 1 < T) commit_x(xnext);
    DL_STAMP(5);
    if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its sc1 stores
    __syncthreads();
    if (publish && tid == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), DL_RLX);
    DL_STAMP(6);
    if (*bail_s) {                                          // a hand-off timed out: poison the outputs so that every loss becomes NaN
      for (int e = tid; e < 16 * ldP; e += blockDim.x)
        if (r0 + e / ldP < N) Ly.out[((size_t)t * Ns + r0 + e / ldP) * ldP + e % ldP] = __builtin_nanf("");
      break;
    }
  }
#ifdef DL_PROF
  if (blockIdx.x == DL_PROF_BLOCK && tid == 0) for (int i = 0; i < 8; ++i) g_dl_prof[i] = prof_[i];
#endif
}

size_t dl_fwd_lds_bytes(int K, int nw, int cg) {
  return ((size_t)16 * (K + 4) + (size_t)nw * cg * 16 * DL_HS + (size_t)nw * 16 * DL_PS + (size_t)nw * cg * 3 * 64 * 4 + (size_t)nw * cg * 16 * 8 + 4) * sizeof(float);
}

bool dl_fwd_supported(const DlFwdArgs& a) {
  if (a.L < 1 || a.L > DL_MAXL || (a.N + 15) / 16 * a.L > 64) return false;
  for (int l = 0; l < a.L; ++l) {
    const DlLayer& y = a.layer[l];
    if (y.H > 256 || y.P > 48 || y.ldI + y.ldP > 80 || y.ldI * 4 > 64 * ((y.H + 31) / 32) || y.ldI * 4 > 64 * ((y.H + 15) / 16) || (y.ldI & 3) || (y.ldP & 3) || (y.ldH & 3) || !y.WpT) return false;
  }
  return true;
}

void launch_dl_fwd(const DlFwdArgs& a, hipStream_t s) {
  int hmax = 1, kmax = 16;
  for (int l = 0; l < a.L; ++l) { hmax = std::max(hmax, a.layer[l].H); kmax = std::max(kmax, a.layer[l].ldI + a.layer[l].ldP); }
  const int nrt = (a.N + 15) / 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dl_fwd<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dl_fwd<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  (void)hipMemsetAsync(a.flags, 0, (size_t)a.L * nrt * sizeof(unsigned), s);
  if (kmax <= 32 && hmax <= 128) {              // small cells (unit tests): one cell group per wave
    const int nw = (hmax + 15) / 16;
    hipLaunchKernelGGL((k_dl_fwd<2, 1>), dim3(nrt * a.L), dim3(64 * nw), dl_fwd_lds_bytes(kmax, nw, 1), s, a);
  } else {
    const int nw = (hmax + 31) / 32;
    hipLaunchKernelGGL((k_dl_fwd<5, 2>), dim3(nrt * a.L), dim3(64 * nw), dl_fwd_lds_bytes(kmax, nw, 2), s, a);
  }
}

}  // namespace rsr
