// dpersist.hip -- the discriminator's forward recurrence as ONE persistent launch (gfx950).
//
// models/discriminator_lstm.py:70-91 runs tf.nn.dynamic_rnn over a stack of LSTMCell(256, use_peepholes, num_proj=40).  When the
// discriminator runs alone (the G-run's D(G(x)) with the freshly updated D) a time step is far too small for a launch: 64 rows x
// 1024 gate columns x K = 40 (+40).  Here the whole recurrence of every layer is one launch of nl x (N/16) x NQ workgroups:
//   workgroup (layer l, 16-row tile r, cell quarter c) keeps its slices of K_h, K_x and W_p in VGPRs for all T steps; per step it
//   computes z = zx + m_{t-1}.K_h (+ x_t.K_x), the cell, and its PARTIAL projection h[:, 64 cells].W_p[64 cells, :] (16 x P floats),
//   which it publishes as 8-byte {tag = t+1, value} granules ("the data is the flag", cdna_hip_programming.md guideline 16 R2).
//   The NQ workgroups of a row tile sum each other's partials in a fixed order (deterministic) to get m_t; the tile of the layer
//   above sums the same granules to get its x_t.  Batch rows never interact, so a tile's cluster is NQ workgroups, placed on one XCD.
// No grid barrier, no flag: a consumer wave re-reads the producer's granules until every tag matches.  The tag is the launch
// GENERATION, a device word the last workgroup to finish increments (every step has its own slot, so the tag only has to tell this
// launch from earlier ones): no memset per launch, nothing frozen into a captured graph.  Every spin is bounded (wall-clock) and
// reports through the sticky `err` word of the control block (Model::check_persist reads and clears it); a failed launch also
// poisons the top layer's output with NaN so the step's losses cannot look healthy.
#include "kernels.h"

namespace rsr {

#include "dpersist_dev.h"

// tools/ubench/dpersist_trace.hip compiles this file with DP_TRACE: thread 0 of every workgroup stamps the 100 MHz counter at the
// phase boundaries of every step; the product build has no such code
#ifdef DP_TRACE
__device__ unsigned g_dp_trace[64][128][12];
// (stamps go to LDS and leave at the end: a global store behind the write-through granule stores would itself stall at issue)
#define DPT(i) do { if (threadIdx.x == 0 && t < 128) dp_tr[t][i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define DPG(i) do { if (threadIdx.x == 256 && t < 128) dp_tr[t][8 + (i)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define DPT_DECL __shared__ unsigned dp_tr[128][12];
#define DPT_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 64) for (int t_ = 0; t_ < 128; ++t_) for (int i_ = 0; i_ < 12; ++i_) g_dp_trace[blockIdx.x][t_][i_] = dp_tr[t_][i_]; } while (0)
#else
#define DPT(i) do { } while (0)
#define DPG(i) do { } while (0)
#define DPT_DECL
#define DPT_FLUSH() do { } while (0)
#endif

// 8 waves: 0-3 compute (wave w: all four gates of cells [16w, 16w+16) of the quarter; waves 0-2 also one 16-column tile of the
// partial projection), 4-7 gather (wave 4+j polls quarter j's granules and hands them over through LDS).  The gather waves issue
// no global stores: vmcnt retires in order, so a wave that has just published would wait for the acknowledgements of its own
// write-through stores before the first polled granule returns (measured: 1.2 us for a sweep of data that was long there).
__device__ __forceinline__ void dp_fwd_body(const DPersistArgs& a, const unsigned gen) {
  __shared__ __attribute__((aligned(16))) float part_m[DP_NQ][DP_KB][64][4];      // swept partials of m_{t-1}
  __shared__ __attribute__((aligned(16))) float part_x[2][DP_NQ][DP_KB][64][4];   // ... of x_{t+1} (layers above 0), by parity of the step
  // the step's stash staged for gather wave 3: [gates i, j, f, o | c | h][16 rows][DP_HS]; array 5 is the h tile the projection reads
  __shared__ __attribute__((aligned(16))) float stage[6][16 * DP_HS];
  __shared__ __attribute__((aligned(16))) float kx_lds[4][4][DP_KB][64][4];       // K_x fragments of the compute waves (off the recurrent path: not worth 48 VGPRs)
  __shared__ __attribute__((aligned(16))) float wp_lds[4][4][64][4];               // W_p fragments of the projecting waves
  __shared__ int dead;
  DPT_DECL
  const int RTn = a.N >> 4, ncl = a.nl * RTn;
  const int cl = blockIdx.x % ncl, cq = blockIdx.x / ncl;           // cluster (layer, row tile) -> same XCD when ncl % 8 == 0
  const int l = cl / RTn, r = cl - l * RTn;
  const DPersistLayer L = a.L[l];                                   // by value: the fields stay in SGPRs (a reference re-reads the kernarg every step)
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, H4 = 4 * H, T = a.T, P = L.P, ldP = L.ldP, I = L.I;
  const int r0 = r * 16;
  const int Ns = a.Ns ? a.Ns : a.N, rb = a.row0 + r0;      // (the rows of this launch inside a taller stash: stride Ns, first row row0; lengths are relative)
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  if (tid == 0) dead = 0;
  const size_t slot_stride_t = (size_t)DP_NQ * DP_SLOT;
  __syncthreads();

  if (w >= 4) {
    // ---------------- gather waves ----------------
    __builtin_amdgcn_s_setprio(3);                                 // the hand-off is the critical path: ahead of the compute waves' run-ahead work
    const int j = w - 4;
    const gu64* gm = (const gu64*)a.gran + ((size_t)(l * RTn + r) * T) * slot_stride_t + (size_t)j * DP_SLOT;          // quarter j of my tile
    const gu64* gx = (const gu64*)a.gran + ((size_t)(max(l - 1, 0) * RTn + r) * T) * slot_stride_t + (size_t)j * DP_SLOT;
    gu64* gout = (gu64*)a.gran + ((size_t)(l * RTn + r) * T) * slot_stride_t + (size_t)cq * DP_SLOT;
    float vm[DP_KB * 4], vx[DP_KB * 4];
    auto put = [&](float (*part)[DP_KB][64][4], const float (&v)[DP_KB * 4]) {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        *reinterpret_cast<float4*>(&part[j][kb][lane][0]) = make_float4(v[kb * 4], v[kb * 4 + 1], v[kb * 4 + 2], v[kb * 4 + 3]);
    };
    auto fail = [&]() { if (lane == 0) { dead = 1; __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } };
    if (l > 0) {                                                   // x_0 for the prologue
      if (!dp_sweep2(nullptr, gen, vm, gx, gen, vx, lane, err)) fail();
      put(part_x[0], vx);
    }
    __syncthreads();                                               // P
    if (dead) return;
    // iteration t delivers m_{t-1} (t > 0) and x_{t+1} (layers above 0) before barrier A(t); t = T: the last m for the tile leader
    const int t_end = cq == 0 ? T + 1 : T;
    for (int t = 0; t < t_end; ++t) {
      const bool wm = t > 0, wx = l > 0 && t + 1 < T;
      DPG(0);
      if (wm || wx) {
        if (!dp_sweep2(wm ? gm + (size_t)(t - 1) * slot_stride_t : nullptr, gen, vm,
                       wx ? gx + (size_t)(t + 1) * slot_stride_t : nullptr, gen, vx, lane, err)) fail();
        if (wm) put(part_m, vm);
        if (wx) put(part_x[(t + 1) & 1], vx);
      }
      DPG(1);
      __syncthreads();                                             // A(t)
      if (dead) return;
      if (t < T) {
        __syncthreads();                                           // B(t): the h tile of step t is in LDS
        DPG(2);
        // Partial projection of this quarter's 64 cells, on the gather waves: the write-through granule stores must not sit in the
        // compute waves' memory queue (whatever vector-memory instruction follows them waits ~0.9 us for their acknowledgement; here
        // that is the next poll, which could not succeed earlier anyway).  TRANSPOSED (W_p^T as the A operand, h^T as B: for the
        // 16x16x4 shapes A[row][k] and B[k][col] sit in the same lanes): lane (q, lr) ends up with m^T[p = 16j + 4q + i][row lr],
        // i = 0..3 = the four consecutive granules of the consumer's lane (q, lr), k-block j: two 16-byte stores of 32 contiguous bytes
        if (j < 3) {
          f32x4 pm = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const float4 af = *reinterpret_cast<const float4*>(&stage[5][lr * DP_HS + 16 * kb + 4 * q]);
            const float4 bf = *reinterpret_cast<const float4*>(&wp_lds[j][kb][lane][0]);
            pm = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.x, af.x, pm, 0, 0, 0);
            pm = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.y, af.y, pm, 0, 0, 0);
            pm = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.z, af.z, pm, 0, 0, 0);
            pm = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.w, af.w, pm, 0, 0, 0);
          }
          gu64* go_ = gout + (size_t)t * slot_stride_t + ((size_t)j * 64 + lane) * 4;
          dp_store2(go_, gen, pm[0], pm[1]);
          dp_store2(go_ + 2, gen, pm[2], pm[3]);
        }
        else {
          // Gather wave 3 (no projection tile) writes the step's stash from its LDS stage: whole 256-byte rows, 1 KB per instruction.
          // From the compute waves the same bytes are 64-byte pieces of 16 rows per instruction and cost them 1000-2200 cycles of
          // issue per step wherever they are placed (profiles/r3_dpersist_trace.txt).  (Plain stores: this wave's next poll waits
          // for their acknowledgement, which comes well before its peer's granules.)
          const int c4 = (lane & 15) * 4, rr = lane >> 4;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = 4 * rg + rr;
            const size_t grow = (size_t)t * Ns + rb + row;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(L.gates + grow * H4 + g * H + cq * 64 + c4) = *reinterpret_cast<const float4*>(&stage[g][row * DP_HS + c4]);
            *reinterpret_cast<float4*>(L.c + (grow + Ns) * H + cq * 64 + c4) = *reinterpret_cast<const float4*>(&stage[4][row * DP_HS + c4]);
            *reinterpret_cast<float4*>(L.h + grow * L.ldH + cq * 64 + c4) = *reinterpret_cast<const float4*>(&stage[5][row * DP_HS + c4]);
          }
        }
        DPG(3);
      }
    }
    return;
  }

  // ---------------- compute waves ----------------
  const int cell = cq * 64 + 16 * w + lr;
  // gate kernels: B fragment of gate g, k-block kb: lane (q, lr) holds K[k0 + 16kb + 4q + u][g*H + cell], u = 0..3 (zero beyond the width)
  float4 kh[4][DP_KB];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      float vh[4], vx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                                 // all eight loads in flight, then the selects (the asm pins the loads
        const int k = 16 * kb + 4 * q + u;                          // as unconditional: hipcc sinks a load into the branch of its select)
        vh[u] = L.K[(size_t)(I + min(k, P - 1)) * H4 + g * H + cell];
        vx[u] = L.K[(size_t)min(k, I - 1) * H4 + g * H + cell];
      }
      asm volatile("" : "+v"(vh[0]), "+v"(vh[1]), "+v"(vh[2]), "+v"(vh[3]), "+v"(vx[0]), "+v"(vx[1]), "+v"(vx[2]), "+v"(vx[3]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = 16 * kb + 4 * q + u;
        vh[u] = k < P ? vh[u] : 0.f;
        vx[u] = k < I ? vx[u] : 0.f;
      }
      kh[g][kb] = make_float4(vh[0], vh[1], vh[2], vh[3]);
      *reinterpret_cast<float4*>(&kx_lds[w][g][kb][lane][0]) = make_float4(vx[0], vx[1], vx[2], vx[3]);
    }
  // projection: wave w < 3 owns output columns [16w, 16w+16): B[k = cell 16kb + 4q + u of this quarter][col 16w + lr]
  {
    const int col = 16 * w + lr;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = L.Wp[(size_t)(cq * 64 + 16 * kb + 4 * q + u) * ldP + min(col, P - 1)];
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = col < P ? v[u] : 0.f;
      *reinterpret_cast<float4*>(&wp_lds[w][kb][lane][0]) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  // The gate products run TRANSPOSED (K^T as the A operand, m^T as B; the 16x16x4 fragments of A[row][k] and B[k][col] sit in the
  // same lanes, so the resident registers serve either way): lane (q, lr) gets z[row lr][cells cb .. cb+3] -- four consecutive cells
  // of ONE row, so zx, the gates, c and h move as 16-byte accesses (6 stores per step instead of 24: a scattered dword store costs
  // ~125 cycles of issue, 3000 per step in profiles/r3_dpersist_trace.txt) and the masks are one comparison.
  const int cb = cq * 64 + 16 * w + 4 * q;
  const float4 pwi = *reinterpret_cast<const float4*>(L.wi + cb), pwf = *reinterpret_cast<const float4*>(L.wf + cb);
  const float4 pwo = *reinterpret_cast<const float4*>(L.wo + cb);
  f32x4 bs[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bs[g] = *reinterpret_cast<const f32x4*>(L.bias + g * H + cb);
  const int lenF = a.len[r0 + lr];
  float cp[4] = {0.f, 0.f, 0.f, 0.f};
  float4 mf[DP_KB];
#pragma unroll
  for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = make_float4(0.f, 0.f, 0.f, 0.f);
  // slot 0 of the carried states is zero (cell.zero_state)
  *reinterpret_cast<float4*>(L.c + (size_t)(rb + lr) * H + cb) = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cq == 0 && w == 0) {
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb)
      if (16 * kb + 4 * q < P) *reinterpret_cast<float4*>(L.mst + (size_t)(rb + lr) * ldP + 16 * kb + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // acc init of the next step: bias + x . K_x.  Layer 0's x is the stack's input, in memory before the launch: its rows travel as
  // plain 16-byte loads issued a whole step before their product (until round 4 that product was a time-batched GEMM in front of
  // the launch, 43 us for 12800 rows, whose 52 MB of output these waves then read back).
  f32x4 accn[4];
  float4 xn[DP_KB];
  auto load_x = [&](int t) {
    const float* xr = L.in + ((size_t)t * Ns + rb + lr) * L.ldI;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) xn[kb] = *reinterpret_cast<const float4*>(xr + min(16 * kb + 4 * q, I - 4));
  };
  // sum of the NQ handed-over partials, in quarter order (k-blocks beyond the width: zero)
  auto sum_parts = [&](float (*part)[DP_KB][64][4], int width, float4 (&s)[DP_KB]) {
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const float4 p0 = *reinterpret_cast<const float4*>(&part[0][kb][lane][0]), p1 = *reinterpret_cast<const float4*>(&part[1][kb][lane][0]);
      const float4 p2 = *reinterpret_cast<const float4*>(&part[2][kb][lane][0]), p3 = *reinterpret_cast<const float4*>(&part[3][kb][lane][0]);
      s[kb] = dp_sel(16 * kb + 4 * q < width,
                     make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                                 ((p0.w + p1.w) + p2.w) + p3.w), make_float4(0.f, 0.f, 0.f, 0.f));
    }
  };
  auto mma_part = [&](f32x4 (&acc)[4], const float4 (&af)[DP_KB], const float4 (&bf)[4][DP_KB]) {     // acc[g] += (af . bf[g])^T
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][kb].x, af[kb].x, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][kb].y, af[kb].y, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][kb].z, af[kb].z, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][kb].w, af[kb].w, acc[g], 0, 0, 0);
    }
  };
  // x-part of step t of a layer above 0 from the handed-over partials of the layer below: accn = bias + mask(x_t) . K_x;
  // layer 0: from the rows load_x fetched (dynamic_rnn does not mask its inputs; the cell discards what lies past a row's length)
  auto next_x = [&](int t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) accn[g] = bs[g];
    float4 xs[DP_KB];
    if (l > 0) {
      sum_parts(part_x[t & 1], I, xs);
      const bool live = t < lenF;                                 // dynamic_rnn's output is zero past the row's length
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) xs[kb] = dp_sel(live, xs[kb], make_float4(0.f, 0.f, 0.f, 0.f));
    } else {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) xs[kb] = dp_sel(16 * kb + 4 * q < I, xn[kb], make_float4(0.f, 0.f, 0.f, 0.f));
    }
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      float4 bf[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bf[g] = *reinterpret_cast<const float4*>(&kx_lds[w][g][kb][lane][0]);
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g].x, xs[kb].x, accn[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g].y, xs[kb].y, accn[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g].z, xs[kb].z, accn[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g].w, xs[kb].w, accn[g], 0, 0, 0);
    }
  };
  // the full m of step t-1 (tile leader only): carried state -> mst[t], masked output -> out[t-1]
  auto store_m = [&](int t, const float4 (&mnew)[DP_KB], bool live_prev) {
    if (cq != 0 || w != 0) return;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb)
      if (16 * kb + 4 * q < P) {
        *reinterpret_cast<float4*>(L.mst + ((size_t)t * Ns + rb + lr) * ldP + 16 * kb + 4 * q) = mf[kb];
        *reinterpret_cast<float4*>(L.out + ((size_t)(t - 1) * Ns + rb + lr) * ldP + 16 * kb + 4 * q) =
            dp_sel(live_prev, mnew[kb], make_float4(0.f, 0.f, 0.f, 0.f));
      }
  };

  if (l == 0) load_x(0);
  __syncthreads();                                                 // P
  if (dead) return;
  next_x(0);
  if (l == 0) load_x(min(1, T - 1));

  float hv[4] = {0.f, 0.f, 0.f, 0.f}, sg[4][4] = {};
  for (int t = 0; t < T; ++t) {
    f32x4 acc[4];
    DPT(0);
    __syncthreads();                                               // A(t): m_{t-1} and x_{t+1} are in LDS
    if (dead) return;
    DPT(1);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = accn[g];
    if (t > 0) {
      float4 ms[DP_KB];
      sum_parts(part_m, P, ms);
      const bool live_prev = (t - 1) < lenF;
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = dp_sel(live_prev, ms[kb], mf[kb]);
      store_m(t, ms, live_prev);
    }
    mma_part(acc, mf, kh);
    DPT(2);
    // the cell, in the accumulator layout: lane = row lr, cells cb .. cb+3
    const bool live = t < lenF;
    const float pi_[4] = {pwi.x, pwi.y, pwi.z, pwi.w}, pf_[4] = {pwf.x, pwf.y, pwf.z, pwf.w}, po_[4] = {pwo.x, pwo.y, pwo.z, pwo.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float cpv = cp[i];
      const float gi = dp_sigmoid(acc[0][i] + pi_[i] * cpv);
      const float gf = dp_sigmoid(acc[2][i] + a.forget_bias + pf_[i] * cpv);
      const float gj = dp_tanh(acc[1][i]);
      const float cn = gf * cpv + gi * gj;
      const float go = dp_sigmoid(acc[3][i] + po_[i] * cn);
      const float hh = go * dp_tanh(cn);
      hv[i] = live ? hh : 0.f;
      sg[0][i] = live ? gi : 0.f; sg[1][i] = live ? gj : 0.f; sg[2][i] = live ? gf : 0.f; sg[3][i] = live ? go : 0.f;
      cp[i] = live ? cn : cpv;
    }
    {
      const int so = lr * DP_HS + 16 * w + 4 * q;
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(&stage[g][so]) = make_float4(sg[g][0], sg[g][1], sg[g][2], sg[g][3]);
      *reinterpret_cast<float4*>(&stage[4][so]) = make_float4(cp[0], cp[1], cp[2], cp[3]);
      *reinterpret_cast<float4*>(&stage[5][so]) = make_float4(hv[0], hv[1], hv[2], hv[3]);
    }
    DPT(3);
    __syncthreads();                                               // B(t): the h tile is in LDS
    DPT(4);
    DPT(7);
    // run ahead while the gather waves poll: the x-part of step t+1 (after their projection and publish, see the backward kernel)
    __builtin_amdgcn_s_sleep(12);
    if (t + 1 < T) next_x(t + 1);
    if (l == 0) load_x(min(t + 2, T - 1));                         // (consumed a step from now; not right behind barrier B: see the backward kernel)
    DPT(5);
    DPT(6);
  }
  DPT_FLUSH();
  if (cq == 0) {                                                   // the last step's m (block-uniform branch)
    __syncthreads();                                               // A(T)
    if (dead) return;
    float4 ms[DP_KB];
    sum_parts(part_m, P, ms);
    const bool live_prev = (T - 1) < lenF;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = dp_sel(live_prev, ms[kb], mf[kb]);
    store_m(T, ms, live_prev);
  }
}

__global__ __launch_bounds__(512, 1) void k_dlstm_fwd(const DPersistArgs a) {
  gu32* ctl = (gu32*)a.ctl;
  // every wave reads the generation itself: it only changes when ALL workgroups have passed their epilogue
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  dp_fwd_body(a, gen);
  if (threadIdx.x == 0) {                                          // (thread 0 leaves the body on every path, failures included)
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[a.nl - 1].out[0] = __builtin_nanf("");                 // (the other workgroups have finished: nobody overwrites it)
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl + DP_CTL_GEN, gen + 1u == 0u ? 1u : gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward recurrence of the same stack as ONE persistent launch (the D-run's BPTT through the discriminator, which runs alone).
// Per step t (descending), workgroup (layer l, 16-row tile r, cell quarter c), everything in the transposed lane layout of the
// forward kernel (lane (q, lr): row lr, cells cb .. cb+3):
//   dm_t  = live(t) ? dout_t + dm_state : 0            dout_t: the top layer reads it from memory (dlogits . W_fc^T, batched),
//                                                      a lower layer sums the dx partials the layer above published for step t
//   dh^T  = W_p[cells, :] . dm_t^T                     12 MFMAs, W_p rows resident as the A operand
//   the cell's gate gradients dz (kernels.hip k_bwd_a2, same formulas), dc carried in registers
//   dm_state partial^T = K_h[:, this wave's gate columns] . dz^T    48 MFMAs: the lane's own dz registers ARE the B fragment
//     (k = the wave's 16 cells of one gate = 4q + i), summed over the 4 compute waves through LDS by the gather waves, published as
//     granules, summed over the 4 quarters by the consumers: dm_state(t-1) = live(t) ? sum : dm_state (masked rows pass it through)
//   dx partial^T = K_x[...] . dz^T (layers above 0)   48 more MFMAs after barrier B, published one step later (the layer below lags)
// dz replaces the gate activations in the stash (the weight-gradient GEMMs read it), dm_t goes to dmt.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dp_bwd_body(const DPersistArgs& a, const unsigned gen) {
  __shared__ __attribute__((aligned(16))) float part_m[DP_NQ][DP_KB][64][4];      // swept partials of dm_state (from step t+1)
  __shared__ __attribute__((aligned(16))) float part_x[2][DP_NQ][DP_KB][64][4];   // ... of dout (dx of the layer above), by parity of the step
  __shared__ __attribute__((aligned(16))) float stage[4][16 * DP_HS];             // dz of the step, staged for gather wave 3
  __shared__ __attribute__((aligned(16))) float psum[4][DP_KB][64][4];            // the compute waves' partial dm_state tiles
  __shared__ __attribute__((aligned(16))) float psum_x[2][4][DP_KB][64][4];       // ... partial dx tiles, by parity of the step
  __shared__ __attribute__((aligned(16))) float kx_lds[4][DP_KB][4][64][4];       // K_x fragments (A operand of the dx product)
  __shared__ int dead;
  DPT_DECL
  const int RTn = a.N >> 4, ncl = a.nl * RTn;
  const int cl = blockIdx.x % ncl, cq = blockIdx.x / ncl;
  const int l = cl / RTn, r = cl - l * RTn;
  const DPersistLayer L = a.L[l];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, P = L.P, ldP = L.ldP, I = L.I;
  const int r0 = r * 16;
  const bool top = l == a.nl - 1;
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  if (tid == 0) dead = 0;
  const size_t slot_stride_t = (size_t)DP_NQ * DP_SLOT;
  // edge 0: dm_state partials of this layer; edge 1: dx partials of this layer = dout of the layer below
  auto edge = [&](int layer, int e) -> gu64* {
    return (gu64*)a.gran + ((size_t)((layer * 2 + e) * RTn + r) * T) * slot_stride_t;
  };
  __syncthreads();

  if (w >= 4) {
    // ---------------- gather waves ----------------
    __builtin_amdgcn_s_setprio(3);
    const int j = w - 4;
    const gu64* gm = edge(l, 0) + (size_t)j * DP_SLOT;
    const gu64* gx = edge(min(l + 1, a.nl - 1), 1) + (size_t)j * DP_SLOT;
    gu64* gout_m = edge(l, 0) + (size_t)cq * DP_SLOT;
    gu64* gout_x = edge(l, 1) + (size_t)cq * DP_SLOT;
    // progress words of this workgroup for the trailing weight-gradient workgroups: [(layer, tile)][T + 1 steps][quarter]
    gu32* dwf = a.dw_ws ? (gu32*)a.dw_flag + ((size_t)(l * RTn + r) * (T + 1)) * 4 + cq : nullptr;
    float vm[DP_KB * 4], vx[DP_KB * 4];
    auto put = [&](float (*part)[DP_KB][64][4], const float (&v)[DP_KB * 4]) {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        *reinterpret_cast<float4*>(&part[j][kb][lane][0]) = make_float4(v[kb * 4], v[kb * 4 + 1], v[kb * 4 + 2], v[kb * 4 + 3]);
    };
    auto fail = [&]() { if (lane == 0) { dead = 1; __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } };
    // the sum of the four compute waves' tile j, published as this quarter's partial
    auto publish = [&](float (*ps)[DP_KB][64][4], gu64* dst) {
      const float4 p0 = *reinterpret_cast<const float4*>(&ps[0][j][lane][0]), p1 = *reinterpret_cast<const float4*>(&ps[1][j][lane][0]);
      const float4 p2 = *reinterpret_cast<const float4*>(&ps[2][j][lane][0]), p3 = *reinterpret_cast<const float4*>(&ps[3][j][lane][0]);
      gu64* go_ = dst + ((size_t)j * 64 + lane) * 4;
      dp_store2(go_, gen, ((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y);
      dp_store2(go_ + 2, gen, ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
    };
    if (!top) {                                                    // dout of step T-1 for the prologue
      if (!dp_sweep2(nullptr, gen, vm, gx + (size_t)(T - 1) * slot_stride_t, gen, vx, lane, err)) fail();
      put(part_x[(T - 1) & 1], vx);
    }
    __syncthreads();                                               // P
    if (dead) return;
    // iteration t delivers dm_state partials of step t+1 (t < T-1) and dout partials of step t-1 (lower layers) before barrier A(t)
    for (int t = T - 1; t >= 0; --t) {
      const bool wm = t < T - 1, wx = !top && t > 0;
      DPG(0);
      if (wm || wx) {
        if (!dp_sweep2(wm ? gm + (size_t)(t + 1) * slot_stride_t : nullptr, gen, vm,
                       wx ? gx + (size_t)(t - 1) * slot_stride_t : nullptr, gen, vx, lane, err)) fail();
        if (wm) put(part_m, vm);
        if (wx) put(part_x[(t - 1) & 1], vx);
      }
      DPG(1);
      __syncthreads();                                             // A(t)
      if (dead) return;
      // dx of step t+1 (in LDS since before A(t)) leaves while the compute waves work: its write-through stores are acknowledged
      // before the dm_state publish below (two publishes back to back cost the second one, and the poll behind it, ~0.75 us)
      if (j < 3 && l > 0 && t < T - 1) publish(psum_x[(t + 1) & 1], gout_x + (size_t)(t + 1) * slot_stride_t);
      __syncthreads();                                             // B(t): psum(t) and the dz stage are in LDS
      DPG(2);
      if (j < 3) {
        publish(psum, gout_m + (size_t)t * slot_stride_t);
      } else {
        // gather wave 3 writes the step's dz over the gate activations: whole 256-byte rows from the LDS stage
        const int c4 = (lane & 15) * 4, rr = lane >> 4;
        if (dwf) {
          // (the weight-gradient workgroups trail this launch, dp_dw_body: dz leaves write-through, and the step's progress word says
          // "everything this workgroup stored up to step t + 1 (dz) / t + 2 (dm) has been acknowledged": this wave's sweep of iteration
          // t ended in s_waitcnt vmcnt(0) behind the dz stores of iteration t + 1, and the compute wave that stores dm(t + 2) has since
          // waited for loads it issued behind them.  No wait is added to the chain.)
          if (lane == 0) __hip_atomic_store(dwf + (size_t)t * 4, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = 4 * rg + rr;
            const size_t grow = (size_t)t * N + r0 + row;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              dp_store4_wt(L.gates + grow * H4 + g * H + cq * 64 + c4, *reinterpret_cast<const float4*>(&stage[g][row * DP_HS + c4]));
          }
        } else {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = 4 * rg + rr;
            const size_t grow = (size_t)t * N + r0 + row;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(L.gates + grow * H4 + g * H + cq * 64 + c4) = *reinterpret_cast<const float4*>(&stage[g][row * DP_HS + c4]);
          }
        }
      }
      DPG(3);
    }
    if (l > 0) {                                                   // dx of step 0, computed after barrier B(0)
      __syncthreads();                                             // C
      if (j < 3) publish(psum_x[0], gout_x);
    }
    return;
  }

  // ---------------- compute waves ----------------
  const int cb = cq * 64 + 16 * w + 4 * q;                          // this lane's four cells
  // A operands, resident.  dh: W_p[cell 16w + lr][p = 16kb + 4q + u]; dm_state: K[I + p][g*H + cells cb .. cb+3], p = 16pt + lr
  float4 wpA[DP_KB], khA[DP_KB][4];
  {
    const float* wrow = L.Wp + (size_t)(cq * 64 + 16 * w + lr) * ldP;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const int k = 16 * kb + 4 * q;
      const float4 v = *reinterpret_cast<const float4*>(wrow + min(k, P - 4));
      wpA[kb] = dp_sel(k < P, v, make_float4(0.f, 0.f, 0.f, 0.f));
    }
#pragma unroll
    for (int pt = 0; pt < DP_KB; ++pt) {
      const int p = 16 * pt + lr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 vh = *reinterpret_cast<const float4*>(L.K + (size_t)(I + min(p, P - 1)) * H4 + g * H + cb);
        const float4 vx = *reinterpret_cast<const float4*>(L.K + (size_t)min(p, I - 1) * H4 + g * H + cb);
        khA[pt][g] = dp_sel(p < P, vh, make_float4(0.f, 0.f, 0.f, 0.f));
        *reinterpret_cast<float4*>(&kx_lds[w][pt][g][lane][0]) = dp_sel(l > 0 && p < I, vx, make_float4(0.f, 0.f, 0.f, 0.f));
      }
    }
  }
  const float4 pwi = *reinterpret_cast<const float4*>(L.wi + cb), pwf = *reinterpret_cast<const float4*>(L.wf + cb);
  const float4 pwo = *reinterpret_cast<const float4*>(L.wo + cb);
  const float pi_[4] = {pwi.x, pwi.y, pwi.z, pwi.w}, pf_[4] = {pwf.x, pwf.y, pwf.z, pwf.w}, po_[4] = {pwo.x, pwo.y, pwo.z, pwo.w};
  const int lenF = a.len[r0 + lr];

  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  float4 mf[DP_KB];                                                 // dm_state of this lane's row: [k = 16kb + 4q + u]
#pragma unroll
  for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = make_float4(0.f, 0.f, 0.f, 0.f);
  // operands of a step, requested one step ahead: the gate activations, c_{t-1} (c_t is last step's c_{t-1}) and the top layer's dout
  f32x4 gn[4], cpn;
  float4 don[DP_KB];
  f32x4 ccur;
  auto prefetch = [&](int t) {
    const size_t row = (size_t)t * N + r0 + lr;
#pragma unroll
    for (int g = 0; g < 4; ++g) gn[g] = *reinterpret_cast<const f32x4*>(L.gates + row * H4 + g * H + cb);
    cpn = *reinterpret_cast<const f32x4*>(L.c + row * H + cb);
    if (top) {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        don[kb] = *reinterpret_cast<const float4*>(a.dout_top + row * a.ld_dout + min(16 * kb + 4 * q, P - 4));
    }
  };
  auto sum_parts = [&](float (*part)[DP_KB][64][4], int width, float4 (&s)[DP_KB]) {
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const float4 p0 = *reinterpret_cast<const float4*>(&part[0][kb][lane][0]), p1 = *reinterpret_cast<const float4*>(&part[1][kb][lane][0]);
      const float4 p2 = *reinterpret_cast<const float4*>(&part[2][kb][lane][0]), p3 = *reinterpret_cast<const float4*>(&part[3][kb][lane][0]);
      s[kb] = dp_sel(16 * kb + 4 * q < width,
                     make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                                 ((p0.w + p1.w) + p2.w) + p3.w), make_float4(0.f, 0.f, 0.f, 0.f));
    }
  };
  ccur = *reinterpret_cast<const f32x4*>(L.c + ((size_t)T * N + r0 + lr) * H + cb);      // c_T
  prefetch(T - 1);
  __syncthreads();                                                 // P
  if (dead) return;

  for (int t = T - 1; t >= 0; --t) {
    DPT(0);
    __syncthreads();                                               // A(t)
    if (dead) return;
    DPT(1);
    const bool live = t < lenF;
    // this step's operands out of the prefetch registers, the next step's requested
    f32x4 gt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gt[g] = gn[g];
    const f32x4 cprev = cpn;
    float4 dout[DP_KB];
    if (top) {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) dout[kb] = dp_sel(16 * kb + 4 * q < P, don[kb], make_float4(0.f, 0.f, 0.f, 0.f));
    } else {
      sum_parts(part_x[t & 1], P, dout);
    }
    // (the next step's operands are requested HERE, on the compute path: behind barrier B the CU's memory pipe belongs to the gather
    // waves' publish and polls -- these loads there, even delayed, cost the hand-off more than they save: 419-427 vs 406 us per launch)
    prefetch(max(t - 1, 0));
    if (t < T - 1) {
      float4 ms[DP_KB];
      sum_parts(part_m, P, ms);
      const bool live_next = (t + 1) < lenF;                        // masked rows pass the carried gradient through
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = dp_sel(live_next, ms[kb], mf[kb]);
    }
    float4 dm[DP_KB];
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb)
      dm[kb] = dp_sel(live, make_float4(dout[kb].x + mf[kb].x, dout[kb].y + mf[kb].y, dout[kb].z + mf[kb].z, dout[kb].w + mf[kb].w),
                      make_float4(0.f, 0.f, 0.f, 0.f));
    if (cq == 0 && w == 0) {                                       // dm_t for the projection's weight gradient
      if (a.dw_ws) {                                               // (read by the trailing weight-gradient workgroups: write-through)
#pragma unroll
        for (int kb = 0; kb < DP_KB; ++kb)
          if (16 * kb + 4 * q < P) dp_store4_wt(L.dmt + ((size_t)t * N + r0 + lr) * ldP + 16 * kb + 4 * q, dm[kb]);
      } else {
#pragma unroll
        for (int kb = 0; kb < DP_KB; ++kb)
          if (16 * kb + 4 * q < P) *reinterpret_cast<float4*>(L.dmt + ((size_t)t * N + r0 + lr) * ldP + 16 * kb + 4 * q) = dm[kb];
      }
    }
    // dh^T[cell][row] = W_p[cell][:] . dm^T
    f32x4 dh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].x, dm[kb].x, dh, 0, 0, 0);
      dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].y, dm[kb].y, dh, 0, 0, 0);
      dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].z, dm[kb].z, dh, 0, 0, 0);
      dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].w, dm[kb].w, dh, 0, 0, 0);
    }
    DPT(2);
    // gate / cell gradients (kernels.hip k_bwd_a2): lane = row lr, cells cb + i
    float dz[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gi = gt[0][i], gj = gt[1][i], gf = gt[2][i], go = gt[3][i];
      const float tc = dp_tanh(ccur[i]);
      const float dao = dh[i] * tc * go * (1.f - go);
      const float dcn = dc[i] + dh[i] * go * (1.f - tc * tc) + dao * po_[i];
      const float daf = dcn * cprev[i] * gf * (1.f - gf);
      const float dai = dcn * gj * gi * (1.f - gi);
      const float dj = dcn * gi * (1.f - gj * gj);
      dz[0][i] = live ? dai : 0.f; dz[1][i] = live ? dj : 0.f; dz[2][i] = live ? daf : 0.f; dz[3][i] = live ? dao : 0.f;
      dc[i] = live ? dcn * gf + dai * pi_[i] + daf * pf_[i] : dc[i];
    }
    ccur = cprev;
    {
      const int so = lr * DP_HS + 16 * w + 4 * q;
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(&stage[g][so]) = make_float4(dz[g][0], dz[g][1], dz[g][2], dz[g][3]);
    }
    DPT(3);
    // partial dm_state^T over this wave's 64 gate columns: the lane's dz[g][0..3] is the B fragment of k-block (gate g)
    f32x4 pa[DP_KB];
#pragma unroll
    for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(khA[pt][g].x, dz[g][0], pa[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(khA[pt][g].y, dz[g][1], pa[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(khA[pt][g].z, dz[g][2], pa[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(khA[pt][g].w, dz[g][3], pa[pt], 0, 0, 0);
    }
#pragma unroll
    for (int pt = 0; pt < DP_KB; ++pt) *reinterpret_cast<f32x4*>(&psum[w][pt][lane][0]) = pa[pt];
    DPT(4);
    __syncthreads();                                               // B(t)
    DPT(5);
    if (l > 0) {                                                   // dx partial^T of this step: published by the gather waves after B(t-1)
      // (first let the gather waves publish: this burst of LDS reads and MFMAs right behind the barrier held their ~350-cycle
      // publish back by 1500 cycles on the shared SIMDs, s_setprio notwithstanding -- profiles/r3_dpersist_trace.txt)
      __builtin_amdgcn_s_sleep(8);
#pragma unroll
      for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 af[DP_KB];
#pragma unroll
        for (int pt = 0; pt < DP_KB; ++pt) af[pt] = *reinterpret_cast<const float4*>(&kx_lds[w][pt][g][lane][0]);
#pragma unroll
        for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pt].x, dz[g][0], pa[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pt].y, dz[g][1], pa[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pt].z, dz[g][2], pa[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pt].w, dz[g][3], pa[pt], 0, 0, 0);
      }
#pragma unroll
      for (int pt = 0; pt < DP_KB; ++pt) *reinterpret_cast<f32x4*>(&psum_x[t & 1][w][pt][lane][0]) = pa[pt];
    }
    DPT(6);
  }
  DPT_FLUSH();
  if (l > 0) __syncthreads();                                      // C: dx of step 0 is in LDS
}

// ------------------------------------------------------------------------------------------------------------------------
// The discriminator's WEIGHT gradients inside the same launch (round 6).  The D-run's BPTT holds 64 of the 256 CUs; its weight
// gradients used to follow it as twelve launches (split-K GEMMs + reductions + column sums: 0.2 ms for 188 k parameters) in front of
// the update the G-run waits for.  Here 4 * nl * N/16 more workgroups ride the launch on idle CUs and TRAIL the recurrence by two
// steps: workgroup (layer l, 16-row tile r, gate g), wave w owns the gate's cells [32w, 32w + 32) and accumulates over all T steps,
// in registers,
//   dK[k][col]   += in[row][k] . dz[row][col]      in = [x_t | m_{t-1} | 1]: the row of ones makes the bias gradient row I + P
//                                                  (96 x 32 per wave: 12 MFMA tiles; rows = the tile's 16 batch rows = 4 MFMAs of k = 4)
//   dw_{i,f,o}[cell] += dz_gate[row][cell] . c[row][cell]   (c_{t-1} for i and f, c_t for o: BNLSTMCell.py:176-207's peepholes)
//   dWp[cell][p] += h[row][cell] . dm[row][p]       (the j gate's workgroup, which has no peephole: wave w owns cells [32w, 32w + 32))
// straight from memory in the MFMA operand layouts (no LDS, no barrier: every wave is its own stream).  dz(t) and dm(t) are the
// recurrence's write-through stores; the wave polls the four quarters' progress words of step t - 2 (see dp_bwd_body) and reads them
// with sc1 loads.  The per-tile partial sums land in dw_ws; k_dw_reduce (kernels.hip) adds the N/16 tiles in a fixed order.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int DW_MT = 6;            // 16-row tiles of the [x | m | 1] operand: three of x (I <= 48), three of m and the ones row (P <= 47)
__device__ __forceinline__ float dp_ld_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool proj>                                                // the j gate has no peephole: its workgroup takes dWp
__device__ __forceinline__ void dp_dw_body(const DPersistArgs& a, const unsigned gen, const int bid) {
  const int RTn = a.N >> 4;
  const int g = bid & 3, cl = bid >> 2;                             // gate (i, j, f, o); (layer, row tile)
  const int l = cl / RTn, r = cl - l * RTn;
  const DPersistLayer L = a.L[l];
  const int lane = threadIdx.x & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, I = L.I, P = L.P, ldP = L.ldP, ldI = L.ldI;
  const int r0 = r * 16;
  const int c0 = 32 * w, col0 = g * H + c0;                         // this wave's 32 cells / gate columns
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  const gu32* flags = (const gu32*)a.dw_flag + ((size_t)(l * RTn + r) * (T + 1)) * 4;

  f32x4 acc[DW_MT][2];
#pragma unroll
  for (int mt = 0; mt < DW_MT; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 accp[2][DP_KB];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int nt = 0; nt < DP_KB; ++nt) accp[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float pp[2] = {0.f, 0.f};
  const size_t coff = g == 3 ? (size_t)N * H : 0;                   // w_o multiplies the NEW cell state: c_t lives one step further

  // A step = one round trip of operand loads + 48 - 72 MFMAs.  The progress word of the NEXT step travels with the operands of this one
  // (polled on its own, a step was two dependent round trips, ~5 us against the recurrence's 4.2 us per step: the launch ended 87 us
  // after its recurrence); only a word that is not there yet -- the wave has caught up with the recurrence -- is polled in a loop.
  struct Ops { float av[DW_MT][4], bv[2][4], ev[DP_KB][4], fv[2][4]; };
  bool ok = true;
  auto wait_for = [&](int t) {                                      // the recurrence is two steps further (or has drained: slot T)
    const gu32* fp = flags + (size_t)(t >= 2 ? t - 2 : T) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 0;; ++spins) {
      u32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(fp) : "memory");
      if (v.x == gen && v.y == gen && v.z == gen && v.w == gen) return;
      __builtin_amdgcn_s_sleep(32);
      if ((spins & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull ||
                                 __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { ok = false; return; }
    }
  };
  // Operand addresses = a wave-uniform base per (buffer, step, row quad s) + a per-lane offset that never changes: every load is
  // unconditional (a clamped offset and a select afterwards: a load under a branch closes the branch with a full wait -- the first
  // version of this loop ran 6 us per step on forty serialised round trips) and leaves in one batch.
  // The [x | m | 1] operand as 3 + 3 tiles: tiles 0-2 = x features 16mt + lr (< I), tiles 3-5 = m features 16(mt-3) + lr (< P), the
  // row of ones at m feature P (P < 48, checked by dpersist_dw_supported).
  unsigned off_x[3], off_m[3], off_d[DP_KB];
  float msk_x[3], msk_m[3], one_m[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int f = 16 * k + lr;
    off_x[k] = (unsigned)((r0 + q) * ldI + min(f, I - 1)); msk_x[k] = f < I ? 1.f : 0.f;
    off_m[k] = (unsigned)((r0 + q) * ldP + min(f, P - 1)); msk_m[k] = f < P ? 1.f : 0.f; one_m[k] = f == P ? 1.f : 0.f;
    off_d[k] = off_m[k];
  }
  const unsigned off_b = (unsigned)((r0 + q) * H4 + col0 + lr);
  const unsigned off_c = (unsigned)((r0 + q) * (proj ? L.ldH : H) + c0 + lr);
  auto load = [&](Ops& o, int t) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const size_t rw = (size_t)t * N + 4 * s;                      // (wave-uniform: scalar arithmetic)
      const float* gb = L.gates + rw * H4;
      o.bv[0][s] = dp_ld_sc1(gb + off_b);
      o.bv[1][s] = dp_ld_sc1(gb + off_b + 16);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const size_t rw = (size_t)t * N + 4 * s;
      const float* xb = L.in + rw * ldI;
      const float* mb = L.mst + rw * ldP;
#pragma unroll
      for (int k = 0; k < 3; ++k) { o.av[k][s] = xb[off_x[k]]; o.av[3 + k][s] = mb[off_m[k]]; }
    }
    if (proj) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const size_t rw = (size_t)t * N + 4 * s;
        const float* hb = L.h + rw * L.ldH;
        const float* db = L.dmt + rw * ldP;
        o.fv[0][s] = hb[off_c]; o.fv[1][s] = hb[off_c + 16];
#pragma unroll
        for (int nt = 0; nt < DP_KB; ++nt) o.ev[nt][s] = dp_ld_sc1(db + off_d[nt]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* cb = L.c + coff + ((size_t)t * N + 4 * s) * H;
        o.fv[0][s] = cb[off_c]; o.fv[1][s] = cb[off_c + 16];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                              // (all requests out before the first select)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { o.av[k][s] *= msk_x[k]; o.av[3 + k][s] = o.av[3 + k][s] * msk_m[k] + one_m[k]; }
      if (proj) {
#pragma unroll
        for (int nt = 0; nt < DP_KB; ++nt) o.ev[nt][s] *= msk_m[nt];
      }
    }
  };
  auto compute = [&](const Ops& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int mt = 0; mt < DW_MT; ++mt) {
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[mt][s], o.bv[0][s], acc[mt][0], 0, 0, 0);
        acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[mt][s], o.bv[1][s], acc[mt][1], 0, 0, 0);
      }
    }
    if (proj) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int nt = 0; nt < DP_KB; ++nt) accp[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.fv[j][s], o.ev[nt][s], accp[j][nt], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) { pp[0] += o.bv[0][s] * o.fv[0][s]; pp[1] += o.bv[1][s] * o.fv[1][s]; }
    }
  };
  // ONE wave of the workgroup polls (32 waves spinning on a tile's progress words -- every wave of its four workgroups -- slowed the
  // recurrence itself: 423 -> 660 us per launch; the words share memory channels with its hand-offs): wave 0 hands what it has seen to
  // the other seven through an LDS word, `prog` = the lowest step whose operands are known to be complete (T: none yet, -1: give up)
  __shared__ int prog;
  if (threadIdx.x == 0) prog = T;
  __syncthreads();
  Ops o;
  if (w == 0) {
    wait_for(T - 1);
    if (lane == 0) __hip_atomic_store(&prog, ok ? T - 1 : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int t = T - 1; t >= 0 && ok; --t) {
      u32x4 fl = {gen, gen, gen, gen};
      if (t >= 1) {
        const gu32* fp = flags + (size_t)(t >= 3 ? t - 3 : T) * 4;  // the word step t - 1 waits for
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(fl) : "v"(fp) : "memory");
      }
      load(o, t);
      compute(o);
      // (the operands above were requested behind the word and have arrived: so has the word; the statement ties its first use to this point)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(fl) : : "memory");
      if (t >= 1) {
        if (!(fl.x == gen && fl.y == gen && fl.z == gen && fl.w == gen)) { __builtin_amdgcn_s_sleep(32); wait_for(t - 1); }
        if (lane == 0) __hip_atomic_store(&prog, ok ? t - 1 : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  } else {
    for (int t = T - 1; t >= 0 && ok; --t) {
      for (;;) {
        const int p_ = __hip_atomic_load(&prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p_ < 0) { ok = false; break; }
        if (p_ <= t) break;
        __builtin_amdgcn_s_sleep(4);
      }
      if (!ok) break;
      load(o, t);
      compute(o);
    }
  }
  if (!ok) return;
  // the tile's partial sums: [I + P + 1][4H] (kernel rows, then the bias row) | [3][H] peepholes w_i, w_f, w_o | [H][ldP] projection
  float* wsb = a.dw_ws + (size_t)(l * RTn + r) * a.dw_stride;
#pragma unroll
  for (int mt = 0; mt < DW_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * (mt % 3) + 4 * q + i;                     // x feature (tiles 0-2) / m feature, the row of ones at P (tiles 3-5)
        const bool on = mt < 3 ? f < I : f <= P;
        if (on) wsb[(size_t)(mt < 3 ? f : I + f) * H4 + col0 + 16 * nt + lr] = acc[mt][nt][i];
      }
  if (proj) {
    float* pw = wsb + (size_t)(I + P + 1) * H4 + 3 * H;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int nt = 0; nt < DP_KB; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (16 * nt + lr < P) pw[(size_t)(c0 + 16 * j + 4 * q + i) * ldP + 16 * nt + lr] = accp[j][nt][i];
  } else {
    float* pw = wsb + (size_t)(I + P + 1) * H4 + (g == 0 ? 0 : g == 2 ? 1 : 2) * H + c0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float v = pp[nt];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);               // the four row groups of a column, fixed order
      if (q == 0) pw[16 * nt + lr] = v;
    }
  }
}

__global__ __launch_bounds__(512, 1) void k_dlstm_bwd(const DPersistArgs a) {
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int nchain = a.nl * (a.N >> 4) * DP_NQ;
  if ((int)blockIdx.x >= nchain) {
    const int bid = (int)blockIdx.x - nchain;
    if ((bid & 3) == 1) dp_dw_body<true>(a, gen, bid); else dp_dw_body<false>(a, gen, bid);
  } else {
    dp_bwd_body(a, gen);
    if (a.dw_ws) {
      // the last progress word: every store of every wave of this workgroup has been acknowledged
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        const int RTn = a.N >> 4, ncl = a.nl * RTn, cl = blockIdx.x % ncl, cq = blockIdx.x / ncl;
        __hip_atomic_store((gu32*)a.dw_flag + ((size_t)cl * (a.T + 1) + a.T) * 4 + cq, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[0].gates[0] = __builtin_nanf("");                      // poisons layer 0's kernel gradient, hence the clipped update
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl + DP_CTL_GEN, gen + 1u == 0u ? 1u : gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// (the backward launch uses two edges per layer: dm_state partials and the dx partials for the layer below)
size_t dpersist_granule_bytes(int nl, int N, int T) { return (size_t)2 * nl * (N / 16) * T * DP_NQ * DP_SLOT * sizeof(unsigned long long); }

int dpersist_grid(int nl, int N) { return nl * (N / 16) * DP_NQ; }
size_t dpersist_lds_bytes() {
  hipFuncAttributes f{}, b{};
  if (hipFuncGetAttributes(&f, (const void*)k_dlstm_fwd) != hipSuccess || hipFuncGetAttributes(&b, (const void*)k_dlstm_bwd) != hipSuccess) return 0;
  return f.sharedSizeBytes > b.sharedSizeBytes ? f.sharedSizeBytes : b.sharedSizeBytes;
}

bool dpersist_supported(const DPersistArgs& a) {
  if (a.nl < 1 || a.nl > DP_MAXL || a.N % 16 != 0 || a.H != 64 * DP_NQ || a.T < 1) return false;
  // every workgroup must be resident at once (they wait for each other), one workgroup per CU: at most 128, and no more than the
  // device has (the static half; Model::init asks the device itself, resident_probe)
  if (dpersist_grid(a.nl, a.N) > 128 || dpersist_grid(a.nl, a.N) > device_cu_count()) return false;
  for (int l = 0; l < a.nl; ++l) {
    const DPersistLayer& L = a.L[l];
    if (L.P > 16 * DP_KB || L.P % 4 != 0 || L.ldP % 4 != 0 || L.P < 4) return false;
    if (l > 0 && (L.I != a.L[l - 1].P)) return false;
    if (L.I > 16 * DP_KB || (l == 0 && (L.I < 4 || L.I % 4 != 0 || L.ldI % 4 != 0))) return false;      // (layer 0 reads 16-byte pieces of its input rows)
  }
  return true;
}

// a.gran: zeroed ONCE at allocation (tag 0 is never a generation); a.ctl: {1, 0, 0, 0} at allocation
void launch_dlstm_bwd(const DPersistArgs& a, hipStream_t s) {
  const int blocks = dpersist_grid(a.nl, a.N) + (a.dw_ws ? dpersist_dw_grid(a.nl, a.N) : 0);
  hipLaunchKernelGGL(k_dlstm_bwd, dim3(blocks), dim3(512), 0, s, a);
  ++g_chain_launches;
}
// the trailing weight-gradient workgroups of the backward launch (dp_dw_body): their number, the floats of one (layer, tile) record of
// dw_ws, the bytes of the progress words, and whether the shapes fit
int dpersist_dw_grid(int nl, int N) { return 4 * nl * (N / 16); }
size_t dpersist_dw_stride(const DPersistArgs& a) { return (size_t)(a.L[0].I + a.L[0].P + 1) * 4 * a.H + 3 * (size_t)a.H + (size_t)a.H * a.L[0].ldP; }
size_t dpersist_dw_flag_bytes(int nl, int N, int T) { return (size_t)nl * (N / 16) * (T + 1) * 4 * sizeof(unsigned); }
bool dpersist_dw_supported(const DPersistArgs& a) {
  for (int l = 0; l < a.nl; ++l) {
    const DPersistLayer& L = a.L[l];
    if (L.I > 48 || L.P > 47 || L.I < 1 || L.I != a.L[0].I || L.P != a.L[0].P || L.ldP != a.L[0].ldP || L.ldI < L.I) return false;
  }
  return a.H == 64 * DP_NQ;
}

// the 2-tile form of the forward launch, stand-alone (RSRGAN_DFWD_T=1: the body that k_glstm_fwd_dt hosts, on the generator's register budget)
__global__ __launch_bounds__(768, 3) void k_dlstm_fwd_t(const DPersistArgs a) {
  __shared__ __attribute__((aligned(16))) DpFwdTLds S;
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) S.dead = 0;
  __syncthreads();
  dp_fwdt_body<false>(a, gen, S, (int)blockIdx.x, false);
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) a.L[a.nl - 1].out[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl + DP_CTL_GEN, gen + 1u == 0u ? 1u : gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
void launch_dlstm_fwd_t(const DPersistArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_dlstm_fwd_t, dim3(a.nl * (a.N / 32) * DP_NQ), dim3(768), 0, s, a);
  ++g_chain_launches;
}

size_t dpersist_fwdt_lds_bytes() { return sizeof(DpFwdTLds); }
int dpersist_trail_grid(int nl, int N) { return nl * (N / 32) * DP_NQ + N / 16; }
size_t dpersist_trail_lds_bytes() { return sizeof(DpTrailLds); }
bool dpersist_trail_supported(const DPersistArgs& a) {
  if (!dpersist_supported(a) || a.N % 32 != 0 || !a.dy || !a.fc_w || !a.dtop) return false;
  if (a.ld_dy % 4 != 0 || a.ld_fcw % 4 != 0 || a.ld_dtop % 4 != 0 || a.fc_P < 1 || a.fc_P > a.ld_dtop || a.ld_dtop > 16 * 8 * DP_FCT) return false;
  return a.L[0].I % 4 == 0 && a.L[0].I >= 4 && a.L[0].I <= a.ld_dy && a.L[0].I <= a.ld_fcw;
}
void launch_dlstm_fwd(const DPersistArgs& a, hipStream_t s) {
  const int blocks = dpersist_grid(a.nl, a.N);
  hipLaunchKernelGGL(k_dlstm_fwd, dim3(blocks), dim3(512), 0, s, a);
  ++g_chain_launches;
}

}  // namespace rsr
