// gpersist.hip -- the GENERATOR's forward recurrence as ONE persistent launch (gfx950).
//
// models/lstm.py:89-112 runs tf.nn.dynamic_rnn over MultiRNNCell[3 x LSTMCell(760, use_peepholes, num_proj=280)] (cell arithmetic:
// models/BNLSTMCell.py:176-217 without the batch norms).  Launch-per-phase (kernels.hip k_fwd_gates + k_fwd_proj) re-reads all
// 23 MB of weights from L2 / Infinity Cache on every wavefront diagonal.  Here the weights stay on chip for all T steps:
//
//   * batch rows never interact, so the batch is cut into ROW GROUPS of 32 rows (two 16-row MFMA tiles) that run independently;
//   * workgroup (row group g, layer l, slice c) owns NT gate tiles of 4 cells (16 gate columns: i, j, f, o of 4 cells, so that one
//     lane of the 16x16x4 accumulator holds all four gates of ONE cell of ONE row) = 4*NT cells; NC = ceil(H / (4 NT)) slices;
//   * 12 waves: R0..R3 keep K_h (the recurrent rows of the kernel, split by k-block over the four waves) in VGPRs and run the
//     critical path (m(t-1).K_h, the cell, the partial projection); X0..X3 keep K_x and run one step AHEAD of the R waves (the
//     x-part of step t+1 only needs the layer below); G0..G3 poll, reduce and publish;
//   * all products run TRANSPOSED (weights = MFMA A operand, activations = B operand), as in dpersist.hip.
//
// Two hand-offs per step and layer, both "the data is the flag" (8-byte {value, tag} granules in 16-byte write-through accesses,
// cdna_hip_programming.md guideline 16 R2):
//   hop 1  every workgroup publishes its PARTIAL projection h[:, its cells] . W_p[its cells, :] (32 x P) cut into (k-block, row tile)
//          chunks of 16 x 16; workgroup c of the layer is the REDUCER of chunk c: it sums the NC partials in slice order
//          (deterministic) -- a reduce-scatter;
//   hop 2  the reducer publishes its chunk of m(t); every workgroup of the layer (for the recurrent product of step t+1) and of the
//          layer above (x of step t) gathers all chunks -- an all-gather.  The chunk's lane layout IS the consumer's B fragment.
// Inside a workgroup the three roles synchronise through monotonic LDS counters (no s_barrier: the X waves are not in lock step).
// Tags: hop 2 has one slot per step, tag = launch generation (dpersist.hip); hop 1 is a ring of two steps, tag = generation and step.
// Every spin is bounded; failures go to the sticky err word of the control block and poison the top layer's output with NaN.
#include "kernels.h"

namespace rsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int GP_NR = GP_ROWS / 16;      // row tiles per group
constexpr int GP_WAVES = 12;             // R0..R3, X0..X3, G0..G3
static_assert(GP_NR == 2, "chunk index = 2 * k-block + row tile");
constexpr int GP_NKB = 18;               // k-blocks of 16 of the recurrent / input width (P, I <= 288)
constexpr int GP_KBW = 5;                // k-blocks per R / X wave (k-block jb belongs to wave jb & 3)
constexpr int GP_SLOT = 2048;            // bytes per chunk slot: 64 lanes x 4 granules
constexpr int GP_NCH = GP_NKB * GP_NR;   // chunk slots per layer and step (the layout's stride; a layer uses its first nkb * NR)
constexpr unsigned GP_SC1 = 16u;         // aux of the raw-buffer builtins: sc1 (agent scope: write-through store / L1-bypassing load)
constexpr unsigned GP_VOL = 1u << 31;    // ... compiler-only: volatile (a polled load must not be hoisted out of its loop)

#ifdef GP_TRACE
__device__ unsigned g_gp_trace[256][24][24];
#define GPT_DECL __shared__ unsigned gp_tr[24][24];
#define GPT(i) do { if ((w & 3) == 0 && lane == 0 && t < 24) gp_tr[t][i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define GPT_FLUSH() do { if (lane == 0) for (int t_ = 0; t_ < 24; ++t_) for (int i_ = i0_; i_ < i1_; ++i_) g_gp_trace[blockIdx.x][t_][i_] = gp_tr[t_][i_]; } while (0)
#else
#define GPT_DECL
#define GPT(i) do { } while (0)
#define GPT_FLUSH() do { } while (0)
#endif

__device__ __forceinline__ float gp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float gp_tanh(float x) {          // (dpersist.hip dp_tanh: ~2e-7 absolute)
  const float x2 = x * x;
  const float ser = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.05396825f * x2)));
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x));
  return fabsf(x) < 0.1f ? ser : big;
}

// ---- intra-workgroup synchronisation: monotonic LDS counters ----
__device__ __forceinline__ void gp_signal(unsigned* cnt, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's LDS writes have landed
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool gp_wait(const unsigned* cnt, unsigned target, const unsigned* dead) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned spins = 0;; ++spins) {
    const unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (v >= target) break;
    if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return false;
    if ((spins & 1023) == 1023 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return false;   // 2 s at 100 MHz
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
  return true;
}

// ---- inter-workgroup transport ----
struct GpBuf { __amdgpu_buffer_rsrc_t rs; };
__device__ __forceinline__ GpBuf gp_buf(const void* p, size_t bytes) {
  GpBuf b; b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); return b;
}
// one chunk slot = two contiguous 1 KB halves, [64 lanes] x {v0, tag, v1, tag} and [64 lanes] x {v2, tag, v3, tag}: every store / load
// instruction moves whole 128-byte lines (with the two 16-byte pieces of a lane side by side each instruction touched 16 lines half
// full: ~700 cycles of issue per store, profiles/r4_gpersist_trace.txt)
__device__ __forceinline__ void gp_publish(const GpBuf& b, unsigned off, int lane, unsigned tag, const f32x4 v) {
  const u32x4 x0 = {__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag};
  const u32x4 x1 = {__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag};
  __builtin_amdgcn_raw_buffer_store_b128(x0, b.rs, off + (unsigned)lane * 16u, 0, GP_SC1);
  __builtin_amdgcn_raw_buffer_store_b128(x1, b.rs, off + 1024u + (unsigned)lane * 16u, 0, GP_SC1);
}
// One wave waits for NS chunk slots (byte offsets off[]) and reads them.  Polling must be CHEAP: a spinning full sweep (20 KB per pass
// and wave, 1800 waves) saturates the fabric and starves every other access of the chip (first version: 700 us per step).  The slots
// are read in full once; if a tag is missing, lane k polls one SENTINEL -- the last 16 bytes of slot k -- with a sleep between polls,
// and the slots are read again when every sentinel carries `tag`.  false on time-out / peer failure.
// POLL_FIRST: the caller arrives before the data as a rule (the hand-offs of the critical path): start with the sentinels.
// BATCH: slots read per round trip (a wave that keeps 100 weight registers cannot hold 10 loads' worth of granules as well).
template <int NS, bool POLL_FIRST, int BATCH = NS>
__device__ __forceinline__ bool gp_sweep(const GpBuf& b, const unsigned (&off)[NS], int ns, int lane, unsigned tag, f32x4 (&v)[NS], gu32* err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // (the offsets as scalar VALUES first: hipcc turns `c ? off[k] : off[0]` into a load through a selected pointer, which keeps the
  // array in scratch / LDS)
  unsigned of[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) of[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)off[k]);
  unsigned so = of[0];
#pragma unroll
  for (int k = 1; k < NS; ++k) so = (lane == k && k < ns) ? of[k] : so;
  so += 1024u + 63u * 16u;
  bool read_now = !POLL_FIRST;
  for (unsigned spins = 0;; ++spins) {
    if (read_now) {
      bool ok = true;
#pragma unroll
      for (int k0 = 0; k0 < NS; k0 += BATCH) {
        u32x4 x[BATCH][2];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int k = k0 + j < NS ? k0 + j : NS - 1;
          const unsigned o = (k < ns ? of[k] : of[0]) + (unsigned)lane * 16u; // (unconditional loads: a slot beyond ns re-reads slot 0)
          x[j][0] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, o, 0, GP_SC1 | GP_VOL);
          x[j][1] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, o + 1024u, 0, GP_SC1 | GP_VOL);
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int k = k0 + j;
          if (k < NS) {
            // (real moves: hipcc otherwise keeps the {value, tag, value, tag} load tuples alive and picks the values out of them
            // where they are used -- twice the registers, in 4-aligned tuples)
            float v0, v1, v2, v3;
            asm volatile("v_mov_b32 %0, %1" : "=v"(v0) : "v"(x[j][0][0]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(v1) : "v"(x[j][0][2]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(v2) : "v"(x[j][1][0]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(v3) : "v"(x[j][1][2]));
            v[k] = f32x4{v0, v1, v2, v3};
            ok &= (k >= ns) || (x[j][0][1] == tag && x[j][0][3] == tag && x[j][1][1] == tag && x[j][1][3] == tag);
          }
        }
      }
      if (__all(ok)) return true;
      asm volatile("" ::: "memory");
      if (spins > 1000000u) return false;
    }
    read_now = true;
    for (unsigned polls = 0;; ++polls) {
      const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(b.rs, so, 0, GP_SC1 | GP_VOL);
      if (__all(y[1] == tag && y[3] == tag)) break;
      asm volatile("" ::: "memory");
      if ((polls & 63) == 63) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull ||          // 1 s at 100 MHz
            __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// LDS of a workgroup (NT = 5: 152 KB)
template <int NT>
struct GpLds {
  float wp[GP_NKB][NT][64];                 // W_p^T fragments: A operand of the partial projection [k-block of P][k-step of 4 cells][lane]
  float mB[GP_NR][GP_NKB][64][4];           // carried m(t-1) as B fragments [row tile][k-block][lane][4]
  float pb[4][NT][GP_NR][64][4];            // accumulator tiles: x-part (X wave w -> R wave w), then the R waves' partial sums
  float st[6][GP_ROWS][4 * NT];             // the step's stash: gates i, j, f, o | c | h   (h also feeds the projection)
  float gs[2][4][64][4];                    // the G waves' partial chunk sums by parity
  float xs[4][NT][64][4];                   // X wave w parks its row-tile-0 accumulators here while it works on row tile 1
  float kx4[2][NT][64][4];                  // the fifth K_x k-block of X waves 0, 1 (k-blocks 16, 17): 100 weight registers do not fit beside the sweeps
  float peep[4 * NT][4];                    // {w_i, w_f, w_o, -} per cell of this slice (one 16-byte read per cell)
  float bias[4 * NT][4];                    // {b_i, b_j, b_f, b_o} per cell: the accumulator registers of a lane
  unsigned cnt_x[4], cnt_p, cnt_h, cnt_m, cnt_g, dead, cnt_s, pad_[6];
};

template <int NT>
__device__ __forceinline__ void gp_fwd_body(const GPersistArgs& a, const unsigned gen, GpLds<NT>& S) {
  constexpr int NR = GP_NR, NU = NT * NR, CW = 4 * NT;
  GPT_DECL
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GP_TRACE
  if (tid == 0) gp_tr[0][21] = (unsigned)__builtin_amdgcn_s_memtime();          // kernel entry -> [0][20]: the prologue (weights into VGPRs / LDS)
#endif
  // block -> (row group, layer, slice): block b runs on XCD b & 7 (observed; speed only) -- a row group owns 8 / groups XCDs
  const int ngr = a.N / GP_ROWS, xpg = 8 / ngr;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = xcd / xpg, idx = slot * xpg + (xcd % xpg);
  if (idx >= a.nl * a.NC) return;
  const int l = idx / a.NC, c = idx - l * a.NC;
  const GPersistLayer L = a.L[l];
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, P = L.P, ldP = L.ldP, I = L.I, NC = a.NC;
  const int nkb = (P + 15) >> 4, nch = nkb * NR, nkbx = (I + 15) >> 4;
  const int row0 = grp * GP_ROWS, cell0 = c * CW;
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  // hop 1: [group][layer][parity][chunk][producer] slots; hop 2: [group][layer][t][chunk] slots
  const size_t g1_per = (size_t)GP_NCH * NC * GP_SLOT, g2_per = (size_t)GP_NCH * GP_SLOT;
  const GpBuf b1 = gp_buf((const char*)a.gran1 + (size_t)(grp * a.nl + l) * 2 * g1_per, 2 * g1_per);
  const GpBuf b2 = gp_buf((const char*)a.gran2 + (size_t)(grp * a.nl + l) * T * g2_per, (size_t)T * g2_per);
  const GpBuf b2x = gp_buf((const char*)a.gran2 + (size_t)(grp * a.nl + (l > 0 ? l - 1 : 0)) * T * g2_per, (size_t)T * g2_per);
  const unsigned tagbase = gen << 11;                                  // hop 1: tag = generation (21 bits) and step + 1

  // ---- cooperative prologue: W_p fragments, peepholes, bias, counters ----
  for (int e = tid; e < GP_NKB * NT * 64; e += GP_WAVES * 64) {
    const int ln = e & 63, ks = (e >> 6) % NT, jb = e / (64 * NT);
    const int col = 16 * jb + (ln & 15), cell = cell0 + 4 * ks + (ln >> 4);
    const float v = L.Wp[(size_t)min(cell, H - 1) * ldP + min(col, P - 1)];
    S.wp[jb][ks][ln] = (col < P && cell < H) ? v : 0.f;
  }
  for (int e = tid; e < 7 * CW; e += GP_WAVES * 64) {
    const int k = e / CW, cl = e - k * CW, cell = min(cell0 + cl, H - 1);
    if (k < 3) S.peep[cl][k] = (k == 0 ? L.wi : k == 1 ? L.wf : L.wo)[cell];
    else S.bias[cl][k - 3] = L.bias[(k - 3) * H + cell];
  }
  if (tid < 16) (&S.cnt_x[0])[tid] = 0u;
  __syncthreads();
  const unsigned* dead = &S.dead;
  auto fail = [&]() {
    if (lane == 0) {
      __hip_atomic_store(&S.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  const int len0 = a.len[row0 + lr], len1 = a.len[row0 + 16 + lr];
  // the stash of step t (gate activations, c, h: 15 KB) from its LDS stage, a quarter per R wave: NT consecutive lanes write one
  // 16 NT-byte row piece.  The R waves do it: they idle during the hand-off and touch no other global memory (the X waves' sweeps
  // queued behind these stores, and a wave that publishes or polls must not have them in front of its hand-off traffic).
  auto stash = [&](int t, int rw) {
#pragma unroll
    for (int it = 0; it < (6 * GP_ROWS * NT + 255) / 256; ++it) {
      const int e = it * 256 + rw * 64 + lane;
      const int cq = e % NT, pr = e / NT, row = pr % GP_ROWS, k = min(pr / GP_ROWS, 5);
      const float4 v = *reinterpret_cast<const float4*>(&S.st[k][row][4 * cq]);
      const size_t rowg = (size_t)t * N + row0 + row;
      float* dst = (k < 4 ? L.gates + rowg * H4 + k * H : k == 4 ? L.c + (rowg + N) * H : L.h + rowg * L.ldH) + cell0 + 4 * cq;
      if (e < 6 * GP_ROWS * NT && cell0 + 4 * cq < H) *reinterpret_cast<float4*>(dst) = v;
    }
  };

  if (w < 4) {
    // =============================== R waves: the critical path ===============================
    // resident K_h fragments: A[row lr = 4 * cell + gate][k = 16 jb + 4 q + u], jb = w + 4 jj
    float4 kh[NT][GP_KBW];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int cell = cell0 + 4 * i + (lr >> 2);
      const float* kr = L.KhT + (size_t)((lr & 3) * H + min(cell, H - 1)) * ldP;
#pragma unroll
      for (int jj = 0; jj < GP_KBW; ++jj) {
        const int k = 16 * (w + 4 * jj) + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(kr + min(k, ldP - 4));
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));              // (the load stays unconditional)
        const bool ok = k < P && cell < H;
        kh[i][jj] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
      }
    }
    float cprev[3] = {0.f, 0.f, 0.f};
    // every LDS access below is one base register + a compile-time offset (hipcc otherwise hoists dozens of loop-invariant addresses
    // out of the step loop and spills them).  Cell units of this wave: u = w + 4 s = (tile (w >> 1) + 2 s, row tile w & 1); projection
    // chunks: ch = w + 4 n = (k-block (w >> 1) + 2 n, row tile w & 1)
    float* const pbw = &S.pb[w][0][0][lane][0];                        // + (i * NR + r) * 256
    const float* const mbw = &S.mB[0][w][lane][0];                     // + (r * GP_NKB + 4 jj) * 256
    const float* const pbc = &S.pb[0][w >> 1][w & 1][lane][0];         // + (k * NU + 2 s * NR) * 256
    const float* const pwc = &S.peep[4 * (w >> 1) + q][0];             // + 32 s
    const int rowc = 16 * (w & 1) + lr, clc = 4 * (w >> 1) + q, lenc = (w & 1) ? len1 : len0;
    for (int t = 0; t < T; ++t) {
#ifdef GP_TRACE
      if (t == 0 && w == 0 && lane == 0) gp_tr[0][20] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
      GPT(0);
      if (!gp_wait(&S.cnt_x[w], (unsigned)t + 1u, dead)) return;       // x-part (+ bias) of step t
      GPT(1);
      if (t > 0 && !gp_wait(&S.cnt_m, 4u * (unsigned)t, dead)) return; // carried m(t-1) is in LDS
      GPT(2);
      __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int r = 0; r < NR; ++r) {                                   // (one row tile at a time: 20 accumulator registers instead of 40)
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = *reinterpret_cast<const f32x4*>(pbw + (i * NR + r) * 256);
        if (t > 0) {
#pragma unroll
          for (int jj = 0; jj < GP_KBW; ++jj) {
            if (w + 4 * jj < nkb) {
              const float4 b = *reinterpret_cast<const float4*>(mbw + (r * GP_NKB + 4 * jj) * 256);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[i][jj].x, b.x, acc[i], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[i][jj].y, b.y, acc[i], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[i][jj].z, b.z, acc[i], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[i][jj].w, b.w, acc[i], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) *reinterpret_cast<f32x4*>(pbw + (i * NR + r) * 256) = acc[i];
      }
      __builtin_amdgcn_s_setprio(0);
      GPT(3);
      gp_signal(&S.cnt_p, lane);
      if (!gp_wait(&S.cnt_p, 4u * ((unsigned)t + 1u), dead)) return;
      GPT(4);
      // the cell, on the accumulator layout: lane (q, lr) of unit (tile i, row tile r) = row 16 r + lr, cell 4 i + q, gates i j f o
      if (t > 0 && !gp_wait(&S.cnt_s, 4u * (unsigned)t, dead)) return; // the stash of step t-1 has left the stage (long ago: right after its cells)
      float* const stc = &S.st[0][rowc][clc];                          // + k * GP_ROWS * CW + 8 s
      const bool live = t < lenc;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (w + 4 * s < NU) {
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(pbc + (0 * NU + 2 * s * NR) * 256), p1 = *reinterpret_cast<const f32x4*>(pbc + (1 * NU + 2 * s * NR) * 256);
          const f32x4 p2 = *reinterpret_cast<const f32x4*>(pbc + (2 * NU + 2 * s * NR) * 256), p3 = *reinterpret_cast<const f32x4*>(pbc + (3 * NU + 2 * s * NR) * 256);
          const f32x4 z = ((p0 + p1) + p2) + p3;
          const float cpv = cprev[s];
          const f32x4 pw = *reinterpret_cast<const f32x4*>(pwc + 32 * s);
          const float gi = gp_sigmoid(z[0] + pw[0] * cpv);
          const float gf = gp_sigmoid(z[2] + a.forget_bias + pw[1] * cpv);
          const float gj = gp_tanh(z[1]);
          const float cn = gf * cpv + gi * gj;
          const float go = gp_sigmoid(z[3] + pw[2] * cn);
          const float hh = go * gp_tanh(cn);
          cprev[s] = live ? cn : cpv;
          stc[0 * GP_ROWS * CW + 8 * s] = live ? gi : 0.f; stc[1 * GP_ROWS * CW + 8 * s] = live ? gj : 0.f;
          stc[2 * GP_ROWS * CW + 8 * s] = live ? gf : 0.f; stc[3 * GP_ROWS * CW + 8 * s] = live ? go : 0.f;
          stc[4 * GP_ROWS * CW + 8 * s] = cprev[s];
          stc[5 * GP_ROWS * CW + 8 * s] = live ? hh : 0.f;
        }
      }
      gp_signal(&S.cnt_h, lane);
      GPT(5);
      if (!gp_wait(&S.cnt_h, 4u * ((unsigned)t + 1u), dead)) return;   // every cell of the step is in the stage
      stash(t, w);
      gp_signal(&S.cnt_s, lane);
    }
#ifdef GP_TRACE
    if (w == 0) { { const int i0_ = 0, i1_ = 6; GPT_FLUSH(); } { const int i0_ = 20, i1_ = 22; GPT_FLUSH(); } }
#endif
    return;
  }

  if (w < 8) {
    // =============================== X waves: one step ahead ===============================
    const int xw = w - 4;
    if (l == 0) {
      // layer 0: the x-part of every step was batched into `gates` (zx = x . K_x + bias); X wave xw fetches the units u = xw (mod 4)
      for (int t = 0; t < T; ++t) {
        float zv[3][4];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int u = min(xw + 4 * s, NU - 1), i = u / NR, r = u - i * NR;
          const int cell = min(cell0 + 4 * i + q, H - 1);
          const float* zr = L.gates + ((size_t)t * N + row0 + 16 * r + lr) * H4 + cell;
#pragma unroll
          for (int g = 0; g < 4; ++g) zv[s][g] = zr[g * H];
        }
        if (t > 0 && !gp_wait(&S.cnt_h, 4u * (unsigned)t, dead)) return;     // the cell of step t-1 has read the tiles
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int u = i * NR + r;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s)
              if (u == xw + 4 * s) v = f32x4{zv[s][0], zv[s][1], zv[s][2], zv[s][3]};
            *reinterpret_cast<f32x4*>(&S.pb[xw][0][0][lane][0] + (i * NR + r) * 256) = v;
          }
        gp_signal(&S.cnt_x[xw], lane);
      }
      return;
    }
    float4 kx[NT][GP_KBW - 1];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int cell = cell0 + 4 * i + (lr >> 2);
      const float* kr = L.KxT + (size_t)((lr & 3) * H + min(cell, H - 1)) * L.ldI;
#pragma unroll
      for (int jj = 0; jj < GP_KBW; ++jj) {
        const int k = 16 * (xw + 4 * jj) + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(kr + min(k, L.ldI - 4));
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        const bool ok = k < I && cell < H;                             // (the copy is zero beyond column I)
        const float4 f = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        if (jj < GP_KBW - 1) kx[i][jj] = f;
        else if (xw < 2) *reinterpret_cast<float4*>(&S.kx4[xw][i][lane][0]) = f;
      }
    }
    const float* const kx4w = &S.kx4[xw & 1][0][lane][0];              // + i * 256
    for (int t = 0; t < T; ++t) {
      // both row tiles BEFORE the wait for the cells of step t-1: behind it only the tile stores are left, so the R waves get the
      // x-part of step t right after their cell of step t-1.  (Row tile 0's accumulators wait in this wave's own LDS stage: 20
      // registers less while row tile 1 is swept.)
      f32x4 acc[NT];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        // x(t) = the masked output of the layer below: chunks (k-block xw + 4 jj, row tile r) of its m(t)
        unsigned off[GP_KBW];
        int ns = 0;
#pragma unroll
        for (int jj = 0; jj < GP_KBW; ++jj) {
          const int jb = xw + 4 * jj;
          off[jj] = (unsigned)(((size_t)t * GP_NCH + min(jb, nkbx - 1) * NR + r) * GP_SLOT);
          ns += jb < nkbx ? 1 : 0;
        }
        f32x4 xv[GP_KBW];
        GPT(8 + 2 * r);
        if (!gp_sweep<GP_KBW, false, 3>(b2x, off, ns, lane, gen, xv, err)) { fail(); return; }
        GPT(9 + 2 * r);
        const bool live = t < (r ? len1 : len0);
#pragma unroll
        for (int i = 0; i < NT; ++i)
          acc[i] = xw == 0 ? *reinterpret_cast<const f32x4*>(&S.bias[q][0] + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(2);                                 // (below the R waves' and the projection's bursts, above every spin loop)
#pragma unroll
        for (int jj = 0; jj < GP_KBW; ++jj) {
          if (xw + 4 * jj < nkbx) {
            const f32x4 b = live ? xv[jj] : f32x4{0.f, 0.f, 0.f, 0.f};
            float4 ka[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) ka[i] = jj < GP_KBW - 1 ? kx[i][jj < GP_KBW - 1 ? jj : 0] : *reinterpret_cast<const float4*>(kx4w + i * 256);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].x, b[0], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].y, b[1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].z, b[2], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].w, b[3], acc[i], 0, 0, 0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        GPT(16 + r);
        if (r == 0) {
#pragma unroll
          for (int i = 0; i < NT; ++i) *reinterpret_cast<f32x4*>(&S.xs[xw][i][lane][0]) = acc[i];
        }
      }
      if (t > 0 && !gp_wait(&S.cnt_h, 4u * (unsigned)t, dead)) return;                   // the cell of step t-1 has read the tiles
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        *reinterpret_cast<f32x4*>(&S.pb[xw][0][0][lane][0] + (i * NR + 0) * 256) = *reinterpret_cast<const f32x4*>(&S.xs[xw][i][lane][0]);
        *reinterpret_cast<f32x4*>(&S.pb[xw][0][0][lane][0] + (i * NR + 1) * 256) = acc[i];
      }
      gp_signal(&S.cnt_x[xw], lane);
      GPT(18);
      GPT(19);
    }
#ifdef GP_TRACE
    if (xw == 0) { { const int i0_ = 8, i1_ = 12; GPT_FLUSH(); } { const int i0_ = 16, i1_ = 20; GPT_FLUSH(); } }
#endif
    return;
  }

  // =============================== G waves: reduce, publish, gather, stash ===============================
  __builtin_amdgcn_s_setprio(1);                                       // (polls sleep between tries; the R waves' MFMA bursts run at 3)
  const int gw = w - 8;
  const bool reducer = c < nch;                                        // this workgroup sums chunk c = (k-block jbr, row tile rr)
  const int jbr = c / NR, rr = c - jbr * NR;
  const int lenr = rr ? len1 : len0;
  const int ppw = (NC + 3) >> 2, pp0 = gw * ppw, pn = max(0, min(ppw, NC - pp0));   // this wave's producers [pp0, pp0 + pn), pn <= 10
  f32x4 mcar = f32x4{0.f, 0.f, 0.f, 0.f};                              // carried state of chunk c (the stash's mst)
  // the carried state of the chunks this wave gathers (ch = gw + 4 n) lives in mB itself: zero before step 0
#pragma unroll
  for (int n = 0; n < 9; ++n)
    if (gw + 4 * n < nch) *reinterpret_cast<f32x4*>(&S.mB[gw & 1][gw >> 1][lane][0] + 2 * n * 256) = f32x4{0.f, 0.f, 0.f, 0.f};
  // slot 0 of the carried states is zero (cell.zero_state)
  if (gw >= 2) {
    for (int e = (gw - 2) * 64 + lane; e < GP_ROWS * NT; e += 128) {
      const int row = e / NT, cq = e - row * NT;
      if (cell0 + 4 * cq < H) *reinterpret_cast<float4*>(L.c + (size_t)(row0 + row) * H + cell0 + 4 * cq) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else if (gw == 1 && reducer) {
    if (16 * jbr + 4 * q < ldP) *reinterpret_cast<float4*>(L.mst + (size_t)(row0 + 16 * rr + lr) * ldP + 16 * jbr + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // The partial projection of this slice's cells and its publication run HERE, not on the R waves: whatever vector-memory access
  // follows the write-through granule stores in a wave's queue waits for their acknowledgement (vmcnt retires in order) -- the R waves
  // touch no global memory at all.  Chunk ch = gw + 4 n = (k-block (gw >> 1) + 2 n of P, row tile gw & 1):
  // m^T[col 16 jb + 4 q + i][row 16 r + lr] = sum_k W_p[cell k][col] h[row][cell k]   (k-blocks beyond P hold zero weights)
  const float* const wpw = &S.wp[gw >> 1][0][lane];                    // + (2 n * NT + ks) * 64
  const int rowp = 16 * (gw & 1) + lr;
  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    f32x4 total = f32x4{0.f, 0.f, 0.f, 0.f};
    GPT(6);
    if (!gp_wait(&S.cnt_h, 4u * ((unsigned)t + 1u), dead)) return;     // the cells of step t: h is in LDS
    GPT(7);
    {
      const float* const sth = &S.st[5][rowp][q];                      // + 4 ks
      float hv[NT];
#pragma unroll
      for (int ks = 0; ks < NT; ++ks) hv[ks] = sth[4 * ks];
      const unsigned tagp = tagbase | ((unsigned)t + 1u);
      const unsigned pub0 = (unsigned)(((size_t)par * GP_NCH + gw) * NC + c) * GP_SLOT;
      __builtin_amdgcn_s_setprio(3);
      // three chunks in flight (the dependent-accumulator latency of the 16x16x4 form is 40 cycles for a 32-cycle issue); a chunk
      // leaves as soon as its NT products are done, so the write-through stores overlap the remaining MFMAs
#pragma unroll
      for (int n0 = 0; n0 < 9; n0 += 3) {
        f32x4 pm[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) pm[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NT; ++ks)
#pragma unroll
          for (int j = 0; j < 3; ++j) pm[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpw[(2 * (n0 + j) * NT + ks) * 64], hv[ks], pm[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (gw + 4 * (n0 + j) < nch) gp_publish(b1, pub0 + (unsigned)(4 * (n0 + j)) * (unsigned)NC * GP_SLOT, lane, tagp, pm[j]);
      }
      __builtin_amdgcn_s_setprio(1);
    }
    GPT(12);
    if (reducer) {
      // hop 1: the NC partial projections of chunk c, summed in slice order
      const unsigned tag1 = tagbase | ((unsigned)t + 1u);
      unsigned off[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) off[k] = (unsigned)(((size_t)par * GP_NCH + c) * NC + min(pp0 + k, NC - 1)) * GP_SLOT;
      f32x4 pv[10];
      if (!gp_sweep<10, true>(b1, off, pn, lane, tag1, pv, err)) { fail(); return; }
      GPT(13);
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 10; ++k)
        if (k < pn) s = k == 0 ? pv[0] : s + pv[k];
      *reinterpret_cast<f32x4*>(&S.gs[par][gw][lane][0]) = s;
      gp_signal(&S.cnt_g, lane);
      if (!gp_wait(&S.cnt_g, 4u * ((unsigned)t + 1u), dead)) return;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(&S.gs[par][0][lane][0]), s1 = *reinterpret_cast<const f32x4*>(&S.gs[par][1][lane][0]);
      const f32x4 s2 = *reinterpret_cast<const f32x4*>(&S.gs[par][2][lane][0]), s3 = *reinterpret_cast<const f32x4*>(&S.gs[par][3][lane][0]);
      total = ((s0 + s1) + s2) + s3;
      // hop 2: chunk c of m(t)
      if (gw == 0) gp_publish(b2, (unsigned)(((size_t)t * GP_NCH + c) * GP_SLOT), lane, gen, total);
      GPT(14);
    }
    if (t + 1 < T) {
      // gather m(t) of this layer for the recurrent product of step t+1: chunks gw, gw + 4, ...; dynamic_rnn carries the state of a
      // finished row through unchanged
      unsigned off[9];
      int ns = 0;
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        const int ch = gw + 4 * n;
        off[n] = (unsigned)(((size_t)t * GP_NCH + min(ch, nch - 1)) * GP_SLOT);
        ns += ch < nch ? 1 : 0;
      }
      f32x4 mv[9];
      if (!gp_sweep<9, true>(b2, off, ns, lane, gen, mv, err)) { fail(); return; }
      GPT(15);
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        const int ch = gw + 4 * n;
        if (ch < nch) {
          const bool live = t < ((ch & 1) ? len1 : len0);
          if (live) *reinterpret_cast<f32x4*>(&S.mB[gw & 1][gw >> 1][lane][0] + 2 * n * 256) = mv[n];
        }
      }
      gp_signal(&S.cnt_m, lane);
    }
    // the step's stash, behind the hand-offs (plain stores: this wave's next poll waits for their acknowledgement, which comes
    // long before its peers' granules)
    if (gw == 1 && reducer) {
      const bool live = t < lenr;
      mcar = live ? total : mcar;
      const f32x4 o = live ? total : f32x4{0.f, 0.f, 0.f, 0.f};
      if (16 * jbr + 4 * q < ldP) {
        const size_t rowg = (size_t)row0 + 16 * rr + lr;
        *reinterpret_cast<f32x4*>(L.mst + ((size_t)(t + 1) * N + rowg) * ldP + 16 * jbr + 4 * q) = mcar;
        *reinterpret_cast<f32x4*>(L.out + ((size_t)t * N + rowg) * ldP + 16 * jbr + 4 * q) = o;
      }
    }
  }
#ifdef GP_TRACE
  if (gw == 0) { { const int i0_ = 6, i1_ = 8; GPT_FLUSH(); } { const int i0_ = 12, i1_ = 16; GPT_FLUSH(); } }
#endif
}

template <int NT>
__global__ __launch_bounds__(GP_WAVES * 64, 3) void k_glstm_fwd(const GPersistArgs a) {
  __shared__ __attribute__((aligned(16))) GpLds<NT> S;
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  gp_fwd_body<NT>(a, gen, S);
  __syncthreads();                                                 // (every wave leaves the body on every path)
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[a.nl - 1].out[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned g1 = gen + 1u;
      __hip_atomic_store(ctl + DP_CTL_GEN, g1 >= (1u << 21) ? 1u : g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int gp_grid(const GPersistArgs& a) {
  const int ngr = a.N / GP_ROWS, xpg = 8 / ngr, nwg = a.nl * a.NC;
  return 8 * ((nwg + xpg - 1) / xpg);
}

bool gpersist_plan(GPersistArgs& a) {
  if (a.nl < 1 || a.nl > GP_MAXL || a.T < 1 || a.T > 2046 || a.H % 4 != 0) return false;
  const int ngr = a.N / GP_ROWS;
  if (a.N % GP_ROWS != 0 || (ngr != 1 && ngr != 2 && ngr != 4 && ngr != 8)) return false;
  a.NT = 5;
  a.NC = (a.H / 4 + a.NT - 1) / a.NT;
  for (int l = 0; l < a.nl; ++l) {
    const GPersistLayer& L = a.L[l];
    if (L.P < 4 || L.P > 16 * GP_NKB || L.P % 4 != 0 || L.ldP % 4 != 0 || L.I > 16 * GP_NKB || L.ldH % 4 != 0) return false;
    if (l > 0 && L.I != a.L[l - 1].P) return false;
    if (((L.P + 15) / 16) * GP_NR > a.NC) return false;            // every chunk needs its reducer
    if (a.NC > 40) return false;                                    // a G wave sums at most 10 producers
  }
  // every workgroup must be resident at once (they wait for each other): one 12-wave workgroup per CU
  return gp_grid(a) <= 256;
}
size_t gpersist_gran1_bytes(const GPersistArgs& a) { return (size_t)(a.N / GP_ROWS) * a.nl * 2 * GP_NCH * a.NC * GP_SLOT; }
size_t gpersist_gran2_bytes(const GPersistArgs& a) { return (size_t)(a.N / GP_ROWS) * a.nl * a.T * GP_NCH * GP_SLOT; }

void launch_glstm_fwd(const GPersistArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_glstm_fwd<5>, dim3(gp_grid(a)), dim3(GP_WAVES * 64), 0, s, a);
  ++g_chain_launches;
}

}  // namespace rsr
