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
//     critical compute (m(t-1).K_h, the cell, the stash); X0..X3 keep K_x and run AHEAD of the R waves (the x-part of step t only
//     needs the layer below); G0..G3 project, publish, poll and reduce -- the R waves never wait for global memory;
//   * all products run TRANSPOSED (weights = MFMA A operand, activations = B operand), as in dpersist.hip;
//   * the two 16-row tiles of a group are two independent LANES of the same workgroup, half a chain apart: while tile 0's
//     hand-off is in flight (G waves 0, 2) the R waves compute tile 1 (whose hand-off G waves 1, 3 run), so a step costs one tile's
//     chain, not compute + hand-off of both (first version: 32 k cycles per step, 21 k of them hand-off with the R waves idle).
//
// Two hand-offs per step, layer and tile, both "the data is the flag" (cdna_hip_programming.md guideline 16 R2) in 16-byte
// write-through accesses -- without tags: a slot holds a sentinel pattern until it is written (see the transport section):
//   hop 1  every workgroup publishes its PARTIAL projection h[:, its cells] . W_p[its cells, :] (16 x P per tile) as 16 x 16 chunks,
//          one per k-block of P; workgroup c of the layer is the REDUCER of the 8-column half (k-block c >> 1, half c & 1) of BOTH
//          tiles: it sums the NC partials in slice order (deterministic) -- a reduce-scatter;
//   hop 2  the reducer publishes its half chunk of m(t); every workgroup of the layer (for the recurrent product of step t+1) and
//          of the layer above (x of step t) gathers all chunks -- an all-gather.  A chunk slot is the accumulator tile itself,
//          [64 lanes][16 bytes]: lane pl = (q, lr) of the producer / consumer fragment owns the four columns 4 q .. 4 q + 3 of row lr, so
//          every store / load instruction moves whole 128-byte lines and a reducer's half (lanes 32 hh ..) is 512 contiguous bytes.
// Inside a workgroup the three roles synchronise through monotonic LDS counters (no s_barrier: the roles are not in lock step).
// Hop 2 has one slot per step (armed by a memset in front of the launch); hop 1 is a ring of GP_R1 steps whose slots the reducer
// re-arms after summing them.  Every spin is bounded; failures go to the sticky err word of the control block and poison the top layer's output with NaN.
#include <type_traits>

#include "kernels.h"

namespace rsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
#include "dpersist_dev.h"            // (the trailing form of the discriminator's BPTT rides inside k_glstm_bwd_dt)

constexpr int GP_NR = GP_ROWS / 16;      // row tiles per group
constexpr int GP_WAVES = 12;             // R0..R3, X0..X3, G0..G3
static_assert(GP_NR == 2, "chunk index = 2 * k-block + row tile");
constexpr int GP_NKB = 18;               // k-blocks of 16 of the recurrent / input width (P, I <= 288)
constexpr int GP_KBW = 5;                // k-blocks per R / X wave (k-block jb belongs to wave jb & 3)
constexpr int GP_SLOT = 1024;            // bytes per chunk slot: a 16 x 16 tile of floats, 16 bytes (four columns of one row) per fragment lane
constexpr unsigned GP_SENT = 0xFFFFFFFFu; // a word of a slot nobody has written yet
#ifndef GP_R1_STEPS
#define GP_R1_STEPS 3
#endif
constexpr int GP_R1 = GP_R1_STEPS;                 // steps in the hop-1 ring: a slot is summed in step t, re-armed at the end of step t + 1, written again in step t + 3
constexpr int GP_NCH = GP_NKB * GP_NR;   // chunk slots per layer and step (the layout's stride; a layer uses its first nkb * NR)
constexpr unsigned GP_SC1 = 16u;         // aux of the raw-buffer builtins: sc1 (agent scope: write-through store / L1-bypassing load)
constexpr unsigned GP_VOL = 1u << 31;    // ... compiler-only: volatile (a polled load must not be hoisted out of its loop)
#ifndef GP_GATE_NUM                      // progressive sweeps start reading when GP_GATE_NUM / GP_GATE_DEN of the awaited sentinel pieces are there
#define GP_GATE_NUM 1
#define GP_GATE_DEN 2
#endif
#ifndef GP_HB10                          // pieces per round trip of a progressive pass (hop 1: 10 per lane, hop 2: 9)
#define GP_HB10 10
#define GP_HB9 9
#endif

#ifdef GP_TRACE
__device__ unsigned g_gp_cnt[8];          // (GP_COUNT builds only: the atomics distort the timeline)          // [0] cached first reads, [1] of them with a stale / missing tag; [2], [3] the same for write-through first reads
__device__ unsigned g_gp_trace[256][24][24];
#define GPT_DECL __shared__ unsigned gp_tr[24][24];
#define GPT(i) do { if ((w & 3) == 0 && lane == 0 && t < 24) gp_tr[t][i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define GPTS(i) do { if ((w & 3) == 0 && lane == 0 && s < 24) gp_tr[s][i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
// (the backward kernel has no register to spare: -DGP_TRACE_NOR / -DGP_TRACE_NOG drop the stamps of its R / X waves or of its G waves,
//  so that the other half is traced on the product's register allocation)
#ifdef GP_TRACE_NOR
#define GPTSR(i) do { } while (0)
#else
#define GPTSR(i) GPTS(i)
#endif
#ifdef GP_TRACE_NOG
#define GPTSG(i) do { } while (0)
#else
#define GPTSG(i) GPTS(i)
#endif
#define GPT_FLUSH() do { if (lane == 0) { _Pragma("unroll 1") for (int t_ = 0; t_ < 24; ++t_) { _Pragma("unroll 1") for (int i_ = i0_; i_ < i1_; ++i_) g_gp_trace[blockIdx.x][t_][i_] = gp_tr[t_][i_]; } } } while (0)
#else
#define GPT_DECL
#define GPT(i) do { } while (0)
#define GPTS(i) do { } while (0)
#define GPTSR(i) do { } while (0)
#define GPTSG(i) do { } while (0)
#define GPT_FLUSH() do { } while (0)
#endif

__device__ __forceinline__ float gp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float gp_tanh(float x) {          // (dpersist.hip dp_tanh: ~2e-7 absolute)
  const float x2 = x * x;
  const float ser = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.05396825f * x2)));
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x));
  return fabsf(x) < 0.1f ? ser : big;
}

// The stash (gate activations, c, h, dz: written once per launch, read by the NEXT launch at the earliest) streams through the
// memory system beside the hand-off rings, which want to stay in the Infinity Cache.  GP_STASH_NT = 1: non-temporal stores / loads
// for it (measured: profiles/r5_hbm_phases.txt).
#ifndef GP_STASH_NT
#define GP_STASH_NT 0
#endif
__device__ __forceinline__ void gp_stash_store(float* dst, const float4& v) {
#if GP_STASH_NT
  __builtin_nontemporal_store(v.x, dst); __builtin_nontemporal_store(v.y, dst + 1); __builtin_nontemporal_store(v.z, dst + 2); __builtin_nontemporal_store(v.w, dst + 3);
#else
  *reinterpret_cast<float4*>(dst) = v;
#endif
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
// The payload carries no tags: a slot holds GP_SENT in every word until its producer has written it, and a 16-byte piece is valid
// when none of its four words is GP_SENT (each 32-bit word lands whole, so a torn piece is an invalid piece).  Half the bytes of the
// {value, tag} granules of dpersist.hip -- the generator's hand-offs move 220 KB into every workgroup per step and are bound by
// that, not by the tag compares (timing ablation: 15.4 -> 14.2 us per step forward, 19.4 -> 16.4 backward).  The price is that
// somebody has to put the sentinels back: the ring slots (hop 1, the backward's input-gradient ring) have exactly one reader, the
// reducer workgroup, whose R waves re-arm them once its G waves have summed them; the all-gathered chunks (hop 2: one slot per step,
// ~40 readers) are re-armed by a memset in front of every launch (launch_glstm_*).  GP_SENT is a NaN pattern no arithmetic
// produces; a value that happens to carry it (a NaN payload passed through from the inputs) is published as the canonical NaN.
struct GpBuf { __amdgpu_buffer_rsrc_t rs; };
__device__ __forceinline__ GpBuf gp_buf(const void* p, size_t bytes) {
  GpBuf b; b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); return b;
}
__device__ __forceinline__ void gp_store(const GpBuf& b, unsigned off, const f32x4& v) {
  u32x4 x = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = x[i] == GP_SENT ? 0x7FC00000u : x[i];
  __builtin_amdgcn_raw_buffer_store_b128(x, b.rs, off, 0, GP_SC1);
}
__device__ __forceinline__ bool gp_valid(const u32x4& x) { return (x[0] != GP_SENT) & (x[1] != GP_SENT) & (x[2] != GP_SENT) & (x[3] != GP_SENT); }
// TAGGED ring slots (round 5): the lowest mantissa bit of every published word carries the PARITY OF THE RING PASS that wrote it
// ((step counter / ring depth) & 1, the counter carried from launch to launch in the control block, GP_CTL_C1 / _C3), so what the
// previous pass left in a slot is told from this pass's data without anybody putting sentinels back: the re-arming stores were half
// of the launches' write-through traffic (profiles/r5_gpersist_trace.txt "no re-arming").  A 32-bit word lands whole, so a torn piece
// is a piece with mixed parities = not valid yet.  The value gives up its last bit: the producer rounds it to a 22-bit mantissa (ties
// to even: <= 1 ulp, unbiased), the consumer clears the tag again (gp_untag), so what is summed does not depend on the pass parity --
// the same inputs give the same bits whatever was launched before.
__device__ __forceinline__ void gp_store_t(const GpBuf& b, unsigned off, const f32x4& v, unsigned tag) {
  u32x4 x = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = ((x[i] + ((x[i] >> 1) & 1u)) & ~1u) | tag;
  __builtin_amdgcn_raw_buffer_store_b128(x, b.rs, off, 0, GP_SC1);
}
template <bool TAGGED>
__device__ __forceinline__ f32x4 gp_untag(const u32x4& x) {
  const unsigned m = TAGGED ? ~1u : ~0u;
  return f32x4{__uint_as_float(x[0] & m), __uint_as_float(x[1] & m), __uint_as_float(x[2] & m), __uint_as_float(x[3] & m)};
}
__device__ __forceinline__ bool gp_valid_t(const u32x4& x, unsigned tag) {
  const unsigned all1 = x[0] & x[1] & x[2] & x[3] & 1u, any1 = (x[0] | x[1] | x[2] | x[3]) & 1u;
  return tag ? all1 != 0u : any1 == 0u;
}
#define GP_CIDX(a) ((a).ngl ? 2 * (a).grp0 : 0)      // (a launch over some row groups only: their rings advance on their own counters)
constexpr int GP_CTL_C1 = 4, GP_CTL_C3 = 5;      // control-block words: steps written so far to the hop-1 ring (mod 2 GP_R1) / the input-gradient rings (mod 2 GP_XR)
// Re-arm the NP producers' 512-byte half chunks at base + p * GP_SLOT (p = 0 .. NP-1): one store covers two producers (a half
// wave each); wave w of the four R waves takes every fourth store.
__device__ __forceinline__ void gp_rearm(const GpBuf& b, unsigned base, int NP, int w, int lane) {
#ifdef GP_NOREARM                   // (harness experiment: rings as deep as the launch is long, armed by a memset in front of every launch)
  return;
#endif
  const u32x4 sent = {GP_SENT, GP_SENT, GP_SENT, GP_SENT};
  for (int k = w; 2 * k < NP; k += 4) {
    const int p = 2 * k + (lane >> 5);
    if (p < NP) __builtin_amdgcn_raw_buffer_store_b128(sent, b.rs, base + (unsigned)p * GP_SLOT + (unsigned)(lane & 31) * 16u, 0, GP_SC1);
  }
}
// One wave waits for NL 16-byte-per-lane pieces (wave-uniform byte offsets lo[], + this lane's lane_off; piece k counts for this lane
// when k < nl, a per-lane number) and hands them to consume(k, piece) in order.  Polling must be CHEAP: a spinning full read (20 KB
// per pass and wave, 1800 waves) saturates the fabric and starves every other access of the chip (first version: 700 us per step).
// So a lane polls one SENTINEL piece (byte offset so, the last 16 bytes of something it waits for; son = this lane has one) with a
// sleep between polls; only when every sentinel piece is valid the pieces are read in full, and re-read in the rare case that a
// store of theirs has not landed yet.  POLL_FIRST: the caller arrives before the data as a rule (the hand-offs of the critical
// path).  BATCH: pieces per round trip.  consume() must be idempotent (a failed pass is repeated).
// CACHED: the first full read uses ordinary (L2-cacheable) loads.  The all-gathered chunks are read by ~19 workgroups per XCD, and
// as write-through (sc1) reads every one of them crossed the fabric.  A line cached before its store landed holds sentinels (the
// L2s are invalidated at the kernel boundary behind the memset), i.e. it is detected like a store that has not landed, and the retry
// reads past the caches (sc1).
// false on time-out / peer failure.
template <int NL, bool POLL_FIRST, int BATCH, bool CACHED, bool TAGGED = false, class F>
__device__ __forceinline__ bool gp_sweep(const GpBuf& b, const unsigned (&lo)[NL], int nl, unsigned lane_off, unsigned so, bool son,
                                         gu32* err, F&& consume, unsigned tag = 0u) {
  static_assert(!(TAGGED && CACHED), "tagged ring slots are read past the caches");
  auto valid = [&](const u32x4& x) { return TAGGED ? gp_valid_t(x, tag) : gp_valid(x); };
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // (the offsets as scalar VALUES first: hipcc turns `c ? lo[k] : lo[0]` into a load through a selected pointer, which keeps the
  // array in scratch / LDS)
  unsigned of[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) of[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)lo[k]);
  auto read_all = [&](auto through_caches) -> bool {
    constexpr unsigned AUX = decltype(through_caches)::value ? GP_VOL : (GP_SC1 | GP_VOL);
    bool ok = true;
#pragma unroll
    for (int k0 = 0; k0 < NL; k0 += BATCH) {
      u32x4 x[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int k = k0 + j < NL ? k0 + j : NL - 1;
        x[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, (k < nl ? of[k] : of[0]) + lane_off, 0, AUX);   // (unconditional)
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int k = k0 + j;
        if (k < NL) {
          ok &= (k >= nl) || valid(x[j]);
          consume(k, gp_untag<TAGGED>(x[j]));
        }
      }
    }
    return __all(ok);
  };
  bool read_now = !POLL_FIRST, first = true;
  for (unsigned spins = 0;; ++spins) {
#ifdef GP_ABL
    if (read_now && POLL_FIRST && ((GP_ABL >> (CACHED ? 2 : 1)) & 1)) return true;      // timing ablation: the sentinels only
    if (!POLL_FIRST && (GP_ABL & 1)) return true;                                        // timing ablation: no x sweeps
#endif
    if (read_now) {
      const bool ok = (CACHED && first) ? read_all(std::true_type{}) : read_all(std::false_type{});
#ifdef GP_COUNT
      if (first && (threadIdx.x & 63) == 0) { atomicAdd(&g_gp_cnt[CACHED ? 0 : 2], 1u); if (!ok) atomicAdd(&g_gp_cnt[CACHED ? 1 : 3], 1u); }
#endif
      first = false;
      if (ok) return true;
      asm volatile("" ::: "memory");
      if (spins > 1000000u) return false;
    }
    read_now = true;
    for (unsigned polls = 0;; ++polls) {
      const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(b.rs, so, 0, GP_SC1 | GP_VOL);
      if (__all(!son || valid(y))) break;
      asm volatile("" ::: "memory");
      if ((polls & 63) == 63) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull ||          // 1 s at 100 MHz
            __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// The PROGRESSIVE form of a sweep (round 5).  gp_sweep above costs a hand-off two round trips after the last piece has landed -- the
// poll that notices it, then the full read (10 pieces per lane: ~4.5 k cycles of the forward period for hop 1, ~2.5 k for hop 2,
// profiles/r4_gpersist_trace.txt "sentinels only") -- although most of the pieces arrived long before the last one (the producers
// finish 7-9 k cycles apart).  Here the data is its own flag all the way: once `gate` of the awaited sentinel pieces are there (cheap
// polling until then: a lane per producer, as before -- a wave that arrives early must not spin on full reads, that is what
// saturated the fabric in the first version), every pass reads the pieces that are still PENDING, keeps the ones that came back
// valid in registers, and asks again for the rest only: a finished piece's load is sent to an out-of-range offset (a raw-buffer load
// beyond num_records returns zeros without a memory access).  What is left behind the last arrival is one round trip of the one or
// two pieces that were missing.  out[] is complete and in slot order when the function returns, so the caller's sum keeps its fixed
// order whatever the arrival order was.  nlw: wave-uniform number of pieces (>= every lane's nl).
template <int NL, int HB>
__device__ __forceinline__ bool gp_sweep_prog(const GpBuf& b, const unsigned (&lo)[NL], int nl, int nlw, unsigned lane_off, unsigned so, bool son,
                                              int gate, gu32* err, f32x4 (&out)[NL]) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned of[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) of[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)lo[k]);
  if (gate > 0) {
    for (unsigned polls = 0;; ++polls) {
      const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(b.rs, so, 0, GP_SC1 | GP_VOL);
      if (__popcll(__ballot(son && gp_valid(y))) >= gate) break;
      asm volatile("" ::: "memory");
      if ((polls & 63) == 63) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  unsigned pend = (unsigned)__builtin_amdgcn_readfirstlane((int)((1u << nlw) - 1u));
  for (unsigned pass = 0;; ++pass) {
#pragma unroll
    for (int k0 = 0; k0 < NL; k0 += HB) {
      if (((pend >> k0) & ((1u << HB) - 1u)) == 0u) continue;                       // (uniform)
      u32x4 t[HB];
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const int k = k0 + j < NL ? k0 + j : NL - 1;
        const bool want = k0 + j < NL && ((pend >> k) & 1u) && k < nl;
        t[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, want ? of[k] + lane_off : OOB, 0, GP_SC1 | GP_VOL);   // (unconditional)
      }
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const int k = k0 + j;
        if (k < NL) {
          const bool ok = __all(k >= nl || gp_valid(t[j]));
          if (((pend >> k) & 1u) && ok) {                                              // (uniform)
            out[k] = f32x4{__uint_as_float(t[j][0]), __uint_as_float(t[j][1]), __uint_as_float(t[j][2]), __uint_as_float(t[j][3])};
            pend &= ~(1u << k);
          }
        }
      }
    }
    if (pend == 0u) return true;
    asm volatile("" ::: "memory");
    if ((pass & 31) == 31) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// LDS of a workgroup (NT = 5: 129 KB)
template <int NT>
struct GpLds {
  // (small, hot arrays first: a DS immediate offset reaches 64 KB; see DpTrailLds)
  unsigned cnt_x[GP_NR][4], cnt_p[GP_NR], cnt_h[GP_NR], cnt_m[GP_NR], cnt_g[GP_NR], cnt_s[GP_NR], dead, cnt_j[GP_NR], pad_[11];
  int len[GP_ROWS];                         // the rows' lengths: read per step (a register that holds one for the whole launch was spilled, and a scratch reload sat in the cell phase and in front of the publication)
  float peep[4 * NT][4];                    // {w_i, w_f, w_o, -} per cell of this slice (one 16-byte read per cell)
  float bias[4 * NT][4];                    // {b_i, b_j, b_f, b_o} per cell: the accumulator registers of a lane
  float gs[2][GP_NR][2][64][4];             // the two reducing G waves' partial sums of this workgroup's half chunk [step parity][tile][wave][half wave = even / odd producers, fragment lane]
  float st[6][GP_ROWS][4 * NT];             // the step's stash: gates i, j, f, o | c | h   (h also feeds the projection)
  float kx4[2][NT][64][4];                  // the fifth K_x k-block of X waves 0, 1 (k-blocks 16, 17): 100 weight registers do not fit beside the sweeps
  float wp[GP_NKB][NT][64];                 // W_p^T fragments: A operand of the partial projection [k-block of P][k-step of 4 cells][lane]
  float mB[GP_NR][GP_NKB][64][4];           // carried m(t-1) as B fragments [row tile][k-block][lane][4]
  float pb[4][NT][GP_NR][64][4];            // accumulator tiles: x-part (X wave w -> R wave w), then the R waves' partial sums
};

// PROG: bit 0 the reducers' hop-1 sweeps, bit 1 the gathers of hop 2 in the progressive form (gp_sweep_prog); RSRGAN_GP_PROG
// RES: a residual stack (models/res_lstm_l.py:101-194): layer l + 1 reads s_l = out_l + s_{l-1} instead of out_l (s_{-1} = the stack's
// input).  The reducer of a half chunk is also the owner of that half chunk of the running sum: it adds its masked m(t) to the half
// chunk of s_{l-1}(t) (layer 0: the input rows in memory; above: what the same-numbered reducer of the layer below published a step
// ago), writes it to res_out and publishes it in the second region of gran2, where the layer above's X waves gather their x(t).
template <int NT, int PROG, bool RES, bool TAG>
__device__ __forceinline__ void gp_fwd_body(const GPersistArgs& a, GpLds<NT>& S, const unsigned c1, const unsigned bid) {
  static_assert(!(TAG && PROG), "the progressive sweeps know the sentinel form only");
  constexpr int NR = GP_NR, NU = NT * NR, CW = 4 * NT;
  GPT_DECL
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GP_TRACE
  if (tid == 0) gp_tr[0][23] = (unsigned)__builtin_amdgcn_s_memtime();          // kernel entry -> [0][22]: the prologue (weights into VGPRs / LDS)
#endif
  // block -> (row group, layer, slice): block b runs on XCD b & 7 (observed; speed only) -- a row group owns 8 / groups XCDs
  // (a.ngl row groups starting at a.grp0, or all of them: a stack whose workgroups do not fit the device at once runs one launch per row group)
  const int ngr = a.N / GP_ROWS, ngl = a.ngl ? a.ngl : ngr, xpg = 8 / ngl;
  const int xcd = bid & 7, slot = bid >> 3;
  const int grp = a.grp0 + xcd / xpg, idx = slot * xpg + (xcd % xpg);
  if (idx >= a.nl * a.NC) return;
  const int l = idx / a.NC, c = idx - l * a.NC;
  const GPersistLayer L = a.L[l];
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, P = L.P, ldP = L.ldP, I = L.I, NC = a.NC;
  const int nkb = (P + 15) >> 4, nkbx = (I + 15) >> 4;
  const int row0 = grp * GP_ROWS, cell0 = c * CW;
  const int nrt = a.nrt ? a.nrt : NR;                                 // live row tiles (GPersistArgs::nrt)
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  // hop 1: [group][layer][parity][tile][k-block][producer] slots; hop 2: [group][layer][t][tile][k-block] slots
  const size_t g1_per = (size_t)GP_NCH * NC * GP_SLOT, g2_per = (size_t)GP_NCH * GP_SLOT;
  const GpBuf b1 = gp_buf((const char*)a.gran1 + (size_t)(grp * a.nl + l) * GP_R1 * g1_per, GP_R1 * g1_per);
  const GpBuf b2 = gp_buf((const char*)a.gran2 + (size_t)(grp * a.nl + l) * T * g2_per, (size_t)T * g2_per);
  // what the layer above gathers as its x(t): the m chunks themselves, or (RES) the running sums in the second region
  const char* const g2x = RES ? (const char*)a.gran2 + (size_t)ngr * a.nl * T * g2_per : (const char*)a.gran2;
  const GpBuf b2x = gp_buf(g2x + (size_t)(grp * a.nl + (l > 0 ? l - 1 : 0)) * T * g2_per, (size_t)T * g2_per);
  const GpBuf b2s = gp_buf(g2x + (size_t)(grp * a.nl + l) * T * g2_per, (size_t)T * g2_per);      // (RES: where this layer publishes s_l)
  const unsigned frag_off = (unsigned)lane * 16u;                     // a fragment lane's bytes in a chunk slot
  const unsigned pair_off = (unsigned)((lane >> 5) * GP_SLOT + (lane & 31) * 16);   // reducer: lanes 0..31 read producer p's half chunk, lanes 32..63 producer p + 1's
  auto slot1 = [&](int par, int r, int jb, int p) { return (unsigned)((((size_t)(par * NR + r) * GP_NKB + jb) * NC + p) * GP_SLOT); };
  auto slot2 = [&](int t, int r, int jb) { return (unsigned)((((size_t)t * NR + r) * GP_NKB + jb) * GP_SLOT); };
  // this workgroup REDUCES the 8-column half (k-block jbr, half hh) of both tiles
  const bool reducer = c < 2 * nkb;
  const int jbr = c >> 1, hh = c & 1;

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
  for (int e = tid; e < GP_NR * GP_NKB * 64; e += GP_WAVES * 64)       // the carried state m(-1) is zero (cell.zero_state)
    *reinterpret_cast<f32x4*>(&S.mB[0][0][0][0] + 4 * e) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 32) (&S.cnt_x[0][0])[tid] = 0u;
  if (tid < GP_ROWS) S.len[tid] = a.len[row0 + tid];
  __syncthreads();
  const unsigned* dead = &S.dead;
  auto fail = [&]() {
    if (lane == 0) {
      __hip_atomic_store(&S.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(err, 1u + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  if (w < 4) {
    // =============================== R waves: the critical compute ===============================
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
    // cell units of a row tile: gate tile w on every wave, gate tile 4 + w' (w' = 0 .. NT - 5) on wave w' as well
    float cprev[NR][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // every LDS access below is one base register + a compile-time offset (hipcc otherwise hoists dozens of loop-invariant addresses
    // out of the step loop and spills them)
    float* const pbw = &S.pb[w][0][0][lane][0];                        // + (i * NR + r) * 256
    const float* const mbw = &S.mB[0][w][lane][0];                     // + (r * GP_NKB + 4 jj) * 256
    const float* const pbc = &S.pb[0][w][0][lane][0];                  // + (k * NU + s * 4 * NR + r) * 256   (gate tile w + 4 s)
    const float* const pwc = &S.peep[4 * w + q][0];                    // + s * 64
    float* const stc = &S.st[0][lr][4 * w + q];                        // + k * GP_ROWS * CW + r * 16 * CW + s * 16
    const bool two = w + 4 < NT;                                       // this wave owns a second gate tile
    for (int t = 0; t < T; ++t) {
#ifdef GP_TRACE
      if (t == 0 && w == 0 && lane == 0) gp_tr[0][22] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r >= nrt) continue;
        GPT(6 * r + 0);
        if (!gp_wait(&S.cnt_x[r][w], (unsigned)t + 1u, dead)) return;     // x-part (+ bias) of step t, tile r
        GPT(6 * r + 1);
        if (t > 0 && !gp_wait(&S.cnt_m[r], 2u * (unsigned)t, dead)) return;   // carried m(t-1) of the tile is in LDS
        GPT(6 * r + 2);
        {
          f32x4 acc[NT];
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[i] = *reinterpret_cast<const f32x4*>(pbw + (i * NR + r) * 256);
          if (t > 0) {
            __builtin_amdgcn_s_setprio(2);
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
            __builtin_amdgcn_s_setprio(0);
          }
#pragma unroll
          for (int i = 0; i < NT; ++i) *reinterpret_cast<f32x4*>(pbw + (i * NR + r) * 256) = acc[i];
        }
        GPT(6 * r + 3);
        gp_signal(&S.cnt_p[r], lane);
        if (!gp_wait(&S.cnt_p[r], 4u * ((unsigned)t + 1u), dead)) return;
        if (t > 0 && !gp_wait(&S.cnt_s[r], 4u * (unsigned)t, dead)) return;  // the stash of step t-1 has left the stage (long ago)
        GPT(6 * r + 4);
        // the cell, on the accumulator layout: lane (q, lr) of (gate tile i, row tile r) = row 16 r + lr, cell 4 i + q, gates i j f o
        const bool live = t < S.len[16 * r + lr];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s == 0 || two) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(pbc + (0 * NU + s * 4 * NR + r) * 256), p1 = *reinterpret_cast<const f32x4*>(pbc + (1 * NU + s * 4 * NR + r) * 256);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(pbc + (2 * NU + s * 4 * NR + r) * 256), p3 = *reinterpret_cast<const f32x4*>(pbc + (3 * NU + s * 4 * NR + r) * 256);
            const f32x4 z = ((p0 + p1) + p2) + p3;
            const float cpv = cprev[r][s];
            const f32x4 pw = *reinterpret_cast<const f32x4*>(pwc + s * 64);
            const float gi = gp_sigmoid(z[0] + pw[0] * cpv);
            const float gf = gp_sigmoid(z[2] + a.forget_bias + pw[1] * cpv);
            const float gj = gp_tanh(z[1]);
            const float cn = gf * cpv + gi * gj;
            const float go = gp_sigmoid(z[3] + pw[2] * cn);
            const float hh = go * gp_tanh(cn);
            cprev[r][s] = live ? cn : cpv;
            float* const d = stc + r * 16 * CW + s * 16;
            d[0 * GP_ROWS * CW] = live ? gi : 0.f; d[1 * GP_ROWS * CW] = live ? gj : 0.f;
            d[2 * GP_ROWS * CW] = live ? gf : 0.f; d[3 * GP_ROWS * CW] = live ? go : 0.f;
            d[4 * GP_ROWS * CW] = cprev[r][s];
            d[5 * GP_ROWS * CW] = live ? hh : 0.f;
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the re-arming stores of the step before: acknowledged long ago)
        gp_signal(&S.cnt_h[r], lane);
        GPT(6 * r + 5);
        if (!gp_wait(&S.cnt_h[r], 4u * ((unsigned)t + 1u), dead)) return;     // every cell of the tile is in the stage
        if ((a.sched & 2) && !gp_wait(&S.cnt_j[r], 2u * ((unsigned)t + 1u), dead)) return;   // (the lane's partial projections have left first)
        // the tile's stash (gate activations, c, h: 7.5 KB) from the LDS stage, a quarter per R wave: NT consecutive lanes write one
        // 16 NT-byte row piece.  The R waves do it: they have nothing in the vector-memory queue that it could delay (the X waves'
        // sweeps queued behind these stores, and a wave that publishes or polls must not have them in front of its hand-off traffic)
#pragma unroll
        for (int it = 0; it < (6 * 16 * NT + 255) / 256; ++it) {
          const int e = it * 256 + w * 64 + lane;
          const int cq = e % NT, pr = e / NT, row = 16 * r + (pr & 15), k = min(pr >> 4, 5);
          const float4 v = *reinterpret_cast<const float4*>(&S.st[k][row][4 * cq]);
          const size_t rowg = (size_t)t * N + row0 + row;
          float* dst = (k < 4 ? L.gates + rowg * H4 + k * H : k == 4 ? L.c + (rowg + N) * H : L.h + rowg * L.ldH) + cell0 + 4 * cq;
#ifdef GP_ABL2
          if (GP_ABL2 & 2) continue;                                     // timing ablation: no stash stores
#endif
          if (e < 6 * 16 * NT && cell0 + 4 * cq < H) gp_stash_store(dst, v);
        }
        gp_signal(&S.cnt_s[r], lane);
        // This workgroup's G waves summed the partial projections of step t-1 before they gathered m(t-1) (the wait at the top):
        // re-arm those ring slots, here, where the wave has nothing urgent to do.  The stores are acknowledged before the wave
        // signals its cells of step t+1, i.e. before this workgroup's partials of step t+1 leave, without which no m(t+1) and
        // hence no partial of step t+2 -- the next write to these slots -- exists.
        if (!TAG && reducer && t > 0) gp_rearm(b1, slot1((t - 1) % GP_R1, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
      }
    }
    if (!TAG && reducer) {                                               // the last step's partials: leave every ring slot armed for the next launch
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r >= nrt) continue;
        if (!gp_wait(&S.cnt_m[r], 2u * (unsigned)T, dead)) return;
        gp_rearm(b1, slot1((T - 1) % GP_R1, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
      }
    }
#ifdef GP_TRACE
    if (w == 0) { { const int i0_ = 0, i1_ = 12; GPT_FLUSH(); } { const int i0_ = 22, i1_ = 24; GPT_FLUSH(); } }
#endif
    return;
  }

  if (w < 8) {
    // =============================== X waves: ahead of the R waves ===============================
    const int xw = w - 4;
    float* const pbx = &S.pb[xw][0][0][lane][0];                       // + (i * NR + r) * 256
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
    const int nsx = (nkbx - xw + 3) >> 2;                              // this wave's k-blocks: xw, xw + 4, ...
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r >= nrt) continue;
        // x(t) = the masked output of the layer below: fragments (k-block xw + 4 jj, tile r) of its m(t), two 16-byte pieces each
        unsigned lo[GP_KBW];
#pragma unroll
        for (int jj = 0; jj < GP_KBW; ++jj) lo[jj] = slot2(t, r, min(xw + 4 * jj, nkbx - 1));
        f32x4 xv[GP_KBW];
        GPT(18 + 2 * r);
        if (l == 0) {
          // layer 0: x(t) is the stack's input (models/lstm.py:82-87: the leaky-ReLU FC over the LPS frame), in memory before the launch:
          // the same fragments as plain loads (16 bytes of one row per lane), no hand-off.  (Until round 4 this product was a
          // time-batched GEMM in front of the launch, 136 us; these waves and 60 % of the MFMA pipe were idle.)
          const float* xr = L.in + ((size_t)t * N + row0 + 16 * r + lr) * L.ldI;
#pragma unroll
          // (rows are zero beyond column I up to ldI, kernels.h layout rule: a piece may straddle I -- res_lstm_l's 257 columns)
          for (int jj = 0; jj < GP_KBW; ++jj)                            // (all five in flight, unconditional)
            xv[jj] = *reinterpret_cast<const f32x4*>(xr + min(16 * min(xw + 4 * jj, nkbx - 1) + 4 * q, L.ldI - 4));
          asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]));
          static_assert(GP_KBW == 5, "the five pieces above");
#pragma unroll
          for (int jj = 0; jj < GP_KBW; ++jj)
            if (16 * min(xw + 4 * jj, nkbx - 1) + 4 * q >= L.ldI) xv[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (!gp_sweep<GP_KBW, false, GP_KBW, true>(b2x, lo, nsx, frag_off, slot2(t, r, min(xw + 4 * (lane >> 1), nkbx - 1)) + (unsigned)(lane & 1) * 512u + 496u,
                                                    lane < 2 * nsx, err, [&](int k, const f32x4& v) { xv[k] = v; })) { fail(); return; }
        GPT(19 + 2 * r);
        // (sched bit 0) not beside the lane's projection burst and publication: the product of step t starts when the G waves have issued
        // the partial projections of step t - 1 (then the lane waits for its hand-off and the other lane's R burst is half a period away)
        if ((a.sched & 1) && t > 0 && !gp_wait(&S.cnt_j[r], 2u * (unsigned)t, dead)) return;
        const bool live = t < S.len[16 * r + lr];
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
          acc[i] = xw == 0 ? *reinterpret_cast<const f32x4*>(&S.bias[q][0] + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);                                 // (below the R waves' and the projection's bursts, above every spin loop)
#pragma unroll
        for (int jj = 0; jj < GP_KBW; ++jj) {
#ifdef GP_ABL2
          if (GP_ABL2 & 1) continue;                                   // timing ablation: no x-part products
#endif
          if (xw + 4 * jj < nkbx) {
            const float b0 = live ? xv[jj][0] : 0.f, b1 = live ? xv[jj][1] : 0.f, b2_ = live ? xv[jj][2] : 0.f, b3 = live ? xv[jj][3] : 0.f;
            float4 ka[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) ka[i] = jj < GP_KBW - 1 ? kx[i][jj < GP_KBW - 1 ? jj : 0] : *reinterpret_cast<const float4*>(kx4w + i * 256);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].x, b0, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].y, b1, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].z, b2_, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].w, b3, acc[i], 0, 0, 0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        if (t > 0 && !gp_wait(&S.cnt_h[r], 4u * (unsigned)t, dead)) return;       // the cells of step t-1 have read the tiles
#pragma unroll
        for (int i = 0; i < NT; ++i) *reinterpret_cast<f32x4*>(pbx + (i * NR + r) * 256) = acc[i];
        gp_signal(&S.cnt_x[r][xw], lane);
      }
    }
#ifdef GP_TRACE
    if (xw == 0) { const int i0_ = 18, i1_ = 22; GPT_FLUSH(); }
#endif
    return;
  }

  // =============================== G waves: project, publish, reduce, gather (tile gw & 1) ===============================
  __builtin_amdgcn_s_setprio(3);                                       // the hand-off is the critical path and these waves issue little (their polls sleep); R bursts run at 2, X bursts at 1
  const int gw = w - 8, r = gw & 1, gp = gw >> 1;                      // this wave's tile, and which of the tile's two G waves it is
  // The partial projection of this slice's cells and its publication run HERE, not on the R waves: whatever vector-memory access
  // follows the write-through granule stores in a wave's queue waits for their acknowledgement (vmcnt retires in order).
  // Chunk n of this wave = k-block gp + 2 n of P, tile r:
  // m^T[col 16 jb + 4 q + i][row 16 r + lr] = sum_k W_p[cell k][col] h[row][cell k]   (k-blocks beyond P hold zero weights)
  const float* const wpw = &S.wp[gp][0][lane];                         // + (2 n * NT + ks) * 64
  const float* const sth = &S.st[5][16 * r + lr][q];                   // + 4 ks
  float* const mbg = &S.mB[r][gp][lane][0];                            // + 2 n * 256
  const int nvg = (nkb - gp + 1) >> 1;                                 // this wave's k-blocks: gp, gp + 2, ...
  // reducer: this wave sums tile r's partials of the producers [pp0, pp0 + pn), two per load (a half wave each: lanes 0..31 the
  // even ones, lanes 32..63 the odd ones), nlr of them in this lane
  const int ppw = (NC + 1) >> 1, pp0 = gp * ppw, pn = max(0, min(ppw, NC - pp0));
  const int nlr = (pn + 1 - (lane >> 5)) >> 1;
  // reducer lane L holds fragment lane pl = 32 hh + (L & 31): row pl & 15, columns 4 (pl >> 4) .. + 3 of the chunk
  const int rrow = row0 + 16 * r + ((32 * hh + (lane & 31)) & 15), rcol = 16 * jbr + 4 * ((32 * hh + (lane & 31)) >> 4);
  const int rlen = a.len[rrow];
  f32x4 mcar = {0.f, 0.f, 0.f, 0.f};                                   // carried state of the reducer's four columns (the stash's mst)
  // slot 0 of the carried states is zero (cell.zero_state)
  for (int e = gw * 64 + lane; e < GP_ROWS * NT; e += 256) {
    const int row = e / NT, cq = e - row * NT;
    if (cell0 + 4 * cq < H) *reinterpret_cast<float4*>(L.c + (size_t)(row0 + row) * H + cell0 + 4 * cq) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (reducer && gp == 1 && lane < 32 && rcol < ldP) *reinterpret_cast<float4*>(L.mst + (size_t)rrow * ldP + rcol) = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r >= nrt) return;                                                // (a tile of padding rows: nothing to project, publish, sum or gather)
  for (int t = 0; t < T; ++t) {
    const int par = t & 1, par1 = (int)((c1 + (unsigned)t) % GP_R1);      // (c1 = 0 without tags)
    const unsigned tag1 = ((c1 + (unsigned)t) / GP_R1) & 1u;
    GPT(12);
    if (!gp_wait(&S.cnt_h[r], 4u * ((unsigned)t + 1u), dead)) return;  // the cells of step t, tile r: h is in LDS
    GPT(13);
    {
      float hv[NT];
#pragma unroll
      for (int ks = 0; ks < NT; ++ks) hv[ks] = sth[4 * ks];
      const unsigned pub0 = slot1(par1, r, gp, c) + frag_off;
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
          for (int j = 0; j < 3; ++j) {
#ifdef GP_ABL2
            if (GP_ABL2 & 4) continue;                                 // timing ablation: no projection products
#endif
            pm[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpw[(2 * (n0 + j) * NT + ks) * 64], hv[ks], pm[j], 0, 0, 0);
          }
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (n0 + j < nvg) { if (TAG) gp_store_t(b1, pub0 + (unsigned)(2 * (n0 + j)) * (unsigned)NC * GP_SLOT, pm[j], tag1); else gp_store(b1, pub0 + (unsigned)(2 * (n0 + j)) * (unsigned)NC * GP_SLOT, pm[j]); }
      }
    }
    if (a.sched & 3) gp_signal(&S.cnt_j[r], lane);
    GPT(14);
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    if (reducer) {
      // hop 1: the NC partial projections of this half chunk, tile r, summed in a fixed order (this wave: its half of the slices,
      // even and odd ones in the two half waves)
      unsigned lo[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) lo[k] = slot1(par1, r, jbr, min(pp0 + 2 * k, NC - 1)) + (unsigned)hh * 512u;
      f32x4 sa = {0.f, 0.f, 0.f, 0.f};
      if (PROG & 1) {
        f32x4 pv[10];
        if (!gp_sweep_prog<10, GP_HB10>(b1, lo, nlr, (pn + 1) >> 1, pair_off, slot1(par1, r, jbr, min(pp0 + lane, NC - 1)) + (unsigned)hh * 512u + 496u, lane < pn,
                                  (pn * GP_GATE_NUM) / GP_GATE_DEN, err, pv)) { fail(); return; }
#pragma unroll
        for (int k = 0; k < 10; ++k) { if (k == 0) sa = k < nlr ? pv[0] : f32x4{0.f, 0.f, 0.f, 0.f}; else if (k < nlr) sa += pv[k]; }      // (slot order: the same bits as the other form)
      } else if (!gp_sweep<10, true, 10, false, TAG>(b1, lo, nlr, pair_off, slot1(par1, r, jbr, min(pp0 + lane, NC - 1)) + (unsigned)hh * 512u + 496u, lane < pn, err,
                                  [&](int k, const f32x4& v) { if (k == 0) sa = k < nlr ? v : f32x4{0.f, 0.f, 0.f, 0.f}; else if (k < nlr) sa += v; }, tag1)) { fail(); return; }
      GPT(15);
      *reinterpret_cast<f32x4*>(&S.gs[par][r][gp][lane][0]) = sa;
      gp_signal(&S.cnt_g[r], lane);
      if (!gp_wait(&S.cnt_g[r], 2u * ((unsigned)t + 1u), dead)) return;
      {
        const float* const gsr = &S.gs[par][r][0][lane & 31][0];
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(gsr), a1 = *reinterpret_cast<const f32x4*>(gsr + 32 * 4);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(gsr + 64 * 4), c1 = *reinterpret_cast<const f32x4*>(gsr + 96 * 4);
        tot = ((a0 + a1) + c0) + c1;
      }
      // hop 2: this half chunk of m(t), tile r
      if (gp == 0 && lane < 32) gp_store(b2, slot2(t, r, jbr) + (unsigned)hh * 512u + (unsigned)lane * 16u, tot);
      GPT(16);
    }
    if (t + 1 < T) {
      // gather m(t) of the tile for the recurrent product of step t+1: k-blocks gp, gp + 2, ... as B fragments; dynamic_rnn carries the
      // state of a finished row through unchanged (the carried state lives in mB itself)
      unsigned lo[9];
#pragma unroll
      for (int n = 0; n < 9; ++n) lo[n] = slot2(t, r, min(gp + 2 * n, nkb - 1));
      f32x4 mv[9];
      if (PROG & 2) {
        if (!gp_sweep_prog<9, GP_HB9>(b2, lo, nvg, nvg, frag_off, slot2(t, r, min(gp + 2 * (lane >> 1), nkb - 1)) + (unsigned)(lane & 1) * 512u + 496u,
                                 lane < 2 * nvg, (2 * nvg * GP_GATE_NUM) / GP_GATE_DEN, err, mv)) { fail(); return; }
      } else if (!gp_sweep<9, true, 9, true>(b2, lo, nvg, frag_off, slot2(t, r, min(gp + 2 * (lane >> 1), nkb - 1)) + (unsigned)(lane & 1) * 512u + 496u,
                                 lane < 2 * nvg, err, [&](int k, const f32x4& v) { mv[k] = v; })) { fail(); return; }
      GPT(17);
      const bool live = t < S.len[16 * r + lr];
#pragma unroll
      for (int n = 0; n < 9; ++n)
        if (n < nvg && live) *reinterpret_cast<f32x4*>(mbg + 2 * n * 256) = mv[n];
    }
    gp_signal(&S.cnt_m[r], lane);                                        // (also after the last step: the R waves re-arm the ring behind it)
    // the reducer's four columns of the stash (carried state, masked output), behind the hand-offs
    if (reducer && gp == 1 && lane < 32) {
      const bool live = t < rlen;
      mcar = live ? tot : mcar;
      if (rcol < ldP) {
        *reinterpret_cast<float4*>(L.mst + ((size_t)(t + 1) * N + rrow) * ldP + rcol) = make_float4(mcar[0], mcar[1], mcar[2], mcar[3]);
        *reinterpret_cast<float4*>(L.out + ((size_t)t * N + rrow) * ldP + rcol) = live ? make_float4(tot[0], tot[1], tot[2], tot[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (RES && reducer && gp == 1) {
      // s_l(t) = out_l(t) + s_{l-1}(t) of this half chunk (res_lstm_l.py:111,121,131,190); the layer above is a step behind, so this
      // sits behind the hand-offs as well.  s_{l-1}(t) was published before this layer's X waves could gather x(t): no real wait.
      const bool act = lane < 32;
      const unsigned off = slot2(t, r, jbr) + (unsigned)hh * 512u + (unsigned)(lane & 31) * 16u;
      f32x4 sb;
      if (l == 0) {
        const float4 v = *reinterpret_cast<const float4*>(L.in + ((size_t)t * N + rrow) * L.ldI + min(rcol, L.ldI - 4));
        sb = rcol < L.ldI ? f32x4{v.x, v.y, v.z, v.w} : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        u32x4 y;
        for (unsigned polls = 0;; ++polls) {
          y = __builtin_amdgcn_raw_buffer_load_b128(b2x.rs, off, 0, GP_SC1 | GP_VOL);
          if (__all(!act || gp_valid(y))) break;
          asm volatile("" ::: "memory");
          if ((polls & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { fail(); return; }
          __builtin_amdgcn_s_sleep(1);
        }
        sb = f32x4{__uint_as_float(y[0]), __uint_as_float(y[1]), __uint_as_float(y[2]), __uint_as_float(y[3])};
      }
      const f32x4 s4 = (t < rlen ? tot : f32x4{0.f, 0.f, 0.f, 0.f}) + sb;
      if (act && rcol < ldP) *reinterpret_cast<float4*>(L.res_out + ((size_t)t * N + rrow) * ldP + rcol) = make_float4(s4[0], s4[1], s4[2], s4[3]);
      if (act && (l + 1 < a.nl || a.fwd_trail)) gp_store(b2s, off, s4);      // (the top layer's: for the FC workgroups of k_glstm_fwd_dt)
    }
  }
#ifdef GP_TRACE
  if (gw == 0) { const int i0_ = 12, i1_ = 18; GPT_FLUSH(); }
#endif
}

template <int NT, int PROG, bool RES, bool TAG>
__global__ __launch_bounds__(GP_WAVES * 64, 3) void k_glstm_fwd(const GPersistArgs a) {
  __shared__ __attribute__((aligned(16))) GpLds<NT> S;
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (counts launches; nothing depends on it)
  // (TAG) steps written to the hop-1 ring by the launches so far, mod 2 GP_R1: every workgroup reads it here, the last one to finish moves it on
  const unsigned c1 = TAG ? __hip_atomic_load(ctl + GP_CTL_C1 + GP_CIDX(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  gp_fwd_body<NT, PROG, RES, TAG>(a, S, c1, blockIdx.x);
  __syncthreads();                                                 // (every wave leaves the body on every path)
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (TAG) __hip_atomic_store(ctl + GP_CTL_C1 + GP_CIDX(a), (c1 + (unsigned)a.T) % (2u * GP_R1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[a.nl - 1].out[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned g1 = gen + 1u;
      __hip_atomic_store(ctl + DP_CTL_GEN, g1 >= (1u << 21) ? 1u : g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// =====================================================================================================================================
// BPTT through the same stack as ONE persistent launch (the mirror of k_glstm_fwd; the launch-per-phase form is kernels.hip
// k_bwd_a2 + k_bwd_bp + k_bwd_b_red: 26 us per wavefront diagonal).  Same decomposition -- workgroup = (row group, layer, slice of
// 4 NT cells), two tile lanes, 12 waves -- and the same two hand-offs per step, now carrying gradients:
//   hop 1  every workgroup publishes its PARTIAL state gradient dz(t)[:, its gate columns] . K_h[:, its gate columns]^T (R waves) and
//          its partial input gradient dz(t) . K_x^T (X waves, to the LAYER BELOW); reducer c sums, for its 8-column half chunk of
//          both tiles, the NC input-gradient partials of the layer above at time t (the top layer reads d(outputs) from memory
//          instead), then the NC state-gradient partials of its own layer from time t + 1, masks finished rows: dm(t);
//   hop 2  it publishes the half chunk and writes it to the stash (dmt: the projection's weight gradient); every workgroup of the
//          layer gathers dm(t), multiplies by its W_p slice (dh), runs the cell's gradient (dz: in place over the gate activations
//          in the stash, and as the B operand of the two products above).
// The state-gradient ring has GP_R1 steps like the forward's (a producer cannot be two steps ahead of a consumer it needs the sum
// from; the third step is the re-arming's slack); the input-gradient ring between two layers has GP_XR steps and an explicit
// back-pressure check: the upper layer does not depend on the lower one, so before re-using a ring slot it polls the lower layer's
// dm chunk of GP_XR - 2 steps ago (published by the reducer AFTER it has summed and re-armed that slot).  Layer 0's input gradient
// is a time-batched GEMM over the dz stash afterwards.
// Cell gradient: kernels.hip k_bwd_a2 (peepholes, o's peephole on the new c, dynamic_rnn masking: a finished row has dz = 0 and
// carries dc through).

// ---- k_glstm_fwd_dt: the generator's forward recurrence with the discriminator's (D(G(x))) a few steps behind it, ONE launch ----
// The FC workgroup of 16-row tile fr follows the generator's top layer: per step it takes the layer's 18 all-gathered chunks of m(t) (RES:
// of the running sum s_top(t)) as they are published (hop-2 slots: valid when no word carries the armed pattern), forms
// y(t) = m(t) . W_out + b (models/lstm.py:121-124) on four waves -- wave w the k-blocks w, w + 4, ..: a partial sum each, which is
// exactly what layer 0 of the discriminator's stack expects from "the layer below" (dpersist_dev.h dp_fwdt_body, xin): four partials as
// generation-tagged granules (bias on wave 0's, gaussian_noise_layer's row noise on wave 1's) -- and three more waves sum the partials
// into y(t) for memory (the mse term, the output FC's gradients) and y(t) + noise (the discriminator's input rows).
struct GpFcLds { float yp[2][4][DP_KB][64][4]; int dead; };
__device__ __forceinline__ void gp_fcf_body(const GPersistArgs& a, const DPersistArgs& d, const unsigned dgen, GpFcLds& S, const int fr, const bool res) {
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (w >= 7) return;                                              // (seven waves; a barrier counts the surviving waves)
  const int T = a.T, N = d.N, Dn = d.L[0].I, K = d.fc_P, RTn = N >> 4;
  const int grp = fr >> 1, r = fr & 1, ltop = a.nl - 1, nkb = (K + 15) >> 4;
  const int ngr = a.N / GP_ROWS;
  gu32* err = (gu32*)d.ctl + DP_CTL_ERR;
  gu32* gerr = (gu32*)a.ctl + DP_CTL_ERR;
  const size_t g2_per = (size_t)GP_NCH * GP_SLOT;
  const char* const g2 = res ? (const char*)a.gran2 + (size_t)ngr * a.nl * T * g2_per : (const char*)a.gran2;
  const GpBuf b2 = gp_buf(g2 + (size_t)(grp * a.nl + ltop) * T * g2_per, (size_t)T * g2_per);
  const size_t slot_stride_t = (size_t)DP_NQ * DP_SLOT;
  gu64* const gout = (gu64*)d.gran + ((size_t)(d.nl * RTn + fr) * T) * slot_stride_t;      // edge nl: layer 0's input
  const int row = 16 * fr + lr;
  const bool deadt = a.nrt && r >= a.nrt;
  auto fail = [&]() { if (lane == 0) { S.dead = 1; __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } };
  if (w < 4) {
    // A operand: W_out[k = 16 jb + 4 q + u][p = 16 ct + lr], jb = w + 4 n (zero beyond K rows / Dn columns)
    float4 wA[5][DP_KB];
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
      for (int ct = 0; ct < DP_KB; ++ct) {
        float v[4];
        const int p = 16 * ct + lr;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = d.fc_w[(size_t)min(16 * (w + 4 * n) + 4 * q + u, K - 1) * d.ld_fcw + min(p, Dn - 1)];
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (16 * (w + 4 * n) + 4 * q + u < K && p < Dn) ? v[u] : 0.f;
        wA[n][ct] = make_float4(v[0], v[1], v[2], v[3]);
      }
    // what this wave's partial carries beside the product: wave 0 the bias, wave 1 the row noise (p = 16 ct + 4 q + i of row lr)
    float4 add[DP_KB];
#pragma unroll
    for (int ct = 0; ct < DP_KB; ++ct) {
      const int p = min(16 * ct + 4 * q, Dn - 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (w == 0) v = *reinterpret_cast<const float4*>(d.fc_b + p);
      if (w == 1 && d.noise) v = *reinterpret_cast<const float4*>(d.noise + (size_t)row * Dn + p);
      add[ct] = dp_sel(16 * ct + 4 * q < Dn, v, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      // this wave's chunks of step t: all in flight, valid when no word carries the armed pattern
      u32x4 x[5];
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      if (deadt) {                                                 // (GPersistArgs::nrt: the generator publishes nothing for a tile of padding rows; m = 0)
#pragma unroll
        for (int n = 0; n < 5; ++n) x[n] = u32x4{0u, 0u, 0u, 0u};
      } else
      for (unsigned polls = 0;; ++polls) {
        bool ok = true;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          const int jb = min(w + 4 * n, nkb - 1);
          x[n] = __builtin_amdgcn_raw_buffer_load_b128(b2.rs, (unsigned)((((size_t)t * GP_NR + r) * GP_NKB + jb) * GP_SLOT) + (unsigned)lane * 16u, 0, GP_SC1 | GP_VOL);
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) ok &= gp_valid(x[n]);
        if (__all(ok)) break;
        asm volatile("" ::: "memory");
        if ((polls & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                   __hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { fail(); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      f32x4 acc[DP_KB];
#pragma unroll
      for (int ct = 0; ct < DP_KB; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        if (w + 4 * n < nkb) {                                      // (wave-uniform)
          const float bz[4] = {__uint_as_float(x[n][0]), __uint_as_float(x[n][1]), __uint_as_float(x[n][2]), __uint_as_float(x[n][3])};
#pragma unroll
          for (int ct = 0; ct < DP_KB; ++ct) {
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][ct].x, bz[0], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][ct].y, bz[1], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][ct].z, bz[2], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][ct].w, bz[3], acc[ct], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < DP_KB; ++ct) {
        *reinterpret_cast<f32x4*>(&S.yp[t & 1][w][ct][lane][0]) = acc[ct];
        gu64* go_ = gout + (size_t)t * slot_stride_t + (size_t)w * DP_SLOT + ((size_t)ct * 64 + lane) * 4;
        dp_store2(go_, dgen, acc[ct][0] + add[ct].x, acc[ct][1] + add[ct].y);
        dp_store2(go_ + 2, dgen, acc[ct][2] + add[ct].z, acc[ct][3] + add[ct].w);
      }
      __syncthreads();                                             // yp[t & 1] is complete (and the summing waves have left yp of step t - 1)
      if (S.dead) return;
    }
    return;
  }
  // waves 4..6: column tile ct of y(t) for memory: p = 16 ct + 4 q + i of row lr
  const int ct = w - 4, p0 = 16 * ct + 4 * q;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), nz = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p0 < Dn) {
    bb = *reinterpret_cast<const float4*>(d.fc_b + p0);
    if (d.noise) nz = *reinterpret_cast<const float4*>(d.noise + (size_t)row * Dn + p0);
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    __syncthreads();
    if (S.dead) return;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(&S.yp[t & 1][0][ct][lane][0]), s1 = *reinterpret_cast<const f32x4*>(&S.yp[t & 1][1][ct][lane][0]);
    const f32x4 s2 = *reinterpret_cast<const f32x4*>(&S.yp[t & 1][2][ct][lane][0]), s3 = *reinterpret_cast<const f32x4*>(&S.yp[t & 1][3][ct][lane][0]);
    const f32x4 y = (((s0 + s1) + s2) + s3) + f32x4{bb.x, bb.y, bb.z, bb.w};
    if (p0 < Dn) {
      *reinterpret_cast<float4*>(d.dy + ((size_t)t * N + row) * d.ld_dy + p0) = make_float4(y[0], y[1], y[2], y[3]);
      *reinterpret_cast<float4*>(d.dtop + ((size_t)t * d.xd_Ns + d.xd_row0 + row) * d.ld_dtop + p0) = make_float4(y[0] + nz.x, y[1] + nz.y, y[2] + nz.z, y[3] + nz.w);
    }
  }
}

template <int NT>
union GpFdtLds { GpLds<NT> g; DpFwdTLds d; GpFcLds f; };
template <int NT, bool RES, bool TAG>
__global__ __launch_bounds__(GP_WAVES * 64, 3) void k_glstm_fwd_dt(const GPersistArgs a, const DPersistArgs d) {
  __shared__ __attribute__((aligned(16))) GpFdtLds<NT> S;
  const int nD = d.nl * (d.N >> 5) * DP_NQ, nReal = nD + (d.N >> 4), dt_pad = (nReal + 7) & ~7;
  const int dbid = (int)blockIdx.x;
  if (dbid < dt_pad) {
    if (dbid >= nReal) return;
    gu32* dctl = (gu32*)d.ctl;
    const unsigned dgen = __hip_atomic_load(dctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) { if (dbid < nD) S.d.dead = 0; else S.f.dead = 0; }
    __syncthreads();
    if (dbid < nD) { if (d.nrt == 1) dp_fwdt_body<true>(d, dgen, S.d, dbid, true); else dp_fwdt_body<false>(d, dgen, S.d, dbid, true); }
    else gp_fcf_body(a, d, dgen, S.f, dbid - nD, RES);
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(dctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)nReal - 1u) {
        if (__hip_atomic_load(dctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) d.L[d.nl - 1].out[0] = __builtin_nanf("");
        __hip_atomic_store(dctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dctl + DP_CTL_GEN, dgen + 1u == 0u ? 1u : dgen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned c1 = TAG ? __hip_atomic_load(ctl + GP_CTL_C1 + GP_CIDX(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  gp_fwd_body<NT, 0, RES, TAG>(a, S.g, c1, blockIdx.x - (unsigned)dt_pad);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - (unsigned)dt_pad - 1u) {
      if (TAG) __hip_atomic_store(ctl + GP_CTL_C1 + GP_CIDX(a), (c1 + (unsigned)a.T) % (2u * GP_R1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[a.nl - 1].out[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned g1 = gen + 1u;
      __hip_atomic_store(ctl + DP_CTL_GEN, g1 >= (1u << 21) ? 1u : g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#ifndef GP_XR_STEPS
#define GP_XR_STEPS 6
#endif
constexpr int GP_XR = GP_XR_STEPS;   // (a slot is re-armed, acknowledged, two steps after it was summed: four steps of slack)

template <int NT>
struct GpLdsB {
  // (small, hot arrays first: a DS immediate offset reaches 64 KB; see DpTrailLds)
  unsigned cnt_p[GP_NR], cnt_h[GP_NR], cnt_m[GP_NR], cnt_g[GP_NR], cnt_z[GP_NR], cnt_f[GP_NR], cnt_q[GP_NR], cnt_d[GP_NR], dead, pad_[3];
  float peep[4 * NT][4];                    // {w_i, w_f, w_o, -} per cell
  float car[4][2][GP_NR][64][2];            // c(t) and the carried dc of an R wave's cells [wave][gate tile w | 4 + w][row tile][lane] (in registers they get spilled, and a scratch reload waits for the wave's write-through stores)
  float gs[2][GP_NR][2][64][4];             // the two reducing G waves' sums [step parity][tile][wave][half wave, fragment lane]
  float st[4][GP_ROWS][4 * NT];             // dz(t) in stash order [gate][row][cell]
  float pfs[GP_NR][5][16][4 * NT];          // the stash of the NEXT step of a tile: gate activations i, j, f, o and c(t-1) (the X waves fetch it a step ahead)
  float wpb[GP_NKB][16][4];                 // ... of the fifth gate tile: [k-block][4 (k quarter) + cell - 16][4 k-steps]; A row 4 q = cell 16 + q, the other rows of that product are never read (every lane of a row group reads the same entry)
  float dzB[GP_NR][NT][64][4];              // dz(t) as B fragments of both gradient products [row tile][gate tile][lane (cell q, row lr)][gate]
  float pd[4][GP_NR][NT][64];               // dh partial sums [R wave = k-blocks w, w + 4, ..][row tile][gate tile][lane]
  float wpa[GP_NKB][64][4];                 // W_p fragments, A operand of dh^T = W_p . dm^T: [k-block of P][lane][4 k-steps]; row 4 q + i = cell 4 i + q (so that accumulator register i of lane (q, lr) is gate tile i's cell q)
  float mB[GP_NR][GP_NKB][64][4];           // dm(t) as B fragments [row tile][k-block][lane][4]
  float kl[8][NT][64][4];                   // weights that do not fit the register file [slot][gate tile][lane][gate]: slots 0..3 R wave w's output tile 12 + w, 4, 5 R waves' tiles 16, 17, 6, 7 X waves' tiles 16, 17
};

// one lane = one sentinel piece (16 bytes at `so` when son): wait until every one is valid
__device__ __forceinline__ bool gp_poll(const GpBuf& b, unsigned so, bool son, gu32* err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned polls = 0;; ++polls) {
    const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(b.rs, so, 0, GP_SC1 | GP_VOL);
    if (__all(!son || gp_valid(y))) return true;
    asm volatile("" ::: "memory");
    if ((polls & 63) == 63) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

#ifndef GP_RES_NB
#define GP_RES_NB 10
#endif
// One wave sums the half chunks of ALL NC producers at base + p * GP_SLOT (off the critical path: a plain loop, four loads in flight --
// the unrolled gp_sweep over 20 pieces cost the backward kernel 8 spilled registers): even producers in lanes 0..31, odd ones in
// 32..63, then the two halves (even first); every lane ends with the total of its (lane & 31) piece.  false: time-out / peer failure.
// NB loads in flight; POLL false: the pieces are there as a rule (the layer above is a diagonal ahead): read first, repeat a batch that is not complete.
template <bool TAGGED = false, int NB = 4, bool POLL = true>
__device__ __forceinline__ bool gp_sum_all(const GpBuf& b, unsigned base, int NC, int lane, gu32* err, f32x4& out, unsigned tag = 0u) {
  auto valid = [&](const u32x4& x) { return TAGGED ? gp_valid_t(x, tag) : gp_valid(x); };
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (!POLL) {}
  else if (!TAGGED) { if (!gp_poll(b, base + (unsigned)min(lane, NC - 1) * GP_SLOT + 496u, lane < NC, err)) return false; }
  else {
    const unsigned so = base + (unsigned)min(lane, NC - 1) * GP_SLOT + 496u;
    for (unsigned polls = 0;; ++polls) {
      const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(b.rs, so, 0, GP_SC1 | GP_VOL);
      if (__all(lane >= NC || valid(y))) break;
      asm volatile("" ::: "memory");
      if ((polls & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return false;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  f32x4 sa = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; 2 * k0 < NC; k0 += NB) {
    for (;;) {
      u32x4 x[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j)
        x[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, base + (unsigned)min(2 * (k0 + j) + (lane >> 5), NC - 1) * GP_SLOT + (unsigned)(lane & 31) * 16u, 0, GP_SC1 | GP_VOL);
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NB; ++j) ok &= (2 * (k0 + j) + (lane >> 5) >= NC) || valid(x[j]);
      if (__all(ok)) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (2 * (k0 + j) + (lane >> 5) < NC) sa += gp_untag<TAGGED>(x[j]);
        break;
      }
      asm volatile("" ::: "memory");
      if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  f32x4 sb;
#pragma unroll
  for (int i = 0; i < 4; ++i) sb[i] = __shfl_xor(sa[i], 32);
  out = lane < 32 ? sa + sb : sb + sa;                                   // (even-producer half first, in both halves of the wave)
  return true;
}

// RES (residual stack, see gp_fwd_body): the gradient of the running sum s_l is what layer l's outputs AND, unchanged, the layer
// below receive: D_l(t) = D_{l+1}(t) + dz_{l+1}(t) . K_x^T, D_top = d(outputs of the stack).  The reducer of a half chunk owns that half
// chunk of D: its second G wave sums ALL input-gradient partials of the layer above (alone: the sum is needed on its own), adds the
// D_{l+1}(t) piece the same-numbered reducer above published a step ago, hands D_l(t) into the state-gradient sum and publishes it
// for the layer below in the second region of gran2 (one slot per step, behind the hand-offs).
template <int NT, int PROG, bool RES, bool TAG>
__device__ __forceinline__ void gp_bwd_body(const GPersistArgs& a, GpLdsB<NT>& S, const unsigned c1, const unsigned c3, const unsigned bid) {
  static_assert(!(TAG && PROG), "the progressive sweeps know the sentinel form only");
  constexpr int NR = GP_NR, CW = 4 * NT;
  GPT_DECL
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GP_TRACE
  if (tid == 0) gp_tr[0][23] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  // (a.ngl row groups starting at a.grp0, or all of them: a stack whose workgroups do not fit the device at once runs one launch per row group)
  const int ngr = a.N / GP_ROWS, ngl = a.ngl ? a.ngl : ngr, xpg = 8 / ngl;
  const int xcd = bid & 7, slot = bid >> 3;
  const int grp = a.grp0 + xcd / xpg, idx = slot * xpg + (xcd % xpg);
  if (idx >= a.nl * a.NC) return;
  const int l = idx / a.NC, c = idx - l * a.NC;
  const GPersistLayer L = a.L[l];
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, P = L.P, ldP = L.ldP, I = L.I, NC = a.NC;
  const int nkb = (P + 15) >> 4, nkbx = (I + 15) >> 4;
  const int row0 = grp * GP_ROWS, cell0 = c * CW;
  const int nrt = a.nrt ? a.nrt : NR;                                 // live row tiles (GPersistArgs::nrt)
  const bool top = l == a.nl - 1;
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  const size_t g1_per = (size_t)GP_NCH * NC * GP_SLOT, g2_per = (size_t)GP_NCH * GP_SLOT;
  const GpBuf b1 = gp_buf((const char*)a.gran1 + (size_t)(grp * a.nl + l) * GP_R1 * g1_per, GP_R1 * g1_per);
  const GpBuf b2 = gp_buf((const char*)a.gran2 + (size_t)(grp * a.nl + l) * T * g2_per, (size_t)T * g2_per);
  const GpBuf b2x = gp_buf((const char*)a.gran2 + (size_t)(grp * a.nl + (l > 0 ? l - 1 : 0)) * T * g2_per, (size_t)T * g2_per);
  // gran3 holds nl + 1 rings per row group: ring l + 1 = what layer l + 1 hands to layer l, ring 0 = what layer 0 hands to its OWN reducers
  const GpBuf b3 = gp_buf((const char*)a.gran3 + (size_t)(grp * (a.nl + 1) + l + 1) * GP_XR * g1_per, GP_XR * g1_per);      // what the layer above hands to this one
  const GpBuf b3x = gp_buf((const char*)a.gran3 + (size_t)(grp * (a.nl + 1) + l) * GP_XR * g1_per, GP_XR * g1_per);         // what this layer hands down
  // (RES) the gradients of the running sums, second region of gran2: this layer's D_l for the layer below, D_{l+1} from the layer above
  const char* const g2d = (const char*)a.gran2 + (size_t)ngr * a.nl * T * g2_per;
  const GpBuf b2d = gp_buf(g2d + (size_t)(grp * a.nl + l) * T * g2_per, (size_t)T * g2_per);
  const GpBuf b2du = gp_buf(g2d + (size_t)(grp * a.nl + (top ? l : l + 1)) * T * g2_per, (size_t)T * g2_per);
  const GpBuf btop = gp_buf(a.dout_top, (size_t)T * N * a.ld_dout * sizeof(float));      // (dout_trail: the top layer's polled reads)
  // Layer 0's input gradient dz_0 . K_x^T (the input FC's d(h0): until round 4 a time-batched GEMM behind the launch, 126 us): its X
  // waves -- idle otherwise -- run the same product as every other layer's and publish it to a ring of their own (ring 0); the
  // layer's reducers, which spend most of a step waiting for the layers above, sum it one step late and write it to din0.
  const bool dx0 = l == 0 && a.din0 != nullptr;
  const bool dxr = dx0 && c < 2 * nkbx;                                  // this workgroup reduces a half chunk of it
  const unsigned frag_off = (unsigned)lane * 16u;
  const unsigned pair_off = (unsigned)((lane >> 5) * GP_SLOT + (lane & 31) * 16);
  auto slot1 = [&](int par, int r, int jb, int p) { return (unsigned)((((size_t)(par * NR + r) * GP_NKB + jb) * NC + p) * GP_SLOT); };   // (both rings)
  auto slot2 = [&](int t, int r, int jb) { return (unsigned)((((size_t)t * NR + r) * GP_NKB + jb) * GP_SLOT); };
  const bool reducer = c < 2 * nkb;
  const int jbr = c >> 1, hh = c & 1;

  // ---- cooperative prologue ----
  for (int e = tid; e < GP_NKB * 80; e += GP_WAVES * 64) {
    const int jb = e / 80, ln = e - jb * 80, at = ln >= 64;
    const int m = ln & 15, kq = at ? (ln - 64) >> 2 : ln >> 4;
    const int cell = cell0 + (at ? 16 + (ln & 3) : 4 * (m & 3) + (m >> 2));
    const bool cok = cell < H && cell < cell0 + CW;
    const float* wr = L.Wp + (size_t)min(cell, H - 1) * ldP;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int col = 16 * jb + 4 * kq + u;
      v[u] = (cok && col < P) ? wr[min(col, P - 1)] : 0.f;
    }
    if (at) *reinterpret_cast<float4*>(&S.wpb[jb][ln - 64][0]) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float4*>(&S.wpa[jb][ln][0]) = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int e = tid; e < 3 * CW; e += GP_WAVES * 64) {
    const int k = e / CW, cl = e - k * CW, cell = min(cell0 + cl, H - 1);
    S.peep[cl][k] = (k == 0 ? L.wi : k == 1 ? L.wf : L.wo)[cell];
  }
  if (tid < 20) (&S.cnt_p[0])[tid] = 0u;                                // (all counters, dead)
  __syncthreads();
  // every counter through ONE base register + a compile-time offset (hipcc otherwise keeps a dozen LDS addresses in registers
  // across the step loop; the zero is opaque to it)
  unsigned zed;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zed));
  unsigned* const cnt = &S.cnt_p[0] + zed;
  constexpr int C_P = 0, C_H = GP_NR, C_M = 2 * GP_NR, C_G = 3 * GP_NR, C_Z = 4 * GP_NR, C_F = 5 * GP_NR, C_Q = 6 * GP_NR, C_D = 7 * GP_NR, C_DEAD = 8 * GP_NR;
  const unsigned* dead = cnt + C_DEAD;
  auto fail = [&]() {
    if (lane == 0) {
      __hip_atomic_store(cnt + C_DEAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(err, 1u + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  const int len0 = a.len[row0 + lr], len1 = a.len[row0 + 16 + lr];

  if (w < 8) {
    // R waves (w < 4): K_h, the state-gradient product, dh and the cell.  X waves: K_x, the input-gradient product for the layer below.
    // Resident weights: output tile pt = ww + 4 jj of the product's P (or I) columns, gate tile i:
    //   A[row lr = column 16 pt + lr][k = cell 4 i + q] for the four k-steps gate 0..3
    const bool isx = w >= 4;
    const int ww = w & 3;
    const bool noprod = isx && l == 0 && !dx0;                           // (layer 0 without a consumer of its input gradient)
    const float* KT = isx ? L.KxT : L.KhT;
    const int ldK = isx ? L.ldI : ldP, PW = isx ? I : P, nkw = isx ? nkbx : nkb;
    float4 kw[GP_KBW - 1][NT];
#pragma unroll
    for (int jj = 0; jj < GP_KBW; ++jj) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int col = 16 * (ww + 4 * jj) + lr, cell = cell0 + 4 * i + q;
        const float* kr = KT + (size_t)min(cell, H - 1) * ldK + min(col, ldK - 1);
        const size_t gs_ = (size_t)H * ldK;
        float4 v = make_float4(kr[0], kr[gs_], kr[2 * gs_], kr[3 * gs_]);
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        const bool ok = col < PW && cell < H && !noprod;
        const float4 f = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        if (jj < (isx ? GP_KBW - 1 : GP_KBW - 2)) kw[jj][i] = f;
        else if (jj == GP_KBW - 2) *reinterpret_cast<float4*>(&S.kl[ww][i][lane][0]) = f;
        else if (ww < 2) *reinterpret_cast<float4*>(&S.kl[(isx ? 6 : 4) + ww][i][lane][0]) = f;
      }
    }
    const float* const k4w = &S.kl[(isx ? 6 : 4) + (ww & 1)][0][lane][0];   // + i * 256
    const float* const k3w = &S.kl[ww][0][lane][0];                     // + i * 256   (R waves)
    const bool five = ww + 16 < nkw;                                    // this wave owns a fifth output tile
    const float* const dzr = &S.dzB[0][0][lane][0];                     // + (r * NT + i) * 256
    const GpBuf& bpub = isx ? b3x : b1;
    // the product of one row tile and its publication: three output tiles in flight, a tile leaves as soon as its 4 NT products are done
    auto product = [&](int r, int ring, unsigned tag) {
      // (opaque copies: hipcc otherwise computes every slot offset and row address of the step loop once, in front of it, and
      // spills them -- a scratch reload behind the write-through stores below waits for their acknowledgement)
      const unsigned ln_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // (the lane id again: two VALU instructions, no register kept, no spill)
      const unsigned fo = ln_ << 4;
      if (isx) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int j0 = 0; j0 < GP_KBW; j0 += 3) {
        if (j0 + 3 >= GP_KBW && ww + 4 * j0 >= nkw) break;              // (uniform)
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#ifdef GP_ABL2
          if ((GP_ABL2 & 8) && isx) continue;                           // timing ablation: no input-gradient products (zeros are published)
#endif
          const float4 b = *reinterpret_cast<const float4*>(dzr + (r * NT + i) * 256);
          float4 ka[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int jj = j0 + j;
            if (jj < GP_KBW - 2 || (jj == GP_KBW - 2 && isx)) ka[j] = kw[jj < GP_KBW - 1 ? jj : 0][i];
            else if (jj == GP_KBW - 2) ka[j] = *reinterpret_cast<const float4*>(k3w + i * 256);
            else if (jj == GP_KBW - 1) ka[j] = *reinterpret_cast<const float4*>(k4w + i * 256);
            else ka[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < GP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j].x, b.x, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < GP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j].y, b.y, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < GP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j].z, b.z, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < GP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j].w, b.w, acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int jj = j0 + j;
          if (jj < GP_KBW && ww + 4 * jj < nkw && (jj < GP_KBW - 1 || five)) { if (TAG) gp_store_t(bpub, slot1(ring, r, ww + 4 * jj, c) + fo, acc[j], tag); else gp_store(bpub, slot1(ring, r, ww + 4 * jj, c) + fo, acc[j]); }
        }
      }
      __builtin_amdgcn_s_setprio(0);
    };

    if (isx) {
      // =============================== X waves ===============================
      const int nsx = (nkbx - ww + 3) >> 2;                             // this wave's output tiles: ww, ww + 4, ...
      // The stash of step t-1 (gate activations, c(t-2): 16 rows x 4 NT cells x 5 values per tile) travels through these waves into
      // LDS a step ahead, as 16-byte row pieces (the mirror of the R waves' stash store): the R waves keep 80 weight registers and
      // cannot hold it, and a load issued in front of the publication below does not wait for a write-through acknowledgement.
      // (two named values, not an array: indexed inside the lambdas the pair lived in SCRATCH -- the load waited for at once, stored,
      //  reloaded a step later, every access a memory round trip beside the hand-off traffic)
      float4 pv0, pv1;
      auto fetch1 = [&](int t, int r, int it, int ln) -> float4 {
        const int e = it * 256 + ww * 64 + ln;
        const int cq = e % NT, pr = min(e / NT, 79), row = pr & 15, k = pr >> 4;
        const size_t rowg = (size_t)t * N + row0 + 16 * r + row;
        const int cell = min(cell0 + 4 * cq, H - 4);
        return *reinterpret_cast<const float4*>((k < 4 ? L.gates + rowg * H4 + k * H : L.c + rowg * H) + cell);
      };
      auto fetch = [&](int t, int r) {
#ifdef GP_ABL2
        if (GP_ABL2 & 32) { pv0 = pv1 = make_float4(0.3f, 0.3f, 0.3f, 0.3f); return; }      // timing ablation: no stash prefetch
#endif
        int ln = lane;
        asm volatile("" : "+v"(ln));
        pv0 = fetch1(t, r, 0, ln);
        pv1 = fetch1(t, r, 1, ln);
      };
      auto stage1 = [&](int r, int it, int ln, const float4& v) {
        const int e = it * 256 + ww * 64 + ln;
        const int cq = e % NT, pr = e / NT, row = pr & 15, k = min(pr >> 4, 4);
        if (e < 5 * 16 * NT) *reinterpret_cast<float4*>(&S.pfs[r][k][row][4 * cq]) = v;
      };
      auto stage = [&](int r) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        stage1(r, 0, ln, pv0);
        stage1(r, 1, ln, pv1);
        gp_signal(cnt + C_F + r, lane);
      };
#pragma unroll
      for (int r = 0; r < NR; ++r) { if (r < nrt) { fetch(T - 1, r); stage(r); } }
      for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (r >= nrt) continue;
          GPTSR(18 + 2 * r);
          // back-pressure: ring slot s % GP_XR was summed AND re-armed by the layer below when it has published its dm(t + GP_XR - 2)
          // (its R waves re-arm a slot at the end of the step that summed it, acknowledged before the next step's partials leave,
          // which the dm of the step after that waits for)
          // (layer 0's own ring: summed a step late, re-armed a step after that: the dm of three steps ago; b2x is this layer's then)
          if (!noprod && s >= GP_XR - (l == 0 ? 3 : 2) && !gp_poll(b2x, slot2(t + GP_XR - (l == 0 ? 3 : 2), r, min(ww + 4 * (lane >> 1), nkbx - 1)) + (unsigned)(lane & 1) * 512u + 496u, lane < 2 * nsx, err)) { fail(); return; }
          if (!gp_wait(cnt + C_H + r, 4u * ((unsigned)s + 1u), dead)) return;       // dz(t) of the tile is in LDS (and the cells have read the stage)
          GPTSR(19 + 2 * r);
          if (t > 0) { fetch(t - 1, r); stage(r); }                       // (needed by the cells a whole hand-off from now)
          if (!noprod) {
            // The R waves' state-gradient product starts at the same moment on the same SIMDs and is on the recurrence's critical
            // path; this one only feeds the layer below, which runs a ring's worth of steps behind: let theirs go first (side by
            // side both took 6-7.6 k cycles for 3.2 k of MFMA issue each, s_setprio notwithstanding).
            if (t > 0 && !gp_wait(cnt + C_Q + r, 4u * ((unsigned)s + 1u), dead)) return;
            product(r, (int)((c3 + (unsigned)s) % GP_XR), ((c3 + (unsigned)s) / GP_XR) & 1u);
            gp_signal(cnt + C_Z + r, lane);                                 // (the product has read dzB)
          }
        }
      }
#ifdef GP_TRACE
      if (ww == 0) { const int i0_ = 18, i1_ = 22; GPT_FLUSH(); }
#endif
      return;
    }

    // =============================== R waves ===============================
    const bool two = w + 4 < NT;                                         // this wave owns a second gate tile
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
        *reinterpret_cast<float2*>(&S.car[w][sl][r][lane][0]) = make_float2(L.c[((size_t)T * N + row0 + 16 * r + lr) * H + min(cell0 + 4 * (w + 4 * sl) + q, H - 1)], 0.f);
    float* const carw = &S.car[w][0][0][lane][0];                        // + (sl * NR + r) * 128
    const float* const pfc = &S.pfs[0][0][lr][4 * w + q];                // + ((r * 5 + k) * 16) * CW + sl * 16
    const float* const mbw = &S.mB[0][w][lane][0];                       // + (r * GP_NKB + 4 jj) * 256
    const float* const wpw = &S.wpa[w][lane][0];                         // + 4 jj * 256
    const float* const wpv = &S.wpb[w][4 * q + (lr >> 2)][0];            // + 4 jj * 64
    float* const pdw = &S.pd[w][0][0][lane];                             // + (r * NT + i) * 64
    const float* const pdc = &S.pd[0][0][w][lane];                       // + (k * NR * NT + r * NT + 4 sl) * 64
    const float* const pwc = &S.peep[4 * w + q][0];                      // + sl * 64
    float* const dzw = &S.dzB[0][w][lane][0];                            // + (r * NT + 4 sl) * 256
    float* const stc = &S.st[0][lr][4 * w + q];                          // + g * GP_ROWS * CW + r * 16 * CW + sl * 16
    for (int s = 0; s < T; ++s) {
      const int t = T - 1 - s;
#ifdef GP_TRACE
      if (s == 0 && w == 0 && lane == 0) gp_tr[0][22] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r >= nrt) continue;
        GPTSR(6 * r + 0);
        if (!gp_wait(cnt + C_M + r, 2u * ((unsigned)s + 1u), dead)) return;      // dm(t) of the tile is in LDS
        GPTSR(6 * r + 1);

        {
          // dh^T[cells][rows] = W_p . dm^T, this wave's k-blocks w, w + 4, ...
          f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
          __builtin_amdgcn_s_setprio(2);
#pragma unroll
          for (int jj = 0; jj < GP_KBW; ++jj) {
            if (w + 4 * jj < nkb) {
              const float4 b = *reinterpret_cast<const float4*>(mbw + (r * GP_NKB + 4 * jj) * 256);
              const float4 a0 = *reinterpret_cast<const float4*>(wpw + 4 * jj * 256), a1 = *reinterpret_cast<const float4*>(wpv + 4 * jj * 64);
              d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, d1, 0, 0, 0);
              d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, d1, 0, 0, 0);
              d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, d1, 0, 0, 0);
              d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, d1, 0, 0, 0);
            }
          }
          __builtin_amdgcn_s_setprio(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) pdw[(r * NT + i) * 64] = d0[i];
          if (NT > 4) pdw[(r * NT + 4) * 64] = d1[0];
        }
        GPTSR(6 * r + 2);
        gp_signal(cnt + C_P + r, lane);
        if (!gp_wait(cnt + C_P + r, 4u * ((unsigned)s + 1u), dead)) return;
        if (l > 0 && s > 0 && !gp_wait(cnt + C_Z + r, 4u * (unsigned)s, dead)) return;     // the X waves have taken dz(t+1)
        if (!gp_wait(cnt + C_F + r, 4u * ((unsigned)s + 1u), dead)) return;                // the stash of step t is in LDS
        GPTSR(6 * r + 3);
        const bool live = t < (r ? len1 : len0);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          if (sl == 0 || two) {
            const float dh = ((pdc[(0 * NR * NT + r * NT + 4 * sl) * 64] + pdc[(1 * NR * NT + r * NT + 4 * sl) * 64]) + pdc[(2 * NR * NT + r * NT + 4 * sl) * 64]) + pdc[(3 * NR * NT + r * NT + 4 * sl) * 64];
            const f32x4 pw = *reinterpret_cast<const f32x4*>(pwc + sl * 64);
            const float* const pq = pfc + r * 5 * 16 * CW + sl * 16;
            const float gi = pq[0], gj = pq[16 * CW], gf = pq[2 * 16 * CW], go = pq[3 * 16 * CW], cp = pq[4 * 16 * CW];
            const float2 cv = *reinterpret_cast<const float2*>(carw + (sl * NR + r) * 128);
            const float cc = cv.x, dcv = cv.y;
            const float tc = gp_tanh(cc);
            const float dao = dh * tc * go * (1.f - go);
            const float dcn = dcv + dh * go * (1.f - tc * tc) + dao * pw[2];
            const float daf = dcn * cp * gf * (1.f - gf);
            const float dai = dcn * gj * gi * (1.f - gi);
            const float dj = dcn * gi * (1.f - gj * gj);
            const float dcx = live ? dcn * gf + dai * pw[0] + daf * pw[1] : dcv;
            *reinterpret_cast<float2*>(carw + (sl * NR + r) * 128) = make_float2(cp, dcx);
            const f32x4 dz = {live ? dai : 0.f, live ? dj : 0.f, live ? daf : 0.f, live ? dao : 0.f};
            *reinterpret_cast<f32x4*>(dzw + (r * NT + 4 * sl) * 256) = dz;
            float* const d = stc + r * 16 * CW + sl * 16;
            d[0 * GP_ROWS * CW] = dz[0]; d[1 * GP_ROWS * CW] = dz[1]; d[2 * GP_ROWS * CW] = dz[2]; d[3 * GP_ROWS * CW] = dz[3];
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the re-arming stores of the step before)
        gp_signal(cnt + C_H + r, lane);
        GPTSR(6 * r + 4);
        if (!gp_wait(cnt + C_H + r, 4u * ((unsigned)s + 1u), dead)) return;       // every cell's dz of the tile is in LDS
        if (t > 0) { product(r, (int)((c1 + (unsigned)s) % GP_R1), ((c1 + (unsigned)s) / GP_R1) & 1u); gp_signal(cnt + C_Q + r, lane); }     // (dm(-1) has no consumer)
        GPTSR(6 * r + 5);
        // dz(t) over the gate activations of the stash, a quarter of the tile per R wave: NT consecutive lanes write one 16 NT-byte row piece
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int it = 0; it < (4 * 16 * NT + 255) / 256; ++it) {
          const int e = it * 256 + w * 64 + ln;
          const int cq = e % NT, pr = e / NT, row = 16 * r + (pr & 15), k = min(pr >> 4, 3);
          const float4 v = *reinterpret_cast<const float4*>(&S.st[k][row][4 * cq]);
          float* dst = L.gates + ((size_t)t * N + row0 + row) * H4 + k * H + cell0 + 4 * cq;
#ifdef GP_ABL2
          if (GP_ABL2 & 16) continue;                                    // timing ablation: no dz stores
#endif
          if (e < 4 * 16 * NT && cell0 + 4 * cq < H) gp_stash_store(dst, v);
        }
        // This workgroup's G waves summed this step's partials before they gathered dm(t) (the wait at the top): re-arm their slots of
        // both rings.  Acknowledged before the wave signals its cells of the next step, i.e. before this workgroup's partials of that
        // step leave, without which no dm of the step after -- and no later write to these slots -- exists.
        if (!TAG && reducer) {
          if (s > 0) gp_rearm(b1, slot1((s - 1) % GP_R1, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
          if (!top) gp_rearm(b3, slot1(s % GP_XR, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
          if (dxr && s > 0) {                                              // (the G wave summed layer 0's input gradient of step s-1 long ago)
            if (!gp_wait(cnt + C_D + r, (unsigned)s, dead)) return;
            gp_rearm(b3x, slot1((s - 1) % GP_XR, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
          }
        }
      }
    }
    if (!TAG && dxr) {                                                     // the last step's slots: leave the ring armed for the next launch
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r >= nrt) continue;
        if (!gp_wait(cnt + C_D + r, (unsigned)T, dead)) return;
        gp_rearm(b3x, slot1((T - 1) % GP_XR, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
      }
    }
#ifdef GP_TRACE
    if (w == 0) { { const int i0_ = 0, i1_ = 12; GPT_FLUSH(); } { const int i0_ = 22, i1_ = 24; GPT_FLUSH(); } }
#endif
    return;
  }

  // =============================== G waves: reduce, publish, gather (tile gw & 1) ===============================
  __builtin_amdgcn_s_setprio(3);
  const int gw = w - 8, r = gw & 1, gp = gw >> 1;
  float* const mbg = &S.mB[r][gp][lane][0];                            // + 2 n * 256
  const int nvg = (nkb - gp + 1) >> 1;                                 // this wave's k-blocks: gp, gp + 2, ...
  const int ppw = (NC + 1) >> 1, pp0 = gp * ppw, pn = max(0, min(ppw, NC - pp0));
  const int nlr = (pn + 1 - (lane >> 5)) >> 1;                           // (two producers per load: lanes 0..31 the even ones, 32..63 the odd ones)
  const int rrow = row0 + 16 * r + ((32 * hh + (lane & 31)) & 15), rcol = 16 * jbr + 4 * ((32 * hh + (lane & 31)) >> 4);
  const int rlen = a.len[rrow];
  // layer 0's input gradient of step sx (time T-1-sx): all NC partials of this half chunk, tile r, by ONE wave (gp == 0; off the
  // critical path, no exchange with the other G wave), even producers in lanes 0..31, odd ones in 32..63, then the two halves
  auto sum_dx0 = [&](int sx) -> bool {
    f32x4 tt;
    if (!gp_sum_all<TAG>(b3x, slot1((int)((c3 + (unsigned)sx) % GP_XR), r, jbr, 0) + (unsigned)hh * 512u, NC, lane, err, tt, ((c3 + (unsigned)sx) / GP_XR) & 1u)) return false;
    if (lane < 32 && rcol < a.ld_din0)
      *reinterpret_cast<float4*>(a.din0 + ((size_t)(T - 1 - sx) * N + rrow) * a.ld_din0 + rcol) = make_float4(tt[0], tt[1], tt[2], tt[3]);
    gp_signal(cnt + C_D + r, lane);
    return true;
  };
  if (r >= nrt) return;                                                // (a tile of padding rows: no gradient to sum, publish or gather)
  for (int s = 0; s < T; ++s) {
    const int t = T - 1 - s;
    // ring positions of this step (c1 = c3 = 0 without tags): what the layer above wrote at step s, what this layer wrote at step s - 1
    const int rx = (int)((c3 + (unsigned)s) % GP_XR), r1 = (int)((c1 + (unsigned)s + 2u * GP_R1 - 1u) % GP_R1);
    const unsigned tagx = ((c3 + (unsigned)s) / GP_XR) & 1u, tag1 = ((c1 + (unsigned)s + 2u * GP_R1 - 1u) / GP_R1) & 1u;
    GPTSG(12);
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    f32x4 Dl = {0.f, 0.f, 0.f, 0.f};                                     // (RES, second G wave) D_l(t) of this lane's four columns
    if (reducer) {
      float4 dtop = make_float4(0.f, 0.f, 0.f, 0.f);
      if (top && rcol < a.ld_dout) {
        if (a.dout_trail == 1) {
          // the discriminator's trailing BPTT (dpersist.hip k_dlstm_bwd_trail) is writing d(outputs) while this launch runs: this
          // lane's piece of step t, past the caches, until none of its words carries the armed pattern (it is there as a rule: that
          // launch runs three times as fast as this one)
          const unsigned off = (unsigned)((((size_t)t * N + rrow) * a.ld_dout + rcol) * sizeof(float));
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          u32x4 y;
          for (unsigned polls = 0;; ++polls) {
            y = __builtin_amdgcn_raw_buffer_load_b128(btop.rs, off, 0, GP_SC1 | GP_VOL);
            if (__all(gp_valid(y))) break;
            asm volatile("" ::: "memory");
            if ((polls & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { fail(); return; }
            __builtin_amdgcn_s_sleep(1);
          }
          dtop = make_float4(__uint_as_float(y[0]), __uint_as_float(y[1]), __uint_as_float(y[2]), __uint_as_float(y[3]));
        } else
          dtop = *reinterpret_cast<const float4*>(a.dout_top + ((size_t)t * N + rrow) * a.ld_dout + rcol);
        // (columns between P and ld_dout of a shared gradient buffer are whatever its last user left there: 0 x NaN would poison dh)
        dtop = make_float4(rcol < P ? dtop.x : 0.f, rcol + 1 < P ? dtop.y : 0.f, rcol + 2 < P ? dtop.z : 0.f, rcol + 3 < P ? dtop.w : 0.f);
      }
      f32x4 sa = {0.f, 0.f, 0.f, 0.f};
      if (RES) {
        if (gp == 1) {
          if (top) Dl = f32x4{dtop.x, dtop.y, dtop.z, dtop.w};
          else {
            f32x4 dxt;
            const unsigned off = slot2(t, r, jbr) + (unsigned)hh * 512u + (unsigned)(lane & 31) * 16u;
            u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(b2du.rs, off, 0, GP_SC1 | GP_VOL);      // (D_{l+1}(t): in flight beside the partials; polled below if it was not there yet)
            if (!gp_sum_all<TAG, GP_RES_NB, false>(b3, slot1(rx, r, jbr, 0) + (unsigned)hh * 512u, NC, lane, err, dxt, tagx)) { fail(); return; }      // (on the reducer's path: ten loads in flight, the same order of the sum)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned polls = 0;; ++polls) {
              if (__all(gp_valid(y))) break;
              y = __builtin_amdgcn_raw_buffer_load_b128(b2du.rs, off, 0, GP_SC1 | GP_VOL);
              if (__all(gp_valid(y))) break;
              asm volatile("" ::: "memory");
              if ((polls & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { fail(); return; }
              __builtin_amdgcn_s_sleep(1);
            }
            Dl = dxt + f32x4{__uint_as_float(y[0]), __uint_as_float(y[1]), __uint_as_float(y[2]), __uint_as_float(y[3])};
          }
        }
        dtop = make_float4(0.f, 0.f, 0.f, 0.f);                          // (it travels inside D_l)
      } else if (!top) {
        // the input-gradient partials of the layer above at time t (published a diagonal ago as a rule: read first, poll if not there)
        unsigned lo[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) lo[k] = slot1(rx, r, jbr, min(pp0 + 2 * k, NC - 1)) + (unsigned)hh * 512u;
        if (!gp_sweep<10, false, 10, false, TAG>(b3, lo, nlr, pair_off, slot1(rx, r, jbr, min(pp0 + lane, NC - 1)) + (unsigned)hh * 512u + 496u, lane < pn, err,
                                     [&](int k, const f32x4& v) { if (k == 0) sa = k < nlr ? v : f32x4{0.f, 0.f, 0.f, 0.f}; else if (k < nlr) sa += v; }, tagx)) { fail(); return; }
      }
      GPTSG(13);
      if (s > 0) {
        // the state-gradient partials of this layer from time t + 1
        unsigned lo[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) lo[k] = slot1(r1, r, jbr, min(pp0 + 2 * k, NC - 1)) + (unsigned)hh * 512u;
        f32x4 ua = {0.f, 0.f, 0.f, 0.f};
        if (PROG & 1) {
          f32x4 pv[10];
          if (!gp_sweep_prog<10, GP_HB10>(b1, lo, nlr, (pn + 1) >> 1, pair_off, slot1((s + GP_R1 - 1) % GP_R1, r, jbr, min(pp0 + lane, NC - 1)) + (unsigned)hh * 512u + 496u, lane < pn,
                                    (pn * GP_GATE_NUM) / GP_GATE_DEN, err, pv)) { fail(); return; }
#pragma unroll
          for (int k = 0; k < 10; ++k) { if (k == 0) ua = k < nlr ? pv[0] : f32x4{0.f, 0.f, 0.f, 0.f}; else if (k < nlr) ua += pv[k]; }
        } else if (!gp_sweep<10, true, 10, false, TAG>(b1, lo, nlr, pair_off, slot1(r1, r, jbr, min(pp0 + lane, NC - 1)) + (unsigned)hh * 512u + 496u, lane < pn, err,
                                    [&](int k, const f32x4& v) { if (k == 0) ua = k < nlr ? v : f32x4{0.f, 0.f, 0.f, 0.f}; else if (k < nlr) ua += v; }, tag1)) { fail(); return; }
        sa += ua;
      }
      if (RES && gp == 1 && lane < 32) sa += Dl;                          // (once: with this wave's even-producer half)
      GPTSG(14);
      *reinterpret_cast<f32x4*>(&S.gs[s & 1][r][gp][lane][0]) = sa;
      gp_signal(cnt + C_G + r, lane);
      if (!gp_wait(cnt + C_G + r, 2u * ((unsigned)s + 1u), dead)) return;
      const float* const gsr = &S.gs[s & 1][r][0][lane & 31][0];
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(gsr), a1 = *reinterpret_cast<const f32x4*>(gsr + 32 * 4);
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(gsr + 64 * 4), c1 = *reinterpret_cast<const f32x4*>(gsr + 96 * 4);
      const bool live = t < rlen;
      tot = (((a0 + a1) + c0) + c1) + f32x4{dtop.x, dtop.y, dtop.z, dtop.w};
      if (!live) tot = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gp == 0 && lane < 32) gp_store(b2, slot2(t, r, jbr) + (unsigned)hh * 512u + (unsigned)lane * 16u, tot);
      GPTSG(15);
    }
    {
      // gather dm(t) of the tile: k-blocks gp, gp + 2, ... as B fragments of dh = dm . W_p^T
      unsigned lo[9];
#pragma unroll
      for (int n = 0; n < 9; ++n) lo[n] = slot2(t, r, min(gp + 2 * n, nkb - 1));
      f32x4 mv[9];
      if (PROG & 2) {
        if (!gp_sweep_prog<9, GP_HB9>(b2, lo, nvg, nvg, frag_off, slot2(t, r, min(gp + 2 * (lane >> 1), nkb - 1)) + (unsigned)(lane & 1) * 512u + 496u,
                                 lane < 2 * nvg, (2 * nvg * GP_GATE_NUM) / GP_GATE_DEN, err, mv)) { fail(); return; }
      } else if (!gp_sweep<9, true, 9, true>(b2, lo, nvg, frag_off, slot2(t, r, min(gp + 2 * (lane >> 1), nkb - 1)) + (unsigned)(lane & 1) * 512u + 496u,
                                 lane < 2 * nvg, err, [&](int k, const f32x4& v) { mv[k] = v; })) { fail(); return; }
      GPTSG(16);
#pragma unroll
      for (int n = 0; n < 9; ++n)
        if (n < nvg) *reinterpret_cast<f32x4*>(mbg + 2 * n * 256) = mv[n];
      gp_signal(cnt + C_M + r, lane);
      GPTSG(17);
    }
    if (reducer && gp == 1 && lane < 32 && rcol < ldP) *reinterpret_cast<float4*>(L.dmt + ((size_t)t * N + rrow) * ldP + rcol) = make_float4(tot[0], tot[1], tot[2], tot[3]);
    if (RES && reducer && gp == 1 && l > 0 && lane < 32) gp_store(b2d, slot2(t, r, jbr) + (unsigned)hh * 512u + (unsigned)lane * 16u, Dl);   // (the layer below is a step behind)
    if (dxr && gp == 0 && s > 0) { if (!sum_dx0(s - 1)) { fail(); return; } }
  }
  if (dxr && gp == 0 && !sum_dx0(T - 1)) { fail(); return; }
#ifdef GP_TRACE
  if (gw == 0) { const int i0_ = 12, i1_ = 18; GPT_FLUSH(); }
#endif
}

template <int NT, int PROG, bool RES, bool TAG>
__global__ __launch_bounds__(GP_WAVES * 64, 3) void k_glstm_bwd(const GPersistArgs a) {
  __shared__ __attribute__((aligned(16))) GpLdsB<NT> S;
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (counts launches)
  // (TAG) the ring step counters: T - 1 state-gradient steps and T input-gradient steps are written per launch
  const unsigned c1 = TAG ? __hip_atomic_load(ctl + GP_CTL_C1 + GP_CIDX(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const unsigned c3 = TAG ? __hip_atomic_load(ctl + GP_CTL_C3 + GP_CIDX(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  gp_bwd_body<NT, PROG, RES, TAG>(a, S, c1, c3, blockIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (TAG) {
        __hip_atomic_store(ctl + GP_CTL_C1 + GP_CIDX(a), (c1 + (unsigned)a.T - 1u) % (2u * GP_R1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctl + GP_CTL_C3 + GP_CIDX(a), (c3 + (unsigned)a.T) % (2u * GP_XR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[0].gates[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned g1 = gen + 1u;
      __hip_atomic_store(ctl + DP_CTL_GEN, g1 >= (1u << 21) ? 1u : g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The generator's BPTT AND the trailing form of the discriminator's (dpersist_dev.h dp_bwdt_body / dp_fcb_body) as ONE launch: the
// first dt_pad workgroups (a multiple of 8: the generator's workgroups keep their XCDs) are the discriminator's, the rest run gp_bwd_body, whose top layer polls the
// gradient the FC workgroups write (GPersistArgs::dout_trail).  Two launches on two streams did the same 0.45 ms faster than one
// after the other -- when their queues let them run side by side: inside a replayed graph, or with other streams of the process
// sharing a hardware queue with either, the generator's launch can be dispatched FIRST and waits a second for a launch that sits
// behind it.  One launch has one dispatch order: the discriminator's workgroups first.
template <int NT, bool RES, bool TAG>
union GpDtLds { GpLdsB<NT> g; DpTrailLds d; };
template <int NT, bool RES, bool TAG>
__global__ __launch_bounds__(GP_WAVES * 64, 3) void k_glstm_bwd_dt(const GPersistArgs a, const DPersistArgs d) {
  __shared__ __attribute__((aligned(16))) GpDtLds<NT, RES, TAG> S;
  const int nD = d.nl * (d.N >> 5) * DP_NQ, nReal = nD + (d.N >> 4), dt_pad = (nReal + 7) & ~7;
  // (a.dout_trail = 1; RSRGAN_TRAIL_DBG sets the timing experiments of DESIGN 6-R5: 2 the discriminator half alone, 3 both halves without
  //  the generator's polls, 4 the generator half alone -- their results are meaningless)
  const int mode = a.dout_trail;
  const int dbid = (int)blockIdx.x;
  const unsigned gbid = blockIdx.x - (unsigned)dt_pad;
  if (dbid < dt_pad) {
    if (dbid >= nReal || mode == 4) return;
    gu32* dctl = (gu32*)d.ctl;
    const unsigned dgen = __hip_atomic_load(dctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) S.d.dead = 0;
    __syncthreads();
    if (dbid < nD) { if (d.nrt == 1) dp_bwdt_body<true>(d, dgen, S.d, dbid); else dp_bwdt_body<false>(d, dgen, S.d, dbid); }
    else dp_fcb_body(d, dgen, S.d, dbid - nD);
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(dctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)nReal - 1u) {
        if (__hip_atomic_load(dctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) d.dy[0] = __builtin_nanf("");
        __hip_atomic_store(dctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dctl + DP_CTL_GEN, dgen + 1u == 0u ? 1u : dgen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  gu32* ctl = (gu32*)a.ctl;
  if (mode == 2) return;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned c1 = TAG ? __hip_atomic_load(ctl + GP_CTL_C1 + GP_CIDX(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const unsigned c3 = TAG ? __hip_atomic_load(ctl + GP_CTL_C3 + GP_CIDX(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  gp_bwd_body<NT, 0, RES, TAG>(a, S.g, c1, c3, gbid);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - (unsigned)dt_pad - 1u) {
      if (TAG) {
        __hip_atomic_store(ctl + GP_CTL_C1 + GP_CIDX(a), (c1 + (unsigned)a.T - 1u) % (2u * GP_R1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctl + GP_CTL_C3 + GP_CIDX(a), (c3 + (unsigned)a.T) % (2u * GP_XR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[0].gates[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned g1 = gen + 1u;
      __hip_atomic_store(ctl + DP_CTL_GEN, g1 >= (1u << 21) ? 1u : g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// =====================================================================================================================================
// UNPROJECTED cells (num_proj=None: the state m IS h; BASELINE.json's "2-layer 512-unit LSTM generator") -- round 5.
// Without a projection there is nothing to reduce: the workgroup that owns 16 cells owns k-block c of the state, so a step has ONE
// hand-off, the all-gather of h(t) (every workgroup publishes its 16 x 16 tile as one chunk, every workgroup of the layer and of the
// layer above gathers all H / 16 of them).  Same decomposition otherwise: workgroup = (row group of 32 rows, layer, slice of 16 cells =
// 64 gate columns = NT 4 gate tiles), two tile lanes, 12 waves -- R: K_h slice (H x 64) resident, recurrent product + cell + stash;
// X: K_x slice, the input product a step ahead; G: publish, gather.  H <= 512 (32 k-blocks: 8 per R / X wave, seven of them in VGPRs,
// the eighth -- the X waves' seventh and eighth -- in LDS), H % 16 == 0.  Cell arithmetic, masking and the stash exactly as gp_fwd_body /
// kernels.hip k_fwd_gates with np_m_out (mst = carried h, out = masked h).
constexpr int NP_NKB = 32, NP_KBW = 8, NP_NCH = NP_NKB * GP_NR;
// NT gate tiles (4 NT cells) per workgroup; KR / KX of an R / X wave's eight k-blocks live in VGPRs, the others in LDS.
//   NT = 4 (16 cells = one whole chunk, H / 16 workgroups per row group and layer), KR = 7, KX = 6: 168 VGPRs and ~55 spilled
//   NT = 2 (8 cells = a half chunk, H / 8 workgroups), KR = KX = 8: every weight in registers, twice the workgroups
template <int NT, int KR, int KX>
struct NpLds {
  float mB[GP_NR][NP_NKB][64][4];           // carried h(t-1) as B fragments [row tile][k-block][lane][4]                 64 KB
  float pb[4][NT][GP_NR][64][4];         // accumulator tiles: x-part (X wave w -> R wave w), then the R waves' partial sums   32 KB
  float st[6][GP_ROWS][4 * NT];          // the step's stash: gates i, j, f, o | c | h                                  12 KB
  float khl[4][NP_KBW - KR > 0 ? NP_KBW - KR : 1][NT][64][4];      // the K_h k-blocks of R wave w that are not in registers
  float kxl[4][NP_KBW - KX > 0 ? NP_KBW - KX : 1][NT][64][4];      // ... K_x, X wave w
  float peep[4 * NT][4], bias[4 * NT][4];
  unsigned cnt_x[GP_NR][4], cnt_p[GP_NR], cnt_h[GP_NR], cnt_m[GP_NR], cnt_s[GP_NR], dead, pad_[15];
};

template <int NT, int KR, int KX>
__device__ __forceinline__ void np_fwd_body(const GPersistArgs& a, NpLds<NT, KR, KX>& S) {
  static_assert(sizeof(NpLds<NT, KR, KX>) <= 160 * 1024, "LDS of the unprojected forward kernel");
  constexpr int NR = GP_NR, NU = NT * NR, CW = 4 * NT;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ngr = a.N / GP_ROWS, xpg = 8 / ngr;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = xcd / xpg, idx = slot * xpg + (xcd % xpg);
  if (idx >= a.nl * a.NC) return;
  const int l = idx / a.NC, c = idx - l * a.NC;
  const GPersistLayer L = a.L[l];
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, P = L.P, ldP = L.ldP, I = L.I;
  const int nkb = (P + 15) >> 4, nkbx = (I + 15) >> 4;
  const int row0 = grp * GP_ROWS, cell0 = c * CW;
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  const size_t g2_per = (size_t)NP_NCH * GP_SLOT;                      // [group][layer][t][tile][k-block] slots
  const GpBuf b2 = gp_buf((const char*)a.gran2 + (size_t)(grp * a.nl + l) * T * g2_per, (size_t)T * g2_per);
  const GpBuf b2x = gp_buf((const char*)a.gran2 + (size_t)(grp * a.nl + (l > 0 ? l - 1 : 0)) * T * g2_per, (size_t)T * g2_per);
  const unsigned frag_off = (unsigned)lane * 16u;
  auto slot2 = [&](int t, int r, int jb) { return (unsigned)((((size_t)t * NR + r) * NP_NKB + jb) * GP_SLOT); };

  for (int e = tid; e < 7 * CW; e += GP_WAVES * 64) {
    const int k = e / CW, cl = e - k * CW, cell = min(cell0 + cl, H - 1);
    if (k < 3) S.peep[cl][k] = (k == 0 ? L.wi : k == 1 ? L.wf : L.wo)[cell];
    else S.bias[cl][k - 3] = L.bias[(k - 3) * H + cell];
  }
  for (int e = tid; e < GP_NR * NP_NKB * 64; e += GP_WAVES * 64)       // the carried state h(-1) is zero (cell.zero_state)
    *reinterpret_cast<f32x4*>(&S.mB[0][0][0][0] + 4 * e) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 32) (&S.cnt_x[0][0])[tid] = 0u;
  __syncthreads();
  const unsigned* dead = &S.dead;
  auto fail = [&]() {
    if (lane == 0) {
      __hip_atomic_store(&S.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  const int len0 = a.len[row0 + lr], len1 = a.len[row0 + 16 + lr];

  if (w < 4) {
    // =============================== R waves ===============================
    // resident K_h fragments: A[row lr = 4 * cell + gate][k = 16 jb + 4 q + u], jb = w + 4 jj; jj = 7 in LDS
    float4 kh[NT][KR];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int cell = cell0 + 4 * i + (lr >> 2);
      const float* kr = L.KhT + (size_t)((lr & 3) * H + min(cell, H - 1)) * ldP;
#pragma unroll
      for (int jj = 0; jj < NP_KBW; ++jj) {
        const int k = 16 * (w + 4 * jj) + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(kr + min(k, ldP - 4));
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        const bool ok = k < P && cell < H;
        const float4 f = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        if (jj < KR) kh[i][jj < KR ? jj : 0] = f;
        else *reinterpret_cast<float4*>(&S.khl[w][jj - KR < 0 ? 0 : jj - KR][i][lane][0]) = f;
      }
    }
    float cprev[NR] = {0.f, 0.f};
    const bool cellw = w < NT;                                        // this wave owns gate tile w (NT = 2: waves 0, 1)
    float* const pbw = &S.pb[w][0][0][lane][0];                        // + (i * NR + r) * 256
    const float* const mbw = &S.mB[0][w][lane][0];                     // + (r * NP_NKB + 4 jj) * 256
    const float* const khw = &S.khl[w][0][0][lane][0];                 // + ((jj - KR) * NT + i) * 256
    const float* const pbc = &S.pb[0][cellw ? w : 0][0][lane][0];      // + (k * NU + r) * 256   (gate tile w)
    const float* const pwc = &S.peep[4 * (cellw ? w : 0) + q][0];
    float* const stc = &S.st[0][lr][4 * (cellw ? w : 0) + q];          // + k * GP_ROWS * CW + r * 16 * CW
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (!gp_wait(&S.cnt_x[r][w], (unsigned)t + 1u, dead)) return;         // x-part (+ bias) of step t, tile r
        if (t > 0 && !gp_wait(&S.cnt_m[r], 2u * (unsigned)t, dead)) return;   // h(t-1) of the tile is in LDS
        {
          f32x4 acc[NT];
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[i] = *reinterpret_cast<const f32x4*>(pbw + (i * NR + r) * 256);
          if (t > 0) {
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int jj = 0; jj < NP_KBW; ++jj) {
              if (w + 4 * jj < nkb) {
                const float4 b = *reinterpret_cast<const float4*>(mbw + (r * NP_NKB + 4 * jj) * 256);
                float4 ka[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) ka[i] = jj < KR ? kh[i][jj < KR ? jj : 0] : *reinterpret_cast<const float4*>(khw + ((jj - KR < 0 ? 0 : jj - KR) * NT + i) * 256);
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].x, b.x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].y, b.y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].z, b.z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].w, b.w, acc[i], 0, 0, 0);
              }
            }
            __builtin_amdgcn_s_setprio(0);
          }
#pragma unroll
          for (int i = 0; i < NT; ++i) *reinterpret_cast<f32x4*>(pbw + (i * NR + r) * 256) = acc[i];
        }
        gp_signal(&S.cnt_p[r], lane);
        if (!gp_wait(&S.cnt_p[r], 4u * ((unsigned)t + 1u), dead)) return;
        if (t > 0 && !gp_wait(&S.cnt_s[r], 4u * (unsigned)t, dead)) return;  // the stash of step t-1 has left the stage
        // the cell of gate tile w: lane (q, lr) = row 16 r + lr, cell 4 w + q, gates i j f o
        const bool live = t < (r ? len1 : len0);
        if (cellw) {
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(pbc + (0 * NU + r) * 256), p1 = *reinterpret_cast<const f32x4*>(pbc + (1 * NU + r) * 256);
          const f32x4 p2 = *reinterpret_cast<const f32x4*>(pbc + (2 * NU + r) * 256), p3 = *reinterpret_cast<const f32x4*>(pbc + (3 * NU + r) * 256);
          const f32x4 z = ((p0 + p1) + p2) + p3;
          const float cpv = cprev[r];
          const f32x4 pw = *reinterpret_cast<const f32x4*>(pwc);
          const float gi = gp_sigmoid(z[0] + pw[0] * cpv);
          const float gf = gp_sigmoid(z[2] + a.forget_bias + pw[1] * cpv);
          const float gj = gp_tanh(z[1]);
          const float cn = gf * cpv + gi * gj;
          const float go = gp_sigmoid(z[3] + pw[2] * cn);
          const float hh = go * gp_tanh(cn);
          cprev[r] = live ? cn : cpv;
          float* const d = stc + r * 16 * CW;
          d[0 * GP_ROWS * CW] = live ? gi : 0.f; d[1 * GP_ROWS * CW] = live ? gj : 0.f;
          d[2 * GP_ROWS * CW] = live ? gf : 0.f; d[3 * GP_ROWS * CW] = live ? go : 0.f;
          d[4 * GP_ROWS * CW] = cprev[r];
          d[5 * GP_ROWS * CW] = live ? hh : 0.f;
        }
        gp_signal(&S.cnt_h[r], lane);
        if (!gp_wait(&S.cnt_h[r], 4u * ((unsigned)t + 1u), dead)) return;     // every cell of the tile is in the stage
        // the tile's stash (gate activations, c, h) from the LDS stage, a quarter per R wave: 4 lanes write one 64-byte row piece
#pragma unroll
        for (int it = 0; it < (6 * 16 * NT + 255) / 256; ++it) {
          const int e = it * 256 + w * 64 + lane;
          const int cq = e % NT, pr = e / NT, row = 16 * r + (pr & 15), k = min(pr >> 4, 5);
          const float4 v = *reinterpret_cast<const float4*>(&S.st[k][row][4 * cq]);
          const size_t rowg = (size_t)t * N + row0 + row;
          float* dst = (k < 4 ? L.gates + rowg * H4 + k * H : k == 4 ? L.c + (rowg + N) * H : L.h + rowg * L.ldH) + cell0 + 4 * cq;
          if (e < 6 * 16 * NT && cell0 + 4 * cq < H) gp_stash_store(dst, v);
        }
        gp_signal(&S.cnt_s[r], lane);
      }
    }
    return;
  }

  if (w < 8) {
    // =============================== X waves: ahead of the R waves ===============================
    const int xw = w - 4;
    float* const pbx = &S.pb[xw][0][0][lane][0];                       // + (i * NR + r) * 256
    float4 kx[NT][KX];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int cell = cell0 + 4 * i + (lr >> 2);
      const float* kr = L.KxT + (size_t)((lr & 3) * H + min(cell, H - 1)) * L.ldI;
#pragma unroll
      for (int jj = 0; jj < NP_KBW; ++jj) {
        const int k = 16 * (xw + 4 * jj) + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(kr + min(k, L.ldI - 4));
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        const bool ok = k < I && cell < H;                             // (the copy is zero beyond column I)
        const float4 f = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        if (jj < KX) kx[i][jj < KX ? jj : 0] = f;
        else *reinterpret_cast<float4*>(&S.kxl[xw][jj - KX < 0 ? 0 : jj - KX][i][lane][0]) = f;
      }
    }
    const float* const kxw = &S.kxl[xw][0][0][lane][0];                // + ((jj - KX) * NT + i) * 256
    const int nsx = (nkbx - xw + 3) >> 2;                              // this wave's k-blocks: xw, xw + 4, ...
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const bool live = t < (r ? len1 : len0);
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
          acc[i] = xw == 0 ? *reinterpret_cast<const f32x4*>(&S.bias[q][0] + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        // x(t) in two halves of four k-blocks (eight pieces at once would not leave the registers for the weights)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (xw + 16 * half >= nkbx) break;                           // (uniform)
          unsigned lo[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) lo[jj] = slot2(t, r, min(xw + 4 * (4 * half + jj), nkbx - 1));
          f32x4 xv[4];
          const int nh = max(0, min(4, nsx - 4 * half));               // pieces of this half
          if (l == 0) {
            const float* xr = L.in + ((size_t)t * N + row0 + 16 * r + lr) * L.ldI;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              xv[jj] = *reinterpret_cast<const f32x4*>(xr + min(16 * min(xw + 4 * (4 * half + jj), nkbx - 1) + 4 * q, L.ldI - 4));
            asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              if (16 * min(xw + 4 * (4 * half + jj), nkbx - 1) + 4 * q >= L.ldI) xv[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
          } else if (!gp_sweep<4, false, 4, true>(b2x, lo, nh, frag_off, slot2(t, r, min(xw + 4 * (4 * half + (lane >> 1)), nkbx - 1)) + (unsigned)(lane & 1) * 512u + 496u,
                                                  lane < 2 * nh, err, [&](int k, const f32x4& v) { xv[k] = v; })) { fail(); return; }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int jg = 4 * half + jj;                              // (compile-time after unrolling)
            if (xw + 4 * jg < nkbx) {
              const float b0 = live ? xv[jj][0] : 0.f, b1 = live ? xv[jj][1] : 0.f, b2_ = live ? xv[jj][2] : 0.f, b3 = live ? xv[jj][3] : 0.f;
              float4 ka[NT];
#pragma unroll
              for (int i = 0; i < NT; ++i) ka[i] = jg < KX ? kx[i][jg < KX ? jg : 0] : *reinterpret_cast<const float4*>(kxw + ((jg - KX < 0 ? 0 : jg - KX) * NT + i) * 256);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].x, b0, acc[i], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].y, b1, acc[i], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].z, b2_, acc[i], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].w, b3, acc[i], 0, 0, 0);
            }
          }
          __builtin_amdgcn_s_setprio(0);
        }
        if (t > 0 && !gp_wait(&S.cnt_h[r], 4u * (unsigned)t, dead)) return;       // the cells of step t-1 have read the tiles
#pragma unroll
        for (int i = 0; i < NT; ++i) *reinterpret_cast<f32x4*>(pbx + (i * NR + r) * 256) = acc[i];
        gp_signal(&S.cnt_x[r][xw], lane);
      }
    }
    return;
  }

  // =============================== G waves: publish, gather (tile gw & 1) ===============================
  __builtin_amdgcn_s_setprio(3);
  const int gw = w - 8, r = gw & 1, gp = gw >> 1;
  const int lenr = r ? len1 : len0;
  const int nvg = (nkb - gp + 1) >> 1;                                 // this wave's k-blocks of the gather: gp, gp + 2, ...
  float* const mbg = &S.mB[r][gp][lane][0];                            // + 2 n * 256
  // this workgroup's own piece of the state: 4 NT cells = quads q < NT of chunk (4 NT c) / 16 (NT = 4: the whole chunk, NT = 2: one
  // half); lane (q, lr), q < NT = row 16 r + lr, its cells 4 q .. 4 q + 3
  const bool pubq = q < NT;
  const float* const sth4 = &S.st[5][16 * r + lr][4 * (pubq ? q : 0)];
  const int srow = row0 + 16 * r + lr, scol = cell0 + 4 * q;
  const int pjb = (CW * c) >> 4;                                       // the chunk this workgroup writes into
  const unsigned poff = (unsigned)(((CW * c) & 15) >> 2) * 256u + (unsigned)lane * 16u;      // ... and its quads' offset inside it (quad-major: 256 bytes per quad)
  f32x4 mcar = {0.f, 0.f, 0.f, 0.f};                                   // carried state of this lane's four cells (the stash's mst)
  for (int e = gw * 64 + lane; e < GP_ROWS * NT; e += 256) {           // slot 0 of the carried states is zero (cell.zero_state)
    const int row = e / NT, cq = e - row * NT;
    if (cell0 + 4 * cq < H) {
      *reinterpret_cast<float4*>(L.c + (size_t)(row0 + row) * H + cell0 + 4 * cq) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(L.mst + (size_t)(row0 + row) * ldP + cell0 + 4 * cq) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int t = 0; t < T; ++t) {
    if (!gp_wait(&S.cnt_h[r], 4u * ((unsigned)t + 1u), dead)) return;  // the cells of step t, tile r: h is in LDS
    const f32x4 hv = *reinterpret_cast<const f32x4*>(sth4);
    if (gp == 0 && pubq) gp_store(b2, slot2(t, r, pjb) + poff, hv);    // the hand-off: this workgroup's quads of h(t)
    if (t + 1 < T) {
      // gather h(t) of the tile for the recurrent product of step t+1; dynamic_rnn carries the state of a finished row through
      // unchanged (the carried state lives in mB itself: a finished row's lanes are not written)
      unsigned lo[16];
#pragma unroll
      for (int n = 0; n < 16; ++n) lo[n] = slot2(t, r, min(gp + 2 * n, nkb - 1));
      const bool live = t < lenr;
      if (!gp_sweep<16, true, 16, true>(b2, lo, nvg, frag_off, slot2(t, r, min(gp + 2 * (lane >> 1), nkb - 1)) + (unsigned)(lane & 1) * 512u + 496u,
                                        lane < 2 * nvg, err, [&](int k, const f32x4& v) { if (k < nvg && live) *reinterpret_cast<f32x4*>(mbg + 2 * k * 256) = v; })) { fail(); return; }
    }
    gp_signal(&S.cnt_m[r], lane);
    if (gp == 1 && pubq && scol < H) {                                 // this workgroup's columns of the carried state / masked output, behind the hand-off
      const bool live = t < lenr;
      mcar = live ? hv : mcar;
      *reinterpret_cast<float4*>(L.mst + ((size_t)(t + 1) * N + srow) * ldP + scol) = make_float4(mcar[0], mcar[1], mcar[2], mcar[3]);
      if (L.out) *reinterpret_cast<float4*>(L.out + ((size_t)t * N + srow) * ldP + scol) = make_float4(hv[0], hv[1], hv[2], hv[3]);
    }
  }
}

template <int NT, int KR, int KX>
__global__ __launch_bounds__(GP_WAVES * 64, 3) void k_glstm_np_fwd(const GPersistArgs a) {
  __shared__ __attribute__((aligned(16))) NpLds<NT, KR, KX> S;
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  np_fwd_body<NT, KR, KX>(a, S);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[a.nl - 1].h[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned g1 = gen + 1u;
      __hip_atomic_store(ctl + DP_CTL_GEN, g1 >= (1u << 21) ? 1u : g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// -------------------------------------------------------------------------------------------------------------------------------------
// BPTT through unprojected cells as ONE persistent launch: the mirror of k_glstm_np_fwd, again with a single hand-off per step.  The
// state gradient of a cell slice needs every workgroup's dz(t+1) . K_h^T restricted to ITS columns, so each workgroup publishes its
// partial over all H state columns (R waves; X waves: dz(t) . K_x^T for the layer below) and the OWNER of 8 cells sums the 512-byte
// pieces of the H / 8 producers in slice order (a reduce-scatter with the owner as reducer); nothing is gathered back -- the owner is
// the only consumer of its dh.  8 cells per workgroup (NT = 2: the K_h^T / K_x^T slices, 64 registers per wave, stay in VGPRs).
// Rings as in k_glstm_bwd: the state-gradient ring GP_R1 steps deep (its re-use is ordered by the recurrence itself), the
// input-gradient ring between two layers GP_XR deep with an explicit check that the slot about to be written has been re-armed
// by its consumer.  Cell gradient, masking, the dz stash: kernels.hip k_bwd_a2 / gp_bwd_body.
constexpr int NPB_NT = 2;
struct NpLdsB {
  float dhS[GP_NR][16][4 * NPB_NT];         // dh(t) of a tile [row][cell]: the reducing G waves -> the cells
  float dzB[GP_NR][NPB_NT][64][4];          // dz(t) as B fragments of both gradient products [tile][gate tile][lane (cell q, row lr)][gate]
  float st[4][GP_ROWS][4 * NPB_NT];         // dz(t) in stash order [gate][row][cell]
  float gs[2][GP_NR][2][64][4];             // the two reducing G waves' sums [step parity][tile][wave][half wave, piece lane]
  float pfs[GP_NR][5][16][4 * NPB_NT];      // the stash of the NEXT step of a tile: gate activations i, j, f, o and c(t-1)
  float peep[4 * NPB_NT][4];
  unsigned cnt_h[GP_NR], cnt_m[GP_NR], cnt_g[GP_NR], cnt_z[GP_NR], cnt_f[GP_NR], cnt_q[GP_NR], dead, pad_[3];
};

__device__ __forceinline__ void np_bwd_body(const GPersistArgs& a, NpLdsB& S) {
  constexpr int NT = NPB_NT, NR = GP_NR, CW = 4 * NT;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ngr = a.N / GP_ROWS, xpg = 8 / ngr;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = xcd / xpg, idx = slot * xpg + (xcd % xpg);
  if (idx >= a.nl * a.NC) return;
  const int l = idx / a.NC, c = idx - l * a.NC;
  const GPersistLayer L = a.L[l];
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, P = L.P, ldP = L.ldP, I = L.I, NC = a.NC;
  const int nkb = (P + 15) >> 4, nkbx = (I + 15) >> 4;
  const int row0 = grp * GP_ROWS, cell0 = c * CW;
  const bool top = l == a.nl - 1;
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  // rings: [group][layer][ring step][tile][k-block][producer] slots of 1 KB
  const size_t g1_per = (size_t)NP_NCH * NC * GP_SLOT;
  const GpBuf b1 = gp_buf((const char*)a.gran1 + (size_t)(grp * a.nl + l) * GP_R1 * g1_per, GP_R1 * g1_per);
  const GpBuf b3 = gp_buf((const char*)a.gran3 + (size_t)(grp * (a.nl + 1) + l + 1) * GP_XR * g1_per, GP_XR * g1_per);      // what the layer above hands to this one
  const GpBuf b3x = gp_buf((const char*)a.gran3 + (size_t)(grp * (a.nl + 1) + l) * GP_XR * g1_per, GP_XR * g1_per);         // what this layer hands down
  const unsigned pair_off = (unsigned)((lane >> 5) * GP_SLOT + (lane & 31) * 16);   // reducer: lanes 0..31 read producer p's half chunk, lanes 32..63 producer p + 1's
  auto slot1 = [&](int par, int r, int jb, int p) { return (unsigned)((((size_t)(par * NR + r) * NP_NKB + jb) * NC + p) * GP_SLOT); };
  // this workgroup OWNS the 8 columns (k-block jbr, half hh) of its layer's state: it sums their pieces of every producer
  const int jbr = (CW * c) >> 4, hh = ((CW * c) & 15) >> 3;

  for (int e = tid; e < 3 * CW; e += GP_WAVES * 64) {
    const int k = e / CW, cl = e - k * CW, cell = min(cell0 + cl, H - 1);
    S.peep[cl][k] = (k == 0 ? L.wi : k == 1 ? L.wf : L.wo)[cell];
  }
  if (tid < 16) (&S.cnt_h[0])[tid] = 0u;                                 // (all counters, dead)
  __syncthreads();
  const unsigned* dead = &S.dead;
  auto fail = [&]() {
    if (lane == 0) {
      __hip_atomic_store(&S.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  const int len0 = a.len[row0 + lr], len1 = a.len[row0 + 16 + lr];

  if (w < 8) {
    // R waves (w < 4): K_h, the state-gradient product and the cell.  X waves: K_x, the input-gradient product for the layer below.
    // Resident weights: output tile pt = ww + 4 jj of the product's P (or I) columns, gate tile i:
    //   A[row lr = column 16 pt + lr][k = cell 4 i + q] for the four k-steps gate 0..3
    const bool isx = w >= 4;
    const int ww = w & 3;
    const bool noprod = isx && l == 0;                                   // (layer 0 feeds nobody: its input is the data)
    const float* KT = isx ? L.KxT : L.KhT;
    const int ldK = isx ? L.ldI : ldP, PW = isx ? I : P, nkw = isx ? nkbx : nkb;
    float4 kw[NP_KBW][NT];
#pragma unroll
    for (int jj = 0; jj < NP_KBW; ++jj) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int col = 16 * (ww + 4 * jj) + lr, cell = cell0 + 4 * i + q;
        const float* kr = KT + (size_t)min(cell, H - 1) * ldK + min(col, ldK - 1);
        const size_t gs_ = (size_t)H * ldK;
        float4 v = make_float4(kr[0], kr[gs_], kr[2 * gs_], kr[3 * gs_]);
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        const bool ok = col < PW && cell < H && !noprod;
        kw[jj][i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
      }
    }
    const float* const dzr = &S.dzB[0][0][lane][0];                     // + (r * NT + i) * 256
    const GpBuf& bpub = isx ? b3x : b1;
    // the product of one row tile and its publication: three output tiles in flight, a tile leaves as soon as its 4 NT products are done
    auto product = [&](int r, int ring) {
      const unsigned ln_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const unsigned fo = ln_ << 4;
      if (isx) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int j0 = 0; j0 < NP_KBW; j0 += 3) {
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const float4 b = *reinterpret_cast<const float4*>(dzr + (r * NT + i) * 256);
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < NP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kw[j0 + j < NP_KBW ? j0 + j : 0][i].x, b.x, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < NP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kw[j0 + j < NP_KBW ? j0 + j : 0][i].y, b.y, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < NP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kw[j0 + j < NP_KBW ? j0 + j : 0][i].z, b.z, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j0 + j < NP_KBW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kw[j0 + j < NP_KBW ? j0 + j : 0][i].w, b.w, acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int jj = j0 + j;
          if (jj < NP_KBW && ww + 4 * jj < nkw) gp_store(bpub, slot1(ring, r, ww + 4 * jj, c) + fo, acc[j]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    };

    if (isx) {
      // =============================== X waves ===============================
      // the stash of step t-1 (gate activations, c(t-2): 16 rows x 4 NT cells x 5 values per tile) travels through these waves into LDS a step ahead
      float4 pv;
      auto fetch = [&](int t, int r) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e = ww * 64 + ln;
        const int cq = e % NT, pr = min(e / NT, 79), row = pr & 15, k = pr >> 4;
        const size_t rowg = (size_t)t * N + row0 + 16 * r + row;
        const int cell = min(cell0 + 4 * cq, H - 4);
        pv = *reinterpret_cast<const float4*>((k < 4 ? L.gates + rowg * H4 + k * H : L.c + rowg * H) + cell);
      };
      auto stage = [&](int r) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e = ww * 64 + ln;
        const int cq = e % NT, pr = e / NT, row = pr & 15, k = min(pr >> 4, 4);
        if (e < 5 * 16 * NT) *reinterpret_cast<float4*>(&S.pfs[r][k][row][4 * cq]) = pv;
        gp_signal(&S.cnt_f[r], lane);
      };
#pragma unroll
      for (int r = 0; r < NR; ++r) { fetch(T - 1, r); stage(r); }
      for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          // back-pressure: the slots of ring step s % GP_XR this wave is about to write (its 8 output chunks, both halves) must have
          // been re-armed by their consumers in the layer below: every word of their last 16 bytes carries the sentinel again
          if (!noprod && s >= GP_XR) {
            const unsigned so = slot1(s % GP_XR, r, min(ww + 4 * (lane >> 1), nkbx - 1), c) + (unsigned)(lane & 1) * 512u + 496u;
            const bool son = lane < 16 && ww + 4 * (lane >> 1) < nkbx;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned polls = 0;; ++polls) {
              const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(b3x.rs, so, 0, GP_SC1 | GP_VOL);
              const bool armed = (y[0] == GP_SENT) & (y[1] == GP_SENT) & (y[2] == GP_SENT) & (y[3] == GP_SENT);
              if (__all(!son || armed)) break;
              asm volatile("" ::: "memory");
              if ((polls & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { fail(); return; }
              __builtin_amdgcn_s_sleep(2);
            }
          }
          if (!gp_wait(&S.cnt_h[r], 4u * ((unsigned)s + 1u), dead)) return;       // dz(t) of the tile is in LDS (and the cells have read the stage)
          if (t > 0) { fetch(t - 1, r); stage(r); }
          if (!noprod) {
            if (t > 0 && !gp_wait(&S.cnt_q[r], 4u * ((unsigned)s + 1u), dead)) return;      // (the R waves' product first: it is on the recurrence's critical path)
            product(r, s % GP_XR);
          }
          gp_signal(&S.cnt_z[r], lane);                                     // (dzB of the tile may be overwritten)
        }
      }
      return;
    }

    // =============================== R waves ===============================
    const bool cellw = w < NT;                                             // this wave owns gate tile w
    float ccur[NR], dcar[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      ccur[r] = L.c[((size_t)T * N + row0 + 16 * r + lr) * H + min(cell0 + 4 * (cellw ? w : 0) + q, H - 1)];
      dcar[r] = 0.f;
    }
    const float* const pfc = &S.pfs[0][0][lr][4 * (cellw ? w : 0) + q];  // + ((r * 5 + k) * 16) * CW
    const float* const pwc = &S.peep[4 * (cellw ? w : 0) + q][0];
    float* const dzw = &S.dzB[0][cellw ? w : 0][lane][0];                 // + r * NT * 256
    float* const stc = &S.st[0][lr][4 * (cellw ? w : 0) + q];             // + g * GP_ROWS * CW + r * 16 * CW
    // this workgroup's pieces of both rings are re-armed by its R waves once its G waves have summed them
    for (int s = 0; s < T; ++s) {
      const int t = T - 1 - s;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (!gp_wait(&S.cnt_m[r], (unsigned)s + 1u, dead)) return;         // dh(t) of the tile is in LDS
        if (s > 0 && !gp_wait(&S.cnt_z[r], 4u * (unsigned)s, dead)) return;  // the X waves have taken dz(t+1)
        if (!gp_wait(&S.cnt_f[r], 4u * ((unsigned)s + 1u), dead)) return;  // the stash of step t is in LDS
        const bool live = t < (r ? len1 : len0);
        if (cellw) {
          const float dh = S.dhS[r][lr][4 * w + q];
          const f32x4 pw = *reinterpret_cast<const f32x4*>(pwc);
          const float* const pq = pfc + r * 5 * 16 * CW;
          const float gi = pq[0], gj = pq[16 * CW], gf = pq[2 * 16 * CW], go = pq[3 * 16 * CW], cp = pq[4 * 16 * CW];
          const float cc = ccur[r], dcv = dcar[r];
          const float tc = gp_tanh(cc);
          const float dao = dh * tc * go * (1.f - go);
          const float dcn = dcv + dh * go * (1.f - tc * tc) + dao * pw[2];
          const float daf = dcn * cp * gf * (1.f - gf);
          const float dai = dcn * gj * gi * (1.f - gi);
          const float dj = dcn * gi * (1.f - gj * gj);
          dcar[r] = live ? dcn * gf + dai * pw[0] + daf * pw[1] : dcv;
          ccur[r] = cp;
          const f32x4 dz = {live ? dai : 0.f, live ? dj : 0.f, live ? daf : 0.f, live ? dao : 0.f};
          *reinterpret_cast<f32x4*>(dzw + r * NT * 256) = dz;
          float* const d = stc + r * 16 * CW;
          d[0 * GP_ROWS * CW] = dz[0]; d[1 * GP_ROWS * CW] = dz[1]; d[2 * GP_ROWS * CW] = dz[2]; d[3 * GP_ROWS * CW] = dz[3];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the re-arming stores of the step before)
        gp_signal(&S.cnt_h[r], lane);
        if (!gp_wait(&S.cnt_h[r], 4u * ((unsigned)s + 1u), dead)) return; // every cell's dz of the tile is in LDS
        if (t > 0) { product(r, s % GP_R1); }                             // (dh(-1) has no consumer)
        gp_signal(&S.cnt_q[r], lane);
        // dz(t) over the gate activations of the stash, a quarter of the tile per R wave
        {
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const int e = w * 64 + ln;
          const int cq = e % NT, pr = e / NT, row = 16 * r + (pr & 15), k = min(pr >> 4, 3);
          const float4 v = *reinterpret_cast<const float4*>(&S.st[k][row][4 * cq]);
          float* dst = L.gates + ((size_t)t * N + row0 + row) * H4 + k * H + cell0 + 4 * cq;
          if (e < 4 * 16 * NT && cell0 + 4 * cq < H) gp_stash_store(dst, v);
        }
        // re-arm what this workgroup's G waves have summed for this step (they did before they handed dh(t) over: the wait at the top)
        if (s > 0) gp_rearm(b1, slot1((s - 1) % GP_R1, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
        if (!top) gp_rearm(b3, slot1(s % GP_XR, r, jbr, 0) + (unsigned)hh * 512u, NC, w, lane);
      }
    }
    return;
  }

  // =============================== G waves: sum this workgroup's pieces (tile gw & 1) ===============================
  __builtin_amdgcn_s_setprio(3);
  const int gw = w - 8, r = gw & 1, gp = gw >> 1;
  const int ppw = (NC + 1) >> 1, pp0 = gp * ppw, pn = max(0, min(ppw, NC - pp0));
  const int nlr = (pn + 1 - (lane >> 5)) >> 1;                           // (two producers per load: lanes 0..31 the even ones, 32..63 the odd ones)
  // reducer lane L holds piece lane L & 31: row (L & 15), cells 4 ((L & 31) >> 4) .. + 3 of this workgroup
  const int rrow = row0 + 16 * r + (lane & 15), rq = (lane & 31) >> 4, rcol = cell0 + 4 * rq;
  const int rlen = a.len[rrow];
  for (int s = 0; s < T; ++s) {
    const int t = T - 1 - s;
    float4 dtop = make_float4(0.f, 0.f, 0.f, 0.f);
    if (top && rcol < a.ld_dout) {
      dtop = *reinterpret_cast<const float4*>(a.dout_top + ((size_t)t * N + rrow) * a.ld_dout + rcol);
      dtop = make_float4(rcol < P ? dtop.x : 0.f, rcol + 1 < P ? dtop.y : 0.f, rcol + 2 < P ? dtop.z : 0.f, rcol + 3 < P ? dtop.w : 0.f);
    }
    f32x4 sa = {0.f, 0.f, 0.f, 0.f};
    if (!top) {
      // the input-gradient partials of the layer above at time t (published a diagonal ago as a rule: read first, poll if not there)
      unsigned lo[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) lo[k] = slot1(s % GP_XR, r, jbr, min(pp0 + 2 * k, NC - 1)) + (unsigned)hh * 512u;
      if (!gp_sweep<16, false, 16, false>(b3, lo, nlr, pair_off, slot1(s % GP_XR, r, jbr, min(pp0 + lane, NC - 1)) + (unsigned)hh * 512u + 496u, lane < pn, err,
                                   [&](int k, const f32x4& v) { if (k == 0) sa = k < nlr ? v : f32x4{0.f, 0.f, 0.f, 0.f}; else if (k < nlr) sa += v; })) { fail(); return; }
    }
    if (s > 0) {
      // the state-gradient partials of this layer from time t + 1
      unsigned lo[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) lo[k] = slot1((s + GP_R1 - 1) % GP_R1, r, jbr, min(pp0 + 2 * k, NC - 1)) + (unsigned)hh * 512u;
      f32x4 ua = {0.f, 0.f, 0.f, 0.f};
      if (!gp_sweep<16, true, 16, false>(b1, lo, nlr, pair_off, slot1((s + GP_R1 - 1) % GP_R1, r, jbr, min(pp0 + lane, NC - 1)) + (unsigned)hh * 512u + 496u, lane < pn, err,
                                  [&](int k, const f32x4& v) { if (k == 0) ua = k < nlr ? v : f32x4{0.f, 0.f, 0.f, 0.f}; else if (k < nlr) ua += v; })) { fail(); return; }
      sa += ua;
    }
    *reinterpret_cast<f32x4*>(&S.gs[s & 1][r][gp][lane][0]) = sa;
    gp_signal(&S.cnt_g[r], lane);
    if (!gp_wait(&S.cnt_g[r], 2u * ((unsigned)s + 1u), dead)) return;
    if (gp == 0) {
      const float* const gsr = &S.gs[s & 1][r][0][lane & 31][0];
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(gsr), a1 = *reinterpret_cast<const f32x4*>(gsr + 32 * 4);
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(gsr + 64 * 4), c1 = *reinterpret_cast<const f32x4*>(gsr + 96 * 4);
      f32x4 tot = (((a0 + a1) + c0) + c1) + f32x4{dtop.x, dtop.y, dtop.z, dtop.w};
      if (!(t < rlen)) tot = f32x4{0.f, 0.f, 0.f, 0.f};
      if (lane < 32) *reinterpret_cast<f32x4*>(&S.dhS[r][lane & 15][4 * rq]) = tot;
      gp_signal(&S.cnt_m[r], lane);
    }
  }
}

__global__ __launch_bounds__(GP_WAVES * 64, 3) void k_glstm_np_bwd(const GPersistArgs a) {
  __shared__ __attribute__((aligned(16))) NpLdsB S;
  gu32* ctl = (gu32*)a.ctl;
  const unsigned gen = __hip_atomic_load(ctl + DP_CTL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  np_bwd_body(a, S);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctl + DP_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      if (__hip_atomic_load(ctl + DP_CTL_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        a.L[0].gates[0] = __builtin_nanf("");
      __hip_atomic_store(ctl + DP_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned g1 = gen + 1u;
      __hip_atomic_store(ctl + DP_CTL_GEN, g1 >= (1u << 21) ? 1u : g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int gp_grid(const GPersistArgs& a) {
  const int ngr = a.ngl ? a.ngl : a.N / GP_ROWS, xpg = 8 / ngr, nwg = a.nl * a.NC;
  return 8 * ((nwg + xpg - 1) / xpg);
}
int gpersist_grid(const GPersistArgs& a) { return gp_grid(a); }
size_t gpersist_lds_bytes() { return sizeof(GpLdsB<5>) > sizeof(GpLds<5>) ? sizeof(GpLdsB<5>) : sizeof(GpLds<5>); }

// ---- how many workgroups of a persistent launch can be resident at once ----
// The persistent recurrences (this file, dpersist.hip) wait for each other inside the launch: every workgroup must be on a CU at the
// same time.  Nothing static tells that: multiProcessorCount is the whole device, a CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK), a
// compute partition or a neighbour process take CUs away silently -- and a launch that does not fit spins into its time-out on every
// step.  So the library ASKS, once per shape at rsrgan_create: `grid` workgroups of the same block size and LDS footprint (one per CU)
// count themselves in and wait until all have arrived or 5 ms have passed.
__global__ void k_resident_probe(unsigned* ctl, unsigned want) {
  extern __shared__ unsigned gp_probe_lds[];
  if (threadIdx.x == 0) {
    gp_probe_lds[0] = 1u;                                              // (the dynamic allocation is what keeps a second workgroup off this CU)
    __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 500000ull) {         // 5 ms at 100 MHz
        __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
int device_cu_count() {
  static const int n = [] {
    int dev = 0; hipDeviceProp_t p{};
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return p.multiProcessorCount;
  }();
  return n;
}
bool resident_probe(int grid, int threads, size_t lds_bytes) {
  static const bool off = [] { const char* e = getenv("RSRGAN_RESIDENT_PROBE"); return e && atoi(e) == 0; }();
  // RSRGAN_RESIDENT_CAP=n: the verdict of a device that can hold n workgroups (what a CU mask or a compute partition makes the probe
  // find; masks are not honoured in every environment, the cap always is -- tests/test_gpu_padrows.py)
  static const int cap = [] { const char* e = getenv("RSRGAN_RESIDENT_CAP"); return e ? atoi(e) : 0; }();
  if (grid < 1) return false;
  if (grid > device_cu_count() || (cap > 0 && grid > cap)) return false;      // (one workgroup per CU at these footprints)
  if (off) return true;
  if (lds_bytes < (size_t)82 * 1024) lds_bytes = (size_t)82 * 1024;    // (more than half a CU's LDS: one probe workgroup per CU, whatever the real kernel's registers allow)
  unsigned* ctl = nullptr;
  if (hipMalloc(&ctl, 2 * sizeof(unsigned)) != hipSuccess) return false;
  bool ok = hipMemset(ctl, 0, 2 * sizeof(unsigned)) == hipSuccess &&
            hipFuncSetAttribute((const void*)k_resident_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_resident_probe, dim3(grid), dim3(threads), lds_bytes, 0, ctl, (unsigned)grid);
    unsigned h[2] = {0u, 1u};
    ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, ctl, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[1] == 0u && h[0] == (unsigned)grid;
  }
  (void)hipGetLastError();
  (void)hipFree(ctl);
  return ok;
}

bool gpersist_plan(GPersistArgs& a) {
  if (a.nl < 1 || a.nl > GP_MAXL || a.T < 1 || a.T > GP_TMAX || a.H % 4 != 0) return false;
  const int ngr = a.N / GP_ROWS;
  if (a.N % GP_ROWS != 0 || (ngr != 1 && ngr != 2 && ngr != 4 && ngr != 8)) return false;
  a.NT = 5;
  a.NC = (a.H / 4 + a.NT - 1) / a.NT;
  for (int l = 0; l < a.nl; ++l) {
    const GPersistLayer& L = a.L[l];
    // (P, I need not be multiples of 4 -- res_lstm_l: 257 --: every row is zero-padded to its ld, and 16-byte pieces stop at the ld)
    if (L.P < 4 || L.P > 16 * GP_NKB || L.ldP % 4 != 0 || L.ldP < L.P || L.ldP - L.P > 3 || L.I > 16 * GP_NKB || L.ldH % 4 != 0) return false;
    if (l == 0 && (L.I < 4 || L.ldI % 4 != 0 || L.ldI < L.I || L.ldI - L.I > 3)) return false;       // (16-byte pieces of the input rows)
    if (l > 0 && L.I != a.L[l - 1].P) return false;
    if (a.res && (L.I != L.P || L.ldI != L.ldP)) return false;          // a running sum: every layer as wide as the stack's input
    if (((L.P + 15) / 16) * 2 > a.NC) return false;                // every 8-column half of a chunk needs its reducer
    if (a.NC > 40) return false;                                    // a G wave sums at most 10 producers
  }
  // every workgroup must be resident at once (they wait for each other): one 12-wave workgroup per CU.  (The static half of the
  // answer; Model::init asks the device itself, resident_probe, before it allocates the hand-off rings.)
  if (gp_grid(a) <= device_cu_count()) return true;
  // one launch per row group (res_lstm_l at 64 rows: 4 layers x 38 slices x 2 row groups = 304 workgroups)
  a.ngl = 1; a.grp0 = 0;
  return gp_grid(a) <= device_cu_count();
}
// The per-launch arming of sentinel slots as a KERNEL, not hipMemsetAsync: inside a replayed hipGraph a fill node in front of a persistent
// launch is not a dependable predecessor (seen twice: as the fork point of two branches it serialised them; and after a device-wide
// synchronisation followed by a blocking copy -- between the instantiation of a step's graphs and a replay -- the replayed launch waited
// a second for pieces that had been overwritten).  16-byte stores, grid-stride.
__global__ void k_arm(uint4* p, size_t n16) {
  const uint4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void gpersist_arm_bytes(void* p, size_t bytes, hipStream_t s) {      // bytes: a multiple of 16
  hipLaunchKernelGGL(k_arm, dim3(1024), dim3(256), 0, s, (uint4*)p, bytes / 16);
}
// the hop-2 slots of the row groups a launch covers (both regions of a residual stack): the other groups' chunks stay -- the FC workgroups
// of k_glstm_fwd_dt read the first group's while the second group's launch runs
static void gp_arm_gran2(const GPersistArgs& a, hipStream_t s) {
  const int ngr = a.N / GP_ROWS, ngl = a.ngl ? a.ngl : ngr;
  const size_t per_grp = (size_t)a.nl * a.T * GP_NCH * GP_SLOT, region = (size_t)ngr * per_grp;
  for (int rg = 0; rg < (a.res ? 2 : 1); ++rg)
    gpersist_arm_bytes((char*)a.gran2 + rg * region + (size_t)a.grp0 * per_grp, (size_t)ngl * per_grp, s);
}
size_t gpersist_gran1_bytes(const GPersistArgs& a) { return (size_t)(a.N / GP_ROWS) * a.nl * GP_R1 * GP_NCH * a.NC * GP_SLOT; }
size_t gpersist_gran2_bytes(const GPersistArgs& a) { return (size_t)(a.res ? 2 : 1) * (a.N / GP_ROWS) * a.nl * a.T * GP_NCH * GP_SLOT; }      // (res: + the running sums' region)
size_t gpersist_gran3_bytes(const GPersistArgs& a) { return (size_t)(a.N / GP_ROWS) * (a.nl + 1) * GP_XR * GP_NCH * a.NC * GP_SLOT; }

// (the memset arms the launch's hop-2 slots; gran1 / gran3 are armed once, gpersist_arm, and re-armed by the kernels themselves)
void gpersist_arm(const GPersistArgs& a, hipStream_t s) {
  (void)hipMemsetAsync(a.gran1, 0xFF, gpersist_gran1_bytes(a), s);
  if (a.gran3) (void)hipMemsetAsync(a.gran3, 0xFF, gpersist_gran3_bytes(a), s);
  (void)hipMemsetAsync(a.ctl + 4, 0, 12 * sizeof(unsigned), s);          // (tagged rings: every word's parity bit is 1 now, pass 0 writes 0)
}
// (-DGP_PROG_ONLY=mask: the progressive sweeps, gp_sweep_prog -- the harness only: measured slower, profiles/r5_gpersist_progressive_sweep_negative.txt)
#ifndef GP_PROG_ONLY
#define GP_PROG_ONLY 0
#endif
void launch_glstm_fwd(const GPersistArgs& a, hipStream_t s) {
  gp_arm_gran2(a, s);
  const dim3 g(gp_grid(a)), b(GP_WAVES * 64);
  if (a.tags && !GP_PROG_ONLY) {
    if (a.res) hipLaunchKernelGGL((k_glstm_fwd<5, 0, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_glstm_fwd<5, 0, false, true>), g, b, 0, s, a);
  } else if (a.res) hipLaunchKernelGGL((k_glstm_fwd<5, GP_PROG_ONLY, true, false>), g, b, 0, s, a);
  else hipLaunchKernelGGL((k_glstm_fwd<5, GP_PROG_ONLY, false, false>), g, b, 0, s, a);
  ++g_chain_launches;
}
// ---- the unprojected form: plan, sizes, launch ----
bool gpersist_np_plan(GPersistArgs& a, int nt_force) {
  if (a.nl < 1 || a.nl > GP_MAXL || a.T < 1 || a.T > GP_TMAX || a.H % 16 != 0 || a.H > 16 * NP_NKB) return false;
  const int ngr = a.N / GP_ROWS;
  if (a.N % GP_ROWS != 0 || (ngr != 1 && ngr != 2 && ngr != 4 && ngr != 8)) return false;
  for (int l = 0; l < a.nl; ++l) {
    const GPersistLayer& L = a.L[l];
    if (L.P != a.H || L.ldP % 4 != 0 || L.ldP < L.P || L.ldH % 4 != 0 || L.I < 4 || L.I > 16 * NP_NKB || L.ldI % 4 != 0 || L.ldI < L.I || L.ldI - L.I > 3) return false;
    if (l > 0 && L.I != a.H) return false;
  }
  // 8 cells per workgroup (every weight in registers) where that many workgroups are resident at once, else 16
  static const int nt_env = [] { const char* e = getenv("RSRGAN_GP_NP_NT"); return e ? atoi(e) : 0; }();
  for (int nt : {2, 4}) {
    if ((nt_env && nt != nt_env) || (nt_force && nt != nt_force)) continue;
    a.NT = nt; a.NC = a.H / (4 * nt);
    if (gp_grid(a) <= device_cu_count()) return true;
  }
  return false;
}
size_t gpersist_np_gran2_bytes(const GPersistArgs& a) { return (size_t)(a.N / GP_ROWS) * a.nl * a.T * NP_NCH * GP_SLOT; }
size_t gpersist_np_lds_bytes() { return sizeof(NpLds<4, 7, 6>); }
void launch_glstm_np_fwd(const GPersistArgs& a, hipStream_t s) {
  gpersist_arm_bytes(a.gran2, gpersist_np_gran2_bytes(a), s);
  if (a.NT == 2) hipLaunchKernelGGL((k_glstm_np_fwd<2, 8, 8>), dim3(gp_grid(a)), dim3(GP_WAVES * 64), 0, s, a);
  else hipLaunchKernelGGL((k_glstm_np_fwd<4, 7, 6>), dim3(gp_grid(a)), dim3(GP_WAVES * 64), 0, s, a);
  ++g_chain_launches;
}
// (the unprojected BPTT: 8 cells per workgroup only; gran1 / gran3 = its two rings, armed once)
size_t gpersist_np_gran1_bytes(const GPersistArgs& a) { return (size_t)(a.N / GP_ROWS) * a.nl * GP_R1 * NP_NCH * a.NC * GP_SLOT; }
size_t gpersist_np_gran3_bytes(const GPersistArgs& a) { return (size_t)(a.N / GP_ROWS) * (a.nl + 1) * GP_XR * NP_NCH * a.NC * GP_SLOT; }
void gpersist_np_arm(const GPersistArgs& a, hipStream_t s) {
  (void)hipMemsetAsync(a.gran1, 0xFF, gpersist_np_gran1_bytes(a), s);
  (void)hipMemsetAsync(a.gran3, 0xFF, gpersist_np_gran3_bytes(a), s);
}
void launch_glstm_np_bwd(const GPersistArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_glstm_np_bwd, dim3(gp_grid(a)), dim3(GP_WAVES * 64), 0, s, a);
  ++g_chain_launches;
}
// ONE launch: the discriminator's trailing BPTT (d: DPersistArgs with the trailing fields) in front of the generator's (a.dout_trail = 1)
int gpersist_dt_grid(const GPersistArgs& a, const DPersistArgs& d) { return gp_grid(a) + ((dpersist_trail_grid(d.nl, d.N) + 7) & ~7); }
void launch_glstm_bwd_dt(const GPersistArgs& a, const DPersistArgs& d, hipStream_t s) {
  gp_arm_gran2(a, s);
  const dim3 g(gpersist_dt_grid(a, d)), b(GP_WAVES * 64);
  if (a.tags) {
    if (a.res) hipLaunchKernelGGL((k_glstm_bwd_dt<5, true, true>), g, b, 0, s, a, d);
    else hipLaunchKernelGGL((k_glstm_bwd_dt<5, false, true>), g, b, 0, s, a, d);
  } else if (a.res) hipLaunchKernelGGL((k_glstm_bwd_dt<5, true, false>), g, b, 0, s, a, d);
  else hipLaunchKernelGGL((k_glstm_bwd_dt<5, false, false>), g, b, 0, s, a, d);
  g_chain_launches += 2;
}
void launch_glstm_fwd_dt(const GPersistArgs& a, const DPersistArgs& d, hipStream_t s) {
  gp_arm_gran2(a, s);
  const dim3 g(gpersist_dt_grid(a, d)), b(GP_WAVES * 64);
  if (a.tags) {
    if (a.res) hipLaunchKernelGGL((k_glstm_fwd_dt<5, true, true>), g, b, 0, s, a, d);
    else hipLaunchKernelGGL((k_glstm_fwd_dt<5, false, true>), g, b, 0, s, a, d);
  } else if (a.res) hipLaunchKernelGGL((k_glstm_fwd_dt<5, true, false>), g, b, 0, s, a, d);
  else hipLaunchKernelGGL((k_glstm_fwd_dt<5, false, false>), g, b, 0, s, a, d);
  g_chain_launches += 2;
}
void launch_glstm_bwd(const GPersistArgs& a, hipStream_t s) {
  gp_arm_gran2(a, s);
  const dim3 g(gp_grid(a)), b(GP_WAVES * 64);
  if (a.tags && !GP_PROG_ONLY) {
    if (a.res) hipLaunchKernelGGL((k_glstm_bwd<5, 0, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_glstm_bwd<5, 0, false, true>), g, b, 0, s, a);
  } else if (a.res) hipLaunchKernelGGL((k_glstm_bwd<5, GP_PROG_ONLY, true, false>), g, b, 0, s, a);
  else hipLaunchKernelGGL((k_glstm_bwd<5, GP_PROG_ONLY, false, false>), g, b, 0, s, a);
  ++g_chain_launches;
}

}  // namespace rsr
