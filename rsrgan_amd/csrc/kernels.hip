// kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the RSRGAN GAN step, except the
// time-batched GEMM (gemm.hip).
//
// Recurrent LSTMP step kernels.  The reference runs tf.nn.dynamic_rnn over
// tf.contrib.rnn.LSTMCell (models/lstm.py:89-112, models/discriminator_lstm.py:70-91;
// cell math as restated in models/BNLSTMCell.py:176-217), i.e. per time step one
// [B,in+P]x[in+P,4H] GEMM, ~12 tiny elementwise kernels and one [B,H]x[H,P] GEMM.
// Here one step is TWO launches ("gates" and "proj"), each able to carry up to
// MAXJ (layer,t) jobs of a wavefront, because on MI355X a dependent kernel boundary
// (~1.5 us) is cheaper than any in-kernel grid barrier (>=4 us).
//
// MFMA use: the per-step GEMMs have M = batch rows (<=64 per tile column), so they use
// v_mfma_f32_16x16x4_f32 (exact fp32) with BOTH operands loaded k-contiguously as
// float4 straight from L2 into VGPRs (no LDS round trip: each operand element is used
// by one wave only).  Lane l holds A[row=l&15][k=4*(l>>4)+u] and B[k=4*(l>>4)+u][col=l&15]
// for the u-th of four consecutive MFMAs, which is why the forward pass keeps
// k-contiguous transposed copies of the kernels (KxT, KhT, WpT) while the backward pass
// reads the TF-layout originals (already k-contiguous for dz.K^T and dm.Wp^T).
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace rsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

// tools/ubench/trace.hip compiles this file with RSR_TRACE: thread 0 of every workgroup stamps the shader clock at the phase
// boundaries of the step kernels into g_trace[block][16] ([0] = 100 MHz real-time counter at entry, [1..9] = s_memtime stamps,
// [14] = HW_ID, [15] = XCC_ID); the product build has no such code
#ifdef RSR_TRACE
__device__ unsigned long long g_trace[8192 * 16];
#define TR_BEGIN() do { if (threadIdx.x == 0 && blockIdx.x < 8192) { unsigned long long* t_ = g_trace + (size_t)blockIdx.x * 16; \
    t_[0] = __builtin_amdgcn_s_memrealtime(); t_[1] = __builtin_amdgcn_s_memtime(); \
    t_[14] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); t_[15] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); \
    for (int i_ = 2; i_ < 14; ++i_) t_[i_] = 0; } } while (0)
#define TR_ID(ptr) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_trace[(size_t)blockIdx.x * 16 + 12] = (unsigned long long)(size_t)(ptr); } while (0)
#define TR(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_trace[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define TR_END() do { if (threadIdx.x == 0 && blockIdx.x < 8192) { g_trace[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime(); \
    g_trace[(size_t)blockIdx.x * 16 + 13] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TR_BEGIN() do { } while (0)
#define TR_ID(ptr) do { } while (0)
#define TR(i) do { } while (0)
#define TR_END() do { } while (0)
#endif

// The job record of a workgroup in ONE memory round trip: lane i of every wave loads dword i of the record (kernarg segment or a
// device table, <= 256 B) and v_readlane moves it into scalar registers.  Reading the fields straight from the by-value
// kernel argument costs 4-6 dependent scalar-cache round trips at the head of every workgroup (s_load as each group of fields
// is first needed, ~0.3-0.5 us each on a cold scalar cache: tools/ubench/trace.hip, 1.5-2 us of "issue" time per workgroup).
// (pointers rebuilt from raw dwords are generic to the compiler, which would make every access a flat_load/flat_store: the
// integer -> address-space-1 pointer -> generic round trip lets InferAddressSpaces turn them back into global accesses)
template <typename T>
__device__ __forceinline__ T* as_global(T* p) { return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)p; }
#define RSR_G(f) J.f = as_global(J.f);
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// DropSpec (kernels.h): the key of this (layer, t) in the current training run, and one element's mask
__device__ __forceinline__ unsigned long long drop_key(const DropSpec& d) {
  return splitmix64(splitmix64(d.seed ^ (*d.ctr * 0xD1342543DE82EF95ull)) + d.tag);
}
__device__ __forceinline__ bool drop_on(unsigned long long key, size_t idx, unsigned thr) { return (unsigned)(splitmix64(key + idx) >> 40) < thr; }
__device__ __forceinline__ void globalize(FwdGateJob& J) {
  RSR_G(x) RSR_G(KxT) RSR_G(m) RSR_G(KhT) RSR_G(Wsw) RSR_G(zx) RSR_G(bias) RSR_G(wf) RSR_G(wi) RSR_G(wo) RSR_G(c_prev) RSR_G(c_out) RSR_G(gates)
  RSR_G(h) RSR_G(len) RSR_G(np_m_out) RSR_G(np_out) RSR_G(np_res_in) RSR_G(np_res_out)
}
__device__ __forceinline__ void globalize(FwdProjJob& J) {
  RSR_G(h) RSR_G(WpT) RSR_G(WpT_sw) RSR_G(m_prev) RSR_G(m_out) RSR_G(out) RSR_G(res_in) RSR_G(res_out) RSR_G(len) RSR_G(bias) RSR_G(noise)
  J.drop.ctr = as_global(J.drop.ctr);
}
__device__ __forceinline__ void globalize(BwdAJob& J) {
  RSR_G(dout) RSR_G(dmst) RSR_G(Wp) RSR_G(Wp_sw) RSR_G(dmt) RSR_G(gates) RSR_G(c_prev) RSR_G(c_cur) RSR_G(wf) RSR_G(wi) RSR_G(wo) RSR_G(dc) RSR_G(len)
  J.drop.ctr = as_global(J.drop.ctr);
}
__device__ __forceinline__ void globalize(BwdBJob& J) { RSR_G(dz) RSR_G(K) RSR_G(Ksw) RSR_G(dx) RSR_G(dmst) RSR_G(len) RSR_G(ws) }
#undef RSR_G
// The job of block `bid` in ONE round trip: the records of ALL MAXJ jobs are requested at once (one dword per lane each) next to
// the scalar load of the block's entry of the job map; `lb` = the block's index inside its job from the job's Place (kernels.h;
// 6 dwords at OFF), or -1 when no job wants this block (idle slot of an XCD group).  Without a map (grid > JOBMAP_MAX) the
// places of all jobs are tested.
template <typename JobT, int OFF>
__device__ __forceinline__ JobT pick_job(const JobT* j, const int& n_ref, const JobMap& map, int bid, int& lb) {
  constexpr int ND = (int)(sizeof(JobT) / 4);
  static_assert(sizeof(JobT) % 4 == 0 && ND <= 64 && OFF + 5 < ND, "job record must fit one dword per lane");
  const int lane = threadIdx.x & 63;
  const int li = lane < ND ? lane : 0;
  unsigned v[MAXJ];
#pragma unroll
  for (int q = 0; q < MAXJ; ++q) v[q] = reinterpret_cast<const unsigned*>(j + q)[li];
  const int x = bid & 7, sr = bid >> 3;
  unsigned vs = v[0];
  lb = -1;
  if (map.valid) {                                       // (uniform)
    const int ji = map.job[min(bid, JOBMAP_MAX - 1)];
#pragma unroll
    for (int q = 1; q < MAXJ; ++q) vs = ji == q ? v[q] : vs;
    const int x0 = __builtin_amdgcn_readlane((int)vs, OFF), nx = __builtin_amdgcn_readlane((int)vs, OFF + 1);
    const int sb = __builtin_amdgcn_readlane((int)vs, OFF + 2);
    lb = ji == 0xFF ? -1 : (sr - sb) * nx + ((x - x0) & 7);
  } else {
    const int n = n_ref;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
      const int x0 = __builtin_amdgcn_readlane((int)v[q], OFF), nx = __builtin_amdgcn_readlane((int)v[q], OFF + 1);
      const int sb = __builtin_amdgcn_readlane((int)v[q], OFF + 2), se = __builtin_amdgcn_readlane((int)v[q], OFF + 3);
      const int nb = __builtin_amdgcn_readlane((int)v[q], OFF + 4);
      const int k = (x - x0) & 7;
      const int l = (sr - sb) * nx + k;
      const bool take = q < n && k < nx && sr >= sb && sr < se && l < nb;       // (uniform; the places of a launch are disjoint)
      vs = take ? v[q] : vs;
      lb = take ? l : lb;
    }
  }
  union U { JobT j; unsigned d[ND]; __device__ U() {} } u;
#pragma unroll
  for (int i = 0; i < ND; ++i) u.d[i] = (unsigned)__builtin_amdgcn_readlane((int)vs, i);
  globalize(u.j);
  return u.j;
}
#define RSR_PICK(JobT, field, lb) pick_job<JobT, (int)(__builtin_offsetof(JobT, field) / 4)>(jobs.j, jobs.n, jobs.map, bid, lb)
#define RSR_PICK_M(JobT, field, mapf, lb) pick_job<JobT, (int)(__builtin_offsetof(JobT, field) / 4)>(jobs.j, jobs.n, jobs.mapf, bid, lb)

template <typename JobT>
__device__ __forceinline__ JobT load_job(const JobT* p) {
  constexpr int ND = (int)(sizeof(JobT) / 4);
  static_assert(sizeof(JobT) % 4 == 0 && ND <= 64, "job record must fit one dword per lane");
  const int lane = threadIdx.x & 63;
  const unsigned v = reinterpret_cast<const unsigned*>(p)[lane < ND ? lane : 0];
  union U { JobT j; unsigned d[ND]; __device__ U() {} } u;
#pragma unroll
  for (int i = 0; i < ND; ++i) u.d[i] = (unsigned)__builtin_amdgcn_readlane((int)v, i);
  globalize(u.j);
  return u.j;
}

// Local block -> (column block, row block): pl.w = the job's column blocks rounded up to a multiple of its XCD slots, so
// lb % nx (= the XCD slot) == cb % nx for every row block and every time step: the slice of a weight matrix a column block
// streams stays in ONE XCD's 4 MiB L2 across the whole recurrence (block id -> XCD is observed behaviour, only speed depends on it).
__device__ __forceinline__ bool tile_of_block(int lb, int nblk_c, const Place& pl, int& cb, int& rb) {
  if (lb < 0) return false;
  rb = lb / pl.w;
  cb = lb - rb * pl.w;
  return cb < nblk_c;
}

// One tile product's operands.  RULE for every load of the step kernels: the address is always valid (row / column / k
// clamped by the caller) and the LOAD IS UNCONDITIONAL; what must not contribute is zeroed afterwards with a select.  A load
// under a branch makes the compiler close the branch with s_waitcnt vmcnt(0), i.e. one full memory round trip per load
// instead of one per batch (seen in the ISA of round 1's kernels: 6-18 serialised round trips per workgroup).
template <int RT>
struct Seg {
  const float* a[RT];   // A rows of this lane (row = l&15 of each 16-row tile, clamped to a real row), k-contiguous
  const float* w;       // W row of this lane (col = l&15, clamped), k-contiguous -- used when wt == nullptr
  const float* wt;      // or: this lane's float4 slot of k-block 0 of the fragment-tiled copy (kernels.h SwizzleJob), next k-block 256 floats on
  int ld;               // row length (multiple of 4); k >= ld contributes nothing
  int nkb;              // ceil(ld/16) k-blocks
  bool wok;             // this lane's output column exists (row-major W only; tiled copies hold zeros there)
};

// operands of the CH k-blocks [j0, j0 + CH) (blocks >= je: clamped loads, zero weights)
template <int RT, int CH, bool TILED>
__device__ __forceinline__ void load_chunk(float4 (&av)[CH][RT], float4 (&bv)[CH], const Seg<RT>& s, int j0, int je, int q) {
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int j = j0 + c, jc = min(j, s.nkb - 1);
    const int k = jc * 16 + 4 * q, kc = min(k, s.ld - 4);
    const float4 b = TILED ? *reinterpret_cast<const float4*>(s.wt + (size_t)jc * 256) : *reinterpret_cast<const float4*>(s.w + kc);
    const bool ok = (j < je) && (k < s.ld) && (TILED || s.wok);
    bv[c] = make_float4(ok ? b.x : 0.f, ok ? b.y : 0.f, ok ? b.z : 0.f, ok ? b.w : 0.f);
#pragma unroll
    for (int i = 0; i < RT; ++i) av[c][i] = *reinterpret_cast<const float4*>(s.a[i] + kc);
  }
}
template <int RT, int CH>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[RT], const float4 (&av)[CH][RT], const float4 (&bv)[CH]) {
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i].x, bv[c].x, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i].y, bv[c].y, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i].z, bv[c].z, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i].w, bv[c].w, acc[i], 0, 0, 0);
  }
}
// acc[i] (16x16, i < RT row tiles) += A_i[.,k] * W[.,k] over the k-blocks [jb,je).  All operand loads of a CH-k-block chunk
// are issued before the first MFMA of the chunk: one L2 round trip per chunk.
template <int RT, int CH, bool TILED>
__device__ __forceinline__ void mma_seg_t(f32x4 (&acc)[RT], const Seg<RT>& s, int jb, int je, int q) {
  for (int j0 = jb; j0 < je; j0 += CH) {
    float4 av[CH][RT], bv[CH];
    load_chunk<RT, CH, TILED>(av, bv, s, j0, je, q);
    __builtin_amdgcn_sched_barrier(0);        // all loads of the chunk are issued before its first MFMA (the scheduler would sink them)
    mma_chunk<RT, CH>(acc, av, bv);
  }
}
// Software-pipelined variant: two register sets; the loads of chunk c+1 are in flight while chunk c feeds the matrix pipe
// (the compiler's counted vmcnt waits only for the set it is about to use).
template <int RT, int CH, bool TILED>
__device__ __forceinline__ void mma_seg_pipe_t(f32x4 (&acc)[RT], const Seg<RT>& s, int jb, int je, int q) {
  float4 a0[CH][RT], b0[CH], a1[CH][RT], b1[CH];
  if (jb >= je) return;                                   // (wave-uniform)
  load_chunk<RT, CH, TILED>(a0, b0, s, jb, je, q);
  for (int j0 = jb; j0 < je; j0 += 2 * CH) {
    load_chunk<RT, CH, TILED>(a1, b1, s, j0 + CH, je, q);           // past je: clamped addresses, zero weights
    __builtin_amdgcn_sched_barrier(0);
    mma_chunk<RT, CH>(acc, a0, b0);
    load_chunk<RT, CH, TILED>(a0, b0, s, j0 + 2 * CH, je, q);
    __builtin_amdgcn_sched_barrier(0);
    mma_chunk<RT, CH>(acc, a1, b1);
  }
}
template <int RT, int CH>
__device__ __forceinline__ void mma_seg(f32x4 (&acc)[RT], const Seg<RT>& s, int jb, int je, int q) {
  if (s.wt) mma_seg_t<RT, CH, true>(acc, s, jb, je, q);             // (uniform per workgroup)
  else mma_seg_t<RT, CH, false>(acc, s, jb, je, q);
}
template <int RT, int CH>
__device__ __forceinline__ void mma_seg_pipe(f32x4 (&acc)[RT], const Seg<RT>& s, int jb, int je, int q) {
  if (s.wt) mma_seg_pipe_t<RT, CH, true>(acc, s, jb, je, q);
  else mma_seg_pipe_t<RT, CH, false>(acc, s, jb, je, q);
}

constexpr int RT = 2;          // 16-row tiles per wave: one W fragment feeds 2 x 4 MFMAs

// ---------------------------------------------------------------------------------------
// forward phase 1: WG (512 threads) = 32 rows x 16 cells; waves = 4 gates x 2 slices of
// K = [x_t | m_{t-1}].  One memory round trip per kernel:
//   - the 32-row A tile [x_t | m_{t-1}] goes global -> LDS by DMA (global_load_lds, no VGPRs) once
//     per WG and is shared by all 8 waves (it used to be re-loaded by every wave);
//   - every wave prefetches its whole weight slice (<= CHB k-blocks) into VGPRs;
//   - every thread prefetches the operands of the one (row, cell) it finishes in the epilogue.
// LDS row stride SA = K + 4 (or + 8) floats with SA/4 odd: the ds_read_b128 fragment reads of 16
// consecutive rows land on distinct 16-B slots.  72 KB at K = 560 -> 2 WGs per CU.
// ---------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS row stride (in float4) of an image whose rows are read as MFMA fragments with ds_read_b128 (lane (q, lr): row lr, float4 q of
// a k-block).  gfx950 serves a ds_read_b128 in four fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...), i.e. every
// group holds all 16 rows, eight of them with float4 q and eight with q + 1, and a group is conflict-free iff its 16 slots
// (address / 16 B) differ mod 16.  slot = lr * stride + 4c + q: stride == 2 (mod 16) puts the first eight on the even slots and the
// others on the odd ones.  An odd stride -- the usual padding rule -- collides 4 to 28 of the 64 lanes (SQ_LDS_BANK_CONFLICT was
// 50 % of SQ_LDS_IDX_ACTIVE in k_fwd_gates and k_bwd_bp, profiles/r2_final_pmc_lds_summary.txt).
__host__ __device__ inline int frag_stride4(int n4) { return n4 + ((18 - (n4 & 15)) & 15); }
__host__ __device__ inline int gates_sa4(int ktot) { return frag_stride4(ktot / 4 + 1); }

// RTG 16-row tiles per wave, RH row halves per workgroup: <.,2,1> = 32 rows x 16 cells on 8 waves (gate x K-half);
// <.,2,2> = 64 rows on 16 waves (gate x K-half x row-half): the two row halves request the same weight lines, so a column
// block's weight slice enters the CU once for 64 rows.
template <int CHB, int RTG, int RH = 1>
__global__ __launch_bounds__(512 * RH, (RH == 1 && RTG == 2 && CHB <= 18) ? 4 : 2) void k_fwd_gates(const FwdGateJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  TR_BEGIN();
  int lb;
  const FwdGateJob J = RSR_PICK(FwdGateJob, pl, lb);
  int cb, rb;
  if (!tile_of_block(lb, J.nblk_c, J.pl, cb, rb)) return;
  TR_ID(J.gates);
  TR(2);
  const int r0 = rb * 16 * RTG * RH, c0 = cb * 16;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, provably uniform -> scalar branches
  const int gate = w & 3, ks = (w >> 2) & 1, rh = w >> 3;
  const int H = J.H, N = J.N, H4 = 4 * H;
  const int ldx = J.x ? J.ldx : 0, ldm = J.ldm, Ktot = ldx + ldm;
  const int SA4 = gates_sa4(Ktot), SA = SA4 * 4;

  // (1) this wave's weight slice -> VGPRs (host guarantees <= CHB k-blocks per wave)
  // (one contiguous 1 KB tile per wave-load from the fragment-tiled copy: zero beyond H cells / Ktot columns)
  const int nkb = (Ktot + 15) >> 4, per = (nkb + 1) >> 1;
  const int jb = ks * per, je = min(nkb, jb + per);
  const float* wt = J.Wsw + (size_t)(gate * J.nblk_c + cb) * nkb * 256 + lane * 4;
  float4 bv[CHB];
#pragma unroll
  for (int c = 0; c < CHB; ++c)         // unconditional, clamped: the product below skips k-blocks >= je
    bv[c] = *reinterpret_cast<const float4*>(wt + (size_t)min(jb + c, nkb - 1) * 256);
  TR(10);
  // (2) A tile: lane p of the linear LDS image <- x / m element (rows >= N and pad columns get a
  // harmless finite dummy; they only ever meet zero weights or unstored rows)
  {
    const int P4 = 16 * RTG * RH * SA4;
    // (row, float4 column) of slot p advance by a fixed (dq, dr) per round: one integer division per thread, none per load
    constexpr int STEP = 512 * RH;
    const int dq = STEP / SA4, dr = STEP - dq * SA4;
    int row = (w * 64 + lane) / SA4, c4 = (w * 64 + lane) - row * SA4;
    const int ldx4 = ldx >> 2, kt4 = Ktot >> 2;
    for (int p0 = w * 64; p0 < P4; p0 += STEP) {
      const int arow = min(r0 + row, N - 1);
      const bool inx = c4 < ldx4, inm = c4 < kt4;
      // (32-bit element offsets and selects: the pointer-valued ?: compiled to a divergent branch per DMA round)
      const int ox = arow * ldx + c4 * 4, om = inm ? arow * ldm + (c4 - ldx4) * 4 : 0;
      const float* src = (inx ? J.x : J.m) + (inx ? ox : om);   // (pad slots and rows >= N: any finite word)
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (size_t)p0 * 4), 16, 0, 0);
      c4 += dr; row += dq;
      const bool wrap = c4 >= SA4;
      c4 -= wrap ? SA4 : 0; row += wrap ? 1 : 0;
    }
  }
  TR(11);
  // (3) epilogue operands of this thread's EPT (row, cell) elements, in flight together with (1) and (2)
  constexpr int EPT = (RTG + 1) / 2;                           // (RTG = 1: 16-row blocks, the upper half of the threads has no element)
  const int er = (tid >> 4) & 15, ec = tid & 15, ecell = c0 + ec;
  float zb[EPT][4], cp[EPT], pwi = 0.f, pwf = 0.f, pwo = 0.f;
  int elen[EPT];
  bool evalid[EPT];
  {
    const int ecl = min(ecell, H - 1);                        // every load below is unconditional, from a clamped address
    pwi = J.wi[ecl]; pwf = J.wf[ecl]; pwo = J.wo[ecl];
    const float* zsrc = J.zx ? J.zx : J.bias;                 // (uniform select of the base pointer)
    const size_t zrow = J.zx ? (size_t)H4 : 0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int erow = r0 + ((tid >> 8) + 2 * RH * u) * 16 + er;
      evalid[u] = (tid >> 8) + 2 * RH * u < RTG * RH && erow < N && ecell < H;
      const int erc = min(erow, N - 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) zb[u][g] = zsrc[(size_t)erc * zrow + g * H + ecl];
      cp[u] = J.c_prev[(size_t)erc * H + ecl];
      elen[u] = J.len[erc];
    }
  }
  TR(3);
  __syncthreads();
  TR(4);
  // (4) MFMAs: A fragments from LDS, B from registers; RTG independent accumulators interleave
  f32x4 acc[RTG];
#pragma unroll
  for (int i = 0; i < RTG; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* abase = smem + (size_t)(rh * 16 * RTG + lr) * SA + jb * 16 + 4 * q;
#pragma unroll
  for (int c = 0; c < CHB; ++c) {
    if (jb + c < je) {
      float4 a[RTG];
#pragma unroll
      for (int i = 0; i < RTG; ++i) a[i] = *reinterpret_cast<const float4*>(abase + (size_t)i * 16 * SA + c * 16);
#pragma unroll
      for (int i = 0; i < RTG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bv[c].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < RTG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bv[c].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < RTG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bv[c].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < RTG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bv[c].w, acc[i], 0, 0, 0);
    }
  }
  TR(5);
  __syncthreads();                                     // everyone is done reading the A tile
  TR(6);
  float (*zs)[RTG * RH][16][17] = reinterpret_cast<float (*)[RTG * RH][16][17]>(smem);      // zs[8][RTG*RH][16][17] aliases it
#pragma unroll
  for (int i = 0; i < RTG; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[w & 7][rh * RTG + i][q * 4 + r][lr] = acc[i][r];
  __syncthreads();
  TR(7);

#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    if (!evalid[u]) continue;
    const int ei = (tid >> 8) + 2 * RH * u;
    const int erow = r0 + ei * 16 + er;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = (zb[u][g] + zs[g][ei][er][ec]) + zs[4 + g][ei][er][ec];
    const size_t ci = (size_t)erow * H + ecell;
    float* g = J.gates + (size_t)erow * H4 + ecell;
    if (J.t < elen[u]) {
      const float cpv = cp[u];
      const float gi = sigmoid_(z[0] + pwi * cpv);
      const float gf = sigmoid_(z[2] + jobs.forget_bias + pwf * cpv);
      const float gj = tanhf(z[1]);
      const float cn = gf * cpv + gi * gj;
      const float go = sigmoid_(z[3] + pwo * cn);
      J.c_out[ci] = cn;
      g[0] = gi; g[H] = gj; g[2 * H] = gf; g[3 * H] = go;
      const float hh = go * tanhf(cn);
      J.h[(size_t)erow * J.ldh + ecell] = hh;
      if (J.np_m_out) {
        const size_t mi = (size_t)erow * J.ldm + ecell;
        J.np_m_out[mi] = hh;
        if (J.np_out) J.np_out[mi] = hh;
        if (J.np_res_out) J.np_res_out[mi] = hh + J.np_res_in[mi];
      }
    } else {                       // dynamic_rnn: t >= len -> state copied through, no gradient
      J.c_out[ci] = cp[u];
      g[0] = 0.f; g[H] = 0.f; g[2 * H] = 0.f; g[3 * H] = 0.f;
      J.h[(size_t)erow * J.ldh + ecell] = 0.f;
      if (J.np_m_out) {
        const size_t mi = (size_t)erow * J.ldm + ecell;
        J.np_m_out[mi] = J.m[mi];
        if (J.np_out) J.np_out[mi] = 0.f;
        if (J.np_res_out) J.np_res_out[mi] = J.np_res_in[mi];
      }
    }
  }
  TR_END();
}

// ---------------------------------------------------------------------------------------
// forward phase 2: 32x16 tile of m_t = h_t.Wp, K (=H) split over NW waves
// ---------------------------------------------------------------------------------------
template <int NW, bool DROP = false>          // DROP: some job of the launch has a DropSpec (a separate instantiation: the default one is untouched)
__global__ __launch_bounds__(64 * NW) void k_fwd_proj(const FwdProjJobs jobs) {
  __shared__ float zs[NW][RT][16][17];
  const int bid = blockIdx.x;
  TR_BEGIN();
  int lb;
  const FwdProjJob J = RSR_PICK(FwdProjJob, pl, lb);
  int cb, rb;
  if (!tile_of_block(lb, J.nblk_c, J.pl, cb, rb)) return;
  TR_ID(J.h);
  const int r0 = rb * 16 * RT, c0 = cb * 16;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, provably uniform -> scalar branches
  const int N = J.N, P = J.P;
  const int p = c0 + lr;
  Seg<RT> s0;
  s0.ld = J.ldh; s0.nkb = (J.ldh + 15) >> 4;
  s0.w = J.WpT + (size_t)min(p, P - 1) * J.ldh; s0.wok = p < P;
  s0.wt = J.WpT_sw ? J.WpT_sw + (size_t)cb * s0.nkb * 256 + lane * 4 : nullptr;
#pragma unroll
  for (int i = 0; i < RT; ++i) s0.a[i] = J.h + (size_t)min(r0 + i * 16 + lr, N - 1) * J.ldh;
  // epilogue operands of this thread's elements, requested before the product (unconditional loads from clamped / stand-in
  // addresses: m_out is always a valid [N][ldm] buffer)
  constexpr int EPT = (RT * 256) / (64 * NW);
  float e_mprev[EPT], e_bias[EPT], e_noise[EPT], e_res[EPT];
  int e_len[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int e = tid + u * 64 * NW;
    const int i = e >> 8, er = (e >> 4) & 15, ec = e & 15;
    const int row = min(r0 + i * 16 + er, N - 1), pp = min(c0 + ec, P - 1);
    const size_t mi = (size_t)row * J.ldm + pp;
    e_mprev[u] = (J.m_prev ? J.m_prev : J.m_out)[mi];
    e_bias[u] = (J.bias ? J.bias : J.WpT)[pp];
    e_len[u] = (J.len ? J.len : reinterpret_cast<const int*>(J.WpT))[row];
    e_noise[u] = (J.noise ? J.noise : J.m_out)[(size_t)row * P + pp];      // (P <= ldm: inside the stand-in too)
    e_res[u] = (J.res_in ? J.res_in : J.m_out)[mi];
  }
  const int per = (s0.nkb + NW - 1) / NW;
  f32x4 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  TR(2);
  mma_seg<RT, 6>(acc, s0, w * per, min(s0.nkb, (w + 1) * per), q);
  TR(5);
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[w][i][q * 4 + r][lr] = acc[i][r];
  __syncthreads();
  TR(7);
  const unsigned long long dkey = (DROP && J.drop.ctr) ? drop_key(J.drop) : 0ull;       // (uniform)
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int e = tid + u * 64 * NW;
    const int i = e >> 8, er = (e >> 4) & 15, ec = e & 15;
    const int row = r0 + i * 16 + er, pp = c0 + ec;
    if (row >= N || pp >= P) continue;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < NW; ++s) v += zs[s][i][er][ec];
    const size_t mi = (size_t)row * J.ldm + pp;
    if (J.bias) v += e_bias[u];
    const bool live = J.len ? (J.t < e_len[u]) : true;
    J.m_out[mi] = live ? v : e_mprev[u];                 // the carried state is never dropped
    float o = live ? v : 0.f;
    if (DROP && J.drop.ctr) o = drop_on(dkey, (size_t)row * P + pp, J.drop.thr) ? o / J.drop.keep : 0.f;
    J.out[(size_t)row * J.ldo + pp] = J.noise ? o + e_noise[u] : o;
    if (J.res_out) J.res_out[mi] = o + e_res[u];
  }
  TR_END();
}

// ---------------------------------------------------------------------------------------
// backward phase A: dh = (mask*(dout+dm_state)).Wp^T, then the cell's gate gradients.  WG = 4 waves on a 32-row x 32-cell tile.
// (Round 1's form -- 16-cell column blocks, every wave re-loading the dm rows as MFMA fragments, an 8-way K split reduced through
// LDS -- spent its ~6 us blocks on instruction issue and latency, tools/ubench/trace.hip; removed in round 3.)  Here
//   - the operand dm = mask.(dout + dmst) [32 x P] is summed ONCE per workgroup from coalesced float4 loads into an LDS image
//     (the column-block-0 workgroup also stores it as dm_t for the weight gradients),
//   - every wave owns one 16x16 output tile with the full K = P in two interleaved accumulators (no cross-wave reduction),
//     weights from the fragment-tiled copy (one contiguous 1 KB wave-load per k-block),
//   - the cell gradients are finished in the accumulator layout (lane: rows 4q..4q+3, cell lr), their operands prefetched
//     before the product.
// ---------------------------------------------------------------------------------------
template <int CHB, bool DROP = false>
__global__ __launch_bounds__(256) void k_bwd_a2(const BwdAJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  TR_BEGIN();
  int lb;
  const BwdAJob J = RSR_PICK(BwdAJob, pl, lb);
  int cb, rb;
  if (!tile_of_block(lb, J.nblk_c, J.pl, cb, rb)) return;
  TR_ID(J.gates);
  TR(2);
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = w & 1, ct = w >> 1;
  const int N = J.N, H = J.H, H4 = 4 * H, ldm = J.ldm;
  const int r0 = rb * 32, c0 = cb * 32;
  const bool noproj = J.Wp == nullptr;          // num_proj=None: dh = mask*(dout + dm_state), no product
  // LDS image of dm: [32 rows][SA4 float4], SA4 odd (conflict-free ds_read_b128 fragment reads); without a projection only the
  // tile's own 32 columns are staged
  // (rows are padded with zeros up to whole k-blocks: the last fragment read of a row must not meet its neighbour or stale LDS)
  const int nkb = (ldm + 15) >> 4;
  const int k4_0 = noproj ? c0 >> 2 : 0, nk4 = noproj ? 8 : ldm >> 2, nk4p = noproj ? 8 : nkb * 4;
  const int SA4 = frag_stride4(nk4p), SA = SA4 * 4;
  // Every load of the workgroup is issued first -- (1) the dm operand, (2) the weight tiles, (3) the epilogue operands -- and only
  // then is (1) consumed: loads return in order, so staging dm into LDS overlaps the landing of (2) and (3).
  // (1) dm: thread = (row tid/8, float4 slot tid%8 + 8 i)
  const int srow = tid >> 3, s8 = tid & 7;
  const int grow = min(r0 + srow, N - 1);
  const float* pm = J.dmst + (size_t)grow * ldm;
  const float* pd = (J.dout ? J.dout : J.dmst) + (size_t)grow * ldm;        // (stand-in when there is no dout: zeroed below)
  constexpr int NI = (CHB * 4 + 7) / 8;
  const int k4max = (ldm >> 2) - 1;
  float4 va[NI], vd[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {          // unconditional loads from clamped slots (one round trip for all of them)
    const int g4 = min(k4_0 + s8 + 8 * i, k4max);
    va[i] = *reinterpret_cast<const float4*>(pm + g4 * 4);
    vd[i] = *reinterpret_cast<const float4*>(pd + g4 * 4);
  }
  const int slen = J.len[grow];                                             // (unconditional: no branch around a load)
  TR(10);
  // (2) this wave's weight tiles: column tile cb*2 + ct of the fragment-tiled copy, all k-blocks
  float4 bv[CHB];
  if (!noproj) {                 // (uniform) unconditional loads from clamped tiles: the product skips k-blocks >= nkb, and
    const int tile_c = min(cb * 2 + ct, ((H + 15) >> 4) - 1);          // cells past H are never stored
    const float* wt = J.Wp_sw + (size_t)tile_c * nkb * 256 + lane * 4;
#pragma unroll
    for (int c = 0; c < CHB; ++c) bv[c] = *reinterpret_cast<const float4*>(wt + (size_t)min(c, nkb - 1) * 256);
  } else {
#pragma unroll
    for (int c = 0; c < CHB; ++c) bv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // (3) epilogue operands of this lane's four (row, cell) elements
  const int ecell = c0 + ct * 16 + lr;
  const bool cok = ecell < H;
  const int ecl = min(ecell, H - 1);
  const float ewo = J.wo[ecl], ewi = J.wi[ecl], ewf = J.wf[ecl];
  float eg[4][4], ecp[4], ecn[4], edc[4];
  int elen[4];
  bool erok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = r0 + rt * 16 + 4 * q + e;
    erok[e] = row < N && cok;
    const int rowc = min(row, N - 1);
    elen[e] = J.len[rowc];
    const float* g = J.gates + (size_t)rowc * H4 + ecl;
    eg[e][0] = g[0]; eg[e][1] = g[H]; eg[e][2] = g[2 * H]; eg[e][3] = g[3 * H];
    const size_t ci = (size_t)rowc * H + ecl;
    ecp[e] = J.c_prev[ci]; ecn[e] = J.c_cur[ci]; edc[e] = J.dc[ci];
  }
  __builtin_amdgcn_sched_barrier(0);       // keep the issue order above: nothing below may be hoisted between the loads
  // (1') dm = mask.(dout + dmst) -> LDS image (and dm_t for the weight gradients from the column-block-0 workgroups)
  {
    const bool live = (r0 + srow < N) & (J.t < slen);
    const float dscale = J.dout ? 1.f : 0.f;
    if (DROP && J.drop.ctr && J.dout) {                  // (uniform) dout through the forward's dropout: mask / keep per element
      const unsigned long long dkey = drop_key(J.drop);
      const float ik = 1.f / J.drop.keep;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int g4 = min(k4_0 + s8 + 8 * i, k4max);
        const size_t e0 = (size_t)grow * J.P + g4 * 4;
        vd[i].x = (g4 * 4 + 0 < J.P && drop_on(dkey, e0 + 0, J.drop.thr)) ? vd[i].x * ik : 0.f;
        vd[i].y = (g4 * 4 + 1 < J.P && drop_on(dkey, e0 + 1, J.drop.thr)) ? vd[i].y * ik : 0.f;
        vd[i].z = (g4 * 4 + 2 < J.P && drop_on(dkey, e0 + 2, J.drop.thr)) ? vd[i].z * ik : 0.f;
        vd[i].w = (g4 * 4 + 3 < J.P && drop_on(dkey, e0 + 3, J.drop.thr)) ? vd[i].w * ik : 0.f;
      }
    }
    float* pt = (cb == 0 && !noproj && r0 + srow < N) ? J.dmt + (size_t)grow * ldm : nullptr;
    float4* dst = reinterpret_cast<float4*>(smem) + (size_t)srow * SA4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c4 = s8 + 8 * i;
      const bool in = live && c4 < nk4 && k4_0 + c4 <= k4max;       // slots past the row's data: zeros
      float4 v = make_float4(va[i].x + dscale * vd[i].x, va[i].y + dscale * vd[i].y, va[i].z + dscale * vd[i].z, va[i].w + dscale * vd[i].w);
      v = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
      if (c4 < nk4p) dst[c4] = v;
      if (pt && c4 < nk4) *reinterpret_cast<float4*>(pt + c4 * 4) = v;
    }
  }
  TR(3);
  __syncthreads();
  TR(4);
  // (4) product: dh[16 x 16] = dm[rows rt*16.., :] . Wp[cells, :]^T
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float dh[4];
  if (!noproj) {
    const float* ab = smem + (size_t)(rt * 16 + lr) * SA + 4 * q;
#pragma unroll
    for (int c = 0; c < CHB; c += 2) {
      if (c < nkb) {
        const float4 a = *reinterpret_cast<const float4*>(ab + c * 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv[c].x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv[c].y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv[c].z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv[c].w, acc0, 0, 0, 0);
      }
      if (c + 1 < CHB && c + 1 < nkb) {
        const float4 a = *reinterpret_cast<const float4*>(ab + (c + 1) * 16);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv[c + 1].x, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv[c + 1].y, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv[c + 1].z, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv[c + 1].w, acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dh[e] = acc0[e] + acc1[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) dh[e] = smem[(size_t)(rt * 16 + 4 * q + e) * SA + ct * 16 + lr];
  }
  TR(5);
  // (5) gate / cell gradients (C layout of the MFMA: lane holds rows 4q + e of column lr)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (!erok[e]) continue;
    const int row = r0 + rt * 16 + 4 * q + e;
    float* g = J.gates + (size_t)row * H4 + ecell;
    if (J.t < elen[e]) {
      const float gi = eg[e][0], gj = eg[e][1], gf = eg[e][2], go = eg[e][3];
      const float tc = tanhf(ecn[e]);
      const float dao = dh[e] * tc * go * (1.f - go);
      const float dcn = edc[e] + dh[e] * go * (1.f - tc * tc) + dao * ewo;
      const float daf = dcn * ecp[e] * gf * (1.f - gf);
      const float dai = dcn * gj * gi * (1.f - gi);
      const float dj = dcn * gi * (1.f - gj * gj);
      J.dc[(size_t)row * H + ecell] = dcn * gf + dai * ewi + daf * ewf;
      g[0] = dai; g[H] = dj; g[2 * H] = daf; g[3 * H] = dao;
    } else {
      g[0] = 0.f; g[H] = 0.f; g[2 * H] = 0.f; g[3 * H] = 0.f;     // dc passes through unchanged
    }
  }
  TR_END();
}

// ---------------------------------------------------------------------------------------
// backward phase B: 32x16 tile of dz_t.K^T, K (=4H) split over NW waves
// ---------------------------------------------------------------------------------------
template <int NW, int VAR = 0, int CH = 6, int RTB = RT>
__global__ __launch_bounds__(NW * 64) void k_bwd_b(const BwdBJobs jobs) {
  __shared__ float zs[NW][RTB][16][17];
  const int bid = blockIdx.x;
  int lb;
  const BwdBJob J = RSR_PICK(BwdBJob, pl, lb);
  int cb, rb;
  if (!tile_of_block(lb, J.nblk_c, J.pl, cb, rb)) return;
  const int r0 = rb * 16 * RTB, n0 = J.n_begin + cb * 16;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, provably uniform -> scalar branches
  const int N = J.N, H4 = J.H4;
  const int n = n0 + lr;
  Seg<RTB> s0;
  s0.ld = H4; s0.nkb = (H4 + 15) >> 4;
  s0.w = J.K + (size_t)min(n, J.n_end - 1) * H4; s0.wok = n < J.n_end;
  s0.wt = J.Ksw ? J.Ksw + (size_t)cb * s0.nkb * 256 + lane * 4 : nullptr;
#pragma unroll
  for (int i = 0; i < RTB; ++i) s0.a[i] = J.dz + (size_t)min(r0 + i * 16 + lr, N - 1) * H4;
  // epilogue read-modify-write operand, prefetched (first RTB*256 threads own one output each): unconditional load from the
  // clamped destination, kept only where the old value takes part (dx += ; masked rows pass the carried gradient through)
  const int e_i = tid >> 8, e_r = (tid >> 4) & 15, e_c = tid & 15;
  const int erow = r0 + e_i * 16 + e_r, enn = n0 + e_c;
  const bool evalid = tid < RTB * 256 && erow < N && enn < J.n_end;
  const int erc = min(erow, N - 1), enc = min(enn, J.n_end - 1);
  const bool to_dx = enc < J.I;
  float* edst = to_dx ? J.dx + (size_t)erc * J.lddx + enc : J.dmst + (size_t)erc * J.ldm + (enc - J.I);
  float eold = *edst;
  const int elen = (J.len ? J.len : reinterpret_cast<const int*>(J.dz))[erc];
  // (bitwise, not && / ?: -- short-circuit control flow lets the optimizer sink the loads above back under a branch)
  const bool keep = (to_dx & (J.dx_accumulate != 0)) | (!to_dx & (J.len != nullptr) & !(J.t < elen));
  eold = keep ? eold : 0.f;
  const int per = (s0.nkb + NW - 1) / NW;
  f32x4 acc[RTB];
#pragma unroll
  for (int i = 0; i < RTB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (VAR & 8) mma_seg<RTB, CH>(acc, s0, w * per, min(s0.nkb, (w + 1) * per), q);      // un-pipelined
  else mma_seg_pipe<RTB, CH>(acc, s0, w * per, min(s0.nkb, (w + 1) * per), q);
#pragma unroll
  for (int i = 0; i < RTB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[w][i][q * 4 + r][lr] = acc[i][r];
  __syncthreads();
  if (evalid) {
    float v = eold;
#pragma unroll
    for (int s2 = 0; s2 < NW; ++s2) v += zs[s2][e_i][e_r][e_c];
    *edst = v;
  }
}

// ---------------------------------------------------------------------------------------
// backward phase B, split-K form.  The per-CU operand pull rate (~12 B/clk) bounds these launches, so the
// tiling minimises bytes: a WG owns ALL 64 rows x 64 output columns x one of KG slices of K (=4H): every
// weight byte is pulled by exactly one WG per launch, the dz slice [64 x kpg*16] goes global -> LDS by DMA
// once and is shared by the 8 waves (4 column tiles x 2 K halves).  Partial tiles land in ws[KG][N][ldw];
// the kernel boundary makes them coherent, and k_bwd_b_red sums them in a fixed order and applies the
// masked epilogue.  123 MB -> ~42 MB of operand traffic per generator-wave launch.
// ---------------------------------------------------------------------------------------
constexpr int BP_RT = 4, BP_CHB = 12;
__host__ __device__ inline int bp_sa4(int kpg) { return frag_stride4(kpg * 4); }      // LDS row stride in float4 (== 2 mod 16)

__global__ __launch_bounds__(512) void k_bwd_bp(const BwdBJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  TR_BEGIN();
  int lb;
  const BwdBJob J = RSR_PICK(BwdBJob, plp, lb);
  if (lb < 0) return;
  TR_ID(J.dz);
  // K slice minor: KG is a multiple of the job's XCD slots, so lb % nx = slice % nx: every workgroup of a slice runs on ONE XCD
  // and that XCD's L2 only ever holds its own share of K (and of dz) across the whole recurrence
  const int kg = lb % J.KG, rem = lb / J.KG;
  const int rg = rem / J.ncg, cg = rem - rg * J.ncg;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = w & 3, kh = w >> 2;
  const int N = J.N, H4 = J.H4, kpg = J.kpg;
  const int nkb = (H4 + 15) >> 4;
  const int j_begin = kg * kpg, j_end = min(nkb, j_begin + kpg);
  const int per = (kpg + 1) >> 1;
  const int jb = j_begin + kh * per, je = min(j_end, jb + per);
  const int r0 = rg * 64;
  const int SA4 = bp_sa4(kpg), SA = SA4 * 4;

  // (1) weight slice of this wave's column tile -> VGPRs
  // (fragment-tiled copy of K's rows [n_begin, n_end): unconditional loads from clamped tiles -- the product skips k-blocks
  //  >= je and columns past n_end are never stored)
  float4 bv[BP_CHB];
  if (J.Ksw) {                                           // (uniform per workgroup: two straight-line load sequences)
    const int tb = min(cg * 4 + ct, ((J.n_end - J.n_begin + 15) >> 4) - 1);      // 16-row block of K counted from n_begin
    const float* wt = J.Ksw + (size_t)tb * nkb * 256 + lane * 4;
#pragma unroll
    for (int c = 0; c < BP_CHB; ++c) bv[c] = *reinterpret_cast<const float4*>(wt + (size_t)min(jb + c, nkb - 1) * 256);
  } else {                                               // row-major K (per-step fully_connected stages): clamped row and k
    const float* wrow = J.K + (size_t)min(J.n_begin + cg * 64 + ct * 16 + lr, J.n_end - 1) * H4;
#pragma unroll
    for (int c = 0; c < BP_CHB; ++c) {
      const int k = (jb + c) * 16 + 4 * q;
      const float4 b = *reinterpret_cast<const float4*>(wrow + min(k, H4 - 4));
      const bool ok = k < H4;
      bv[c] = make_float4(ok ? b.x : 0.f, ok ? b.y : 0.f, ok ? b.z : 0.f, ok ? b.w : 0.f);
    }
  }
  // (2) dz slice [64 rows][kpg*16] -> LDS (rows >= N / columns past H4 get a finite dummy: they meet zero weights
  //     or unstored rows only)
  {
    const int P4 = 64 * SA4;
    const int dq = 512 / SA4, dr = 512 - dq * SA4;       // slot p -> (row, float4 column) advances by (dq, dr) per round
    int row = (w * 64 + lane) / SA4, c4 = (w * 64 + lane) - row * SA4;
    const int k4max = (H4 >> 2) - 1 - j_begin * 4;       // last float4 column of dz inside this K slice
    for (int p0 = w * 64; p0 < P4; p0 += 512) {
      const float* src = J.dz + (size_t)min(r0 + row, N - 1) * H4 + (size_t)(j_begin * 16 + min(c4, k4max) * 4);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (size_t)p0 * 4), 16, 0, 0);
      c4 += dr; row += dq;
      const bool wrap = c4 >= SA4;
      c4 -= wrap ? SA4 : 0; row += wrap ? 1 : 0;
    }
  }
  TR(3);
  __syncthreads();
  TR(4);
  // (3) MFMAs
  f32x4 acc[BP_RT];
#pragma unroll
  for (int i = 0; i < BP_RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* abase = smem + (size_t)lr * SA + (jb - j_begin) * 16 + 4 * q;
#pragma unroll
  for (int c = 0; c < BP_CHB; ++c) {
    if (jb + c < je) {
      float4 a[BP_RT];
#pragma unroll
      for (int i = 0; i < BP_RT; ++i) a[i] = *reinterpret_cast<const float4*>(abase + (size_t)i * 16 * SA + c * 16);
#pragma unroll
      for (int i = 0; i < BP_RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bv[c].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < BP_RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bv[c].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < BP_RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bv[c].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < BP_RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bv[c].w, acc[i], 0, 0, 0);
    }
  }
  TR(5);
  __syncthreads();                                       // done reading the dz slice
  TR(6);
  float (*zs)[BP_RT][16][17] = reinterpret_cast<float (*)[BP_RT][16][17]>(smem);     // zs[8][4][16][17] aliases it
#pragma unroll
  for (int i = 0; i < BP_RT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[w][i][q * 4 + r][lr] = acc[i][r];
  __syncthreads();
  // (4) partial tile [64 x 64] = sum of the two K halves -> ws[kg]
  const int ncols = J.n_end - J.n_begin;
  for (int e = tid; e < 64 * 64; e += 512) {
    const int row = e >> 6, col = e & 63;
    const int grow = r0 + row, gcol = cg * 64 + col;
    if (grow < N && gcol < ncols) {
      const int i = row >> 4, rr = row & 15, t2 = col >> 4, cc = col & 15;
      J.ws[((size_t)kg * N + grow) * J.ldw + gcol] = zs[t2][i][rr][cc] + zs[4 + t2][i][rr][cc];
    }
  }
  TR_END();
}

__global__ __launch_bounds__(256) void k_bwd_b_red(const BwdBJobs jobs) {
  const int bid = blockIdx.x;
  int lb;
  const BwdBJob J = RSR_PICK_M(BwdBJob, plr, mapr, lb);
  if (lb < 0) return;
  const int ncols = J.n_end - J.n_begin;
  const int e = lb * 256 + threadIdx.x;
  if (e >= J.N * ncols) return;
  const int row = e / ncols, col = e - row * ncols;
  const int nn = J.n_begin + col;
  // every load first (the KG partials, the old value, the row's length: independent, one round trip), then the fixed-order sum
  const bool to_dx = nn < J.I;
  float* dst = to_dx ? J.dx + (size_t)row * J.lddx + nn : J.dmst + (size_t)row * J.ldm + (nn - J.I);
  const float old = *dst;
  const int lenv = (J.len ? J.len : reinterpret_cast<const int*>(J.ws))[row];     // (fully_connected stages have no lengths)
  constexpr int KGMAX = 16;
  float part[KGMAX];
#pragma unroll
  for (int g = 0; g < KGMAX; ++g) part[g] = J.ws[((size_t)min(g, J.KG - 1) * J.N + row) * J.ldw + col];
  const bool keep = (to_dx & (J.dx_accumulate != 0)) | (!to_dx & (J.len != nullptr) & !(J.t < lenv));   // masked row: the carried gradient passes through
  float v = keep ? old : 0.f;
#pragma unroll
  for (int g = 0; g < KGMAX; ++g) v += g < J.KG ? part[g] : 0.f;
  *dst = v;
}

// ---------------------------------------------------------------------------------------
// placement of the jobs of a launch (kernels.h Place)
// ---------------------------------------------------------------------------------------
static int g_xcd_groups = -1;          // RSRGAN_XCD_GROUPS=0: every job spans all 8 XCD slots (round 1's contiguous layout)
// phase 1: XCD slots per job.  A job worth >= 12 % of the launch's work gets a group of slots of its own, sized by its share
// (largest remainders, at least one, 8 in all); the others ("fillers") span all 8 slots after the groups' rounds.
static void plan_groups(int n, const double* cost, int* nx, int* x0, bool* grouped, double balance_tol = 1.35) {
  if (g_xcd_groups < 0) { const char* e = getenv("RSRGAN_XCD_GROUPS"); g_xcd_groups = e ? atoi(e) : 1; }
  double total = 0.0;
  for (int j = 0; j < n; ++j) total += cost[j];
  int ng = 0;
  double cg = 0.0;
  for (int j = 0; j < n; ++j) { grouped[j] = g_xcd_groups && total > 0.0 && cost[j] >= 0.12 * total; ng += grouped[j]; cg += grouped[j] ? cost[j] : 0.0; }
  if (ng < 2 || ng > 8) {
    for (int j = 0; j < n; ++j) { grouped[j] = false; nx[j] = 8; x0[j] = 0; }
    return;
  }
  double frac[MAXJ];
  int sum = 0;
  for (int j = 0; j < n; ++j) {
    nx[j] = 8; x0[j] = 0; frac[j] = -1.0;
    if (!grouped[j]) continue;
    const double ideal = 8.0 * cost[j] / cg;
    nx[j] = std::max(1, (int)ideal);
    frac[j] = ideal - nx[j];
    sum += nx[j];
  }
  while (sum < 8) {                      // hand the remaining slots to the largest remainders
    int best = -1;
    for (int j = 0; j < n; ++j) if (grouped[j] && (best < 0 || frac[j] > frac[best])) best = j;
    nx[best]++; frac[best] -= 1.0; ++sum;
  }
  while (sum > 8) {                      // (only when many small groups were rounded up to one slot each)
    int best = -1;
    for (int j = 0; j < n; ++j) if (grouped[j] && nx[j] > 1 && (best < 0 || frac[j] < frac[best])) best = j;
    if (best < 0) break;
    nx[best]--; frac[best] += 1.0; --sum;
  }
  // a group must not carry more than its share: the busiest XCD's work within balance_tol of an even spread, else no groups
  double worst = 0.0;
  for (int j = 0; j < n; ++j) if (grouped[j]) worst = std::max(worst, cost[j] / nx[j]);
  if (worst > balance_tol * cg / 8.0) {
    for (int j = 0; j < n; ++j) { grouped[j] = false; nx[j] = 8; x0[j] = 0; }
    return;
  }
  int x = 0;
  for (int j = 0; j < n; ++j) if (grouped[j]) { x0[j] = x; x += nx[j]; }
}
// phase 2: rounds.  Groups start at round 0 side by side; fillers follow, each on all 8 slots.  Returns the grid size.
static int plan_rounds(int n, const int* nb, const int* w, const int* nx, const int* x0, const bool* grouped, Place** out) {
  int cur = 0;
  for (int j = 0; j < n; ++j)
    if (grouped[j]) { *out[j] = Place{x0[j], nx[j], 0, (nb[j] + nx[j] - 1) / nx[j], nb[j], w[j]}; cur = std::max(cur, out[j]->se); }
  for (int j = 0; j < n; ++j)
    if (!grouped[j]) { const int r = (nb[j] + 7) / 8; *out[j] = Place{0, 8, cur, cur + r, nb[j], w[j]}; cur += r; }
  return 8 * std::max(cur, 1);
}
int plan_places(int n, const PlanItem* items, Place* out) {
  double cost[MAXJ]; int nx[MAXJ], x0[MAXJ], nb[MAXJ], w[MAXJ]; bool grouped[MAXJ]; Place* po[MAXJ];
  for (int j = 0; j < n; ++j) cost[j] = items[j].cost;
  plan_groups(n, cost, nx, x0, grouped);
  for (int j = 0; j < n; ++j) {
    w[j] = (items[j].ncol + nx[j] - 1) / nx[j] * nx[j];
    nb[j] = w[j] * items[j].nrow;
    po[j] = out + j;
  }
  return plan_rounds(n, nb, w, nx, x0, grouped, po);
}
// block id -> job table of a launch from the jobs' places
template <typename GetPlace>
static void fill_map(JobMap& m, int n, int grid, GetPlace pl) {
  m.valid = grid <= JOBMAP_MAX;
  if (!m.valid) return;
  for (int b = 0; b < grid; ++b) {
    const int x = b & 7, sr = b >> 3;
    unsigned char ji = 0xFF;
    for (int q = 0; q < n; ++q) {
      const Place& p = pl(q);
      const int k = (x - p.x0) & 7, l = (sr - p.sb) * p.nx + k;
      if (k < p.nx && sr >= p.sb && sr < p.se && l < p.nb) { ji = (unsigned char)q; break; }
    }
    m.job[b] = ji;
  }
}
// tile kernels: column blocks x row blocks of `rows` rows, cost ~ blocks x K
template <typename JobsT, typename CostF>
static int place_tiles(JobsT& jobs, int rows, CostF kcost) {
  PlanItem it[MAXJ]; Place pl[MAXJ];
  for (int i = 0; i < jobs.n; ++i) {
    it[i].ncol = jobs.j[i].nblk_c; it[i].nrow = (jobs.j[i].N + rows - 1) / rows; it[i].mult = 0;
    it[i].cost = (double)it[i].ncol * it[i].nrow * kcost(jobs.j[i]);
  }
  const int grid = plan_places(jobs.n, it, pl);
  for (int i = 0; i < jobs.n; ++i) jobs.j[i].pl = pl[i];
  fill_map(jobs.map, jobs.n, grid, [&](int q) -> const Place& { return jobs.j[q].pl; });
  return grid;
}

size_t bwd_b_plan(BwdBJobs& jobs, float* ws_base) {
  size_t off = 0;
  const int n = jobs.n;
  double cost[MAXJ]; int nx[MAXJ], x0[MAXJ], nb[MAXJ], w1[MAXJ], nbr[MAXJ]; bool grouped[MAXJ], nogroup[MAXJ]; Place* pp[MAXJ]; Place* pr[MAXJ];
  for (int i = 0; i < n; ++i) cost[i] = (double)(jobs.j[i].n_end - jobs.j[i].n_begin) * jobs.j[i].N * jobs.j[i].H4;
  plan_groups(n, cost, nx, x0, grouped);
  // a group's workgroups must fit ONE round at one per CU (32 CUs per XCD): three equal layers split 3 / 3 / 2 put 36 on the
  // 2-XCD group's CUs and the launch took 23 us instead of 14 -- then no groups (slice = block id % 8 keeps its K-slice affinity)
  static int bp_groups = -1;
  if (bp_groups < 0) { const char* e = getenv("RSRGAN_BP_GROUPS"); bp_groups = e ? atoi(e) : 1; }
  if (!bp_groups) for (int i = 0; i < n; ++i) { grouped[i] = false; nx[i] = 8; x0[i] = 0; }
  constexpr int kpg_target = 24;     // k-blocks per K slice: 24 -> 99 KB LDS (1 WG/CU); 11 / 12 measured slower; round 3:
                                     // 22 / 20 / 19 (243-270 workgroups per generator diagonal) 8.02 / 8.24 / 8.24 ms per step, 27 / 32 7.48 / 7.50, 24 7.48
  for (int i = 0; i < n; ++i) {
    BwdBJob& b = jobs.j[i];
    const int nkb = (b.H4 + 15) >> 4, ncols = b.n_end - b.n_begin;
    int KG = std::max((nkb + kpg_target - 1) / kpg_target, std::min(8, nkb / 8));
    KG = std::max(1, KG);
    // a multiple of the job's XCD slots (slice lb % KG then stays on one XCD), while k_bwd_b_red holds <= 16 partials in registers
    const int KGm = (KG + nx[i] - 1) / nx[i] * nx[i];
    if (KGm <= 16 && KGm <= nkb) KG = KGm;
    KG = std::min(16, KG);
    b.kpg = (nkb + KG - 1) / KG;
    b.KG = (nkb + b.kpg - 1) / b.kpg;
    b.ncg = (ncols + 63) / 64; b.nrg = (b.N + 63) / 64;
    b.ldw = (ncols + 3) & ~3;
    b.ws = ws_base ? ws_base + off : nullptr;
    off += (size_t)b.KG * b.N * b.ldw;
    nb[i] = b.KG * b.ncg * b.nrg; w1[i] = 1; pp[i] = &b.plp;
    nbr[i] = (b.N * ncols + 255) / 256; nogroup[i] = false; pr[i] = &b.plr;
  }
  {
    bool over = false;
    for (int i = 0; i < n; ++i) over = over || (grouped[i] && (nb[i] + nx[i] - 1) / nx[i] > 32);
    if (over) {                    // re-plan without groups (KG depends on nx: recompute)
      for (int i = 0; i < n; ++i) { grouped[i] = false; nx[i] = 8; x0[i] = 0; }
      off = 0;
      for (int i = 0; i < n; ++i) {
        BwdBJob& b = jobs.j[i];
        const int nkb = (b.H4 + 15) >> 4;
        int KG = std::max((nkb + kpg_target - 1) / kpg_target, std::min(8, nkb / 8));
        KG = std::min(16, std::max(1, KG));
        b.kpg = (nkb + KG - 1) / KG;
        b.KG = (nkb + b.kpg - 1) / b.kpg;
        b.ws = ws_base ? ws_base + off : nullptr;
        off += (size_t)b.KG * b.N * b.ldw;
        nb[i] = b.KG * b.ncg * b.nrg;
      }
    }
  }
  const int gp = plan_rounds(n, nb, w1, nx, x0, grouped, pp);
  int nx8[MAXJ], x00[MAXJ];
  for (int i = 0; i < n; ++i) { nx8[i] = 8; x00[i] = 0; }
  const int gr = plan_rounds(n, nbr, w1, nx8, x00, nogroup, pr);       // the reduce: elementwise, contiguous (grouping it by layer: 8.67 vs 8.59 ms/step)
  fill_map(jobs.map, n, gp, [&](int q) -> const Place& { return jobs.j[q].plp; });
  fill_map(jobs.mapr, n, gr, [&](int q) -> const Place& { return jobs.j[q].plr; });
  return off;
}

// ---- the floor of a dependent launch, measured live for bench.py's latency bound: n launches of 256 workgroups in one stream,
// mode 0: empty kernels (the kernel boundary alone); mode 1: every workgroup reads what its predecessor launch wrote (one dependent
// operand round trip: 1 KB per wave), adds, stores -- the shape of the lightest recurrence step
__global__ __launch_bounds__(256) void k_floor(const float* __restrict__ in, float* __restrict__ out, int mode) {
  if (mode == 0) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float4 v = reinterpret_cast<const float4*>(in)[i];
  reinterpret_cast<float4*>(out)[i] = make_float4(v.x + 1.f, v.y, v.z, v.w);
}
void launch_floor_chain(float* a, float* b, int n, int mode, hipStream_t s) {
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_floor, dim3(256), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, mode);
}
long long g_chain_launches = 0;     // launches of the recurrence kernels issued by the host since the last reset (bench.py: latency bound)
void launch_bwd_b_splitk(const BwdBJobs& jobs, hipStream_t s) {
  g_chain_launches += 2;
  int bp = 8, br = 8, kpg_max = 1, nbp = 0;
  for (int i = 0; i < jobs.n; ++i) {
    const BwdBJob& b = jobs.j[i];
    bp = std::max(bp, 8 * b.plp.se); br = std::max(br, 8 * b.plr.se);
    kpg_max = std::max(kpg_max, b.kpg); nbp += b.plp.nb;
  }
  size_t lds = (size_t)64 * bp_sa4(kpg_max) * 16;
  lds = (lds + 8191) / 8192 * 8192;
  lds = std::max(lds, (size_t)8 * BP_RT * 16 * 17 * sizeof(float));
  if (nbp <= 256) lds = std::max(lds, (size_t)84 * 1024);       // one workgroup per CU: with <= 256 of them none should share a CU's matrix pipe
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_bp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_bwd_bp, dim3(bp), dim3(512), lds, s, jobs);
  hipLaunchKernelGGL(k_bwd_b_red, dim3(br), dim3(256), 0, s, jobs);
}

// the launchers place the jobs (Place) and size the grid; `kb_max` is the largest k-block count over the jobs, which picks the K split.
constexpr int g_gates_rows = 32;       // rows per k_fwd_gates workgroup (64-row / 16-wave blocks measured slower twice; removed)
int fwd_gates_rows() { return g_gates_rows; }
void launch_fwd_gates(const FwdGateJobs& jobs_in, int total_blocks, int kb_max, hipStream_t s) {
  ++g_chain_launches;
  FwdGateJobs jobs = jobs_in;
  // a launch that would fill less than half of the chip with 32-row blocks (the discriminator alone) runs 16-row blocks: twice
  // the workgroups, half the MFMA chain each (its time is the latency of one block, not bytes)
  int blocks32 = 0;
  for (int i = 0; i < jobs.n; ++i) blocks32 += jobs.j[i].nblk_c * ((jobs.j[i].N + 31) / 32);
  const bool small = g_gates_rows == 32 && blocks32 <= 128 && kb_max <= 36;
  if (small) {
    total_blocks = place_tiles(jobs, 16, [](const FwdGateJob& j) { return (double)((j.x ? j.ldx : 0) + j.ldm); });
    size_t lds = (size_t)16 * gates_sa4(kb_max * 16) * 16;
    lds = (lds + 8191) / 8192 * 8192;
    lds = std::max(lds, (size_t)8 * 16 * 17 * sizeof(float));
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_gates<18, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
    hipLaunchKernelGGL((k_fwd_gates<18, 1>), dim3(total_blocks), dim3(512), lds, s, jobs);
    return;
  }
  total_blocks = place_tiles(jobs, g_gates_rows, [](const FwdGateJob& j) { return (double)((j.x ? j.ldx : 0) + j.ldm); });
  // dynamic LDS: the widest job's A tile (rows x SA floats) rounded to the 8 KB DMA granule of the
  // 8 waves, and at least the reduction buffer
  const int ktot = kb_max * 16;
  const int rtg = g_gates_rows / 16;
  size_t lds = (size_t)16 * rtg * gates_sa4(ktot) * 16;
  const size_t granule = rtg == 2 ? 8192 : 16384;          // one DMA round of all waves of the workgroup
  lds = (lds + granule - 1) / granule * granule;
  if (lds < 8 * rtg * 16 * 17 * sizeof(float)) lds = 8 * rtg * 16 * 17 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_gates<18, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (kb_max > 36) {        // K = [x | m] wider than 576 floats (e.g. 512-cell layers without projection): 32 k-blocks per wave, 1 WG/CU
    static bool attr2 = false;
    if (!attr2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_gates<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr2 = true; }
    size_t l2 = (size_t)32 * gates_sa4(ktot) * 16;
    l2 = (l2 + 8191) / 8192 * 8192;
    hipLaunchKernelGGL((k_fwd_gates<32, 2>), dim3(total_blocks), dim3(512), l2, s, jobs);
  } else hipLaunchKernelGGL((k_fwd_gates<18, 2>), dim3(total_blocks), dim3(512), lds, s, jobs);
}
void launch_fwd_proj(const FwdProjJobs& jobs_in, int total_blocks, int kb_max, hipStream_t s) {
  ++g_chain_launches;
  FwdProjJobs jobs = jobs_in;
  total_blocks = place_tiles(jobs, 32, [](const FwdProjJob& j) { return (double)j.ldh; });
  bool drop = false;
  for (int i = 0; i < jobs.n; ++i) drop |= jobs.j[i].drop.ctr != nullptr;
  if (drop) {
    if (kb_max <= 24) hipLaunchKernelGGL((k_fwd_proj<4, true>), dim3(total_blocks), dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((k_fwd_proj<8, true>), dim3(total_blocks), dim3(512), 0, s, jobs);
  } else if (kb_max <= 24)
    hipLaunchKernelGGL(k_fwd_proj<4>, dim3(total_blocks), dim3(256), 0, s, jobs);
  else
    hipLaunchKernelGGL(k_fwd_proj<8>, dim3(total_blocks), dim3(512), 0, s, jobs);
}
int bwd_a_cells() { return 32; }      // k_bwd_a2: 32 x 32 tiles, the jobs' nblk_c counts 32-cell blocks
void launch_bwd_a(const BwdAJobs& jobs_in, int total_blocks, int kb_max, hipStream_t s) {
  ++g_chain_launches;
  BwdAJobs jobs = jobs_in;
  total_blocks = place_tiles(jobs, 32, [](const BwdAJob& j) { return (double)(j.Wp ? j.ldm : 16); });
  // dynamic LDS: 32 rows x (widest dm row + pad) of the jobs in the launch (without a projection: 32 x 36 floats)
  int sa4 = frag_stride4(8);
  for (int i = 0; i < jobs.n; ++i)
    if (jobs.j[i].Wp) sa4 = std::max(sa4, frag_stride4(((jobs.j[i].ldm + 15) >> 4) * 4));
  const size_t lds = (size_t)32 * sa4 * 16;
  bool drop = false;
  for (int i = 0; i < jobs.n; ++i) drop |= jobs.j[i].drop.ctr != nullptr;
  if (drop) {
    if (kb_max <= 4) hipLaunchKernelGGL((k_bwd_a2<4, true>), dim3(total_blocks), dim3(256), lds, s, jobs);
    else if (kb_max <= 18) hipLaunchKernelGGL((k_bwd_a2<18, true>), dim3(total_blocks), dim3(256), lds, s, jobs);
    else hipLaunchKernelGGL((k_bwd_a2<24, true>), dim3(total_blocks), dim3(256), lds, s, jobs);
  } else if (kb_max <= 4) hipLaunchKernelGGL(k_bwd_a2<4>, dim3(total_blocks), dim3(256), lds, s, jobs);
  else if (kb_max <= 18) hipLaunchKernelGGL(k_bwd_a2<18>, dim3(total_blocks), dim3(256), lds, s, jobs);
  else hipLaunchKernelGGL(k_bwd_a2<24>, dim3(total_blocks), dim3(256), lds, s, jobs);
}
void launch_bwd_b(const BwdBJobs& jobs_in, int total_blocks, int kb_max, hipStream_t s) {
  ++g_chain_launches;
  BwdBJobs jobs = jobs_in;
  total_blocks = place_tiles(jobs, kb_max <= 64 ? 16 : 32, [](const BwdBJob& j) { return (double)j.H4; });
  if (kb_max <= 64)  // small K (the discriminator alone): 8 waves x <= 8 k-blocks, one load round, no pipeline, 16-row tiles
    hipLaunchKernelGGL((k_bwd_b<8, 8, 8, 1>), dim3(total_blocks), dim3(512), 0, s, jobs);
  else               // 8 waves split K; each runs a double-buffered 6-k-block register pipeline
    hipLaunchKernelGGL((k_bwd_b<8, 0, 6>), dim3(total_blocks), dim3(512), 0, s, jobs);
}
// ---------------------------------------------------------------------------------------
// layout kernels: batch-major caller buffers <-> time-major padded internal buffers
// ---------------------------------------------------------------------------------------
__global__ void k_pack_tm(const float* __restrict__ src, float* __restrict__ dst, int B, int T, int D, int ld) {
  // one block per (t, b): coalesced read of the 257-float frame, coalesced write
  const int t = blockIdx.x, b = blockIdx.y;
  const float* s = src + ((size_t)b * T + t) * D;
  float* d = dst + ((size_t)t * B + b) * ld;
  for (int i = threadIdx.x; i < D; i += blockDim.x) d[i] = s[i];
}
__global__ void k_unpack_bm(const float* __restrict__ src, int ld, float* __restrict__ dst, int B, int T, int D) {
  const int t = blockIdx.x, b = blockIdx.y;
  const float* s = src + ((size_t)t * B + b) * ld;
  float* d = dst + ((size_t)b * T + t) * D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) d[i] = s[i];
}
// Everything a call stages from the caller's buffers in ONE launch (was two k_pack_tm + up to four 5 us device-to-device copies, all
// eager: caller pointers never enter a captured graph): z = 0, 1 the batch-major -> time-major packs, z = 2 the small copies.
__global__ void k_stage_inputs(const StageJobs j) {
  const int z = blockIdx.z;
  if (z < 2) {
    const StagePack& p = j.pack[z];
    if (!p.src) return;
    const int t = blockIdx.x, b = blockIdx.y;
    const float* s = p.src + ((size_t)b * j.T + t) * p.D;
    float* d = p.dst + ((size_t)t * j.B + b) * p.ld;
    for (int i = threadIdx.x; i < p.D; i += blockDim.x) d[i] = s[i];
    return;
  }
  if (blockIdx.x || blockIdx.y) return;                               // the copies (a few KB in all): one block
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (!j.copy[c].src) continue;
    const unsigned* s = (const unsigned*)j.copy[c].src;
    unsigned* d = (unsigned*)j.copy[c].dst;
    for (int i = threadIdx.x; i < j.copy[c].n; i += blockDim.x) d[i] = s[i];
  }
}
void launch_stage_inputs(const StageJobs& j, hipStream_t s) {
  // (Bt < B: the model pads its row count for the persistent recurrences; rows [Bt, B) of the packs stay zero, their lengths 0)
  hipLaunchKernelGGL(k_stage_inputs, dim3(j.T, j.Bt > 0 ? j.Bt : j.B, 3), dim3(128), 0, s, j);
}
void launch_pack_tm(const float* src, float* dst, int B, int T, int D, int ld, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_tm, dim3(T, B), dim3(D >= 128 ? 128 : 64), 0, s, src, dst, B, T, D, ld);
}
void launch_unpack_bm(const float* src, int ld, float* dst, int B, int T, int D, hipStream_t s, int Bt) {
  hipLaunchKernelGGL(k_unpack_bm, dim3(T, Bt > 0 ? Bt : B), dim3(D >= 128 ? 128 : 64), 0, s, src, ld, dst, B, T, D);
}

// discriminator input (gan_rnn_placeholder.py:205-213 + utils/ops.py:19-30): rows [0,B) real
// = labels + noise_r, rows [B,2B) (or [0,B) when !with_real) fake = G(x) + noise_f; the noise
// is [B,D], broadcast over time.
__global__ void k_build_d_input(const float* __restrict__ lab, const float* __restrict__ y,
                                const float* __restrict__ nr, const float* __restrict__ nf,
                                float* __restrict__ xd, int B, int T, int D, int ld, int with_real) {
  const int Nd = with_real ? 2 * B : B;
  const size_t total = (size_t)T * Nd * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const size_t r = i / D;
    const int bb = (int)(r % Nd), t = (int)(r / Nd);
    float v;
    if (with_real && bb < B) {
      v = lab[((size_t)t * B + bb) * ld + d] + (nr ? nr[bb * D + d] : 0.f);
    } else {
      const int b = with_real ? bb - B : bb;
      v = y[((size_t)t * B + b) * ld + d] + (nf ? nf[b * D + d] : 0.f);
    }
    xd[r * ld + d] = v;
  }
}
void launch_build_d_input(const float* lab, const float* y, const float* nr, const float* nf, float* xd,
                          int B, int T, int D, int ld, bool with_real, hipStream_t s) {
  const size_t total = (size_t)T * (with_real ? 2 * B : B) * D;
  const int blocks = (int)min((size_t)2048, (total + 255) / 256);
  hipLaunchKernelGGL(k_build_d_input, dim3(blocks), dim3(256), 0, s, lab, y, nr, nf, xd, B, T, D, ld, with_real ? 1 : 0);
}

__global__ void k_add_noise_rows(const float* __restrict__ src, const float* __restrict__ noise, float* __restrict__ dst,
                                 int B, int T, int D, int ld, int Ns, int row0) {
  const size_t total = (size_t)T * B * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const size_t r = i / D;
    const int b = (int)(r % B), t = (int)(r / B);
    dst[((size_t)t * Ns + row0 + b) * ld + d] = src[r * ld + d] + (noise ? noise[b * D + d] : 0.f);
  }
}
void launch_add_noise_rows(const float* src, const float* noise, float* dst, int B, int T, int D, int ld, int Ns, int row0, hipStream_t s) {
  const size_t total = (size_t)T * B * D;
  const int blocks = (int)min((size_t)2048, (total + 255) / 256);
  hipLaunchKernelGGL(k_add_noise_rows, dim3(blocks), dim3(256), 0, s, src, noise, dst, B, T, D, ld, Ns, row0);
}

__global__ void k_transpose(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) dst[(size_t)c * ldd + r] = tile[tx][i];
  }
}
void launch_transpose(const float* src, int lds_, float* dst, int ldd, int R, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, lds_, dst, ldd, R, C);
}
// all transposed weight copies of a network in ONE launch (after every optimizer step)
__global__ void k_transpose_many(TransposeList tl) {
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < tl.n && (int)blockIdx.x >= tl.j[j + 1].blk_base) ++j;
  const TransposeJob& J = tl.j[j];
  const int b = blockIdx.x - J.blk_base, nbc = (J.C + 31) / 32;
  const int c0 = (b % nbc) * 32, r0 = (b / nbc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < J.R && c < J.C) ? J.src[(size_t)r * J.lds + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < J.C && r < J.R) J.dst[(size_t)c * J.ldd + r] = tile[tx][i];
  }
}
void launch_transpose_many(TransposeList& tl, hipStream_t s) {
  if (tl.n == 0) return;
  int base = 0;
  for (int i = 0; i < tl.n; ++i) { tl.j[i].blk_base = base; base += ((tl.j[i].C + 31) / 32) * ((tl.j[i].R + 31) / 32); }
  hipLaunchKernelGGL(k_transpose_many, dim3(base), dim3(256), 0, s, tl);
}

// every fragment-tiled weight copy of a network in ONE launch (after every optimizer step): one thread per float4 of the
// tiled image (lane slot l of tile (ct, kb)), reads are 64-B runs per 16 lanes, writes are fully coalesced
__global__ __launch_bounds__(256) void k_swizzle_many(SwizzleList sl) {
  int j = 0;
  while (j + 1 < sl.n && (int)blockIdx.x >= sl.j[j + 1].blk_base) ++j;
  const SwizzleJob& J = sl.j[j];
  const size_t i4 = (size_t)(blockIdx.x - J.blk_base) * 256 + threadIdx.x;
  const size_t tile = i4 >> 6;
  if (tile >= (size_t)J.nct * J.nkb) return;
  const int l = (int)(i4 & 63), q = l >> 4, lr = l & 15;
  const int ct = (int)(tile / J.nkb), kb = (int)(tile - (size_t)ct * J.nkb);
  const int c = ct * 16 + lr, k0 = kb * 16 + 4 * q;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (J.gates == 0) {
    if (c < J.C) {
      const float* r = J.src + (size_t)(J.c0 + c) * J.ld;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + u < J.K1) v[u] = r[k0 + u];
    }
  } else {
    const int ncb16 = ((J.H + 15) >> 4) << 4;
    const int g = c / ncb16, cell = c - g * ncb16;
    if (g < J.gates && cell < J.H) {
      const float* col = J.src + (size_t)g * J.H + cell;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        if (k < J.kx) { if (k < J.I) v[u] = col[(size_t)k * J.ld]; }
        else if (k - J.kx < J.P) v[u] = col[(size_t)(J.I + k - J.kx) * J.ld];
      }
    }
  }
  *reinterpret_cast<float4*>(J.dst + i4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
void launch_swizzle_many(SwizzleList& sl, hipStream_t s) {
  if (sl.n == 0) return;
  int base = 0;
  for (int i = 0; i < sl.n; ++i) { sl.j[i].blk_base = base; base += (int)(((size_t)sl.j[i].nct * sl.j[i].nkb * 64 + 255) / 256); }
  hipLaunchKernelGGL(k_swizzle_many, dim3(base), dim3(256), 0, s, sl);
}

// ---------------------------------------------------------------------------------------
// R-CED (models/rced.py:90-102): tf.contrib.layers.conv2d([S, fw], SAME, stride 1) on NHWC [R, S, W, C] as a GEMM over
// the patch matrix.  Positions p = (r*S + h)*W + w; patch column k = (dh*fw + dw)*C + c = the row-major order of the
// filter tensor [S, fw, C, Cout].  SAME padding: (k-1)/2 leading zeros per axis (TF pads the extra element of an even
// extent at the end).  Element (r, h, w, c) of the source lives at src[r*row_stride + (h*W + w)*ldc + c] (layer 0 reads
// the fed [R][S*W] rows directly with ldc = 1).
// ---------------------------------------------------------------------------------------
template <int V>      // V = 4: C % 4 == 0 (aligned float4 along c), V = 1: any C
__global__ __launch_bounds__(256) void k_im2col(const float* __restrict__ src, size_t row_stride, int ldc, int C, int S, int W,
                                                int kh, int kw, float* __restrict__ col, int ldk, size_t M) {
  const int K = kh * kw * C, nv = ldk / V;
  const size_t total = M * (size_t)nv;
  const int pt = (kh - 1) / 2, pl = (kw - 1) / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / nv;
    const int k = (int)(i - p * nv) * V;
    const int w = (int)(p % W);
    const size_t rh = p / W;
    const int h = (int)(rh % S);
    const size_t r = rh / S;
    float v[V];
#pragma unroll
    for (int u = 0; u < V; ++u) v[u] = 0.f;
    if (k < K) {
      const int c = k % C, dd = k / C, dw = dd % kw, dh = dd / kw;
      const int hh = h + dh - pt, ww = w + dw - pl;
      if (hh >= 0 && hh < S && ww >= 0 && ww < W) {
        const float* q = src + r * row_stride + ((size_t)hh * W + ww) * ldc + c;
        if (V == 4) { const float4 t = *reinterpret_cast<const float4*>(q); v[0] = t.x; v[1 % V] = t.y; v[2 % V] = t.z; v[3 % V] = t.w; }
        else v[0] = *q;
      }
    }
    float* o = col + p * ldk + k;
    if (V == 4) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
    else *o = v[0];
  }
}
void launch_im2col(const float* src, size_t row_stride, int ldc, int C, int S, int W, int kh, int kw, float* col, int ldk, size_t M,
                   hipStream_t s) {
  if (C % 4 == 0 && ldc % 4 == 0) {
    const size_t total = M * (size_t)(ldk / 4);
    hipLaunchKernelGGL(k_im2col<4>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, src, row_stride, ldc, C,
                       S, W, kh, kw, col, ldk, M);
  } else {
    const size_t total = M * (size_t)ldk;
    hipLaunchKernelGGL(k_im2col<1>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, src, row_stride, ldc, C,
                       S, W, kh, kw, col, ldk, M);
  }
}
// adjoint of the patch matrix, as a gather (deterministic): dst[p][c] = sum over (dh, dw) of dcol[p'][(dh*kw+dw)*C + c] with
// p' = the output position whose patch element (dh, dw) is p.  dst is [M][ldc] (pad columns written as 0).
__global__ __launch_bounds__(256) void k_col2im(const float* __restrict__ dcol, int ldk, int C, int S, int W, int kh, int kw,
                                                float* __restrict__ dst, int ldc, size_t M) {
  const int nv = ldc / 4;
  const size_t total = M * (size_t)nv;
  const int pt = (kh - 1) / 2, pl = (kw - 1) / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / nv;
    const int c = (int)(i - p * nv) * 4;
    const int w = (int)(p % W);
    const size_t rh = p / W;
    const int h = (int)(rh % S);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      for (int dh = 0; dh < kh; ++dh) {
        const int ho = h - dh + pt;                  // output row whose patch row dh is h
        if (ho < 0 || ho >= S) continue;
        for (int dw = 0; dw < kw; ++dw) {
          const int wo = w - dw + pl;
          if (wo < 0 || wo >= W) continue;
          const size_t po = (rh - h + ho) * W + wo;
          const float4 t = *reinterpret_cast<const float4*>(dcol + po * ldk + (size_t)(dh * kw + dw) * C + c);
          acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
      }
    }
    *reinterpret_cast<float4*>(dst + p * ldc + c) = acc;
  }
}
// [R][ld_src] rows of n scalars -> [R*n][4] positions x (value, 0, 0, 0): the single-channel input of R-CED's first conv2d as NHWC with C' = 4
__global__ void k_expand_c4(const float* __restrict__ src, int ld_src, int n, float4* __restrict__ dst, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / n;
    dst[i] = make_float4(src[r * ld_src + (i - r * n)], 0.f, 0.f, 0.f);
  }
}
void launch_expand_c4(const float* src, int ld_src, int n, float* dst, size_t rows, hipStream_t s) {
  const size_t total = rows * (size_t)n;
  hipLaunchKernelGGL(k_expand_c4, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, src, ld_src, n,
                     reinterpret_cast<float4*>(dst), total);
}
void launch_col2im(const float* dcol, int ldk, int C, int S, int W, int kh, int kw, float* dst, int ldc, size_t M, hipStream_t s) {
  const size_t total = M * (size_t)(ldc / 4);
  hipLaunchKernelGGL(k_col2im, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, dcol, ldk, C, S, W, kh, kw,
                     dst, ldc, M);
}

__global__ void k_zero_many(ZeroList zl) {
  const int j = blockIdx.y;
  if (j >= zl.n) return;
  float* p = zl.p[j];
  const size_t n = zl.len[j];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
void launch_zero_many(const ZeroList& zl, hipStream_t s) {
  if (zl.n == 0) return;
  hipLaunchKernelGGL(k_zero_many, dim3(32, zl.n), dim3(256), 0, s, zl);
}

__global__ void k_fill(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill(float* p, size_t n, float v, hipStream_t s) {
  if (n == 0) return;
  const int blocks = (int)min((size_t)2048, (n + 255) / 256);
  hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, s, p, n, v);
}

__global__ void k_lrelu_bwd(const float* __restrict__ hval, float* __restrict__ d, size_t rows, int cols, int ld, float alpha) {
  const size_t total = rows * (size_t)cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols; const int c = (int)(i % cols);
    const size_t o = r * ld + c;
    if (!(hval[o] > 0.f)) d[o] *= alpha;
  }
}
void launch_lrelu_bwd(const float* hval, float* d, size_t rows, int cols, int ld, float alpha, hipStream_t s) {
  const size_t total = rows * cols;
  const int blocks = (int)min((size_t)2048, (total + 255) / 256);
  hipLaunchKernelGGL(k_lrelu_bwd, dim3(blocks), dim3(256), 0, s, hval, d, rows, cols, ld, alpha);
}

__global__ void k_drop_tick(unsigned long long* ctr) { if (threadIdx.x == 0) *ctr += 1ull; }
void launch_drop_tick(unsigned long long* ctr, hipStream_t s) { hipLaunchKernelGGL(k_drop_tick, dim3(1), dim3(64), 0, s, ctr); }
__global__ __launch_bounds__(256) void k_dropout_fwd(float* __restrict__ y, size_t rows, int cols, int ld, unsigned long long key, unsigned thr,
                                                     float keep) {
  const size_t n = rows * (size_t)cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cols;
    float* p = y + r * ld + (i - r * cols);
    const bool on = (unsigned)(splitmix64(key + i) >> 40) < thr;
    *p = on ? *p / keep : 0.f;
  }
}
void launch_dropout_fwd(float* y, size_t rows, int cols, int ld, unsigned long long key, unsigned thr, float keep, hipStream_t s) {
  const size_t n = rows * (size_t)cols;
  if (n == 0) return;
  hipLaunchKernelGGL(k_dropout_fwd, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, y, rows, cols, ld, key, thr, keep);
}
__global__ __launch_bounds__(256) void k_dropout_bwd(const float* __restrict__ y, float* __restrict__ d, size_t rows, int cols, int ld, float keep) {
  const size_t n = rows * (size_t)cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cols, o = r * ld + (i - r * cols);
    d[o] = y[o] > 0.f ? d[o] / keep : 0.f;
  }
}
void launch_dropout_bwd(const float* y, float* d, size_t rows, int cols, int ld, float keep, hipStream_t s) {
  const size_t n = rows * (size_t)cols;
  if (n == 0) return;
  hipLaunchKernelGGL(k_dropout_bwd, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, y, d, rows, cols, ld, keep);
}

// column sums in two deterministic stages (bias and peephole gradients)
constexpr int CS_SLICES = 64;
__global__ __launch_bounds__(256) void k_colsum1(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                 float* __restrict__ scratch, int rows, int cols) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int per = (rows + CS_SLICES - 1) / CS_SLICES;
  const int rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float s = 0.f;
  if (c < cols) {
    for (int r = rbeg + rl; r < rend; r += 4) {
      const float v = a[(size_t)r * lda + c];
      s += b ? v * b[(size_t)r * ldb + c] : v;
    }
  }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols)
    scratch[(size_t)blockIdx.y * cols + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ void k_colsum2(const float* __restrict__ scratch, float* __restrict__ out, int cols) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int i = 0; i < CS_SLICES; ++i) s += scratch[(size_t)i * cols + c];
  out[c] = s;
}
// Tall, narrow matrices (R-CED bias gradients: 10^5..10^6 rows x <= 32 columns): more row slices, and a second stage that
// splits the slices over four thread groups (fixed summation order: slice-group partials, then groups 0..3).
__global__ __launch_bounds__(256) void k_colsum1_n(const float* __restrict__ a, int lda, float* __restrict__ scratch, int rows, int cols,
                                                   int per) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float s = 0.f;
  if (c < cols) {
#pragma unroll 4
    for (int r = rbeg + rl; r < rend; r += 4) s += a[(size_t)r * lda + c];
  }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols)
    scratch[(size_t)blockIdx.y * cols + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void k_colsum2_n(const float* __restrict__ scratch, float* __restrict__ out, int cols, int slices) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols) {
#pragma unroll 4
    for (int i = g; i < slices; i += 4) s += scratch[(size_t)i * cols + c];
  }
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < cols) out[c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
void launch_colsum_tall(const float* a, int lda, float* out, int rows, int cols, float* scratch, size_t scratch_floats, hipStream_t s) {
  int slices = (int)std::min<size_t>(512, scratch_floats / (size_t)std::max(cols, 1));
  slices = std::max(1, std::min(slices, (rows + 63) / 64));
  const int per = (rows + slices - 1) / slices;
  slices = (rows + per - 1) / per;
  hipLaunchKernelGGL(k_colsum1_n, dim3((cols + 63) / 64, slices), dim3(256), 0, s, a, lda, scratch, rows, cols, per);
  hipLaunchKernelGGL(k_colsum2_n, dim3((cols + 63) / 64), dim3(256), 0, s, scratch, out, cols, slices);
}

// One pass over dZ [rows][4H] and the cell stash: db[4H] = colsum(dZ); dw_i = colsum(dai*c_prev); dw_f = colsum(daf*c_prev);
// dw_o = colsum(dao*c_cur).  scratch: CS_SLICES x 7H floats.
__global__ __launch_bounds__(256) void k_lstm_colsums1(const float* __restrict__ dz, const float* __restrict__ cprev,
                                                       const float* __restrict__ ccur, float* __restrict__ scratch, int rows, int H) {
  __shared__ float red[7][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int per = (rows + CS_SLICES - 1) / CS_SLICES;
  const int rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < H) {
#pragma unroll 4
    for (int r = rbeg + rl; r < rend; r += 4) {
      const float* z = dz + (size_t)r * 4 * H + c;
      const float di = z[0], dj = z[H], df = z[2 * H], d_o = z[3 * H];
      const float cp = cprev[(size_t)r * H + c], cc = ccur[(size_t)r * H + c];
      s[0] += di; s[1] += dj; s[2] += df; s[3] += d_o; s[4] += di * cp; s[5] += df * cp; s[6] += d_o * cc;
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) red[k][rl][threadIdx.x & 63] = s[k];
  __syncthreads();
  if (rl == 0 && c < H) {
    const int x = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 7; ++k)
      scratch[((size_t)blockIdx.y * 7 + k) * H + c] = ((red[k][0][x] + red[k][1][x]) + red[k][2][x]) + red[k][3][x];
  }
}
__global__ void k_lstm_colsums2(const float* __restrict__ scratch, float* __restrict__ db, float* __restrict__ dwi,
                                float* __restrict__ dwf, float* __restrict__ dwo, int H) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int i = 0; i < CS_SLICES; ++i)
#pragma unroll
    for (int k = 0; k < 7; ++k) s[k] += scratch[((size_t)i * 7 + k) * H + c];
  db[c] = s[0]; db[H + c] = s[1]; db[2 * H + c] = s[2]; db[3 * H + c] = s[3];
  dwi[c] = s[4]; dwf[c] = s[5]; dwo[c] = s[6];
}
// the same for up to 4 layers of one shape in one launch each (blockIdx.z = layer; scratch: n x CS_SLICES x 7H floats)
__global__ __launch_bounds__(256) void k_lstm_colsums1_b(const ColsumsBatch bt, float* __restrict__ scratch, int rows, int H) {
  __shared__ float red[7][4][64];
  const int p = blockIdx.z;
  const float* __restrict__ dz = bt.dz[p]; const float* __restrict__ cprev = bt.cprev[p]; const float* __restrict__ ccur = bt.ccur[p];
  scratch += (size_t)p * CS_SLICES * 7 * H;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int per = (rows + CS_SLICES - 1) / CS_SLICES;
  const int rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < H) {
#pragma unroll 4
    for (int r = rbeg + rl; r < rend; r += 4) {
      const float* z = dz + (size_t)r * 4 * H + c;
      const float di = z[0], dj = z[H], df = z[2 * H], d_o = z[3 * H];
      const float cp = cprev[(size_t)r * H + c], cc = ccur[(size_t)r * H + c];
      s[0] += di; s[1] += dj; s[2] += df; s[3] += d_o; s[4] += di * cp; s[5] += df * cp; s[6] += d_o * cc;
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) red[k][rl][threadIdx.x & 63] = s[k];
  __syncthreads();
  if (rl == 0 && c < H) {
    const int x = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 7; ++k)
      scratch[((size_t)blockIdx.y * 7 + k) * H + c] = ((red[k][0][x] + red[k][1][x]) + red[k][2][x]) + red[k][3][x];
  }
}
__global__ void k_lstm_colsums2_b(const float* __restrict__ scratch, const ColsumsBatch bt, int H) {
  const int p = blockIdx.y;
  scratch += (size_t)p * CS_SLICES * 7 * H;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int i = 0; i < CS_SLICES; ++i)
#pragma unroll
    for (int k = 0; k < 7; ++k) s[k] += scratch[((size_t)i * 7 + k) * H + c];
  float* db = bt.db[p];
  db[c] = s[0]; db[H + c] = s[1]; db[2 * H + c] = s[2]; db[3 * H + c] = s[3];
  bt.dwi[p][c] = s[4]; bt.dwf[p][c] = s[5]; bt.dwo[p][c] = s[6];
}
void launch_lstm_colsums_batch(const ColsumsBatch& bt, int rows, int H, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(k_lstm_colsums1_b, dim3((H + 63) / 64, CS_SLICES, bt.n), dim3(256), 0, s, bt, scratch, rows, H);
  hipLaunchKernelGGL(k_lstm_colsums2_b, dim3((H + 255) / 256, bt.n), dim3(256), 0, s, scratch, bt, H);
}
void launch_lstm_colsums(const float* dz, const float* cprev, const float* ccur, float* db, float* dwi, float* dwf, float* dwo,
                         int rows, int H, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(k_lstm_colsums1, dim3((H + 63) / 64, CS_SLICES), dim3(256), 0, s, dz, cprev, ccur, scratch, rows, H);
  hipLaunchKernelGGL(k_lstm_colsums2, dim3((H + 255) / 256), dim3(256), 0, s, scratch, db, dwi, dwf, dwo, H);
}

void launch_colsum(const float* a, int lda, const float* b, int ldb, float* out, int rows, int cols, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(k_colsum1, dim3((cols + 63) / 64, CS_SLICES), dim3(256), 0, s, a, lda, b, ldb, scratch, rows, cols);
  hipLaunchKernelGGL(k_colsum2, dim3((cols + 255) / 256), dim3(256), 0, s, scratch, out, cols);
}

// ---------------------------------------------------------------------------------------
// losses (models/gan_rnn_placeholder.py:244-260)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  // deterministic: fixed shuffle tree per wave, then fixed-order sum of the 16 wave totals
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) s += red[i];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(1024) void k_lsgan(const float* __restrict__ logits, int ldl, float* __restrict__ dlogits,
                                                int T, int Nd, int n_real, const float* __restrict__ t_real,
                                                const float* __restrict__ t_fake, float* __restrict__ loss3,
                                                int clip_on, float clip_lo, float clip_hi, int Bp, int Bt) {
  __shared__ float red[16];
  const int rows = T * Nd;
  const float tr = *t_real, tf = *t_fake;
  // (Bp > 0: each half of Bp rows per frame holds Bt utterances and Bp - Bt padding rows, which are no part of any mean)
  const int nr_ = Bp > 0 ? n_real / Bp * Bt : n_real, nf_ = Bp > 0 ? (Nd - n_real) / Bp * Bt : Nd - n_real;
  const float cr = (float)T * (float)nr_, cf = (float)T * (float)nf_;
  float sr = 0.f, sf = 0.f;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    const bool real = (r % Nd) < n_real;
    const bool valid = Bp <= 0 || ((r % Nd) % Bp) < Bt;
    const float raw = logits[(size_t)r * ldl];
    // discriminator_dnn.py:93 tf.clip_by_value(y, -0.5, 1.5): value clipped, gradient passes where lo <= y <= hi
    const float val = clip_on ? fminf(fmaxf(raw, clip_lo), clip_hi) : raw;
    const float d = valid ? val - (real ? tr : tf) : 0.f;
    if (real) sr += d * d; else sf += d * d;
    if (dlogits) dlogits[(size_t)r * ldl] = (clip_on && (raw < clip_lo || raw > clip_hi)) ? 0.f : 2.f * d / (real ? cr : cf);
  }
  sr = block_sum_1024(sr, red);
  sf = block_sum_1024(sf, red);
  if (threadIdx.x == 0) {
    const float lr_ = n_real > 0 ? sr / cr : 0.f;
    const float lf_ = (Nd - n_real) > 0 ? sf / cf : 0.f;
    loss3[0] = lr_; loss3[1] = lf_; loss3[2] = lr_ + lf_;
  }
}
void launch_lsgan(const float* logits, int ldl, float* dlogits, int T, int Nd, int n_real,
                  const float* t_real, const float* t_fake, float* loss3, hipStream_t s, bool clip_on, float clip_lo, float clip_hi,
                  int Bp, int Bt) {
  hipLaunchKernelGGL(k_lsgan, dim3(1), dim3(1024), 0, s, logits, ldl, dlogits, T, Nd, n_real, t_real, t_fake, loss3,
                     clip_on ? 1 : 0, clip_lo, clip_hi, Bp, Bt);
}

// The head of discriminator_lstm in ONE pass over its top layer's outputs (models/discriminator_lstm.py:93-104: fully_connected to one
// logit; gan_rnn_placeholder.py:244-250: the LSGAN terms over ALL frames, padded ones included) and, when gradients are wanted, its
// backward half as well: logits, the three losses, dlogits, d(outputs) = dlogits . W^T and the FC's own gradients (dW = out^T dlogits,
// db = sum dlogits).  Was gemm_n32 + k_lsgan (one block) + gemm16 + split-K reduce + two column-sum kernels + gemm16: seven launches,
// 85 us between the discriminator's forward and backward recurrences with the chip idle.  One row per thread, one wave per block; the
// waves' partials are summed in a fixed order by k_dhead2 (deterministic).
__global__ __launch_bounds__(64) void k_dhead1(const DHeadArgs a) {
  __shared__ float wsh[DH_MAXR + 1];
  const int lane = threadIdx.x;
  const int rows = a.T * a.Nd, dR = a.dR;
  if (lane < dR) wsh[lane] = a.w[(size_t)lane * a.ldw];
  if (lane == 0) wsh[DH_MAXR] = a.b[0];
  __syncthreads();
  const int r = blockIdx.x * 64 + lane;
  const bool on = r < rows;
  const float tr = *a.t_real, tf = *a.t_fake;
  const int nr_ = a.Bp > 0 ? a.n_real / a.Bp * a.Bt : a.n_real, nf_ = a.Bp > 0 ? (a.Nd - a.n_real) / a.Bp * a.Bt : a.Nd - a.n_real;
  const float cr = (float)a.T * (float)nr_, cf = (float)a.T * (float)nf_;
  float x[DH_MAXR];
  float logit = wsh[DH_MAXR];
  const float* xr = a.top + (size_t)(on ? r : 0) * a.ldt;
#pragma unroll
  for (int c = 0; c < DH_MAXR; c += 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < dR) v = *reinterpret_cast<const float4*>(xr + c);          // (dR % 4 == 0: uniform)
    x[c] = v.x; x[c + 1] = v.y; x[c + 2] = v.z; x[c + 3] = v.w;
  }
#pragma unroll
  for (int c = 0; c < DH_MAXR; ++c) if (c < dR) logit += x[c] * wsh[c];
  const bool real = on && (r % a.Nd) < a.n_real;
  const bool valid = a.Bp <= 0 || ((r % a.Nd) % a.Bp) < a.Bt;          // (padding rows of a row-padded model: no part of any mean)
  const float d = valid ? logit - (real ? tr : tf) : 0.f;
  float sr = (on && real) ? d * d : 0.f, sf = (on && !real) ? d * d : 0.f;
  const float dl = on ? 2.f * d / (real ? cr : cf) : 0.f;
  if (on) {
    a.logits[(size_t)r * a.ldl] = logit;
    if (a.want_grads) {
      a.dlogits[(size_t)r * a.ldl] = dl;
      float* o = a.dout + (size_t)r * a.ldo;
#pragma unroll
      for (int c = 0; c < DH_MAXR; c += 4)
        if (c < dR) *reinterpret_cast<float4*>(o + c) = make_float4(dl * wsh[c], dl * wsh[c + 1], dl * wsh[c + 2], dl * wsh[c + 3]);
    }
  }
  // the wave's partials: [0] sum over real rows, [1] over fake rows, [2] sum dlogits, [3 + c] sum out[:, c] * dlogits
  auto wsum = [&](float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; };
  float* part = a.part + (size_t)blockIdx.x * (DH_MAXR + 3);
  sr = wsum(sr); sf = wsum(sf);
  if (lane == 0) { part[0] = sr; part[1] = sf; }
  if (a.want_wgrads) {
    const float sd = wsum(dl);
    if (lane == 0) part[2] = sd;
#pragma unroll
    for (int c = 0; c < DH_MAXR; ++c)
      if (c < dR) { const float v = wsum(x[c] * dl); if (lane == 0) part[3 + c] = v; }
  }
}
// 16 lanes per quantity: lane j sums the partials of blocks j, j + 16, ... in order, then a fixed shuffle tree over the 16
__global__ __launch_bounds__(1024) void k_dhead2(const DHeadArgs a, int nblocks) {
  __shared__ float tot[DH_MAXR + 3];
  const int q = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int nq = a.want_wgrads ? 3 + a.dR : 2;
  float s = 0.f;
  if (q < nq)
    for (int b = j; b < nblocks; b += 16) s += a.part[(size_t)b * (DH_MAXR + 3) + q];
  s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
  if (q < nq && j == 0) {
    tot[q] = s;
    if (a.want_wgrads) {
      if (q == 2) a.gb[0] = s;
      if (q >= 3) a.gw[(size_t)(q - 3) * a.ldw] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nr_ = a.Bp > 0 ? a.n_real / a.Bp * a.Bt : a.n_real, nf_ = a.Bp > 0 ? (a.Nd - a.n_real) / a.Bp * a.Bt : a.Nd - a.n_real;
    const float cr = (float)a.T * (float)nr_, cf = (float)a.T * (float)nf_;
    const float lr_ = a.n_real > 0 ? tot[0] / cr : 0.f;
    const float lf_ = (a.Nd - a.n_real) > 0 ? tot[1] / cf : 0.f;
    a.loss3[0] = lr_; a.loss3[1] = lf_; a.loss3[2] = lr_ + lf_;
  }
}
void launch_dhead(const DHeadArgs& a, hipStream_t s) {
  const int rows = a.T * a.Nd, nb = (rows + 63) / 64;
  hipLaunchKernelGGL(k_dhead1, dim3(nb), dim3(64), 0, s, a);
  hipLaunchKernelGGL(k_dhead2, dim3(1), dim3(1024), 0, s, a, nb);
}

// models/gan.py:158-175: joint[r] = concat(x[r][off : off+dim], tail[r][0 : Dt]); rows [row0, row0+R) of `joint`
__global__ void k_build_joint(const float* __restrict__ x, int ldx, int off, int dim, const float* __restrict__ tail, int ldt, int Dt,
                              float* __restrict__ joint, int ldj, int row0, int R) {
  const int W = dim + Dt;
  const size_t total = (size_t)R * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / W), c = (int)(i % W);
    joint[(size_t)(row0 + r) * ldj + c] = c < dim ? x[(size_t)r * ldx + off + c] : tail[(size_t)r * ldt + (c - dim)];
  }
}
void launch_build_joint(const float* x, int ldx, int off, int dim, const float* tail, int ldt, int Dt, float* joint, int ldj,
                        int row0, int R, hipStream_t s) {
  const size_t total = (size_t)R * (dim + Dt);
  const int blocks = (int)min((size_t)2048, (total + 255) / 256);
  hipLaunchKernelGGL(k_build_joint, dim3(blocks), dim3(256), 0, s, x, ldx, off, dim, tail, ldt, Dt, joint, ldj, row0, R);
}
__global__ void k_slice_cols(const float* __restrict__ src, int lds_, int off, float* __restrict__ dst, int ldd, int R, int C) {
  const size_t total = (size_t)R * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i % C);
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds_ + off + c];
  }
}
void launch_slice_cols(const float* src, int lds_, int off, float* dst, int ldd, int R, int C, hipStream_t s) {
  const size_t total = (size_t)R * C;
  const int blocks = (int)min((size_t)2048, (total + 255) / 256);
  hipLaunchKernelGGL(k_slice_cols, dim3(blocks), dim3(256), 0, s, src, lds_, off, dst, ldd, R, C);
}

constexpr int MSE_BLOCKS = 256;
__global__ __launch_bounds__(256) void k_mse1(const float* __restrict__ y, const float* __restrict__ lab, int ld,
                                              float* __restrict__ dy, int rows, int D, const float* __restrict__ lambda,
                                              int accumulate, float* __restrict__ scratch, int Bp, int Bt) {
  __shared__ float red[16];
  const size_t total = (size_t)rows * D;
  const int rows_eff = Bp > 0 ? rows / Bp * Bt : rows;                  // (Bp > 0: rows [Bt, Bp) of every frame are padding)
  const float scale = dy ? (*lambda) / (float)rows_eff : 0.f;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / D; const int c = (int)(i % D);
    const size_t o = r * ld + c;
    const bool valid = Bp <= 0 || (int)(r % Bp) < Bt;
    const float d = valid ? y[o] - lab[o] : 0.f;
    s += d * d;
    if (dy) dy[o] = (accumulate ? dy[o] : 0.f) + scale * d;
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}
__global__ void k_mse2(const float* __restrict__ scratch, int rows, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < MSE_BLOCKS; ++i) s += scratch[i];
    *out = 0.5f * s / (float)rows;       // 0.5 * D * sum / (rows * D)
  }
}
void launch_mse(const float* y, const float* lab, int ld, float* dy, int rows, int D, const float* lambda,
                bool accumulate, float* loss_out, float* scratch, hipStream_t s, int Bp, int Bt) {
  hipLaunchKernelGGL(k_mse1, dim3(MSE_BLOCKS), dim3(256), 0, s, y, lab, ld, dy, rows, D, lambda, accumulate ? 1 : 0, scratch, Bp, Bt);
  hipLaunchKernelGGL(k_mse2, dim3(1), dim3(64), 0, s, scratch, Bp > 0 ? rows / Bp * Bt : rows, loss_out);
}

__global__ void k_g_total(float* l4, const float* lambda) {
  if (threadIdx.x == 0) l4[3] = l4[0] + (*lambda) * l4[1] + l4[2];
}
void launch_g_total(float* l4, const float* lambda, hipStream_t s) {
  hipLaunchKernelGGL(k_g_total, dim3(1), dim3(64), 0, s, l4, lambda);
}
__global__ void k_copy_f(const float* src, float* dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
void launch_copy_f(const float* src, float* dst, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_f, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
}

// ---------------------------------------------------------------------------------------
// optimizer: per-tensor clip_by_norm (gan_rnn_placeholder.py:178-182), SGD (:144,183),
// TF-form Adam (:147,184), EMA (:149-150,185-186).  One block per 4096-float chunk; a chunk
// never straddles two tensors.
// ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, ChunkTable ct, float* __restrict__ partial) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const float* p = g + ct.off[c];
  const int n = ct.len[c];
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < n; i += 1024) {       // len and off are multiples of 4
    const float4 v = *reinterpret_cast<const float4*>(p + i);
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) partial[c] = s;
}
void launch_sumsq(const float* g, const ChunkTable& ct, float* partial, hipStream_t s) {
  hipLaunchKernelGGL(k_sumsq, dim3(ct.n_chunks), dim3(256), 0, s, g, ct, partial);
}

// The discriminator's weight gradients come out of k_dlstm_bwd as one record of partial sums per (layer, 16-row tile) (dpersist.hip
// dp_dw_body).  One block per 4096-float chunk of the gradient buffer: the chunk's floats = the sum of the tiles' records in tile order
// (deterministic), written to the buffer, and -- k_sumsq's job -- the chunk's sum of squares for the per-tensor clip.
__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ ws, size_t stride, int ntiles, const long long* __restrict__ src,
                                                   float* __restrict__ g, ChunkTable ct, float* __restrict__ partial) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int t = ct.tensor[c];
  const int off = ct.off[c], n = ct.len[c];
  const long long so = src[t];
  float s = 0.f;
  if (so < 0) {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(g + off + i);
      s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    const float* p = ws + so + (off - ct.off[ct.t_first[t]]);
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      float4 a = *reinterpret_cast<const float4*>(p + i);
      for (int r = 1; r < ntiles; ++r) {
        const float4 b = *reinterpret_cast<const float4*>(p + (size_t)r * stride + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      *reinterpret_cast<float4*>(g + off + i) = a;
      s += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) partial[c] = s;
}
void launch_dw_reduce(const float* ws, size_t stride, int ntiles, const long long* src, float* g, const ChunkTable& ct, float* partial, hipStream_t s) {
  hipLaunchKernelGGL(k_dw_reduce, dim3(ct.n_chunks), dim3(256), 0, s, ws, stride, ntiles, src, g, ct, partial);
}

__global__ __launch_bounds__(256) void k_l2(const float* __restrict__ w, float* __restrict__ g, ChunkTable ct,
                                            const float* __restrict__ l2_scale, float* __restrict__ partial) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const bool on = ct.t_l2[ct.tensor[c]] != 0;
  const int off = ct.off[c], n = ct.len[c];
  const float sc = *l2_scale;
  float s = 0.f;
  if (on) {
    for (int i = threadIdx.x; i < n; i += 256) {
      const float v = w[off + i];
      s += v * v;
      g[off + i] += sc * v;
    }
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) partial[c] = s;
}
void launch_l2(const float* w, float* g, const ChunkTable& ct, const float* l2_scale, float* partial, hipStream_t s) {
  hipLaunchKernelGGL(k_l2, dim3(ct.n_chunks), dim3(256), 0, s, w, g, ct, l2_scale, partial);
}
__global__ __launch_bounds__(1024) void k_l2_total(const float* __restrict__ partial, int n, const float* __restrict__ l2_scale, float* out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) *out = 0.5f * (*l2_scale) * s;
}
void launch_l2_total(const float* partial, int n, const float* l2_scale, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_l2_total, dim3(1), dim3(1024), 0, s, partial, n, l2_scale, out);
}

__device__ __forceinline__ float tensor_clip_scale(const ChunkTable& ct, const float* partial, int c, float clip, float* red) {
  const int t = ct.tensor[c];
  const int first = ct.t_first[t], cnt = ct.t_count[t];
  float s = 0.f;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) s += partial[first + i];
  s = block_sum_1024(s, red);
  if (!(clip > 0.f)) return 1.0f;       // models/gan.py:140-143 applies the averaged gradients unclipped
  // tf.clip_by_norm: t * clip * min(rsqrt(sum(t*t)), 1/clip)
  const float inv = s > 0.f ? 1.0f / sqrtf(s) : __builtin_inff();
  return clip * fminf(inv, 1.0f / clip);
}

__global__ __launch_bounds__(256) void k_apply_sgd(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ ema,
                                                   ChunkTable ct, const float* __restrict__ partial, const float* __restrict__ dyn) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const float scale = tensor_clip_scale(ct, partial, c, dyn[DYN_CLIP], red);
  const float lr = dyn[DYN_D_LR], dec = dyn[DYN_EMA];
  const int off = ct.off[c], n = ct.len[c];
  for (int i = threadIdx.x; i < n; i += 256) {
    const float nw = w[off + i] - lr * (g[off + i] * scale);
    w[off + i] = nw;
    if (ema) ema[off + i] = dec * ema[off + i] + (1.f - dec) * nw;
  }
}
void launch_apply_sgd(float* w, const float* g, float* ema, const ChunkTable& ct, const float* partial, const float* dyn, hipStream_t s) {
  hipLaunchKernelGGL(k_apply_sgd, dim3(ct.n_chunks), dim3(256), 0, s, w, g, ema, ct, partial, dyn);
}

__global__ __launch_bounds__(256) void k_apply_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ ema, ChunkTable ct,
                                                    const float* __restrict__ partial, const float* __restrict__ dyn, int lrt_slot) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const float scale = tensor_clip_scale(ct, partial, c, dyn[DYN_CLIP], red);
  const float b1 = dyn[DYN_B1], b2 = dyn[DYN_B2], eps = dyn[DYN_EPS], lrt = dyn[lrt_slot], dec = dyn[DYN_EMA];
  const int off = ct.off[c], n = ct.len[c];
  for (int i = threadIdx.x; i < n; i += 256) {
    const float gg = g[off + i] * scale;
    const float mm = b1 * m[off + i] + (1.f - b1) * gg;
    const float vv = b2 * v[off + i] + (1.f - b2) * gg * gg;
    m[off + i] = mm; v[off + i] = vv;
    const float nw = w[off + i] - lrt * mm / (sqrtf(vv) + eps);      // eps outside the sqrt (TF form)
    w[off + i] = nw;
    if (ema) ema[off + i] = dec * ema[off + i] + (1.f - dec) * nw;
  }
}
void launch_apply_adam(float* w, const float* g, float* m, float* v, float* ema, const ChunkTable& ct,
                       const float* partial, const float* dyn, hipStream_t s, int lrt_slot) {
  hipLaunchKernelGGL(k_apply_adam, dim3(ct.n_chunks), dim3(256), 0, s, w, g, m, v, ema, ct, partial, dyn, lrt_slot);
}

// Adam's beta powers (tf.train.AdamOptimizer): t += 1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t), on the
// device so that a captured graph advances it on every replay.
__global__ void k_adam_tick(float* dyn, int* t, double b1, double b2, int lr_slot, int lrt_slot) {
  if (threadIdx.x == 0) {
    const int tt = *t + 1;
    *t = tt;
    dyn[lrt_slot] = (float)((double)dyn[lr_slot] * sqrt(1.0 - pow(b2, (double)tt)) / (1.0 - pow(b1, (double)tt)));
  }
}
void launch_adam_tick(float* dyn, int* t, double b1, double b2, hipStream_t s, int lr_slot, int lrt_slot) {
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, s, dyn, t, b1, b2, lr_slot, lrt_slot);
}

__global__ void k_pad_copy(const float* __restrict__ dense_c, float* __restrict__ dense_m, float* __restrict__ padded,
                           int rows, int cols, int ld, int to_padded) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols; const int c = (int)(i % cols);
    if (to_padded) padded[r * ld + c] = dense_c[i];
    else dense_m[i] = padded[r * ld + c];
  }
}
void launch_pad_copy(const float* dense, float* padded, int rows, int cols, int ld, bool to_padded, hipStream_t s) {
  const size_t total = (size_t)rows * cols;
  const int blocks = (int)min((size_t)1024, (total + 255) / 256);
  hipLaunchKernelGGL(k_pad_copy, dim3(blocks), dim3(256), 0, s, dense, const_cast<float*>(dense), padded, rows, cols, ld, to_padded ? 1 : 0);
}

}  // namespace rsr
