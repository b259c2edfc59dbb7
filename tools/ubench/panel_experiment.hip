// panel_experiment.hip -- ROUND 2 EXPERIMENT, NOT PART OF THE PRODUCT LIBRARY (results: profiles/r2_ubench_gates_ablation_*.txt,
// DESIGN.md 6).  Candidate heavy recurrent step kernels: a pipelined 64-row "panel" MFMA product with two epilogues.
//
// The two dominant phases of an LSTMP time step (models/lstm.py:89-112; cell math models/BNLSTMCell.py:176-217) are products
//     C[64 rows x 48 columns] = A[64 x K] . W^T,   A rows and W rows both k-contiguous in HBM,
// with M = the batch rows of one tower (64 at BASELINE.json's size), followed by pointwise work:
//   gates : A = [x_t | m_{t-1}],  W = rows of KxT|KhT of 12 cells x 4 gates  -> gate activations, c_t, h_t
//   bwd B : A = dz[:, K slice],   W = rows of K (48 outputs), split-K         -> partial tile (k_bwd_b_red sums them)
// (the light phases -- projection, backward phase A -- stay on kernels.hip's K-split-over-waves kernels: their K is long and
// their output narrow, so one memory round trip + an LDS reduction beats a chunk pipeline).
// Round 1's kernels loaded everything, synchronised, then ran the MFMAs (19 us per generator diagonal, two rounds of 288
// workgroups at one per CU).  Here a workgroup is 64 rows x 48 columns with one wave per 16x16 output tile (12 waves, 3 per
// SIMD); A and W stream through a 4-stage LDS ring of 64-float K chunks filled by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPRs); the fragments of chunk c+1 are read into a second register set while the MFMAs of chunk c run; one raw s_barrier per
// chunk; counted vmcnt keeps two chunks in flight across it (cdna_hip_programming.md 5 "Pipelining across barriers").  Weights
// enter a CU once for all 64 rows, and a generator diagonal is ONE round of <= 256 workgroups.
// LDS image of a stage: (64 + 48) rows of SA4 = 17 float4 (16 data + 1 pad; an odd stride keeps the ds_read_b128 fragment reads
// of 16 consecutive rows at 2 LDS passes per lane group -- the best any row-major image gets, tools/ubench); the DMA image is
// linear in p = row*SA4 + c4, and lanes that fall on pad slots, rows >= N, columns past the job or k past the segment read a
// 16-byte zero word instead (exact zeros in the product, nothing stored).
// MFMA: v_mfma_f32_16x16x4_f32, both fragments float4 along k (lane l: row/col l&15, k-slot l>>4), exact fp32.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "../../rsrgan_amd/csrc/kernels.h"

namespace rsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ float g_pn_zeros[64];      // DMA source for padding / masked rows (zero-initialised)
constexpr int PN_KC = 4;               // 16-float k-blocks per chunk
constexpr int PN_KCF = PN_KC * 16;     // 64 floats of K per chunk
constexpr int PN_SA4 = PN_KC * 4 + 1;  // float4 per LDS image row (odd)

// tools/ubench includes this file with PN_ABLATE defined to measure the kernels with parts switched off (bit 1: no MFMA,
// 2: no DMA, 4: no fragment reads, 8: no epilogue, 16: no epilogue-operand prefetch, 32/64/128: early returns); the product
// build has no such code.
#ifdef PN_ABLATE
__device__ int g_pn_ablate = 0;
#define PN_ON(bit) (!(pn_ab & (bit)))
#define PN_AB_PARAM , int pn_ab
#define PN_AB_ARG , pn_ab
#define PN_AB_LOAD const int pn_ab = g_pn_ablate;
#else
#define PN_ON(bit) true
#define PN_AB_PARAM
#define PN_AB_ARG
#define PN_AB_LOAD
#endif

template <int NCT, int S_ = 4>
struct PG {
  static constexpr int S = S_;                       // ring stages: S-1 chunks are in flight or landed ahead of the MFMAs
  static constexpr int NC = 16 * NCT, NW = 4 * NCT, NT = 64 * NW, ROWS = 64 + NC, P4 = ROWS * PN_SA4;
  static constexpr int NI = (P4 + NT - 1) / NT;      // DMA instructions per wave per chunk
  static constexpr int STAGE4 = NI * NT;             // float4 slots per stage (the tail past P4 only ever receives zeros)
  static constexpr int LDS_BYTES = S * STAGE4 * 16;
  static constexpr int ZLD = NC + 1;                 // row stride of the accumulator exchange buffer (aliases the ring)
};

// per-lane DMA sources of the NI slots this lane fills in every chunk; K = segment 0 (k0 floats) then segment 1 (k1 floats)
template <int NCT>
struct Slots {
  const float* p0[PG<NCT>::NI];     // row pointer in segment 0, nullptr = zeros
  const float* p1[PG<NCT>::NI];     // row pointer in segment 1
  int kofs[PG<NCT>::NI];            // k offset of the slot inside a chunk, or a huge value for pad slots
};
struct KSeg { int k0, k1, nch0, nch; };      // segment lengths (floats, multiples of 4), chunks of segment 0, chunks in total
__device__ __forceinline__ KSeg pn_kseg(int k0, int k1) {
  KSeg s; s.k0 = k0; s.k1 = k1; s.nch0 = (k0 + PN_KCF - 1) / PN_KCF; s.nch = s.nch0 + (k1 + PN_KCF - 1) / PN_KCF;
  return s;
}

// FA(i, row, p0, p1): A row pointers of panel row `row` (0..63) for DMA slot i; FW(j, p0, p1): W row pointers of panel column j
template <int NCT, typename FA, typename FW>
__device__ __forceinline__ void pn_slots(Slots<NCT>& sl, int tid, FA fa, FW fw) {
  typedef PG<NCT> G;
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {
    const int p = i * G::NT + tid;
    const int row = p / PN_SA4, c4 = p - row * PN_SA4;
    sl.p0[i] = nullptr; sl.p1[i] = nullptr;
    sl.kofs[i] = (c4 < PN_KC * 4) ? c4 * 4 : (1 << 28);
    if (p < G::P4) {
      if (row < 64) fa(i, row, sl.p0[i], sl.p1[i]);
      else fw(row - 64, sl.p0[i], sl.p1[i]);
    }
  }
}

template <int NCT, int S>
__device__ __forceinline__ void pn_issue(float* smem, const Slots<NCT>& sl, int c, const KSeg& ks, const float* zeros, int w PN_AB_PARAM) {
  typedef PG<NCT, S> G;
  const bool s1 = c >= ks.nch0;                               // wave-uniform
  const int kb = (s1 ? c - ks.nch0 : c) * PN_KCF, klen = s1 ? ks.k1 : ks.k0;
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {
    const int k = kb + sl.kofs[i];
    const float* base = s1 ? sl.p1[i] : sl.p0[i];
    const float* src = (base != nullptr && k < klen) ? base + k : zeros;
#ifdef PN_ABLATE
    if (!PN_ON(512)) src = sl.p0[i] ? sl.p0[i] + (c * PN_KCF + (sl.kofs[i] & 63)) : zeros;
#endif
    if (PN_ON(2)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + ((size_t)(c % S) * G::STAGE4 + i * G::NT + w * 64) * 4), 16, 0, 0);
  }
}
__device__ __forceinline__ int pn_nkb(int c, const KSeg& ks) {        // k-blocks of chunk c that hold data (wave-uniform)
  const bool s1 = c >= ks.nch0;
  const int rem = (s1 ? ks.k1 - (c - ks.nch0) * PN_KCF : ks.k0 - c * PN_KCF);
  return min(PN_KC, (rem + 15) >> 4);
}
__device__ __forceinline__ int pn_nkb_dummy() { return PN_KC;
}

template <int OFF>
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
struct Frag { f32x4 a[PN_KC], b[PN_KC]; };
__device__ __forceinline__ void pn_read(Frag& f, unsigned sa, unsigned sb PN_AB_PARAM) {
#ifdef PN_ABLATE
  for (int i = 0; i < PN_KC; ++i) { f.a[i] = f32x4{1.f, 1.f, 1.f, 1.f}; f.b[i] = f.a[i]; }
  if (!PN_ON(4)) return;
#endif
  f.a[0] = lds_read16<0>(sa); f.b[0] = lds_read16<0>(sb);
  f.a[1] = lds_read16<64>(sa); f.b[1] = lds_read16<64>(sb);
  f.a[2] = lds_read16<128>(sa); f.b[2] = lds_read16<128>(sb);
  f.a[3] = lds_read16<192>(sa); f.b[3] = lds_read16<192>(sb);
}
template <int KB0, int KB1>
__device__ __forceinline__ void pn_mfma(f32x4& acc0, f32x4& acc1, const Frag& f, int nkb PN_AB_PARAM) {
#pragma unroll
  for (int kb = KB0; kb < KB1; ++kb) {
    if (kb < nkb && PN_ON(1)) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[kb].x, f.b[kb].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[kb].y, f.b[kb].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[kb].z, f.b[kb].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[kb].w, f.b[kb].w, acc1, 0, 0, 0);
    }
  }
}
// wait until at most `chunks` of this wave's newest DMA chunks are outstanding (wave-uniform argument)
template <int NI>
__device__ __forceinline__ void pn_wait_dma(int chunks) {
  if (chunks >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");
  else if (chunks == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
  else if (chunks == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One pipeline step: the MFMAs of chunk c (fragments in `cur`) around the hand-over of chunk c+1 (-> `nxt`).
//  * DMA of chunk c+S-1 goes first: its stage held chunk c-1, whose fragment reads every wave completed before it passed the
//    barrier of step c-1, so no barrier is needed in front of it;
//  * the barrier sits between the two MFMA halves: after it the wave only issues 8 ds_reads and goes back to MFMAs.
template <int NCT, int S>
__device__ __forceinline__ void pn_step(f32x4& acc0, f32x4& acc1, Frag& cur, Frag& nxt, float* smem, const Slots<NCT>& sl, int c,
                                        const KSeg& ks, const float* zeros, int w, unsigned aoff, unsigned boff PN_AB_PARAM) {
  typedef PG<NCT, S> G;
  if (c + S - 1 < ks.nch) pn_issue<NCT, S>(smem, sl, c + S - 1, ks, zeros, w PN_AB_ARG);
  int nkb = pn_nkb(c, ks);
#ifdef PN_ABLATE
  if (!PN_ON(256)) nkb = PN_KC;
#endif
  pn_mfma<0, PN_KC / 2>(acc0, acc1, cur, nkb PN_AB_ARG);
  if (c + 1 < ks.nch) {
    pn_wait_dma<G::NI>(min(ks.nch - (c + 2), S - 2));       // chunk c+1 of this wave has landed
    __builtin_amdgcn_s_barrier();                           // ... and everyone else's part of it
    const unsigned so = (unsigned)((c + 1) % S) * (G::STAGE4 * 16);
#ifdef PN_ABLATE
    if (!PN_ON(1024)) pn_read(nxt, (aoff & 0xff) + so, (aoff & 0xff) + so PN_AB_ARG); else
#endif
    pn_read(nxt, aoff + so, boff + so PN_AB_ARG);
  }
  pn_mfma<PN_KC / 2, PN_KC>(acc0, acc1, cur, nkb PN_AB_ARG);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// acc (16x16 tile of this wave: rows rt*16.., columns ct*16..) = sum over K.  On return every wave has passed a barrier after its
// last LDS read, so the ring may be overwritten.
template <int NCT, int S>
__device__ __forceinline__ f32x4 pn_product(float* smem, const Slots<NCT>& sl, const KSeg& ks, const float* zeros, int w, int lane PN_AB_PARAM) {
  typedef PG<NCT, S> G;
  const int rt = w & 3, ct = w >> 2, lr = lane & 15, q = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned aoff = lds0 + (unsigned)(((rt * 16 + lr) * PN_SA4 + q) * 16);
  const unsigned boff = lds0 + (unsigned)(((64 + ct * 16 + lr) * PN_SA4 + q) * 16);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (ks.nch <= 0) return acc0;
#pragma unroll
  for (int i = 0; i < S - 1; ++i)
    if (i < ks.nch) pn_issue<NCT, S>(smem, sl, i, ks, zeros, w PN_AB_ARG);
  pn_wait_dma<G::NI>(min(ks.nch - 1, S - 2));
  __builtin_amdgcn_s_barrier();
  Frag fa, fb;
  pn_read(fa, aoff, boff PN_AB_ARG);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < ks.nch; c += 2) {
    pn_step<NCT, S>(acc0, acc1, fa, fb, smem, sl, c, ks, zeros, w, aoff, boff PN_AB_ARG);
    if (c + 1 < ks.nch) pn_step<NCT, S>(acc0, acc1, fb, fa, smem, sl, c + 1, ks, zeros, w, aoff, boff PN_AB_ARG);
  }
  __builtin_amdgcn_s_barrier();
  return acc0 + acc1;
}

// accumulators -> zs[64][NC+1] (aliases the ring; the caller syncs before reading)
template <int NCT>
__device__ __forceinline__ void pn_spill(float* smem, const f32x4& acc, int w, int lane) {
  typedef PG<NCT> G;
  const int rt = w & 3, ct = w >> 2, lr = lane & 15, q = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) smem[(rt * 16 + q * 4 + r) * G::ZLD + ct * 16 + lr] = acc[r];
}

__device__ __forceinline__ float pn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename JobT>
__device__ __forceinline__ int pn_find_job(const JobT* j, int n, int bid) {
  int ji = 0;
#pragma unroll
  for (int q = 1; q < MAXJ; ++q)
    if (q < n && bid >= j[q].blk_base) ji = q;
  return ji;
}
// blocks of a job: nrg row groups x roundup8(ncg) virtual column groups (block id % 8 == column group % 8)
__device__ __forceinline__ bool pn_tile(int lb, int ncg, int& cg, int& rg) {
  const int ncg8 = (ncg + 7) & ~7;
  cg = lb % ncg8; rg = lb / ncg8;
  return cg < ncg;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward phase 1 (LSTMCell gates + cell update), 64 rows x 12 cells x 4 gates per workgroup, 12 waves
// ---------------------------------------------------------------------------------------------------------------------------
template <int GT_NCT, int GT_S>
__global__ __launch_bounds__(256 * GT_NCT) void k_pn_gates(const FwdGateJobs jobs) {
  typedef PG<GT_NCT, GT_S> G;
  constexpr int GT_NCELL = 4 * GT_NCT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  PN_AB_LOAD
  if (!PN_ON(32)) return;
  const FwdGateJob& J = jobs.j[pn_find_job(jobs.j, jobs.n, bid)];
  const int H = J.H, N = J.N, H4 = 4 * H;
  int cg, rg;
  if (!pn_tile(bid - J.blk_base, (H + GT_NCELL - 1) / GT_NCELL, cg, rg)) return;
  if (!PN_ON(64)) { if (H == 123456) J.h[0] = 1.f; return; }
  const int r0 = rg * 64, c0 = cg * GT_NCELL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ldx = J.x ? J.ldx : 0, ldm = J.ldm;

  // epilogue operands of this thread's (row, cell), requested before the product
  const int er = tid / GT_NCELL, ecl = tid - er * GT_NCELL, erow = r0 + er, ecell = c0 + ecl;
  const bool evalid = erow < N && ecell < H;
  float zb[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f, pwi = 0.f, pwf = 0.f, pwo = 0.f, mprev = 0.f, resin = 0.f;
  int elen = 0;
  if (evalid && PN_ON(16)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) zb[g] = J.zx ? J.zx[(size_t)erow * H4 + g * H + ecell] : J.bias[g * H + ecell];
    cp = J.c_prev[(size_t)erow * H + ecell];
    elen = J.len[erow];
    pwi = J.wi[ecell]; pwf = J.wf[ecell]; pwo = J.wo[ecell];
    if (J.np_m_out) {
      mprev = J.m[(size_t)erow * ldm + ecell];
      if (J.np_res_out) resin = J.np_res_in[(size_t)erow * ldm + ecell];
    }
  }
  Slots<GT_NCT> sl;
  {
    const float* jx = J.x; const float* jm = J.m; const float* jkx = J.KxT; const float* jkh = J.KhT;
    pn_slots<GT_NCT>(sl, tid,
        [=](int, int row, const float*& p0, const float*& p1) {
          const int arow = r0 + row;
          if (arow < N) { p0 = ldx ? jx + (size_t)arow * ldx : nullptr; p1 = jm + (size_t)arow * ldm; }
        },
        [=](int j, const float*& p0, const float*& p1) {
          const int gate = j / GT_NCELL, cell = c0 + (j - gate * GT_NCELL);
          if (cell < H) {
            const size_t gcol = (size_t)gate * H + cell;
            p0 = ldx ? jkx + gcol * ldx : nullptr; p1 = jkh + gcol * ldm;
          }
        });
  }
  if (!PN_ON(128)) { if (sl.p0[0] == (const float*)16 || sl.p1[0] == (const float*)16 || zb[0] == 123.456f) J.h[0] = cp; return; }
  const f32x4 acc = pn_product<GT_NCT, GT_S>(smem, sl, pn_kseg(ldx, ldm), g_pn_zeros, w, lane PN_AB_ARG);
  if (!PN_ON(8)) { if (acc[0] == 123.456f) J.h[0] = acc[1]; return; }
  pn_spill<GT_NCT>(smem, acc, w, lane);
  __syncthreads();
  if (!evalid) return;
  float z[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) z[g] = zb[g] + smem[er * G::ZLD + g * GT_NCELL + ecl];
  const size_t ci = (size_t)erow * H + ecell;
  float* go_ = J.gates + (size_t)erow * H4 + ecell;
  const size_t mi = (size_t)erow * ldm + ecell;
  if (J.t < elen) {
    const float gi = pn_sigmoid(z[0] + pwi * cp);
    const float gf = pn_sigmoid(z[2] + jobs.forget_bias + pwf * cp);
    const float gj = tanhf(z[1]);
    const float cn = gf * cp + gi * gj;
    const float go = pn_sigmoid(z[3] + pwo * cn);
    J.c_out[ci] = cn;
    go_[0] = gi; go_[H] = gj; go_[2 * H] = gf; go_[3 * H] = go;
    const float hh = go * tanhf(cn);
    J.h[(size_t)erow * J.ldh + ecell] = hh;
    if (J.np_m_out) {
      J.np_m_out[mi] = hh; J.np_out[mi] = hh;
      if (J.np_res_out) J.np_res_out[mi] = hh + resin;
    }
  } else {                       // dynamic_rnn: t >= len -> state copied through, zero output, no gradient
    J.c_out[ci] = cp;
    go_[0] = 0.f; go_[H] = 0.f; go_[2 * H] = 0.f; go_[3 * H] = 0.f;
    J.h[(size_t)erow * J.ldh + ecell] = 0.f;
    if (J.np_m_out) {
      J.np_m_out[mi] = mprev; J.np_out[mi] = 0.f;
      if (J.np_res_out) J.np_res_out[mi] = resin;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward phase B, split-K: partial tile [64 rows x 48 outputs] of dz_t[:, slice] . K[:, slice]^T -> ws[ks][N][ldw];
// k_bwd_b_red (kernels.hip) sums the KG partials in fixed order and applies the dynamic_rnn mask.  12 waves.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int BB_NCT = 3, BB_NC = 16 * BB_NCT;
__global__ __launch_bounds__(PG<BB_NCT>::NT) void k_pn_bwd_b(const BwdBJobs jobs) {
  typedef PG<BB_NCT> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  PN_AB_LOAD
  int ji = 0;
#pragma unroll
  for (int qq = 1; qq < MAXJ; ++qq)
    if (qq < jobs.n && bid >= jobs.j[qq].blk_base_p) ji = qq;
  const BwdBJob& J = jobs.j[ji];
  const int lb = bid - J.blk_base_p;
  const int per_ks = J.ncg * J.nrg;
  const int ks = lb / per_ks, rem = lb - ks * per_ks;
  const int rg = rem / J.ncg, cg = rem - rg * J.ncg;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = J.N, H4 = J.H4;
  const int kbeg = ks * J.kpg * PN_KCF, klen = min(H4 - kbeg, J.kpg * PN_KCF);     // kpg = chunks per K slice
  const int r0 = rg * 64, n0 = J.n_begin + cg * BB_NC;
  Slots<BB_NCT> sl;
  {
    const float* jdz = J.dz; const float* jk = J.K; const int nend = J.n_end;
    pn_slots<BB_NCT>(sl, tid,
        [=](int, int row, const float*& p0, const float*& p1) { if (r0 + row < N) p0 = jdz + (size_t)(r0 + row) * H4 + kbeg; },
        [=](int j, const float*& p0, const float*& p1) { if (n0 + j < nend) p0 = jk + (size_t)(n0 + j) * H4 + kbeg; });
  }
  const f32x4 acc = pn_product<BB_NCT, 4>(smem, sl, pn_kseg(klen, 0), g_pn_zeros, w, lane PN_AB_ARG);
  pn_spill<BB_NCT>(smem, acc, w, lane);
  __syncthreads();
  for (int e = tid; e < 64 * (BB_NC / 4); e += G::NT) {
    const int row = e / (BB_NC / 4), c4 = (e - row * (BB_NC / 4)) * 4;
    const int grow = r0 + row, gcol = cg * BB_NC + c4;
    if (grow >= N || gcol >= J.ldw) continue;
    float* dst = J.ws + ((size_t)ks * N + grow) * J.ldw + gcol;
    const float* z = smem + row * G::ZLD + c4;
    *reinterpret_cast<float4*>(dst) = make_float4(z[0], z[1], z[2], z[3]);      // ldw is a multiple of 4; columns >= ncols hold exact zeros
  }
}

// ------------------------------------------------------------------------------------------------------------ host side
static bool g_panel = false;     // RSRGAN_PANEL=1 selects the panel kernels for gates / backward B (measured: no faster)
bool panel_kernels() {
  static int init = -1;
  if (init < 0) { const char* e = getenv("RSRGAN_PANEL"); g_panel = e && atoi(e) != 0; init = 1; }
  return g_panel;
}
template <typename K>
static void pn_attr(K kern, int bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
static void pn_init() {
  static bool done = false;
  if (done) return;
  typedef PG<3, 4> G34; typedef PG<1, 3> G13; typedef PG<1, 2> G12;
  pn_attr(&k_pn_gates<3, 4>, G34::LDS_BYTES); pn_attr(&k_pn_gates<1, 3>, G13::LDS_BYTES); pn_attr(&k_pn_gates<1, 2>, G12::LDS_BYTES);
  pn_attr(&k_pn_bwd_b, PG<BB_NCT>::LDS_BYTES);
  done = true;
}
int pn_blocks(int ncols, int per_wg, int N) { return ((((ncols + per_wg - 1) / per_wg) + 7) & ~7) * ((N + 63) / 64); }
static int g_gates_nct = 3, g_gates_s = 4;        // RSRGAN_PN_GATES="nct,s": 64 rows x 16*nct columns per workgroup, s ring stages
static void pn_gates_cfg() {
  static bool done = false;
  if (done) return;
  if (const char* e = getenv("RSRGAN_PN_GATES")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && a == 1 && (b == 2 || b == 3)) { g_gates_nct = a; g_gates_s = b; } }
  done = true;
}
void pn_set_gates_cfg(int nct, int s) { pn_gates_cfg(); if (nct == 1 && (s == 2 || s == 3)) { g_gates_nct = 1; g_gates_s = s; } else { g_gates_nct = 3; g_gates_s = 4; } }
int pn_gates_blocks(int H, int N) { pn_gates_cfg(); return pn_blocks(H, 4 * g_gates_nct, N); }

void launch_pn_gates(const FwdGateJobs& jobs, int total_blocks, hipStream_t s) {
  pn_init(); pn_gates_cfg();
  typedef PG<3, 4> G34; typedef PG<1, 3> G13; typedef PG<1, 2> G12;
  if (g_gates_nct == 3) hipLaunchKernelGGL((k_pn_gates<3, 4>), dim3(total_blocks), dim3(G34::NT), G34::LDS_BYTES, s, jobs);
  else if (g_gates_s == 3) hipLaunchKernelGGL((k_pn_gates<1, 3>), dim3(total_blocks), dim3(G13::NT), G13::LDS_BYTES, s, jobs);
  else hipLaunchKernelGGL((k_pn_gates<1, 2>), dim3(total_blocks), dim3(G12::NT), G12::LDS_BYTES, s, jobs);
}
// split-K plan of backward phase B for the panel kernel: KG slices of kpg 96-float chunks, 48-column groups, 64-row groups
size_t pn_bwd_b_plan(BwdBJobs& jobs, float* ws_base) {
  size_t off = 0;
  int bp = 0, br = 0;
  for (int i = 0; i < jobs.n; ++i) {
    BwdBJob& b = jobs.j[i];
    const int nch = (b.H4 + PN_KCF - 1) / PN_KCF, ncols = b.n_end - b.n_begin;
    int KG = std::max(1, (nch + 6) / 7);               // about 7 chunks (448 floats of K) per workgroup
    b.kpg = (nch + KG - 1) / KG;
    b.KG = (nch + b.kpg - 1) / b.kpg;
    b.ncg = (ncols + BB_NC - 1) / BB_NC; b.nrg = (b.N + 63) / 64;
    b.ldw = (ncols + 3) & ~3;
    b.ws = ws_base ? ws_base + off : nullptr;
    off += (size_t)b.KG * b.N * b.ldw;
    b.blk_base_p = bp; bp += b.KG * b.ncg * b.nrg;
    b.blk_base_r = br; br += (b.N * ncols + 255) / 256;
  }
  return off;
}
int pn_bwd_b_blocks(const BwdBJobs& jobs) {
  int bp = 0;
  for (int i = 0; i < jobs.n; ++i) bp = std::max(bp, jobs.j[i].blk_base_p + jobs.j[i].KG * jobs.j[i].ncg * jobs.j[i].nrg);
  return bp;
}
void launch_pn_bwd_b(const BwdBJobs& jobs, hipStream_t s) {
  pn_init();
  hipLaunchKernelGGL(k_pn_bwd_b, dim3(pn_bwd_b_blocks(jobs)), dim3(PG<BB_NCT>::NT), PG<BB_NCT>::LDS_BYTES, s, jobs);
  launch_bwd_b_red(jobs, s);
}

}  // namespace rsr
