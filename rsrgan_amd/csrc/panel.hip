// panel.hip -- the recurrent step kernels of round 2: ONE pipelined 64-row "panel" MFMA product with four epilogues.
//
// Every phase of an LSTMP time step (models/lstm.py:89-112; cell math models/BNLSTMCell.py:176-217) is a product
//     C[64 rows x NC columns] = A[64 x K] . W^T,   A rows and W rows both k-contiguous in HBM,
// with M = the batch rows of one tower (64 at BASELINE.json's size), followed by pointwise work:
//   gates : A = [x_t | m_{t-1}],           W = rows of KxT|KhT of 12 cells x 4 gates -> gate activations, c_t, h_t
//   proj  : A = h_t,                       W = rows of WpT (16 outputs)              -> m_t, dynamic_rnn masking, residual, FC stage
//   bwd A : A = mask*[dm_state | dout],    W = rows of Wp|Wp (16 cells)              -> dh -> gate gradients dz, dc
//   bwd B : A = dz[:, K slice],            W = rows of K (48 outputs), split-K       -> partial tile (k_bwd_b_red sums them)
// Round 1's kernels loaded everything, synchronised, then ran the MFMAs (19 us per generator diagonal, two rounds of 288
// workgroups at one per CU).  Here a workgroup is 64 rows x NC columns with one wave per 16x16 output tile (4 x NCT waves), and A
// and W stream through a 3-stage LDS ring of 96-float K chunks filled by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs), one
// raw s_barrier per chunk, counted vmcnt so that the next chunk stays in flight across the barrier
// (cdna_hip_programming.md 5 "Pipelining across barriers"): operand pull (~12 B/clk/CU from the fabric, profiles/r2_ubench_*)
// overlaps the MFMAs, weights enter a CU once for all 64 rows, and a generator diagonal is ONE round of <= 256 workgroups.
// LDS image of a stage: (64 + NC) rows of SA4 = 25 float4 (24 data + 1 pad; odd stride keeps the ds_read_b128 fragment reads of 16
// consecutive rows at 2 LDS passes per lane group); the DMA image is linear in p = row*SA4 + c4, lanes that fall on pad slots,
// rows >= N, columns past the job or k >= K read a 16-byte zero word instead (exact zeros in the product, nothing stored).
// MFMA: v_mfma_f32_16x16x4_f32, both fragments float4 along k (lane l: row/col l&15, k-slot l>>4), exact fp32.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace rsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int PN_KC = 6;               // 16-float k-blocks per chunk
constexpr int PN_KCF = PN_KC * 16;     // 96 floats of K per chunk
constexpr int PN_SA4 = PN_KC * 4 + 1;  // float4 per LDS image row (odd)
constexpr int PN_S = 3;                // ring stages

// tools/ubench includes this file with PN_ABLATE defined to measure the kernels with parts switched off (bit 1: no MFMA,
// 2: no DMA, 4: no fragment reads, 8: no epilogue, 16: no epilogue-operand prefetch); the product build has no such code.
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

template <int NCT>
struct PG {
  static constexpr int NC = 16 * NCT, NW = 4 * NCT, NT = 64 * NW, ROWS = 64 + NC, P4 = ROWS * PN_SA4;
  static constexpr int NI = (P4 + NT - 1) / NT;      // DMA instructions per wave per chunk
  static constexpr int STAGE4 = NI * NT;             // float4 slots per stage (the tail past P4 only ever receives zeros)
  static constexpr int LDS_BYTES = PN_S * STAGE4 * 16;
  static constexpr int ZLD = NC + 1;                 // row stride of the accumulator exchange buffer (aliases the ring)
};

// per-lane DMA sources of the NI slots this lane fills in every chunk
template <int NCT>
struct Slots {
  const float* p0[PG<NCT>::NI];     // segment 0 row pointer (k < ka0), nullptr = zeros
  const float* p1[PG<NCT>::NI];     // segment 1 row pointer, pre-offset by -ka0
  int kofs[PG<NCT>::NI];            // k offset of the slot inside a chunk, or a huge value for pad slots
};

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

template <int NCT>
__device__ __forceinline__ void pn_issue(float* smem, int st, const Slots<NCT>& sl, int c, int ka0, int ktot, const float* zeros, int w PN_AB_PARAM) {
  typedef PG<NCT> G;
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {
    const int k = c * PN_KCF + sl.kofs[i];
    const float* base = k < ka0 ? sl.p0[i] : sl.p1[i];
    const float* src = (base != nullptr && k < ktot) ? base + k : zeros;
    if (PN_ON(2)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + ((size_t)st * G::STAGE4 + i * G::NT + w * 64) * 4), 16, 0, 0);
  }
}

template <int OFF>
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// acc (16x16 tile of this wave: rows rt*16.., columns ct*16..) = sum over K.  On return every wave has passed a barrier after its
// last LDS read, so the ring may be overwritten.
template <int NCT>
__device__ __forceinline__ f32x4 pn_product(float* smem, const Slots<NCT>& sl, int ka0, int ktot, const float* zeros, int w, int lane PN_AB_PARAM) {
  typedef PG<NCT> G;
  const int rt = w & 3, ct = w >> 2, lr = lane & 15, q = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned aoff = lds0 + (unsigned)(((rt * 16 + lr) * PN_SA4 + q) * 16);
  const unsigned boff = lds0 + (unsigned)(((64 + ct * 16 + lr) * PN_SA4 + q) * 16);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nch = (ktot + PN_KCF - 1) / PN_KCF;
  if (nch > 0) pn_issue<NCT>(smem, 0, sl, 0, ka0, ktot, zeros, w PN_AB_ARG);
  if (nch > 1) pn_issue<NCT>(smem, 1, sl, 1, ka0, ktot, zeros, w PN_AB_ARG);
  int st = 0;
  for (int c = 0; c < nch; ++c) {
    // chunk c of THIS wave has landed once at most the NI instructions of chunk c+1 are outstanding
    if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // everyone's part of chunk c landed; everyone finished reading stage (c-1)%3
    if (c + 2 < nch) pn_issue<NCT>(smem, st == 0 ? 2 : st - 1, sl, c + 2, ka0, ktot, zeros, w PN_AB_ARG);
    const unsigned sa = aoff + (unsigned)st * (G::STAGE4 * 16), sb = boff + (unsigned)st * (G::STAGE4 * 16);
    f32x4 a[PN_KC], b[PN_KC];
#ifdef PN_ABLATE
    for (int i = 0; i < PN_KC; ++i) { a[i] = f32x4{1.f, 1.f, 1.f, 1.f}; b[i] = a[i]; }
    if (PN_ON(4)) {
#endif
    a[0] = lds_read16<0>(sa); b[0] = lds_read16<0>(sb);
    a[1] = lds_read16<64>(sa); b[1] = lds_read16<64>(sb);
    a[2] = lds_read16<128>(sa); b[2] = lds_read16<128>(sb);
    a[3] = lds_read16<192>(sa); b[3] = lds_read16<192>(sb);
    a[4] = lds_read16<256>(sa); b[4] = lds_read16<256>(sb);
    a[5] = lds_read16<320>(sa); b[5] = lds_read16<320>(sb);
#ifdef PN_ABLATE
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int nkb = min(PN_KC, (ktot - c * PN_KCF + 15) >> 4);        // wave-uniform: k-blocks of this chunk that hold data
#pragma unroll
    for (int kb = 0; kb < PN_KC; ++kb) {
      if (kb < nkb && PN_ON(1)) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].x, b[kb].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].y, b[kb].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].z, b[kb].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].w, b[kb].w, acc1, 0, 0, 0);
      }
    }
    st = st == 2 ? 0 : st + 1;
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
constexpr int GT_NCT = 3, GT_NCELL = 4 * GT_NCT;
__global__ __launch_bounds__(PG<GT_NCT>::NT) void k_pn_gates(const FwdGateJobs jobs) {
  typedef PG<GT_NCT> G;
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
  const int ldx = J.x ? J.ldx : 0, ldm = J.ldm, ktot = ldx + ldm;

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
          if (arow < N) { p0 = ldx ? jx + (size_t)arow * ldx : nullptr; p1 = jm + (size_t)arow * ldm - ldx; }
        },
        [=](int j, const float*& p0, const float*& p1) {
          const int gate = j / GT_NCELL, cell = c0 + (j - gate * GT_NCELL);
          if (cell < H) {
            const size_t gcol = (size_t)gate * H + cell;
            p0 = ldx ? jkx + gcol * ldx : nullptr; p1 = jkh + gcol * ldm - ldx;
          }
        });
  }
  if (!PN_ON(128)) { if (sl.p0[0] == (const float*)16 || sl.p1[3] == (const float*)16 || zb[0] == 123.456f) J.h[0] = cp; return; }
  const f32x4 acc = pn_product<GT_NCT>(smem, sl, ldx, ktot, jobs.zeros, w, lane PN_AB_ARG);
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
// forward phase 2 (projection m_t = h_t.Wp, dynamic_rnn masking, residual add; also the per-step fully_connected stage),
// 64 rows x 16 outputs per workgroup, 4 waves
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PJ_NCT = 1;
__global__ __launch_bounds__(PG<PJ_NCT>::NT) void k_pn_proj(const FwdProjJobs jobs) {
  typedef PG<PJ_NCT> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  PN_AB_LOAD
  const FwdProjJob& J = jobs.j[pn_find_job(jobs.j, jobs.n, bid)];
  const int N = J.N, P = J.P;
  int cg, rg;
  if (!pn_tile(bid - J.blk_base, (P + 15) / 16, cg, rg)) return;
  const int r0 = rg * 64, c0 = cg * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ldh = J.ldh;
  // 4 outputs per thread: (row = e >> 4, column = e & 15), e = tid + 256 u
  float pm[4], pres[4], pnoise[4], pbias = 0.f;
  int plen[4];
  const int ecc = tid & 15, epp = c0 + ecc;
  if (epp < P && J.bias) pbias = J.bias[epp];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = r0 + ((tid + 256 * u) >> 4);
    pm[u] = 0.f; pres[u] = 0.f; pnoise[u] = 0.f; plen[u] = 1 << 30;
    if (row < N && epp < P) {
      const size_t mi = (size_t)row * J.ldm + epp;
      if (J.m_prev) pm[u] = J.m_prev[mi];
      if (J.res_out) pres[u] = J.res_in[mi];
      if (J.noise) pnoise[u] = J.noise[(size_t)row * P + epp];
      if (J.len) plen[u] = J.len[row];
    }
  }
  Slots<PJ_NCT> sl;
  {
    const float* jh = J.h; const float* jw = J.WpT;          // scalar copies: a lane-dependent choice between J.h and J.WpT must not
    pn_slots<PJ_NCT>(sl, tid,                                // turn into a vector load of the pointer from the kernarg segment
        [=](int, int row, const float*& p0, const float*& p1) { if (r0 + row < N) p0 = jh + (size_t)(r0 + row) * ldh; },
        [=](int j, const float*& p0, const float*& p1) { if (c0 + j < P) p0 = jw + (size_t)(c0 + j) * ldh; });
  }
  const f32x4 acc = pn_product<PJ_NCT>(smem, sl, ldh, ldh, jobs.zeros, w, lane PN_AB_ARG);
  pn_spill<PJ_NCT>(smem, acc, w, lane);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int er = (tid + 256 * u) >> 4, row = r0 + er;
    if (row >= N || epp >= P) continue;
    const float v = smem[er * G::ZLD + ecc] + pbias;
    const bool live = J.t < plen[u];
    const size_t mi = (size_t)row * J.ldm + epp;
    J.m_out[mi] = live ? v : pm[u];
    J.out[(size_t)row * J.ldo + epp] = (live ? v : 0.f) + pnoise[u];
    if (J.res_out) J.res_out[mi] = (live ? v : 0.f) + pres[u];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward phase A: dm = mask*(dout_t + dm_state); dh = dm.Wp^T; gate gradients -> dz (in place of the activations), dc.
// 64 rows x 16 cells per workgroup, 4 waves; the product runs over K = [dm_state | dout] against [Wp | Wp].
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int BA_NCT = 1;
__global__ __launch_bounds__(PG<BA_NCT>::NT) void k_pn_bwd_a(const BwdAJobs jobs) {
  typedef PG<BA_NCT> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  PN_AB_LOAD
  const BwdAJob& J = jobs.j[pn_find_job(jobs.j, jobs.n, bid)];
  const int N = J.N, H = J.H, H4 = 4 * H, ldm = J.ldm;
  const int ncg = (H + 15) / 16;
  int cg, rg;
  if (!pn_tile(bid - J.blk_base, ncg, cg, rg)) return;
  const int r0 = rg * 64, c0 = cg * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool noproj = J.Wp == nullptr;                    // num_proj=None: dh = mask*(dout + dm_state), no product
  const float* jdm = J.dmst; const float* jdo = J.dout; const float* jwp = J.Wp; const int* jlen = J.len; const int jt = J.t;
  // (1) lengths of the A rows this lane's DMA slots cover (unconditional, clamped: all in flight together)
  int slen[G::NI];
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {
    const int row = min((i * G::NT + tid) / PN_SA4, 63);
    slen[i] = jlen[min(r0 + row, N - 1)];
  }
  // (2) this workgroup's share of dmt = mask*(dout + dm_state) (operand of the projection's weight gradient): the ldm/4 float4
  //     columns are dealt over the column groups, normally one float4 per thread, loaded now and stored after the product
  const int l4 = ldm >> 2, per = (l4 + ncg - 1) / ncg;
  const int d_row = tid / per, d_c4 = cg * per + (tid - d_row * per), d_arow = r0 + d_row;
  const bool d_ok = !noproj && tid < 64 * per && d_arow < N && d_c4 < l4;
  // every load below is unconditional (clamped indices, a zero word where the job has no dout): hipcc's vmcnt bookkeeping is
  // conservative across branches, and one conditional load here makes the first use of slen[] wait for ALL of them
  const float* zsrc = jobs.zeros;
  const size_t d_off = (size_t)min(d_arow, N - 1) * ldm + min(d_c4, l4 - 1) * 4;
  const int d_len = jlen[min(d_arow, N - 1)];
  const float4 d_v = *reinterpret_cast<const float4*>(jdm + d_off);
  const float4 d_d = *reinterpret_cast<const float4*>(jdo ? jdo + d_off : zsrc);
  // (3) epilogue operands: 4 (row, cell) elements per thread
  const int ecc = tid & 15, ecell = c0 + ecc, ecellc = min(ecell, H - 1);
  float eg[4][4], ecp[4], ecn[4], edc[4], edh0[4], edh1[4];
  int elen[4];
  const float ewo = J.wo[ecellc], ewi = J.wi[ecellc], ewf = J.wf[ecellc];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int rowc = min(r0 + ((tid + 256 * u) >> 4), N - 1);
    const float* g = J.gates + (size_t)rowc * H4 + ecellc;
    eg[u][0] = g[0]; eg[u][1] = g[H]; eg[u][2] = g[2 * H]; eg[u][3] = g[3 * H];
    const size_t ci = (size_t)rowc * H + ecellc;
    ecp[u] = J.c_prev[ci]; ecn[u] = J.c_cur[ci]; edc[u] = J.dc[ci];
    elen[u] = jlen[rowc];
    edh0[u] = 0.f; edh1[u] = 0.f;
    if (noproj) {                      // (num_proj=None layers skip the product, so nothing waits on these)
      edh0[u] = jdm[(size_t)rowc * ldm + ecellc];
      edh1[u] = jdo ? jdo[(size_t)rowc * ldm + ecellc] : 0.f;
    }
  }
  if (!noproj) {
    const int ktot = jdo ? 2 * ldm : ldm;
    Slots<BA_NCT> sl;
    pn_slots<BA_NCT>(sl, tid,
        [=](int i, int row, const float*& p0, const float*& p1) {
          const int arow = r0 + row;
          if (arow < N && jt < slen[i]) {
            p0 = jdm + (size_t)arow * ldm;
            p1 = jdo ? jdo + (size_t)arow * ldm - ldm : nullptr;
          }
        },
        [=](int j, const float*& p0, const float*& p1) {
          if (c0 + j < H) { p0 = jwp + (size_t)(c0 + j) * ldm; p1 = p0 - ldm; }
        });
    const f32x4 acc = pn_product<BA_NCT>(smem, sl, ldm, ktot, jobs.zeros, w, lane PN_AB_ARG);
    pn_spill<BA_NCT>(smem, acc, w, lane);
    __syncthreads();
    if (d_ok) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (jt < d_len) v = make_float4(d_v.x + d_d.x, d_v.y + d_d.y, d_v.z + d_d.z, d_v.w + d_d.w);
      *reinterpret_cast<float4*>(J.dmt + (size_t)d_arow * ldm + d_c4 * 4) = v;
    }
    for (int e = tid + G::NT; e < 64 * per; e += G::NT) {        // only when H is tiny (ldm/4 columns over very few column groups)
      const int row = e / per, c4 = cg * per + (e - row * per), arow = r0 + row;
      if (arow >= N || c4 >= l4) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (jt < jlen[arow]) {
        v = *reinterpret_cast<const float4*>(jdm + (size_t)arow * ldm + c4 * 4);
        if (jdo) {
          const float4 d = *reinterpret_cast<const float4*>(jdo + (size_t)arow * ldm + c4 * 4);
          v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
        }
      }
      *reinterpret_cast<float4*>(J.dmt + (size_t)arow * ldm + c4 * 4) = v;
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int er = (tid + 256 * u) >> 4, row = r0 + er;
    if (row >= N || ecell >= H) continue;
    float* g = J.gates + (size_t)row * H4 + ecell;
    if (jt < elen[u]) {
      const float dh = noproj ? edh0[u] + edh1[u] : smem[er * G::ZLD + ecc];
      const float gi = eg[u][0], gj = eg[u][1], gf = eg[u][2], go = eg[u][3];
      const float tc = tanhf(ecn[u]);
      const float dao = dh * tc * go * (1.f - go);
      const float dcn = edc[u] + dh * go * (1.f - tc * tc) + dao * ewo;
      const float daf = dcn * ecp[u] * gf * (1.f - gf);
      const float dai = dcn * gj * gi * (1.f - gi);
      const float dj = dcn * gi * (1.f - gj * gj);
      J.dc[(size_t)row * H + ecell] = dcn * gf + dai * ewi + daf * ewf;
      g[0] = dai; g[H] = dj; g[2 * H] = daf; g[3 * H] = dao;
    } else {
      g[0] = 0.f; g[H] = 0.f; g[2 * H] = 0.f; g[3 * H] = 0.f;       // dc passes through unchanged
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
  const f32x4 acc = pn_product<BB_NCT>(smem, sl, klen, klen, jobs.zeros, w, lane PN_AB_ARG);
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
static bool g_panel = true;      // RSRGAN_PANEL=0 selects round 1's step kernels (kernels.hip) for A/B measurements
bool panel_kernels() {
  static int init = -1;
  if (init < 0) { const char* e = getenv("RSRGAN_PANEL"); g_panel = !(e && atoi(e) == 0); init = 1; }
  return g_panel;
}
template <typename K>
static void pn_attr(K kern, int bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
static void pn_init() {
  static bool done = false;
  if (done) return;
  pn_attr(&k_pn_gates, PG<GT_NCT>::LDS_BYTES); pn_attr(&k_pn_proj, PG<PJ_NCT>::LDS_BYTES);
  pn_attr(&k_pn_bwd_a, PG<BA_NCT>::LDS_BYTES); pn_attr(&k_pn_bwd_b, PG<BB_NCT>::LDS_BYTES);
  done = true;
}
int pn_blocks(int ncols, int per_wg, int N) { return ((((ncols + per_wg - 1) / per_wg) + 7) & ~7) * ((N + 63) / 64); }
int pn_gates_blocks(int H, int N) { return pn_blocks(H, GT_NCELL, N); }
int pn_proj_blocks(int P, int N) { return pn_blocks(P, 16, N); }
int pn_bwd_a_blocks(int H, int N) { return pn_blocks(H, 16, N); }

void launch_pn_gates(const FwdGateJobs& jobs, int total_blocks, hipStream_t s) {
  pn_init();
  hipLaunchKernelGGL(k_pn_gates, dim3(total_blocks), dim3(PG<GT_NCT>::NT), PG<GT_NCT>::LDS_BYTES, s, jobs);
}
void launch_pn_proj(const FwdProjJobs& jobs, int total_blocks, hipStream_t s) {
  pn_init();
  hipLaunchKernelGGL(k_pn_proj, dim3(total_blocks), dim3(PG<PJ_NCT>::NT), PG<PJ_NCT>::LDS_BYTES, s, jobs);
}
void launch_pn_bwd_a(const BwdAJobs& jobs, int total_blocks, hipStream_t s) {
  pn_init();
  hipLaunchKernelGGL(k_pn_bwd_a, dim3(total_blocks), dim3(PG<BA_NCT>::NT), PG<BA_NCT>::LDS_BYTES, s, jobs);
}
// split-K plan of backward phase B for the panel kernel: KG slices of kpg 96-float chunks, 48-column groups, 64-row groups
size_t pn_bwd_b_plan(BwdBJobs& jobs, float* ws_base) {
  size_t off = 0;
  int bp = 0, br = 0;
  for (int i = 0; i < jobs.n; ++i) {
    BwdBJob& b = jobs.j[i];
    const int nch = (b.H4 + PN_KCF - 1) / PN_KCF, ncols = b.n_end - b.n_begin;
    int KG = std::max(1, (nch + 4) / 5);               // about 5 chunks (480 floats of K) per workgroup
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
