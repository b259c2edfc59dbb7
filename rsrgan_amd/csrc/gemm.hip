// gemm.hip -- time-batched fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak on MI355X), hand-written.
// Used for everything that is NOT on the serial recurrence: the input/output fully_connected layers
// (models/lstm.py:82-87,121-124), the x-part of every LSTM kernel batched over all T*B frames, all weight/data gradients
// batched over time, the frame-level DNN stacks (models/dnn.py, discriminator_dnn.py) and -- through the row maps -- the
// strided 1-D convolutions of the SEGAN-style networks (models/generator.py, discriminator.py: a downconv is the GEMM of an
// overlapping-window VIEW of the channels-last activation with the [kwidth*Cin][Cout] filter).
//
// k_gemm (round 3; replaces the 128x128x16 single-buffer kernel and the hipBLASLt route of round 2):
//   * block = 4 waves, tile BM x BN = 128x128 (2x2 waves of 64x64), 96x128 (1x4 waves of 96x32) or 128x96 (4x1 waves of 32x96):
//     the host takes the shape that wastes the fewest MFMAs on padding (M = 560 = 5.83 x 96; N = 280 = 2.92 x 96).
//   * k-tile 32 deep, LDS double-buffered, ONE barrier per k-tile.  The LDS image is k-major (S[k][x]); the MFMA fragment of
//     lane l (row/col l&31, k = l>>5) is 32 consecutive floats per half-wave = conflict-free ds_read_b32, software-pipelined
//     one k-pair ahead of the MFMAs.
//       - x-contiguous operands ([K][X] in memory: weights [in][out], the stashes of the weight gradients) ARE that image:
//         they go global -> LDS by DMA (global_load_lds, 1 KB per wave-instruction, no VGPRs).  Chunks outside the operand
//         (K tail, column padding) are redirected to a 16-byte zero chunk: no predication anywhere.
//       - k-contiguous operands ([X][K]: activations of a forward layer) are loaded as float4 with clamped addresses at the top
//         of the k-tile, held across the MFMA phase and transposed into the image afterwards (row stride BX+1: conflict-free
//         ds_write_b32); no load sits under a branch.
//   * stream-K work split instead of a tile grid: W = 512 workers (two per CU).  Whole rounds of tiles are data-parallel; the
//     last W..2W tiles are cut into W equal runs of (tile, k-tile) units, so every CU gets the same number of MFMAs whatever
//     the tile count (120 tiles on 256 CUs is the weight-gradient case).  A worker writes a tile it owns completely straight
//     through the epilogue; pieces go to a work space in the accumulator layout (coalesced 16-byte stores) and k_gemm_fixup
//     sums the pieces of a tile in k order (fixed order: deterministic, no float atomics) and runs the same epilogue.
#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace rsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int GK = 32;                       // k-tile depth
#ifndef RSR_GEMM_MINW
#define RSR_GEMM_MINW 2                      // waves per SIMD the register budget leaves room for (one block of 6 waves per CU)
#endif
#ifndef RSR_GEMM_ABL                         // tools/ubench/gemm_bench.hip builds timing-only variants with parts of the k-loop switched
#define RSR_GEMM_ABL 0                       // off (1: no DMA, 2: no fragment reads after the first stage, 4: no barrier); product: 0
#endif
constexpr int BK = 16;                       // k-tile of the narrow kernel below
__device__ __attribute__((aligned(16))) float g_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};
#if RSR_GEMM_ABL & 8
__device__ unsigned long long g_gemm_clk[4];      // shader clocks / 100 MHz ticks of worker 0 (micro-benchmark: the clock the chip held)
#endif

struct GemmArgs {
  const float* A; const float* A2; const float* B; const float* bias; float* C; float* ws;
  int lda, lda2, M1, ldb, ldc, M, N, K;
  int act, accumulate; float alpha;
  int tiles_m, tiles_n, NT, NK, W, n_dp;     // tile grid, tiles, k-tiles, workers, tiles of the data-parallel rounds
  int Uq, Ur;                                // units of the stream-K region = Uq * W + Ur (all unit arithmetic is 32-bit)
  GemmRowMap ma;                             // row map of A (kernels.h), used by the MAP instantiations only
};
// a BATCH of nb same-shaped products in one launch (round 5: the three layers' weight gradients [x | m]^T dZ): tile column tn
// belongs to problem tn / tn1, whose operands replace A / A2 / B / C -- one stream-K unit space over all problems' tiles.  A second
// kernel argument of the batched instantiation only (NoBatch elsewhere: the other kernels' argument block and code stay as they were --
// with the table inside GemmArgs every k_gemm lost ~5-10 %: d(h0) 141 -> 156 us, the SEGAN step 19.7 -> 20.8 ms)
struct GemmBatch { int nb, tn1; const float* Ab[GEMM_MAXB]; const float* A2b[GEMM_MAXB]; const float* Bb[GEMM_MAXB]; float* Cb[GEMM_MAXB]; };
struct NoBatch { int nb; };
// first unit of worker w: floor(w * U / W)
__device__ __forceinline__ int worker_lo(const GemmArgs& g, int w) { return w * g.Uq + (w * g.Ur) / g.W; }
// tile index -> (tile row, tile column): groups of 4 tile rows, column-major inside a group, so that the ~32 tiles an XCD works on
// at a time (consecutive workers share an XCD) form a 4 x 8 block: 12 operand panels through that XCD's L2 instead of 33
__device__ __forceinline__ void tile_rc(const GemmArgs& g, int t, int& tm, int& tn) {
  const int per = 4 * g.tiles_n, grp = t / per, first = grp * 4, rows = min(4, g.tiles_m - first), r = t - grp * per;
  tn = r / rows; tm = first + (r - tn * rows);
}

// address (float offset) of row `r` of a mapped operand: row r is sample r / rows_per, position r % rows_per
__device__ __forceinline__ long long map_row(const GemmRowMap& m, int r) {
  const int q = r / m.rows_per, p = r - q * m.rows_per;
  return (long long)q * m.outer + (long long)p * m.inner;
}

// One operand tile (BX rows or columns x GK) of a k-tile, global -> LDS by DMA (global_load_lds, 16 bytes per lane, 1 KB per
// wave-instruction), issued by the block's NL loader waves: loader `lw` owns the wave-instructions j = lw + NL u (image chunks
// 64 j .. 64 j + 63).  No staging registers, no ds_write, no predication: a chunk outside the operand (K tail, column padding, no
// successor tile) is fetched from a 16-byte zero chunk instead.
//  !KC (x contiguous in memory, element(x,k) = P[k*ld + x]): the image is k-major S[k][x] = the memory layout, row by row.
//  KC  (k contiguous, element(x,k) = P[x*ld + k]): the image is S[x][32 k] with the eight 16-byte chunks of a row stored at
//       position kq ^ ((x >> 1) & 7) (the DMA writes lane-linear, so the permutation sits on the SOURCE side: the lane that fills
//       position pos of row x fetches k-chunk pos ^ ((x >> 1) & 7)); the MFMA fragment is then one 8-byte read per k-quad (2-way
//       bank conflict, irrelevant at the fp32 MFMA rate) instead of a 32-way conflicted column walk.
#ifndef RSR_GEMM_NL
#define RSR_GEMM_NL 2
#endif
constexpr int NL = RSR_GEMM_NL;                                    // loader waves per block
// LDS ring: k-tiles resident (one being multiplied, the others landing): as many as fit beside ~16 KB of slack, at most 4
// (measured at 4096^3, 128 x 128 tiles: 105 / 115 / 117 TFLOP/s with 2 / 3 / 4)
constexpr int ring_depth(int bm, int bn) { return (144 * 1024) / ((bm + bn) * GK * 4) >= 4 ? 4 : (144 * 1024) / ((bm + bn) * GK * 4); }
template <bool KC, int BX, bool MAP = false, int NLW = NL>
struct Stage {
  static constexpr int NI = BX * GK / 256 / NLW;         // wave-instructions per loading wave and k-tile (NLW waves share the tile)
  static constexpr int FLOATS = GK * BX;
  const float* p[NI];                                    // this lane's source of instruction u at the current k-tile
  int mp[MAP && !KC ? NI : 1], mq[MAP && !KC ? NI : 1];  // mapped k-major operand: position / sample of the chunk's k row

  static __device__ __forceinline__ int kc_kq(int lane, int lw) { return (lane & 7) ^ ((4 * lw + (lane >> 4)) & 7); }   // same for every u
  __device__ __forceinline__ void init(const float* P, int ld, const float* P2, int ld2, int X1, int x0, int X, int k_first,
                                       int lane, int lw, const GemmRowMap& map) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int c = 64 * (lw + NLW * u) + lane;
      if (KC) {
        const int row = min(x0 + (c >> 3), X - 1);
        p[u] = P + (MAP ? map_row(map, row) : (long long)row * ld) + k_first + 4 * kc_kq(lane, lw);
      } else {
        const int kr = c / (BX / 4), x = x0 + (c - kr * (BX / 4)) * 4;
        const int xs = x < ((X + 3) & ~3) ? x : 0;
        if (MAP) {                                       // row index = k: (sample, position) kept incrementally
          const int kk = k_first + kr;
          mq[u] = kk / map.rows_per; mp[u] = kk - mq[u] * map.rows_per;
          p[u] = P + xs;
        } else if (P2 && xs >= X1) p[u] = P2 + (long long)(k_first + kr) * ld2 + (xs - X1);
        else p[u] = P + (long long)(k_first + kr) * ld + xs;
      }
    }
  }
  // all of this loader's instructions of the k-tile starting at k0 (K = reduction length; `on` false: a k-tile past the end of
  // the run, zeros -- the instruction count per k-tile is fixed, the loaders wait with COUNTED vmcnt); `dst` = this operand's image
  __device__ __forceinline__ void issue(int k0, int K, bool on, float* dst, int lane, int lw, int ld, int ld2, int X1, int x0, int X,
                                        const float* P2, const GemmRowMap& map) {
#pragma unroll
    for (int u = 0; u < NI; ++u) issue_one(u, k0, K, on, dst, lane, lw, ld, ld2, X1, x0, X, P2, map);
  }
  // instruction u alone (k_gemm_s spreads them over the MFMA stream of the k-tile before)
  __device__ __forceinline__ void issue_one(int u, int k0, int K, bool on, float* dst, int lane, int lw, int ld, int ld2, int X1, int x0, int X,
                                            const float* P2, const GemmRowMap& map) {
    {
      const int j = lw + NLW * u, c = 64 * j + lane;
      bool v;
      const float* src;
      if (KC) {
        v = on && k0 + 4 * kc_kq(lane, lw) < ((K + 3) & ~3);    // (k..k+3 < ld: zero padding of the row)
        src = p[u];
        p[u] += GK;
      } else {
        const int kr = c / (BX / 4), x = x0 + (c - kr * (BX / 4)) * 4;
        v = on && x < ((X + 3) & ~3) && (k0 + kr < K);
        if (MAP) {
          src = p[u] + (long long)mq[u] * map.outer + (long long)mp[u] * map.inner;
          mp[u] += GK;
          const int cq = mp[u] / map.rows_per;
          mp[u] -= cq * map.rows_per; mq[u] += cq;
        } else {
          src = p[u];
          p[u] += GK * ((P2 && x >= X1) ? ld2 : ld);
        }
      }
      src = v ? src : g_zero_chunk;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + 256 * j), 16, 0, 0);
    }
  }
};

// C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int RT, int CT>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[RT][CT], int m_base, int n_base, int l31, int lh, const GemmArgs& g, float* __restrict__ C) {
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n_base + j * 32 + l31;
    const bool cok = col < g.N;
    const int cc = cok ? col : g.N - 1;
    const float bv = g.bias ? g.bias[cc] : 0.f;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
      if (g.act == 1) {                                  // utils/ops.py:120-121 tf.maximum(x, alpha*x)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], g.alpha * v[r]);
      } else if (g.act == 2) {                           // tf.nn.relu (models/dnn.py:36)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      const int row0 = m_base + i * 32 + 4 * lh;
      if (g.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(row0 + (r & 3) + 8 * (r >> 2), g.M - 1);
          v[r] += C[(size_t)row * g.ldc + cc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        if (cok && row < g.M) C[(size_t)row * g.ldc + col] = v[r];
      }
    }
  }
}

// Block = 4 MFMA waves + NL loader waves.  An LDS-DMA costs the issuing wave ~60-180 cycles, a 32x32x2 MFMA occupies the matrix pipe
// for 64: with the DMA in the MFMA waves' own instruction stream (round 3 first form, one chunk behind every fourth MFMA) the pipe
// idled behind every DMA -- 112 TFLOP/s at 4096^3 with one wave per SIMD against 133 with the DMA ablated and 143 with DMA and
// fragment reads ablated (tools/ubench/gemm_bench.hip `abl`).  A loader wave stalls on its own; the SIMD issues the MFMA wave next to it.
template <bool AKC, bool BKC, int RT, int CT, int WM, bool MAPA, class BT = NoBatch>
__global__ __launch_bounds__(64 * (4 + NL), RSR_GEMM_MINW) void k_gemm(const GemmArgs g, const BT bt) {
  constexpr bool BATCH = std::is_same<BT, GemmBatch>::value;
  constexpr int WN = 4 / WM, BM = WM * RT * 32, BN = WN * CT * 32;
  typedef Stage<AKC, BM, MAPA> SA;
  typedef Stage<BKC, BN, false> SB;
  constexpr int BUF = SA::FLOATS + SB::FLOATS, NBUF = ring_depth(BM, BN);
  static_assert(NBUF >= 2, "tile too large for the LDS ring");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid - wr * WN;
  // logical worker: consecutive workers on one XCD (block b runs on XCD b % 8), so that the tiles sharing an operand panel
  // share an L2
  const int W = g.W;
  const int bid = blockIdx.x;
  const int w = (W & 7) ? bid : (bid & 7) * (W >> 3) + (bid >> 3);
  const int u_lo = worker_lo(g, w), u_hi = worker_lo(g, w + 1);
#if RSR_GEMM_ABL & 8
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  int u = u_lo;
  int tdp = w;                                            // whole tiles of the data-parallel rounds: w, w + W, ... < n_dp
  while (tdp < g.n_dp || u < u_hi) {
    int t, i0, i1;
    if (tdp < g.n_dp) { t = tdp; tdp += W; i0 = 0; i1 = g.NK; }
    else {
      const int ts = u / g.NK;
      t = g.n_dp + ts; i0 = u - ts * g.NK;
      i1 = min(g.NK, i0 + (u_hi - u));
    }
    int tm, tn;
    tile_rc(g, t, tm, tn);
    const float* pA = g.A; const float* pA2 = g.A2; const float* pB = g.B; float* pC = g.C;
    if constexpr (BATCH) {
      const GemmBatch& gb = reinterpret_cast<const GemmBatch&>(bt);
      const int b = tn / gb.tn1; tn -= b * gb.tn1; pA = gb.Ab[b]; pA2 = gb.A2b[b]; pB = gb.Bb[b]; pC = gb.Cb[b];
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if (wid >= 4) {
      // ---- loader wave: runs NBUF-1 k-tiles ahead of the MFMA waves through a ring of NBUF LDS buffers.  One barrier per k-tile:
      // behind it the MFMA waves are done with k-tile kt (its buffer is free) and -- by the loader's COUNTED vmcnt just before it
      // -- k-tile kt+1 has landed, while the DMAs of kt+2.. stay in flight across the barrier (operands come through L2 from
      // HBM / MALL: with one k-tile = 1.7 us of prefetch distance the MFMA waves waited on every tile)
      const int lw = wid - 4;
      SA sa; SB sb;
      sa.init(pA, g.lda, pA2, g.lda2, g.M1, m0, g.M, i0 * GK, lane, lw, g.ma);
      sb.init(pB, g.ldb, nullptr, 0, 0, n0, g.N, i0 * GK, lane, lw, g.ma);
      constexpr int NIT = SA::NI + SB::NI;                 // DMA instructions per k-tile of this wave
      static_assert((NBUF - 2) * NIT < 64, "vmcnt is a 6-bit counter");
#pragma unroll
      for (int d = 0; d < NBUF - 1; ++d) {
        sa.issue((i0 + d) * GK, g.K, i0 + d < i1, smem + d * BUF, lane, lw, g.lda, g.lda2, g.M1, m0, g.M, pA2, g.ma);
        sb.issue((i0 + d) * GK, g.K, i0 + d < i1, smem + d * BUF + SA::FLOATS, lane, lw, g.ldb, 0, 0, n0, g.N, nullptr, g.ma);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NIT) : "memory");
      __builtin_amdgcn_s_barrier();
      int slot = NBUF - 1;                                 // ring slot of k-tile kt + NBUF - 1
      for (int kt = i0; kt < i1; ++kt) {
        float* bn = smem + slot * BUF;
        const int kn = kt + NBUF - 1;
        if (!(RSR_GEMM_ABL & 1)) {
          sa.issue(kn * GK, g.K, kn < i1, bn, lane, lw, g.lda, g.lda2, g.M1, m0, g.M, pA2, g.ma);
          sb.issue(kn * GK, g.K, kn < i1, bn + SA::FLOATS, lane, lw, g.ldb, 0, 0, n0, g.N, nullptr, g.ma);
        }
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NIT) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (zero fills of the slots past the run: done before the ring is reused)
    } else {
      // ---- MFMA wave
      f32x16 acc[RT][CT];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      __builtin_amdgcn_s_barrier();
      // fragment addresses.  PAIRED (some operand k-contiguous): MFMA step 2s+e of the k-tile takes k = 4s + 2*(lane>>5) + e, so a
      // k-contiguous operand reads its two values of a k-quad as one 8-byte word; otherwise step 2s+e takes k = 4s + 2e + (lane>>5).
      constexpr bool PAIRED = AKC || BKC;
      const int ra = wr * RT * 32 + l31, rb = wc * CT * 32 + l31;          // row / column of tile 0 inside the block tile
      int slot = 0;
      for (int kt = i0; kt < i1; ++kt) {
        const float* bc = smem + slot * BUF;
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        // The k-tile is ONE pinned instruction stream: NM MFMAs in k order; the fragments of stage s+1 (4 k = 2 MFMA steps) are
        // read one per MFMA gap while stage s multiplies.  Left alone hipcc sinks every read to just before its MFMAs and reuses
        // one register set (read -> wait -> MFMAs -> read ...: a bubble per k-pair on a wave that is alone on its SIMD); read as
        // a cluster at the head of a stage they cost ~450 cycles per k-tile of 4096.
        constexpr int NS = GK / 4;
        constexpr int RA = AKC ? RT : 2 * RT, RB = BKC ? CT : 2 * CT, NR = RA + RB;      // fragment reads per stage
        constexpr int NMS = 2 * RT * CT, RPG = (NR + NMS - 1) / NMS;                     // MFMAs per stage, reads per gap
        float fa[2][2][RT], fb[2][2][CT];                                    // [stage parity][step in stage][tile]
        auto read_one = [&](int st, int par, int r) {
          if (r < RA) {
            if (AKC) {
              const int x = ra + r * 32;
              const float* v = bc + x * 32 + ((st ^ ((x >> 1) & 7)) << 2) + 2 * lh;      // (two float reads: through a float2
              fa[par][0][r] = v[0]; fa[par][1][r] = v[1];                               //  pointer hipcc drains the DMA queue first)
            } else {
              const int e = r / RT, i = r - e * RT;
              fa[par][e][i] = bc[(4 * st + (PAIRED ? 2 * lh + e : 2 * e + lh)) * BM + ra + i * 32];
            }
          } else {
            const float* bb = bc + SA::FLOATS;
            const int q = r - RA;
            if (BKC) {
              const int x = rb + q * 32;
              const float* v = bb + x * 32 + ((st ^ ((x >> 1) & 7)) << 2) + 2 * lh;
              fb[par][0][q] = v[0]; fb[par][1][q] = v[1];
            } else {
              const int e = q / CT, j = q - e * CT;
              fb[par][e][j] = bb[(4 * st + (PAIRED ? 2 * lh + e : 2 * e + lh)) * BN + rb + j * 32];
            }
          }
        };
#pragma unroll
        for (int r = 0; r < NR; ++r) read_one(0, 0, r);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
              for (int j = 0; j < CT; ++j) {
                const int m = (e * RT + i) * CT + j;                         // (compile-time after unrolling)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(RSR_GEMM_ABL & 2) ? 0 : (st & 1)][e][i], fb[(RSR_GEMM_ABL & 2) ? 0 : (st & 1)][e][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < NS && !(RSR_GEMM_ABL & 2)) {
#pragma unroll
                  for (int k = 0; k < RPG; ++k)
                    if (m * RPG + k < NR) read_one(st + 1, (st + 1) & 1, m * RPG + k);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
        }
        __builtin_amdgcn_s_barrier();        // (every fragment read above has been waited for by the MFMA that takes it)
      }

      if (i0 == 0 && i1 == g.NK) {
        gemm_epilogue<RT, CT>(acc, m0 + wr * RT * 32, n0 + wc * CT * 32, l31, lh, g, pC);
      } else {
        // a piece of a tile: raw accumulators, lane-native order ([tile][quad][thread] float4: coalesced)
        const int slot = 2 * w + (u == u_lo ? 0 : 1);
        float4* q = reinterpret_cast<float4*>(g.ws) + (size_t)slot * (RT * CT * 4 * 256) + tid;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
              q[((i * CT + j) * 4 + r4) * 256] = make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]);
      }
    }
    if (t >= g.n_dp) u += i1 - i0;
  }
#if RSR_GEMM_ABL & 8
  if (bid == 0 && tid == 0) { g_gemm_clk[0] = __builtin_amdgcn_s_memtime() - clk0; g_gemm_clk[1] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// k_gemm_s: the same product on 256-wide tiles with FOUR waves that load for themselves (no loader waves): one wave per SIMD owns the
// whole 512-register file (accumulators of a 128 x 128 or 64 x 128 sub-tile in the AGPR half), every wave issues its quarter of the
// next k-tile's DMA inside its own MFMA stream -- one 1 KB instruction per DPM MFMAs -- into the second of two LDS buffers; one
// vmcnt(0) + barrier per k-tile.  k_gemm at 128 x 128 is ingest-bound (8 B/clk/CU against ~6.8); a 256 x 256 x 32 k-tile needs
// 64 KB per 16384 MFMA cycles = 4 B/clk, 128 x 256 needs 6 (section 6-R3: the design point of the vendor library's kernels).
// ------------------------------------------------------------------------------------------------------------------------
template <bool AKC, bool BKC, int RT, int CT, int WM, bool MAPA, class BT = NoBatch>
__global__ __launch_bounds__(256, 1) void k_gemm_s(const GemmArgs g, const BT bt) {
  constexpr bool BATCH = std::is_same<BT, GemmBatch>::value;
  constexpr int WN = 4 / WM, BM = WM * RT * 32, BN = WN * CT * 32;
  typedef Stage<AKC, BM, MAPA, 4> SA;
  typedef Stage<BKC, BN, false, 4> SB;
  constexpr int BUF = SA::FLOATS + SB::FLOATS;
  static_assert(2 * BUF * 4 <= 160 * 1024, "two k-tiles must fit the LDS");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid - wr * WN;
  const int W = g.W;
  const int bid = blockIdx.x;
  const int w = (W & 7) ? bid : (bid & 7) * (W >> 3) + (bid >> 3);
  const int u_lo = worker_lo(g, w), u_hi = worker_lo(g, w + 1);
  int u = u_lo;
  int tdp = w;
  while (tdp < g.n_dp || u < u_hi) {
    int t, i0, i1;
    if (tdp < g.n_dp) { t = tdp; tdp += W; i0 = 0; i1 = g.NK; }
    else {
      const int ts = u / g.NK;
      t = g.n_dp + ts; i0 = u - ts * g.NK;
      i1 = min(g.NK, i0 + (u_hi - u));
    }
    int tm, tn;
    tile_rc(g, t, tm, tn);
    const float* pA = g.A; const float* pA2 = g.A2; const float* pB = g.B; float* pC = g.C;
    if constexpr (BATCH) {
      const GemmBatch& gb = reinterpret_cast<const GemmBatch&>(bt);
      const int b = tn / gb.tn1; tn -= b * gb.tn1; pA = gb.Ab[b]; pA2 = gb.A2b[b]; pB = gb.Bb[b]; pC = gb.Cb[b];
    }
    const int m0 = tm * BM, n0 = tn * BN;
    SA sa; SB sb;
    sa.init(pA, g.lda, pA2, g.lda2, g.M1, m0, g.M, i0 * GK, lane, wid, g.ma);
    sb.init(pB, g.ldb, nullptr, 0, 0, n0, g.N, i0 * GK, lane, wid, g.ma);
    constexpr int NIT = SA::NI + SB::NI;                   // DMA instructions per wave and k-tile
    __builtin_amdgcn_s_barrier();                          // the previous tile's last k-tile has been read by every wave
    sa.issue(i0 * GK, g.K, true, smem, lane, wid, g.lda, g.lda2, g.M1, m0, g.M, pA2, g.ma);
    sb.issue(i0 * GK, g.K, true, smem + SA::FLOATS, lane, wid, g.ldb, 0, 0, n0, g.N, nullptr, g.ma);
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr bool PAIRED = AKC || BKC;
    const int ra = wr * RT * 32 + l31, rb = wc * CT * 32 + l31;
    // swizzle term of the k-contiguous images: ((x >> 1) & 7) with x = ra + 32 r is the same for every sub-tile r -- spelled out, a
    // fragment address is ONE register per stage plus an immediate (r * 4 KB); left to the compiler every (stage, r) pair got its
    // own hoisted register and the 256 x 256 tile spilled
    const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;
    int slot = 0;
    for (int kt = i0; kt < i1; ++kt) {
      const float* bc = smem + slot * BUF;
      float* bnx = smem + (slot ^ 1) * BUF;
      slot ^= 1;
      const bool more = kt + 1 < i1;                       // (uniform) the next k-tile of this run: its DMA rides this k-tile's MFMAs
      constexpr int NS = GK / 4;
      constexpr int RA = AKC ? RT : 2 * RT, RB = BKC ? CT : 2 * CT, NR = RA + RB;
      constexpr int NMS = 2 * RT * CT, RPG = (NR + NMS - 1) / NMS;
      constexpr int NM = NS * NMS;                         // MFMAs per k-tile; DMA instruction d rides MFMA ((2d+1) NM) / (2 NIT)
      static_assert(NM >= NIT, "at most one DMA instruction per MFMA");
      float fa[2][2][RT], fb[2][2][CT];
      auto read_one = [&](int st, int par, int r) {
        if (r < RA) {
          if (AKC) {
            const float* v = bc + ra * 32 + ((st ^ swa) << 2) + 2 * lh + r * 1024;
            fa[par][0][r] = v[0]; fa[par][1][r] = v[1];
          } else {
            const int e = r / RT, i = r - e * RT;
            fa[par][e][i] = bc[(4 * st + (PAIRED ? 2 * lh + e : 2 * e + lh)) * BM + ra + i * 32];
          }
        } else {
          const float* bb = bc + SA::FLOATS;
          const int q = r - RA;
          if (BKC) {
            const float* v = bb + rb * 32 + ((st ^ swb) << 2) + 2 * lh + q * 1024;
            fb[par][0][q] = v[0]; fb[par][1][q] = v[1];
          } else {
            const int e = q / CT, j = q - e * CT;
            fb[par][e][j] = bb[(4 * st + (PAIRED ? 2 * lh + e : 2 * e + lh)) * BN + rb + j * 32];
          }
        }
      };
#pragma unroll
      for (int r = 0; r < NR; ++r) read_one(0, 0, r);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) {
              const int m = (e * RT + i) * CT + j;         // MFMA index inside the stage (compile-time after unrolling)
              const int gm = st * NMS + m;                 // ... inside the k-tile
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1][e][i], fb[st & 1][e][j], acc[i][j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              const int d = (gm * NIT) / NM;
              if (gm == ((2 * d + 1) * NM) / (2 * NIT) && more) {      // behind an MFMA: the matrix pipe has work while the DMA issues
                if (d < SA::NI) sa.issue_one(d, (kt + 1) * GK, g.K, true, bnx, lane, wid, g.lda, g.lda2, g.M1, m0, g.M, pA2, g.ma);
                else sb.issue_one(d - SA::NI, (kt + 1) * GK, g.K, true, bnx + SA::FLOATS, lane, wid, g.ldb, 0, 0, n0, g.N, nullptr, g.ma);
                __builtin_amdgcn_sched_barrier(0);
              }
              if (st + 1 < NS) {
#pragma unroll
                for (int k = 0; k < RPG; ++k)
                  if (m * RPG + k < NR) read_one(st + 1, (st + 1) & 1, m * RPG + k);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of k-tile kt+1 has landed
      __builtin_amdgcn_s_barrier();                        // ... everybody's; and k-tile kt has been read by every wave
    }
    if (i0 == 0 && i1 == g.NK) {
      gemm_epilogue<RT, CT>(acc, m0 + wr * RT * 32, n0 + wc * CT * 32, l31, lh, g, pC);
    } else {
      const int pslot = 2 * w + (u == u_lo ? 0 : 1);
      float4* q = reinterpret_cast<float4*>(g.ws) + (size_t)pslot * (RT * CT * 4 * 256) + tid;
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            q[((i * CT + j) * 4 + r4) * 256] = make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]);
    }
    if (t >= g.n_dp) u += i1 - i0;
  }
}

// tile `ts` of the stream-K region: the workers whose runs cut it, in k order; first run of a worker -> slot 2w, last -> 2w+1
template <int RT, int CT, int WM, class BT = NoBatch>
__global__ __launch_bounds__(256) void k_gemm_fixup(const GemmArgs g, const BT bt) {
  constexpr bool BATCH = std::is_same<BT, GemmBatch>::value;
  constexpr int WN = 4 / WM, BM = WM * RT * 32, BN = WN * CT * 32;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = tid >> 6, wr = wid / WN, wc = wid - wr * WN;
  const int ts = blockIdx.x;
  // the workers holding the tile's first and last unit: largest w with worker_lo(w) <= u
  const int ua = ts * g.NK, ub = ua + g.NK - 1;
  auto owner = [&](int u) {
    int w = g.Uq > 0 ? min(g.W - 1, u / g.Uq) : g.W - 1;
    while (w > 0 && worker_lo(g, w) > u) --w;
    while (w + 1 < g.W && worker_lo(g, w + 1) <= u) ++w;
    return w;
  };
  const int wa = owner(ua), wb = owner(ub);
  if (wa == wb) return;                                   // one worker owned the whole tile and wrote it
  // Large tiles (k_gemm_s: up to 4 x 4 sub-tiles per wave = 64 float4 per thread and piece): one block per ROW of sub-tiles
  // (blockIdx.y) -- RT times the blocks, and a piece row is small enough to keep two pieces in flight (round 5: the 192 x 256
  // tiles of the batched dK launch have 108 tiles for 256 CUs: 83 us as one block per tile)
  constexpr bool SPLITI = RT * CT * 4 > 16;
  constexpr int RTL = SPLITI ? 1 : RT;
  const int isel = SPLITI ? blockIdx.y : 0;
  f32x16 acc[RTL][CT];
#pragma unroll
  for (int i = 0; i < RTL; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // pieces in k order, TWO in flight when the tile is small enough to hold them in registers (a tile of a long-K product is cut into
  // 3-10 pieces: one dependent round trip per piece made this launch 20 us in the SEGAN step); the sums keep their order
  constexpr int NQ4 = RT * CT * 4, NQL = RTL * CT * 4;
  constexpr bool PAIR = NQL <= 16;
  auto slot_of = [&](int w) { return 2 * w + (worker_lo(g, w) / g.NK == ts ? 0 : 1); };
  for (int w = wa; w <= wb; w += PAIR ? 2 : 1) {
    const float4* q0 = reinterpret_cast<const float4*>(g.ws) + (size_t)slot_of(w) * (NQ4 * 256) + (size_t)isel * (CT * 4 * 256) + tid;
    const bool two = PAIR && w + 1 <= wb;
    const float4* q1 = reinterpret_cast<const float4*>(g.ws) + (size_t)slot_of(two ? w + 1 : w) * (NQ4 * 256) + (size_t)isel * (CT * 4 * 256) + tid;
    float4 v0[NQL], v1[PAIR ? NQL : 1];
#pragma unroll
    for (int x = 0; x < NQL; ++x) v0[x] = q0[x * 256];
    if (PAIR) {
#pragma unroll
      for (int x = 0; x < NQL; ++x) v1[x] = q1[x * 256];             // (unconditional: the last odd piece is read twice, added once)
    }
#pragma unroll
    for (int i = 0; i < RTL; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 v = v0[(i * CT + j) * 4 + r4];
          acc[i][j][4 * r4] += v.x; acc[i][j][4 * r4 + 1] += v.y; acc[i][j][4 * r4 + 2] += v.z; acc[i][j][4 * r4 + 3] += v.w;
        }
    if (PAIR && two) {
#pragma unroll
      for (int i = 0; i < RTL; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = v1[(i * CT + j) * 4 + r4];
            acc[i][j][4 * r4] += v.x; acc[i][j][4 * r4 + 1] += v.y; acc[i][j][4 * r4 + 2] += v.z; acc[i][j][4 * r4 + 3] += v.w;
          }
    }
  }
  const int t = g.n_dp + ts;
  int tm, tn;
  tile_rc(g, t, tm, tn);
  float* pC = g.C;
  if constexpr (BATCH) { const GemmBatch& gb = reinterpret_cast<const GemmBatch&>(bt); const int b = tn / gb.tn1; tn -= b * gb.tn1; pC = gb.Cb[b]; }
  gemm_epilogue<RTL, CT>(acc, tm * BM + wr * RT * 32 + isel * 32, tn * BN + wc * CT * 32, l31, lh, g, pC);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_gemm16: the 128 x 128 x 16 kernel of rounds 1-2 (register-staged operands, single LDS buffer, 3-4 blocks per CU, deterministic
// split-K + k_splitk_reduce).  It stays the kernel of the products with little work per output tile -- short K (a tile of k_gemm
// pays a ring fill and a 6-wave block's epilogue per tile, with ONE block per CU: 12800 x 1024 x 40 took 125 us against 18 here)
// and small outputs with a long K (many pieces per tile) -- see launch_gemm_mapped for the rule.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, LDT = 132;
constexpr int GBK = 16;               // k-tile of k_gemm16

// Load a 128(x) x 16(k) operand tile into registers (2 float4 per thread).
//  KC  (k contiguous): element(x,k) = P[x*ld + k]  -> thread: x = idx>>2, k4 = (idx&3)*4
//  !KC (x contiguous): element(x,k) = P[k*ld + x]  -> thread: k = idx>>5, x4 = (idx&31)*4
// `P2`/`ld2`/`X1` (x-contiguous operands only): rows x >= X1 of the operand come from a second matrix P2 (column
// x - X1); X1 % 4 == 0 so a float4 never straddles.  This is how dK = [x_t | m_{t-1}]^T dZ reads its two stashes.
constexpr int GNF = BM * GBK / 4 / 256;      // float4 per thread and operand tile
template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int x0, int X, int k0, int K,
                                          int tid, float4 (&r)[GNF], const float* __restrict__ P2 = nullptr, int ld2 = 0,
                                          int X1 = 0) {
#pragma unroll
  for (int u = 0; u < GNF; ++u) {
    const int idx = tid + 256 * u;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      const int x = x0 + idx / (GBK / 4), k = k0 + (idx % (GBK / 4)) * 4;
      if (x < X && k < K) v = *reinterpret_cast<const float4*>(P + (size_t)x * ld + k);   // k..k+3 < ld (zero pad)
    } else {
      const int k = k0 + (idx >> 5), x = x0 + (idx & 31) * 4;
      if (k < K && x < X)                                                                 // x..x+3 < ld (zero pad)
        v = (P2 && x >= X1) ? *reinterpret_cast<const float4*>(P2 + (size_t)k * ld2 + (x - X1))
                            : *reinterpret_cast<const float4*>(P + (size_t)k * ld + x);
    }
    r[u] = v;
  }
}
template <bool KC>
__device__ __forceinline__ void store_tile(float (*S)[LDT], int tid, const float4 (&r)[GNF]) {
#pragma unroll
  for (int u = 0; u < GNF; ++u) {
    const int idx = tid + 256 * u;
    if (KC) {
      const int x = idx / (GBK / 4), k = (idx % (GBK / 4)) * 4;
      S[k + 0][x] = r[u].x; S[k + 1][x] = r[u].y; S[k + 2][x] = r[u].z; S[k + 3][x] = r[u].w;
    } else {
      const int k = idx >> 5, x = (idx & 31) * 4;
      *reinterpret_cast<float4*>(&S[k][x]) = r[u];
    }
  }
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void gemm16_body(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                            float* __restrict__ C, int ldc, int M, int N, int K,
                                            const float* __restrict__ bias, int act, float alpha, int accumulate,
                                            float* __restrict__ ws, int ldw, int kt_per_split,
                                            const float* __restrict__ A2, int lda2, int M1, const int zs) {      // zs: this block's k split
  __shared__ __attribute__((aligned(16))) float As[GBK][LDT];
  __shared__ __attribute__((aligned(16))) float Bs[GBK][LDT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // k-contiguous operands may be read up to their zero-padded width
  const int KA = AKC ? ((K + 3) & ~3) : K;
  const int KB = BKC ? ((K + 3) & ~3) : K;
  const int MA = AKC ? M : ((M + 3) & ~3);
  const int NB = BKC ? N : ((N + 3) & ~3);

  // split-K: blockIdx.z owns k-tiles [kt0, kt1); partial tiles go to ws[z][M][ldw] and are summed in
  // a fixed order by k_splitk_reduce (deterministic, unlike float atomics)
  const int nk_all = (K + GBK - 1) / GBK;
  const int kt0 = zs * kt_per_split;
  const int nk = min(nk_all, kt0 + kt_per_split);
  float4 ra[GNF], rb[GNF];
  load_tile<AKC>(A, lda, m0, MA, kt0 * GBK, KA, tid, ra, A2, lda2, M1);
  load_tile<BKC>(B, ldb, n0, NB, kt0 * GBK, KB, tid, rb);
  for (int kt = kt0; kt < nk; ++kt) {
    store_tile<AKC>(As, tid, ra);
    store_tile<BKC>(Bs, tid, rb);
    __syncthreads();
    if (kt + 1 < nk) {
      load_tile<AKC>(A, lda, m0, MA, (kt + 1) * GBK, KA, tid, ra, A2, lda2, M1);
      load_tile<BKC>(B, ldb, n0, NB, (kt + 1) * GBK, KB, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < GBK / 2; ++kk) {
      const int k = 2 * kk + lh;
      const float a0 = As[k][wr * 64 + l31], a1 = As[k][wr * 64 + 32 + l31];
      const float b0 = Bs[k][wc * 64 + l31], b1 = Bs[k][wc * 64 + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + l31;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= M) continue;
        if (ws) {
          ws[((size_t)zs * M + row) * ldw + col] = acc[i][j][r];
          continue;
        }
        float v = acc[i][j][r] + bv;
        if (act == 1) v = fmaxf(v, alpha * v);            // utils/ops.py:120-121 tf.maximum(x, alpha*x)
        else if (act == 2) v = fmaxf(v, 0.f);             // tf.nn.relu (models/dnn.py:36)
        float* c = C + (size_t)row * ldc + col;
        if (accumulate) v += *c;
        *c = v;
      }
    }
}
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void k_gemm16(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, int M, int N, int K,
                                              const float* __restrict__ bias, int act, float alpha, int accumulate,
                                              float* __restrict__ ws, int ldw, int kt_per_split,
                                              const float* __restrict__ A2, int lda2, int M1) {
  gemm16_body<AKC, BKC>(A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, accumulate, ws, ldw, kt_per_split, A2, lda2, M1, blockIdx.z);
}
// Up to GEMM16_MAXB products of ONE shape in one launch (the weight gradients of a stack's layers: [x | m]^T dZ, h^T dm -- short
// launches that each left most of the chip idle and paid a launch + a reduce launch of their own): blockIdx.z = problem * nsplit + split.
__global__ __launch_bounds__(256) void k_gemm16_b(const Gemm16Batch bt, int lda, int ldb, int ldc, int M, int N, int K, int accumulate,
                                                  float* __restrict__ ws, int ldw, int kt_per_split, int nsplit, int lda2, int M1) {
  const int p = blockIdx.z / nsplit, zs = blockIdx.z - p * nsplit;
  gemm16_body<false, false>(bt.A[p], lda, bt.B[p], ldb, bt.C[p], ldc, M, N, K, nullptr, 0, 0.f, accumulate,
                            ws ? ws + (size_t)p * nsplit * M * ldw : nullptr, ldw, kt_per_split, bt.A2[p], lda2, M1, zs);
}
__global__ __launch_bounds__(256) void k_splitk_reduce_b(const float* __restrict__ ws, int ldw, int splits, const Gemm16Batch bt, int ldc,
                                                         int M, int N, int accumulate) {
  const int p = blockIdx.y;
  const float* w = ws + (size_t)p * splits * M * ldw;
  float* C = bt.C[p];
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / N), col = (int)(i % N);
    float v = 0.f;
#pragma unroll 8
    for (int z = 0; z < splits; ++z) v += w[((size_t)z * M + row) * ldw + col];
    float* c = C + (size_t)row * ldc + col;
    if (accumulate) v += *c;
    *c = v;
  }
}

__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ ws, int ldw, int splits, float* __restrict__ C,
                                                       int ldc, int M, int N, const float* __restrict__ bias, int act,
                                                       float alpha, int accumulate) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / N), col = (int)(i % N);
    float v = 0.f;
#pragma unroll 8
    for (int z = 0; z < splits; ++z) v += ws[((size_t)z * M + row) * ldw + col];
    if (bias) v += bias[col];
    if (act == 1) v = fmaxf(v, alpha * v);
    else if (act == 2) v = fmaxf(v, 0.f);
    float* c = C + (size_t)row * ldc + col;
    if (accumulate) v += *c;
    *c = v;
  }
}


// Narrow-N variant for outputs with N <= 32 columns (R-CED: a conv2d has 12..32 filters; 128-wide tiles would spend 75-90 %
// of the MFMA work on padding): 256 x 32 x 16 block tile, 4 waves each 64 rows x 32 columns = 2 MFMA tiles.  B is [K][N]
// (n contiguous) only.  Same k-major LDS image, register prefetch and deterministic split-K as k_gemm.
constexpr int NBM = 256, NBN = 32, NLDA = 260, NLDB = 36;
template <bool AKC>
__global__ __launch_bounds__(256) void k_gemm_n32(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                  float* __restrict__ C, int ldc, int M, int N, int K,
                                                  const float* __restrict__ bias, int act, float alpha, int accumulate,
                                                  float* __restrict__ ws, int ldw, int kt_per_split) {
  __shared__ __attribute__((aligned(16))) float As[BK][NLDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK][NLDB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.y * NBM;
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int KA = AKC ? ((K + 3) & ~3) : K;
  const int MA = AKC ? M : ((M + 3) & ~3);
  const int NB = (N + 3) & ~3;
  const int nk_all = (K + BK - 1) / BK;
  const int kt0 = blockIdx.z * kt_per_split;
  const int nk = min(nk_all, kt0 + kt_per_split);
  float4 ra[4], rb;
  auto load = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (AKC) {
        const int x = m0 + (idx >> 2), k = k0 + (idx & 3) * 4;
        if (x < MA && k < KA) v = *reinterpret_cast<const float4*>(A + (size_t)x * lda + k);
      } else {
        const int k = k0 + (idx >> 6), x = m0 + (idx & 63) * 4;
        if (k < K && x < MA) v = *reinterpret_cast<const float4*>(A + (size_t)k * lda + x);
      }
      ra[u] = v;
    }
    rb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 128) {
      const int k = k0 + (tid >> 3), x = (tid & 7) * 4;
      if (k < K && x < NB) rb = *reinterpret_cast<const float4*>(B + (size_t)k * ldb + x);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      if (AKC) {
        const int x = idx >> 2, k = (idx & 3) * 4;
        As[k + 0][x] = ra[u].x; As[k + 1][x] = ra[u].y; As[k + 2][x] = ra[u].z; As[k + 3][x] = ra[u].w;
      } else {
        const int k = idx >> 6, x = (idx & 63) * 4;
        *reinterpret_cast<float4*>(&As[k][x]) = ra[u];
      }
    }
    if (tid < 128) *reinterpret_cast<float4*>(&Bs[tid >> 3][(tid & 7) * 4]) = rb;
  };
  load(kt0);
  for (int kt = kt0; kt < nk; ++kt) {
    store();
    __syncthreads();
    if (kt + 1 < nk) load(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = 2 * kk + lh;
      const float a0 = As[k][w * 64 + l31], a1 = As[k][w * 64 + 32 + l31];
      const float b0 = Bs[k][l31];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = l31;
  if (col >= N) return;
  const float bv = bias ? bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + w * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= M) continue;
      if (ws) { ws[((size_t)blockIdx.z * M + row) * ldw + col] = acc[i][r]; continue; }
      float v = acc[i][r] + bv;
      if (act == 1) v = fmaxf(v, alpha * v);
      else if (act == 2) v = fmaxf(v, 0.f);
      float* c = C + (size_t)row * ldc + col;
      if (accumulate) v += *c;
      *c = v;
    }
}


// Split-K factor for an under-filled output grid (weight gradients: few tiles, K = T*B).  Model: the busiest CU runs
// ceil(tiles*s/256) workgroups of ceil(nk/s) k-tiles (~1.2 us each; 25 % slower when fewer than two workgroups per CU
// hide each other's latency), then the reduce streams s partial images at ~4 TB/s.
static int pick_splits(int tiles, int nk, size_t out_bytes, int max_splits, double tile_us = 1.2, int min_per = 4) {
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const int per = (nk + sp - 1) / sp;
    if (sp > 1 && per < min_per) break;
    const int wgs = tiles * sp;
    double cost = (double)((wgs + 255) / 256) * per * tile_us * (wgs < 512 ? 1.25 : 1.0);
    if (sp > 1) cost += 2.8 + (double)(sp + 1) * out_bytes / 4.0e6;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
  }
  return best;
}


thread_local int g_gemm_workers = 256;     // (per host thread: Model::g_backward narrows it around the launches that share the chip with the next D(real))
                                            // worker slots of a launch = one block (4 MFMA + 2 loader waves, ring of LDS buffers) per CU
namespace {

struct Plan { int W, n_dp, whole; double cost; };
// cost of a fix-up launch in the plan: fixed part (us) and bytes per us.  (The launches measure ~20 us in the SEGAN step, yet pricing
// them at 10 - 40 us made that step slower, 25.1 -> 26.0 ms: cutting tiles is right even then.)
constexpr double g_fix_us = 4.0, g_fix_bw = 3.0e6;
// Time model of one configuration (us).  The MFMA waves run at ~0.92 of the matrix rate when fed; a CU ingests operands at
// ~14 GB/s (L2 -> LDS DMA, measured: 117 TFLOP/s at 128 x 128 tiles = 0.031 B/FLOP), so small tiles are ingest-bound; a cut
// tile costs its pieces a write and a read plus the fix-up launch; whole-tile mode costs the idle CUs of the last round.
Plan plan_cfg(int M, int N, int K, int bm, int bn, int workers, float* ws, size_t ws_floats) {
  const int tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn, NT = tm * tn, NK = (K + GK - 1) / GK;
  const double flops = 2.0 * tm * bm * (double)tn * bn * NK * GK;
  const double t_mfma = flops / (157.3e6 * 0.92), t_in = flops * (2.0 / bm + 2.0 / bn) / 3.7e6;
  const double t_full = std::max(t_mfma, t_in);            // us with all 256 CUs busy
  const long long units = (long long)NT * NK;
  int W = workers;
  if (units < 8LL * W) W = (int)std::max<long long>(8, (units / 8) & ~7LL);
  if (units < 64) W = 1;
  Plan whole{std::min(W, NT), NT, 1, 0.0};
  if (whole.W >= 8) whole.W &= ~7;
  whole.cost = t_full * 256.0 / NT * ((NT + whole.W - 1) / whole.W);     // a worker = a CU: its tiles in sequence
  Plan sk{W, 0, 0, 1e30};
  while (ws && sk.W > 8 && 2 * (size_t)sk.W * bm * bn > ws_floats) sk.W -= 8;
  if (ws && 2 * (size_t)sk.W * bm * bn <= ws_floats && NT < 8 * sk.W && sk.W > 1) {
    sk.n_dp = NT >= 2 * sk.W ? (NT / sk.W - 1) * sk.W : 0;
    const long long U = (long long)(NT - sk.n_dp) * NK;
    const bool cut = !(U % sk.W == 0 && (U / sk.W) % NK == 0);
    const double pieces = cut ? (double)(sk.W + (NT - sk.n_dp)) * bm * bn * 4.0 : 0.0;       // bytes, upper estimate
    sk.cost = t_full * 256.0 / std::min(sk.W, 256) + (cut ? g_fix_us + 2.0 * pieces / g_fix_bw : 0.0);
  }
  return sk.cost < whole.cost ? sk : whole;
}

template <bool AKC, bool BKC, int RT, int CT, int WM, bool MAPA>
void launch_cfg(GemmArgs& g, const Plan& pl, hipStream_t s, float* ws, GemmBatch* bt = nullptr) {
  constexpr int WN = 4 / WM, BM = WM * RT * 32, BN = WN * CT * 32;
  constexpr size_t lds = (size_t)ring_depth(BM, BN) * (BM + BN) * GK * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<AKC, BKC, RT, CT, WM, MAPA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = ((g.N + BN - 1) / BN) * (bt ? bt->nb : 1);     // (a batch: every problem's tile columns side by side)
  g.NT = g.tiles_m * g.tiles_n;
  g.NK = (g.K + GK - 1) / GK;
  if ((long long)g.NT * g.NK >= (1LL << 30)) { fprintf(stderr, "rsrgan: GEMM beyond the 32-bit (tile, k-tile) unit arithmetic\n"); abort(); }
  g.ws = ws; g.W = pl.W; g.n_dp = pl.n_dp;
  const int U = (g.NT - g.n_dp) * g.NK;
  g.Uq = U / g.W; g.Ur = U % g.W;
  const bool cut = U > 0 && !(U % g.W == 0 && (U / g.W) % g.NK == 0);      // some tile is cut: sum its pieces
  if (bt) {
    if constexpr (!AKC && !BKC && !MAPA && RT == 2 && CT == 2 && WM == 2) {
      static bool attr_b = false;
      if (!attr_b) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<AKC, BKC, RT, CT, WM, MAPA, GemmBatch>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_b = true; }
      bt->tn1 = (g.N + BN - 1) / BN;
      hipLaunchKernelGGL((k_gemm<AKC, BKC, RT, CT, WM, MAPA, GemmBatch>), dim3(g.W), dim3(64 * (4 + NL)), lds, s, g, *bt);
      if (cut) hipLaunchKernelGGL((k_gemm_fixup<RT, CT, WM, GemmBatch>), dim3(g.NT - g.n_dp, RT * CT * 4 > 16 ? RT : 1), dim3(256), 0, s, g, *bt);
    }
    return;
  }
  hipLaunchKernelGGL((k_gemm<AKC, BKC, RT, CT, WM, MAPA>), dim3(g.W), dim3(64 * (4 + NL)), lds, s, g, NoBatch{0});
  if (cut) hipLaunchKernelGGL((k_gemm_fixup<RT, CT, WM>), dim3(g.NT - g.n_dp, RT * CT * 4 > 16 ? RT : 1), dim3(256), 0, s, g, NoBatch{0});
}

template <bool AKC, bool BKC, int RT, int CT, int WM, bool MAPA, class BT = NoBatch>
void launch_cfg_s(GemmArgs& g, const Plan& pl, hipStream_t s, float* ws, BT* bt = nullptr) {
  constexpr int WN = 4 / WM, BM = WM * RT * 32, BN = WN * CT * 32;
  constexpr size_t lds = (size_t)2 * (BM + BN) * GK * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_s<AKC, BKC, RT, CT, WM, MAPA, BT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = ((g.N + BN - 1) / BN) * (bt ? bt->nb : 1);
  g.NT = g.tiles_m * g.tiles_n;
  g.NK = (g.K + GK - 1) / GK;
  if ((long long)g.NT * g.NK >= (1LL << 30)) { fprintf(stderr, "rsrgan: GEMM beyond the 32-bit (tile, k-tile) unit arithmetic\n"); abort(); }
  g.ws = ws; g.W = pl.W; g.n_dp = pl.n_dp;
  const int U = (g.NT - g.n_dp) * g.NK;
  g.Uq = U / g.W; g.Ur = U % g.W;
  BT btv{}; if (bt) { btv = *bt; }
  if constexpr (std::is_same<BT, GemmBatch>::value) btv.tn1 = (g.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_gemm_s<AKC, BKC, RT, CT, WM, MAPA, BT>), dim3(g.W), dim3(256), lds, s, g, btv);
  if (U > 0 && !(U % g.W == 0 && (U / g.W) % g.NK == 0))
    hipLaunchKernelGGL((k_gemm_fixup<RT, CT, WM, BT>), dim3(g.NT - g.n_dp, RT * CT * 4 > 16 ? RT : 1), dim3(256), 0, s, g, btv);
}

template <bool AKC, bool BKC, bool MAPA>
void launch_layout(GemmArgs& g, hipStream_t s, float* ws, size_t ws_floats) {
  // (192 x 128 and 256 x 128 tiles were measured too: +4 % at 4096^3, slower on every shape of the training steps -- fewer, larger
  //  pieces to fix up -- and past the 256-register budget of a 6-wave block: they spill.  Not instantiated.)
  // 256 x 32 / 256 x 64: the window-view products of the first SEGAN layers have 16..64 output channels and ~1e5..1e6 rows
  // 256 x 256 / 128 x 256 / 256 x 128: k_gemm_s (four self-loading waves); RSRGAN_GEMM_SELF=0 leaves them out
  static const int cfgs[][2] = {{128, 128}, {96, 128}, {128, 96}, {256, 64}, {256, 32}, {256, 256}, {128, 256}, {256, 128}};
  static int self = -1;
  if (self < 0) { const char* e = getenv("RSRGAN_GEMM_SELF"); self = e ? atoi(e) : 1; }
  int best = 0;
  Plan bp{};
  for (int c = 0; c < (self ? 8 : 5); ++c) {
    // the 256-wide tiles pay off with at least two full rounds of whole tiles (measured: 4096^3 118-121 -> 131-133 TFLOP/s,
    // 32768 x 1024 x 1024 105-109 -> 116-118; 6400 x 1024 x 1024, 200 tiles of 128 x 256: 87 either way, the frame-level step 2 % slower)
    if (c >= 5 && (long long)((g.M + cfgs[c][0] - 1) / cfgs[c][0]) * ((g.N + cfgs[c][1] - 1) / cfgs[c][1]) < 2LL * g_gemm_workers) continue;
    Plan pl = plan_cfg(g.M, g.N, g.K, cfgs[c][0], cfgs[c][1], g_gemm_workers, ws, ws_floats);
    if (c == 0 || pl.cost < 0.97 * bp.cost) { best = c; bp = pl; }
  }
  switch (best) {
    case 1: launch_cfg<AKC, BKC, 3, 1, 1, MAPA>(g, bp, s, ws); break;
    case 2: launch_cfg<AKC, BKC, 1, 3, 4, MAPA>(g, bp, s, ws); break;
    case 3: launch_cfg<AKC, BKC, 2, 2, 4, MAPA>(g, bp, s, ws); break;
    case 4: launch_cfg<AKC, BKC, 2, 1, 4, MAPA>(g, bp, s, ws); break;
    case 5: launch_cfg_s<AKC, BKC, 4, 4, 2, MAPA>(g, bp, s, ws); break;
    case 6: launch_cfg_s<AKC, BKC, 2, 4, 2, MAPA>(g, bp, s, ws); break;
    case 7: launch_cfg_s<AKC, BKC, 4, 2, 2, MAPA>(g, bp, s, ws); break;
    default: launch_cfg<AKC, BKC, 2, 2, 2, MAPA>(g, bp, s, ws); break;
  }
}
}  // namespace

static void launch_gemm16(const float* A, int lda, const float* A2, int lda2, int M1, bool a_kc, const float* B, int ldb, bool b_kc,
                          float* C, int ldc, int M, int N, int K, const float* bias, int act, float alpha, bool accumulate,
                          hipStream_t s, float* ws, size_t ws_floats) {
  const int gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
  const int nk = (K + GBK - 1) / GBK;
  int splits = 1;
  const int ldw = (N + 3) & ~3;
  if (ws && gx * gy < 192 && nk * GBK >= 128) {
    int cap = 64;
    while (cap > 1 && (size_t)cap * M * ldw > ws_floats) --cap;
    splits = pick_splits(gx * gy, nk, (size_t)M * ldw * sizeof(float), cap, 1.2 * GBK / 16, GBK == 16 ? 4 : 2);
  }
  const int per = std::max(1, (nk + splits - 1) / splits);
  splits = std::max(1, (nk + per - 1) / per);
  float* w = splits > 1 ? ws : nullptr;
  dim3 grid(gx, gy, splits), block(256);
  const int acc = accumulate ? 1 : 0;
  if (a_kc && !b_kc)
    hipLaunchKernelGGL((k_gemm16<true, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, nullptr, 0, 0);
  else if (a_kc && b_kc)
    hipLaunchKernelGGL((k_gemm16<true, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, nullptr, 0, 0);
  else if (!a_kc && !b_kc)
    hipLaunchKernelGGL((k_gemm16<false, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, A2, lda2, M1);
  else
    hipLaunchKernelGGL((k_gemm16<false, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, A2, lda2, M1);
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    const int blocks = (int)std::min<size_t>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, s, ws, ldw, splits, C, ldc, M, N, bias, act, alpha, acc);
  }
}


// The batch of launch_gemm16: operands x-contiguous ([K][M], [K][N]), no bias / activation.  Same split rule, over all problems' tiles.
void launch_gemm16_batch(const Gemm16Batch& bt, int lda, int lda2, int M1, int ldb, int ldc, int M, int N, int K, bool accumulate,
                         hipStream_t s, float* ws, size_t ws_floats) {
  if (bt.n <= 0 || M <= 0 || N <= 0 || K <= 0) return;
  const int gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
  const int nk = (K + GBK - 1) / GBK;
  int splits = 1;
  const int ldw = (N + 3) & ~3;
  if (ws && gx * gy * bt.n < 192 && nk * GBK >= 128) {
    int cap = 64;
    while (cap > 1 && (size_t)cap * bt.n * M * ldw > ws_floats) --cap;
    splits = pick_splits(gx * gy * bt.n, nk, (size_t)M * ldw * sizeof(float), cap, 1.2 * GBK / 16, GBK == 16 ? 4 : 2);
  }
  const int per = std::max(1, (nk + splits - 1) / splits);
  splits = std::max(1, (nk + per - 1) / per);
  float* w = splits > 1 ? ws : nullptr;
  hipLaunchKernelGGL(k_gemm16_b, dim3(gx, gy, splits * bt.n), dim3(256), 0, s, bt, lda, ldb, ldc, M, N, K, accumulate ? 1 : 0, w, ldw, per,
                     splits, lda2, M1);
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    const int blocks = (int)std::min<size_t>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(k_splitk_reduce_b, dim3(blocks, bt.n), dim3(256), 0, s, ws, ldw, splits, bt, ldc, M, N, accumulate ? 1 : 0);
  }
}

void launch_gemm_mapped(const float* A, int lda, const GemmRowMap& ma, const float* A2, int lda2, int M1, bool a_kc, const float* B, int ldb,
                        bool b_kc, float* C, int ldc, int M, int N, int K, const float* bias, int act, float alpha,
                        bool accumulate, hipStream_t s, float* ws, size_t ws_floats) {
  if (M <= 0 || N <= 0 || K <= 0) return;
  const bool mapped = ma.rows_per > 0;
  if (N <= NBN && !b_kc && !A2 && M >= NBM && !mapped) {         // narrow output: 256 x 32 tiles
    const int gy = (M + NBM - 1) / NBM, nk = (K + BK - 1) / BK, ldw = (N + 3) & ~3;
    int splits = 1;
    if (ws && gy < 192 && nk >= 8) {
      int cap = 64;
      while (cap > 1 && (size_t)cap * M * ldw > ws_floats) --cap;
      splits = pick_splits(gy, nk, (size_t)M * ldw * sizeof(float), cap);
    }
    const int per = std::max(1, (nk + splits - 1) / splits);
    splits = std::max(1, (nk + per - 1) / per);
    float* w = splits > 1 ? ws : nullptr;
    dim3 grid(1, gy, splits), block(256);
    const int acc = accumulate ? 1 : 0;
    if (a_kc) hipLaunchKernelGGL((k_gemm_n32<true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per);
    else hipLaunchKernelGGL((k_gemm_n32<false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per);
    if (splits > 1) {
      const size_t total = (size_t)M * N;
      const int blocks = (int)std::min<size_t>(2048, (total + 255) / 256);
      hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, s, ws, ldw, splits, C, ldc, M, N, bias, act, alpha, acc);
    }
    return;
  }
  // k_gemm (stream-K, loader waves, LDS ring) for the window views and for the products with enough work per tile and enough
  // tiles; k_gemm16 for the rest (measured, tools/ubench/gemm_bench.hip; the two are within noise of each other in between)
  const double outs = (double)M * N;
  if (!mapped && !(K >= 256 && outs >= 4.0e6) && !(K >= 2048 && outs >= 1.5e6)) {
    launch_gemm16(A, lda, a_kc ? nullptr : A2, lda2, M1, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, act, alpha, accumulate, s, ws, ws_floats);
    return;
  }
  GemmArgs g{};
  g.A = A; g.A2 = a_kc ? nullptr : A2; g.B = B; g.bias = bias; g.C = C;
  g.lda = lda; g.lda2 = lda2; g.M1 = M1; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.act = act; g.accumulate = accumulate ? 1 : 0; g.alpha = alpha;
  g.ma = ma;
  if (mapped) {                 // the SEGAN-style convolutions: a window view times a [K][N] filter / gradient (b_kc never occurs)
    g.A2 = nullptr;
    if (a_kc) launch_layout<true, false, true>(g, s, ws, ws_floats);
    else launch_layout<false, false, true>(g, s, ws, ws_floats);
  }
  else if (a_kc && !b_kc) launch_layout<true, false, false>(g, s, ws, ws_floats);
  else if (a_kc && b_kc) launch_layout<true, true, false>(g, s, ws, ws_floats);
  else if (!a_kc && !b_kc) launch_layout<false, false, false>(g, s, ws, ws_floats);
  else launch_layout<false, true, false>(g, s, ws, ws_floats);
}

// nb same-shaped products C_b = [A_b | A2_b]^T-style (x-contiguous operands, no bias / activation) as ONE stream-K launch of k_gemm
// at 128 x 128 tiles: the unit space runs over all problems' tiles, so a worker's run is nb times longer and one fix-up launch
// serves them all (round 5: the generator's three dK products, 120 tiles each: 3 x 259 us -> 703 us in tools/ubench/gemm_bench explore).
// false: not applicable (the caller launches the products one by one).
bool launch_gemm_batch(int nb, const float* const* A, int lda, const float* const* A2, int lda2, int M1, const float* const* B, int ldb,
                       float* const* C, int ldc, int M, int N, int K, bool accumulate, hipStream_t s, float* ws, size_t ws_floats) {
  static const bool on = [] { const char* e = getenv("RSRGAN_GEMM_BATCH"); return !e || atoi(e) != 0; }();
  if (!on || nb < 2 || nb > GEMM_MAXB || M <= 0 || N <= 0 || K <= 0 || !ws) return false;
  const double outs = (double)M * N;
  if (!(K >= 256 && outs >= 4.0e6) && !(K >= 2048 && outs >= 1.5e6)) return false;      // (k_gemm16's products: launch_gemm16_batch)
  GemmArgs g{};
  g.A = A[0]; g.A2 = A2 ? A2[0] : nullptr; g.B = B[0]; g.bias = nullptr; g.C = C[0];
  g.lda = lda; g.lda2 = lda2; g.M1 = M1; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.act = 0; g.accumulate = accumulate ? 1 : 0; g.alpha = 0.f;
  g.ma = GemmRowMap{0, 0, 0};
  GemmBatch bt{};
  bt.nb = nb;
  for (int b = 0; b < nb; ++b) { bt.Ab[b] = A[b]; bt.A2b[b] = A2 ? A2[b] : nullptr; bt.Bb[b] = B[b]; bt.Cb[b] = C[b]; }
  // 192 x 256 tiles on the four self-loading waves of k_gemm_s (accumulators in AGPRs): 0.018 B/FLOP of operand ingest instead of
  // 0.031 at 128 x 128, and M = 560 is 2.9 tiles of 192 (2.8 % padding) against 4.4 of 128 (12.5 %).  RSRGAN_GEMM_BATCH=2: the 128 x 128 form.
  static const int form = [] { const char* e = getenv("RSRGAN_GEMM_BATCH"); return e ? atoi(e) : 1; }();
  const int pad192 = (M + 191) / 192 * 192, pad128 = (M + 127) / 128 * 128;
  if (form == 1 && pad192 < pad128 && 2 * (size_t)g_gemm_workers * 192 * 256 <= ws_floats) {
    const int tn1 = (N + 255) / 256;
    // k_gemm_s fills a CU's register file: nothing of the side stream (dWp, column sums, the FCs' gradients) co-resides with it, and
    // behind the launch that work is exposed.  RSRGAN_GEMM_BATCH_W workers (a multiple of 8) leave the other CUs to the side stream.
    static const int bw = [] { const char* e = getenv("RSRGAN_GEMM_BATCH_W"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 256 ? v & ~7 : 0; }();
    const Plan pl = plan_cfg(M, nb * tn1 * 256, K, 192, 256, bw ? std::min(bw, g_gemm_workers) : g_gemm_workers, ws, ws_floats);
    if (!pl.whole) { launch_cfg_s<false, false, 3, 4, 2, false, GemmBatch>(g, pl, s, ws, &bt); return true; }
  }
  const int tn1 = (N + 127) / 128;
  const Plan pl = plan_cfg(M, nb * tn1 * 128, K, 128, 128, g_gemm_workers, ws, ws_floats);
  launch_cfg<false, false, 2, 2, 2, false>(g, pl, s, ws, &bt);
  return true;
}

void launch_gemm2(const float* A, int lda, const float* A2, int lda2, int M1, bool a_kc, const float* B, int ldb, bool b_kc,
                  float* C, int ldc, int M, int N, int K, const float* bias, int act, float alpha, bool accumulate,
                  hipStream_t s, float* ws, size_t ws_floats) {
  const GemmRowMap none{0, 0, 0};
  launch_gemm_mapped(A, lda, none, A2, lda2, M1, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, act, alpha, accumulate, s, ws, ws_floats);
}

void launch_gemm(const float* A, int lda, bool a_kc, const float* B, int ldb, bool b_kc, float* C, int ldc,
                 int M, int N, int K, const float* bias, int act, float alpha, bool accumulate, hipStream_t s,
                 float* ws, size_t ws_floats) {
  launch_gemm2(A, lda, nullptr, 0, 0, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, act, alpha, accumulate, s, ws, ws_floats);
}

}  // namespace rsr
