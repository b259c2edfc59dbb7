// gemm.hip -- time-batched fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak on
// MI355X).  Used for everything that is NOT on the serial recurrence: the input/output
// fully_connected layers (models/lstm.py:82-87,121-124), the x-part of every LSTM kernel
// batched over all T*B frames, and all weight/data gradients batched over time.
//
// 128x128x16 block tile, 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 MFMA tiles.
// Operands are staged global -> VGPR (prefetched one k-tile ahead) -> LDS in a k-major image
// As[k][m], Bs[k][n] (row stride 132 floats: 16-B aligned rows, 4-bank shift per k) so that the
// MFMA fragment reads (lane l: row/col = l&31, k = l>>5) are 32 consecutive floats per half-wave
// = conflict-free ds_read_b32.
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <algorithm>
#include <array>
#include <cstdlib>
#include <map>

#include "kernels.h"

namespace rsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;
#ifndef RSR_GBK
#define RSR_GBK 16
#endif
constexpr int GBK = RSR_GBK;          // k-tile of k_gemm; 32 measured equal (115-127 TF at 4096^3, 79-87 at 6400x3040x280) at lower occupancy

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
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, int M, int N, int K,
                                              const float* __restrict__ bias, int act, float alpha, int accumulate,
                                              float* __restrict__ ws, int ldw, int kt_per_split,
                                              const float* __restrict__ A2, int lda2, int M1) {
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
  const int kt0 = blockIdx.z * kt_per_split;
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
          ws[((size_t)blockIdx.z * M + row) * ldw + col] = acc[i][j][r];
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

// dst[r][0:w1] = a[r][0:w1], dst[r][w1:w1+w2] = b[r][0:w2]   (w1 % 4 == 0; float4 moves; pad columns zero)
__global__ __launch_bounds__(256) void k_concat_cols(const float* __restrict__ a, int lda, int w1, const float* __restrict__ b, int ldb, int w2,
                                                     float* __restrict__ dst, int ldd, int rows) {
  const int c4n = ldd >> 2;
  const size_t n = (size_t)rows * c4n;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    float4 v;
    if (c < w1) v = *reinterpret_cast<const float4*>(a + r * lda + c);
    else {
      const int cb = c - w1;                      // (ldb is padded to 4: the last float4 of b may read its zero padding)
      v = cb < w2 ? *reinterpret_cast<const float4*>(b + r * ldb + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    *reinterpret_cast<float4*>(dst + r * ldd + c) = v;
  }
}

// ---- plain library GEMM for the big epilogue-free products (weight gradients dK = [x|m]^T.dZ: 560 x 3040 x 6400, data gradients):
// hipBLASLt, resolved at run time (dlopen; the header only supplies types).  Without the library, or with RSRGAN_BLAS=0,
// everything stays on k_gemm.  One algorithm per (shape, layout), taken once from the heuristic and cached. ----
namespace {
struct LtPlan { hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t a, b, c; hipblasLtMatmulAlgo_t algo; size_t ws; bool ok; };
struct Blas {
  bool tried = false, ok = false;
  hipblasLtHandle_t handle = nullptr;
  std::map<hipStream_t, void*> ws;          // one work space per stream that issues products (concurrent streams must not share one)
  size_t ws_bytes = 32u << 20;
  decltype(&hipblasLtCreate) create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
  decltype(&hipblasLtMatmul) matmul = nullptr;
  std::map<std::array<int64_t, 8>, LtPlan> plans;
};
Blas g_blas;
template <class F>
bool lt_sym(void* lib, const char* name, F& f) { f = reinterpret_cast<F>(dlsym(lib, name)); return f != nullptr; }
bool blas_ready() {
  if (g_blas.tried) return g_blas.ok;
  g_blas.tried = true;
  const char* e = getenv("RSRGAN_BLAS");
  if (e && atoi(e) == 0) return false;
  void* lib = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return false;
  Blas& g = g_blas;
  if (!lt_sym(lib, "hipblasLtCreate", g.create) || !lt_sym(lib, "hipblasLtMatmulDescCreate", g.desc_create) ||
      !lt_sym(lib, "hipblasLtMatmulDescSetAttribute", g.desc_set) || !lt_sym(lib, "hipblasLtMatrixLayoutCreate", g.layout_create) ||
      !lt_sym(lib, "hipblasLtMatmulPreferenceCreate", g.pref_create) || !lt_sym(lib, "hipblasLtMatmulPreferenceSetAttribute", g.pref_set) ||
      !lt_sym(lib, "hipblasLtMatmulPreferenceDestroy", g.pref_destroy) || !lt_sym(lib, "hipblasLtMatmulAlgoGetHeuristic", g.heuristic) ||
      !lt_sym(lib, "hipblasLtMatmul", g.matmul))
    return false;
  if (g.create(&g.handle) != HIPBLAS_STATUS_SUCCESS || !g.handle) return false;
  if (const char* w = getenv("RSRGAN_BLAS_WS")) g.ws_bytes = (size_t)atoll(w);      // 0: only algorithms without a work space
  g.ok = true;
  return true;
}
// row-major C[M][N] = op(A).op(B)  ==  column-major C^T[N][M] = op(B)^T.op(A)^T: the library's first operand is B, its second A.
// B stored [N][K] (k contiguous) is the column-major K x N matrix (transpose it), stored [K][N] it is N x K already;
// A stored [M][K] is the column-major K x M matrix (as is), stored [K][M] it is M x K (transpose it).
// bias (per output column = per row of the library's D) and relu ride the library's epilogue (BIAS / RELU_BIAS)
bool blas_gemm(const float* A, int lda, bool a_kc, const float* B, int ldb, bool b_kc, float* C, int ldc, int M, int N, int K, hipStream_t s,
               const float* bias = nullptr, bool relu = false) {
  Blas& g = g_blas;
  const uint32_t epi = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS) : HIPBLASLT_EPILOGUE_DEFAULT;
  if (relu && !bias) return false;
  const std::array<int64_t, 8> key = {M, N, K, lda, ldb, ldc, (a_kc ? 1 : 0) | (b_kc ? 2 : 0), (int64_t)epi};
  auto it = g.plans.find(key);
  if (it == g.plans.end()) {
    LtPlan p{}; p.ok = false;
    const int32_t opB = b_kc ? HIPBLAS_OP_T : HIPBLAS_OP_N, opA = a_kc ? HIPBLAS_OP_N : HIPBLAS_OP_T;
    hipblasLtMatmulPreference_t pref = nullptr;
    hipblasLtMatmulHeuristicResult_t res[1];
    int found = 0;
    if (g.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS &&
        g.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opB, sizeof(opB)) == HIPBLAS_STATUS_SUCCESS &&
        g.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opA, sizeof(opA)) == HIPBLAS_STATUS_SUCCESS &&
        g.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS &&
        (!bias || g.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) == HIPBLAS_STATUS_SUCCESS) &&
        g.layout_create(&p.a, HIP_R_32F, b_kc ? K : N, b_kc ? N : K, ldb) == HIPBLAS_STATUS_SUCCESS &&
        g.layout_create(&p.b, HIP_R_32F, a_kc ? K : M, a_kc ? M : K, lda) == HIPBLAS_STATUS_SUCCESS &&
        g.layout_create(&p.c, HIP_R_32F, N, M, ldc) == HIPBLAS_STATUS_SUCCESS &&
        g.pref_create(&pref) == HIPBLAS_STATUS_SUCCESS &&
        g.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &g.ws_bytes, sizeof(g.ws_bytes)) == HIPBLAS_STATUS_SUCCESS &&
        g.heuristic(g.handle, p.desc, p.a, p.b, p.c, p.c, pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS && found > 0 &&
        res[0].workspaceSize <= g.ws_bytes) {
      p.algo = res[0].algo; p.ws = res[0].workspaceSize; p.ok = true;
    }
    if (pref) g.pref_destroy(pref);
    it = g.plans.emplace(key, p).first;
  }
  const LtPlan& p = it->second;
  if (!p.ok) return false;
  const float one = 1.f, zero = 0.f;
  void* wsp = nullptr;
  if (p.ws) {
    auto w = g.ws.find(s);
    if (w == g.ws.end()) {
      void* mem = nullptr;
      if (g.ws.size() >= 8) {                     // streams of models long gone: start over (never inside a capture: the first
        (void)hipDeviceSynchronize();             // product of a stream runs in a segment's eager first pass)
        for (auto& kv : g.ws) (void)hipFree(kv.second);
        g.ws.clear();
      }
      if (hipMalloc(&mem, g.ws_bytes) != hipSuccess) return false;
      w = g.ws.emplace(s, mem).first;
    }
    wsp = w->second;
  }
  if (bias && g.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return false;
  return g.matmul(g.handle, p.desc, &one, B, p.a, A, p.b, &zero, C, p.c, C, p.c, &p.algo, wsp, p.ws, s) == HIPBLAS_STATUS_SUCCESS;
}
}  // namespace

void launch_gemm2(const float* A, int lda, const float* A2, int lda2, int M1, bool a_kc, const float* B, int ldb, bool b_kc,
                  float* C, int ldc, int M, int N, int K, const float* bias, int act, float alpha, bool accumulate,
                  hipStream_t s, float* ws, size_t ws_floats) {
  if (M <= 0 || N <= 0) return;
  // (measured, MI355X: the library wins where the output alone fills the chip -- 6400 x 1024 x 1024: 101 vs 90 TFLOP/s, 280 x 3040 x 6400:
  //  105 vs 91 -- and loses on small outputs with a long K, where k_gemm's split-K is the better plan: 760 x 280 x 6400 40 vs 53)
  const int ldcat = (M + 3) & ~3;
  const bool cat = A2 && !a_kc && ws && (size_t)K * ldcat <= ws_floats;      // two-source operand: stack it once, one product
  const double out_elems = (double)((A2 && !cat) ? std::min(M1, M - M1) : M) * N;
  const bool epi_ok = act == 0 || (act == 2 && bias);               // none / bias / bias + relu (leaky-relu stays on k_gemm)
  if (epi_ok && !(bias && A2) && !accumulate && out_elems >= 0.8e6 && K >= 256 && N > NBN && (!A2 || !a_kc) && blas_ready()) {
    bool ok;
    if (cat) {                  // [x_t | m_{t-1}] as one [K][M] operand in the (otherwise unused) split-K work space: one 560-row
      const size_t total = (size_t)K * (ldcat >> 2);      // product runs at 131 TFLOP/s, two 280-row halves at 105
      hipLaunchKernelGGL(k_concat_cols, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, A, lda, M1, A2, lda2,
                         M - M1, ws, ldcat, K);
      ok = blas_gemm(ws, ldcat, false, B, ldb, b_kc, C, ldc, M, N, K, s);
    } else if (A2) {
      ok = blas_gemm(A, lda, a_kc, B, ldb, b_kc, C, ldc, M1, N, K, s) &&
           blas_gemm(A2, lda2, a_kc, B, ldb, b_kc, C + (size_t)M1 * ldc, ldc, M - M1, N, K, s);
    } else {
      ok = blas_gemm(A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, s, bias, act == 2);
    }
    if (ok) return;
    // (no algorithm for this shape, or a failed call: k_gemm for this product)
  }
  if (N <= NBN && !b_kc && !A2 && M >= NBM) {         // narrow output: 256 x 32 tiles
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
    hipLaunchKernelGGL((k_gemm<true, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, nullptr, 0, 0);
  else if (a_kc && b_kc)
    hipLaunchKernelGGL((k_gemm<true, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, nullptr, 0, 0);
  else if (!a_kc && !b_kc)
    hipLaunchKernelGGL((k_gemm<false, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, A2, lda2, M1);
  else
    hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, act, alpha, acc, w, ldw, per, A2, lda2, M1);
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    const int blocks = (int)std::min<size_t>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, s, ws, ldw, splits, C, ldc, M, N, bias, act, alpha, acc);
  }
}

void launch_gemm(const float* A, int lda, bool a_kc, const float* B, int ldb, bool b_kc, float* C, int ldc,
                 int M, int N, int K, const float* bias, int act, float alpha, bool accumulate, hipStream_t s,
                 float* ws, size_t ws_floats) {
  launch_gemm2(A, lda, nullptr, 0, 0, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, act, alpha, accumulate, s, ws, ws_floats);
}

}  // namespace rsr
