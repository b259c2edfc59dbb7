// gemm_bench.hip -- correctness screen + timing of csrc/gemm.hip (k_gemm, stream-K) next to hipBLASLt (measurement only: the
// product never links the library).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm_bench.hip -lhipblaslt -o gemm_bench
// Run: ./gemm_bench [check] [time] [abl] [workers=N]   (abl: timing of the headline shapes only; builds with -DRSR_GEMM_ABL=n
// switch parts of the k-loop off)
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../rsrgan_amd/csrc/gemm.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using rsr::GemmRowMap;

// reference: one thread per output element, double accumulation, same addressing conventions as launch_gemm_mapped
__global__ void k_ref(const float* A, int lda, GemmRowMap ma, const float* A2, int lda2, int M1, int a_kc, const float* B, int ldb, int b_kc,
                      double* C, int M, int N, int K) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  double s = 0.0;
  for (int k = 0; k < K; ++k) {
    float a;
    if (a_kc) {
      const long long ro = ma.rows_per > 0 ? (long long)(m / ma.rows_per) * ma.outer + (long long)(m % ma.rows_per) * ma.inner : (long long)m * lda;
      a = A[ro + k];
    } else {
      const long long ro = ma.rows_per > 0 ? (long long)(k / ma.rows_per) * ma.outer + (long long)(k % ma.rows_per) * ma.inner : (long long)k * lda;
      a = (A2 && m >= M1) ? A2[(long long)k * lda2 + (m - M1)] : A[ro + m];
    }
    const float b = b_kc ? B[(long long)n * ldb + k] : B[(long long)k * ldb + n];
    s += (double)a * (double)b;
  }
  C[i] = s;
}

static int pad4(int x) { return (x + 3) & ~3; }
static float* dev_rand(size_t n, unsigned seed, bool zero = false) {
  std::vector<float> h(n);
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  for (auto& v : h) v = zero ? 0.f : d(g);
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return p;
}
// operand stored [rows][pad4(cols)] with zero padding columns
static bool g_zero_fill = false;
static float* dev_mat(int rows, int cols, unsigned seed) {
  const int ld = pad4(cols);
  std::vector<float> h((size_t)rows * ld, 0.f);
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) h[(size_t)r * ld + c] = g_zero_fill ? 0.f : d(g);
  float* p; CK(hipMalloc(&p, h.size() * sizeof(float)));
  CK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return p;
}

static float* g_ws = nullptr;
static const size_t g_ws_floats = (size_t)16 << 20;

static int check_one(int M, int N, int K, bool akc, bool bkc, int act, bool accumulate, bool two_src, const char* tag) {
  float* A = akc ? dev_mat(M, K, 1) : dev_mat(K, M, 1);
  float* B = bkc ? dev_mat(N, K, 2) : dev_mat(K, N, 2);
  const int lda = akc ? pad4(K) : pad4(M), ldb = bkc ? pad4(K) : pad4(N), ldc = pad4(N);
  float* A2 = nullptr; int lda2 = 0, M1 = 0;
  if (two_src) { M1 = (M / 2) & ~3; A2 = dev_mat(K, M - M1, 5); lda2 = pad4(M - M1); }
  float* bias = dev_mat(1, N, 3);
  float* C = dev_rand((size_t)M * ldc, 4);
  std::vector<float> c0((size_t)M * ldc);
  CK(hipMemcpy(c0.data(), C, c0.size() * sizeof(float), hipMemcpyDeviceToHost));
  double* R; CK(hipMalloc(&R, (size_t)M * N * sizeof(double)));
  GemmRowMap none{0, 0, 0};
  hipLaunchKernelGGL(k_ref, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, 0, A, lda, none, A2, lda2, M1, akc ? 1 : 0, B, ldb, bkc ? 1 : 0, R, M, N, K);
  rsr::launch_gemm2(A, lda, A2, lda2, M1, akc, B, ldb, bkc, C, ldc, M, N, K, bias, act, 0.3f, accumulate, 0, g_ws, g_ws_floats);
  CK(hipDeviceSynchronize());
  std::vector<float> c((size_t)M * ldc), hb(pad4(N));
  std::vector<double> r((size_t)M * N);
  CK(hipMemcpy(c.data(), C, c.size() * sizeof(float), hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), R, r.size() * sizeof(double), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), bias, hb.size() * sizeof(float), hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 1;
  int bad_pad = 0;
  for (int m = 0; m < M; ++m) {
    for (int n = 0; n < ldc; ++n) {
      const float got = c[(size_t)m * ldc + n];
      if (n >= N) { if (got != c0[(size_t)m * ldc + n]) ++bad_pad; continue; }
      double want = r[(size_t)m * N + n] + hb[n];
      if (act == 1) want = std::max(want, 0.3 * want); else if (act == 2) want = std::max(want, 0.0);
      if (accumulate) want += c0[(size_t)m * ldc + n];
      maxerr = std::max(maxerr, std::fabs(want - got)); maxref = std::max(maxref, std::fabs(want));
    }
  }
  const bool ok = maxerr / maxref < 2e-5 && bad_pad == 0;
  printf("%s %-10s M=%5d N=%5d K=%5d akc=%d bkc=%d act=%d acc=%d two=%d : rel err %.2e pad-writes %d\n", ok ? "ok  " : "FAIL", tag, M, N, K, akc, bkc, act,
         accumulate, two_src, maxerr / maxref, bad_pad);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(bias)); CK(hipFree(C)); CK(hipFree(R)); if (A2) CK(hipFree(A2));
  return ok ? 0 : 1;
}

// SEGAN-style downconv as a mapped GEMM: x [Bn][Lp][Cin] (already zero-padded), stride 2, kw taps:
//   forward  y[b*Lo + o][co] = sum_k xwin[b, o][k] W[k][co]      (a_kc, rows mapped: outer = Lp*Cin, inner = 2*Cin)
//   wgrad    dW[k][co] = sum_r xwin[r][k] dY[r][co]              (!a_kc, k index mapped)
static int check_mapped(int Bn, int Lo, int Cin, int Cout, int kw) {
  const int Lp = 2 * Lo + kw;                       // padded length (windows 2*o .. 2*o + kw - 1 stay inside)
  const int K = kw * Cin, Mr = Bn * Lo;
  float* x = dev_rand((size_t)Bn * Lp * Cin + 64, 11);
  float* Wt = dev_mat(K, Cout, 12);
  float* dY = dev_mat(Mr, Cout, 13);
  const int ldc = pad4(Cout);
  GemmRowMap map{Lo, (long long)Lp * Cin, 2LL * Cin};
  int fails = 0;
  {
    float* C = dev_rand((size_t)Mr * ldc, 14);
    double* R; CK(hipMalloc(&R, (size_t)Mr * Cout * sizeof(double)));
    hipLaunchKernelGGL(k_ref, dim3((unsigned)(((size_t)Mr * Cout + 255) / 256)), dim3(256), 0, 0, x, 0, map, nullptr, 0, 0, 1, Wt, ldc, 0, R, Mr, Cout, K);
    rsr::launch_gemm_mapped(x, 0, map, nullptr, 0, 0, true, Wt, ldc, false, C, ldc, Mr, Cout, K, nullptr, 0, 0.f, false, 0, g_ws, g_ws_floats);
    CK(hipDeviceSynchronize());
    std::vector<float> c((size_t)Mr * ldc); std::vector<double> r((size_t)Mr * Cout);
    CK(hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), R, r.size() * 8, hipMemcpyDeviceToHost));
    double e = 0, s = 1;
    for (int m = 0; m < Mr; ++m) for (int n = 0; n < Cout; ++n) { e = std::max(e, std::fabs(r[(size_t)m * Cout + n] - c[(size_t)m * ldc + n])); s = std::max(s, std::fabs(r[(size_t)m * Cout + n])); }
    printf("%s mapped fwd   B=%d Lo=%d Cin=%d Cout=%d kw=%d (M=%d N=%d K=%d): rel err %.2e\n", e / s < 2e-5 ? "ok  " : "FAIL", Bn, Lo, Cin, Cout, kw, Mr, Cout, K, e / s);
    fails += !(e / s < 2e-5);
    CK(hipFree(C)); CK(hipFree(R));
  }
  {
    float* C = dev_rand((size_t)K * ldc, 15);
    double* R; CK(hipMalloc(&R, (size_t)K * Cout * sizeof(double)));
    hipLaunchKernelGGL(k_ref, dim3((unsigned)(((size_t)K * Cout + 255) / 256)), dim3(256), 0, 0, x, 0, map, nullptr, 0, 0, 0, dY, ldc, 0, R, K, Cout, Mr);
    rsr::launch_gemm_mapped(x, 0, map, nullptr, 0, 0, false, dY, ldc, false, C, ldc, K, Cout, Mr, nullptr, 0, 0.f, false, 0, g_ws, g_ws_floats);
    CK(hipDeviceSynchronize());
    std::vector<float> c((size_t)K * ldc); std::vector<double> r((size_t)K * Cout);
    CK(hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), R, r.size() * 8, hipMemcpyDeviceToHost));
    double e = 0, s = 1;
    for (int m = 0; m < K; ++m) for (int n = 0; n < Cout; ++n) { e = std::max(e, std::fabs(r[(size_t)m * Cout + n] - c[(size_t)m * ldc + n])); s = std::max(s, std::fabs(r[(size_t)m * Cout + n])); }
    printf("%s mapped wgrad B=%d Lo=%d Cin=%d Cout=%d kw=%d (M=%d N=%d K=%d): rel err %.2e\n", e / s < 2e-5 ? "ok  " : "FAIL", Bn, Lo, Cin, Cout, kw, K, Cout, Mr, e / s);
    fails += !(e / s < 2e-5);
    CK(hipFree(C)); CK(hipFree(R));
  }
  CK(hipFree(x)); CK(hipFree(Wt)); CK(hipFree(dY));
  return fails;
}

// ---- hipBLASLt side (row-major C = op(A) op(B) as column-major C^T = op(B)^T op(A)^T)
static hipblasLtHandle_t g_lt = nullptr;
static void* g_ltws = nullptr;
static double time_lt(const float* A, int lda, bool a_kc, const float* B, int ldb, bool b_kc, float* C, int ldc, int M, int N, int K, int reps) {
  if (!g_lt) { if (hipblasLtCreate(&g_lt) != HIPBLAS_STATUS_SUCCESS) return -1; CK(hipMalloc(&g_ltws, 64u << 20)); }
  hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t la, lb, lc; hipblasLtMatmulPreference_t pref;
  const int32_t opB = b_kc ? HIPBLAS_OP_T : HIPBLAS_OP_N, opA = a_kc ? HIPBLAS_OP_N : HIPBLAS_OP_T;
  size_t wsb = 64u << 20;
  hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F);
  hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opB, sizeof(opB));
  hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opA, sizeof(opA));
  hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, b_kc ? K : N, b_kc ? N : K, ldb);
  hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, a_kc ? K : M, a_kc ? M : K, lda);
  hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, N, M, ldc);
  hipblasLtMatmulPreferenceCreate(&pref);
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb));
  hipblasLtMatmulHeuristicResult_t res[1]; int found = 0;
  if (hipblasLtMatmulAlgoGetHeuristic(g_lt, desc, la, lb, lc, lc, pref, 1, res, &found) != HIPBLAS_STATUS_SUCCESS || !found) return -1;
  const float one = 1.f, zero = 0.f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipblasLtMatmul(g_lt, desc, &one, B, la, A, lb, &zero, C, lc, C, lc, &res[0].algo, g_ltws, res[0].workspaceSize, 0);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipblasLtMatmul(g_lt, desc, &one, B, la, A, lb, &zero, C, lc, C, lc, &res[0].algo, g_ltws, res[0].workspaceSize, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static void time_one(int M, int N, int K, bool akc, bool bkc, int reps = 30) {
  float* A = akc ? dev_mat(M, K, 1) : dev_mat(K, M, 1);
  float* B = bkc ? dev_mat(N, K, 2) : dev_mat(K, N, 2);
  const int lda = akc ? pad4(K) : pad4(M), ldb = bkc ? pad4(K) : pad4(N), ldc = pad4(N);
  float* C; CK(hipMalloc(&C, (size_t)M * ldc * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) rsr::launch_gemm(A, lda, akc, B, ldb, bkc, C, ldc, M, N, K, nullptr, 0, 0.f, false, 0, g_ws, g_ws_floats);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) rsr::launch_gemm(A, lda, akc, B, ldb, bkc, C, ldc, M, N, K, nullptr, 0, 0.f, false, 0, g_ws, g_ws_floats);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double lt = time_lt(A, lda, akc, B, ldb, bkc, C, ldc, M, N, K, reps);
  const double fl = 2.0 * M * N * K;
  printf("M=%5d N=%5d K=%5d akc=%d bkc=%d : k_gemm %.3f ms %6.1f TF | hipBLASLt %.3f ms %6.1f TF\n", M, N, K, akc, bkc, ms, fl / ms / 1e9, lt, lt > 0 ? fl / lt / 1e9 : 0.0);
#if RSR_GEMM_ABL & 8
  { unsigned long long c[4]; rsr::launch_gemm(A, lda, akc, B, ldb, bkc, C, ldc, M, N, K, nullptr, 0, 0.f, false, 0, g_ws, g_ws_floats); CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(rsr::g_gemm_clk), sizeof(c)));
    printf("   worker 0: %llu shader clocks in %llu ticks of 100 MHz -> %.0f MHz\n", c[0], c[1], c[1] ? 100.0 * c[0] / c[1] : 0.0); }
#endif
  fflush(stdout);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
}

int main(int argc, char** argv) {
  bool do_check = argc < 2, do_time = argc < 2, do_abl = false;
  for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "check")) do_check = true; if (!strcmp(argv[i], "time")) do_time = true; if (!strcmp(argv[i], "abl")) do_abl = true;
    if (!strcmp(argv[i], "zero")) g_zero_fill = true;
    if (!strncmp(argv[i], "workers=", 8)) rsr::g_gemm_workers = atoi(argv[i] + 8); }
  CK(hipMalloc(&g_ws, g_ws_floats * sizeof(float)));
  int fails = 0;
  if (do_check) {
    const int shapes[][3] = {{128, 128, 16}, {300, 257, 40}, {1600, 3040, 280}, {37, 1, 40}, {560, 3040, 640}, {257, 280, 1003}, {5, 7, 3},
                             {760, 280, 6400}, {96, 128, 32}, {97, 129, 33}, {6400, 280, 257}, {1000, 40, 143}, {4096, 1024, 512}, {80, 1024, 12800},
                             {560, 3040, 6400}, {6400, 1024, 1024}};
    for (auto& sh : shapes)
      for (int lay = 0; lay < 4; ++lay) {
        const bool akc = lay & 1, bkc = lay & 2;
        fails += check_one(sh[0], sh[1], sh[2], akc, bkc, 1, false, false, "bias+lrelu");
        if (sh[0] * (long long)sh[1] < 2000000) fails += check_one(sh[0], sh[1], sh[2], akc, bkc, 0, true, false, "accumulate");
      }
    fails += check_one(560, 3040, 6400, false, false, 0, false, true, "two-source");
    fails += check_one(537, 300, 77, false, false, 2, false, true, "two-source");
    fails += check_one(560, 3040, 200, false, true, 0, true, true, "two-source");
    fails += check_mapped(3, 50, 16, 32, 20);
    fails += check_mapped(2, 9, 512, 1024, 31);
    fails += check_mapped(4, 257, 64, 128, 31);
    fails += check_mapped(2, 8212, 16, 32, 31);
    fails += check_mapped(5, 2, 32, 48, 3);
    printf("%s: %d failures\n", fails ? "CHECK FAILED" : "CHECK PASSED", fails);
  }
  if (do_abl) {                 // -DRSR_GEMM_ABL=n builds: results are wrong by construction, only the time means something
    printf("ablation mask %d\n", RSR_GEMM_ABL);
    time_one(4096, 4096, 4096, false, false); time_one(4096, 4096, 4096, true, false); time_one(560, 3040, 6400, false, false);
  }
  for (int i = 1; i < argc; ++i)
    if (!strcmp(argv[i], "explore")) {
      // what does the weight-gradient product [x | m]^T dZ (560 x 3040 x 6400, both operands row-major over time) wait for?  The same
      // product with a longer reduction (longer runs per stream-K worker), with three layers' columns side by side (what a batched
      // launch of the stack's three products would be), with M a whole number of tiles, and one 800-deep slab of it
      const int shapes[][3] = {{560, 3040, 6400}, {560, 3040, 19200}, {560, 9120, 6400}, {640, 3072, 6400}, {512, 3072, 6400}, {576, 3040, 6400},
                               {560, 3040, 800}, {560, 3040, 1600}, {1120, 3040, 6400}, {6400, 280, 3040}, {6400, 280, 9120}};
      for (auto& sh : shapes) time_one(sh[0], sh[1], sh[2], false, false);
      time_one(6400, 280, 3040, true, true);
      // d(h0) = dZ_0 . K_x^T: 150 tiles of 128 x 96 -- stream-K over 256 workers + fix-up, or one round of whole tiles on fewer workers?
      for (int wk : {256, 152, 200}) { rsr::g_gemm_workers = wk; printf("workers=%d: ", wk); time_one(6400, 280, 3040, true, true); }
      rsr::g_gemm_workers = 256;
    }
  if (do_time) {
    const int shapes[][3] = {{4096, 4096, 4096}, {560, 3040, 6400}, {6400, 3040, 280}, {6400, 1024, 1024}, {6400, 1024, 2828}, {1024, 1024, 6400},
                             {760, 280, 6400}, {6400, 280, 257}, {32768, 1024, 1024}, {80, 1024, 12800}};
    for (auto& sh : shapes)
      for (int lay = 0; lay < 3; ++lay) {
        const bool akc = lay == 0 || lay == 1, bkc = lay == 1;
        time_one(sh[0], sh[1], sh[2], akc, bkc);
      }
  }
  return fails ? 1 : 0;
}
