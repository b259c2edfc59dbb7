// icache.hip -- stand-alone micro-benchmark (not part of the product): what does straight-line code cost at the start of a kernel?
// The same N VALU instructions per wave as one straight run (N x 4 B of code, every instruction fetched cold after a kernel
// boundary) and as a loop over a 32-instruction body, 256 one-wave-per-SIMD workgroups, replayed from a hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R32(x) R16(x) R16(x)
#define R256(x) R16(R16(x))
#define R1024(x) R4(R256(x))
#define INSN "v_add_f32 %0, %0, %0\n"

template <int KB>
__global__ __launch_bounds__(256) void k_straight(float* sink) {
  float x = (float)threadIdx.x;
  if (KB >= 1) asm volatile(R256(INSN) : "+v"(x));
  if (KB >= 2) asm volatile(R256(INSN) : "+v"(x));
  if (KB >= 4) asm volatile(R256(INSN) R256(INSN) : "+v"(x));
  if (KB >= 8) asm volatile(R1024(INSN) : "+v"(x));
  if (KB >= 16) asm volatile(R1024(INSN) R1024(INSN) : "+v"(x));
  if (x == 123.456f) sink[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_loop(float* sink, int iters) {
  float x = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) asm volatile(R32(INSN) : "+v"(x));
  if (x == 123.456f) sink[threadIdx.x] = x;
}
__global__ void k_empty() {}

static float time_graph(hipStream_t s, int per_graph, int replays, const std::function<void()>& f) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < per_graph; ++i) f();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1000.f / (per_graph * replays);
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* sink; CK(hipMalloc(&sink, 4096));
  const float e = time_graph(s, 200, 5, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); });
  printf("empty kernel: %.2f us per launch\n", e);
  auto row = [&](int kb, float st, float st_alt) {
    const float lp = time_graph(s, 100, 5, [&] { hipLaunchKernelGGL(k_loop, dim3(256), dim3(256), 0, s, sink, kb * 8); });
    printf("%2d KB of code (%5d v_add per wave): straight %.2f us | straight, alternating with a 4 KB kernel %.2f us | loop of 32 %.2f us  -> cold fetch costs %.2f us\n", kb, kb * 256, st, st_alt,
           lp, st - lp);
  };
#define ROW(KB) row(KB, time_graph(s, 100, 5, [&] { hipLaunchKernelGGL(k_straight<KB>, dim3(256), dim3(256), 0, s, sink); }), \
                    time_graph(s, 100, 5, [&] { hipLaunchKernelGGL(k_straight<KB>, dim3(256), dim3(256), 0, s, sink); hipLaunchKernelGGL(k_straight<4>, dim3(256), dim3(256), 0, s, sink); }) - \
                    time_graph(s, 100, 5, [&] { hipLaunchKernelGGL(k_straight<4>, dim3(256), dim3(256), 0, s, sink); }))
  ROW(1); ROW(2); ROW(4); ROW(8); ROW(16);
  return 0;
}
