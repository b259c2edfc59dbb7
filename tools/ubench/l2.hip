// l2.hip -- stand-alone micro-benchmark (not part of the product): what survives in the per-XCD L2s across a kernel boundary?
//   A  per-launch time of an XCD-affine weight pull vs total footprint (8..64 MB = 1..8 MB per XCD), launches replayed from a
//      hipGraph (no host launch cost), against the in-kernel (L2-hot) re-read time of the same bytes
//   B  the same with a dirty-data writer kernel between pulls (plain / nt / sc1 stores) and with a second weight set alternating
//   C  block -> XCD map of graph-replayed launches (HW_REG_XCC_ID vs blockIdx % 8)
//   D  access shape: fully coalesced 16 B/lane vs the MFMA row-fragment shape (lane l: row l&15, 16 B at k-slot l>>4)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 l2.hip -o l2      Run: ./l2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// block b pulls slice (b%8)*(nslices/8) + b/8 : blocks with equal b % 8 (one XCD under round-robin dispatch) own one contiguous
// eighth of W.  FRAG: the 64 lanes of a wave read 16 rows x 64 B (row stride = rowbytes) instead of 1 KB contiguous.
template <int LOADS, bool FRAG>
__global__ __launch_bounds__(512) void k_pull(const float* W, size_t slice_floats, int nslices, int passes, int rowbytes, float* sink) {
  const int b = blockIdx.x % nslices;
  const int x = b & 7, i = b >> 3;
  const size_t s = (size_t)x * (nslices / 8) + i;
  const float* base = W + s * slice_floats;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(slice_floats * 4), 0x00020000);
  float acc = 0.f;
  const int per_sweep = 512 * 4 * LOADS;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int p = 0; p < passes; ++p) {
    for (size_t o = 0; o < slice_floats; o += per_sweep) {
      u32x4 v[LOADS];
#pragma unroll
      for (int l = 0; l < LOADS; ++l) {
        unsigned off;
        if (FRAG) {
          // a wave owns 16 rows; load l covers 64 B of k per row: rows (wv*16 + lane&15), k bytes l*64 + (lane>>4)*16
          const unsigned row = (unsigned)(wv * 16 + (lane & 15));
          off = (unsigned)(o * 4) + row * (unsigned)rowbytes + (unsigned)(l * 64 + (lane >> 4) * 16);
        } else {
          off = (unsigned)((o + (size_t)l * 2048 + threadIdx.x * 4) * 4);
        }
        v[l] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      }
#pragma unroll
      for (int l = 0; l < LOADS; ++l) acc += __uint_as_float(v[l].x) + __uint_as_float(v[l].w);
    }
    asm volatile("" : "+v"(acc));
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

// writes n floats (mode 0 plain, 1 nt, 2 sc1), 16 B per lane
__global__ __launch_bounds__(256) void k_write(float* p, size_t n, int mode) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(n * 4), 0x00020000);
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
    if (mode == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, (unsigned)(i * 4), 0, 0);
    else if (mode == 1) __builtin_amdgcn_raw_buffer_store_b128(v, r, (unsigned)(i * 4), 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, (unsigned)(i * 4), 0, 16);
  }
}

__global__ void k_xcc(int* out) {
  if (threadIdx.x == 0) {
    unsigned x = __builtin_amdgcn_s_getreg((20u) | (0u << 6) | ((4u - 1u) << 11));       // HW_REG_XCC_ID bits [3:0]
    out[blockIdx.x] = (int)x;
  }
}
__global__ void k_empty() {}

static float time_graph(hipStream_t s, int per_graph, int replays, const std::function<void()>& f) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < per_graph; ++i) f();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1000.f / (per_graph * replays);
}

int main(int argc, char** argv) {
  const bool only_e = argc > 1 && argv[1][0] == 'E';
  const int GN = only_e ? 10 : 100, GR = only_e ? 2 : 5;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t maxb = (size_t)160 << 20;
  float* W; CK(hipMalloc(&W, maxb)); CK(hipMemset(W, 0, maxb));
  float* W2; CK(hipMalloc(&W2, maxb)); CK(hipMemset(W2, 0, maxb));
  float* D; CK(hipMalloc(&D, (size_t)64 << 20)); CK(hipMemset(D, 0, (size_t)64 << 20));
  float* sink; CK(hipMalloc(&sink, 1 << 16));
  const float empty = time_graph(s, 200, 5, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s); });
  printf("empty kernel in a graph: %.2f us per launch\n", empty);

  if (!only_e) {
  printf("== A: XCD-affine pull, 256 blocks x 512 threads, replayed from a hipGraph (us per launch; 'hot' = extra in-kernel pass)\n");
  for (int mb : {4, 8, 16, 20, 24, 28, 32, 40, 48, 64, 96, 128}) {
    const size_t total = (size_t)mb << 20;
    const int nslices = 256;
    const size_t slice = total / 4 / nslices;
    auto pull = [&](const float* w, int passes) { hipLaunchKernelGGL((k_pull<10, false>), dim3(256), dim3(512), 0, s, w, slice, nslices, passes, 0, sink); };
    const float p1 = time_graph(s, 100, 5, [&] { pull(W, 1); });
    const float p3 = time_graph(s, 100, 5, [&] { pull(W, 3); });
    const float alt = time_graph(s, 100, 5, [&] { pull(W, 1); pull(W2, 1); });      // per PAIR (time_graph divides by the number of calls of the lambda)
    printf("total %3d MB (%4.1f MB/XCD): b2b %.2f us -> pull %.2f us = %.1f TB/s | in-kernel hot pass %.2f us = %.1f TB/s | two sets alternating: %.2f us per pair\n",
           mb, mb / 8.0, p1, p1 - empty, total / ((p1 - empty) * 1e6), (p3 - p1) / 2, total / ((p3 - p1) / 2 * 1e6), alt);
  }
  printf("== B: 20 MB pull with a writer kernel in between (us per pair; writer alone in brackets)\n");
  {
    const size_t total = (size_t)20 << 20; const int nslices = 256; const size_t slice = total / 4 / nslices;
    auto pull = [&] { hipLaunchKernelGGL((k_pull<10, false>), dim3(256), dim3(512), 0, s, W, slice, nslices, 1, 0, sink); };
    for (int wmb : {1, 4, 8, 16}) {
      for (int mode = 0; mode < 3; ++mode) {
        const size_t n = ((size_t)wmb << 20) / 4;
        const float w = time_graph(s, 100, 5, [&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, s, D, n, mode); });
        const float both = time_graph(s, 100, 5, [&] { pull(); hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, s, D, n, mode); });
        printf("writer %2d MB %-5s: pair %.2f us (writer alone %.2f) -> pull %.2f us\n", wmb, mode == 0 ? "plain" : mode == 1 ? "nt" : "sc1", both, w, both - w - empty);
      }
    }
  }
  printf("== C: block -> XCD map under graph replay (HW_REG_XCC_ID of blocks 0..9, by size of the PRECEDING kernel's grid)\n");
  {
    int* xo; CK(hipMalloc(&xo, 4096 * 4));
    for (int prev : {0, 256, 300, 301, 303, 1024}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      if (prev) hipLaunchKernelGGL(k_empty, dim3(prev), dim3(512), 0, s);
      hipLaunchKernelGGL(k_xcc, dim3(1024), dim3(512), 0, s, xo);
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      std::vector<int> h(1024);
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), xo, 1024 * 4, hipMemcpyDeviceToHost));
        int consistent = 1;
        for (int b = 8; b < 1024; ++b) consistent &= (h[b] == h[b - 8]);
        printf("prev grid %4d, replay %d: ids %d %d %d %d %d %d %d %d %d %d  (period-8 everywhere: %s)\n", prev, rep, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9],
               consistent ? "yes" : "NO");
      }
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(xo));
  }
  printf("== D: access shape, 20 MB, 256 blocks (slice = 128 rows x 640 B): coalesced vs MFMA row-fragment loads\n");
  {
    const size_t total = (size_t)20 << 20; const int nslices = 256; const size_t slice = total / 4 / nslices;     // 80 KB = 128 rows x 640 B
    for (int frag = 0; frag < 2; ++frag) {
      auto pull = [&](int passes) {
        if (frag) hipLaunchKernelGGL((k_pull<10, true>), dim3(256), dim3(512), 0, s, W, slice, nslices, passes, 640, sink);
        else hipLaunchKernelGGL((k_pull<10, false>), dim3(256), dim3(512), 0, s, W, slice, nslices, passes, 0, sink);
      };
      const float p1 = time_graph(s, 100, 5, [&] { pull(1); });
      const float p3 = time_graph(s, 100, 5, [&] { pull(3); });
      printf("%s: b2b %.2f us (pull %.2f) | hot pass %.2f us\n", frag ? "row-fragment (16 rows x 64 B per wave-load)" : "coalesced (1 KB per wave-load)        ", p1, p1 - empty, (p3 - p1) / 2);
    }
  }
  }
  printf("== E: what breaks L2 retention?  20 MB pull alternating with a second launch (us per PAIR, graph replay)\n");
  {
    const size_t total = (size_t)20 << 20; const int nslices = 256; const size_t slice = total / 4 / nslices;
    auto pullw = [&](const float* w, float* snk) { hipLaunchKernelGGL((k_pull<10, false>), dim3(256), dim3(512), 0, s, w, slice, nslices, 1, 0, snk); };
    const float e1 = time_graph(s, GN, GR, [&] { pullw(W, sink); pullw(W, sink); });
    const float e2 = time_graph(s, GN, GR, [&] { pullw(W, sink); pullw(W, sink + 1024); });
    const float e3 = time_graph(s, GN, GR, [&] { pullw(W, sink); hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s); });
    const float e4 = time_graph(s, GN, GR, [&] { pullw(W, sink); hipLaunchKernelGGL((k_pull<10, false>), dim3(256), dim3(512), 0, s, W2, (size_t)64, nslices, 1, 0, sink); });
    const float e5 = time_graph(s, GN, GR, [&] { pullw(W, sink); pullw(W + ((size_t)32 << 18), sink); });
    const float e6 = time_graph(s, GN, GR, [&] { pullw(W, sink); hipLaunchKernelGGL((k_pull<5, false>), dim3(256), dim3(512), 0, s, W, slice, nslices, 1, 0, sink); });
    const float e7 = time_graph(s, GN, GR, [&] { pullw(W, sink); hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, s, D, (size_t)1 << 14, 0); });
    const float e8 = time_graph(s, GN, GR, [&] { pullw(W, sink); pullw(W2, sink); pullw(W, sink); pullw(W2, sink); }) / 2.f;
    printf("e1 same W / same W                    : %.2f\n", e1);
    printf("e2 same W, other sink pointer         : %.2f\n", e2);
    printf("e3 W / empty kernel                   : %.2f\n", e3);
    printf("e4 W / same kernel on 64 KB of W2     : %.2f\n", e4);
    printf("e5 W / 20 MB at W + 32 MB (same alloc): %.2f\n", e5);
    printf("e6 W / same W, other instantiation    : %.2f\n", e6);
    printf("e7 W / 64 KB plain-store writer       : %.2f\n", e7);
    printf("e8 W / W2 (20 MB each)                : %.2f\n", e8);
  }
  return 0;
}
