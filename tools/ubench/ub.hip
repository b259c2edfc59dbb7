// ub.hip -- stand-alone micro-benchmarks that decide the schedule of the recurrence on MI355X (gfx950):
//   t1  dependent-launch floor vs grid / block / kernarg size / prologue shape (eager and hipGraph)
//   t2  do a launch's weights survive in the XCD L2s across a kernel boundary?  cold / back-to-back / in-kernel re-read
//   t3  XCD-hierarchical grid barrier (MI355X_MICROARCH.md "barrier-xcd") with and without a published payload,
//       payload checked word by word every iteration
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ub.hip -o ub      Run: ./ub [t1|t2|t3|all]
// Not part of the product library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define KA_ABLATE 1
#include "../../rsrgan_amd/csrc/kernels.hip"
#ifndef DL_PROF_BLOCK
#define DL_PROF_BLOCK 0
#endif
#define DL_PROF 1
#include "../../rsrgan_amd/csrc/dlstm.hip"
#define PN_ABLATE 1
#include "panel_experiment.hip"

// ------------------------------------------------------------------------------------------------ t1
struct BigJob { const float* p[20]; int v[12]; int nblk_c, blk_base; };      // ~216 B, like FwdGateJob
struct BigJobs { int n; float fb; BigJob j[8]; };                            // ~1.7 KB by value
struct PtrJobs { const BigJobs* tab; int d; };

__global__ void k_empty(int* sink) { if (sink == (int*)1) *sink = 0; }
template <int NT>
__global__ __launch_bounds__(NT) void k_empty_nt(int* sink) { if (sink == (int*)1) *sink = 0; }

// the product kernels' prologue: search the by-value job list, then dereference the job
template <int NT>
__global__ __launch_bounds__(NT) void k_big_byvalue(const BigJobs jobs, float* sink) {
  const int bid = blockIdx.x;
  int ji = 0;
#pragma unroll
  for (int q = 1; q < 8; ++q)
    if (q < jobs.n && bid >= jobs.j[q].blk_base) ji = q;
  const BigJob& J = jobs.j[ji];
  if (J.v[0] == 12345) sink[bid] = J.p[0][threadIdx.x];
}
// the same through a device-resident table
template <int NT>
__global__ __launch_bounds__(NT) void k_big_table(const PtrJobs a, float* sink) {
  const BigJobs& jobs = a.tab[a.d];
  const int bid = blockIdx.x;
  int ji = 0;
#pragma unroll
  for (int q = 1; q < 8; ++q)
    if (q < jobs.n && bid >= jobs.j[q].blk_base) ji = q;
  const BigJob& J = jobs.j[ji];
  if (J.v[0] == 12345) sink[bid] = J.p[0][threadIdx.x];
}
// one dependent global load + store per thread (a minimal "real" body)
template <int NT>
__global__ __launch_bounds__(NT) void k_touch(const float* in, float* out) {
  const int i = blockIdx.x * NT + threadIdx.x;
  out[i] = in[i] + 1.f;
}



#include <functional>
static float time_loop(hipStream_t s, int iters, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) f();
  CK(hipStreamSynchronize(s));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return best * 1000.f / iters;
}

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

static void t1(hipStream_t s) {
  printf("== t1: dependent launch floor (us per launch, back-to-back on one stream, best of 3 x 2000)\n");
  float* buf; CK(hipMalloc(&buf, 1 << 22)); CK(hipMemset(buf, 0, 1 << 22));
  float* buf2; CK(hipMalloc(&buf2, 1 << 22));
  BigJobs bj{}; bj.n = 7;
  for (int i = 0; i < 8; ++i) { bj.j[i].blk_base = i * 40; for (auto& p : bj.j[i].p) p = buf; }
  BigJobs* tab; CK(hipMalloc(&tab, sizeof(BigJobs) * 4)); CK(hipMemcpy(tab, &bj, sizeof(bj), hipMemcpyHostToDevice));
  PtrJobs pj{tab, 0};
  const int N = 2000;
  for (int grid : {64, 256, 304, 608}) {
    float a = time_loop(s, N, [&] { hipLaunchKernelGGL(k_empty_nt<256>, dim3(grid), dim3(256), 0, s, (int*)nullptr); });
    float b = time_loop(s, N, [&] { hipLaunchKernelGGL(k_empty_nt<512>, dim3(grid), dim3(512), 0, s, (int*)nullptr); });
    float c = time_loop(s, N, [&] { hipLaunchKernelGGL(k_big_byvalue<512>, dim3(grid), dim3(512), 0, s, bj, buf2); });
    float d = time_loop(s, N, [&] { hipLaunchKernelGGL(k_big_table<512>, dim3(grid), dim3(512), 0, s, pj, buf2); });
    float e = time_loop(s, N, [&] { hipLaunchKernelGGL(k_touch<512>, dim3(grid), dim3(512), 0, s, buf, buf2); });
    float e2 = time_loop(s, N, [&] { hipLaunchKernelGGL(k_touch<256>, dim3(grid), dim3(256), 0, s, buf, buf2); });
    float f = time_loop(s, N, [&] { hipLaunchKernelGGL(k_big_byvalue<512>, dim3(grid), dim3(512), 84 * 1024, s, bj, buf2); });
    printf("grid %4d: empty/256thr %.2f  empty/512thr %.2f  by-value-jobs/512 %.2f  table-jobs/512 %.2f  touch/512 %.2f  touch/256 %.2f  by-value+84KB-LDS %.2f\n",
           grid, a, b, c, d, e, e2, f);
  }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big_byvalue<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  {
    float g1 = time_graph(s, 200, 10, [&] { hipLaunchKernelGGL(k_empty_nt<512>, dim3(304), dim3(512), 0, s, (int*)nullptr); });
    float g2 = time_graph(s, 200, 10, [&] { hipLaunchKernelGGL(k_big_byvalue<512>, dim3(304), dim3(512), 0, s, bj, buf2); });
    float g3 = time_graph(s, 200, 10, [&] { hipLaunchKernelGGL(k_touch<512>, dim3(304), dim3(512), 0, s, buf, buf2); });
    printf("hipGraph (200 nodes, 10 replays) grid 304: empty/512 %.2f  by-value-jobs/512 %.2f  touch/512 %.2f\n", g1, g2, g3);
  }
  // alternate two different kernels (as gates/proj alternate)
  {
    float x = time_loop(s, N, [&] {
      hipLaunchKernelGGL(k_big_byvalue<512>, dim3(304), dim3(512), 0, s, bj, buf2);
      hipLaunchKernelGGL(k_big_byvalue<256>, dim3(152), dim3(256), 0, s, bj, buf2);
    });
    printf("alternating by-value 512thr x304 / 256thr x152: %.2f us per pair\n", x);
  }
  CK(hipFree(buf)); CK(hipFree(buf2)); CK(hipFree(tab));
}

// ------------------------------------------------------------------------------------------------ t2
// Block b pulls a contiguous slice of W; slices are laid out so that blocks with equal b % 8 (same XCD under
// round-robin dispatch) own one contiguous eighth of W.  PASSES re-reads inside the launch.
template <int LOADS, bool SC1>
__global__ __launch_bounds__(512) void k_pull(const float* W, size_t slice_floats, int nslices, int passes, float* sink) {
  const int b = blockIdx.x % nslices;                    // blocks beyond nslices share slices (second row block)
  const int x = b & 7, i = b >> 3;
  const size_t s = (size_t)x * (nslices / 8) + i;
  const float* base = W + s * slice_floats;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(slice_floats * 4), 0x00020000);
  float acc = 0.f;
  const int per_sweep = 512 * 4 * LOADS;
  for (int p = 0; p < passes; ++p) {
    for (size_t o = 0; o < slice_floats; o += per_sweep) {
      u32x4 v[LOADS];
#pragma unroll
      for (int l = 0; l < LOADS; ++l) {
        const unsigned off = (unsigned)((o + (size_t)l * 2048 + threadIdx.x * 4) * 4);
        if (SC1) v[l] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);      // out-of-range -> 0
        else v[l] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      }
#pragma unroll
      for (int l = 0; l < LOADS; ++l) acc += __uint_as_float(v[l].x) + __uint_as_float(v[l].w);
    }
    asm volatile("" : "+v"(acc));
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}
__global__ void k_flush(float* p, size_t n) {       // streams a large buffer through the L2s / MALL
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = p[i] * 0.5f + 1.f;
}

static void t2(hipStream_t s) {
  printf("== t2: weight pull per launch: cold vs back-to-back vs in-kernel re-read (256 blocks x 512 thr)\n");
  const size_t total = (size_t)20 << 20;                 // 20 MiB of "weights" (G forward set is 20.4 MB)
  float* W; CK(hipMalloc(&W, total)); CK(hipMemset(W, 0, total));
  float* sink; CK(hipMalloc(&sink, 4096 * 4));
  const size_t flush_n = (size_t)160 << 20;              // 640 MB > MALL
  float* F; CK(hipMalloc(&F, flush_n * 4)); CK(hipMemset(F, 0, flush_n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nblk : {256, 512}) {
    const int nslices = 256;
    const size_t slice = total / 4 / nslices;            // floats: 80 KB per slice
    auto launch = [&](int passes, bool sc1) {
      if (sc1) hipLaunchKernelGGL((k_pull<10, true>), dim3(nblk), dim3(512), 0, s, W, slice, nslices, passes, sink);
      else hipLaunchKernelGGL((k_pull<10, false>), dim3(nblk), dim3(512), 0, s, W, slice, nslices, passes, sink);
    };
    // cold: flush between launches, time only the pull
    float cold = 0.f;
    for (int r = 0; r < 5; ++r) {
      hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, s, F, flush_n);
      CK(hipEventRecord(e0, s)); launch(1, false); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); cold += ms * 1000.f / 5;
    }
    // MALL-warm, L2 state unknown: back-to-back launches
    float b2b = time_loop(s, 500, [&] { launch(1, false); });
    float b2b_sc1 = time_loop(s, 500, [&] { launch(1, true); });
    // in-kernel re-reads
    float p1 = time_loop(s, 200, [&] { launch(1, false); });
    float p9 = time_loop(s, 200, [&] { launch(9, false); });
    float q9 = time_loop(s, 200, [&] { launch(9, true); });
    printf("blocks %d (slice 80 KB, %s): cold(HBM) %.2f us | back-to-back %.2f | b2b sc1-loads %.2f | in-kernel extra pass %.2f (plain) %.2f (sc1)"
           "  => per-CU B/clk@2.4GHz: b2b %.1f, in-kernel %.1f\n",
           nblk, nblk == 256 ? "each line one reader" : "two readers per line", cold, b2b, b2b_sc1, (p9 - p1) / 8, (q9 - p1) / 8,
           (nblk / 256.0) * 81920.0 / (b2b * 2400.0), (nblk / 256.0) * 81920.0 / ((p9 - p1) / 8 * 2400.0));
  }
  // with a dirty-data writer in between (like the stash writes of the gates kernel): does it evict?
  {
    const int nslices = 256; const size_t slice = total / 4 / nslices;
    float both = time_loop(s, 300, [&] {
      hipLaunchKernelGGL((k_pull<10, false>), dim3(256), dim3(512), 0, s, W, slice, nslices, 1, sink);
      hipLaunchKernelGGL(k_flush, dim3(256), dim3(256), 0, s, F, (size_t)1 << 20);       // 4 MB read+write
    });
    float fl = time_loop(s, 300, [&] { hipLaunchKernelGGL(k_flush, dim3(256), dim3(256), 0, s, F, (size_t)1 << 20); });
    printf("pull + 4MB rw kernel alternating: %.2f us per pair (rw kernel alone %.2f)\n", both, fl);
  }
  CK(hipFree(W)); CK(hipFree(sink)); CK(hipFree(F));
}

// ------------------------------------------------------------------------------------------------ t3
constexpr unsigned SPIN_LIMIT = 1u << 21;
struct BarState {                       // every word on its own 128-B line
  unsigned grp[8][32];                  // arrivals per group (block id % 8)
  unsigned top[32];                     // groups complete
  unsigned gen[8][32];                  // release generation per group
  unsigned err[32];
};
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// arrive: one lane per workgroup.  Payload published before must be visible: either sc1 (write-through) stores drained by
// s_waitcnt vmcnt(0) in every storing wave + __syncthreads (release_fence = false), or a release fence here.
__device__ __forceinline__ void bar_arrive(BarState* B, unsigned it, unsigned n_in_group, bool release_fence) {
  const unsigned g = blockIdx.x & 7;
  if (release_fence) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  const unsigned old = __hip_atomic_fetch_add(&B->grp[g][0], 1u, RLX_AGENT);
  if (old + 1 == n_in_group * (it + 1)) {
    const unsigned o2 = __hip_atomic_fetch_add(&B->top[0], 1u, RLX_AGENT);
    if (o2 + 1 == 8 * (it + 1)) {
#pragma unroll
      for (int x = 0; x < 8; ++x) __hip_atomic_store(&B->gen[x][0], it + 1, RLX_AGENT);
    }
  }
}
__device__ __forceinline__ bool bar_wait(BarState* B, unsigned it, bool acquire_fence) {
  const unsigned g = blockIdx.x & 7;
  unsigned spins = 0;
  while (__hip_atomic_load(&B->gen[g][0], RLX_AGENT) < it + 1) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_LIMIT) { __hip_atomic_store(&B->err[0], 1u, RLX_AGENT); return false; }
  }
  if (acquire_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

// mode bits: 1 = acquire fence after the wait, 2 = release fence before the arrive, 4 = publish `wr` floats per WG with
// sc1 stores (else plain stores), 8 = read `rd` floats of the shared vector after the barrier, 16 = read with sc1 loads,
// 32 = check every word read
__global__ __launch_bounds__(512) void k_bar(BarState* B, float* shared, int iters, int mode, int wr, int rd, unsigned* bad, float* sink) {
  __shared__ int ok_s;
  const unsigned n_in_group = (gridDim.x + 7 - (blockIdx.x & 7)) / 8;
  float acc = 0.f;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    float* buf = shared + (size_t)(it & 1) * rd;
    if (wr) {
      // the shared vector of iteration `it`: word i holds (it+1)*1e-3 + i ; every WG writes its own range
      const size_t w0 = (size_t)blockIdx.x * wr;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, rd * 4, 0x00020000);
      for (int i = threadIdx.x * 4; i < wr; i += 512 * 4) {
        const size_t o = (w0 + i) % rd;
        u32x4 v;
        v.x = __float_as_uint((float)(it + 1) + (float)(o + 0) * 1e-6f); v.y = __float_as_uint((float)(it + 1) + (float)(o + 1) * 1e-6f);
        v.z = __float_as_uint((float)(it + 1) + (float)(o + 2) * 1e-6f); v.w = __float_as_uint((float)(it + 1) + (float)(o + 3) * 1e-6f);
        if (mode & 4) __builtin_amdgcn_raw_buffer_store_b128(v, r, (unsigned)(o * 4), 0, 16);
        else __builtin_amdgcn_raw_buffer_store_b128(v, r, (unsigned)(o * 4), 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      bar_arrive(B, (unsigned)it, n_in_group, (mode & 2) != 0);
      ok_s = bar_wait(B, (unsigned)it, (mode & 1) != 0) ? 1 : 0;
    }
    __syncthreads();
    if (!ok_s) break;
    if (mode & 8) {
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, rd * 4, 0x00020000);
      const int covered = min(rd, (int)gridDim.x * wr);       // words actually written this iteration
      for (int i0 = 0; i0 < rd; i0 += 512 * 4 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int l = 0; l < 8; ++l)
          v[l] = (mode & 16) ? __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)((i0 + l * 2048 + threadIdx.x * 4) * 4), 0, 16)
                             : __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)((i0 + l * 2048 + threadIdx.x * 4) * 4), 0, 0);
#pragma unroll
        for (int l = 0; l < 8; ++l) {
          const int i = i0 + l * 2048 + threadIdx.x * 4;
          acc += __uint_as_float(v[l].x) + __uint_as_float(v[l].w);
          if ((mode & 32) && i + 3 < covered) {
            if (__uint_as_float(v[l].x) != (float)(it + 1) + (float)(i + 0) * 1e-6f) ++nbad;
            if (__uint_as_float(v[l].w) != (float)(it + 1) + (float)(i + 3) * 1e-6f) ++nbad;
          }
        }
      }
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

static void t3(hipStream_t s) {
  printf("== t3: XCD-hierarchical grid barrier, 512-thread WGs, us per iteration (2000 iterations, best of 3)\n");
  BarState* B; CK(hipMalloc(&B, sizeof(BarState)));
  const int RD = 64 * 560;          // one layer's [x_t | m_{t-1}] for 64 rows = 143 KB
  float* shared; CK(hipMalloc(&shared, (size_t)2 * RD * 3 * 4)); CK(hipMemset(shared, 0, (size_t)2 * RD * 3 * 4));
  unsigned* bad; CK(hipMalloc(&bad, 4));
  float* sink; CK(hipMalloc(&sink, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { const char* name; int mode, wr, rd; };
  const V vs[] = {
      {"counter only (no fences)", 0, 0, 0},
      {"+ acquire fence", 1, 0, 0},
      {"+ release + acquire fences", 3, 0, 0},
      {"plain 560-float publish, release+acquire, read 143KB plain, checked", 1 | 2 | 8 | 32, 560, RD},
      {"sc1 560-float publish, no release, acquire, read 143KB plain, checked", 1 | 4 | 8 | 32, 560, RD},
      {"sc1 560-float publish, no fences, read 143KB sc1, checked", 4 | 8 | 16 | 32, 560, RD},
      {"sc1 publish 3.5K floats/WG (14KB), no fences, read 143KB sc1", 4 | 8 | 16, 3584, RD},
      {"plain publish 3.5K floats/WG (14KB), rel+acq, read 143KB plain", 1 | 2 | 8, 3584, RD},
      {"sc1 560-float publish, no fences, read 430KB sc1, checked", 4 | 8 | 16 | 32, 1680, 3 * RD},
  };
  for (int nwg : {256, 240}) {
    for (const V& v : vs) {
      float best = 1e30f; unsigned hbad = 0, herr = 0;
      const int iters = 2000;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(B, 0, sizeof(BarState), s)); CK(hipMemsetAsync(bad, 0, 4, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_bar, dim3(nwg), dim3(512), 0, s, B, shared, iters, v.mode, v.wr, v.rd, bad, sink);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); hbad += hb;
        BarState hB; CK(hipMemcpy(&hB, B, sizeof(hB), hipMemcpyDeviceToHost)); herr |= hB.err[0];
        if (herr) break;
      }
      printf("nwg %d  %-72s %7.2f us/iter  bad_words=%u timeout=%u\n", nwg, v.name, best * 1000.f / iters, hbad, herr);
      if (herr) { printf("barrier timed out -- stopping t3\n"); return; }
    }
  }
  CK(hipFree(B)); CK(hipFree(shared)); CK(hipFree(bad)); CK(hipFree(sink));
}

// ------------------------------------------------------------------------------------------------ t4
// the panel gates kernel (rsrgan_amd/csrc/panel.hip) on one generator diagonal of BASELINE's size, parts switched off
static void t4(hipStream_t s) {
  using namespace rsr;
  printf("== t4: k_pn_gates on one diagonal (3 generator layers N=64 H=760 K=560|280 + 4 discriminator jobs), us per launch in a hipGraph\n");
  std::vector<void*> bufs;
  auto dal = [&](size_t n, float v) { float* p; CK(hipMalloc(&p, n * 4)); std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = v * (float)((i * 2654435761u >> 20) & 255) / 256.f;
                                       CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); bufs.push_back(p); return p; };
  int* len; CK(hipMalloc(&len, 128 * 4)); { std::vector<int> h(128, 1000); CK(hipMemcpy(len, h.data(), 128 * 4, hipMemcpyHostToDevice)); }
  float* zeros; CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
  auto mk = [&](FwdGateJob& a, int N, int H, int ldx, int ldm, bool zx) {
    a = FwdGateJob{};
    a.x = zx ? nullptr : dal((size_t)N * ldx, 1.f); a.KxT = dal((size_t)4 * H * ldx, 0.05f); a.ldx = ldx;
    a.m = dal((size_t)N * ldm, 1.f); a.KhT = dal((size_t)4 * H * ldm, 0.05f); a.ldm = ldm;
    a.zx = zx ? dal((size_t)N * 4 * H, 0.1f) : nullptr; a.bias = dal(4 * H, 0.1f);
    a.wf = dal(H, 0.1f); a.wi = dal(H, 0.1f); a.wo = dal(H, 0.1f);
    a.c_prev = dal((size_t)N * H, 0.5f); a.c_out = dal((size_t)N * H, 0.f); a.gates = dal((size_t)N * 4 * H, 0.f);
    a.h = dal((size_t)N * ((H + 3) & ~3), 0.f); a.ldh = (H + 3) & ~3; a.len = len; a.t = 0; a.N = N; a.H = H;
  };
  auto build = [&](FwdGateJobs& gj, int ng, int nd, int N) {
    gj = FwdGateJobs{}; gj.forget_bias = 1.f;
    int base = 0;
    for (int l = ng - 1; l >= 0; --l) { FwdGateJob& a = gj.j[gj.n++]; mk(a, N, 760, 280, 280, l == 0); a.blk_base = base; base += pn_gates_blocks(760, N); }
    for (int l = 0; l < nd; ++l) { FwdGateJob& a = gj.j[gj.n++]; mk(a, N, 256, 40, 40, false); a.blk_base = base; base += pn_gates_blocks(256, N); }
    return base;
  };
  struct Cfg { const char* name; int ng, nd, N; };
  for (int tile : {34, 13, 12})
  for (const Cfg& c : {Cfg{"3 G layers + 4 D jobs", 3, 4, 64}, Cfg{"3 G layers", 3, 0, 64}, Cfg{"4 D jobs only", 0, 4, 64}}) {
    pn_set_gates_cfg(tile / 10, tile % 10);
    printf("-- tile config (nct,s) = (%d,%d)\n", tile / 10, tile % 10);
    FwdGateJobs gj; const int blocks = build(gj, c.ng, c.nd, c.N);
    for (int ab : {0, 1, 2, 1 | 4, 1 | 2 | 4, 8 | 16, 1 | 2 | 4 | 8 | 16, 128 | 16}) {
      CK(hipMemcpyToSymbol(HIP_SYMBOL(rsr::g_pn_ablate), &ab, sizeof(int)));
      launch_pn_gates(gj, blocks, s); CK(hipStreamSynchronize(s));
      float us = time_graph(s, 50, 6, [&] { launch_pn_gates(gj, blocks, s); });
      printf("%-24s blocks %3d  ablate %3d (%s%s%s%s%s%s%s%s) : %6.2f us\n", c.name, blocks, ab, ab & 1 ? "noMFMA " : "", ab & 2 ? "noDMA " : "", ab & 4 ? "noLDSread " : "",
             ab & 8 ? "noEpilogue " : "", ab & 16 ? "noPrefetch " : "", ab & 32 ? "return-at-entry " : "", ab & 64 ? "return-after-job-lookup " : "",
             ab & 128 ? "return-before-product " : "", us);
      if (ab >= 256) printf("      (256 = no k-block guards, 512 = unchecked DMA sources, 1024 = broadcast fragment reads)\n");
    }
    {   // the same grid with a small LDS request / fewer threads, returning at entry: what does dispatching the workgroups cost?
      int ab = 32; CK(hipMemcpyToSymbol(HIP_SYMBOL(rsr::g_pn_ablate), &ab, sizeof(int)));
      for (int lds : {0, 64 * 1024, 147456}) {
        float us = time_graph(s, 50, 6, [&] { hipLaunchKernelGGL((k_pn_gates<3, 4>), dim3(blocks), dim3(768), lds, s, gj); });
        printf("%-24s blocks %3d  return-at-entry, 768 threads, dynamic LDS %6d B : %6.2f us\n", c.name, blocks, lds, us);
      }
    }
  }
  int z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(rsr::g_pn_ablate), &z, sizeof(int)));
  for (void* p : bufs) CK(hipFree(p));
}

// ------------------------------------------------------------------------------------------------ t5
// what does one s_barrier cost in a 4 / 8 / 12-wave workgroup, alone and with a little VALU work between barriers?
template <int NT>
__global__ __launch_bounds__(NT) void k_barloop(int iters, int work, float* sink) {
  float x = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int j = 0; j < work; ++j) x = x * 1.0001f + 0.5f;
    asm volatile("" : "+v"(x));
    __builtin_amdgcn_s_barrier();
  }
  if (x == 123.456f) sink[blockIdx.x] = x;
}
static void t5(hipStream_t s) {
  printf("== t5: s_barrier cost: us per launch (192 workgroups, hipGraph), iters barriers with `work` dependent FMAs in between\n");
  float* sink; CK(hipMalloc(&sink, 4096 * 4));
  for (int work : {0, 32, 128}) {
    for (int iters : {0, 10, 100}) {
      float a = time_graph(s, 50, 6, [&] { hipLaunchKernelGGL(k_barloop<256>, dim3(192), dim3(256), 0, s, iters, work, sink); });
      float b = time_graph(s, 50, 6, [&] { hipLaunchKernelGGL(k_barloop<512>, dim3(192), dim3(512), 0, s, iters, work, sink); });
      float c = time_graph(s, 50, 6, [&] { hipLaunchKernelGGL(k_barloop<768>, dim3(192), dim3(768), 0, s, iters, work, sink); });
      printf("work %3d iters %3d : 256 thr %6.2f  512 thr %6.2f  768 thr %6.2f us\n", work, iters, a, b, c);
    }
  }
  CK(hipFree(sink));
}

// ------------------------------------------------------------------------------------------------ t6
// the persistent discriminator recurrence (rsrgan_amd/csrc/dlstm.hip) at BASELINE's size, with per-phase shader-cycle counters
static void t6(hipStream_t s) {
  using namespace rsr;
  const int N = 64, T = 100, H = 256, P = 40, I = 40, L = 2;
  printf("== t6: k_dl_fwd, N=%d rows, %d layers LSTMP(%d, proj %d), T=%d (profiled workgroup: block %d)\n", N, L, H, P, T, DL_PROF_BLOCK);
  auto dal = [&](size_t n, float v) { float* p; CK(hipMalloc(&p, n * 4)); std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = v * ((float)((i * 2654435761u >> 20) & 255) / 128.f - 1.f);
                                       CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p; };
  DlFwdArgs a{};
  a.L = L; a.N = N; a.Ns = N; a.T = T; a.forget_bias = 1.f;
  int* len; CK(hipMalloc(&len, N * 4)); { std::vector<int> h(N, T); CK(hipMemcpy(len, h.data(), N * 4, hipMemcpyHostToDevice)); }
  a.len = len;
  unsigned* flags; CK(hipMalloc(&flags, (DL_MAXL * 64 + 64) * 4)); a.flags = flags; a.err = flags + DL_MAXL * 64;
  CK(hipMalloc(&a.dump, 512 * 4));
  float* prev_out = nullptr;
  for (int l = 0; l < L; ++l) {
    DlLayer& y = a.layer[l];
    y.I = I; y.H = H; y.P = P; y.ldI = 40; y.ldP = 40; y.ldH = 256;
    y.in = l == 0 ? dal((size_t)T * N * 40, 1.f) : prev_out;
    y.KxT = dal((size_t)4 * H * 40, 0.1f); y.KhT = dal((size_t)4 * H * 40, 0.1f); y.WpT = dal((size_t)P * 256, 0.1f);
    y.bias = dal(4 * H, 0.1f); y.wf = dal(H, 0.1f); y.wi = dal(H, 0.1f); y.wo = dal(H, 0.1f);
    y.gates = dal((size_t)T * N * 4 * H, 0.f); y.c = dal((size_t)(T + 1) * N * H, 0.f); y.h = dal((size_t)T * N * 256, 0.f);
    y.mst = dal((size_t)(T + 1) * N * 40, 0.f); y.out = dal((size_t)T * N * 40, 0.f);
    prev_out = y.out;
  }
  if (!dl_fwd_supported(a)) { printf("not supported\n"); return; }
  for (int i = 0; i < 3; ++i) launch_dl_fwd(a, s);
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < 10; ++i) launch_dl_fwd(a, s);
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long prof[16]; CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(rsr::g_dl_prof), sizeof(prof)));
  unsigned err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
  printf("k_dl_fwd: %.1f us per launch = %.2f us per step (err flag %u)\n", ms * 100.f, ms * 100.f / T, err);
  const char* nm[8] = {"x wait + issue", "gates MFMA", "cell update + stash stores", "projection MFMA + partial store", "barrier 1", "reduce + m_t + x commit", "drain + barrier 2 + flag", "loop top"};
  unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += prof[i];
  for (int i = 0; i < 8; ++i) printf("  phase %-34s %8.0f cycles per step (%4.1f %%)\n", nm[i], (double)prof[i] / T, 100.0 * prof[i] / (double)tot);
  printf("  total %.0f cycles per step\n", (double)tot / T);
}

// ------------------------------------------------------------------------------------------------ t7
// round 1's backward phase A kernel on one generator diagonal (3 layers, N=64, H=760, P=280), parts switched off
static void t7(hipStream_t s) {
  using namespace rsr;
  printf("== t7: k_bwd_a<8> on one diagonal (3 generator layers N=64 H=760 P=280 [+ 2 discriminator jobs]), us per launch in a hipGraph\n");
  auto dal = [&](size_t n, float v) { float* p; CK(hipMalloc(&p, n * 4)); std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = v * ((float)((i * 2654435761u >> 20) & 255) / 256.f);
                                       CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p; };
  int* len; CK(hipMalloc(&len, 128 * 4)); { std::vector<int> h(128, 1000); CK(hipMemcpy(len, h.data(), 128 * 4, hipMemcpyHostToDevice)); }
  // a stash big enough to be cold in every cache: 40 time steps per layer, one launch reads a different step each time
  const int NT = 40;
  auto mk = [&](BwdAJob& a, int N, int H, int P, int tstep, float* gates, float* cbuf, float* dcb, float* dm, float* dout, float* dmt, float* Wp, float* pe) {
    a = BwdAJob{};
    a.dout = dout; a.dmst = dm; a.Wp = Wp; a.dmt = dmt;
    a.gates = gates + (size_t)tstep * N * 4 * H; a.c_prev = cbuf + (size_t)tstep * N * H; a.c_cur = cbuf + (size_t)(tstep + 1) * N * H;
    a.wf = pe; a.wi = pe + H; a.wo = pe + 2 * H; a.dc = dcb; a.len = len; a.ldm = (P + 3) & ~3; a.P = P; a.t = 0; a.N = N; a.H = H;
    a.nblk_c = (H + 15) / 16;
  };
  struct L { float *gates, *c, *dc, *dm, *dout, *dmt, *Wp, *pe; int N, H, P; };
  std::vector<L> Ls;
  for (int i = 0; i < 5; ++i) {
    L l; l.N = 64; l.H = i < 3 ? 760 : 256; l.P = i < 3 ? 280 : 40;
    l.gates = dal((size_t)NT * l.N * 4 * l.H, 0.5f); l.c = dal((size_t)(NT + 1) * l.N * l.H, 0.5f); l.dc = dal((size_t)l.N * l.H, 0.1f);
    l.dm = dal((size_t)l.N * l.P, 0.1f); l.dout = dal((size_t)l.N * l.P, 0.1f); l.dmt = dal((size_t)l.N * l.P, 0.f);
    l.Wp = dal((size_t)l.H * l.P, 0.05f); l.pe = dal(3 * l.H, 0.1f);
    Ls.push_back(l);
  }
  for (int nd : {2, 0}) {
    for (int ab : {0, 1, 2, 4, 8, 16, 4 | 8, 2 | 4 | 8 | 16, 1 | 2 | 4 | 8 | 16, 32}) {
      CK(hipMemcpyToSymbol(HIP_SYMBOL(rsr::g_ka_ablate), &ab, sizeof(int)));
      int tstep = 0;
      auto launch = [&] {
        BwdAJobs aj{}; int base = 0;
        for (int i = 0; i < 3 + nd; ++i) { BwdAJob& a = aj.j[aj.n++]; const L& l = Ls[i]; mk(a, l.N, l.H, l.P, tstep, l.gates, l.c, l.dc, l.dm, l.dout, l.dmt, l.Wp, l.pe);
                                             a.blk_base = base; base += job_blocks(a.nblk_c, l.N); }
        launch_bwd_a(aj, base, 0, s);
        tstep = (tstep + 7) % NT;
      };
      launch(); CK(hipStreamSynchronize(s));
      float us = time_graph(s, 40, 6, launch);
      printf("3 G layers + %d D jobs  ablate %2d (%s%s%s%s%s%s) : %6.2f us\n", nd, ab, ab & 1 ? "noMFMA " : "", ab & 2 ? "noOperandLoads " : "", ab & 4 ? "noEpilogueLoads " : "",
             ab & 8 ? "noEpilogueStores " : "", ab & 16 ? "noDmtStore " : "", ab & 32 ? "return-at-entry" : "", us);
    }
  }
  int z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(rsr::g_ka_ablate), &z, sizeof(int)));
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  if (!strcmp(what, "t1") || !strcmp(what, "all")) t1(s);
  if (!strcmp(what, "t2") || !strcmp(what, "all")) t2(s);
  if (!strcmp(what, "t3") || !strcmp(what, "all")) t3(s);
  if (!strcmp(what, "t4") || !strcmp(what, "all")) t4(s);
  if (!strcmp(what, "t5") || !strcmp(what, "all")) t5(s);
  if (!strcmp(what, "t6") || !strcmp(what, "all")) t6(s);
  if (!strcmp(what, "t7") || !strcmp(what, "all")) t7(s);
  CK(hipStreamDestroy(s));
  return 0;
}
