// persist.hip -- grid-barrier micro-benchmarks (rsrgan_microbench kind 4): what does one device-wide barrier cost on
// MI355X when every CU holds one resident workgroup, and what does a barrier + broadcast read of the step's activations
// cost?  Decides between launch-per-phase (kernels.hip) and a weights-stationary persistent recurrence.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace rsr {

constexpr unsigned SPIN_LIMIT = 1u << 22;     // bail out instead of hanging the GPU

// flat barrier: one monotonically increasing counter in device memory
__device__ __forceinline__ bool grid_barrier_flat(unsigned* ctr, unsigned target, int* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);            // agent scope by default
    unsigned spins = 0;
    while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

// two-level barrier: 8 per-XCD counters (workgroup b sits on XCD b % 8), the last arriver of each XCD bumps the top counter
__device__ __forceinline__ bool grid_barrier_2lvl(unsigned* ctrs, unsigned iter, unsigned per_xcd, int* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7;
    unsigned* mine = ctrs + 32 * (1 + x);                      // 128-B apart
    const unsigned prev = __atomic_fetch_add(mine, 1u, __ATOMIC_ACQ_REL);
    if (prev + 1 == (iter + 1) * per_xcd) __atomic_fetch_add(ctrs, 1u, __ATOMIC_RELEASE);
    unsigned spins = 0;
    while (__atomic_load_n(ctrs, __ATOMIC_ACQUIRE) < (iter + 1) * 8) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

// variant 0: flat barrier only; 1: two-level barrier only; 2/3: flat / two-level + exchange: every workgroup writes `wr_floats`
// of the shared vector, then after the barrier reads all `rd_floats` of it (the [64 x 560] activations of a layer).
__global__ __launch_bounds__(512) void k_gridbar(unsigned* ctrs, float* buf, int iters, int variant, int wr_floats, int rd_floats,
                                                 float* sink, int* err) {
  float acc = 0.f;
  const int nwg = gridDim.x;
  for (int it = 0; it < iters; ++it) {
    if (variant >= 2) {
      float* dst = buf + (size_t)(it & 1) * rd_floats;
      for (int i = threadIdx.x; i < wr_floats; i += blockDim.x) {
        const int o = (blockIdx.x * wr_floats + i) % rd_floats;
        dst[o] = acc + (float)it;
      }
    }
    const bool ok = (variant & 1) ? grid_barrier_2lvl(ctrs, (unsigned)it, (unsigned)(nwg / 8), err)
                                  : grid_barrier_flat(ctrs, (unsigned)(it + 1) * nwg, err);
    if (!ok) break;
    if (variant >= 2) {
      const float4* src = reinterpret_cast<const float4*>(buf + (size_t)(it & 1) * rd_floats);
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = threadIdx.x; i < rd_floats / 4; i += blockDim.x) {
        const float4 v = src[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      acc = acc * 0.5f + 1e-9f * (s.x + s.y + s.z + s.w);
    }
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

// kind 5: flag exchange inside small groups (no global barrier).  `gsz` workgroups form a group; per iteration each writes
// `wr_floats` to its slot, release-stores its flag, acquire-spins on the flags of the other members, then reads all slots.
// variant 0: a group = consecutive block ids (its members sit on different XCDs); variant 1: a group = ids with equal
// id % 8 (same XCD under round-robin dispatch).  Two slot sets alternate so a fast member cannot overwrite unread data.
__global__ __launch_bounds__(512) void k_flagx(unsigned* flags, float* slots, int iters, int variant, int gsz, int wr_floats,
                                               float* sink, int* err) {
  const int b = blockIdx.x;
  int grp, mem;
  if (variant == 0) { grp = b / gsz; mem = b % gsz; }
  else { const int x = b & 7, i = b >> 3; grp = x * ((gridDim.x / 8 + gsz - 1) / gsz) + i / gsz; mem = i % gsz; }   // needs gridDim.x >= 8 * gsz
  unsigned* gflags = flags + (size_t)grp * gsz * 32;            // one 128-B line per flag
  float* gslots = slots + (size_t)grp * gsz * 2 * wr_floats;
  float acc = 0.f;
  __shared__ int bail;
  if (threadIdx.x == 0) bail = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    float* mine = gslots + ((size_t)(it & 1) * gsz + mem) * wr_floats;
    for (int i = threadIdx.x; i < wr_floats; i += blockDim.x) mine[i] = acc + (float)(it + mem);
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_store_n(gflags + mem * 32, (unsigned)(it + 1), __ATOMIC_RELEASE);
    }
    if ((int)threadIdx.x < gsz && (int)threadIdx.x != mem) {
      unsigned spins = 0;
      while (__atomic_load_n(gflags + threadIdx.x * 32, __ATOMIC_ACQUIRE) < (unsigned)(it + 1)) {
        if (++spins > SPIN_LIMIT) { *err = 1; bail = 1; break; }
      }
    }
    __syncthreads();
    if (bail) break;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const float* all = gslots + (size_t)(it & 1) * gsz * wr_floats;
    float s = 0.f;
    for (int i = threadIdx.x; i < gsz * wr_floats; i += blockDim.x) s += all[i];
    acc = acc * 0.5f + 1e-6f * s;
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

int flagx_microbench(int variant, int nwg, int iters, int gsz, int wr_floats, float* out_us) {
  unsigned* flags = nullptr; float* slots = nullptr; float* sink = nullptr; int* err = nullptr;
  if (hipMalloc((void**)&flags, (size_t)nwg * 32 * sizeof(unsigned)) != hipSuccess) return -1;
  if (hipMalloc((void**)&slots, (size_t)nwg * 2 * wr_floats * sizeof(float) + 64) != hipSuccess) return -1;
  if (hipMalloc((void**)&sink, nwg * sizeof(float)) != hipSuccess) return -1;
  if (hipMalloc((void**)&err, sizeof(int)) != hipSuccess) return -1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  int herr = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(flags, 0, (size_t)nwg * 32 * sizeof(unsigned));
    (void)hipMemset(slots, 0, (size_t)nwg * 2 * wr_floats * sizeof(float));
    (void)hipMemset(err, 0, sizeof(int));
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k_flagx, dim3(nwg), dim3(512), 0, nullptr, flags, slots, iters, variant, gsz, wr_floats, sink, err);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
    (void)hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
    if (herr) break;
  }
  *out_us = best * 1000.f / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(flags); (void)hipFree(slots); (void)hipFree(sink); (void)hipFree(err);
  return herr ? -2 : 0;
}

int gridbar_microbench(int variant, int nwg, int iters, int wr_floats, int rd_floats, float* out_us) {
  unsigned* ctrs = nullptr; float* buf = nullptr; float* sink = nullptr; int* err = nullptr;
  if (hipMalloc((void**)&ctrs, 32 * 9 * sizeof(unsigned)) != hipSuccess) return -1;
  if (hipMalloc((void**)&buf, (size_t)2 * rd_floats * sizeof(float) + 64) != hipSuccess) return -1;
  if (hipMalloc((void**)&sink, nwg * sizeof(float)) != hipSuccess) return -1;
  if (hipMalloc((void**)&err, sizeof(int)) != hipSuccess) return -1;
  (void)hipMemset(buf, 0, (size_t)2 * rd_floats * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  int herr = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(ctrs, 0, 32 * 9 * sizeof(unsigned));
    (void)hipMemset(err, 0, sizeof(int));
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k_gridbar, dim3(nwg), dim3(512), 0, nullptr, ctrs, buf, iters, variant, wr_floats, rd_floats, sink, err);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
    (void)hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
    if (herr) break;
  }
  *out_us = best * 1000.f / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(ctrs); (void)hipFree(buf); (void)hipFree(sink); (void)hipFree(err);
  return herr ? -2 : 0;
}

}  // namespace rsr
