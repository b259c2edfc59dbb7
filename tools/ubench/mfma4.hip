// Layout probe for v_mfma_f32_4x4x1_16B_f32 on gfx950: which lanes feed which block, what cbsz/abid broadcast, where D lands.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma4.hip -o tools/ubench/mfma4 && tools/ubench/mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

int main() {
  std::vector<float> a(64), b(64), d(256);
  for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 100.f * (1 + l); }
  float *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
    if (mode == 1) hipLaunchKernelGGL((k<4, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
    if (mode == 2) hipLaunchKernelGGL((k<4, 3>), dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    // hypothesis: D[lane = 4*blk + j][reg i] = A[src_blk*4 + i] * B[4*blk + j], src_blk = blk (mode 0) or ABID (cbsz = 4)
    int ok = 0;
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {
        const int blk = l / 4, sb = mode == 0 ? blk : (mode == 1 ? 0 : 3);
        ok += d[l * 4 + i] == a[sb * 4 + i] * b[l];
      }
    printf("mode %d: hypothesis D[lane][i] = A[srcblk*4+i]*B[lane] matches %d / 256\n", mode, ok);
    if (ok != 256)
      for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
  }
  return 0;
}
