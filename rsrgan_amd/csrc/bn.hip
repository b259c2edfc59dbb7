// bn.hip -- tf.contrib.layers.batch_norm(is_training, scale=True, renorm=True) as the frame-level generators and
// discriminator_dnn use it (models/dnn.py:56-61, models/discriminator_dnn.py:36-41, models/rced.py:67-72): TF 1.4's
// layers/normalization.py BatchNormalization with renorm, restated (see oracle/bn_renorm.py for the algorithm and the order of
// the update ops).  All of it is HBM-bound column statistics + elementwise work over [rows][cols] fp32 activations:
//   forward   k_bn_stats1 (row-slice partial sums, shifted by the first row) -> k_bn_stats2 (moments in double, the renorm
//             corrections r, d from the state, the affine y = z*a + b) -> k_bn_apply (y = relu(z*a + b))
//   backward  k_bn_bwd1 (partials of sum dy', sum dy'.xhat with dy' = dy.[y > 0]) -> k_bn_bwd2 (dbeta, dgamma, per-column
//             constants) -> k_bn_bwd3 (dz in place)
//   state     k_bn_commit (the UPDATE_OPS of one call, `times` times)
// Every reduction has a fixed order (slices, then a serial sum), so results do not depend on the launch geometry.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace rsr {

constexpr float BN_EPS = 1e-3f;            // contrib.layers.batch_norm epsilon
constexpr double BN_DECAY = 0.999;         // decay
constexpr double BN_RENORM_DECAY = 0.99;   // renorm_decay

// partial sums of one row slice: scratch[slice][2][cols] = sum (z - z0), sum (z - z0)^2 with z0 = the call's first row
__global__ __launch_bounds__(256) void k_bn_stats1(const float* __restrict__ z, int ld, int rows, int cols, int per,
                                                   float* __restrict__ scratch) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float s1 = 0.f, s2 = 0.f;
  if (c < cols) {
    const float z0 = z[c];
#pragma unroll 4
    for (int r = rbeg + rl; r < rend; r += 4) {
      const float v = z[(size_t)r * ld + c] - z0;
      s1 += v; s2 += v * v;
    }
  }
  red[0][rl][threadIdx.x & 63] = s1; red[1][rl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rl < 2 && c < cols) {
    const int l = threadIdx.x & 63;
    scratch[((size_t)blockIdx.y * 2 + rl) * cols + c] = ((red[rl][0][l] + red[rl][1][l]) + red[rl][2][l]) + red[rl][3][l];
  }
}

// moments, corrections and the affine of one call.  stat rows: 0 mean, 1 stddev, 2 r, 3 d, 4 a, 5 b  (y = z*a + b)
__global__ __launch_bounds__(256) void k_bn_stats2(const float* __restrict__ scratch, int slices, const float* __restrict__ z, int rows,
                                                   int cols, BnVars v, float* __restrict__ stat, int ldc) {
  // 64 columns x 4 slice lanes per workgroup: the (up to 512) slice partials of a column are summed by four threads with
  // several loads in flight each, then combined in a fixed order (one thread walking them was ~100 us: 512 dependent round trips)
  __shared__ double red2[2][4][64];
  const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
  {
    double p1 = 0.0, p2 = 0.0;
    if (c < cols) {
#pragma unroll 8
      for (int i = rl; i < slices; i += 4) { p1 += scratch[((size_t)i * 2) * cols + c]; p2 += scratch[((size_t)i * 2 + 1) * cols + c]; }
    }
    red2[0][rl][l] = p1; red2[1][rl][l] = p2;
  }
  __syncthreads();
  if (rl != 0 || c >= cols) return;
  const double s1 = ((red2[0][0][l] + red2[0][1][l]) + red2[0][2][l]) + red2[0][3][l];
  const double s2 = ((red2[1][0][l] + red2[1][1][l]) + red2[1][2][l]) + red2[1][3][l];
  const double m0 = s1 / rows;
  const double var = fmax(s2 / rows - m0 * m0, 0.0);
  const double mean = (double)z[c] + m0;
  const double sd = sqrt(var + (double)BN_EPS);
  const double mixed_mean = (double)v.rm[c] + (1.0 - (double)v.rmw[0]) * mean;
  const double mixed_sd = (double)v.rs[c] + (1.0 - (double)v.rsw[0]) * sd;
  const double r = sd / mixed_sd, d = (mean - mixed_mean) / mixed_sd;
  const double g = v.gamma[c];
  const double a = r * g / sd;
  stat[c] = (float)mean; stat[ldc + c] = (float)sd; stat[2 * ldc + c] = (float)r; stat[3 * ldc + c] = (float)d;
  stat[4 * ldc + c] = (float)a; stat[5 * ldc + c] = (float)(d * g + (double)v.beta[c] - mean * a);
}

// is_training=False: y = (z - moving_mean) / sqrt(moving_variance + eps) * gamma + beta
__global__ __launch_bounds__(256) void k_bn_infer_coef(int cols, BnVars v, float* __restrict__ stat, int ldc) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const double a = (double)v.gamma[c] / sqrt((double)v.mv[c] + (double)BN_EPS);
  stat[4 * ldc + c] = (float)a; stat[5 * ldc + c] = (float)((double)v.beta[c] - (double)v.mm[c] * a);
}

__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ z, int ld, float* __restrict__ y, int ldy, size_t rows, int cols,
                                                  const float* __restrict__ a, const float* __restrict__ b, int relu) {
  const int c4 = cols >> 2;                       // cols is padded to a multiple of 4 by the caller's leading dimensions
  const size_t n = rows * (size_t)c4;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    const float4 zv = *reinterpret_cast<const float4*>(z + r * ld + c);
    const float4 av = *reinterpret_cast<const float4*>(a + c), bv = *reinterpret_cast<const float4*>(b + c);
    float4 o = make_float4(zv.x * av.x + bv.x, zv.y * av.y + bv.y, zv.z * av.z + bv.z, zv.w * av.w + bv.w);
    if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    *reinterpret_cast<float4*>(y + r * ldy + c) = o;
  }
}

// backward partials: scratch[slice][2][cols] = sum dy', sum dy'.xhat ; dy' = dy where y > 0 (the ReLU that follows), else 0
__global__ __launch_bounds__(256) void k_bn_bwd1(const float* __restrict__ dy, int ldd, const float* __restrict__ y, int ldy,
                                                 const float* __restrict__ z, int ldz, const float* __restrict__ stat, int ldc, int rows,
                                                 int cols, int per, int relu, float* __restrict__ scratch) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float s1 = 0.f, s2 = 0.f;
  if (c < cols) {
    const float mean = stat[c], isd = 1.f / stat[ldc + c];
#pragma unroll 4
    for (int r = rbeg + rl; r < rend; r += 4) {
      float g = dy[(size_t)r * ldd + c];
      if (relu && !(y[(size_t)r * ldy + c] > 0.f)) g = 0.f;
      s1 += g; s2 += g * ((z[(size_t)r * ldz + c] - mean) * isd);
    }
  }
  red[0][rl][threadIdx.x & 63] = s1; red[1][rl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rl < 2 && c < cols) {
    const int l = threadIdx.x & 63;
    scratch[((size_t)blockIdx.y * 2 + rl) * cols + c] = ((red[rl][0][l] + red[rl][1][l]) + red[rl][2][l]) + red[rl][3][l];
  }
}

// sums -> dbeta, dgamma (assigned or accumulated) and the per-column constants of dz: sums[0] = s1/n, sums[1] = s2/n
__global__ __launch_bounds__(256) void k_bn_bwd2(const float* __restrict__ scratch, int slices, int rows, int cols,
                                                 const float* __restrict__ stat, int ldc, float* __restrict__ dbeta,
                                                 float* __restrict__ dgamma, int accumulate, float* __restrict__ sums) {
  __shared__ double red2[2][4][64];
  const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
  {
    double p1 = 0.0, p2 = 0.0;
    if (c < cols) {
#pragma unroll 8
      for (int i = rl; i < slices; i += 4) { p1 += scratch[((size_t)i * 2) * cols + c]; p2 += scratch[((size_t)i * 2 + 1) * cols + c]; }
    }
    red2[0][rl][l] = p1; red2[1][rl][l] = p2;
  }
  __syncthreads();
  if (rl != 0 || c >= cols) return;
  const double s1 = ((red2[0][0][l] + red2[0][1][l]) + red2[0][2][l]) + red2[0][3][l];
  const double s2 = ((red2[1][0][l] + red2[1][1][l]) + red2[1][2][l]) + red2[1][3][l];
  if (dbeta) {
    const double gb = s1, gg = (double)stat[2 * ldc + c] * s2 + (double)stat[3 * ldc + c] * s1;
    dbeta[c] = (float)(accumulate ? (double)dbeta[c] + gb : gb);
    dgamma[c] = (float)(accumulate ? (double)dgamma[c] + gg : gg);
  }
  sums[c] = (float)(s1 / rows); sums[ldc + c] = (float)(s2 / rows);
}

// dz = gamma*r/stddev * (dy' - s1/n - xhat*s2/n), in place over dy.  stat[4] = a = gamma*r/stddev.
__global__ __launch_bounds__(256) void k_bn_bwd3(float* __restrict__ dy, int ldd, const float* __restrict__ y, int ldy,
                                                 const float* __restrict__ z, int ldz, const float* __restrict__ stat, int ldc,
                                                 const float* __restrict__ sums, size_t rows, int cols, int relu) {
  const int c4 = cols >> 2;
  const size_t n = rows * (size_t)c4;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    float4 g = *reinterpret_cast<const float4*>(dy + r * ldd + c);
    if (relu) {
      const float4 yv = *reinterpret_cast<const float4*>(y + r * ldy + c);
      g = make_float4(yv.x > 0.f ? g.x : 0.f, yv.y > 0.f ? g.y : 0.f, yv.z > 0.f ? g.z : 0.f, yv.w > 0.f ? g.w : 0.f);
    }
    const float4 zv = *reinterpret_cast<const float4*>(z + r * ldz + c);
    const float4 mean = *reinterpret_cast<const float4*>(stat + c), sd = *reinterpret_cast<const float4*>(stat + ldc + c);
    const float4 a = *reinterpret_cast<const float4*>(stat + 4 * ldc + c);
    const float4 m1 = *reinterpret_cast<const float4*>(sums + c), m2 = *reinterpret_cast<const float4*>(sums + ldc + c);
    // (padding columns carry stddev = 0 and a = 0: they must come out as 0, not NaN)
    const float4 isd = make_float4(sd.x > 0.f ? 1.f / sd.x : 0.f, sd.y > 0.f ? 1.f / sd.y : 0.f, sd.z > 0.f ? 1.f / sd.z : 0.f,
                                   sd.w > 0.f ? 1.f / sd.w : 0.f);
    float4 o;
    o.x = a.x * (g.x - m1.x - (zv.x - mean.x) * isd.x * m2.x);
    o.y = a.y * (g.y - m1.y - (zv.y - mean.y) * isd.y * m2.y);
    o.z = a.z * (g.z - m1.z - (zv.z - mean.z) * isd.z * m2.z);
    o.w = a.w * (g.w - m1.w - (zv.w - mean.w) * isd.w * m2.w);
    *reinterpret_cast<float4*>(dy + r * ldd + c) = o;
  }
}

// ---- tall narrow activations (the R-CED feature maps: 1.6 M positions x 12-32 channels, leading dimension = the padded channel
// count) ----  The wide kernels above give a column to a lane: with 12-32 columns most lanes of a wave idle and every load is 4 bytes
// (0.3 TB/s).  Here the matrix is one dense run of 16-byte quads: a thread keeps ONE quad column (its per-column constants live in
// registers) and walks rows, 256/(ld/4) rows per workgroup pass, four rows in flight.  Partials keep the [slice][2][cols] layout, so
// k_bn_stats2 / k_bn_bwd2 finish them unchanged; the order of every sum is fixed by (rows, ld, slices).
__device__ __forceinline__ float4 bn_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <bool BWD>
__global__ __launch_bounds__(256) void k_bn_part_narrow(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ stat, int ldc, int ld, int rows, int cols, int per, int relu,
                                                        float* __restrict__ scratch) {
  __shared__ float red[2][256][4];
  const int q = ld >> 2, R = 256 / q, t = threadIdx.x, rr = t / q, cq = t - rr * q;
  const int rbeg = blockIdx.x * per, rend = min(rows, rbeg + per);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rr < R) {
    float4 m, k;                               // forward: m = the call's first row (the shift); backward: m = mean, k = 1/stddev
    if (BWD) {
      m = bn_ld4(stat + cq * 4);
      const float4 sd = bn_ld4(stat + ldc + cq * 4);
      k = make_float4(1.f / sd.x, 1.f / sd.y, 1.f / sd.z, 1.f / sd.w);
    } else {
      m = bn_ld4(z + cq * 4); k = m;
    }
    const size_t step = (size_t)R * ld;
    const float* pz = z + (size_t)(rbeg + rr) * ld + cq * 4;
    const float* pd = BWD ? dy + (size_t)(rbeg + rr) * ld + cq * 4 : pz;
    const float* py = BWD ? y + (size_t)(rbeg + rr) * ld + cq * 4 : pz;
    auto acc = [&](float4 zv, float4 g, float4 yv) {
      if (BWD) {
        if (relu) { g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f; }
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x += g.x * ((zv.x - m.x) * k.x); s2.y += g.y * ((zv.y - m.y) * k.y);
        s2.z += g.z * ((zv.z - m.z) * k.z); s2.w += g.w * ((zv.w - m.w) * k.w);
      } else {
        const float vx = zv.x - m.x, vy = zv.y - m.y, vz = zv.z - m.z, vw = zv.w - m.w;
        s1.x += vx; s1.y += vy; s1.z += vz; s1.w += vw;
        s2.x += vx * vx; s2.y += vy * vy; s2.z += vz * vz; s2.w += vw * vw;
      }
    };
    int r = rbeg + rr;
    for (; r + 3 * R < rend; r += 4 * R) {
      float4 zv[4], gv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        zv[u] = bn_ld4(pz + u * step);
        if (BWD) { gv[u] = bn_ld4(pd + u * step); yv[u] = relu ? bn_ld4(py + u * step) : zv[u]; } else { gv[u] = zv[u]; yv[u] = zv[u]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc(zv[u], gv[u], yv[u]);
      pz += 4 * step; pd += 4 * step; py += 4 * step;
    }
    for (; r < rend; r += R) {
      const float4 zv = bn_ld4(pz);
      acc(zv, BWD ? bn_ld4(pd) : zv, BWD && relu ? bn_ld4(py) : zv);
      pz += step; pd += step; py += step;
    }
  }
  *reinterpret_cast<float4*>(red[0][t]) = s1; *reinterpret_cast<float4*>(red[1][t]) = s2;
  __syncthreads();
  if (t < 2 * ld) {
    const int which = t / ld, c = t - which * ld;
    if (c < cols) {
      float sum = 0.f;
      for (int i = 0; i < R; ++i) sum += red[which][i * q + (c >> 2)][c & 3];
      scratch[((size_t)blockIdx.x * 2 + which) * cols + c] = sum;
    }
  }
}

// y = relu(z*a + b) (forward) / dz in place over dy (backward) on the same (row lane, quad column) decomposition
template <bool BWD>
__global__ __launch_bounds__(256) void k_bn_elem_narrow(const float* __restrict__ z, float* __restrict__ out, const float* __restrict__ y,
                                                        const float* __restrict__ stat, int ldc, const float* __restrict__ sums, int ld, int rows,
                                                        int relu) {
  const int q = ld >> 2, R = 256 / q, t = threadIdx.x, rr = t / q, cq = t - rr * q;
  if (rr >= R) return;
  const float4 a = bn_ld4(stat + 4 * ldc + cq * 4);
  float4 b, mean, isd, m1, m2;
  if (BWD) {
    mean = bn_ld4(stat + cq * 4);
    const float4 sd = bn_ld4(stat + ldc + cq * 4);
    // (padding columns carry stddev = 0 and a = 0: they must come out as 0, not NaN)
    isd = make_float4(sd.x > 0.f ? 1.f / sd.x : 0.f, sd.y > 0.f ? 1.f / sd.y : 0.f, sd.z > 0.f ? 1.f / sd.z : 0.f, sd.w > 0.f ? 1.f / sd.w : 0.f);
    m1 = bn_ld4(sums + cq * 4); m2 = bn_ld4(sums + ldc + cq * 4);
    b = a;
  } else {
    b = bn_ld4(stat + 5 * ldc + cq * 4);
    mean = isd = m1 = m2 = a;
  }
  auto one = [&](float4 zv, float4 g, float4 yv) {
    float4 o;
    if (BWD) {
      if (relu) { g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f; }
      o.x = a.x * (g.x - m1.x - (zv.x - mean.x) * isd.x * m2.x);
      o.y = a.y * (g.y - m1.y - (zv.y - mean.y) * isd.y * m2.y);
      o.z = a.z * (g.z - m1.z - (zv.z - mean.z) * isd.z * m2.z);
      o.w = a.w * (g.w - m1.w - (zv.w - mean.w) * isd.w * m2.w);
    } else {
      o = make_float4(zv.x * a.x + b.x, zv.y * a.y + b.y, zv.z * a.z + b.z, zv.w * a.w + b.w);
      if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    }
    return o;
  };
  const size_t gstep = (size_t)gridDim.x * R;                 // rows between two passes of this thread
  size_t r = (size_t)blockIdx.x * R + rr;
  for (; r + 3 * gstep < (size_t)rows; r += 4 * gstep) {
    float4 zv[4], gv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = (r + u * gstep) * ld + cq * 4;
      zv[u] = bn_ld4(z + o);
      if (BWD) { gv[u] = bn_ld4(out + o); yv[u] = relu ? bn_ld4(y + o) : zv[u]; } else { gv[u] = zv[u]; yv[u] = zv[u]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(out + (r + u * gstep) * ld + cq * 4) = one(zv[u], gv[u], yv[u]);
  }
  for (; r < (size_t)rows; r += gstep) {
    const size_t o = r * ld + cq * 4;
    const float4 zv = bn_ld4(z + o);
    *reinterpret_cast<float4*>(out + o) = one(zv, BWD ? bn_ld4(out + o) : zv, BWD && relu ? bn_ld4(y + o) : zv);
  }
}

static bool bn_narrow(int rows, int cols, int l0, int l1, int l2) {
  static int minrows = -1;                 // RSRGAN_BN_NARROW: least row count that takes this form (0 = never; tests set 1)
  if (minrows < 0) { const char* e = getenv("RSRGAN_BN_NARROW"); minrows = e ? std::max(0, atoi(e)) : 4096; }
  const int ld = (cols + 3) & ~3;
  return minrows > 0 && ld <= 64 && rows >= minrows && l0 == ld && l1 == ld && l2 == ld;
}

// UPDATE_OPS of one call, `times` times: assign_moving_average(v, value, decay, zero_debias=False) is v -= (v - value)*(1 - decay).
// One workgroup per layer: every thread reads the two scalar weights before thread 0 rewrites them.
__global__ __launch_bounds__(256) void k_bn_commit(int cols, BnVars v, const float* __restrict__ stat, int ldc, int times) {
  const double w_mean0 = v.rmw[0], w_sd0 = v.rsw[0];
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    const double mean = stat[c], sd = stat[ldc + c];
    double rm = v.rm[c], rs = v.rs[c], mm = v.mm[c], mv = v.mv[c], wm = w_mean0, ws = w_sd0;
    for (int t = 0; t < times; ++t) {
      rm -= (rm - mean) * (1.0 - BN_RENORM_DECAY); wm -= (wm - 1.0) * (1.0 - BN_RENORM_DECAY);
      rs -= (rs - sd) * (1.0 - BN_RENORM_DECAY); ws -= (ws - 1.0) * (1.0 - BN_RENORM_DECAY);
      const double new_mean = rm / wm, new_sd = rs / ws;
      mm -= (mm - new_mean) * (1.0 - BN_DECAY);
      mv -= (mv - (new_sd * new_sd - (double)BN_EPS)) * (1.0 - BN_DECAY);
    }
    v.rm[c] = (float)rm; v.rs[c] = (float)rs; v.mm[c] = (float)mm; v.mv[c] = (float)mv;
  }
  if (threadIdx.x == 0) {
    double wm = w_mean0, ws = w_sd0;
    for (int t = 0; t < times; ++t) { wm -= (wm - 1.0) * (1.0 - BN_RENORM_DECAY); ws -= (ws - 1.0) * (1.0 - BN_RENORM_DECAY); }
    v.rmw[0] = (float)wm; v.rsw[0] = (float)ws;
  }
}

// ---- few rows (the shipped frame-level batch: 256 frames): one launch per direction.  A workgroup owns 16 columns and all rows of
// every call (the discriminator's real | fake halves are consecutive calls): 16 row lanes sum a column, LDS reduces them in a fixed
// order, the column threads finish the moments / gradients, then the workgroup re-reads its (L2-resident) columns for the
// elementwise pass. ----
static int bn_small_rows() { static int v = -1; if (v < 0) { const char* e = getenv("RSRGAN_BN_SMALL_ROWS"); v = e ? atoi(e) : 384; } return v; }

constexpr int BNS_CW = 16, BNS_RL = 16;          // small path: 16 columns x 16 row lanes per workgroup (cols / 16 workgroups)

__device__ __forceinline__ float bns_reduce(float (*red)[BNS_CW], int l) {      // fixed-order sum over the row lanes
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < BNS_RL; ++i) t += red[i][l];
  return t;
}

__global__ __launch_bounds__(256) void k_bn_fwd_small(const float* __restrict__ z, int ldz, float* __restrict__ y, int ldy, int rows, int cols,
                                                      int calls, BnVars v, float* __restrict__ stat, int ldc, int training, int relu) {
  __shared__ float red[2][BNS_RL][BNS_CW];
  __shared__ float ab[2][BNS_CW];
  const int l = threadIdx.x & (BNS_CW - 1), rl = threadIdx.x / BNS_CW, c = blockIdx.x * BNS_CW + l;
  const bool cok = c < cols;
  for (int k = 0; k < calls; ++k) {
    const float* zk = z + (size_t)k * rows * ldz;
    float* yk = y + (size_t)k * rows * ldy;
    float* st = stat + (size_t)k * BN_STAT_ROWS * ldc;
    if (training) {
      float s1 = 0.f, s2 = 0.f;
      const float z0 = cok ? zk[c] : 0.f;
      if (cok) {
#pragma unroll 8
        for (int r = rl; r < rows; r += BNS_RL) { const float d = zk[(size_t)r * ldz + c] - z0; s1 += d; s2 += d * d; }
      }
      red[0][rl][l] = s1; red[1][rl][l] = s2;
      __syncthreads();
      if (rl == 0 && cok) {
        const double t1 = (double)bns_reduce(red[0], l), t2 = (double)bns_reduce(red[1], l);
        const double m0 = t1 / rows, var = fmax(t2 / rows - m0 * m0, 0.0), mean = (double)z0 + m0, sd = sqrt(var + (double)BN_EPS);
        const double mixed_mean = (double)v.rm[c] + (1.0 - (double)v.rmw[0]) * mean;
        const double mixed_sd = (double)v.rs[c] + (1.0 - (double)v.rsw[0]) * sd;
        const double r = sd / mixed_sd, d = (mean - mixed_mean) / mixed_sd, g = v.gamma[c], a = r * g / sd;
        const float af = (float)a, bf = (float)(d * g + (double)v.beta[c] - mean * a);
        st[c] = (float)mean; st[ldc + c] = (float)sd; st[2 * ldc + c] = (float)r; st[3 * ldc + c] = (float)d;
        st[4 * ldc + c] = af; st[5 * ldc + c] = bf;
        ab[0][l] = af; ab[1][l] = bf;
      }
    } else if (rl == 0 && cok) {
      const double a = (double)v.gamma[c] / sqrt((double)v.mv[c] + (double)BN_EPS);
      const float af = (float)a, bf = (float)((double)v.beta[c] - (double)v.mm[c] * a);
      st[4 * ldc + c] = af; st[5 * ldc + c] = bf;
      ab[0][l] = af; ab[1][l] = bf;
    }
    __syncthreads();
    if (cok) {
      const float a = ab[0][l], b = ab[1][l];
#pragma unroll 8
      for (int r = rl; r < rows; r += BNS_RL) {
        const float o = zk[(size_t)r * ldz + c] * a + b;
        yk[(size_t)r * ldy + c] = relu ? fmaxf(o, 0.f) : o;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_small(float* __restrict__ dy, int ldd, const float* __restrict__ y, int ldy,
                                                      const float* __restrict__ z, int ldz, int rows, int cols, int calls,
                                                      const float* __restrict__ stat, int ldc, float* __restrict__ dbeta,
                                                      float* __restrict__ dgamma, int relu) {
  __shared__ float red[2][BNS_RL][BNS_CW];
  __shared__ float ms[2][BNS_CW];
  const int l = threadIdx.x & (BNS_CW - 1), rl = threadIdx.x / BNS_CW, c = blockIdx.x * BNS_CW + l;
  const bool cok = c < cols;
  double gb = 0.0, gg = 0.0;                       // (column thread rl == 0: dbeta / dgamma summed over the calls in call order)
  for (int k = 0; k < calls; ++k) {
    float* dk = dy + (size_t)k * rows * ldd;
    const float* yk = y + (size_t)k * rows * ldy;
    const float* zk = z + (size_t)k * rows * ldz;
    const float* st = stat + (size_t)k * BN_STAT_ROWS * ldc;
    const float mean = cok ? st[c] : 0.f, isd = cok ? 1.f / st[ldc + c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (cok) {
#pragma unroll 8
      for (int r = rl; r < rows; r += BNS_RL) {
        float g = dk[(size_t)r * ldd + c];
        if (relu && !(yk[(size_t)r * ldy + c] > 0.f)) g = 0.f;
        s1 += g; s2 += g * ((zk[(size_t)r * ldz + c] - mean) * isd);
      }
    }
    red[0][rl][l] = s1; red[1][rl][l] = s2;
    __syncthreads();
    if (rl == 0 && cok) {
      const double t1 = (double)bns_reduce(red[0], l), t2 = (double)bns_reduce(red[1], l);
      gb += t1; gg += (double)st[2 * ldc + c] * t2 + (double)st[3 * ldc + c] * t1;
      ms[0][l] = (float)(t1 / rows); ms[1][l] = (float)(t2 / rows);
    }
    __syncthreads();
    if (cok) {
      const float a = st[4 * ldc + c], m1 = ms[0][l], m2 = ms[1][l];
#pragma unroll 8
      for (int r = rl; r < rows; r += BNS_RL) {
        float g = dk[(size_t)r * ldd + c];
        if (relu && !(yk[(size_t)r * ldy + c] > 0.f)) g = 0.f;
        dk[(size_t)r * ldd + c] = a * (g - m1 - (zk[(size_t)r * ldz + c] - mean) * isd * m2);
      }
    }
    __syncthreads();
  }
  if (dbeta && rl == 0 && cok) { dbeta[c] = (float)gb; dgamma[c] = (float)gg; }
}

// several layers' update ops in one launch (blockIdx.x = entry)
__global__ __launch_bounds__(256) void k_bn_commit_many(BnCommitList cl) {
  const BnCommit e = cl.e[blockIdx.x];
  const BnVars& v = e.v;
  const double w_mean0 = v.rmw[0], w_sd0 = v.rsw[0];
  __syncthreads();
  for (int c = threadIdx.x; c < e.cols; c += 256) {
    double rm = v.rm[c], rs = v.rs[c], mm = v.mm[c], mv = v.mv[c], wm = w_mean0, ws = w_sd0;
    for (int u = 0; u < 2; ++u) {                 // update 0: call 0 `times0` times, update 1: call 1 `times1` times
      if ((u == 0 ? e.times0 : e.times1) == 0) continue;          // (single-call layers have no second statistics slot)
      const float* st = e.stat + (size_t)u * BN_STAT_ROWS * e.ldc;
      const double mean = st[c], sd = st[e.ldc + c];
      for (int t = 0; t < (u == 0 ? e.times0 : e.times1); ++t) {
        rm -= (rm - mean) * (1.0 - BN_RENORM_DECAY); wm -= (wm - 1.0) * (1.0 - BN_RENORM_DECAY);
        rs -= (rs - sd) * (1.0 - BN_RENORM_DECAY); ws -= (ws - 1.0) * (1.0 - BN_RENORM_DECAY);
        const double new_mean = rm / wm, new_sd = rs / ws;
        mm -= (mm - new_mean) * (1.0 - BN_DECAY);
        mv -= (mv - (new_sd * new_sd - (double)BN_EPS)) * (1.0 - BN_DECAY);
      }
    }
    v.rm[c] = (float)rm; v.rs[c] = (float)rs; v.mm[c] = (float)mm; v.mv[c] = (float)mv;
  }
  if (threadIdx.x == 0) {
    double wm = w_mean0, ws = w_sd0;
    for (int t = 0; t < e.times0 + e.times1; ++t) { wm -= (wm - 1.0) * (1.0 - BN_RENORM_DECAY); ws -= (ws - 1.0) * (1.0 - BN_RENORM_DECAY); }
    v.rmw[0] = (float)wm; v.rsw[0] = (float)ws;
  }
}
void launch_bn_commit_many(const BnCommitList& cl, hipStream_t s) {
  if (cl.n > 0) hipLaunchKernelGGL(k_bn_commit_many, dim3(cl.n), dim3(256), 0, s, cl);
}

static int bn_narrow_grid(int rows, int ld) {          // four rows per thread and pass, at most 8192 workgroups
  const int R = 256 / (ld >> 2);
  return std::max(1, std::min(8192, (rows + 4 * R - 1) / (4 * R)));
}

static int bn_slices(int rows, int cols, size_t scratch_floats, int* per) {
  int slices = (int)std::min<size_t>(512, scratch_floats / ((size_t)2 * std::max(cols, 1)));
  slices = std::max(1, std::min(slices, (rows + 63) / 64));
  *per = (rows + slices - 1) / slices;
  return (rows + *per - 1) / *per;
}

// `calls` consecutive calls of `rows` rows each (statistics slots 0 .. calls-1 of `stat`)
void launch_bn_forward(const float* z, int ldz, float* y, int ldy, int rows, int cols, const BnVars& v, float* stat, int ldc, bool training,
                       bool relu, float* scratch, size_t scratch_floats, hipStream_t s, int calls) {
  if (rows <= bn_small_rows() && cols >= 64) {
    hipLaunchKernelGGL(k_bn_fwd_small, dim3((cols + BNS_CW - 1) / BNS_CW), dim3(256), 0, s, z, ldz, y, ldy, rows, cols, calls, v, stat, ldc,
                       training ? 1 : 0, relu ? 1 : 0);
    return;
  }
  for (int k = 1; k < calls; ++k)
    launch_bn_forward(z + (size_t)k * rows * ldz, ldz, y + (size_t)k * rows * ldy, ldy, rows, cols, v, stat + (size_t)k * BN_STAT_ROWS * ldc, ldc,
                      training, relu, scratch, scratch_floats, s, 1);
  if (training) {
    int per;
    const int slices = bn_slices(rows, cols, scratch_floats, &per);
    if (bn_narrow(rows, cols, ldz, ldy, ldz))
      hipLaunchKernelGGL(k_bn_part_narrow<false>, dim3(slices), dim3(256), 0, s, z, nullptr, nullptr, nullptr, 0, ldz, rows, cols, per, 0, scratch);
    else
      hipLaunchKernelGGL(k_bn_stats1, dim3((cols + 63) / 64, slices), dim3(256), 0, s, z, ldz, rows, cols, per, scratch);
    hipLaunchKernelGGL(k_bn_stats2, dim3((cols + 63) / 64), dim3(256), 0, s, scratch, slices, z, rows, cols, v, stat, ldc);
  } else {
    hipLaunchKernelGGL(k_bn_infer_coef, dim3((cols + 255) / 256), dim3(256), 0, s, cols, v, stat, ldc);
  }
  const int cp = (cols + 3) & ~3;
  const size_t n = (size_t)rows * (cp >> 2);
  const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  if (bn_narrow(rows, cols, ldz, ldy, ldz)) {
    hipLaunchKernelGGL(k_bn_elem_narrow<false>, dim3(bn_narrow_grid(rows, ldz)), dim3(256), 0, s, z, y, nullptr, stat, ldc, nullptr, ldz, rows,
                       relu ? 1 : 0);
    return;
  }
  hipLaunchKernelGGL(k_bn_apply, dim3(grid), dim3(256), 0, s, z, ldz, y, ldy, (size_t)rows, cp, stat + 4 * ldc, stat + 5 * ldc, relu ? 1 : 0);
}

// dy: gradient w.r.t. y (the layer's output AFTER its ReLU when relu) -> overwritten by the gradient w.r.t. z.
// dbeta / dgamma may be null (data gradient only).  sums: 2*ldc floats of work space.
void launch_bn_backward(float* dy, int ldd, const float* y, int ldy, const float* z, int ldz, int rows, int cols, const float* stat, int ldc,
                        float* dbeta, float* dgamma, bool accumulate, bool relu, float* sums, float* scratch, size_t scratch_floats,
                        hipStream_t s, int calls) {
  if (rows <= bn_small_rows() && cols >= 64 && !accumulate) {
    hipLaunchKernelGGL(k_bn_bwd_small, dim3((cols + BNS_CW - 1) / BNS_CW), dim3(256), 0, s, dy, ldd, y, ldy, z, ldz, rows, cols, calls, stat, ldc, dbeta,
                       dgamma, relu ? 1 : 0);
    return;
  }
  // (call 0 assigns dbeta / dgamma, the later calls accumulate: the order of the small path)
  if (calls > 1) {
    launch_bn_backward(dy, ldd, y, ldy, z, ldz, rows, cols, stat, ldc, dbeta, dgamma, accumulate, relu, sums, scratch, scratch_floats, s, 1);
    for (int k = 1; k < calls; ++k)
      launch_bn_backward(dy + (size_t)k * rows * ldd, ldd, y + (size_t)k * rows * ldy, ldy, z + (size_t)k * rows * ldz, ldz, rows, cols,
                         stat + (size_t)k * BN_STAT_ROWS * ldc, ldc, dbeta, dgamma, true, relu, sums, scratch, scratch_floats, s, 1);
    return;
  }
  int per;
  const int slices = bn_slices(rows, cols, scratch_floats, &per);
  const bool narrow = bn_narrow(rows, cols, ldd, ldy, ldz);
  if (narrow)
    hipLaunchKernelGGL(k_bn_part_narrow<true>, dim3(slices), dim3(256), 0, s, z, dy, y, stat, ldc, ldz, rows, cols, per, relu ? 1 : 0, scratch);
  else
    hipLaunchKernelGGL(k_bn_bwd1, dim3((cols + 63) / 64, slices), dim3(256), 0, s, dy, ldd, y, ldy, z, ldz, stat, ldc, rows, cols, per,
                       relu ? 1 : 0, scratch);
  hipLaunchKernelGGL(k_bn_bwd2, dim3((cols + 63) / 64), dim3(256), 0, s, scratch, slices, rows, cols, stat, ldc, dbeta, dgamma,
                     accumulate ? 1 : 0, sums);
  const int cp = (cols + 3) & ~3;
  const size_t n = (size_t)rows * (cp >> 2);
  const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  if (narrow) {
    hipLaunchKernelGGL(k_bn_elem_narrow<true>, dim3(bn_narrow_grid(rows, ldz)), dim3(256), 0, s, z, dy, y, stat, ldc, sums, ldz, rows, relu ? 1 : 0);
    return;
  }
  hipLaunchKernelGGL(k_bn_bwd3, dim3(grid), dim3(256), 0, s, dy, ldd, y, ldy, z, ldz, stat, ldc, sums, (size_t)rows, cp, relu ? 1 : 0);
}

void launch_bn_commit(int cols, const BnVars& v, const float* stat, int ldc, int times, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_commit, dim3(1), dim3(256), 0, s, cols, v, stat, ldc, times);
}

}  // namespace rsr
