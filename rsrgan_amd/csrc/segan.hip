// segan.hip -- the small kernels of the SEGAN-style conv G/D (models/segan.py, generator.py:AEGenerator, discriminator.py,
// utils/bnorm.py:VBN; BASELINE.json configs[4]).  The strided convolutions themselves are GEMMs of window views (gemm.hip:
// launch_gemm_mapped); what lives here is everything around them, all HBM-bound elementwise / column-reduction work on
// channels-last [rows = batch x position][channels] fp32 buffers, plus the direct kernels of the single-channel ends of the
// networks (a 1-channel window view is not 16-byte aligned, and those layers are < 1 % of the FLOPs).
// Every reduction is two-stage with a fixed order (no float atomics): bit-reproducible.
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <vector>
#include "segan.h"

namespace rsr {

// ---- dense [B][L][C] -> [B][pf + L + pb][C] with zero rows in front / behind (the window views never leave the buffer)
__global__ __launch_bounds__(256) void k_pad_rows(const float* __restrict__ src, float* __restrict__ dst, int B, int L, int C4, int pf, int pb) {
  const size_t Lp = (size_t)pf + L + pb, n = (size_t)B * Lp * C4;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t row = i / C4, c = i - row * C4;
    const size_t b = row / Lp;
    const long long r = (long long)(row - b * Lp) - pf;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 0 && r < L) v = reinterpret_cast<const float4*>(src)[((size_t)b * L + r) * C4 + c];
    reinterpret_cast<float4*>(dst)[i] = v;
  }
}
void launch_pad_rows(const float* src, float* dst, int B, int L, int C, int pf, int pb, hipStream_t s) {
  const size_t n = (size_t)B * (pf + L + pb) * (C / 4);
  hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, src, dst, B, L, C / 4, pf, pb);
}

// ---- single-channel input, stride-2 SAME convolution (utils/ops.py:78-100 with Cin = 1): z[b,o,c] = sum_dk x[b, 2o+dk-pl] W[dk][c] + bias[c]
__global__ __launch_bounds__(256) void k_conv1_fwd(const float* __restrict__ x, int ldx, int L, int Lo, int k, int pl, const float* __restrict__ W, int ldw,
                                                   const float* __restrict__ bias, int C, float* __restrict__ z, int ldz, size_t rows) {
  extern __shared__ float sw[];                          // W [k][C] (+ bias [C])
  for (int i = threadIdx.x; i < k * C; i += 256) sw[i] = W[(size_t)(i / C) * ldw + (i % C)];
  for (int i = threadIdx.x; i < C; i += 256) sw[k * C + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  for (size_t r = blockIdx.x * (size_t)256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
    const size_t b = r / Lo;
    const int o = (int)(r - b * Lo);
    const float* xb = x + b * ldx;
    float* zr = z + r * ldz;
    for (int c0 = 0; c0 < C; c0 += 16) {
      float acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = sw[k * C + c0 + c];
      for (int dk = 0; dk < k; ++dk) {
        const int i = 2 * o + dk - pl;
        const float xv = (i >= 0 && i < L) ? xb[i] : 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(xv, sw[dk * C + c0 + c], acc[c]);
      }
#pragma unroll
      for (int c = 0; c < 16; c += 4) *reinterpret_cast<float4*>(zr + c0 + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    }
  }
}
void launch_conv1_fwd(const float* x, int ldx, int B, int L, int k, const float* W, int ldw, const float* bias, int C, float* z, int ldz, hipStream_t s) {
  const int Lo = (L + 1) / 2, total = std::max((Lo - 1) * 2 + k - L, 0), pl = total / 2;
  const size_t rows = (size_t)B * Lo;
  hipLaunchKernelGGL(k_conv1_fwd, dim3((unsigned)std::min<size_t>((rows + 255) / 256, 4096)), dim3(256), (size_t)(k + 1) * C * sizeof(float), s,
                     x, ldx, L, Lo, k, pl, W, ldw, bias, C, z, ldz, rows);
}

// its weight gradient dW[dk][c] = sum_{b,o} x[b, 2o+dk-pl] dz[b,o,c]: partials per row chunk, then a fixed-order sum
// Workgroup = one batch row x a range of output positions, walked in sub-chunks of 512 positions staged in LDS (the gradient rows
// as float4, the input window once): thread = filter element (dk, c).  (First form: every thread walked 512 rows straight from
// global memory with a dependent fma chain and an integer division per row: 480 us per call, 1.4 ms of the SEGAN step.)
constexpr int C1_SUB = 512;
__global__ __launch_bounds__(256) void k_conv1_wgrad_part(const float* __restrict__ x, int ldx, int L, int Lo, int k, int pl, const float* __restrict__ dz, int ldz,
                                                          int C, int chunk, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* dzs = sm;                                         // [C1_SUB][C]
  float* xs = sm + (size_t)C1_SUB * C;                     // [2 * C1_SUB + k]
  const int b = blockIdx.y, o0 = blockIdx.x * chunk, o1 = min(Lo, o0 + chunk);
  const int ne = k * C;
  constexpr int NE = 4;                                    // elements threadIdx.x + 256 u (k * C <= 1024)
  float acc[NE] = {0.f, 0.f, 0.f, 0.f};
  int dk[NE], cc[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) { const int e = min(threadIdx.x + 256 * u, ne - 1); dk[u] = e / C; cc[u] = e - dk[u] * C; }
  for (int s0 = o0; s0 < o1; s0 += C1_SUB) {
    const int n = min(C1_SUB, o1 - s0);
    __syncthreads();
    for (int i = threadIdx.x; i < n * C / 4; i += 256)
      reinterpret_cast<float4*>(dzs)[i] = *reinterpret_cast<const float4*>(dz + ((size_t)b * Lo + s0) * ldz + (size_t)(i * 4 / C) * ldz + (i * 4 % C));
    for (int i = threadIdx.x; i < 2 * n + k; i += 256) {
      const int xi = 2 * s0 - pl + i;
      xs[i] = (xi >= 0 && xi < L) ? x[(size_t)b * ldx + xi] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      if ((int)threadIdx.x + 256 * u >= ne) continue;
      float a_ = acc[u];
#pragma unroll 8
      for (int o = 0; o < n; ++o) a_ = fmaf(xs[2 * o + dk[u]], dzs[o * C + cc[u]], a_);
      acc[u] = a_;
    }
  }
  float* po = part + ((size_t)b * gridDim.x + blockIdx.x) * ne;
#pragma unroll
  for (int u = 0; u < NE; ++u)
    if ((int)threadIdx.x + 256 * u < ne) po[threadIdx.x + 256 * u] = acc[u];
}
__global__ __launch_bounds__(256) void k_sum_parts(const float* __restrict__ part, int nparts, int n, int C, float* __restrict__ out, int ldo) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float acc = 0.f;
  for (int p = 0; p < nparts; ++p) acc += part[(size_t)p * n + e];
  out[(size_t)(e / C) * ldo + (e % C)] = acc;
}
void launch_conv1_wgrad(const float* x, int ldx, int B, int L, int k, const float* dz, int ldz, int C, float* dW, int ldw, float* scratch, size_t scratch_floats,
                        hipStream_t s) {
  const int Lo = (L + 1) / 2, total = std::max((Lo - 1) * 2 + k - L, 0), pl = total / 2;
  if (k * C > 1024 || C % 4 || ldz % 4) { fprintf(stderr, "rsrgan: conv1 weight gradient needs k * C <= 1024 and C, ldz multiples of 4\n"); abort(); }
  int chunk = 1024;                                        // positions per workgroup: ~B * Lo / 1024 workgroups
  while ((size_t)B * ((Lo + chunk - 1) / chunk) * k * C > scratch_floats) chunk *= 2;
  const int nch = (Lo + chunk - 1) / chunk;
  const size_t lds = ((size_t)C1_SUB * C + 2 * C1_SUB + k) * sizeof(float);
  hipLaunchKernelGGL(k_conv1_wgrad_part, dim3(nch, B), dim3(256), lds, s, x, ldx, L, Lo, k, pl, dz, ldz, C, chunk, scratch);
  hipLaunchKernelGGL(k_sum_parts, dim3((k * C + 255) / 256), dim3(256), 0, s, scratch, B * nch, k * C, C, dW, ldw);
}

// ---- single-channel OUTPUT of the transposed stride-2 convolution: t[b,i] = bias + sum over (o, dk) with 2o + dk - pl = i of s[b,o,:] . W[dk][:]
// (the last deconv of the generator, utils/ops.py:277-311 with one output channel, and the data gradient of a Cin = 1 downconv)
__global__ __launch_bounds__(256) void k_tconv1(const float* __restrict__ S, int lds, int Ls, int C, int Lt, int k, int pl, const float* __restrict__ W,
                                                int ldw, const float* __restrict__ bias, float* __restrict__ t, int ldt, size_t n) {
  extern __shared__ float sw[];
  for (int i = threadIdx.x; i < k * C; i += 256) sw[i] = W[(size_t)(i / C) * ldw + (i % C)];
  __syncthreads();
  const float bv = bias ? bias[0] : 0.f;
  for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    const size_t b = idx / Lt;
    const int i = (int)(idx - b * Lt);
    float acc = bv;
    for (int dk = (i + pl) & 1; dk < k; dk += 2) {
      const int o2 = i + pl - dk;
      if (o2 < 0 || o2 >= 2 * Ls) continue;
      const float* sr = S + (b * Ls + (o2 >> 1)) * lds;
      const float* wr = sw + dk * C;
      for (int c = 0; c < C; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(sr + c);
        acc = fmaf(v.x, wr[c], acc); acc = fmaf(v.y, wr[c + 1], acc); acc = fmaf(v.z, wr[c + 2], acc); acc = fmaf(v.w, wr[c + 3], acc);
      }
    }
    t[b * ldt + i] = acc;
  }
}
void launch_tconv1(const float* S, int lds, int B, int Ls, int C, int Lt, int k, const float* W, int ldw, const float* bias, float* t, int ldt,
                   hipStream_t s) {
  const int total = std::max((Ls - 1) * 2 + k - Lt, 0), pl = total / 2;
  const size_t n = (size_t)B * Lt;
  hipLaunchKernelGGL(k_tconv1, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), (size_t)k * C * sizeof(float), s, S, lds, Ls, C, Lt, k,
                     pl, W, ldw, bias, t, ldt, n);
}

// ---- Wt_e[(rr, a)][b] = W[dk = 2 (n_e - 1 - rr) + e][b][a]: the filter of one parity class of a transposed convolution as a GEMM operand
__global__ __launch_bounds__(256) void k_prep_tconv(const float* __restrict__ W, int ldw, int nb, int na, int e, int ne, float* __restrict__ dst, int ldd) {
  const size_t n = (size_t)ne * na * nb;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i % nb);
    const size_t ra = i / nb;
    const int a = (int)(ra % na), rr = (int)(ra / na);
    const int dk = 2 * (ne - 1 - rr) + e;
    dst[ra * ldd + b] = W[((size_t)dk * nb + b) * ldw + a];
  }
}
void launch_prep_tconv(const float* W, int ldw, int nb, int na, int e, int ne, float* dst, int ldd, hipStream_t s) {
  const size_t n = (size_t)ne * na * nb;
  hipLaunchKernelGGL(k_prep_tconv, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, W, ldw, nb, na, e, ne, dst, ldd);
}
// every parity-class operand of a net in ONE launch (a weight refresh used to be 40 + 20 launches of 2-60 us)
__global__ __launch_bounds__(256) void k_prep_tconv_many(const PrepTconvList pl) {
  // block = one 32 (a) x 32 (b) tile of one tap rr of one job, transposed through LDS: reads run along a (contiguous in W), writes
  // along b (contiguous in dst).  (Thread-per-element with b fastest read W with a stride of ldw floats: 60 us for the 512 x 1024 layer.)
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < pl.n && (int)blockIdx.x >= pl.first[j + 1]) ++j;
  const PrepTconvJob J = pl.j[j];
  const int ta = (J.na + 31) / 32, tb = (J.nb + 31) / 32;
  int t = (int)blockIdx.x - pl.first[j];
  const int rr = t / (ta * tb); t -= rr * ta * tb;
  const int a0 = (t / tb) * 32, b0 = (t % tb) * 32;
  const int dk = 2 * (J.ne - 1 - rr) + J.e;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int b = b0 + ty + 8 * k, a = a0 + tx;
    tile[ty + 8 * k][tx] = (b < J.nb && a < J.na) ? J.W[((size_t)dk * J.nb + b) * J.ldw + a] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = a0 + ty + 8 * k, b = b0 + tx;
    if (a < J.na && b < J.nb) J.dst[((size_t)rr * J.na + a) * J.ldd + b] = tile[tx][ty + 8 * k];
  }
}
void launch_prep_tconv_many(PrepTconvList& pl, hipStream_t s) {
  int blocks = 0;
  for (int j = 0; j < pl.n; ++j) {
    pl.first[j] = blocks;
    blocks += pl.j[j].ne * ((pl.j[j].na + 31) / 32) * ((pl.j[j].nb + 31) / 32);
  }
  if (blocks) hipLaunchKernelGGL(k_prep_tconv_many, dim3(blocks), dim3(256), 0, s, pl);
}

// ---- T[b,i,:] = T_e[b, (i - i0_e) / 2, :] + bias, e = (i + pl) & 1: the two parity classes of a transposed convolution back in position order
__global__ __launch_bounds__(256) void k_interleave(const float* __restrict__ T0, const float* __restrict__ T1, int Q0, int Q1, int i00, int i01, int pl,
                                                    const float* __restrict__ bias, float* __restrict__ T, int Lt, int C4, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t row = i / C4;
    const int c = (int)(i - row * C4);
    const size_t b = row / Lt;
    const int p = (int)(row - b * Lt);
    const int e = (p + pl) & 1;
    const float4 v = e ? reinterpret_cast<const float4*>(T1)[(b * Q1 + ((p - i01) >> 1)) * C4 + c]
                       : reinterpret_cast<const float4*>(T0)[(b * Q0 + ((p - i00) >> 1)) * C4 + c];
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = reinterpret_cast<const float4*>(bias)[c];
    reinterpret_cast<float4*>(T)[i] = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
  }
}
void launch_interleave(const float* T0, const float* T1, int Q0, int Q1, int i00, int i01, int pl, const float* bias, float* T, int B, int Lt, int C,
                       hipStream_t s) {
  const size_t n = (size_t)B * Lt * (C / 4);
  hipLaunchKernelGGL(k_interleave, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, T0, T1, Q0, Q1, i00, i01, pl, bias, T, Lt,
                     C / 4, n);
}

// ---- PReLU (utils/ops.py:123-134: relu(x) + alpha (x - |x|) / 2) / leaky-ReLU (tf.maximum(x, a x), :120-121) into a column range of a wider buffer
__global__ __launch_bounds__(256) void k_act_fwd(const float* __restrict__ z, int C, const float* __restrict__ alpha, float leak, float* __restrict__ out,
                                                 int ldo, int coff, size_t rows) {
  const size_t n = rows * C;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const float v = z[i], a = alpha ? alpha[c] : leak;
    out[r * ldo + coff + c] = v > 0.f ? v : a * v;
  }
}
void launch_act_fwd(const float* z, int C, const float* alpha, float leak, float* out, int ldo, int coff, size_t rows, hipStream_t s) {
  const size_t n = rows * C;
  hipLaunchKernelGGL(k_act_fwd, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, z, C, alpha, leak, out, ldo, coff, rows);
}
__global__ __launch_bounds__(256) void k_copy_cols(const float* __restrict__ src, int lds, int soff, float* __restrict__ dst, int ldd, int doff, int C,
                                                   size_t rows, int accumulate) {
  const size_t n = rows * C;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const float v = src[r * lds + soff + c];
    float* d = dst + r * ldd + doff + c;
    *d = accumulate ? *d + v : v;
  }
}
void launch_copy_cols(const float* src, int lds, int soff, float* dst, int ldd, int doff, int C, size_t rows, bool accumulate, hipStream_t s) {
  const size_t n = rows * C;
  hipLaunchKernelGGL(k_copy_cols, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, src, lds, soff, dst, ldd, doff, C, rows,
                     accumulate ? 1 : 0);
}
// dz = dy[:, coff : coff + C] * act'(z) (+ extra): act' = 1 (z > 0), a (z < 0); PReLU at exactly 0: a / 2 (d|x|/dx = 0 in TensorFlow),
// leaky-ReLU at 0: 1 (tf.maximum sends a tie to its first argument)
__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ dy, int ldy, int coff, const float* __restrict__ z, int C,
                                                 const float* __restrict__ alpha, float leak, const float* __restrict__ extra, float* __restrict__ dz,
                                                 size_t rows) {
  const size_t n = rows * C;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const float v = z[i], a = alpha ? alpha[c] : leak;
    const float d = v > 0.f ? 1.f : (v < 0.f ? a : (alpha ? 0.5f * a : 1.f));
    dz[i] = dy[r * ldy + coff + c] * d + (extra ? extra[i] : 0.f);
  }
}
void launch_act_bwd(const float* dy, int ldy, int coff, const float* z, int C, const float* alpha, float leak, const float* extra, float* dz, size_t rows,
                    hipStream_t s) {
  const size_t n = rows * C;
  hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, dy, ldy, coff, z, C, alpha, leak, extra, dz, rows);
}

// ---- column reductions over `rows` rows of P consecutive passes (segments of rows_per rows): two outputs per (pass, column)
//  MODE 0: (sum a, -)                       bias gradients
//  MODE 1: (sum a[:, coff+c] min(b, 0), -)  d alpha of PReLU (a = dy with leading dimension lda, b = z)
//  MODE 2: (sum a, sum a^2)                 VBN batch statistics (bnorm.py:40-41)
//  MODE 3: (sum g, sum g (a - mu)), g = dy * leaky'(a * sc + sh)        VBN backward (b = dy; per-pass coefficient rows in `coef`)
template <int MODE>
__global__ __launch_bounds__(256) void k_colred_part(const float* __restrict__ a, int lda, int coff, const float* __restrict__ b, int ldb, int C,
                                                     size_t rows_per, int chunk, int chunks_per, const float* __restrict__ coef, int ldcoef, float leak,
                                                     float* __restrict__ part) {
  // block = (pass, chunk of rows); thread t -> column t % C (C <= 256 handled by a column loop), row lane t / C
  const int pass = blockIdx.x / chunks_per, ch = blockIdx.x - pass * chunks_per;
  const size_t r0 = (size_t)pass * rows_per + (size_t)ch * chunk, r1 = min((size_t)(pass + 1) * rows_per, r0 + chunk);
  __shared__ float red[2][256];
  const int lanes = C >= 256 ? 1 : 256 / C;              // row lanes per column group
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int cw = min(C - c0, 256);
    const int lc = threadIdx.x % cw, lr = threadIdx.x / cw;
    float s0 = 0.f, s1 = 0.f;
    if (lr < lanes || cw == 256) {
      const int c = c0 + lc;
      float mu = 0.f, sc = 0.f, sh = 0.f;
      if (MODE == 3) { mu = coef[(size_t)(pass * 8 + 0) * ldcoef + c]; sc = coef[(size_t)(pass * 8 + 3) * ldcoef + c]; sh = coef[(size_t)(pass * 8 + 4) * ldcoef + c]; }
      const size_t step = cw == 256 ? 1 : lanes;
      auto acc1 = [&](float av, float bv) {
        if (MODE == 0) s0 += av;
        else if (MODE == 1) s0 += av * fminf(bv, 0.f);
        else if (MODE == 2) { s0 += av; s1 = fmaf(av, av, s1); }
        else { const float g = bv * (fmaf(av, sc, sh) >= 0.f ? 1.f : leak); s0 += g; s1 = fmaf(g, av - mu, s1); }
      };
      constexpr bool TWO = MODE == 1 || MODE == 3;
      size_t r = r0 + lr;
      for (; r + 7 * step < r1; r += 8 * step) {           // eight rows in flight, accumulated in row order (same bits as one at a time)
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = a[(r + u * step) * lda + coff + c]; bv[u] = TWO ? b[(r + u * step) * ldb + c] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc1(av[u], bv[u]);
      }
      for (; r < r1; r += step) acc1(a[r * lda + coff + c], TWO ? b[r * ldb + c] : 0.f);
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < cw) {                              // fixed order over the row lanes
      float t0 = 0.f, t1 = 0.f;
      const int nl = cw == 256 ? 1 : lanes;
      for (int l = 0; l < nl; ++l) { t0 += red[0][l * cw + threadIdx.x]; t1 += red[1][l * cw + threadIdx.x]; }
      part[((size_t)blockIdx.x * 2 + 0) * C + c0 + threadIdx.x] = t0;
      part[((size_t)blockIdx.x * 2 + 1) * C + c0 + threadIdx.x] = t1;
    }
    __syncthreads();
  }
}
// The same reductions with 16-byte loads: thread = (group of 4 columns, row lane), four rows in flight per thread (the scalar form
// above keeps one 4-byte load in flight per thread: 0.5-0.8 TB/s on the [3 x 262144][16] tensors of the first SEGAN blocks, 25 % of
// the SEGAN step in round 3's first profile).  Needs C, lda, ldb, coff multiples of 4.
template <int MODE>
__global__ __launch_bounds__(256) void k_colred_part4(const float* __restrict__ a, int lda, int coff, const float* __restrict__ b, int ldb, int C,
                                                      size_t rows_per, int chunk, int chunks_per, const float* __restrict__ coef, int ldcoef, float leak,
                                                      float* __restrict__ part) {
  const int pass = blockIdx.x / chunks_per, ch = blockIdx.x - pass * chunks_per;
  const size_t r0 = (size_t)pass * rows_per + (size_t)ch * chunk, r1 = min((size_t)(pass + 1) * rows_per, r0 + chunk);
  __shared__ double red[2][4][256];
  const int C4 = C >> 2;
  for (int g0 = 0; g0 < C4; g0 += 256) {
    const int gw = min(C4 - g0, 256);                    // column groups of this round
    const int lanes = 256 / gw;                          // row lanes per column group
    const int lg = threadIdx.x % gw, lr = threadIdx.x / gw;
    // accumulators in double: the VBN-backward sums cancel to 1e-3 of their terms (the d beta of the 16384-sample parity case lost
    // its 2e-3 against the fp64 oracle with fp32 accumulators in this summation order); the kernel is bandwidth-bound either way
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (lr < lanes) {
      const int c = (g0 + lg) * 4;
      float mu[4] = {0, 0, 0, 0}, sc[4] = {0, 0, 0, 0}, sh[4] = {0, 0, 0, 0};
      if (MODE == 3) {
        const float4 m4 = *reinterpret_cast<const float4*>(coef + (size_t)(pass * 8 + 0) * ldcoef + c);
        const float4 c4 = *reinterpret_cast<const float4*>(coef + (size_t)(pass * 8 + 3) * ldcoef + c);
        const float4 h4 = *reinterpret_cast<const float4*>(coef + (size_t)(pass * 8 + 4) * ldcoef + c);
        mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w; sc[0] = c4.x; sc[1] = c4.y; sc[2] = c4.z; sc[3] = c4.w;
        sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
      }
      auto acc1 = [&](const float4 av4, const float4 bv4) {
        const float av[4] = {av4.x, av4.y, av4.z, av4.w}, bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (MODE == 0) s0[e] += av[e];
          else if (MODE == 1) s0[e] += av[e] * fminf(bv[e], 0.f);
          else if (MODE == 2) { s0[e] += av[e]; s1[e] += (double)av[e] * av[e]; }
          else { const float g = bv[e] * (fmaf(av[e], sc[e], sh[e]) >= 0.f ? 1.f : leak); s0[e] += g; s1[e] += (double)g * (av[e] - mu[e]); }
        }
      };
      constexpr bool TWO = MODE == 1 || MODE == 3;
      size_t r = r0 + lr;
      for (; r + 3 * (size_t)lanes < r1; r += 4 * (size_t)lanes) {        // four rows in flight, summed in row order
        float4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = *reinterpret_cast<const float4*>(a + (r + (size_t)u * lanes) * lda + coff + c);
          bv[u] = TWO ? *reinterpret_cast<const float4*>(b + (r + (size_t)u * lanes) * ldb + c) : av[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc1(av[u], bv[u]);
      }
      for (; r < r1; r += lanes) {
        const float4 av = *reinterpret_cast<const float4*>(a + r * lda + coff + c);
        const float4 bv = TWO ? *reinterpret_cast<const float4*>(b + r * ldb + c) : av;
        acc1(av, bv);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][e][threadIdx.x] = s0[e]; red[1][e][threadIdx.x] = s1[e]; }
    __syncthreads();
    if (threadIdx.x < gw) {                              // fixed order over the row lanes
      double t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
      for (int l = 0; l < lanes; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) { t0[e] += red[0][e][l * gw + threadIdx.x]; t1[e] += red[1][e][l * gw + threadIdx.x]; }
      *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + (g0 + threadIdx.x) * 4) = make_float4((float)t0[0], (float)t0[1], (float)t0[2], (float)t0[3]);
      *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + (g0 + threadIdx.x) * 4) = make_float4((float)t1[0], (float)t1[1], (float)t1[2], (float)t1[3]);
    }
    __syncthreads();
  }
}
// out[(pass, which)][c] (+)= sum over the chunks, in double and in a fixed order: 32 outputs per workgroup, 8 chunk lanes each
// (one thread per output walked up to 256 dependent loads: 25 us per call, 2.5 ms of the SEGAN step)
__global__ __launch_bounds__(256) void k_colred_final(const float* __restrict__ part, int chunks_per, int C, int P, float* __restrict__ out, int ldo,
                                                      int accumulate, int nout) {
  __shared__ double red[8][32];
  const int o = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;                     // (pass, which, column); one-output modes write row `pass` only
  const bool in = i < P * 2 * C;
  const int c = in ? i % C : 0, which = in ? (i / C) & 1 : 0, pass = in ? i / (2 * C) : 0;
  double acc = 0.0;
  if (in && which < nout)
    for (int ch = l; ch < chunks_per; ch += 8) acc += (double)part[((size_t)(pass * chunks_per + ch) * 2 + which) * C + c];
  red[l][o] = acc;
  __syncthreads();
  if (l == 0 && in && which < nout) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][o];
    float* op = out + (size_t)(pass * nout + which) * ldo + c;
    *op = accumulate ? *op + (float)t : (float)t;
  }
}
void launch_colred(int mode, const float* a, int lda, int coff, const float* b, int ldb, int C, size_t rows_per, int P, const float* coef, int ldcoef,
                   float leak, float* out, int ldo, bool accumulate, float* scratch, size_t scratch_floats, hipStream_t s) {
  const bool two = mode == 1 || mode == 3;
  // RSRGAN_COLRED_VEC: bit m = the 16-byte form for mode m.  Mode 2 (the VBN statistics) stays on the scalar form by default: the
  // reference's E[h^2] - E[h]^2 (bnorm.py:40-41) is ill-conditioned in fp32 where |mean| >> sigma (the first two blocks on waveform
  // input), and the 16384-sample parity case holds its 2e-3 against the fp64 oracle only with sums that differ from the 16-byte
  // form's (correctly rounded) ones by one ulp -- 9e-4 .. 3e-3 on four tensors otherwise (tests/test_gpu_segan.py)
  static int vecmask = -1;
  if (vecmask < 0) { const char* e = getenv("RSRGAN_COLRED_VEC"); vecmask = e ? atoi(e) : 11; }
  const bool vec = ((vecmask >> mode) & 1) && C % 4 == 0 && lda % 4 == 0 && coff % 4 == 0 && (!two || ldb % 4 == 0) && (mode != 3 || ldcoef % 4 == 0);
  // chunk: ~2048 workgroups per launch, <= 256 partials per (pass, column) (the final sum is one thread per column), scratch permitting
  int chunk = vec ? 64 : 256;
  const int max_chunks = vec ? 256 : 128;
  while (((rows_per + chunk - 1) / chunk) * (size_t)P * 2 * C > scratch_floats || (rows_per + chunk - 1) / chunk > (size_t)max_chunks ||
         (vec && ((rows_per + chunk - 1) / chunk) * (size_t)P > 4096))
    chunk *= 2;
  const int chunks_per = (int)((rows_per + chunk - 1) / chunk);
  dim3 grid(P * chunks_per), block(256);
#define RSR_COLRED(K, M) hipLaunchKernelGGL(K<M>, grid, block, 0, s, a, lda, coff, b, ldb, C, rows_per, chunk, chunks_per, coef, ldcoef, leak, scratch)
  if (vec) {
    if (mode == 0) RSR_COLRED(k_colred_part4, 0);
    else if (mode == 1) RSR_COLRED(k_colred_part4, 1);
    else if (mode == 2) RSR_COLRED(k_colred_part4, 2);
    else RSR_COLRED(k_colred_part4, 3);
  } else {
    if (mode == 0) RSR_COLRED(k_colred_part, 0);
    else if (mode == 1) RSR_COLRED(k_colred_part, 1);
    else if (mode == 2) RSR_COLRED(k_colred_part, 2);
    else RSR_COLRED(k_colred_part, 3);
  }
#undef RSR_COLRED
  hipLaunchKernelGGL(k_colred_final, dim3((P * 2 * C + 31) / 32), dim3(256), 0, s, scratch, chunks_per, C, P, out, ldo, accumulate ? 1 : 0, mode >= 2 ? 2 : 1);
}

// ---- virtual batch norm (utils/bnorm.py).  Pass 0 of a call group is the reference ("dummy") pass; the live passes mix their
// own statistics with the reference's by c = 1 / (B + 1).  coef rows per pass: 0 mu, 1 q (mean of squares), 2 s = (eps + q - mu^2)^-1/2,
// 3 sc = gamma s, 4 sh = beta - mu sc, 5 k1, 6 k2 (backward: dh = g sc + k1 + k2 h), 7 unused
__global__ __launch_bounds__(256) void k_vbn_coef(const float* __restrict__ sums, int lds, int P, int C, float inv_rows, float cnew, float eps,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ ref_coef,
                                                  float* __restrict__ coef, int ldc) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  // ref_coef != null: every pass of this group is live against that reference (the G-run: reference statistics computed before)
  float mref = 0.f, qref = 0.f;
  for (int p = 0; p < P; ++p) {
    float m = sums[(size_t)(p * 2) * lds + c] * inv_rows, q = sums[(size_t)(p * 2 + 1) * lds + c] * inv_rows;
    const bool live = ref_coef != nullptr || p > 0;
    if (ref_coef) { mref = ref_coef[c]; qref = ref_coef[ldc + c]; }
    if (live) { m = cnew * m + (1.f - cnew) * mref; q = cnew * q + (1.f - cnew) * qref; }
    else { mref = m; qref = q; }
    const float sd = 1.f / sqrtf(eps + q - m * m), sc = gamma[c] * sd;
    float* o = coef + (size_t)p * 8 * ldc + c;
    o[0] = m; o[ldc] = q; o[2 * ldc] = sd; o[3 * ldc] = sc; o[4 * ldc] = beta[c] - m * sc;
  }
}
void launch_vbn_coef(const float* sums, int lds, int P, int C, size_t rows_per, int B, float eps, const float* gamma, const float* beta,
                     const float* ref_coef, float* coef, int ldc, hipStream_t s) {
  hipLaunchKernelGGL(k_vbn_coef, dim3((C + 255) / 256), dim3(256), 0, s, sums, lds, P, C, 1.f / (float)rows_per, 1.f / (B + 1.f), eps, gamma, beta,
                     ref_coef, coef, ldc);
}
// y = leaky(h * sc + sh), per pass coefficients
__global__ __launch_bounds__(256) void k_vbn_apply(const float* __restrict__ h, int C, size_t rows_per, const float* __restrict__ coef, int ldc, float leak,
                                                   float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const int p = (int)(r / rows_per);
    const float v = fmaf(h[i], coef[(size_t)(p * 8 + 3) * ldc + c], coef[(size_t)(p * 8 + 4) * ldc + c]);   // (explicit fma: the backward kernels decide the kink with the same rounding)
    y[i] = v >= 0.f ? v : leak * v;
  }
}
void launch_vbn_apply(const float* h, int C, size_t rows_per, int P, const float* coef, int ldc, float leak, float* y, hipStream_t s) {
  const size_t n = rows_per * P * C;
  hipLaunchKernelGGL(k_vbn_apply, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, h, C, rows_per, coef, ldc, leak, y, n);
}
// backward coefficients from S1 = sum g, S2 = sum g (h - mu) per pass (launch_colred mode 3):
//   dL/dmu = -sc S1 + mu s^3 gamma S2 ; dL/dq = -1/2 s^3 gamma S2 ; a live pass hands (1 - c) of both to the reference pass, whose own
//   statistics take them in full; dh = g sc + k1 + k2 h with k1 = dL/dm_batch / rows, k2 = 2 dL/dq_batch / rows.
//   dgamma (+)= sum_p s S2 ; dbeta (+)= sum_p S1.   first_live: index of the first live pass (0: all live, reference constant)
__global__ __launch_bounds__(256) void k_vbn_bwd_coef(const float* __restrict__ sums, int lds, int P, int first_live, int C, float inv_rows, float cnew,
                                                      const float* __restrict__ gamma, float* __restrict__ coef, int ldc, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float dg = 0.f, db = 0.f, dmref = 0.f, dqref = 0.f;
  for (int p = P - 1; p >= 0; --p) {                     // live passes first, the reference pass (p = 0 when first_live = 1) last
    const float S1 = sums[(size_t)(p * 2) * lds + c], S2 = sums[(size_t)(p * 2 + 1) * lds + c];
    float* o = coef + (size_t)p * 8 * ldc + c;
    const float mu = o[0], sd = o[2 * ldc], sc = o[3 * ldc], g = gamma[c];
    const float s3 = sd * sd * sd;
    float dmu = -sc * S1 + mu * s3 * g * S2, dq = -0.5f * s3 * g * S2;
    dg += sd * S2; db += S1;
    const bool live = p >= first_live;
    float dmb, dqb;
    if (live) { dmb = cnew * dmu; dqb = cnew * dq; dmref += (1.f - cnew) * dmu; dqref += (1.f - cnew) * dq; }
    else { dmb = dmu + dmref; dqb = dq + dqref; }
    o[5 * ldc] = dmb * inv_rows; o[6 * ldc] = 2.f * dqb * inv_rows;
  }
  if (dgamma) { dgamma[c] = accumulate ? dgamma[c] + dg : dg; dbeta[c] = accumulate ? dbeta[c] + db : db; }
}
void launch_vbn_bwd_coef(const float* sums, int lds, int P, int first_live, int C, size_t rows_per, int B, const float* gamma, float* coef, int ldc,
                         float* dgamma, float* dbeta, bool accumulate, hipStream_t s) {
  hipLaunchKernelGGL(k_vbn_bwd_coef, dim3((C + 255) / 256), dim3(256), 0, s, sums, lds, P, first_live, C, 1.f / (float)rows_per, 1.f / (B + 1.f), gamma,
                     coef, ldc, dgamma, dbeta, accumulate ? 1 : 0);
}
__global__ __launch_bounds__(256) void k_vbn_bwd_apply(const float* __restrict__ h, const float* __restrict__ dy, int C, size_t rows_per,
                                                       const float* __restrict__ coef, int ldc, float leak, float* __restrict__ dh, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const float* o = coef + (size_t)(r / rows_per) * 8 * ldc + c;
    const float hv = h[i], sc = o[3 * ldc];
    const float g = dy[i] * (fmaf(hv, sc, o[4 * ldc]) >= 0.f ? 1.f : leak);
    dh[i] = g * sc + o[5 * ldc] + o[6 * ldc] * hv;
  }
}
void launch_vbn_bwd_apply(const float* h, const float* dy, int C, size_t rows_per, int P, const float* coef, int ldc, float leak, float* dh, hipStream_t s) {
  const size_t n = rows_per * P * C;
  hipLaunchKernelGGL(k_vbn_bwd_apply, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, h, dy, C, rows_per, coef, ldc, leak, dh, n);
}

// ---- discriminator input: joint[p][b][:] = concat(x[b], tail_p[b]) + noise_p[b]   (segan.py:183-199, discriminator.py:74)
__global__ __launch_bounds__(256) void k_build_joint1(const float* __restrict__ x, int Lx, const float* __restrict__ tail, int U, const float* __restrict__ noise,
                                                      float* __restrict__ joint, size_t n) {
  const int Lj = Lx + U;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t b = i / Lj;
    const int p = (int)(i - b * Lj);
    joint[i] = (p < Lx ? x[b * Lx + p] : tail[b * U + (p - Lx)]) + (noise ? noise[i] : 0.f);
  }
}
void launch_build_joint1(const float* x, int Lx, const float* tail, int U, const float* noise, float* joint, int B, hipStream_t s) {
  const size_t n = (size_t)B * (Lx + U);
  hipLaunchKernelGGL(k_build_joint1, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, x, Lx, tail, U, noise, joint, n);
}

// ---- logits_conv (conv1d, 31 taps, ONE kernel, stride 1, SAME, no bias: discriminator.py:88-91) + squeeze + fully_connected(1):
// out[r][p] = sum_{dk,c} h[r, p + dk - pl, c] W[dk][c] ; logit[r] = sum_p out[r][p] wfc[p] + bfc.  h [R][Ld][C] dense.
__global__ __launch_bounds__(256) void k_dhead_fwd(const float* __restrict__ h, int Ld, int C, int k, const float* __restrict__ W, const float* __restrict__ wfc,
                                                   int ldfc, const float* __restrict__ bfc, float* __restrict__ conv_out, float* __restrict__ logits) {
  const int r = blockIdx.x, pl = (k - 1) / 2;
  __shared__ float red[256];
  float logit = 0.f;
  for (int p = 0; p < Ld; ++p) {
    float acc = 0.f;
    const int d0 = max(0, pl - p), d1 = min(k, Ld + pl - p);       // taps inside the row
    const float* base = h + ((size_t)r * Ld + (p + d0 - pl)) * C;  // contiguous (d1 - d0) * C floats
    const float* w = W + (size_t)d0 * C;
    const int n = (d1 - d0) * C;
    for (int i = threadIdx.x; i < n; i += 256) acc = fmaf(base[i], w[i], acc);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) { conv_out[(size_t)r * Ld + p] = red[0]; logit += red[0] * wfc[(size_t)p * ldfc]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) logits[r] = logit + bfc[0];
}
void launch_dhead_fwd(const float* h, int R, int Ld, int C, int k, const float* W, const float* wfc, int ldfc, const float* bfc, float* conv_out, float* logits,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_dhead_fwd, dim3(R), dim3(256), 0, s, h, Ld, C, k, W, wfc, ldfc, bfc, conv_out, logits);
}
// backward: dconv[r][p] = dlogit[r] wfc[p] ; dwfc[p] = sum_r dlogit[r] conv_out[r][p] ; dbfc = sum_r dlogit[r] ;
// dW[dk][c] = sum_{r,p} dconv[r][p] h[r, p+dk-pl, c] ; dh[r,l,c] = sum_p dconv[r][p] W[l - p + pl][c]
__global__ __launch_bounds__(256) void k_dhead_bwd_small(const float* __restrict__ dlogit, int R, int Ld, const float* __restrict__ conv_out,
                                                         float* __restrict__ dwfc, int ldfc, float* __restrict__ dbfc) {
  const int p = threadIdx.x;
  if (p < Ld) { float a = 0.f; for (int r = 0; r < R; ++r) a += dlogit[r] * conv_out[(size_t)r * Ld + p]; dwfc[(size_t)p * ldfc] = a; }
  if (p == 255) { float a = 0.f; for (int r = 0; r < R; ++r) a += dlogit[r]; dbfc[0] = a; }
}
__global__ __launch_bounds__(256) void k_dhead_bwd_w(const float* __restrict__ dlogit, const float* __restrict__ wfc, int ldfc, const float* __restrict__ h,
                                                     int R, int Ld, int C, int k, float* __restrict__ dW) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= k * C) return;
  const int dk = e / C, c = e - dk * C, pl = (k - 1) / 2;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) {
    const float dl = dlogit[r];
    for (int p = 0; p < Ld; ++p) {
      const int l = p + dk - pl;
      if (l >= 0 && l < Ld) acc = fmaf(dl * wfc[(size_t)p * ldfc], h[((size_t)r * Ld + l) * C + c], acc);
    }
  }
  dW[e] = acc;
}
__global__ __launch_bounds__(256) void k_dhead_bwd_x(const float* __restrict__ dlogit, const float* __restrict__ wfc, int ldfc, const float* __restrict__ W,
                                                     int Ld, int C, int k, float* __restrict__ dh, size_t n) {
  const int pl = (k - 1) / 2;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t rl = i / C;
    const int l = (int)(rl % Ld);
    const size_t r = rl / Ld;
    float acc = 0.f;
    for (int p = 0; p < Ld; ++p) {
      const int dk = l - p + pl;
      if (dk >= 0 && dk < k) acc = fmaf(wfc[(size_t)p * ldfc], W[(size_t)dk * C + c], acc);
    }
    dh[i] = acc * dlogit[r];
  }
}
void launch_dhead_bwd(const float* dlogit, int R, int Ld, int C, int k, const float* h, const float* conv_out, const float* W, const float* wfc, int ldfc,
                      float* dW, float* dwfc, float* dbfc, float* dh, hipStream_t s) {
  if (dW) {
    hipLaunchKernelGGL(k_dhead_bwd_small, dim3(1), dim3(256), 0, s, dlogit, R, Ld, conv_out, dwfc, ldfc, dbfc);
    hipLaunchKernelGGL(k_dhead_bwd_w, dim3((k * C + 255) / 256), dim3(256), 0, s, dlogit, wfc, ldfc, h, R, Ld, C, k, dW);
  }
  const size_t n = (size_t)R * Ld * C;
  hipLaunchKernelGGL(k_dhead_bwd_x, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, dlogit, wfc, ldfc, W, Ld, C, k, dh, n);
}

// ---- losses (segan.py:226-235).  logits [P][B]; mode 0 (D-run): pass 1 = real (target 1), pass 2 = fake (target 0) -> loss3 = {d_rl, d_fk, sum},
// dlogits of pass 0 = 0;  mode 1 (G-run): pass `fake` against target 1 -> loss3[0] = g_adv
__global__ __launch_bounds__(256) void k_segan_lsgan(const float* __restrict__ logits, int B, int mode, int fake_pass, int P, float* __restrict__ dlogits,
                                                     float* __restrict__ loss3) {
  __shared__ float red[2][256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < P * B; i += 256) {
    const int p = i / B;
    float d = 0.f;
    if (mode == 0) {
      if (p == 1) { const float e = logits[i] - 1.f; a += e * e; d = 2.f * e / B; }
      else if (p == 2) { const float e = logits[i]; b += e * e; d = 2.f * e / B; }
    } else if (p == fake_pass) { const float e = logits[i] - 1.f; a += e * e; d = 2.f * e / B; }
    if (dlogits) dlogits[i] = d;
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; } __syncthreads(); }
  if (threadIdx.x == 0) {
    if (mode == 0) { loss3[0] = red[0][0] / B; loss3[1] = red[1][0] / B; loss3[2] = loss3[0] + loss3[1]; }
    else loss3[0] = red[0][0] / B;
  }
}
void launch_segan_lsgan(const float* logits, int B, int mode, int fake_pass, int P, float* dlogits, float* loss3, hipStream_t s) {
  hipLaunchKernelGGL(k_segan_lsgan, dim3(1), dim3(256), 0, s, logits, B, mode, fake_pass, P, dlogits, loss3);
}
// g_l1 = lambda mean|G - labels| ; dG (+)= lambda sign(G - labels) / n ; loss3 = {g_adv (given), g_l1, sum}
__global__ __launch_bounds__(256) void k_segan_l1(const float* __restrict__ G, const float* __restrict__ lab, int n, const float* __restrict__ lambda,
                                                  float* __restrict__ dG, int accumulate, float* __restrict__ loss3) {
  __shared__ float red[256];
  const float lam = lambda[0];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float e = G[i] - lab[i];
    a += fabsf(e);
    if (dG) { const float d = lam * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) / n; dG[i] = accumulate ? dG[i] + d : d; }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) { loss3[1] = lam * red[0] / n; loss3[2] = loss3[0] + loss3[1]; }
}
void launch_segan_l1(const float* G, const float* lab, int n, const float* lambda, float* dG, bool accumulate, float* loss3, hipStream_t s) {
  hipLaunchKernelGGL(k_segan_l1, dim3(1), dim3(256), 0, s, G, lab, n, lambda, dG, accumulate ? 1 : 0, loss3);
}

// out[0] = sum of src[rows][cols]: 256 block partials, then one block (fixed order)
__global__ __launch_bounds__(256) void k_sum_all(const float* __restrict__ src, size_t n, int cols, int ld, float* __restrict__ out, int final_pass) {
  __shared__ float red[256];
  float a = 0.f;
  if (final_pass) { for (size_t i = threadIdx.x; i < n; i += 256) a += src[i]; }
  else for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += src[(i / cols) * ld + (i % cols)];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) out[final_pass ? 0 : blockIdx.x] = red[0];
}
void launch_sum_all(const float* src, int rows, int cols, int ld, float* out, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_all, dim3(256), dim3(256), 0, s, src, (size_t)rows * cols, cols, ld, scratch, 0);
  hipLaunchKernelGGL(k_sum_all, dim3(1), dim3(256), 0, s, scratch, (size_t)256, 1, 1, out, 1);
}

// ---- tf.train.RMSPropOptimizer(lr): ms = 0.9 ms + 0.1 g^2 ; w -= lr g / sqrt(ms + 1e-10)   (segan.py:123-124; TF 1.4 ApplyRMSProp, momentum 0)
__global__ __launch_bounds__(256) void k_rmsprop(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ ms, const float* __restrict__ lr,
                                                 float decay, float eps, size_t n) {
  const float l = lr[0];
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gv = g[i];
    const float m = decay * ms[i] + (1.f - decay) * gv * gv;
    ms[i] = m;
    w[i] -= l * gv / sqrtf(m + eps);
  }
}
void launch_rmsprop(float* w, const float* g, float* ms, const float* lr, float decay, float eps, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_rmsprop, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, w, g, ms, lr, decay, eps, n);
}

}  // namespace rsr
