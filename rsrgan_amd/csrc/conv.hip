// conv.hip -- implicit-GEMM conv2d for the R-CED generator (models/rced.py:90-102): kernel [S, fw] with S = the whole height,
// stride 1, SAME, NHWC.  No patch matrix is materialised: a workgroup keeps all S rows of a TW-column strip of one frame
// (plus the fw-1 halo columns) in LDS, and because the channel axis is innermost, the (dw, ci) part of a patch is a CONTIGUOUS
// window of an image row: A[m][k] = row[h_m + dh - pt][(wl_m)*C' + k], k in [0, fw*C').  C' = C rounded so that C'/4 is odd:
// the 16 lanes of an MFMA operand then start 4*odd floats apart and their float4 reads hit 16 disjoint bank groups.  The filter is
// pre-arranged by k_conv_prep as Ft[dh][co][k'] (k' = dw*C' + ci, zero rows at the pads), so both MFMA operands are k-contiguous
// float4 (v_mfma_f32_16x16x4_f32, four MFMAs per float4 pair) exactly as in the recurrence kernels.
// The same kernel computes the data gradient: d(in) = conv_SAME(d(out), flipped filter with in/out channels swapped).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace rsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int conv_cpad(int C) { const int c4 = (C + 3) & ~3; return ((c4 / 4) & 1) ? c4 : c4 + 4; }   // C'/4 odd (C = 1: one real channel + 3 zero ones)
__host__ __device__ inline int conv_kp(int fw, int C) { return (fw * conv_cpad(C) + 15) / 16 * 16; }   // k' extent, whole k-blocks
__host__ __device__ inline int conv_ldf(int fw, int C) { const int k = conv_kp(fw, C) + 4; return ((k / 4) & 1) ? k : k + 4; }

// Ft[dh][n][k'] for n < 32: n = output channel, k' = dw*C' + c.  flip = 0: Ft = F[dh][dw][c][n] (forward, C = Cin, N = Cout);
// flip = 1: Ft = F[S-1-dh][fw-1-dw][n][c] (data gradient: C = Cout of the layer, N = Cin).  F is the [S*fw*Cin][ldf_src] operand.
__global__ void k_conv_prep(const float* __restrict__ F, int ldf_src, int S, int fw, int Cin, int Cout, int flip, float* __restrict__ Ft) {
  const int C = flip ? Cout : Cin, N = flip ? Cin : Cout;
  const int Cp = conv_cpad(C), ldf = conv_ldf(fw, C);
  const size_t total = (size_t)S * 32 * ldf;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kp = (int)(i % ldf);
    const int n = (int)((i / ldf) % 32);
    const int dh = (int)(i / ((size_t)ldf * 32));
    const int dw = kp / Cp, c = kp - dw * Cp;
    float v = 0.f;
    if (n < N && dw < fw && c < C) {
      if (!flip) v = F[((size_t)(dh * fw + dw) * Cin + c) * ldf_src + n];
      else v = F[((size_t)((S - 1 - dh) * fw + (fw - 1 - dw)) * Cin + n) * ldf_src + c];
    }
    Ft[i] = v;
  }
}

// grid: (strips per frame, frames).  512 threads = 8 waves; wave w owns the 16-position tiles w, w+8, w+16, .. (interleaved, so every
// wave gets its share of the edge rows whose padding tiles are skipped) of the strip's
// S*TW positions and both 16-column halves of the (<= 32) output channels.
template <int RT, int NT>
__global__ __launch_bounds__(512) void k_conv_fwd(const float* __restrict__ in, int ldc_in, int C, const float* __restrict__ Ft,
                                                  const float* __restrict__ bias, int relu, float* __restrict__ out, int ldc_out, int N,
                                                  int S, int W, int fw, int TW, const float* __restrict__ mask, int wbase, int wend, int FB, int R) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Cp = conv_cpad(C), nkb = conv_kp(fw, C) / 16, ldf = conv_ldf(fw, C);
  const int pt = (S - 1) / 2, pl = (fw - 1) / 2;
  const int rowlen = (TW + fw - 1) * Cp + 16;             // +16: the last k-block of the last position may run past its window
  // FB frames per workgroup (1 but for narrow remainder strips: a one-column strip of ONE frame is 11 positions): image rows
  // [fb][h], ONE zero row ZR = FB*S behind them; position m = (fb*S + h)*TW + wl
  const int ZR = FB * S;
  float* img = smem;                                      // [FB*S + 1][rowlen], row ZR = zeros
  float* fts = smem + (size_t)(ZR + 1) * rowlen;          // [32][ldf] filter slice of one dh
  const int r0 = blockIdx.y * FB, w0 = wbase + blockIdx.x * TW;      // this launch covers the columns [wbase, wend)
  const int tw = min(TW, wend - w0);
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  // image strip -> LDS: position x of a row is image column w0 - pl + x; every slot (halo, channel pads, zero row) is written
  {
    // six loads in flight per thread, UNCONDITIONAL from clamped / stand-in addresses, the zeros selected afterwards (a load under
    // a branch in a loop with a run-time bound is one global round trip per iteration: up to 13 of them per workgroup)
    const int cp4 = Cp / 4, row4 = rowlen / 4;
    const int total4 = (ZR + 1) * row4;
    constexpr int SU = 6;
    for (int i0 = tid; i0 < total4; i0 += 512 * SU) {
      float4 v[SU];
      bool ok[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = min(i0 + u * 512, total4 - 1);
        const int hr = i / row4, e = i - hr * row4;          // hr = fb*S + h: frames are consecutive in `in`, so is (r0*S + hr)
        const int x = e / cp4, c = (e - x * cp4) * 4;
        const int wcol = w0 - pl + x;
        ok[u] = hr < ZR && r0 * S + hr < R * S && x < TW + fw - 1 && wcol >= 0 && wcol < W && c < C;
        const float* src = ok[u] ? in + ((size_t)(r0 * S + hr) * W + wcol) * ldc_in + c : in;
        v[u] = *reinterpret_cast<const float4*>(src);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + u * 512;
        const float4 z = make_float4(ok[u] ? v[u].x : 0.f, ok[u] ? v[u].y : 0.f, ok[u] ? v[u].z : 0.f, ok[u] ? v[u].w : 0.f);
        if (i < total4) *reinterpret_cast<float4*>(img + (size_t)i * 4) = z;
      }
    }
  }
  // per-lane position of each row tile: m = (wv*RT + i)*16 + lr -> (h, wl); positions >= S*tw are parked on the zero row
  const int M = ZR * TW;
  int ph[RT], pofs[RT], pfr[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int m = (i * 8 + wv) * 16 + lr;
    const int hr = m / TW, wl = m - hr * TW;
    const int fb = hr / S, h = hr - fb * S;
    const bool ok = m < M && wl < tw && r0 + fb < R;
    ph[i] = ok ? h : -1000;
    pfr[i] = fb * S;
    pofs[i] = wl * Cp + 4 * q;
  }
  f32x4 acc[RT][NT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The filter slice of row dh+1 travels to registers while row dh is multiplied (<= 5 float4 per thread: 32 x ldf <= 32 x 280
  // floats over 512 threads; checked on the host): staging it synchronously between two barriers cost ~2 us of each of the S rows,
  // 17 % of the kernel at S = 11 (SQ_VALU_MFMA_BUSY_CYCLES 69 % of the CU-cycles, profiles/r3_rced_pmc.txt).
  constexpr int FSL = 5;
  const int n4 = NT * 16 * ldf / 4;
  // (named scalars: hipcc leaves a float4 array that lives across the row loop in scratch memory)
  static_assert(FSL == 5, "five named slots below");
  const int fi0 = min(tid, n4 - 1), fi1 = min(tid + 512, n4 - 1), fi2 = min(tid + 1024, n4 - 1), fi3 = min(tid + 1536, n4 - 1),
            fi4 = min(tid + 2048, n4 - 1);
  float4 fr0, fr1, fr2, fr3, fr4;
  {
    const float4* src = reinterpret_cast<const float4*>(Ft);
    fr0 = src[fi0]; fr1 = src[fi1]; fr2 = src[fi2]; fr3 = src[fi3]; fr4 = src[fi4];
  }
  for (int dh = 0; dh < S; ++dh) {
    __syncthreads();                                      // image ready (first pass) / previous filter slice consumed
    {
      float4* dst = reinterpret_cast<float4*>(fts);
      if (tid < n4) dst[tid] = fr0;
      if (tid + 512 < n4) dst[tid + 512] = fr1;
      if (tid + 1024 < n4) dst[tid + 1024] = fr2;
      if (tid + 1536 < n4) dst[tid + 1536] = fr3;
      if (tid + 2048 < n4) dst[tid + 2048] = fr4;
    }
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(Ft + (size_t)min(dh + 1, S - 1) * 32 * ldf);      // (the last row re-reads itself)
      fr0 = src[fi0]; fr1 = src[fi1]; fr2 = src[fi2]; fr3 = src[fi3]; fr4 = src[fi4];
    }
    // image rows h with 0 <= h + dh - pt < S take part in this filter row: positions [vlo, vhi) of the strip; a 16-position tile
    // outside that range only multiplies the zero row -- skipped (wave-uniform: 25 % of the MFMAs at S = 11)
    const int vlo = max(0, pt - dh) * TW, vhi = min(S, S + pt - dh) * TW;
    const float* arow[RT];
    bool live[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int hh = ph[i] + dh - pt;
      arow[i] = img + (size_t)((hh >= 0 && hh < S) ? pfr[i] + hh : ZR) * rowlen + pofs[i];
      const int t0 = (i * 8 + wv) * 16;
      live[i] = FB > 1 ? t0 < M : (t0 + 15 >= vlo && t0 < vhi);       // (the row ranges below are one frame's)
    }
    const float* b0 = fts + (size_t)lr * ldf + 4 * q;
    const float* b1 = b0 + (size_t)16 * ldf;
    // (fragments stay single 16-byte reads although their odd row strides collide on half of the LDS cycles: two conflict-free
    // 8-byte reads per fragment measured 629 vs 608 ms per step of the R-CED variant -- the loop is instruction-, not LDS-bound)
    for (int kb = 0; kb < nkb; ++kb) {
      float4 bv[NT];
      bv[0] = *reinterpret_cast<const float4*>(b0 + kb * 16);
      if (NT == 2) bv[NT - 1] = *reinterpret_cast<const float4*>(b1 + kb * 16);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        if (!live[i]) continue;
        const float4 av = *reinterpret_cast<const float4*>(arow[i] + kb * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[j].w, acc[i][j], 0, 0, 0);
      }
    }
  }
  // epilogue: C/D map of the 16x16 MFMA: row = 4*(lane>>4) + e, col = lane&15
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = j * 16 + lr;
      if (co >= N) continue;
      const float bv = bias ? bias[co] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = (i * 8 + wv) * 16 + 4 * q + e;
        const int h = m / TW, wl = m - h * TW;              // h = fb*S + row: frames are consecutive in `out`
        if (m >= M || wl >= tw || r0 * S + h >= R * S) continue;
        float v = acc[i][j][e] + bv;
        if (relu) v = fmaxf(v, 0.f);
        const size_t oi = ((size_t)(r0 * S + h) * W + w0 + wl) * ldc_out + co;
        if (mask && !(mask[oi] > 0.f)) v = 0.f;       // data gradient: relu' of the layer below, its activations laid out like `out`
        out[oi] = v;
      }
    }
}

// Weight gradient without a patch matrix: dF[dh][k'][co] = sum over positions of A[m][dh, k'] * d[m][co].  One workgroup owns DH
// consecutive filter rows for a group of frames (and one column strip): per frame it stages the image strip and the d strip in
// LDS ONCE and accumulates its DH [K' x 32] tiles in registers (wave w: k'-tiles w*KT.., both output-channel halves; the MFMA k
// axis runs over positions; per filter row only the 16-position blocks that touch non-padding rows).  Partials
// [group][strip][dh][K'][32] are summed in a fixed order by k_conv_wgrad_red, which also undoes the k' padding.
template <int KT, int NT, int NWV, int DH>
__global__ __launch_bounds__(64 * NWV) void k_conv_wgrad(const float* __restrict__ in, int ldc_in, int C, const float* __restrict__ d, int ldc_d,
                                                         int N, float* __restrict__ part, int S, int W, int fw, int TW, int R, int fpg,
                                                         float* __restrict__ bpart) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Cp = conv_cpad(C), KP = conv_kp(fw, C);
  const int pt = (S - 1) / 2, pl = (fw - 1) / 2;
  const int rowlen = (TW + fw - 1) * Cp + 16;
  const int MP = (S * TW + 15) / 16 * 16;                 // positions padded to whole 16-blocks
  float* img = smem;                                      // [S + 1][rowlen], row S = zeros
  float* ds = smem + (size_t)(S + 1) * rowlen;            // [MP][36] gradient strip (4 positions = 144 floats apart: 16-bank steps)
  int* tab = reinterpret_cast<int*>(ds + (size_t)MP * 36); // [MP] h*rowlen + wl*C': LDS offset of position m's window in its own image row
  int* gtab = tab + MP;                                   // [MP] offset of position m in one frame of d (-1: outside the strip)
  // filter rows of this workgroup: dh0 + dd * dstep, INTERLEAVED over the row groups (rows near the middle meet the fewest padding
  // positions: contiguous groups {0..5} / {6..10} of an 11-row filter carry 51 / 40 row-blocks, interleaved 48 / 43)
  const int dh0 = blockIdx.x, dstep = gridDim.x, grp = blockIdx.y, strip = blockIdx.z;
  const int w0 = strip * TW, tw = min(TW, W - w0);
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc[DH][KT][NT];
#pragma unroll
  for (int dd = 0; dd < DH; ++dd)
#pragma unroll
    for (int i = 0; i < KT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[dd][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int M = S * TW;
  for (int m = tid; m < MP; m += 64 * NWV) {
    const int h = m / TW, wl = m - h * TW;
    if constexpr (DH == 1) {        // one filter row per workgroup: the final window offset (zero row for padding) is tabulated once
      const int hh = h + dh0 - pt;
      tab[m] = ((m < M && hh >= 0 && hh < S) ? hh : S) * rowlen + wl * Cp;
    } else {
      tab[m] = (m < M ? h : 0) * rowlen + wl * Cp;
    }
    gtab[m] = (m < M && wl < tw) ? (h * W + w0 + wl) * ldc_d : -1;
  }
  // staging map of this thread, frame independent: float4 slot e of every image row (x = strip column, c = channel)
  const int cp4 = Cp / 4, row4 = rowlen / 4;
  const int e0 = tid < row4 ? tid : -1;
  const int sx = e0 >= 0 ? e0 / cp4 : 0, sc = e0 >= 0 ? (e0 - sx * cp4) * 4 : 0;
  const int swcol = w0 - pl + sx;
  const bool s_in = e0 >= 0 && sx < TW + fw - 1 && swcol >= 0 && swcol < W && sc < C;
  __syncthreads();                                        // tab / gtab visible
  bool kt_ok[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i) kt_ok[i] = (wv + NWV * i) * 16 < KP;      // k'-tile wv + NWV*i: a second tile only where the first round left some
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);             // this thread's channels 4 (tid & 7) .. + 3 over its positions of every frame
  for (int f = 0; f < fpg; ++f) {
    const int r = grp * fpg + f;
    if (r >= R) break;
    __syncthreads();
    {
      // image strip: thread e0 owns float4 slot e0 of every row (row4 <= 512 is checked on the host); rows are independent loads
      // All loads of a frame are in flight together: six image rows per batch and up to four gradient slots, UNCONDITIONAL from
      // clamped / stand-in addresses, the zeros selected afterwards.  (Round 2's loops loaded under a branch inside a loop with a
      // run-time bound: one global round trip per image row, 12 per frame -- 40 % of the kernel's time at W = 257.)
      const float* dsrc = d + (size_t)r * S * W * ldc_d;
      constexpr int DSLOTS = 4;                              // MP * 8 <= 4 * 1024 (checked on the host: conv_wgrad_supported)
      float4 dv[DSLOTS];
      int dgo[DSLOTS];
      // (bias gradient = column sums of d: the row-group-0 workgroups see every gradient element of their frames exactly once on its
      //  way to LDS -- a separate tall column sum re-read the 2.3 GB tensor per layer, 9 ms of the 608 ms step)
#pragma unroll
      for (int u = 0; u < DSLOTS; ++u) {                     // 8 float4 = 32 channels per position
        const int i = tid + u * 64 * NWV;
        const int m = min(i >> 3, MP - 1), c = (i & 7) * 4;
        dgo[u] = (i < MP * 8 && c < N) ? gtab[m] : -1;
        dv[u] = *reinterpret_cast<const float4*>(dsrc + (dgo[u] >= 0 ? dgo[u] + c : 0));
      }
      if (e0 >= 0) {
        const float* src = s_in ? in + ((size_t)r * S * W + swcol) * ldc_in + sc : in;
        const size_t hs = s_in ? (size_t)W * ldc_in : 0;
        for (int h0 = 0; h0 <= S; h0 += 6) {
          float4 v[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)min(h0 + u, S - 1) * hs);
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int h = h0 + u;
            const bool ok = s_in && h < S;
            const float4 z = make_float4(ok ? v[u].x : 0.f, ok ? v[u].y : 0.f, ok ? v[u].z : 0.f, ok ? v[u].w : 0.f);
            if (h <= S) *reinterpret_cast<float4*>(img + (size_t)h * rowlen + (size_t)e0 * 4) = z;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < DSLOTS; ++u) {
        const int i = tid + u * 64 * NWV;
        const int m = i >> 3, c = (i & 7) * 4;
        const bool ok = dgo[u] >= 0;
        const float4 z = make_float4(ok ? dv[u].x : 0.f, ok ? dv[u].y : 0.f, ok ? dv[u].z : 0.f, ok ? dv[u].w : 0.f);
        if (i < MP * 8) *reinterpret_cast<float4*>(ds + (size_t)m * 36 + c) = z;
        bsum.x += z.x; bsum.y += z.y; bsum.z += z.z; bsum.w += z.w;
      }
    }
    __syncthreads();
    if (!kt_ok[0]) continue;          // (wave-uniform) a wave without a k'-tile only stages: its MFMAs would take matrix-pipe time from the others
    if constexpr (DH > 1) {
      // Several filter rows per workgroup, the rows INSIDE the loop over the 16-position blocks: the gradient operand (and the
      // window-offset table entries) of a block are read once for all DH rows -- 8 + 4 + 4*DH scalar LDS reads per 8*DH MFMAs
      // (KT = 1, NT = 2) instead of 16 per 8 -- and a frame is staged once per DH rows.  No explicit software pipeline: the four
      // waves of a SIMD cover each other's LDS latency, and the registers go to the DH accumulator sets.
      int vlo[DH], vhi[DH], sh[DH];
      int lo = MP, hi = 0;
#pragma unroll
      for (int dd = 0; dd < DH; ++dd) {
        const int dh = dh0 + dd * dstep;
        vlo[dd] = max(0, pt - dh) * TW; vhi[dd] = dh < S ? min(S, S + pt - dh) * TW : 0;      // (rows past S: empty range)
        sh[dd] = (dh - pt) * rowlen;
        if (vhi[dd] > vlo[dd]) { lo = min(lo, vlo[dd] / 16); hi = max(hi, min(MP / 16, (vhi[dd] + 15) / 16)); }
      }
      const int zoff = S * rowlen;
      const int a_lane = wv * 16 + lr, b_lane = 4 * q * 36 + lr, t_lane = 4 * q;
      const float* dsl = ds + b_lane;
      const float* imgl = img + a_lane;
      // software pipeline inside the wave: the gradient operand / table entries of block mb+1 and the image operand of row dd+1
      // are requested before the MFMAs of (mb, dd) issue (loads are unconditional -- rows that only meet padding read the zero
      // row -- so that nothing but the MFMAs sits under a branch)
      float b_n[4][NT];
      int tb_n[4];
      auto load_b = [&](int mb, float (&b)[4][NT], int (&tb)[4]) {
        const int mbc = min(mb, hi - 1);
        const float* dp = dsl + mbc * (16 * 36);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tb[j] = tab[mbc * 16 + t_lane + j];
#pragma unroll
          for (int n = 0; n < NT; ++n) b[j][n] = dp[j * 36 + n * 16];
        }
      };
      auto load_a = [&](int mb, int dd, const int (&tb)[4], float (&a)[4][KT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mb * 16 + t_lane + j;
          const int off = (m >= vlo[dd] && m < vhi[dd]) ? tb[j] + sh[dd] : zoff;
#pragma unroll
          for (int i = 0; i < KT; ++i) a[j][i] = imgl[off + (kt_ok[i] ? i * NWV * 16 : 0)];
        }
      };
      load_b(lo, b_n, tb_n);
      for (int mb = lo; mb < hi; ++mb) {
        float b[4][NT], a_n[4][KT];
        int tb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tb[j] = tb_n[j];
#pragma unroll
          for (int n = 0; n < NT; ++n) b[j][n] = b_n[j][n];
        }
        load_a(mb, 0, tb, a_n);
        load_b(mb + 1, b_n, tb_n);
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) {
          float a[4][KT];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < KT; ++i) a[j][i] = a_n[j][i];
          if (dd + 1 < DH) load_a(mb, dd + 1, tb, a_n);
          __builtin_amdgcn_sched_barrier(0);
          if (mb * 16 >= vhi[dd] || mb * 16 + 16 <= vlo[dd]) continue;           // (uniform) the block only meets padding rows of this filter row
#pragma unroll
          for (int i = 0; i < KT; ++i) {
            if (!kt_ok[i]) continue;                                                // (wave-uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int n = 0; n < NT; ++n) acc[dd][i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], b[j][n], acc[dd][i][n], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) {
      const int dh = dh0 + dd * dstep;
      if (dh >= S) break;
      // positions whose patch row dh is not padding: [vlo, vhi) -> blocks [mb_lo, nmb); the rest would add zeros
      const int vlo = max(0, pt - dh) * TW, vhi = min(S, S + pt - dh) * TW;
      const int mb_lo = vlo / 16, nmb = min(MP / 16, (vhi + 15) / 16);
      // software pipeline over the 16-position blocks: window offsets two blocks ahead, operands one block ahead, MFMAs now
      // (this lane's four positions on the MFMA k axis are m = mb*16 + 4q + j)
      // address arithmetic is kept to a handful of 32-bit VALU ops per block (the loop is VALU-, not MFMA-bound otherwise):
      // lane constants + block-linear terms; a position outside [vlo, vhi) reads the zero row
      int off_n[4];
      float a_c[4][KT], b_c[4][NT], a_n[4][KT], b_n[4][NT];
      const int shift = (dh - pt) * rowlen, zoff = S * rowlen;
      const int a_lane = wv * 16 + lr, b_lane = 4 * q * 36 + lr, t_lane = 4 * q;
      const float* dsl = ds + b_lane;
      const float* imgl = img + a_lane;
      auto load_off = [&](int mb, int (&off)[4]) {
        const int m0 = min(mb, nmb - 1) * 16 + t_lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = m0 + j;
          if constexpr (DH == 1) off[j] = tab[m];
          else off[j] = (m >= vlo && m < vhi) ? tab[m] + shift : zoff;
        }
      };
      auto load_ops = [&](int mb, const int (&off)[4], float (&a)[4][KT], float (&b)[4][NT]) {
        const float* dp = dsl + min(mb, nmb - 1) * (16 * 36);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int n = 0; n < NT; ++n) b[j][n] = dp[j * 36 + n * 16];
#pragma unroll
          for (int i = 0; i < KT; ++i) a[j][i] = kt_ok[i] ? imgl[off[j] + i * NWV * 16] : 0.f;
        }
      };
      auto mma = [&](const float (&a)[4][KT], const float (&b)[4][NT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < KT; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[dd][i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], b[j][n], acc[dd][i][n], 0, 0, 0);
      };
      {
        int off0[4];
        load_off(mb_lo, off0);
        load_ops(mb_lo, off0, a_c, b_c);
        load_off(mb_lo + 1, off_n);
      }
      for (int mb = mb_lo; mb < nmb; mb += 2) {
        int off_nn[4];
        load_ops(mb + 1, off_n, a_n, b_n);
        load_off(mb + 2, off_nn);
        mma(a_c, b_c);
        if (mb + 1 < nmb) {
          load_ops(mb + 2, off_nn, a_c, b_c);
          load_off(mb + 3, off_n);
          mma(a_n, b_n);
        }
      }
    }
    }   // (rows-outside form)
  }
  if (bpart && blockIdx.x == 0) {                          // (uniform) fixed-order sum over the 64 * NWV / 8 threads of a channel group
    __syncthreads();
    float4* sb = reinterpret_cast<float4*>(smem);
    sb[tid] = bsum;
    __syncthreads();
    if (tid < 8) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < 8 * NWV; ++j) { const float4 v = sb[tid + 8 * j]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      *reinterpret_cast<float4*>(bpart + ((size_t)grp * gridDim.z + strip) * 32 + 4 * tid) = t;
    }
  }
  // partial tiles -> part[((grp*nstrips + strip)*S + dh)][k'][32]
#pragma unroll
  for (int dd = 0; dd < DH; ++dd) {
    const int dh = dh0 + dd * dstep;
    if (dh >= S) break;
    float* po = part + ((size_t)(grp * gridDim.z + strip) * S + dh) * (size_t)KP * 32;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      const int kt = wv + NWV * i;
      if (kt * 16 >= KP) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) po[(size_t)(kt * 16 + 4 * q + e) * 32 + j * 16 + lr] = j < NT ? acc[dd][i][j < NT ? j : 0][e] : 0.f;
    }
  }
}
// ---- weight gradient on v_mfma_f32_4x4x1_16B_f32 (output widths 12, 20, 24: no padded MFMA columns, see k_conv_fwd4) ----
// D[lane][i] = A[4*abid + i] * B[lane]: the gradient vector of ONE position in the A slot (lane l holds d[m][channel l]: one LDS
// read serves every channel group), 64 consecutive k' of that position's window in the B slot (a contiguous run of the image row:
// conflict-free 4-byte reads) -> lane = filter element k', registers = four output channels.  The MFMA k axis is ONE position, so
// there are no padded positions at all: a wave walks exactly the rows a filter row touches and the columns of the strip.
// Wave roles (NWV = 8, 10 or 12 waves, chosen so that every wave has a role and the accumulators fit): k' group (64 k' each) x row set (RW of the workgroup's DH filter rows) x position part (columns wl = ps,
// ps + PS, ..); waves beyond NKG * NRS * PS only stage.  Every (row set, position part) writes its own partial slot (the reducer
// sums them in a fixed order).  Same staging, LDS layout and bias sums as k_conv_wgrad.
template <int NCG, int RW, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_conv_wgrad4(const float* __restrict__ in, int ldc_in, int C, const float* __restrict__ d, int ldc_d,
                                                      int N, float* __restrict__ part, int S, int W, int fw, int TW, int R, int fpg,
                                                      float* __restrict__ bpart, int DH, int NKG, int PS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Cp = conv_cpad(C), KP = conv_kp(fw, C);
  const int pt = (S - 1) / 2, pl = (fw - 1) / 2;
  const int rowlen = (TW + fw - 1) * Cp + 16;
  const int MP = (S * TW + 15) / 16 * 16;
  float* img = smem;                                      // [S + 1][rowlen], row S = zeros
  float* ds = smem + (size_t)(S + 1) * rowlen;            // [MP][36] gradient strip
  int* gtab = reinterpret_cast<int*>(ds + (size_t)MP * 36) + MP;      // [MP] offset of position m in one frame of d (-1: outside the strip)
  const int dh0 = blockIdx.x, dstep = gridDim.x, grp = blockIdx.y, strip = blockIdx.z;
  const int w0 = strip * TW, tw = min(TW, W - w0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NRS = (DH + RW - 1) / RW;
  const int kg = wv % NKG, rs = (wv / NKG) % NRS, ps = wv / (NKG * NRS);
  const bool active = ps < PS;
  f32x4 acc[RW][NCG];
#pragma unroll
  for (int j = 0; j < RW; ++j)
#pragma unroll
    for (int c = 0; c < NCG; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int M = S * TW;
  for (int m = tid; m < MP; m += 64 * NWV) {
    const int h = m / TW, wl = m - h * TW;
    gtab[m] = (m < M && wl < tw) ? (h * W + w0 + wl) * ldc_d : -1;
  }
  const int cp4 = Cp / 4, row4 = rowlen / 4;
  const int e0 = tid < row4 ? tid : -1;
  const int sx = e0 >= 0 ? e0 / cp4 : 0, sc = e0 >= 0 ? (e0 - sx * cp4) * 4 : 0;
  const int swcol = w0 - pl + sx;
  const bool s_in = e0 >= 0 && sx < TW + fw - 1 && swcol >= 0 && swcol < W && sc < C;
  __syncthreads();
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nwl = active ? (tw - ps + PS - 1) / PS : 0;      // this wave's columns of a strip row
  // lane -> filter element WITHOUT the channel pads of the LDS image (C' = 20 for 16 channels: 13 x 20 = 260 k' would need a fifth
  // 64-lane group for 4 lanes): kq = dw*C + ci  ->  window offset dw*C' + ci, which is also its row in the partial slot
  const int kq = 64 * kg + lane;
  const int kdw = kq / C, loff = min(kdw, fw - 1) * Cp + (kq - kdw * C);
  for (int f = 0; f < fpg; ++f) {
    const int r = grp * fpg + f;
    if (r >= R) break;
    __syncthreads();
    {
      // image strip: thread e0 owns float4 slot e0 of every row (row4 <= 512 is checked on the host); rows are independent loads
      // All loads of a frame are in flight together: six image rows per batch and up to four gradient slots, UNCONDITIONAL from
      // clamped / stand-in addresses, the zeros selected afterwards.  (Round 2's loops loaded under a branch inside a loop with a
      // run-time bound: one global round trip per image row, 12 per frame -- 40 % of the kernel's time at W = 257.)
      const float* dsrc = d + (size_t)r * S * W * ldc_d;
      constexpr int DSLOTS = (64 + NWV - 1) / NWV;            // MP * 8 <= 4096 <= DSLOTS * 64 * NWV
      float4 dv[DSLOTS];
      int dgo[DSLOTS];
      // (bias gradient = column sums of d: the row-group-0 workgroups see every gradient element of their frames exactly once on its
      //  way to LDS -- a separate tall column sum re-read the 2.3 GB tensor per layer, 9 ms of the 608 ms step)
#pragma unroll
      for (int u = 0; u < DSLOTS; ++u) {                     // 8 float4 = 32 channels per position
        const int i = tid + u * 64 * NWV;
        const int m = min(i >> 3, MP - 1), c = (i & 7) * 4;
        dgo[u] = (i < MP * 8 && c < N) ? gtab[m] : -1;
        dv[u] = *reinterpret_cast<const float4*>(dsrc + (dgo[u] >= 0 ? dgo[u] + c : 0));
      }
      if (e0 >= 0) {
        const float* src = s_in ? in + ((size_t)r * S * W + swcol) * ldc_in + sc : in;
        const size_t hs = s_in ? (size_t)W * ldc_in : 0;
        for (int h0 = 0; h0 <= S; h0 += 6) {
          float4 v[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)min(h0 + u, S - 1) * hs);
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int h = h0 + u;
            const bool ok = s_in && h < S;
            const float4 z = make_float4(ok ? v[u].x : 0.f, ok ? v[u].y : 0.f, ok ? v[u].z : 0.f, ok ? v[u].w : 0.f);
            if (h <= S) *reinterpret_cast<float4*>(img + (size_t)h * rowlen + (size_t)e0 * 4) = z;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < DSLOTS; ++u) {
        const int i = tid + u * 64 * NWV;
        const int m = i >> 3, c = (i & 7) * 4;
        const bool ok = dgo[u] >= 0;
        const float4 z = make_float4(ok ? dv[u].x : 0.f, ok ? dv[u].y : 0.f, ok ? dv[u].z : 0.f, ok ? dv[u].w : 0.f);
        if (i < MP * 8) *reinterpret_cast<float4*>(ds + (size_t)m * 36 + c) = z;
        bsum.x += z.x; bsum.y += z.y; bsum.z += z.z; bsum.w += z.w;
      }
    }
    __syncthreads();
    if (!active) continue;                                 // (wave-uniform)
    // The filter rows of this wave INSIDE the loop over positions: the gradient vector of a position is read once for all of them
    // (1 + RW LDS reads per RW * NCG products instead of 2 per NCG -- with 12-24 output channels the kernel was bound by the LDS
    // pipe, not by the matrix pipe: SQ_VALU_MFMA_BUSY 0.35-0.53 of the CU cycles, profiles/r3_rced_pmc_final.txt).  Which rows
    // an image row h meets is uniform: one branch per row and four-position step.
    // (measured and not kept: a single position stream per filter row with scalar carries and operands requested a step ahead, 560
    //  vs 535 ms per step; the last 1-3 columns of a row as one zero-padded step, 550; their operands requested together, +-0;
    //  two operand sets inside a row -- step s + 1 requested before the products of step s -- in this rows-inside form: 476 vs 457)
    int dhj[RW], sh[RW];
    bool rowok[RW];
    int hlo = S, hhi = 0;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int dd = rs * RW + j;
      dhj[j] = dh0 + dd * dstep;
      rowok[j] = dd < DH && dhj[j] < S;
      sh[j] = (dhj[j] - pt) * rowlen;
      if (rowok[j]) { hlo = min(hlo, max(0, pt - dhj[j])); hhi = max(hhi, min(S, S + pt - dhj[j])); }
    }
    const int sa = PS * 36, sb = PS * Cp;
    for (int h = hlo; h < hhi; ++h) {
      bool on[RW];
#pragma unroll
      for (int j = 0; j < RW; ++j) { const int hh = h + dhj[j] - pt; on[j] = rowok[j] && hh >= 0 && hh < S; }
      const float* pa = ds + (size_t)(h * TW + ps) * 36 + (lane & 31);
      const float* pb = img + (size_t)h * rowlen + ps * Cp + loff;
#define RSR_W4(c) if (NCG > c) acc[j][NCG > c ? c : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[u], bv[j][u], acc[j][NCG > c ? c : 0], 4, c, 0);
      int t = 0;
      for (; t + 4 <= nwl; t += 4) {
        float av[4], bv[RW][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = pa[u * sa];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
          if (!on[j]) continue;
#pragma unroll
          for (int u = 0; u < 4; ++u) bv[j][u] = pb[sh[j] + u * sb];
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
          if (!on[j]) continue;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            RSR_W4(0) RSR_W4(1) RSR_W4(2) RSR_W4(3) RSR_W4(4) RSR_W4(5) RSR_W4(6) RSR_W4(7)
          }
        }
        pa += 4 * sa; pb += 4 * sb;
      }
      for (; t < nwl; ++t) {
        float av[1], bv[RW][1];
        constexpr int u = 0;
        av[0] = pa[0];
#pragma unroll
        for (int j = 0; j < RW; ++j) bv[j][0] = on[j] ? pb[sh[j]] : 0.f;
#pragma unroll
        for (int j = 0; j < RW; ++j) {
          if (!on[j]) continue;
          RSR_W4(0) RSR_W4(1) RSR_W4(2) RSR_W4(3) RSR_W4(4) RSR_W4(5) RSR_W4(6) RSR_W4(7)
        }
        pa += sa; pb += sb;
      }
#undef RSR_W4
    }
  }
  if (bpart && blockIdx.x == 0) {                          // (uniform) fixed-order sum over the 64 * NWV / 8 threads of a channel group
    __syncthreads();
    float4* sb4 = reinterpret_cast<float4*>(smem);
    sb4[tid] = bsum;
    __syncthreads();
    if (tid < 8) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < 8 * NWV; ++j) { const float4 v = sb4[tid + 8 * j]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      *reinterpret_cast<float4*>(bpart + ((size_t)grp * gridDim.z + strip) * 32 + 4 * tid) = t;
    }
  }
  // partial slot of (frame group, strip, position part): part[((grp*nstrips + strip)*PS + ps)][dh][k'][32]; rows of this slot that
  // belong to other row sets are written by their waves (same ps)
  if (!active) return;
  if (kq >= fw * C) return;
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    const int dd = rs * RW + j, dh = dh0 + dd * dstep;
    if (dd >= DH || dh >= S) continue;
    float* po = part + (((size_t)(grp * gridDim.z + strip) * PS + ps) * S + dh) * (size_t)KP * 32 + (size_t)loff * 32;
#pragma unroll
    for (int c = 0; c < NCG; ++c) *reinterpret_cast<float4*>(po + 4 * c) = make_float4(acc[j][c][0], acc[j][c][1], acc[j][c][2], acc[j][c][3]);
  }
}

// dW[(dh*fw + dw)*C + c][co] = sum over partial tiles p of part[p][dh][dw*C' + c][co]   (fixed order)
__global__ void k_conv_wgrad_red(const float* __restrict__ part, int nparts, int S, int fw, int C, int N, float* __restrict__ dW, int ldw,
                                 const float* __restrict__ bpart, float* __restrict__ db, int nbparts) {
  const int Cp = conv_cpad(C), KP = conv_kp(fw, C);
  const int total = S * fw * C * N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (db && i >= total && i < total + N) {                 // the bias gradient from the row-group-0 workgroups' column sums
    float s = 0.f;
    for (int p = 0; p < nbparts; ++p) s += bpart[(size_t)p * 32 + (i - total)];
    db[i - total] = s;
    return;
  }
  if (i >= total) return;
  const int co = i % N, k = i / N;
  const int c = k % C, dd = k / C, dw = dd % fw, dh = dd / fw;
  const size_t off = ((size_t)dh * KP + dw * Cp + c) * 32 + co, stride = (size_t)S * KP * 32;
  float s = 0.f;
#pragma unroll 8
  for (int p = 0; p < nparts; ++p) s += part[p * stride + off];
  dW[(size_t)k * ldw + co] = s;
}

size_t conv_prep_floats(int S, int fw, int C) { return (size_t)S * 32 * conv_ldf(fw, C); }

// ---- the same convolution on v_mfma_f32_4x4x1_16B_f32, for output widths that are no multiple of 16 (12, 20, 24 channels) ----
// k_conv_fwd pads the output channels to 16 or 32 MFMA columns: 20 channels use 62 % of the columns, 12 and 24 use 75 %.  The
// 4x4x1 instruction is sixteen 4x4 outer products at the same peak rate (512 flop per 8 cycles); measured lane map (tools/ubench/
// mfma4.hip): D[lane][reg i] = A[4*srcblk + i] * B[lane], srcblk = the lane's own block, or with cbsz = 4 the block `abid` for
// every lane.  So with the FILTER in the A slot (lane l holds Ft[channel l][k]: one register serves all channel groups, the
// immediate abid picks channels 4c..4c+3) and the PATCH in the B slot (lane = one of 64 consecutive positions), a lane ends up
// with four consecutive output channels of its own position: no padded columns for any N % 4 == 0, and a 16-byte store per lane.
// Cost: four times the MFMA instructions per flop and ~2.4x the LDS reads per flop of the 16x16x4 form -- both under their limits.
// Workgroup = 8 waves: waves w and w + 4 share the position groups {w & 3, (w & 3) + 4, ..} and split every filter row's k'
// range in halves (balanced for any group count); the upper half hands its accumulators over through LDS at the end.
template <int G, int NCG, int KS = 2>          // KS: waves per position-group set (k' split); NSET = 8 / KS sets of G groups
__global__ __launch_bounds__(512) void k_conv_fwd4(const float* __restrict__ in, int ldc_in, int C, const float* __restrict__ Ft,
                                                   const float* __restrict__ bias, int relu, float* __restrict__ out, int ldc_out, int N,
                                                   int S, int W, int fw, int TW, const float* __restrict__ mask, int wbase, int wend, int FB, int R) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Cp = conv_cpad(C), ldf = conv_ldf(fw, C);
  const int nk4 = fw * Cp / 4;                            // float4 steps of one filter row (no rounding to whole 16-float k-blocks here)
  const int pt = (S - 1) / 2, pl = (fw - 1) / 2;
  const int rowlen = (TW + fw - 1) * Cp + 16;
  // FB frames per workgroup (1 but for narrow remainder strips: a one-column strip of ONE frame is 11 positions): image rows
  // [fb][h], ONE zero row ZR = FB*S behind them; position m = (fb*S + h)*TW + wl
  const int ZR = FB * S;
  float* img = smem;                                      // [FB*S + 1][rowlen], row ZR = zeros
  float* fts = smem + (size_t)(ZR + 1) * rowlen;          // [32][ldf] filter slice of one dh
  const int r0 = blockIdx.y * FB, w0 = wbase + blockIdx.x * TW;      // this launch covers the columns [wbase, wend)
  const int tw = min(TW, wend - w0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // KS = 2: sets {w&3} of groups {w&3, (w&3)+4, ..}, k' halves.  KS = 4 (512-position workgroups): TWO interleaved sets (even / odd
  // groups: the rows a filter row does not touch are a prefix or a suffix of the strip, so both sets lose the same number of
  // groups +-1 and the skipped groups shorten the row for every wave) and k' quarters: 0.86 of the KS = 2 time at S = 11.
  constexpr int NSET = 8 / KS;
  const int wg = wv & (NSET - 1), half = wv / NSET;       // set, k' part
  {
    // six loads in flight per thread, UNCONDITIONAL from clamped / stand-in addresses, the zeros selected afterwards (a load under
    // a branch in a loop with a run-time bound is one global round trip per iteration: up to 13 of them per workgroup)
    const int cp4 = Cp / 4, row4 = rowlen / 4;
    const int total4 = (ZR + 1) * row4;
    constexpr int SU = 6;
    for (int i0 = tid; i0 < total4; i0 += 512 * SU) {
      float4 v[SU];
      bool ok[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = min(i0 + u * 512, total4 - 1);
        const int hr = i / row4, e = i - hr * row4;          // hr = fb*S + h: frames are consecutive in `in`, so is (r0*S + hr)
        const int x = e / cp4, c = (e - x * cp4) * 4;
        const int wcol = w0 - pl + x;
        ok[u] = hr < ZR && r0 * S + hr < R * S && x < TW + fw - 1 && wcol >= 0 && wcol < W && c < C;
        const float* src = ok[u] ? in + ((size_t)(r0 * S + hr) * W + wcol) * ldc_in + c : in;
        v[u] = *reinterpret_cast<const float4*>(src);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + u * 512;
        const float4 z = make_float4(ok[u] ? v[u].x : 0.f, ok[u] ? v[u].y : 0.f, ok[u] ? v[u].z : 0.f, ok[u] ? v[u].w : 0.f);
        if (i < total4) *reinterpret_cast<float4*>(img + (size_t)i * 4) = z;
      }
    }
  }
  const int M = ZR * TW;
  int ph[G], pofs[G], pfr[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int m = (i * NSET + wg) * 64 + lane;
    const int hr = m / TW, wl = m - hr * TW;
    const int fb = hr / S, h = hr - fb * S;
    const bool ok = m < M && wl < tw && r0 + fb < R;
    ph[i] = ok ? h : -1000;
    pfr[i] = fb * S;
    pofs[i] = wl * Cp;
  }
  f32x4 acc[G][NCG];
#pragma unroll
  for (int i = 0; i < G; ++i)
#pragma unroll
    for (int c = 0; c < NCG; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NROW = NCG > 4 ? 32 : 16;                 // filter rows staged per dh
  const int n4 = NROW * ldf / 4;
  const int fi0 = min(tid, n4 - 1), fi1 = min(tid + 512, n4 - 1), fi2 = min(tid + 1024, n4 - 1), fi3 = min(tid + 1536, n4 - 1),
            fi4 = min(tid + 2048, n4 - 1);
  float4 fr0, fr1, fr2, fr3, fr4;
  {
    const float4* src = reinterpret_cast<const float4*>(Ft);
    fr0 = src[fi0]; fr1 = src[fi1]; fr2 = src[fi2]; fr3 = src[fi3]; fr4 = src[fi4];
  }
  // the k loop walks the REAL channel quads of a filter row: quad qi = dw*C4 + j sits at k' = dw*C' + 4j; the pad quads of the LDS
  // image (C' = 20 / 28 / 36 for 16 / 24 / 32 channels: every 5th / 7th / 9th quad) only multiply zeros of the filter
  const int C4 = max(1, C / 4), nq = fw * C4;
  const int s_lo = nq * half / KS, s_hi = nq * (half + 1) / KS;
  const int j_lo = s_lo % C4, off_lo = (s_lo / C4) * Cp + 4 * j_lo;
  (void)nk4;
  const float* fb = fts + (size_t)(lane & (NROW - 1)) * ldf;
  for (int dh = 0; dh < S; ++dh) {
    __syncthreads();
    {
      float4* dst = reinterpret_cast<float4*>(fts);
      if (tid < n4) dst[tid] = fr0;
      if (tid + 512 < n4) dst[tid + 512] = fr1;
      if (tid + 1024 < n4) dst[tid + 1024] = fr2;
      if (tid + 1536 < n4) dst[tid + 1536] = fr3;
      if (tid + 2048 < n4) dst[tid + 2048] = fr4;
    }
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(Ft + (size_t)min(dh + 1, S - 1) * 32 * ldf);
      fr0 = src[fi0]; fr1 = src[fi1]; fr2 = src[fi2]; fr3 = src[fi3]; fr4 = src[fi4];
    }
    const int vlo = max(0, pt - dh) * TW, vhi = min(S, S + pt - dh) * TW;
    const float* arow[G];
    bool live[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int hh = ph[i] + dh - pt;
      arow[i] = img + (size_t)((hh >= 0 && hh < S) ? pfr[i] + hh : ZR) * rowlen + pofs[i];
      const int t0 = (i * NSET + wg) * 64;
      live[i] = FB > 1 ? t0 < M : (t0 + 63 >= vlo && t0 < vhi && t0 < M);             // (wave-uniform) a group outside the rows this filter row touches: only zeros
    }
    // One uniform branch per live group and step around its 4 * NCG products (a branch per group AND component cost a fifth of
    // the issue slots: SQ_INSTS_SALU 1.2e9 next to 2.9e9 MFMAs; one specialised copy of the loop per live mask doubled the
    // accumulator registers).  Unrolled by two: the operands of step s + 1 are requested before the products of step s.
    auto prod = [&](const float4& fv, const float4 (&av)[G]) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (!live[i]) continue;
#define RSR_C4(comp)                                                                                                          \
        acc[i][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][0], 4, 0, 0);                              \
        if (NCG > 1) acc[i][NCG > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][NCG > 1 ? 1 : 0], 4, 1, 0); \
        if (NCG > 2) acc[i][NCG > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][NCG > 2 ? 2 : 0], 4, 2, 0); \
        if (NCG > 3) acc[i][NCG > 3 ? 3 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][NCG > 3 ? 3 : 0], 4, 3, 0); \
        if (NCG > 4) acc[i][NCG > 4 ? 4 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][NCG > 4 ? 4 : 0], 4, 4, 0); \
        if (NCG > 5) acc[i][NCG > 5 ? 5 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][NCG > 5 ? 5 : 0], 4, 5, 0); \
        if (NCG > 6) acc[i][NCG > 6 ? 6 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][NCG > 6 ? 6 : 0], 4, 6, 0); \
        if (NCG > 7) acc[i][NCG > 7 ? 7 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(fv.comp, av[i].comp, acc[i][NCG > 7 ? 7 : 0], 4, 7, 0);
        RSR_C4(x) RSR_C4(y) RSR_C4(z) RSR_C4(w)
#undef RSR_C4
      }
    };
    bool any = false;
#pragma unroll
    for (int i = 0; i < G; ++i) any |= live[i];
    if (any && s_lo < s_hi) {
      int off = off_lo, jq = j_lo, left = s_hi - s_lo;       // the next quad to request (uniform); past the end the last one is re-read
      auto fetch = [&](float4& fv, float4 (&av)[G]) {
        fv = *reinterpret_cast<const float4*>(fb + off);
#pragma unroll
        for (int i = 0; i < G; ++i) av[i] = *reinterpret_cast<const float4*>(arow[i] + off);
        if (--left > 0) {
          off += 4;
          if (++jq == C4) { jq = 0; off += Cp - 4 * C4; }
        }
      };
      float4 f0, f1;
      float4 a0[G], a1[G];
      fetch(f0, a0);
      int st = s_lo;
      for (; st + 1 < s_hi; st += 2) {
        fetch(f1, a1);
        prod(f0, a0);
        fetch(f0, a0);
        prod(f1, a1);
      }
      if (st < s_hi) prod(f0, a0);
    }
  }
  // the k' parts hand their sums down a tree (part p + h -> part p, h = KS/2, .., 1) through LDS, over the (now idle) image:
  // red[(part - h) * NSET + set][i][c][lane] float4
  float4* red = reinterpret_cast<float4*>(smem);
#pragma unroll
  for (int hh = KS / 2; hh >= 1; hh >>= 1) {
    __syncthreads();
    if (half >= hh && half < 2 * hh) {
#pragma unroll
      for (int i = 0; i < G; ++i)
#pragma unroll
        for (int c = 0; c < NCG; ++c)
          red[((size_t)(((half - hh) * NSET + wg) * G + i) * NCG + c) * 64 + lane] = make_float4(acc[i][c][0], acc[i][c][1], acc[i][c][2], acc[i][c][3]);
    }
    __syncthreads();
    if (half < hh) {
#pragma unroll
      for (int i = 0; i < G; ++i)
#pragma unroll
        for (int c = 0; c < NCG; ++c) {
          const float4 u = red[((size_t)((half * NSET + wg) * G + i) * NCG + c) * 64 + lane];
          acc[i][c][0] += u.x; acc[i][c][1] += u.y; acc[i][c][2] += u.z; acc[i][c][3] += u.w;
        }
    }
  }
  if (half) return;
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int m = (i * NSET + wg) * 64 + lane;
    const int h = m / TW, wl = m - h * TW;                  // h = fb*S + row
    if (m >= M || wl >= tw || r0 * S + h >= R * S) continue;
    const size_t o0 = ((size_t)(r0 * S + h) * W + w0 + wl) * ldc_out;
#pragma unroll
    for (int c = 0; c < NCG; ++c) {
      if (4 * c >= N) continue;
      float4 v = make_float4(acc[i][c][0], acc[i][c][1], acc[i][c][2], acc[i][c][3]);
      if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (mask) {
        const float4 k = *reinterpret_cast<const float4*>(mask + o0 + 4 * c);
        v = make_float4(k.x > 0.f ? v.x : 0.f, k.y > 0.f ? v.y : 0.f, k.z > 0.f ? v.z : 0.f, k.w > 0.f ? v.w : 0.f);
      }
      *reinterpret_cast<float4*>(out + o0 + 4 * c) = v;
    }
  }
}

void launch_conv_prep(const float* F, int ldf_src, int S, int fw, int Cin, int Cout, bool flip, float* Ft, hipStream_t s) {
  const size_t total = conv_prep_floats(S, fw, flip ? Cout : Cin);
  hipLaunchKernelGGL(k_conv_prep, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, F, ldf_src, S, fw, Cin, Cout, flip ? 1 : 0, Ft);
}

// Strip width TW and row tiles per wave RT (4 or 6: 512 / 768 positions per workgroup) of the forward kernel: even strips
// (W / nstrips, not a fixed 64: a 257-wide frame is 4 x 65, not 5 x 64), the fewest strips whose workgroup fits the LDS.
static bool conv_fwd_plan(int C, int S, int W, int fw, int& TW, int& RT, size_t& lds) {
  double best = 0.0;
  for (int rt : {4, 6}) {
    const int twmax = 8 * rt * 16 / S;
    if (twmax < 1) continue;
    for (int ns = (W + twmax - 1) / twmax; ns <= W; ++ns) {
      const int tw = (W + ns - 1) / ns;
      const size_t l = ((size_t)(S + 1) * ((tw + fw - 1) * conv_cpad(C) + 16) + (size_t)32 * conv_ldf(fw, C)) * sizeof(float);
      if (l > 160 * 1024) continue;
      const double eff = (double)W * S / ((double)ns * 8 * rt * 16);       // useful share of the MFMA row slots
      if (eff > best + 1e-9) { best = eff; TW = tw; RT = rt; lds = l; }
      break;
    }
  }
  return best > 0.0;
}
// true if the implicit kernel covers this shape (else the caller uses the patch-matrix path)
bool conv_fwd_supported(int C, int N, int S, int W, int fw) {
  if ((C % 4 && C != 1) || N > 32 || !(S & 1) || !(fw & 1)) return false;      // C == 1: the caller pads the input to [positions][4]
  if (32 * conv_ldf(fw, C) / 4 > 5 * 512) return false;                          // k_conv_fwd keeps a filter slice in 5 float4 per thread
  int TW, RT; size_t lds;
  return conv_fwd_plan(C, S, W, fw, TW, RT, lds);
}

template <int G, int NCG, int KS>
static void launch_conv_fwd4_t(dim3 grid, size_t lds, hipStream_t s, const float* in, int ldc_in, int C, const float* Ft, const float* bias, int rl,
                               float* out, int ldc_out, int N, int S, int W, int fw, int TW, const float* mask, int wbase, int wend, int FB, int R) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fwd4<G, NCG, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((k_conv_fwd4<G, NCG, KS>), grid, dim3(512), lds, s, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, S, W, fw, TW, mask, wbase, wend, FB, R);
}
static bool launch_conv_fwd4(int G, int ncg, dim3 grid, size_t lds, hipStream_t s, const float* in, int ldc_in, int C, const float* Ft,
                             const float* bias, int rl, float* out, int ldc_out, int N, int S, int W, int fw, int TW, const float* mask, int wbase, int wend, int FB, int R) {
#define RSR_L4(g, n, ks) if (G == g && ncg == n) { launch_conv_fwd4_t<g, n, ks>(grid, lds, s, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, S, W, fw, TW, mask, wbase, wend, FB, R); return true; }
  RSR_L4(2, 1, 2) RSR_L4(2, 2, 2) RSR_L4(2, 3, 2) RSR_L4(2, 4, 2) RSR_L4(2, 5, 2) RSR_L4(2, 6, 2) RSR_L4(2, 7, 2) RSR_L4(2, 8, 2)
  RSR_L4(3, 1, 2) RSR_L4(3, 2, 2) RSR_L4(3, 3, 2) RSR_L4(3, 4, 2) RSR_L4(3, 5, 2) RSR_L4(3, 6, 2) RSR_L4(3, 7, 2) RSR_L4(3, 8, 2)
  RSR_L4(4, 1, 4) RSR_L4(4, 2, 4) RSR_L4(4, 3, 4) RSR_L4(4, 4, 4) RSR_L4(4, 5, 4) RSR_L4(4, 6, 4) RSR_L4(4, 7, 4) RSR_L4(4, 8, 4)
#undef RSR_L4
  return false;
}

// one launch over the columns [wbase, wend) in strips of TW
static void conv_fwd_range(int wbase, int wend, int TW, int RT, size_t lds, const float* in, int ldc_in, int C, const float* Ft, const float* bias,
                           int rl, float* out, int ldc_out, int N, int R, int S, int W, int fw, hipStream_t s, const float* mask, int FB = 1) {
  const bool small = RT == 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fwd<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fwd<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fwd<6, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fwd<6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  dim3 grid((wend - wbase + TW - 1) / TW, (R + FB - 1) / FB);
  // RSRGAN_CONV4: 0 = never the 4x4x1 form, 1 (default) = every output width that is a multiple of 4 except 16 (12 / 20 / 24 waste
  // 16-wide MFMA columns; at 32 the 4x4x1 kernel is ahead since it walks only the real channel quads, 16.4 vs 18.0 ms per call;
  // at 16 -- layers whose input has no channel pads -- the 16x16x4 kernel is, 8.7 vs 9.3), 2 = every multiple of 4
  static int conv4 = -1;
  if (conv4 < 0) { const char* e = getenv("RSRGAN_CONV4"); conv4 = e ? atoi(e) : 1; }
  if (conv4 && N % 4 == 0 && ldc_out % 4 == 0 && (conv4 > 1 || N != 16) && (!bias || ((size_t)bias & 15) == 0)) {
    // RSRGAN_CONV4_KS=4: two interleaved group sets x k' quarters for 512-position workgroups (0.86 of the MFMAs of the default
    // four sets x k' halves at S = 11, measured slower: 566 vs 558 ms per step of the R-CED variant)
    static int ks4 = -1;
    if (ks4 < 0) { const char* e = getenv("RSRGAN_CONV4_KS"); ks4 = e ? atoi(e) : 2; }
    const int G = small ? (ks4 == 4 ? 4 : 2) : 3, ncg = N / 4;
    const size_t lds4 = std::max(lds, (size_t)8 * ncg * 1024 * (G == 3 ? 2 : G == 4 ? 2 : 1));      // the tree's widest round
    if (lds4 <= 160 * 1024 && launch_conv_fwd4(G, ncg, grid, lds4, s, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, S, W, fw, TW, mask, wbase, wend, FB, R)) return;
  }
  if (small && N <= 16) hipLaunchKernelGGL((k_conv_fwd<4, 1>), grid, dim3(512), lds, s, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, S, W, fw, TW, mask, wbase, wend, FB, R);
  else if (small) hipLaunchKernelGGL((k_conv_fwd<4, 2>), grid, dim3(512), lds, s, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, S, W, fw, TW, mask, wbase, wend, FB, R);
  else if (N <= 16) hipLaunchKernelGGL((k_conv_fwd<6, 1>), grid, dim3(512), lds, s, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, S, W, fw, TW, mask, wbase, wend, FB, R);
  else hipLaunchKernelGGL((k_conv_fwd<6, 2>), grid, dim3(512), lds, s, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, S, W, fw, TW, mask, wbase, wend, FB, R);
}

void launch_conv_fwd(const float* in, int ldc_in, int C, const float* Ft, const float* bias, bool relu, float* out, int ldc_out, int N,
                     int R, int S, int W, int fw, hipStream_t s, const float* mask) {
  int TW = W, RT = 4; size_t lds = 0;
  if (!conv_fwd_plan(C, S, W, fw, TW, RT, lds)) return;
  const int rl = relu ? 1 : 0;
  // Row-aligned strips (RSRGAN_CONV_ROWS=0 turns them off): with 64-column strips a 16- or 64-position group never straddles two
  // image rows, so the rows a filter row does not touch are skipped exactly and every lane of a full strip is a real position
  // (a 257-wide frame cut into 6 x 43 columns used 69 % / 80 % of the position slots of the 4x4x1 / 16x16x4 kernel).  The W % 64
  // columns that are left take a second launch with their own plan.
  static int rows = -1;
  if (rows < 0) { const char* e = getenv("RSRGAN_CONV_ROWS"); rows = e ? atoi(e) : 1; }
  const size_t lds64 = ((size_t)(S + 1) * ((64 + fw - 1) * conv_cpad(C) + 16) + (size_t)32 * conv_ldf(fw, C)) * sizeof(float);
  if (rows && W > 64 && S * 64 <= 8 * 6 * 16 && lds64 <= 160 * 1024) {
    const int wmain = W / 64 * 64;
    conv_fwd_range(0, wmain, 64, 6, lds64, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, R, S, W, fw, s, mask);
    if (wmain < W) {
      int TWr = W - wmain, RTr = 4; size_t ldsr = 0;
      if (!conv_fwd_plan(C, S, W - wmain, fw, TWr, RTr, ldsr)) return;
      // a narrow remainder (one column of a 257-wide frame = 11 positions) takes several frames per workgroup: as many as fill the
      // position slots and fit the LDS (one workgroup per frame cost 0.9 ms per layer, a quarter of a 64-column strip launch)
      int FB = 1;
      if ((W - wmain + TWr - 1) / TWr == 1) {
        const size_t rowb = ((size_t)(TWr + fw - 1) * conv_cpad(C) + 16) * sizeof(float), filt = (size_t)32 * conv_ldf(fw, C) * sizeof(float);
        const int cap = 8 * RTr * 16 / (S * TWr);
        const int fit = (int)((128 * 1024 - filt - rowb) / (rowb * S));
        FB = std::max(1, std::min(std::min(cap, fit), 16));
        ldsr = ((size_t)FB * S + 1) * rowb + filt;
      }
      conv_fwd_range(wmain, W, TWr, RTr, ldsr, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, R, S, W, fw, s, mask, FB);
    }
    return;
  }
  conv_fwd_range(0, W, TW, RT, lds, in, ldc_in, C, Ft, bias, rl, out, ldc_out, N, R, S, W, fw, s, mask);
}

// Grid of the weight-gradient kernel: ceil(S / DH) filter-row groups x frame groups x strips, at most 256 workgroups (one round
// on the 256 CUs); DH (filter rows per workgroup, 1..3) is chosen to minimise the frames a workgroup has to stage.
static int wgrad_tw(int S, int W) {          // narrower strips than the forward kernel (the d strip is in LDS too), evenly split
  if (S * W <= 8 * 4 * 16) return W;
  const int ns = (W + 31) / 32;
  return (W + ns - 1) / ns;
}
static int g_wgrad_dhmax = -1;        // RSRGAN_WGRAD_DH: most filter rows per workgroup for multi-strip frames (3 = round 2's form)
static void wgrad_plan(int R, int S, int nstrips, int KP, int& DH, int& fpg, int& groups) {
  if (g_wgrad_dhmax < 0) { const char* e = getenv("RSRGAN_WGRAD_DH"); g_wgrad_dhmax = e ? atoi(e) : 6; }
  int best = 1 << 30;
  DH = 1; fpg = R; groups = 1;
  // multi-strip frames with one k'-tile per wave (K' <= 256): 4 or 6 rows per workgroup in the rows-inside form (its accumulators
  // fit the 128 registers of a 16-wave workgroup); the measure is the LDS operand reads + staged frames per MFMA
  if (nstrips > 1 && KP > 16 * 16 && g_wgrad_dhmax >= 3) {        // two k'-tile rounds: 3 rows (accumulators: 3 x 2 x NT tiles)
    const int ng = (S + 2) / 3, gmax = std::max(1, 256 / (ng * nstrips));
    DH = 3; fpg = std::max(1, (R + gmax - 1) / gmax); groups = (R + fpg - 1) / fpg;
    return;
  }
  if (nstrips > 1 && KP <= 16 * 16 && g_wgrad_dhmax > 3) {
    for (int dhc : {6, 4}) {
      if (dhc > g_wgrad_dhmax) continue;
      const int ng = (S + dhc - 1) / dhc;
      if (ng * dhc - S >= dhc / 2 + 1) continue;                  // (too many empty rows in the last group)
      const int gmax = std::max(1, 256 / (ng * nstrips));
      const int f = std::max(1, (R + gmax - 1) / gmax);
      DH = dhc; fpg = f; groups = (R + f - 1) / f;
      return;
    }
  }
  // measured: with one strip per frame (W = 40) one filter row per workgroup is faster (4.20 vs 4.39 ms/step) although it stages
  // three times as many frames; with 9 strips (W = 257) three rows per workgroup win (26.2 vs 28.0 ms/step)
  for (int dhc = 1; dhc <= (nstrips > 1 ? 3 : 1); ++dhc) {
    const int ng = (S + dhc - 1) / dhc;
    const int gmax = std::max(1, 256 / (ng * nstrips));
    const int f = std::max(1, (R + gmax - 1) / gmax);
    if (f < best) { best = f; DH = dhc; fpg = f; groups = (R + f - 1) / f; }
  }
}
// k_conv_wgrad4: position parts per (k' group, row set) of a 16-wave workgroup
// (workgroup of 8, 10 or 12 waves: the largest of those that the roles k' group x row set x position part fill completely, else 8)
static int wgrad4_waves(int KP, int DH, int& PS) {
  const int roles = ((KP + 63) / 64) * ((DH + 2) / 3);
  for (int w : {12, 10, 8})
    if (roles <= w && w % roles == 0) { PS = w / roles; return w; }
  PS = std::max(1, 8 / roles);
  return roles <= 8 ? 8 : (roles <= 10 ? 10 : 12);
}
// the plan of the 4x4x1 form: its 64-lane groups cover fw*C filter elements (no channel pads), and up to four of them take the
// six-rows-per-workgroup plan whatever the padded K' is
static void wgrad4_plan(int C, int R, int S, int nstrips, int fw, int& DH, int& fpg, int& groups, int& nkg, int& PS, int& nwv) {
  nkg = (fw * C + 63) / 64;
  const int KP = conv_kp(fw, C);
  wgrad_plan(R, S, nstrips, nkg <= 4 ? std::min(KP, 256) : KP, DH, fpg, groups);
  nwv = wgrad4_waves(64 * nkg, DH, PS);
}
static size_t conv_wgrad_ws_floats_at(int C, int R, int S, int W, int fw) {
  const int TW = wgrad_tw(S, W), nstrips = (W + TW - 1) / TW;
  int DH, fpg, groups, nkg, PS, nwv;
  wgrad_plan(R, S, nstrips, conv_kp(fw, C), DH, fpg, groups);
  // filter partials (one slot per position part in the 4x4x1 form) + one 32-float bias partial per (group, strip)
  const size_t a = (size_t)groups * nstrips * ((size_t)S * conv_kp(fw, C) + 1) * 32;
  wgrad4_plan(C, R, S, nstrips, fw, DH, fpg, groups, nkg, PS, nwv);
  const size_t b = (size_t)groups * nstrips * ((size_t)PS * S * conv_kp(fw, C) + 1) * 32;
  return std::max(a, b);
}
// The work space for up to R frames.  The planners' group count is NOT monotonic in the frame count (groups = ceil(R / ceil(R / gmax)):
// 57 frames at gmax = 28 make 19 groups, 28 frames make 28), and the C ABI accepts any T <= max_frames: size it for the worst R' <= R.
size_t conv_wgrad_ws_floats(int C, int R, int S, int W, int fw) {
  size_t need = 0;
  for (int r = 1; r <= R; ++r) need = std::max(need, conv_wgrad_ws_floats_at(C, r, S, W, fw));
  return need;
}
bool conv_wgrad_supported(int C, int N, int S, int W, int fw) {
  if (!conv_fwd_supported(C, N, S, W, fw)) return false;
  const int TW = wgrad_tw(S, W);
  const int MP = (S * TW + 15) / 16 * 16;
  const size_t lds = ((size_t)(S + 1) * ((TW + fw - 1) * conv_cpad(C) + 16) + (size_t)MP * 38) * sizeof(float);
  return lds <= 160 * 1024 && MP * 8 <= 4 * 1024 && conv_kp(fw, C) <= 16 * 2 * 16 && ((TW + fw - 1) * conv_cpad(C) + 16) / 4 <= 512 && (TW + fw) * conv_cpad(C) < (1 << 20);
}
template <int KT, int NT>
static void wgrad_launch_dh(int DH, dim3 grid, size_t lds, hipStream_t s, const float* in, int ldc_in, int C, const float* d, int ldc_d, int N,
                            float* ws, int S, int W, int fw, int TW, int R, int fpg, float* bpart) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad<KT, NT, 16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad<KT, NT, 16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad<KT, NT, 16, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (KT == 1) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad<KT, NT, 16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad<KT, NT, 16, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    attr = true;
  }
  if (DH == 1) hipLaunchKernelGGL((k_conv_wgrad<KT, NT, 16, 1>), grid, dim3(1024), lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  else if (DH == 2) hipLaunchKernelGGL((k_conv_wgrad<KT, NT, 16, 2>), grid, dim3(1024), lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  else if (DH == 3) hipLaunchKernelGGL((k_conv_wgrad<KT, NT, 16, 3>), grid, dim3(1024), lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  else if constexpr (KT == 1) {
    if (DH == 4) hipLaunchKernelGGL((k_conv_wgrad<KT, NT, 16, 4>), grid, dim3(1024), lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
    else hipLaunchKernelGGL((k_conv_wgrad<KT, NT, 16, 6>), grid, dim3(1024), lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  }
}
template <int NCG, int NWV>
static void launch_conv_wgrad4_t(dim3 grid, size_t lds, hipStream_t s, const float* in, int ldc_in, int C, const float* d, int ldc_d, int N, float* ws,
                                 int S, int W, int fw, int TW, int R, int fpg, float* bpart, int DH, int nkg, int PS) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad4<NCG, 3, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((k_conv_wgrad4<NCG, 3, NWV>), grid, dim3(64 * NWV), lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart, DH, nkg, PS);
}
static void launch_conv_wgrad4(int ncg, int nwv, dim3 grid, size_t lds, hipStream_t s, const float* in, int ldc_in, int C, const float* d, int ldc_d, int N,
                               float* ws, int S, int W, int fw, int TW, int R, int fpg, float* bpart, int DH, int nkg, int PS) {
#define RSR_W(n, w) if (ncg == n && nwv == w) { launch_conv_wgrad4_t<n, w>(grid, lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart, DH, nkg, PS); return; }
#define RSR_WN(n) RSR_W(n, 8) RSR_W(n, 10) RSR_W(n, 12)
  RSR_WN(1) RSR_WN(2) RSR_WN(3) RSR_WN(4) RSR_WN(5) RSR_WN(6) RSR_WN(7) RSR_WN(8)
#undef RSR_WN
#undef RSR_W
}
void launch_conv_wgrad(const float* in, int ldc_in, int C, const float* d, int ldc_d, int N, float* dW, int ldw, float* ws, int R, int S,
                       int W, int fw, hipStream_t s, float* db) {
  const int TW = wgrad_tw(S, W), nstrips = (W + TW - 1) / TW;
  int DH, fpg, groups;
  const int MP = (S * TW + 15) / 16 * 16, KP = conv_kp(fw, C);
  const size_t lds = ((size_t)(S + 1) * ((TW + fw - 1) * conv_cpad(C) + 16) + (size_t)MP * 38) * sizeof(float);
  // every output width that is a multiple of 4 goes to the 4x4x1 form (measured: 502 ms per R-CED step against 509 with only the
  // widths that waste 16-wide MFMA columns; unlike the forward kernel it has no padded positions to pay for).  RSRGAN_WGRAD4: 0 =
  // never, 1 = widths that are no multiple of 16, 2 = all (default); RSRGAN_CONV4=0 alone turns both directions off.
  static int w4 = -1;
  if (w4 < 0) { const char* e = getenv("RSRGAN_CONV4"); const char* e2 = getenv("RSRGAN_WGRAD4"); w4 = e2 ? atoi(e2) : ((e && !atoi(e)) ? 0 : 2); }
  int nkg = 1, PS = 1, nwv4 = 8;
  wgrad4_plan(C, R, S, nstrips, fw, DH, fpg, groups, nkg, PS, nwv4);
  const bool use4 = w4 && N % 4 == 0 && N <= 32 && (w4 > 1 || N % 16 != 0) && nkg * ((DH + 2) / 3) <= 12 &&
                    ((TW + fw - 1) * conv_cpad(C) + 16) / 4 <= 64 * nwv4;
  if (!use4) { PS = 1; wgrad_plan(R, S, nstrips, KP, DH, fpg, groups); }
  dim3 grid((S + DH - 1) / DH, groups, nstrips);
  const int nparts = groups * nstrips * PS;
  float* bpart = db ? ws + (size_t)nparts * S * KP * 32 : nullptr;
  if (use4) {
    launch_conv_wgrad4(N / 4, nwv4, grid, lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart, DH, nkg, PS);
    const int total4 = S * fw * C * N;
    hipLaunchKernelGGL(k_conv_wgrad_red, dim3((total4 + N + 255) / 256), dim3(256), 0, s, ws, nparts, S, fw, C, N, dW, ldw, bpart, db, groups * nstrips);
    return;
  }
  const bool k2 = KP > 16 * 16, n1 = N <= 16;          // 16 waves: one k'-tile per wave up to K' = 256, two beyond
  if (k2 && n1) wgrad_launch_dh<2, 1>(DH, grid, lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  else if (k2) wgrad_launch_dh<2, 2>(DH, grid, lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  else if (n1) wgrad_launch_dh<1, 1>(DH, grid, lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  else wgrad_launch_dh<1, 2>(DH, grid, lds, s, in, ldc_in, C, d, ldc_d, N, ws, S, W, fw, TW, R, fpg, bpart);
  const int total = S * fw * C * N;
  hipLaunchKernelGGL(k_conv_wgrad_red, dim3((total + N + 255) / 256), dim3(256), 0, s, ws, groups * nstrips, S, fw, C, N, dW, ldw, bpart, db, groups * nstrips);
}

}  // namespace rsr
