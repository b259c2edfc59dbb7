// bwd_ab_experiment.hip -- NOT part of the product: the fused backward kernel measured in round 2 and not kept.
//
// Idea: phases A (dh = dm.Wp^T, cell gradients -> dz) and B (dz.K^T, split over cell groups) of a backward diagonal in ONE launch,
// so that a diagonal costs fused + reduce instead of k_bwd_a2 + k_bwd_bp + k_bwd_b_red.  A workgroup owns 32 cells (128 columns of
// dz), 64 rows and 192 of the 560 output columns; the three workgroups sharing a cell group recompute the same dz (the price of not
// synchronising across workgroups), the one with og == 0 stores it.  Needs a cell-major tiled copy of K (k_swizzle_many mode
// gates < 0), dz in its own stash (others still read the activations) and a double-buffered carried dc.
//
// Result on MI355X (B=64, T=100, reference-true nets; parity-green on tests/test_gpu_fullsize.py, test_gpu_golden.py,
// test_gpu_placement.py with the path forced on): k_bwd_ab 21.1 us per diagonal vs 8.65 + 13.4 us for the two kernels it
// replaces, step 8.26 vs 8.21 ms.  Timing ablation (tools/ab_ablate.sh at the time): with EVERY global load and store removed the
// launch still takes 15.5 us -- the 3x recomputed phase-A product (1.9 us of MFMA per workgroup), two block barriers, the LDS
// round trips of dm / dz and the phase-B product (5.1 us) -- so saving one launch (~3 us) and one job lookup cannot pay for it.
// First version 21.9 us; Wp tiles through LDS by DMA (every wave had pulled its own copy: 4x the bytes) and float4 partial
// stores through wave-private LDS patches (scalar 64-byte pieces cost 2 us) brought 0.8 us.
//
// The code below is the kernel, its planner and the declarations as they were wired into kernels.hip / kernels.h; the host side
// (model.cpp) filled BwdABJob from LayerRun / FcStage per diagonal and launched k_bwd_ab + k_bwd_b_red.

#if 0   // ---- kernels.h
// Backward phases A and B in ONE launch (k_bwd_ab): a workgroup owns 32 cells of a layer (all four gates: 128 columns of dz), 64 rows
// and up to 192 output columns of [dx_t | dm_rec].  It computes dh = dm . Wp^T for its cells, the gate gradients dz (kept in LDS,
// laid out as the A operand of the second product), then its partial [64 x 192] of dz . K^T over its 128 columns.  The partials of
// the KG = ceil(H / 32) cell groups are summed by k_bwd_b_red as before.  The OG workgroups that share a cell group recompute the
// same dz; the one with og == 0 stores it (dzo) together with dc (double-buffered: the others still read the old value) and dm_t.
// mode 1 = a fully_connected stage riding the wave: no phase A, dz is read from `gates` ([N][H4]), K is row-major.
struct BwdABJob {
  const float* dout; const float* dmst; const float* Wp_sw; float* dmt;
  const float* gates;   // [N][4H] gate activations (mode 0) / the stage's input gradient [N][H4] (mode 1)
  float* dzo;           // [N][4H] out: dz
  const float* c_prev; const float* c_cur; const float* wf; const float* wi; const float* wo;
  const float* dc_in; float* dc_out; const int* len;
  const float* Kc;      // cell-major fragment-tiled copy of rows [n_begin, n_end) of K: tiles [row block][cell block x gate][256]
  const float* K;       // row-major [rows][H4] (mode 1)
  float* dx; float* ws;
  int ldm, P, t, N, H, H4, I, n_begin, n_end, lddx, ldw, KG, OG, nrg, mode;
  Place pl;
};
struct BwdABJobs { int n; int ablate; BwdABJob j[MAXJ]; JobMap map; };
constexpr int AB_CELLS = 32, AB_COLS = 192;
size_t bwd_ab_plan(BwdABJobs& jobs, float* ws_base);          // fills KG / OG / nrg / ldw / ws / pl; returns the floats of partials
void launch_bwd_ab(const BwdABJobs& jobs, hipStream_t s);
void bwd_b_red_plan(BwdBJobs& jobs);                          // places of k_bwd_b_red for jobs whose ws / ldw / KG are set
void launch_bwd_b_red(const BwdBJobs& jobs, hipStream_t s);
#endif

#if 0   // ---- kernels.hip
__device__ __forceinline__ void globalize(BwdABJob& J) {
#define RSR_G(f) J.f = as_global(J.f);
  RSR_G(dout) RSR_G(dmst) RSR_G(Wp_sw) RSR_G(dmt) RSR_G(gates) RSR_G(dzo) RSR_G(c_prev) RSR_G(c_cur) RSR_G(wf) RSR_G(wi) RSR_G(wo)
  RSR_G(dc_in) RSR_G(dc_out) RSR_G(len) RSR_G(Kc) RSR_G(K) RSR_G(dx) RSR_G(ws)
#undef RSR_G
}

// ---------------------------------------------------------------------------------------
// backward phases A + B in one launch (kernels.h BwdABJob).  512 threads = 8 waves.
//   phase A: wave (rt = w & 3, ct = w >> 2) owns the 16 x 16 tile (rows rt*16.., cells ct*16..) of dh = dm . Wp^T with the whole
//            K = P in registers (as k_bwd_a2), finishes the cell gradients in the accumulator layout and writes the four dz values
//            of every (row, cell) into the LDS image dzs[row tile][k-block = cell tile x gate][row][cell]
//   phase B: wave (rh = w & 1, cgp = w >> 1) owns rows rh*32.. and output column tiles cgp*3 .. +3 of the workgroup's 192:
//            6 accumulators over the 8 k-blocks, A fragments from dzs, B from the cell-major tiled copy of K (registers, requested
//            before phase A starts so that they land under it)
// ---------------------------------------------------------------------------------------
constexpr int AB_NKP = 18;            // k-blocks of phase A (P <= 288)
constexpr int AB_DZS = 20;            // floats per (row, k-block) of dzs: 16 + 4 (float4 fragment reads spread over the banks)
__host__ __device__ inline int ab_sa4(int ldm) { return (((ldm + 15) >> 4) * 4) | 1; }
// first LDS region: the dm image [64][SA] in phase A, then 8 wave-private [32][52] patches for the partial stores
__host__ __device__ inline size_t ab_dm_floats(int ldm) { const size_t a = (size_t)64 * ab_sa4(ldm) * 4, b = (size_t)8 * 32 * 52; return a > b ? a : b; }
__host__ __device__ inline size_t ab_lds_floats(int ldm) {      // + dzs + the two Wp cell tiles (all k-blocks)
  return ab_dm_floats(ldm) + (size_t)4 * 8 * 16 * AB_DZS + (size_t)2 * ((ldm + 15) >> 4) * 256;
}

__global__ __launch_bounds__(512, 2) void k_bwd_ab(const BwdABJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  int lb;
  const BwdABJob J = RSR_PICK(BwdABJob, pl, lb);
  if (lb < 0) return;
  const int abl = jobs.ablate;
  const int kg = lb % J.KG, rem = lb / J.KG;
  const int og = rem % J.OG, rg = rem / J.OG;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = J.N, H = J.H, H4 = J.H4, ldm = J.ldm;
  const int r0 = rg * 64;
  const int ncols = J.n_end - J.n_begin;
  const int SA4 = ab_sa4(ldm), SA = SA4 * 4;
  float* dms = smem;                                     // [64][SA]   dm = mask.(dout + dmst)
  float* dzs = smem + ab_dm_floats(ldm);                 // [4][8][16][AB_DZS]
  float* wps = dzs + (size_t)4 * 8 * 16 * AB_DZS;        // [2 cell tiles][nkb][256] fragment tiles of Wp (mode 0)
  const int rh = w & 1, cgp = w >> 1;                    // phase B roles
  const int nnb = (ncols + 15) >> 4;
  float4 bw[3][8];

  if (J.mode == 0) {
    const int rt = w & 3, ct = w >> 2;                   // phase A roles
    const int nkb = (ldm + 15) >> 4;
    // (1) every phase-A load first: dm operand, Wp tiles, the operands of this lane's four (row, cell) elements
    const int srow = tid >> 3, s8 = tid & 7;
    const int grow = min(r0 + srow, N - 1);
    const float* pm = J.dmst + (size_t)grow * ldm;
    const float* pd = (J.dout ? J.dout : J.dmst) + (size_t)grow * ldm;
    constexpr int NI = (AB_NKP * 4 + 7) / 8;
    const int k4max = (ldm >> 2) - 1, nk4 = ldm >> 2, nk4p = nkb * 4;
    float4 va[NI], vd[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int g4 = (abl & 16) ? 0 : min(s8 + 8 * i, k4max);
      va[i] = *reinterpret_cast<const float4*>(pm + g4 * 4);
      vd[i] = *reinterpret_cast<const float4*>(pd + g4 * 4);
    }
    const int slen = J.len[grow];
    // Wp tiles of the workgroup's two cell blocks -> LDS by DMA, once (every wave reading its own copy from global was 4x the bytes)
    if (!(abl & 4)) {
      const int ntile = 2 * nkb;                          // 1 KB each; wave w moves tiles w, w + 8, ..
      const int ncb16 = (H + 15) >> 4;
      for (int tl = w; tl < ntile; tl += 8) {
        const int tct = tl / nkb, tkb = tl - tct * nkb;
        const float* src = J.Wp_sw + ((size_t)min(kg * 2 + tct, ncb16 - 1) * nkb + tkb) * 256 + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wps + (size_t)tl * 256), 16, 0, 0);
      }
    }
    const int ecell = kg * AB_CELLS + ct * 16 + lr;
    const bool cok = ecell < H;
    const int ecl = min(ecell, H - 1);
    const float ewo = J.wo[ecl], ewi = J.wi[ecl], ewf = J.wf[ecl];
    float eg[4][4], ecp[4], ecn[4], edc[4];
    int elen[4];
    bool erok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = r0 + rt * 16 + 4 * q + e;
      erok[e] = (row < N) & cok;
      const int rowc = (abl & 1) ? 0 : min(row, N - 1);
      elen[e] = J.len[rowc];
      const float* g = J.gates + (size_t)rowc * H4 + ((abl & 1) ? 0 : ecl);
      eg[e][0] = g[0]; eg[e][1] = g[H]; eg[e][2] = g[2 * H]; eg[e][3] = g[3 * H];
      const size_t ci = (abl & 1) ? 0 : (size_t)rowc * H + ecl;
      ecp[e] = J.c_prev[ci]; ecn[e] = J.c_cur[ci]; edc[e] = J.dc_in[ci];
    }
    __builtin_amdgcn_sched_barrier(0);
    // (1') dm -> LDS (and dm_t for the projection's weight gradient from one workgroup per row group)
    {
      const bool live = (r0 + srow < N) & (J.t < slen);
      const float dscale = J.dout ? 1.f : 0.f;
      float* pt = (kg == 0 && og == 0 && r0 + srow < N) ? J.dmt + (size_t)grow * ldm : nullptr;
      float4* dst = reinterpret_cast<float4*>(dms) + (size_t)srow * SA4;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c4 = s8 + 8 * i;
        const bool in = live & (c4 < nk4);
        float4 v = make_float4(va[i].x + dscale * vd[i].x, va[i].y + dscale * vd[i].y, va[i].z + dscale * vd[i].z, va[i].w + dscale * vd[i].w);
        v = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
        if (c4 < nk4p) dst[c4] = v;
        if (pt && c4 < nk4) *reinterpret_cast<float4*>(pt + c4 * 4) = v;
      }
    }
    // (3) phase-B weights: requested now, consumed after phase A
    {
      const int nkbc = J.KG * 8;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const int nb = min(og * (AB_COLS / 16) + cgp * 3 + ci, nnb - 1);
        const float* wt = J.Kc + ((size_t)nb * nkbc + (size_t)kg * 8) * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) bw[ci][j] = (abl & 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(wt + (size_t)j * 256);
      }
    }
    __syncthreads();
    // (4) dh[16 x 16] = dm[rows rt*16.., :] . Wp[cells, :]^T
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    {
      const float* ab = dms + (size_t)(rt * 16 + lr) * SA + 4 * q;
      const float* wb = wps + (size_t)ct * nkb * 256 + lane * 4;
#pragma unroll
      for (int c = 0; c < AB_NKP; c += 2) {
        if (c < nkb) {
          const float4 a = *reinterpret_cast<const float4*>(ab + c * 16);
          const float4 b = *reinterpret_cast<const float4*>(wb + (size_t)c * 256);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc0, 0, 0, 0);
        }
        if (c + 1 < AB_NKP && c + 1 < nkb) {
          const float4 a = *reinterpret_cast<const float4*>(ab + (c + 1) * 16);
          const float4 b = *reinterpret_cast<const float4*>(wb + (size_t)(c + 1) * 256);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);
        }
      }
    }
    // (5) cell gradients in the accumulator layout (lane: rows 4q + e, cell lr) -> dzs (every workgroup), dz / dc (og == 0)
    {
      float* zt = dzs + ((size_t)(rt * 8 + ct * 4) * 16) * AB_DZS + lr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dh = acc0[e] + acc1[e];
        float dai = 0.f, dj = 0.f, daf = 0.f, dao = 0.f, dcn_out = edc[e];
        if (erok[e] & (J.t < elen[e])) {
          const float gi = eg[e][0], gj = eg[e][1], gf = eg[e][2], go = eg[e][3];
          const float tc = tanhf(ecn[e]);
          dao = dh * tc * go * (1.f - go);
          const float dcn = edc[e] + dh * go * (1.f - tc * tc) + dao * ewo;
          daf = dcn * ecp[e] * gf * (1.f - gf);
          dai = dcn * gj * gi * (1.f - gi);
          dj = dcn * gi * (1.f - gj * gj);
          dcn_out = dcn * gf + dai * ewi + daf * ewf;
        }
        float* zr = zt + (size_t)(4 * q + e) * AB_DZS;
        zr[0] = dai; zr[(size_t)16 * AB_DZS] = dj; zr[(size_t)32 * AB_DZS] = daf; zr[(size_t)48 * AB_DZS] = dao;
        if (og == 0 && erok[e]) {
          const int row = r0 + rt * 16 + 4 * q + e;
          float* g = J.dzo + (size_t)row * H4 + ecell;
          g[0] = dai; g[H] = dj; g[2 * H] = daf; g[3 * H] = dao;
          J.dc_out[(size_t)row * H + ecell] = dcn_out;          // (masked rows: the carried gradient passes through)
        }
      }
    }
  } else {
    // fully_connected stage: dz slice [64 rows][8 k-blocks] global -> dzs; weights from the row-major K
    const int nkb = (H4 + 15) >> 4;
    for (int i = tid; i < 64 * 8 * 4; i += 512) {
      const int f4 = i & 3, j = (i >> 2) & 7, row = i >> 5;
      const int k = (kg * 8 + j) * 16 + f4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < N && k < H4 && kg * 8 + j < nkb) v = *reinterpret_cast<const float4*>(J.gates + (size_t)(r0 + row) * H4 + k);
      *reinterpret_cast<float4*>(dzs + ((size_t)((row >> 4) * 8 + j) * 16 + (row & 15)) * AB_DZS + f4 * 4) = v;
    }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const int n = min(og * AB_COLS + cgp * 48 + ci * 16 + lr, ncols - 1);
      const float* wrow = J.K + (size_t)(J.n_begin + n) * H4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = (kg * 8 + j) * 16 + 4 * q;
        const float4 b = *reinterpret_cast<const float4*>(wrow + min(k, H4 - 4));
        const bool ok = k < H4;
        bw[ci][j] = make_float4(ok ? b.x : 0.f, ok ? b.y : 0.f, ok ? b.z : 0.f, ok ? b.w : 0.f);
      }
    }
  }
  __syncthreads();
  // (6) partial [64 x 192] of dz . K^T over this workgroup's 8 k-blocks
  f32x4 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) acc[i][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* ab = dzs + ((size_t)(rh * 2 * 8) * 16 + lr) * AB_DZS + 4 * q;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float4 a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(ab + ((size_t)(i * 8 + j) * 16) * AB_DZS);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bw[ci][j].x, acc[i][ci], 0, 0, 0);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bw[ci][j].y, acc[i][ci], 0, 0, 0);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bw[ci][j].z, acc[i][ci], 0, 0, 0);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bw[ci][j].w, acc[i][ci], 0, 0, 0);
    }
  }
  // (7) partial tile -> ws[kg][row][col].  The accumulators hold (rows 4q + e, column lr): scalar stores would be 64-byte pieces
  // (2 us of the launch); each wave turns its 32 x 48 patch through its own slice of the (now free) dm image and stores float4 rows.
  {
    float* ts = smem + (size_t)w * (32 * 52);             // [32 rows][48 cols + 4 pad] per wave (8 x 6.5 KB <= the dm image: mode 0
                                                          //  holds 64 x SA >= 64 x 52 floats; mode 1 launches get the same size)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int e = 0; e < 4; ++e) ts[(size_t)(i * 16 + 4 * q + e) * 52 + ci * 16 + lr] = acc[i][ci][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the wave's own LDS writes (no other wave touches this slice)
    const int colbase = og * AB_COLS + cgp * 48;
#pragma unroll
    for (int u = 0; u < 6; ++u) {                         // 32 rows x 12 float4 = 384 = 6 per lane
      const int idx = u * 64 + lane, rr = idx / 12, c4 = idx - rr * 12;
      const int row = r0 + rh * 32 + rr, col = colbase + c4 * 4;
      if (row < N && col < ncols && !(abl & 8))
        *reinterpret_cast<float4*>(J.ws + ((size_t)kg * N + row) * J.ldw + col) = *reinterpret_cast<const float4*>(ts + (size_t)rr * 52 + c4 * 4);
    }
  }
}


// k_bwd_ab: KG = groups of 32 cells (mode 1: of 8 k-blocks), OG = groups of 192 output columns, nrg = groups of 64 rows
size_t bwd_ab_plan(BwdABJobs& jobs, float* ws_base) {
  size_t off = 0;
  const int n = jobs.n;
  double cost[MAXJ]; int nx[MAXJ], x0[MAXJ], nb[MAXJ], w1[MAXJ]; bool grouped[MAXJ]; Place* pp[MAXJ];
  for (int i = 0; i < n; ++i) cost[i] = (double)(jobs.j[i].n_end - jobs.j[i].n_begin) * jobs.j[i].N * jobs.j[i].H4;
  plan_groups(n, cost, nx, x0, grouped);
  for (int i = 0; i < n; ++i) {
    BwdABJob& b = jobs.j[i];
    const int ncols = b.n_end - b.n_begin;
    b.KG = b.mode == 0 ? (b.H + AB_CELLS - 1) / AB_CELLS : (((b.H4 + 15) >> 4) + 7) / 8;
    b.OG = (ncols + AB_COLS - 1) / AB_COLS; b.nrg = (b.N + 63) / 64;
    b.ldw = (ncols + 3) & ~3;
    b.ws = ws_base ? ws_base + off : nullptr;
    off += (size_t)b.KG * b.N * b.ldw;
    nb[i] = b.KG * b.OG * b.nrg; w1[i] = 1; pp[i] = &b.pl;
  }
  bool over = false;           // a group's workgroups must fit one round at one per CU (32 CUs per XCD)
  for (int i = 0; i < n; ++i) over = over || (grouped[i] && (nb[i] + nx[i] - 1) / nx[i] > 32);
  if (over) for (int i = 0; i < n; ++i) { grouped[i] = false; nx[i] = 8; x0[i] = 0; }
  const int gp = plan_rounds(n, nb, w1, nx, x0, grouped, pp);
  fill_map(jobs.map, n, gp, [&](int q) -> const Place& { return jobs.j[q].pl; });
  return off;
}
void launch_bwd_ab(const BwdABJobs& jobs, hipStream_t s) {
  int grid = 8;
  size_t lds = 0;
  for (int i = 0; i < jobs.n; ++i) {
    grid = std::max(grid, 8 * jobs.j[i].pl.se);
    lds = std::max(lds, ab_lds_floats(jobs.j[i].ldm) * sizeof(float));
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_ab), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("RSRGAN_AB_ABLATE"); ablate = e ? atoi(e) : 0; }
  BwdABJobs jj = jobs; jj.ablate = ablate;
  hipLaunchKernelGGL(k_bwd_ab, dim3(grid), dim3(512), lds, s, jj);
}

// the reduce launch alone (after k_bwd_ab): places of k_bwd_b_red for jobs whose ws / ldw / KG are already set
void bwd_b_red_plan(BwdBJobs& jobs) {
  const int n = jobs.n;
  int nbr[MAXJ], w1[MAXJ], nx8[MAXJ], x00[MAXJ]; bool nogroup[MAXJ]; Place* pr[MAXJ];
  for (int i = 0; i < n; ++i) {
    BwdBJob& b = jobs.j[i];
    nbr[i] = (b.N * (b.n_end - b.n_begin) + 255) / 256; w1[i] = 1; nx8[i] = 8; x00[i] = 0; nogroup[i] = false; pr[i] = &b.plr;
  }
  const int gr = plan_rounds(n, nbr, w1, nx8, x00, nogroup, pr);
  fill_map(jobs.mapr, n, gr, [&](int q) -> const Place& { return jobs.j[q].plr; });
}
void launch_bwd_b_red(const BwdBJobs& jobs, hipStream_t s) {
  int br = 8;
  for (int i = 0; i < jobs.n; ++i) br = std::max(br, 8 * jobs.j[i].plr.se);
  hipLaunchKernelGGL(k_bwd_b_red, dim3(br), dim3(256), 0, s, jobs);
}

#endif
