// dpersist_dev.h -- device-side pieces of the persistent discriminator recurrences shared by dpersist.hip and gpersist.hip (the
// generator's BPTT launch hosts the trailing form of the discriminator's: k_glstm_bwd_dt).  Included inside namespace rsr.
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int DP_NQ = 4;            // cell quarters per row tile (H = 64 * DP_NQ)
constexpr int DP_KB = 3;            // k-blocks of 16 of the recurrent / input width (P <= 48)
constexpr int DP_SLOT = DP_KB * 64 * 4;     // granules per (layer, tile, step, quarter): the consumer's A-fragment order [kb][lane][u]
constexpr int DP_HS = 72;           // LDS row stride of the h tile (floats): 18 float4, == 2 mod 16 (conflict-free fragment reads)

__device__ __forceinline__ float4 dp_sel(bool c, const float4 a, const float4 b) {      // componentwise: a float4 ?: becomes a pointer select + scratch
  return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
// Cell non-linearities on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): the cell phase of a workgroup is
// 1024 elements x 5 transcendentals on ONE wave per SIMD, and libm's expf / tanhf / IEEE division made it 1.4 us of a step
// (profiles/r3_dpersist_trace.txt).  |error| <= ~2e-7 absolute: tanh switches to its odd series below 0.1 where 1 - 2/(1+e^2x) cancels.
__device__ __forceinline__ float dp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float dp_tanh(float x) {
  const float x2 = x * x;
  const float ser = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.05396825f * x2)));
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x));
  return fabsf(x) < 0.1f ? ser : big;
}

// 16-byte write-through accesses carrying two granules each (8-byte scalar sc1 stores are one fabric write per lane: 2.7x the time
// per byte, MI355X_MICROARCH.md "stores of each flavour"; the 8-byte halves of a 16-byte sc1 access are observed untorn).  Inline
// asm: the compiler has no 16-byte agent-scope atomic.  (Its vmcnt bookkeeping does not see these: the loads wait inside the asm,
// and an untracked store only makes a later compiler-placed wait stricter, vmcnt retiring in order.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dp_store2(gu64* p, unsigned tag, float v0, float v1) {
  const u32x4 x = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  // (s_nop: a store of more than 8 bytes followed by a write of its data registers needs a wait state; the compiler inserts it
  // for its own stores, not behind an asm statement)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// 16 bytes of ordinary data write-through (what another workgroup of the SAME launch reads behind a progress word: sc1 stores + sc1 loads)
__device__ __forceinline__ void dp_store4_wt(float* p, const float4 v) {
  const u32x4 x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// the 12 granules of this lane in one slot: 6 loads in flight, one wait
__device__ __forceinline__ void dp_load12(const gu64* g, int lane, u32x4 (&x)[6]) {
  const gu64* p0 = g + (size_t)lane * 4;
  asm volatile(
      "global_load_dwordx4 %0, %6, off sc1\n\t"
      "global_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %6, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %3, %6, off offset:2064 sc1\n\t"
      "global_load_dwordx4 %4, %7, off sc1\n\t"
      "global_load_dwordx4 %5, %7, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5])
      : "v"(p0), "v"(p0 + 512)
      : "memory");
}
// two slots in one round trip
__device__ __forceinline__ void dp_load24(const gu64* g0, const gu64* g1, int lane, u32x4 (&x)[6], u32x4 (&y)[6]) {
  const gu64* p0 = g0 + (size_t)lane * 4;
  const gu64* p1 = g1 + (size_t)lane * 4;
  asm volatile(
      "global_load_dwordx4 %0, %12, off sc1\n\t"
      "global_load_dwordx4 %1, %12, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %12, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %3, %12, off offset:2064 sc1\n\t"
      "global_load_dwordx4 %4, %13, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off offset:16 sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %14, off offset:16 sc1\n\t"
      "global_load_dwordx4 %8, %14, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %9, %14, off offset:2064 sc1\n\t"
      "global_load_dwordx4 %10, %15, off sc1\n\t"
      "global_load_dwordx4 %11, %15, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]),
        "=&v"(y[4]), "=&v"(y[5])
      : "v"(p0), "v"(p0 + 512), "v"(p1), "v"(p1 + 512)
      : "memory");
}
__device__ __forceinline__ bool dp_take(const u32x4 (&x)[6], unsigned tag, float (&v)[DP_KB * 4]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    v[2 * k] = __uint_as_float(x[k].x); v[2 * k + 1] = __uint_as_float(x[k].z);
    ok &= x[k].y == tag && x[k].w == tag;
  }
  return __all(ok);
}
// One wave re-reads granules until all carry their tag: 12 of slot `gm_` (tag tm; skipped when null) and 12 of slot `gx_` (tag tx;
// skipped when null).  false on time-out (or when another workgroup has failed).
__device__ __forceinline__ bool dp_sweep2(const gu64* gm_, unsigned tm, float (&vm)[DP_KB * 4], const gu64* gx_, unsigned tx, float (&vx)[DP_KB * 4],
                                          int lane, gu32* err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool dm = gm_ == nullptr, dx = gx_ == nullptr;                   // (wave-uniform)
  for (unsigned spins = 0;; ++spins) {
    u32x4 x[6], y[6];
    if (!dm && !dx) { dp_load24(gm_, gx_, lane, x, y); dm = dp_take(x, tm, vm); dx = dp_take(y, tx, vx); }
    else if (!dm) { dp_load12(gm_, lane, x); dm = dp_take(x, tm, vm); }
    else if (!dx) { dp_load12(gx_, lane, x); dx = dp_take(x, tx, vx); }
    if (dm && dx) return true;
    if ((spins & 63) == 63) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull ||                     // 1 s at 100 MHz (a peer that is merely not scheduled yet must not fail the launch)
          __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The TRAILING form of the backward launch (round 5; the G-run: gan_rnn_placeholder.py:246-256 differentiates g_adv through the
// discriminator INTO the generator).  The generator's BPTT (gpersist.hip k_glstm_bwd) consumes d(outputs)(t) = dy(t) . W_out^T in the
// same descending time order in which the discriminator's BPTT produces d g_adv / d y(t): launched one after the other the second
// waits 0.38 ms for the first (+ two GEMMs) although it could start a few steps behind it.  This form rides INSIDE the generator's
// launch (gpersist.hip k_glstm_bwd_dt: its first workgroups run the bodies below, the rest gp_bwd_body): two launches on two streams
// gave the same overlap, but nothing makes two hardware queues dispatch them in the right order -- one launch has one order.
//   * every workgroup of the launch must be resident at once and k_glstm_bwd holds 228 of the 256 CUs, so the discriminator's BPTT
//     runs on HALF its usual workgroups: workgroup (layer, PAIR of 16-row tiles, cell quarter), 16 for 64 rows, 12 waves (the launch's
//     block size): waves 0-3 compute tile 0, waves 8-11 tile 1 (same cells, same weights, a tile's state each), waves 4-7 gather and
//     publish for both tiles in turn -- a step is two phases (barriers A, B per tile), the hand-off of one tile travels while the
//     other computes;
//   * plus one FC workgroup per row tile that turns layer 0's input-gradient partials into the generator's top-layer gradient:
//       dy(t) = [lambda-weighted mse term, written by k_mse1 before the launch] + sum of the four quarters' dx0 partials  (-> memory:
//               the output FC's parameter gradients read it afterwards)
//       d(outputs)(t) = dy(t) . W_out^T  (16 x Dout . Dout x P: 12 MFMAs per 16-column tile)  -> dtop, write-through
//     dtop is armed with the all-ones pattern at the head of the run; the reducers of the generator's top layer poll their own 16-byte
//     piece of step t (GPersistArgs::dout_trail);
//   * the launch's register budget is the generator's: 168 per wave.  A SPILLED register here costs BOTH halves dearly (a scratch
//     reload is a memory round trip of microseconds while the generator floods the fabric: 32 spilled registers made the launch 2.4 ms
//     instead of 1.5), so nothing sits in registers that need not: the K_h fragments (state-gradient product, critical path) live in
//     LDS, the input-gradient product moved to the gather waves (dz staged in LDS, K_x fragments requested per phase while the compute
//     waves work), the step's operands are requested one phase ahead into the registers the phase has just freed.
// No weight gradients here (the G-run takes none of the discriminator): no dz stash, no dmt.  16 + 4 + 228 = 248 workgroups; this half
// runs at ~9 us per step, the generator's at 15.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int DP_TPW = 2;           // 16-row tiles per workgroup of the trailing form
struct DpTrailLds {
  // (the small, hot arrays first: a DS instruction's immediate offset reaches 64 KB -- behind that the compiler keeps every address of an
  //  unrolled access in a register of its own, and spills them)
  float stage[4][16 * DP_HS];                     // dz of the step and tile [gate][row][cell of the quarter]: the gather waves' B operand of the input-gradient product
  float psum[4][DP_KB][64][4];                    // the compute waves' partial dm_state tiles (one tile at a time)
  float part_m[DP_TPW][DP_NQ][DP_KB][64][4];      // swept partials of dm_state (from step t+1), per tile
  float kx_lds[4][DP_KB][4][64][4];               // K_h fragments (A operand of the state-gradient product; the K_x ones are read from memory by the gather waves)
  float part_x[DP_TPW][2][DP_NQ][DP_KB][64][4];   // swept partials of dout (dx of the layer above), per tile, by parity of the step
  int dead;
};

template <bool DEAD1>
__device__ __forceinline__ void dp_bwdt_body(const DPersistArgs& a, const unsigned gen, DpTrailLds& S, const int bid) {
  const int RPn = a.N >> 5, RTn = a.N >> 4, ncl = a.nl * RPn;       // tile pairs
  const int cl = bid % ncl, cq = bid / ncl;
  const int l = cl / RPn, rp = cl - l * RPn;
  const DPersistLayer L = a.L[l];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, H4 = 4 * H, T = a.T, N = a.N, P = L.P, ldP = L.ldP, I = L.I;
  const bool top = l == a.nl - 1;
  constexpr bool dead1 = DEAD1;                                     // tile 1 of the pair: padding rows only (DPersistArgs::nrt == 1: the caller picks the instantiation)
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  const size_t slot_stride_t = (size_t)DP_NQ * DP_SLOT;
  auto edge = [&](int layer, int e, int r) -> gu64* {
    return (gu64*)a.gran + ((size_t)((layer * 2 + e) * RTn + r) * T) * slot_stride_t;
  };

  if (w >= 4 && w < 8) {
    // ---------------- gather waves (4-7) ----------------
    __builtin_amdgcn_s_setprio(3);
    const int j = w - 4;
    float vm[DP_KB * 4], vx[DP_KB * 4];
    auto put = [&](float (*part)[DP_KB][64][4], const float (&v)[DP_KB * 4]) {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        *reinterpret_cast<float4*>(&part[j][kb][lane][0]) = make_float4(v[kb * 4], v[kb * 4 + 1], v[kb * 4 + 2], v[kb * 4 + 3]);
    };
    auto fail = [&]() { if (lane == 0) { S.dead = 1; __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } };
    auto publish = [&](float (*ps)[DP_KB][64][4], gu64* dst) {
      const float4 p0 = *reinterpret_cast<const float4*>(&ps[0][j][lane][0]), p1 = *reinterpret_cast<const float4*>(&ps[1][j][lane][0]);
      const float4 p2 = *reinterpret_cast<const float4*>(&ps[2][j][lane][0]), p3 = *reinterpret_cast<const float4*>(&ps[3][j][lane][0]);
      gu64* go_ = dst + ((size_t)j * 64 + lane) * 4;
      dp_store2(go_, gen, ((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y);
      dp_store2(go_ + 2, gen, ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
    };
    // The input-gradient product dx partial^T = K_x . dz^T of this quarter (every layer: layer 0's goes to the FC workgroups) runs HERE, on
    // the gather waves: the compute waves have no register left for it (168 per wave beside the generator's BPTT; spilled registers
    // cost BOTH halves of the launch dearly: 1.5 -> 2.4 ms), these waves have.  Wave j < 3 owns output rows p = 16 j .. 16 j + 15 over
    // ALL 64 cells x 4 gates of the quarter (the sum over the four compute waves happens inside the accumulator): K_x fragments
    // resident, A[p = 16 j + lr][cell 16 wc + 4 q + u] of gate g (rows p >= I are row I - 1 again: every consumer masks them); B = the
    // dz tile the compute waves staged in LDS before barrier B.  The result waits in registers for its slot a step later.
    // (the fragments are 64 registers: beside the 48 of a sweep's loads too many to keep -- they are requested behind barrier A, while the
    //  compute waves work, 16 KB per wave and phase out of the L2, and have arrived by barrier B)
    float4 kxA[4][4];
    const float* const kxrow = L.K + (size_t)min(16 * j + lr, I - 1) * H4 + cq * 64 + 4 * q;
    auto kx_request = [&]() {
      const float* kq = kxrow;
      asm volatile("" : "+v"(kq));                                  // (opaque: the same addresses every step -- hoisted out of the loop the fragments are resident again)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int wc_ = 0; wc_ < 4; ++wc_)
          kxA[g][wc_] = *reinterpret_cast<const float4*>(kq + g * H + 16 * wc_);
    };
    f32x4 dxacc[DP_TPW];
    auto dx_product = [&](f32x4& acc) {
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int wc_ = 0; wc_ < 4; ++wc_) {
          const float4 bz = *reinterpret_cast<const float4*>(&S.stage[g][lr * DP_HS + 16 * wc_ + 4 * q]);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kxA[g][wc_].x, bz.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kxA[g][wc_].y, bz.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kxA[g][wc_].z, bz.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kxA[g][wc_].w, bz.w, acc, 0, 0, 0);
        }
    };
    auto publish_dx = [&](const f32x4& acc, gu64* dst) {
      gu64* go_ = dst + ((size_t)j * 64 + lane) * 4;
      dp_store2(go_, gen, acc[0], acc[1]);
      dp_store2(go_ + 2, gen, acc[2], acc[3]);
    };
    if (!top) {                                                    // dout of step T-1 for the prologue
#pragma unroll
      for (int i = 0; i < DP_TPW; ++i) {
        if (i == 1 && dead1) continue;
        const gu64* gx = edge(l + 1, 1, 2 * rp + i) + (size_t)j * DP_SLOT;
        if (!dp_sweep2(nullptr, gen, vm, gx + (size_t)(T - 1) * slot_stride_t, gen, vx, lane, err)) fail();
        put(S.part_x[i][(T - 1) & 1], vx);
      }
    }
    __syncthreads();                                               // P
    if (S.dead) return;
    for (int t = T - 1; t >= 0; --t) {
#pragma unroll
      for (int i = 0; i < DP_TPW; ++i) {
        const int r = 2 * rp + i;
        const gu64* gm = edge(l, 0, r) + (size_t)j * DP_SLOT;
        const gu64* gx = edge(min(l + 1, a.nl - 1), 1, r) + (size_t)j * DP_SLOT;
        gu64* gout_m = edge(l, 0, r) + (size_t)cq * DP_SLOT;
        gu64* gout_x = edge(l, 1, r) + (size_t)cq * DP_SLOT;
        const bool wm = t < T - 1, wx = !top && t > 0;
        if (i == 1 && dead1) {                                     // (the padding tile's phase: the two barriers)
          __syncthreads();
          if (S.dead) return;
          __syncthreads();
          continue;
        }
        if (wm || wx) {
          if (!dp_sweep2(wm ? gm + (size_t)(t + 1) * slot_stride_t : nullptr, gen, vm,
                         wx ? gx + (size_t)(t - 1) * slot_stride_t : nullptr, gen, vx, lane, err)) fail();
          if (wm) put(S.part_m[i], vm);
          if (wx) put(S.part_x[i][(t - 1) & 1], vx);
        }
        __syncthreads();                                           // A(t, i)
        if (S.dead) return;
        if (j < 3 && t < T - 1) publish_dx(dxacc[i], gout_x + (size_t)(t + 1) * slot_stride_t);      // dx of step t+1 (in registers since B(t+1, i))
        if (j < 3) kx_request();
        __syncthreads();                                           // B(t, i): psum(t, i) and the dz tile are in LDS
        if (j < 3) {
          publish(S.psum, gout_m + (size_t)t * slot_stride_t);
          dx_product(dxacc[i]);                                    // (done before this wave arrives at the next barrier, behind which the dz tile is overwritten)
        }
      }
    }
    __syncthreads();                                               // C
    if (j < 3) {
#pragma unroll
      for (int i = 0; i < DP_TPW; ++i)
        if (!(i == 1 && dead1)) publish_dx(dxacc[i], edge(l, 1, 2 * rp + i) + (size_t)cq * DP_SLOT);
    }
    return;
  }

  // ---------------- compute waves: 0-3 take tile 0, 8-11 tile 1 (the same cells and weights, a tile's state each) ----------------
  const int wc = w & 3;
  auto compute = [&](auto my_c) {
  constexpr int my = decltype(my_c)::value;
  const int cb = cq * 64 + 16 * wc + 4 * q;                         // this lane's four cells
  // W_p rows in registers, the K_h fragments (A operand of the state-gradient product, on the step's critical path) in LDS -- the launch
  // shares the 168-register budget of the generator's BPTT, and whatever this half reads from MEMORY on its critical path (spilled
  // registers, weights) costs a round trip of microseconds while that one floods the fabric with its hand-offs.  The K_x fragments of
  // the input-gradient product, off the critical path behind barrier B, are read from memory there (12 KB per wave and phase, L2).
  float4 wpA[DP_KB];
  {
    const float* wrow = L.Wp + (size_t)(cq * 64 + 16 * wc + lr) * ldP;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const int k = 16 * kb + 4 * q;
      const float4 v = *reinterpret_cast<const float4*>(wrow + min(k, P - 4));
      wpA[kb] = dp_sel(k < P, v, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (my == 0) {                                                  // (one copy of the fragments serves both tiles)
#pragma unroll
      for (int pt = 0; pt < DP_KB; ++pt) {
        const int p = 16 * pt + lr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 vh = *reinterpret_cast<const float4*>(L.K + (size_t)(I + min(p, P - 1)) * H4 + g * H + cb);
          *reinterpret_cast<float4*>(&S.kx_lds[wc][pt][g][lane][0]) = dp_sel(p < P, vh, make_float4(0.f, 0.f, 0.f, 0.f));
        }
      }
    }
  }
  const float4 pwi = *reinterpret_cast<const float4*>(L.wi + cb), pwf = *reinterpret_cast<const float4*>(L.wf + cb);
  const float4 pwo = *reinterpret_cast<const float4*>(L.wo + cb);
  const float pi_[4] = {pwi.x, pwi.y, pwi.z, pwi.w}, pf_[4] = {pwf.x, pwf.y, pwf.z, pwf.w}, po_[4] = {pwo.x, pwo.y, pwo.z, pwo.w};
  const int rowb = 32 * rp + 16 * my + lr;                          // this lane's batch row
  const int lenF = a.len[rowb];
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  float4 mf[DP_KB];
#pragma unroll
  for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto sum_parts = [&](float (*part)[DP_KB][64][4], int width, float4 (&s)[DP_KB]) {
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const float4 p0 = *reinterpret_cast<const float4*>(&part[0][kb][lane][0]), p1 = *reinterpret_cast<const float4*>(&part[1][kb][lane][0]);
      const float4 p2 = *reinterpret_cast<const float4*>(&part[2][kb][lane][0]), p3 = *reinterpret_cast<const float4*>(&part[3][kb][lane][0]);
      s[kb] = dp_sel(16 * kb + 4 * q < width,
                     make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                                 ((p0.w + p1.w) + p2.w) + p3.w), make_float4(0.f, 0.f, 0.f, 0.f));
    }
  };
  f32x4 ccur = *reinterpret_cast<const f32x4*>(L.c + ((size_t)T * N + rowb) * H + cb);      // c_T
  // operands of my tile's next step: requested at the END of a phase, when the registers of the phase are dead -- they travel while the
  // other tile has the workgroup (this launch shares the 168-register budget of the generator's BPTT)
  f32x4 gt[4], cprev;
  float4 don[DP_KB];
  auto load_ops = [&](int t) {
    const size_t row = (size_t)t * N + rowb;
#pragma unroll
    for (int g = 0; g < 4; ++g) gt[g] = *reinterpret_cast<const f32x4*>(L.gates + row * H4 + g * H + cb);
    cprev = *reinterpret_cast<const f32x4*>(L.c + row * H + cb);
    if (top) {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        don[kb] = *reinterpret_cast<const float4*>(a.dout_top + row * a.ld_dout + min(16 * kb + 4 * q, P - 4));
    }
  };
  load_ops(T - 1);
  __syncthreads();                                                 // P
  if (S.dead) return;

  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int i = 0; i < DP_TPW; ++i) {
      if (i == my && !(my == 1 && dead1)) {
        __syncthreads();                                           // A(t, i)
        if (S.dead) return;
        const bool live = t < lenF;
        float4 dout[DP_KB];
        if (top) {
#pragma unroll
          for (int kb = 0; kb < DP_KB; ++kb) dout[kb] = dp_sel(16 * kb + 4 * q < P, don[kb], make_float4(0.f, 0.f, 0.f, 0.f));
        } else {
          sum_parts(S.part_x[i][t & 1], P, dout);
        }
        if (t < T - 1) {
          float4 ms[DP_KB];
          sum_parts(S.part_m[i], P, ms);
          const bool live_next = (t + 1) < lenF;                    // masked rows pass the carried gradient through
#pragma unroll
          for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = dp_sel(live_next, ms[kb], mf[kb]);
        }
        float4 dm[DP_KB];
#pragma unroll
        for (int kb = 0; kb < DP_KB; ++kb)
          dm[kb] = dp_sel(live, make_float4(dout[kb].x + mf[kb].x, dout[kb].y + mf[kb].y, dout[kb].z + mf[kb].z, dout[kb].w + mf[kb].w),
                          make_float4(0.f, 0.f, 0.f, 0.f));
        f32x4 dh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DP_KB; ++kb) {
          dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].x, dm[kb].x, dh, 0, 0, 0);
          dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].y, dm[kb].y, dh, 0, 0, 0);
          dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].z, dm[kb].z, dh, 0, 0, 0);
          dh = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].w, dm[kb].w, dh, 0, 0, 0);
        }
        // gate / cell gradients (kernels.hip k_bwd_a2): lane = row lr, cells cb + u
        float dz[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float gi = gt[0][u], gj = gt[1][u], gf = gt[2][u], go = gt[3][u];
          const float tc = dp_tanh(ccur[u]);
          const float dao = dh[u] * tc * go * (1.f - go);
          const float dcn = dc[u] + dh[u] * go * (1.f - tc * tc) + dao * po_[u];
          const float daf = dcn * cprev[u] * gf * (1.f - gf);
          const float dai = dcn * gj * gi * (1.f - gi);
          const float dj = dcn * gi * (1.f - gj * gj);
          dz[0][u] = live ? dai : 0.f; dz[1][u] = live ? dj : 0.f; dz[2][u] = live ? daf : 0.f; dz[3][u] = live ? dao : 0.f;
          dc[u] = live ? dcn * gf + dai * pi_[u] + daf * pf_[u] : dc[u];
        }
        ccur = cprev;
        f32x4 pa[DP_KB];
#pragma unroll
        for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 kh[DP_KB];
          __builtin_amdgcn_sched_barrier(0);                        // (one gate's fragments at a time: read ahead, the four sets are 48 registers again)
#pragma unroll
          for (int pt = 0; pt < DP_KB; ++pt) kh[pt] = *reinterpret_cast<const float4*>(&S.kx_lds[wc][pt][g][lane][0]);
#pragma unroll
          for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[pt].x, dz[g][0], pa[pt], 0, 0, 0);
#pragma unroll
          for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[pt].y, dz[g][1], pa[pt], 0, 0, 0);
#pragma unroll
          for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[pt].z, dz[g][2], pa[pt], 0, 0, 0);
#pragma unroll
          for (int pt = 0; pt < DP_KB; ++pt) pa[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[pt].w, dz[g][3], pa[pt], 0, 0, 0);
        }
#pragma unroll
        for (int pt = 0; pt < DP_KB; ++pt) *reinterpret_cast<f32x4*>(&S.psum[wc][pt][lane][0]) = pa[pt];
        {                                                          // the dz tile for the gather waves' input-gradient product
          const int so = lr * DP_HS + 16 * wc + 4 * q;
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(&S.stage[g][so]) = make_float4(dz[g][0], dz[g][1], dz[g][2], dz[g][3]);
        }
        __syncthreads();                                           // B(t, i)
        load_ops(max(t - 1, 0));                                   // (my next step's operands travel while the other tile has the workgroup)
      } else {
        __syncthreads();                                           // A(t, i), the other tile's
        if (S.dead) return;
        __syncthreads();                                           // B(t, i), the other tile's
      }
    }
  }
  __syncthreads();                                                 // C
  };
  if (w < 4) compute(std::integral_constant<int, 0>{}); else compute(std::integral_constant<int, 1>{});
}

// The FC workgroup of row tile fr (see above): per step, the four gather waves take a quarter's dx0 partials each; then every wave
// forms dy(t) of the tile and the 16-column tiles w, w + 8, w + 16 of dy(t) . W_out^T.
constexpr int DP_FCT = 3;           // 16-column tiles of d(outputs) per wave: P_fc <= 16 * 8 * DP_FCT
__device__ __forceinline__ void dp_fcb_body(const DPersistArgs& a, const unsigned gen, DpTrailLds& S, const int fr) {
  const int RTn = a.N >> 4;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, N = a.N, I = a.L[0].I, Pf = a.fc_P, ldt = a.ld_dtop;
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  const size_t slot_stride_t = (size_t)DP_NQ * DP_SLOT;
  const gu64* gx = (const gu64*)a.gran + ((size_t)((0 * 2 + 1) * RTn + fr) * T) * slot_stride_t + (size_t)(w & 3) * DP_SLOT;
  float (*part)[DP_KB][64][4] = S.part_m[0];
  if (a.nrt == 1 && (fr & 1)) {
    // a tile of padding rows (DPersistArgs::nrt): nobody publishes an input gradient for it and the generator does not read its rows of
    // d(outputs); they leave the armed pattern all the same (zeros: the gradient of rows of length 0), dy stays what the mse term left
    for (int e = tid; e < T * 16 * (ldt >> 2); e += (int)blockDim.x) {
      const int c4 = e % (ldt >> 2), rt = e / (ldt >> 2), rr = rt & 15, t = rt >> 4;
      float* dst = a.dtop + ((size_t)t * N + 16 * fr + rr) * ldt + 4 * c4;
      const f32x4 x = {0.f, 0.f, 0.f, 0.f};                          // (write-through like the live tiles': a generator lane that does run polls these rows past the caches)
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(x) : "memory");
    }
    return;
  }
  if (w >= 8) return;                                              // (eight waves; a barrier counts the surviving waves)
  // A operand: W_out[c = 16 ct + lr][p = 16 kb + 4 q + u] (zero beyond P_fc rows / I columns)
  float4 wA[DP_FCT][DP_KB];
#pragma unroll
  for (int n = 0; n < DP_FCT; ++n)
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const int c = 16 * (w + 8 * n) + lr, p = 16 * kb + 4 * q;
      const float4 v = *reinterpret_cast<const float4*>(a.fc_w + (size_t)min(c, Pf - 1) * a.ld_fcw + min(p, I - 4));
      wA[n][kb] = dp_sel(c < Pf && p < I, v, make_float4(0.f, 0.f, 0.f, 0.f));
    }
  const int row = 16 * fr + lr;
  float4 dyn[DP_KB];
  auto load_dy = [&](int t) {
    const float* p_ = a.dy + ((size_t)t * N + row) * a.ld_dy;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) dyn[kb] = *reinterpret_cast<const float4*>(p_ + min(16 * kb + 4 * q, I - 4));
  };
  load_dy(T - 1);
  __syncthreads();                                                 // (dead = 0 is visible)
  for (int t = T - 1; t >= 0; --t) {
    if (w >= 4) {
      float vm[DP_KB * 4], vx[DP_KB * 4];
      if (!dp_sweep2(nullptr, gen, vm, gx + (size_t)t * slot_stride_t, gen, vx, lane, err)) {
        if (lane == 0) { S.dead = 1; __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        *reinterpret_cast<float4*>(&part[w - 4][kb][lane][0]) = make_float4(vx[kb * 4], vx[kb * 4 + 1], vx[kb * 4 + 2], vx[kb * 4 + 3]);
    }
    __syncthreads();                                               // A: the four quarters' partials of step t are in LDS
    if (S.dead) return;
    float4 dyv[DP_KB];
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const float4 p0 = *reinterpret_cast<const float4*>(&part[0][kb][lane][0]), p1 = *reinterpret_cast<const float4*>(&part[1][kb][lane][0]);
      const float4 p2 = *reinterpret_cast<const float4*>(&part[2][kb][lane][0]), p3 = *reinterpret_cast<const float4*>(&part[3][kb][lane][0]);
      const float4 dx = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
      dyv[kb] = dp_sel(16 * kb + 4 * q < I, make_float4(dyn[kb].x + dx.x, dyn[kb].y + dx.y, dyn[kb].z + dx.z, dyn[kb].w + dx.w), make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (w == 0) {
      float* p_ = a.dy + ((size_t)t * N + row) * a.ld_dy;
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        if (16 * kb + 4 * q < I) *reinterpret_cast<float4*>(p_ + 16 * kb + 4 * q) = dyv[kb];
    }
    load_dy(max(t - 1, 0));
#pragma unroll
    for (int n = 0; n < DP_FCT; ++n) {
      const int c0 = 16 * (w + 8 * n) + 4 * q;                      // this lane's four columns of row `row`
      if (16 * (w + 8 * n) < ldt) {                                 // (wave-uniform)
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DP_KB; ++kb) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][kb].x, dyv[kb].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][kb].y, dyv[kb].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][kb].z, dyv[kb].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[n][kb].w, dyv[kb].w, acc, 0, 0, 0);
        }
        if (c0 < ldt) {
          u32x4 x = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = x[u] == 0xFFFFFFFFu ? 0x7FC00000u : x[u];      // (the armed pattern is nobody's value)
          float* dst = a.dtop + ((size_t)t * N + row) * ldt + c0;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(x) : "memory");
        }
      }
    }
    __syncthreads();                                               // B: `part` may be overwritten
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The FORWARD twin of the trailing form: two 16-row tiles per workgroup, 12 waves, the 168-register budget of the generator's launches
// (see above).  Waves 0-3 compute tile 0, 8-11 tile 1, 4-7 gather / project / publish / write the stash for both tiles in turn.  K_h
// fragments in LDS; the x-part of a tile's next step (K_x fragments requested from memory, 12 KB per wave) runs in the OTHER tile's
// phase; W_p fragments in the projecting waves' registers.  xin: layer 0 takes its input like the layers above take theirs -- four
// partial sums published as granules by whoever produces it (edge nl: the FC workgroups that follow the generator's top layer) --
// instead of reading rows from memory; such an input is not masked by the rows' lengths (dynamic_rnn does not mask its inputs).
// ------------------------------------------------------------------------------------------------------------------------
struct DpFwdTLds {
  float bp[64][8];                                // per cell of the quarter: {b_i, b_j, b_f, b_o, w_i, w_f, w_o, -} (28 registers a compute wave does not have)
  float stage[6][16 * DP_HS];                     // the phase's stash [gates i, j, f, o | c | h][row][cell of the quarter]; array 5 is the h tile the projection reads
  float part_m[DP_TPW][DP_NQ][DP_KB][64][4];      // swept partials of m(t-1), per tile
  float kh_lds[4][4][DP_KB][64][4];               // K_h fragments [compute wave][gate][k-block][lane]
  float part_x[DP_TPW][2][DP_NQ][DP_KB][64][4];   // swept partials of x(t+1), per tile, by parity of the step
  int dead;
};

template <bool DEAD1>
__device__ __forceinline__ void dp_fwdt_body(const DPersistArgs& a, const unsigned gen, DpFwdTLds& S, const int bid, const bool xin) {
  const int RPn = a.N >> 5, RTn = a.N >> 4, ncl = a.nl * RPn;
  const int cl = bid % ncl, cq = bid / ncl;
  const int l = cl / RPn, rp = cl - l * RPn;
  const DPersistLayer L = a.L[l];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, H4 = 4 * H, T = a.T, P = L.P, ldP = L.ldP, I = L.I;
  const bool xg = l > 0 || xin;                                     // the input arrives as granules
  const int Ns = a.Ns ? a.Ns : a.N, rw0 = a.row0;                   // (the rows of this launch inside a taller stash: stride Ns, first row row0; lengths are relative)
  gu32* err = (gu32*)a.ctl + DP_CTL_ERR;
  const size_t slot_stride_t = (size_t)DP_NQ * DP_SLOT;
  auto edge = [&](int layer, int r) -> gu64* { return (gu64*)a.gran + ((size_t)(layer * RTn + r) * T) * slot_stride_t; };
  const int lx = l > 0 ? l - 1 : a.nl;                              // whose granules are my input (edge nl: the producer of layer 0's input)
  constexpr bool dead1 = DEAD1;                                     // tile 1 of the pair: padding rows only (DPersistArgs::nrt == 1: the caller picks the instantiation)

  if (w >= 4 && w < 8) {
    // ---------------- gather waves ----------------
    __builtin_amdgcn_s_setprio(3);
    const int j = w - 4;
    float vm[DP_KB * 4], vx[DP_KB * 4];
    auto put = [&](float (*part)[DP_KB][64][4], const float (&v)[DP_KB * 4]) {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb)
        *reinterpret_cast<float4*>(&part[j][kb][lane][0]) = make_float4(v[kb * 4], v[kb * 4 + 1], v[kb * 4 + 2], v[kb * 4 + 3]);
    };
    auto fail = [&]() { if (lane == 0) { S.dead = 1; __hip_atomic_store(err, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } };
    // W_p^T fragments of this wave's 16 output columns (j < 3): A[col 16 j + lr][cell 16 kb + 4 q + u of the quarter]
    float4 wpA[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int col = 16 * min(j, 2) + lr;
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = L.Wp[(size_t)(cq * 64 + 16 * kb + 4 * q + u) * ldP + min(col, P - 1)];
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      wpA[kb] = dp_sel(col < P, make_float4(v[0], v[1], v[2], v[3]), make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (xg) {                                                      // x_0 for the prologue
#pragma unroll
      for (int i = 0; i < DP_TPW; ++i) {
        if (i == 1 && dead1) continue;
        if (!dp_sweep2(nullptr, gen, vm, edge(lx, 2 * rp + i) + (size_t)j * DP_SLOT, gen, vx, lane, err)) fail();
        put(S.part_x[i][0], vx);
      }
    }
    __syncthreads();                                               // P
    if (S.dead) return;
    const int t_end = cq == 0 ? T + 1 : T;
    for (int t = 0; t < t_end; ++t) {
#pragma unroll
      for (int i = 0; i < DP_TPW; ++i) {
        const int r = 2 * rp + i, r0 = 16 * r;
        const gu64* gm = edge(l, r) + (size_t)j * DP_SLOT;
        const gu64* gx = edge(lx, r) + (size_t)j * DP_SLOT;
        gu64* gout = edge(l, r) + (size_t)cq * DP_SLOT;
        const bool wm = t > 0, wx = xg && t + 1 < T;
        if (i == 1 && dead1) {                                     // (the padding tile's phase: its barriers -- tile 0's x-part runs in it)
          __syncthreads();
          if (S.dead) return;
          if (t < T) __syncthreads();
          continue;
        }
        if (wm || wx) {
          if (!dp_sweep2(wm ? gm + (size_t)(t - 1) * slot_stride_t : nullptr, gen, vm,
                         wx ? gx + (size_t)(t + 1) * slot_stride_t : nullptr, gen, vx, lane, err)) fail();
          if (wm) put(S.part_m[i], vm);
          if (wx) put(S.part_x[i][(t + 1) & 1], vx);
        }
        __syncthreads();                                           // A(t, i)
        if (S.dead) return;
        if (t < T) {
          __syncthreads();                                         // B(t, i): the h tile and the stash of (t, i) are in LDS
          if (j < 3) {
            f32x4 pm = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
              const float4 af = *reinterpret_cast<const float4*>(&S.stage[5][lr * DP_HS + 16 * kb + 4 * q]);
              pm = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].x, af.x, pm, 0, 0, 0);
              pm = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].y, af.y, pm, 0, 0, 0);
              pm = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].z, af.z, pm, 0, 0, 0);
              pm = __builtin_amdgcn_mfma_f32_16x16x4f32(wpA[kb].w, af.w, pm, 0, 0, 0);
            }
            gu64* go_ = gout + (size_t)t * slot_stride_t + ((size_t)j * 64 + lane) * 4;
            dp_store2(go_, gen, pm[0], pm[1]);
            dp_store2(go_ + 2, gen, pm[2], pm[3]);
          } else {
            // gather wave 3 writes the phase's stash from its LDS stage: whole 256-byte rows (read before this wave arrives at the next barrier)
            const int c4 = (lane & 15) * 4, rr = lane >> 4;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int row = 4 * rg + rr;
              const size_t grow = (size_t)t * Ns + rw0 + r0 + row;
#pragma unroll
              for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(L.gates + grow * H4 + g * H + cq * 64 + c4) = *reinterpret_cast<const float4*>(&S.stage[g][row * DP_HS + c4]);
              *reinterpret_cast<float4*>(L.c + (grow + Ns) * H + cq * 64 + c4) = *reinterpret_cast<const float4*>(&S.stage[4][row * DP_HS + c4]);
              *reinterpret_cast<float4*>(L.h + grow * L.ldH + cq * 64 + c4) = *reinterpret_cast<const float4*>(&S.stage[5][row * DP_HS + c4]);
            }
          }
        }
      }
    }
    return;
  }

  // ---------------- compute waves: 0-3 take tile 0, 8-11 tile 1 ----------------
  const int wc = w & 3;
  auto compute = [&](auto my_c) {
  constexpr int my = decltype(my_c)::value;
  const int cell = cq * 64 + 16 * wc + lr;                          // (fragment loads: the cell of this lane's A rows)
  const int cb = cq * 64 + 16 * wc + 4 * q;                         // this lane's four cells in the accumulator layout
  const int rowl = 32 * rp + 16 * my + lr, rowb = rw0 + rowl;      // the lane's row in the launch / in the stash
  if (my == 0) {                                                    // one copy of the K_h fragments serves both tiles
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) {
        float vh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vh[u] = L.K[(size_t)(I + min(16 * kb + 4 * q + u, P - 1)) * H4 + g * H + cell];
        asm volatile("" : "+v"(vh[0]), "+v"(vh[1]), "+v"(vh[2]), "+v"(vh[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u) vh[u] = 16 * kb + 4 * q + u < P ? vh[u] : 0.f;
        *reinterpret_cast<float4*>(&S.kh_lds[wc][g][kb][lane][0]) = make_float4(vh[0], vh[1], vh[2], vh[3]);
      }
  }
  if (my == 0 && lane < 16) {                                       // bias and peepholes of this wave's 16 cells -> LDS
    const int c_ = cq * 64 + 16 * wc + lane;
    *reinterpret_cast<float4*>(&S.bp[16 * wc + lane][0]) = make_float4(L.bias[c_], L.bias[H + c_], L.bias[2 * H + c_], L.bias[3 * H + c_]);
    *reinterpret_cast<float4*>(&S.bp[16 * wc + lane][4]) = make_float4(L.wi[c_], L.wf[c_], L.wo[c_], 0.f);
  }
  const float* const bpl = &S.bp[16 * wc + 4 * q][0];               // + 8 u: cell cb + u
  const int lenF = a.len[rowl];
  float cp[4] = {0.f, 0.f, 0.f, 0.f};
  float4 mf[DP_KB];
#pragma unroll
  for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(L.c + (size_t)rowb * H + cb) = make_float4(0.f, 0.f, 0.f, 0.f);       // slot 0 of the carried states is zero (cell.zero_state)
  if (cq == 0 && wc == 0) {
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb)
      if (16 * kb + 4 * q < P) *reinterpret_cast<float4*>(L.mst + (size_t)rowb * ldP + 16 * kb + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x4 accn[4];
  float4 xn[DP_KB];
  auto load_x = [&](int t) {
    const float* xr = L.in + ((size_t)t * Ns + rowb) * L.ldI;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) xn[kb] = *reinterpret_cast<const float4*>(xr + min(16 * kb + 4 * q, I - 4));
  };
  auto sum_parts = [&](float (*part)[DP_KB][64][4], int width, float4 (&s_)[DP_KB]) {
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
      const float4 p0 = *reinterpret_cast<const float4*>(&part[0][kb][lane][0]), p1 = *reinterpret_cast<const float4*>(&part[1][kb][lane][0]);
      const float4 p2 = *reinterpret_cast<const float4*>(&part[2][kb][lane][0]), p3 = *reinterpret_cast<const float4*>(&part[3][kb][lane][0]);
      s_[kb] = dp_sel(16 * kb + 4 * q < width,
                      make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                                  ((p0.w + p1.w) + p2.w) + p3.w), make_float4(0.f, 0.f, 0.f, 0.f));
    }
  };
  // x-part of step t: accn = bias + x_t . K_x (the K_x fragments of this wave's cells: requested here, 12 KB out of the L2)
  const float* const kxp = L.K + cell;                              // + (16 kb + 4 q + u) * H4 + g * H
  auto next_x = [&](int t) {
    const float* kq = kxp;
    asm volatile("" : "+v"(kq));                                    // (opaque: hoisted out of the loop the fragments are 48 registers held forever)
    float4 kx[DP_KB][4];
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = kq[(size_t)min(16 * kb + 4 * q + u, I - 1) * H4 + g * H];
        kx[kb][g] = make_float4(v[0], v[1], v[2], v[3]);
      }
    float4 xs[DP_KB];
    if (xg) {
      sum_parts(S.part_x[my][t & 1], I, xs);
      const bool livex = l == 0 || t < lenF;                        // dynamic_rnn's OUTPUT is zero past the row's length; the stack's input is not masked
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) xs[kb] = dp_sel(livex, xs[kb], make_float4(0.f, 0.f, 0.f, 0.f));
    } else {
#pragma unroll
      for (int kb = 0; kb < DP_KB; ++kb) xs[kb] = dp_sel(16 * kb + 4 * q < I, xn[kb], make_float4(0.f, 0.f, 0.f, 0.f));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 bq = *reinterpret_cast<const float4*>(bpl + 8 * u);
      accn[0][u] = bq.x; accn[1][u] = bq.y; accn[2][u] = bq.z; accn[3][u] = bq.w;
    }
    // (rows k >= I of the fragments repeat row I - 1: the x values there are zero)
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[kb][g].x, xs[kb].x, accn[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[kb][g].y, xs[kb].y, accn[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[kb][g].z, xs[kb].z, accn[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) accn[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[kb][g].w, xs[kb].w, accn[g], 0, 0, 0);
    }
  };
  auto store_m = [&](int t, const float4 (&mnew)[DP_KB], bool live_prev) {
    if (cq != 0 || wc != 0) return;
#pragma unroll
    for (int kb = 0; kb < DP_KB; ++kb)
      if (16 * kb + 4 * q < P) {
        *reinterpret_cast<float4*>(L.mst + ((size_t)t * Ns + rowb) * ldP + 16 * kb + 4 * q) = mf[kb];
        *reinterpret_cast<float4*>(L.out + ((size_t)(t - 1) * Ns + rowb) * ldP + 16 * kb + 4 * q) =
            dp_sel(live_prev, mnew[kb], make_float4(0.f, 0.f, 0.f, 0.f));
      }
  };

  if (!xg) load_x(0);
  __syncthreads();                                                 // P
  if (S.dead) return;
  constexpr bool idle = my == 1 && dead1;                           // this wave's tile is padding: barriers only
  if (!idle) next_x(0);
  if (!xg) load_x(min(1, T - 1));

  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int i = 0; i < DP_TPW; ++i) {
      __syncthreads();                                             // A(t, i)
      if (S.dead) return;
      if (idle) {
        __syncthreads();                                           // B(t, i)
      } else if (i == my) {
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = accn[g];
        if (t > 0) {
          float4 ms[DP_KB];
          sum_parts(S.part_m[my], P, ms);
          const bool live_prev = (t - 1) < lenF;
#pragma unroll
          for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = dp_sel(live_prev, ms[kb], mf[kb]);
          store_m(t, ms, live_prev);
        }
#pragma unroll
        for (int kb = 0; kb < DP_KB; ++kb) {
          float4 kh[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) kh[g] = *reinterpret_cast<const float4*>(&S.kh_lds[wc][g][kb][lane][0]);
          const float mv[4] = {mf[kb].x, mf[kb].y, mf[kb].z, mf[kb].w};
          const float* khf = &kh[0].x;
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(khf[4 * g + u], mv[u], acc[g], 0, 0, 0);
        }
        // the cell, in the accumulator layout: lane = row lr, cells cb .. cb+3
        const bool live = t < lenF;
        float hv[4], sg[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 pw = *reinterpret_cast<const float4*>(bpl + 8 * u + 4);      // {w_i, w_f, w_o} of cell cb + u
          const float cpv = cp[u];
          const float gi = dp_sigmoid(acc[0][u] + pw.x * cpv);
          const float gf = dp_sigmoid(acc[2][u] + a.forget_bias + pw.y * cpv);
          const float gj = dp_tanh(acc[1][u]);
          const float cn = gf * cpv + gi * gj;
          const float go = dp_sigmoid(acc[3][u] + pw.z * cn);
          const float hh = go * dp_tanh(cn);
          hv[u] = live ? hh : 0.f;
          sg[0][u] = live ? gi : 0.f; sg[1][u] = live ? gj : 0.f; sg[2][u] = live ? gf : 0.f; sg[3][u] = live ? go : 0.f;
          cp[u] = live ? cn : cpv;
        }
        {
          const int so = lr * DP_HS + 16 * wc + 4 * q;
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(&S.stage[g][so]) = make_float4(sg[g][0], sg[g][1], sg[g][2], sg[g][3]);
          *reinterpret_cast<float4*>(&S.stage[4][so]) = make_float4(cp[0], cp[1], cp[2], cp[3]);
          *reinterpret_cast<float4*>(&S.stage[5][so]) = make_float4(hv[0], hv[1], hv[2], hv[3]);
        }
        __syncthreads();                                           // B(t, i)
      } else {
        // the other tile's phase: my tile's x-part of the step it runs next -- step t + 1 behind my phase (t, 0), step t in front of my phase (t, 1)
        const int tn = my == 0 ? t + 1 : t;
        if (tn < T && (my == 0 || t > 0)) {
          next_x(tn);
          if (!xg) load_x(min(tn + 1, T - 1));
        }
        __syncthreads();                                           // B(t, i), the other tile's
      }
    }
  }
  if (cq == 0) {                                                   // the last step's m (block-uniform branch)
#pragma unroll
    for (int i = 0; i < DP_TPW; ++i) {
      __syncthreads();                                             // A(T, i)
      if (S.dead) return;
      if (i == my && !idle) {
        float4 ms[DP_KB];
        sum_parts(S.part_m[my], P, ms);
        const bool live_prev = (T - 1) < lenF;
#pragma unroll
        for (int kb = 0; kb < DP_KB; ++kb) mf[kb] = dp_sel(live_prev, ms[kb], mf[kb]);
        store_m(T, ms, live_prev);
      }
    }
  }
  };
  if (w < 4) compute(std::integral_constant<int, 0>{}); else compute(std::integral_constant<int, 1>{});
}
