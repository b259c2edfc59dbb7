// kernels.h -- host-callable launchers of the gfx950 kernels (kernels.hip, gemm.hip).
// Every matrix is row-major with a leading dimension that is a multiple of 4
// floats, 16-byte aligned, and ZERO in its padding columns (the arena is
// memset once and kernels only ever write valid columns).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsr {

constexpr int MAXJ = 8;   // (layer, t) jobs fused into one launch of a step kernel

// Placement of a job's workgroups in a launch (filled by the launchers, read by pick_job in kernels.hip).  Block id b has the XCD
// slot x = b & 7 (observed on MI355X: block b runs on XCD (b + c) % 8 for a constant c, tools/ubench/l2.hip) and the round
// s = b >> 3.  A job owns the slots [x0, x0 + nx) (mod 8) during the rounds [sb, se): its local block is
// lb = (s - sb) * nx + ((x - x0) & 7), valid while lb < nb, and the tile kernels read lb as (column block lb % w, row block lb / w)
// with w = the column-block count rounded up to a multiple of nx -- so a column block (= a slice of the layer's weights) always
// runs on the same XCD, and a HEAVY job (a generator layer) owns a group of XCDs of its own: that group's L2 holds only this
// layer's weights and operands (2.3 MB instead of an eighth of every layer's plus all the streaming traffic), the blocks that
// produce a layer's h / m are the ones whose XCD consumes them in the next launch, and the heavy blocks sit one per CU.
// nx = 8, x0 = 0 is the plain contiguous layout (lb = b - 8 sb).
struct Place { int x0, nx, sb, se, nb, w; };
// block id -> job index (0xFF = no job), filled from the places by the launchers: the kernels look their job up with one scalar
// load instead of testing all MAXJ places (valid = 0: grid larger than the table, the kernels test the places)
constexpr int JOBMAP_MAX = 768;
struct JobMap { int valid; unsigned char job[JOBMAP_MAX]; };
struct PlanItem { int ncol, nrow; double cost; int mult; };       // mult: nx must divide it (0 = any nx)
// assigns every job a Place (`nb_of(job, nx)` = its block count when it owns nx slots per round); returns the grid size
int plan_places(int n, const PlanItem* items, Place* out);

// device-resident scalars (model.dyn): what the reference changes with tf.assign between steps
enum { DYN_G_LR = 0, DYN_D_LR = 1, DYN_LAMBDA = 2, DYN_D_REAL = 3, DYN_D_FAKE = 4, DYN_L2 = 5, DYN_CLIP = 6,
       DYN_B1 = 7, DYN_B2 = 8, DYN_EPS = 9, DYN_EMA = 10, DYN_ADAM_LRT = 11, DYN_ADAM_LRT_D = 12, DYN_COUNT = 16 };

// ---------------------------------------------------------------- step kernels
// Forward phase 1: z = zx | bias  (+ x_t . Kx) + m_{t-1} . Kh ; gates; c_t ; h_t
struct FwdGateJob {
  const float* x;      // [N][ldx] layer input at t, or nullptr when zx holds the x-part
  const float* KxT;    // [4H][ldx]   (k-contiguous transposed copy of kernel rows 0..I)
  const float* m;      // [N][ldm] carried recurrent state m_{t-1}
  const float* KhT;    // [4H][ldm]
  const float* Wsw;    // fragment-tiled copy of [KxT | KhT] (or of KhT alone when zx holds the x-part): tiles [gate][cell block][k-block][256], see SwizzleJob
  const float* zx;     // [N][4H] precomputed x_t.Kx + bias, or nullptr
  const float* bias;   // [4H] (used when zx == nullptr)
  const float* wf; const float* wi; const float* wo;   // peepholes [H]
  const float* c_prev; // [N][H]
  float* c_out;        // [N][H]
  float* gates;        // [N][4H]  out: sigma(i), tanh(j), sigma(f), sigma(o)   (may alias zx)
  float* h;            // [N][ldh] out: sigma(o)*tanh(c)
  const int* len;      // [N]
  int ldx, ldm, ldh, t, N, H;
  int nblk_c;
  Place pl;
  // num_proj=None layers (m = h): the epilogue also writes the carried state, the masked output and the residual sum
  float* np_m_out; float* np_out; const float* np_res_in; float* np_res_out;
};
struct FwdGateJobs { int n; float forget_bias; FwdGateJob j[MAXJ]; JobMap map; };

// tf.contrib.rnn.DropoutWrapper(cell, output_keep_prob) (models/lstm.py:99-102, res_lstm_l.py:96-99): the output of (layer, t)
// times mask / keep, a new mask in every training run.  mask[row][col] = [ (splitmix64(key + row*P + col) >> 40) < thr ] with
// key = splitmix64(splitmix64(seed ^ run * 0xD1342543DE82EF95) + tag); `run` is read from device memory (*ctr, advanced by
// k_drop_tick once per training run) so that a replayed hipGraph draws new masks.  tests/helpers.py dropout_mask restates it.
struct DropSpec {
  const unsigned long long* ctr;
  unsigned long long seed, tag;
  unsigned thr;
  float keep;
};
void launch_drop_tick(unsigned long long* ctr, hipStream_t s);

// Forward phase 2: m_t = h_t . Wp ; dynamic_rnn masking ; optional residual add
struct FwdProjJob {
  const float* h;       // [N][ldh]
  const float* WpT;     // [P][ldh]
  const float* WpT_sw;  // fragment-tiled copy of WpT: tiles [column block][k-block][256], or nullptr (fully_connected stages)
  const float* m_prev;  // [N][ldm]
  float* m_out;         // [N][ldm] carried state
  float* out;           // [N][ldm] masked output (0 for t >= len)
  const float* res_in;  // [N][ldm] or nullptr
  float* res_out;       // [N][ldm] = out + res_in
  const int* len;       // nullptr: every row is live (fully_connected stage)
  const float* bias;    // [P] added to the product (fully_connected stage), or nullptr
  const float* noise;   // [N][P] added to `out` only (gaussian_noise_layer on D's input), or nullptr
  int ldh, ldm, ldo, P, t, N;     // ldm: stride of m_prev/m_out/res_*, ldo: stride of out
  int nblk_c;
  Place pl;
  DropSpec drop;        // DropoutWrapper(output_keep_prob) on `out` (and on the residual sum's `out` term); ctr == nullptr: off
};
struct FwdProjJobs { int n; FwdProjJob j[MAXJ]; JobMap map; };

// Backward phase A: dm = mask*(dout_t + dm_state); dh = dm . Wp^T ; gate grads -> dz ; dc
struct BwdAJob {
  const float* dout;    // [N][ldm] grad of the masked output at t (nullptr = 0)
  const float* dmst;    // [N][ldm] carried grad of the m state
  const float* Wp;      // [H][ldm]
  const float* Wp_sw;   // fragment-tiled copy of Wp: tiles [cell block][k-block][256], or nullptr
  float* dmt;           // [N][ldm] out: total dm at t (0 on masked rows)
  float* gates;         // [N][4H]  in: activations, out: dz = (dai, dj, daf, dao)
  const float* c_prev;  // [N][H]   c_{t-1} (state before the step)
  const float* c_cur;   // [N][H]   c_t
  const float* wf; const float* wi; const float* wo;
  float* dc;            // [N][H]   carried grad of c (in/out)
  const int* len;
  int ldm, P, t, N, H;
  int nblk_c;
  Place pl;
  DropSpec drop;        // the forward job's: dout is multiplied by mask / keep as it is loaded
};
struct BwdAJobs { int n; BwdAJob j[MAXJ]; JobMap map; };

// Backward phase B: [dx_t | dm_rec] = dz_t . K^T restricted to kernel rows [n_begin, n_end)
struct BwdBJob {
  const float* dz;      // [N][4H]
  const float* K;       // [(I+R)][4H] TF-layout kernel
  const float* Ksw;     // fragment-tiled copy of rows [n_begin, n_end) of K: tiles [row block from n_begin][k-block][256], or nullptr
  float* dx;            // [N][lddx] receives columns n < I (nullptr if n_begin >= I)
  float* dmst;          // [N][ldm]  columns n >= I:  dmst = (mask ? 0 : dmst) + value
  const int* len;
  int I, n_begin, n_end, lddx, ldm, t, N, H4;
  int dx_accumulate;    // dx += instead of =
  int nblk_c;
  Place pl;             // k_bwd_b (one launch of 32x16 / 16x16 tiles)
  // split-K form (k_bwd_bp + k_bwd_b_red): partial tiles ws[KG][N][ldw], KG groups of kpg k-blocks
  float* ws;
  int ldw, KG, kpg, ncg, nrg;
  Place plp, plr;       // k_bwd_bp (lb -> K slice lb % KG, output tile lb / KG) and k_bwd_b_red (256 outputs per block)
};
struct BwdBJobs { int n; BwdBJob j[MAXJ]; JobMap map, mapr; };      // map: k_bwd_b or k_bwd_bp (whichever runs), mapr: k_bwd_b_red

// The launchers place the jobs (Place above) and size the grid themselves; `total_blocks` arguments are ignored hints.
// kb_max = largest 16-float k-block count of any job in the launch (selects how many waves split K).
inline int job_blocks(int nblk_c, int N, int rows = 32) { return ((nblk_c + 7) & ~7) * ((N + rows - 1) / rows); }
int fwd_gates_rows();
void launch_fwd_gates(const FwdGateJobs& jobs, int total_blocks, int kb_max, hipStream_t s);
void launch_fwd_proj(const FwdProjJobs& jobs, int total_blocks, int kb_max, hipStream_t s);
int bwd_a_cells();                 // cells per column block of the phase-A kernel in use (BwdAJob::nblk_c = ceil(H / bwd_a_cells()))
void launch_bwd_a(const BwdAJobs& jobs, int total_blocks, int kb_max, hipStream_t s);
void launch_bwd_b(const BwdBJobs& jobs, int total_blocks, int kb_max, hipStream_t s);
// split-K phase B: fills ws/ldw/KG/... of every job (ws_base: >= bwd_b_ws_floats(jobs) floats) and launches both kernels
size_t bwd_b_plan(BwdBJobs& jobs, float* ws_base);
void launch_bwd_b_splitk(const BwdBJobs& jobs, hipStream_t s);
// ---- persistent discriminator recurrence (dpersist.hip) ----
constexpr int DP_MAXL = 3;
constexpr int DP_CTL_GEN = 0, DP_CTL_DONE = 1, DP_CTL_ERR = 2, DP_CTL_WORDS = 4;   // err: 0, or 1 + the first workgroup whose bounded wait expired
struct DPersistLayer {
  const float *K, *bias, *wi, *wf, *wo, *Wp;      // TF-layout kernel [(I+P)][4H], bias [4H], peepholes [H], projection [H][ldP]
  float *gates, *c, *h, *mst, *out;               // the layer's stash (model.h LstmStash)
  const float* in;                                // layer 0: the stack's input [T][N][ldI] (forward)
  float* dmt;                                     // backward: [T][N][ldP] total dm per step (the projection's weight gradient reads it)
  int I, P, ldP, ldH, ldI;
};
struct DPersistArgs {
  DPersistLayer L[DP_MAXL];
  int nl, N, T, H;
  const int* len;
  unsigned long long* gran;                       // dpersist_granule_bytes(nl, N, T), zeroed once at allocation
  unsigned* ctl;                                  // control block [DP_CTL_*]: generation (starts at 1), finished workgroups, sticky error
  float forget_bias;
  const float* dout_top;                          // backward: [T][N][ld_dout] gradient of the top layer's masked outputs
  int ld_dout;
  // the trailing form of the backward launch (dpersist_dev.h, inside gpersist.hip k_glstm_bwd_dt: the G-run, no weight gradients): layer 0's input gradient is added
  // to dy [T][N][ld_dy] step by step, and d(generator outputs)(t) = dy(t) . fc_w^T (fc_w: the generator's output FC [fc_P][ld_fcw])
  // lands in dtop [T][N][ld_dtop], armed with 0xFF bytes by the caller, where k_glstm_bwd's top layer polls it
  float* dy;
  const float* fc_w;
  float* dtop;
  int ld_dy, ld_fcw, ld_dtop, fc_P;
  // the trailing form of the FORWARD launch (inside gpersist.hip k_glstm_fwd_dt: D(G(x)) a few steps behind the generator): the FC
  // workgroups form y(t) = m_top(t) . fc_w + fc_b from the generator's top-layer chunks (fc_w [fc_P][ld_fcw], y -> dy [T][N][ld_dy]),
  // hand y(t) + noise to layer 0 as four partial sums and write it to dtop [T][xd_Ns][ld_dtop] at rows xd_row0.. (the input rows the
  // discriminator's weight gradients read); noise [N][I] or null
  const float* fc_b;
  const float* noise;
  int xd_Ns, xd_row0;
  // forward launches over the rows [row0, row0 + N) of a stash that is Ns rows tall (0: N) -- the D-run's two discriminator calls write
  // the two halves of the stacked stash its BPTT reads (model.cpp Model::d_backward); len points at the launch's first row
  int Ns, row0;
  // the trailing forms (dp_fwdt_body / dp_bwdt_body / dp_fcb_body): 1 = the second 16-row tile of every tile pair holds padding rows only
  // (GPersistArgs::nrt): its phase is two barriers -- no sweep, no product, no publication, no stash rows
  int nrt;
  // the stand-alone backward launch with the weight gradients inside (dpersist.hip dp_dw_body; null: none): per (layer, 16-row tile)
  // partial sums [dw_stride floats], progress words [(layer, tile)][T + 1][4] zeroed once at allocation.  Needs L[l].in / ldI of
  // every layer (layer l > 0: the masked outputs of the layer below)
  float* dw_ws;
  unsigned* dw_flag;
  size_t dw_stride;
};
int dpersist_dw_grid(int nl, int N);
size_t dpersist_dw_stride(const DPersistArgs& a);
size_t dpersist_dw_flag_bytes(int nl, int N, int T);
bool dpersist_dw_supported(const DPersistArgs& a);
size_t dpersist_granule_bytes(int nl, int N, int T);
int dpersist_trail_grid(int nl, int N);
size_t dpersist_trail_lds_bytes();
bool dpersist_trail_supported(const DPersistArgs& a);
bool dpersist_supported(const DPersistArgs& a);
int dpersist_grid(int nl, int N);                 // workgroups of a launch over N rows
size_t dpersist_lds_bytes();
void launch_dlstm_fwd(const DPersistArgs& a, hipStream_t s);
void launch_dlstm_fwd_t(const DPersistArgs& a, hipStream_t s);    // the 2-tile, 12-wave form (N % 32 == 0), stand-alone
void launch_dlstm_bwd(const DPersistArgs& a, hipStream_t s);      // gates: activations in, dz out; needs c, dmt, dout_top
// ---- persistent generator recurrence (gpersist.hip): the forward pass of a stack of large projected LSTM cells ----
constexpr int GP_MAXL = 4;
constexpr int GP_ROWS = 32;                       // batch rows per row group (two 16-row MFMA tiles)
constexpr int GP_THREADS = 768;                   // 12 waves per workgroup
struct GPersistLayer {
  const float *KxT, *KhT;                         // k-contiguous transposed copies of the kernel: [4H][ldI], [4H][ldP] (zero padding)
  const float *bias, *wi, *wf, *wo, *Wp;          // bias [4H], peepholes [H], projection [H][ldP]
  float *gates, *c, *h, *mst, *out;               // the layer's stash (model.h LstmStash)
  const float* in;                                // layer 0: the stack's input [T][N][ldI] (forward)
  float* dmt;                                     // backward: [T][N][ldP] total dm per step (the projection's weight gradient reads it)
  int I, P, ldI, ldP, ldH;
  float* res_out;                                 // residual stacks (GPersistArgs::res): [T][N][ldP] s_l = out_l + s_{l-1}: the next layer's input, the output FC's for the top layer
};
struct GPersistArgs {
  GPersistLayer L[GP_MAXL];
  int nl, N, T, H;
  int NT, NC;                                     // 4-cell gate tiles per workgroup, workgroups per (row group, layer)
  const int* len;
  unsigned long long *gran1, *gran2;              // hop 1 (partial projections, ring of three steps: armed once, gpersist_arm), hop 2 (m chunks, one slot per step: armed by every launch)
  unsigned* ctl;                                  // control block [DP_CTL_*]
  float forget_bias;
  unsigned long long* gran3;                      // backward: the partial input gradients a layer hands to the layer below -- layer 0 to its own reducers (nl + 1 rings of GP_XR steps); armed once
  const float* dout_top;                          // backward: [T][N][ld_dout] gradient of the top layer's masked outputs
  int ld_dout;
  float* din0;                                    // backward: [T][N][ld_din0] gradient of the stack's input (null: not wanted)
  int ld_din0;
  // models/res_lstm_l.py:101-194: inputs_{l+1} = outputs_l + inputs_l (inputs_1 = the stack's input, L[0].in; needs I == P everywhere).
  // gran2 then holds a second region of the same size: the running sums s_l (forward) / their gradients (backward), one slot per step.
  int res;
  // ring slots told apart by the parity of the ring pass in every word's lowest bit instead of sentinels that somebody has to put back
  // (gpersist.hip gp_store_t; the projected kernels only): no re-arming stores
  int tags;
  // dout_top is being written WHILE this launch runs (dpersist.hip k_dlstm_bwd_trail, the G-run): armed with 0xFF bytes, the top
  // layer's reducers poll their 16-byte piece of a step past the caches until no word carries that pattern
  int dout_trail;
  // k_glstm_fwd_dt: the discriminator's forward recurrence follows inside this launch (RES: the top layer publishes its running sum too)
  int fwd_trail;
  // a launch over ngl row groups starting at grp0 (0: all): gpersist_plan sets ngl = 1 when the whole stack's workgroups do not fit the device
  int ngl, grp0;
  // live 16-row tiles of every row group (0: both).  1: rows 16..31 of the group are padding rows of a padded model (Bt <= 16: their length
  // is 0 in every batch and everything they own in the stashes stays at the zeros of the allocation), so their tile lane does not run at
  // all: no recurrent product, no cells, no hand-off pieces -- half the bytes of a step on the fabric
  int nrt;
  // bit 0: a lane's off-chain work waits for the lane's publication (gp_fwd_body: the X waves' product of step t + 1 and the R waves' stash
  // store of step t start when the G waves have issued the partial projections of step t: LDS counter cnt_j) instead of colliding with it
  int sched;
};
constexpr int GP_TMAX = 2046;                     // longest launch (slot offsets are 32-bit; a longer batch takes the launch-per-phase path)
bool gpersist_plan(GPersistArgs& a);              // fills NT / NC; false: shape not supported
void gpersist_arm_bytes(void* p, size_t bytes, hipStream_t s);      // 0xFF-fill as a kernel (a fill node is no dependable predecessor of a persistent launch inside a replayed graph)
extern thread_local int g_gemm_workers;                      // worker slots of a stream-K GEMM launch (gemm.hip)
int gpersist_grid(const GPersistArgs& a);         // workgroups of a launch (all must be resident at once)
int gpersist_dt_grid(const GPersistArgs& a, const DPersistArgs& d);      // ... of k_glstm_bwd_dt / k_glstm_fwd_dt
void launch_glstm_fwd_dt(const GPersistArgs& a, const DPersistArgs& d, hipStream_t s);   // the generator's forward recurrence with D(G(x)) trailing it (ONE launch; a.fwd_trail = 1)
size_t dpersist_fwdt_lds_bytes();
void launch_glstm_bwd_dt(const GPersistArgs& a, const DPersistArgs& d, hipStream_t s);   // the generator's BPTT with the discriminator's trailing BPTT in front (ONE launch; a.dout_trail = 1, d.dtop armed)
size_t gpersist_lds_bytes();                      // LDS of a workgroup (the larger of the two kernels')
int device_cu_count();                            // hipDeviceProp_t::multiProcessorCount of the current device (queried once)
// true when `grid` workgroups of `threads` threads and `lds_bytes` of LDS are on the device AT THE SAME TIME (asked of the device itself:
// a probe launch; RSRGAN_RESIDENT_PROBE=0 trusts multiProcessorCount alone)
bool resident_probe(int grid, int threads, size_t lds_bytes);
size_t gpersist_gran1_bytes(const GPersistArgs& a);
size_t gpersist_gran2_bytes(const GPersistArgs& a);
size_t gpersist_gran3_bytes(const GPersistArgs& a);
void gpersist_arm(const GPersistArgs& a, hipStream_t s);          // once after allocation: the "not written" pattern in every slot of gran1 / gran3
void launch_glstm_fwd(const GPersistArgs& a, hipStream_t s);
// the unprojected form (num_proj=None: P == H <= 512, H % 16 == 0): one hand-off per step, the all-gather of h (round 5)
bool gpersist_np_plan(GPersistArgs& a, int nt_force = 0);      // nt_force: 2 / 4 = that many gate tiles per workgroup only (0: the smaller one that fits the device)
size_t gpersist_np_gran2_bytes(const GPersistArgs& a);
size_t gpersist_np_lds_bytes();
void launch_glstm_np_fwd(const GPersistArgs& a, hipStream_t s);
size_t gpersist_np_gran1_bytes(const GPersistArgs& a);            // the unprojected BPTT (NT == 2 only): state-gradient ring, input-gradient rings
size_t gpersist_np_gran3_bytes(const GPersistArgs& a);
void gpersist_np_arm(const GPersistArgs& a, hipStream_t s);
void launch_glstm_np_bwd(const GPersistArgs& a, hipStream_t s);   // gates: activations in, dz out; needs c, dout_top; no input gradient for layer 0
void launch_glstm_bwd(const GPersistArgs& a, hipStream_t s);      // gates: activations in, dz out; needs c, dmt, dout_top; no input gradient for layer 0
extern long long g_chain_launches;
void launch_floor_chain(float* a, float* b, int n, int mode, hipStream_t s);

// ---------------------------------------------------------------- batched GEMM
// C[M,N] (+)= A.B (+bias)(lrelu).  a_kc: A(m,k)=A[m*lda+k] else A[k*lda+m];
// b_kc: B(k,n)=B[n*ldb+k] else B[k*ldb+n].
// Row map of a GEMM operand (rows_per > 0): row r lives at float offset (r / rows_per) * outer + (r % rows_per) * inner instead of
// r * ld -- an overlapping-window view of a channels-last activation [sample][position][channel] (SEGAN-style strided conv1d as a
// GEMM: inner = stride * channels, a row = kwidth * channels contiguous floats).  For a k-contiguous operand the mapped index is the
// row (m or n); for an x-contiguous operand it is k.
struct GemmRowMap { int rows_per; long long outer; long long inner; };
void launch_gemm_mapped(const float* A, int lda, const GemmRowMap& ma, const float* A2, int lda2, int M1, bool a_kc, const float* B, int ldb,
                        bool b_kc, float* C, int ldc, int M, int N, int K, const float* bias, int act, float alpha,
                        bool accumulate, hipStream_t s, float* ws, size_t ws_floats);
// up to 4 products of one shape in one k_gemm16 launch (+ one reduce launch): C[p] (+)= [A[p] | A2[p]]^T . B[p], operands [K][M1 | M - M1], [K][N]
constexpr int GEMM16_MAXB = 4;
struct Gemm16Batch { const float* A[GEMM16_MAXB]; const float* A2[GEMM16_MAXB]; const float* B[GEMM16_MAXB]; float* C[GEMM16_MAXB]; int n; };
void launch_gemm16_batch(const Gemm16Batch& bt, int lda, int lda2, int M1, int ldb, int ldc, int M, int N, int K, bool accumulate,
                         hipStream_t s, float* ws, size_t ws_floats);
void launch_gemm2(const float* A, int lda, const float* A2, int lda2, int M1, bool a_kc, const float* B, int ldb, bool b_kc,
                  float* C, int ldc, int M, int N, int K, const float* bias, int act, float alpha, bool accumulate,
                  hipStream_t s, float* ws, size_t ws_floats);   // rows >= M1 of an m-contiguous A come from A2
constexpr int GEMM_MAXB = 4;
bool launch_gemm_batch(int nb, const float* const* A, int lda, const float* const* A2, int lda2, int M1, const float* const* B, int ldb,
                       float* const* C, int ldc, int M, int N, int K, bool accumulate, hipStream_t s, float* ws, size_t ws_floats);
void launch_gemm(const float* A, int lda, bool a_kc, const float* B, int ldb, bool b_kc,
                 float* C, int ldc, int M, int N, int K,
                 const float* bias, int act, float alpha, bool accumulate, hipStream_t s,
                 float* splitk_ws = nullptr, size_t splitk_ws_floats = 0);

// ---------------------------------------------------------------- layout / pointwise
struct StagePack { const float* src; float* dst; int D, ld; };              // [B][T][D] -> [T][B][ld]
struct StageCopy { const void* src; void* dst; int n; };                     // n 32-bit words
struct StageJobs { StagePack pack[2]; StageCopy copy[4]; int B, T; int Bt; };      // Bt > 0: the caller has Bt <= B rows (destination stride B)
void launch_stage_inputs(const StageJobs& j, hipStream_t s);
void launch_pack_tm(const float* src_bm, float* dst_tm, int B, int T, int D, int ld, hipStream_t s);      // [B,T,D] -> [T][B][ld]
void launch_unpack_bm(const float* src_tm, int ld, float* dst_bm, int B, int T, int D, hipStream_t s, int Bt = 0);    // [T][B][ld] -> [Bt or B,T,D]
// xd[t][b] = labels[t][b] + noise_r[b]   (b <  B)   (only when with_real)
// xd[t][B+b or b] = y[t][b] + noise_f[b]
void launch_build_d_input(const float* lab_tm, const float* y_tm, const float* noise_r, const float* noise_f,
                          float* xd, int B, int T, int D, int ld, bool with_real, hipStream_t s);
// dst[t][row0 + b] = src[t][b] + noise[b]   for b < B; dst has Ns rows per time step
void launch_add_noise_rows(const float* src_tm, const float* noise, float* dst, int B, int T, int D, int ld,
                           int Ns, int row0, hipStream_t s);
void launch_transpose(const float* src, int lds, float* dst, int ldd, int R, int C, hipStream_t s);        // dst[c][r] = src[r][c]
void launch_fill(float* p, size_t n, float v, hipStream_t s);
// Fragment-tiled ("swizzled") weight copies.  The step kernels feed v_mfma_f32_16x16x4_f32 with a B fragment per lane
// l = (q = l >> 4, lr = l & 15): four consecutive k of output column lr.  Read from a row-major k-contiguous matrix that is 16 rows
// x 64 B per wave-load, which the memory pipeline serves at ~15 B/clk/CU even from a hot L2 (tools/ubench/l2.hip section D:
// 3.4x slower than a contiguous 1 KB wave-load).  A tiled copy stores the 16 columns x 16 k of one MFMA k-block as
// [q][lr][4] = 256 floats, so lane l reads bytes [16 l, 16 l + 16) of a contiguous 1 KB tile; tiles of one column block are
// contiguous along k.  Logical matrix M[c][k], c < 16*nct, k < 16*nkb, zero outside the source:
//   gates == 0: M[c][k] = src[(c0 + c) * ld + k]                       (c < C, k < K1)               -- rows of a k-contiguous matrix
//   gates >= 1: c = g * 16*ncb + cell (ncb = ceil(H/16) per gate); column of the source = g * H + cell;
//               k < kx: row k (valid k < I) ; k >= kx: row I + (k - kx) (valid k - kx < P) ; M = src[row * ld + column]
//               (the [x | m] concatenation of the gates product over a TF-layout kernel [(I+P)][4H]; kx = padded width of x, or 0)
struct SwizzleJob { const float* src; float* dst; int ld, gates, H, C, c0, K1, I, P, kx, nct, nkb, blk_base; };
struct SwizzleList { int n; SwizzleJob j[24]; };
void launch_swizzle_many(SwizzleList& sl, hipStream_t s);
inline size_t swizzle_floats(int nct, int nkb) { return (size_t)nct * nkb * 256; }
struct TransposeJob { const float* src; float* dst; int lds, ldd, R, C, blk_base; };   // dst[c][r] = src[r][c]
struct TransposeList { int n; TransposeJob j[16]; };
void launch_transpose_many(TransposeList& tl, hipStream_t s);
// implicit-GEMM conv2d (conv.hip): Ft = prepared filter (launch_conv_prep), forward or data gradient (flip)
size_t conv_prep_floats(int S, int fw, int C);
void launch_conv_prep(const float* F, int ldf_src, int S, int fw, int Cin, int Cout, bool flip, float* Ft, hipStream_t s);
bool conv_fwd_supported(int C, int N, int S, int W, int fw);
void launch_conv_fwd(const float* in, int ldc_in, int C, const float* Ft, const float* bias, bool relu, float* out, int ldc_out, int N,
                     int R, int S, int W, int fw, hipStream_t s,
                     const float* mask = nullptr /* [positions][ldc_out]: out = 0 where mask <= 0 (relu' fused into the data gradient) */);
bool conv_wgrad_supported(int C, int N, int S, int W, int fw);
size_t conv_wgrad_ws_floats(int C, int R, int S, int W, int fw);
void launch_conv_wgrad(const float* in, int ldc_in, int C, const float* d, int ldc_d, int N, float* dW, int ldw, float* ws, int R, int S,
                       int W, int fw, hipStream_t s,
                       float* db = nullptr /* also the bias gradient colsum(d)[0:N] */);
// R-CED patch matrix (conv2d SAME as GEMM) and its adjoint; col2im needs C % 4 == 0
void launch_im2col(const float* src, size_t row_stride, int ldc, int C, int S, int W, int kh, int kw, float* col, int ldk, size_t M,
                   hipStream_t s);
void launch_expand_c4(const float* src, int ld_src, int n, float* dst, size_t rows, hipStream_t s);
void launch_col2im(const float* dcol, int ldk, int C, int S, int W, int kh, int kw, float* dst, int ldc, size_t M, hipStream_t s);
struct ZeroList { int n; float* p[32]; unsigned len[32]; };        // many small buffers zeroed by ONE launch
void launch_zero_many(const ZeroList& zl, hipStream_t s);
struct ColsumsBatch { const float* dz[4]; const float* cprev[4]; const float* ccur[4]; float* db[4]; float* dwi[4]; float* dwf[4]; float* dwo[4]; int n; };
void launch_lstm_colsums_batch(const ColsumsBatch& bt, int rows, int H, float* scratch, hipStream_t s);      // scratch: n x 64 x 7H floats
void launch_lstm_colsums(const float* dz, const float* cprev, const float* ccur, float* db, float* dwi, float* dwf, float* dwo,
                         int rows, int H, float* scratch /* >= 64*7*H floats */, hipStream_t s);
void launch_lrelu_bwd(const float* hval, float* d, size_t rows, int cols, int ld, float alpha, hipStream_t s); // d *= (h>0?1:alpha)
// tf.nn.dropout (models/dnn.py:116-121, discriminator_dnn.py:100-105): y = y / keep * mask in place; the mask of element (r, c) is
// [ (splitmix64(key + r*cols + c) >> 40) < thr ] with thr = floor(keep * 2^24) -- oracle side: tests/helpers.py dropout_mask
void launch_dropout_fwd(float* y, size_t rows, int cols, int ld, unsigned long long key, unsigned thr, float keep, hipStream_t s);
// backward through dropout + the ReLU under it, given the dropped output y: d = y > 0 ? d / keep : 0
void launch_dropout_bwd(const float* y, float* d, size_t rows, int cols, int ld, float keep, hipStream_t s);
// col sums over `rows` rows: out[c] = sum_r a[r*lda+c] * (b ? b[r*ldb+c] : 1)
void launch_colsum_tall(const float* a, int lda, float* out, int rows, int cols, float* scratch, size_t scratch_floats, hipStream_t s);

// ---- batch_norm(renorm=True, scale=True) of the frame-level nets (bn.hip) ----
struct BnVars {           // one layer's <scope>/BatchNorm/* variables (device pointers into the ParamSet)
  float *beta, *gamma;                                  // [C] trainable
  float *mm, *mv, *rm, *rmw, *rs, *rsw;                  // moving_mean, moving_variance, renorm_mean, renorm_mean_weight [1], renorm_stddev, renorm_stddev_weight [1]
};
constexpr int BN_STAT_ROWS = 6;                          // per call: mean, stddev, r, d, a, b   (y = z*a + b), each [ldc]
// `calls` consecutive calls of `rows` rows each with their own batch moments (statistics slots 0 .. calls-1 of `stat`)
void launch_bn_forward(const float* z, int ldz, float* y, int ldy, int rows, int cols, const BnVars& v, float* stat, int ldc, bool training,
                       bool relu, float* scratch, size_t scratch_floats, hipStream_t s, int calls = 1);
void launch_bn_backward(float* dy, int ldd, const float* y, int ldy, const float* z, int ldz, int rows, int cols, const float* stat, int ldc,
                        float* dbeta, float* dgamma, bool accumulate, bool relu, float* sums, float* scratch, size_t scratch_floats,
                        hipStream_t s, int calls = 1);
void launch_bn_commit(int cols, const BnVars& v, const float* stat, int ldc, int times, hipStream_t s);
struct BnCommit { BnVars v; const float* stat; int cols, ldc, times0, times1; };   // update ops: call 0 x times0, then call 1 x times1
struct BnCommitList { int n; BnCommit e[24]; };
void launch_bn_commit_many(const BnCommitList& cl, hipStream_t s);
void launch_colsum(const float* a, int lda, const float* b, int ldb, float* out, int rows, int cols,
                   float* scratch /* >= 64*cols floats */, hipStream_t s);

// LSGAN: logits [T*Nd][ldl] col 0.  rows with (r % Nd) < n_real use target_real and go to
// loss[0], the others target_fake and loss[1]; loss[2] = loss[0]+loss[1].  Means over T*n_real /
// T*(Nd-n_real) entries.  dlogits[r][0] = 2*(l-target)/count  (nullptr = skip).
// discriminator_lstm's head, forward and backward in one pass (kernels.hip k_dhead1): top [T*Nd][ldt] -> logits, losses, dlogits, dout,
// the FC's gradients gw [dR][ldw] / gb.  part: ceil(T*Nd / 64) x (DH_MAXR + 3) floats.  dR % 4 == 0, dR <= DH_MAXR.
constexpr int DH_MAXR = 64;
struct DHeadArgs {
  const float* top; int ldt; const float* w; int ldw; const float* b;
  float* logits; int ldl; float* dlogits; float* dout; int ldo; float* gw; float* gb;
  int T, Nd, n_real, dR; const float* t_real; const float* t_fake; float* loss3; float* part; int want_grads, want_wgrads;
  int Bp, Bt;                       // Bp > 0: every Bp rows of a frame are Bt utterances + padding rows that count in no mean
};
void launch_dhead(const DHeadArgs& a, hipStream_t s);
void launch_lsgan(const float* logits, int ldl, float* dlogits, int T, int Nd, int n_real,
                  const float* target_real, const float* target_fake, float* loss3, hipStream_t s,
                  bool clip_on = false, float clip_lo = 0.f, float clip_hi = 0.f, int Bp = 0, int Bt = 0);
void launch_build_joint(const float* x, int ldx, int off, int dim, const float* tail, int ldt, int Dt, float* joint, int ldj,
                        int row0, int R, hipStream_t s);
void launch_slice_cols(const float* src, int lds, int off, float* dst, int ldd, int R, int C, hipStream_t s);
// g_mse = 0.5*D*mean((y-lab)^2) ; dy (+)= lambda*(y-lab)/(B*T)   (dy nullptr = loss only)
void launch_mse(const float* y, const float* lab, int ld, float* dy, int rows, int D, const float* lambda,
                bool accumulate, float* loss_out, float* scratch /* >= 1024 floats */, hipStream_t s, int Bp = 0, int Bt = 0);   // Bp > 0: rows [Bt, Bp) of every Bp are padding
// losses[3] = adv + lambda*mse + l2
void launch_g_total(float* l4 /* adv,mse,l2,total */, const float* lambda, hipStream_t s);
void launch_copy_f(const float* src, float* dst, int n, hipStream_t s);

// ---------------------------------------------------------------- optimizer
struct ChunkTable {          // device arrays, one entry per 4096-float chunk of the flat buffer
  const int* tensor;         // tensor index of the chunk
  const int* off;            // float offset in the flat buffer
  const int* len;            // valid floats in the chunk
  const int* t_first;        // [n_tensors] first chunk of tensor
  const int* t_count;        // [n_tensors] chunks of tensor
  const int* t_l2;           // [n_tensors] 1 if the tensor takes the L2 term ("bias" not in name)
  int n_chunks, n_tensors;
};
void launch_sumsq(const float* g, const ChunkTable& ct, float* partial, hipStream_t s);
// adds the N/16 per-tile records of dw_ws into the gradient buffer (fixed order) and leaves every chunk's sum of squares in `partial`
// (k_sumsq's output: the update that follows in the same segment skips that launch); src[tensor] = the tensor's float offset inside a
// layer-0-tile-0 record (+ layer * ntiles * stride), or -1 for tensors the launch did not compute (their gradient is only squared)
void launch_dw_reduce(const float* ws, size_t stride, int ntiles, const long long* src, float* g, const ChunkTable& ct, float* partial, hipStream_t s);
// l2: g += l2_scale*w on flagged tensors; partial[c] = sumsq(w) of flagged chunks (else 0)
void launch_l2(const float* w, float* g, const ChunkTable& ct, const float* l2_scale, float* partial, hipStream_t s);
void launch_l2_total(const float* partial, int n_chunks, const float* l2_scale, float* out, hipStream_t s);
// dyn: device scalars {lr, clip, beta1, beta2, eps, lr_t(adam), ema_decay}
void launch_apply_sgd(float* w, const float* g, float* ema, const ChunkTable& ct, const float* partial,
                      const float* dyn, hipStream_t s);
void launch_apply_adam(float* w, const float* g, float* m, float* v, float* ema, const ChunkTable& ct,
                       const float* partial, const float* dyn, hipStream_t s, int lrt_slot = DYN_ADAM_LRT);
void launch_adam_tick(float* dyn, int* t, double b1, double b2, hipStream_t s, int lr_slot = DYN_G_LR, int lrt_slot = DYN_ADAM_LRT);
// dense <-> padded flat copies (checkpoint / parity injection)
void launch_pad_copy(const float* dense, float* padded, int rows, int cols, int ld, bool to_padded, hipStream_t s);

}  // namespace rsr
