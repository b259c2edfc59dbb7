// segan.h -- host state and kernel launchers of the SEGAN-style conv G/D (models/segan.py; BASELINE.json configs[4]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/rsrgan.h"
#include "model.h"

namespace rsr {

// segan.hip
void launch_pad_rows(const float* src, float* dst, int B, int L, int C, int pf, int pb, hipStream_t s);
void launch_conv1_fwd(const float* x, int ldx, int B, int L, int k, const float* W, int ldw, const float* bias, int C, float* z, int ldz, hipStream_t s);
void launch_conv1_wgrad(const float* x, int ldx, int B, int L, int k, const float* dz, int ldz, int C, float* dW, int ldw, float* scratch, size_t scratch_floats,
                        hipStream_t s);
void launch_tconv1(const float* S, int lds, int B, int Ls, int C, int Lt, int k, const float* W, int ldw, const float* bias, float* t, int ldt,
                   hipStream_t s);
void launch_sum_all(const float* src, int rows, int cols, int ld, float* out, float* scratch, hipStream_t s);
void launch_prep_tconv(const float* W, int ldw, int nb, int na, int e, int ne, float* dst, int ldd, hipStream_t s);
struct PrepTconvJob { const float* W; float* dst; int ldw, nb, na, e, ne, ldd; };
struct PrepTconvList { int n; int first[44]; PrepTconvJob j[44]; };      // <= 2 operands x 22 layers per launch (kernel argument: < 4 KB)
void launch_prep_tconv_many(PrepTconvList& pl, hipStream_t s);
void launch_interleave(const float* T0, const float* T1, int Q0, int Q1, int i00, int i01, int pl, const float* bias, float* T, int B, int Lt, int C,
                       hipStream_t s);
void launch_act_fwd(const float* z, int C, const float* alpha, float leak, float* out, int ldo, int coff, size_t rows, hipStream_t s);
void launch_copy_cols(const float* src, int lds, int soff, float* dst, int ldd, int doff, int C, size_t rows, bool accumulate, hipStream_t s);
void launch_act_bwd(const float* dy, int ldy, int coff, const float* z, int C, const float* alpha, float leak, const float* extra, float* dz, size_t rows,
                    hipStream_t s);
void launch_colred(int mode, const float* a, int lda, int coff, const float* b, int ldb, int C, size_t rows_per, int P, const float* coef, int ldcoef,
                   float leak, float* out, int ldo, bool accumulate, float* scratch, size_t scratch_floats, hipStream_t s);
void launch_vbn_coef(const float* sums, int lds, int P, int C, size_t rows_per, int B, float eps, const float* gamma, const float* beta,
                     const float* ref_coef, float* coef, int ldc, hipStream_t s);
void launch_vbn_apply(const float* h, int C, size_t rows_per, int P, const float* coef, int ldc, float leak, float* y, hipStream_t s);
void launch_vbn_bwd_coef(const float* sums, int lds, int P, int first_live, int C, size_t rows_per, int B, const float* gamma, float* coef, int ldc,
                         float* dgamma, float* dbeta, bool accumulate, hipStream_t s);
void launch_vbn_bwd_apply(const float* h, const float* dy, int C, size_t rows_per, int P, const float* coef, int ldc, float leak, float* dh, hipStream_t s);
void launch_build_joint1(const float* x, int Lx, const float* tail, int U, const float* noise, float* joint, int B, hipStream_t s);
void launch_dhead_fwd(const float* h, int R, int Ld, int C, int k, const float* W, const float* wfc, int ldfc, const float* bfc, float* conv_out, float* logits,
                      hipStream_t s);
void launch_dhead_bwd(const float* dlogit, int R, int Ld, int C, int k, const float* h, const float* conv_out, const float* W, const float* wfc, int ldfc,
                      float* dW, float* dwfc, float* dbfc, float* dh, hipStream_t s);
void launch_segan_lsgan(const float* logits, int B, int mode, int fake_pass, int P, float* dlogits, float* loss3, hipStream_t s);
void launch_segan_l1(const float* G, const float* lab, int n, const float* lambda, float* dG, bool accumulate, float* loss3, hipStream_t s);
void launch_rmsprop(float* w, const float* g, float* ms, const float* lr, float decay, float eps, size_t n, hipStream_t s);

struct SameGeom { int out, pl, pr; };
inline SameGeom same_pad(int L, int k, int stride = 2) {
  SameGeom g;
  g.out = (L + stride - 1) / stride;
  const int total = std::max((g.out - 1) * stride + k - L, 0);
  g.pl = total / 2; g.pr = total - g.pl;
  return g;
}

// A stride-2 conv layer of either net: filter [k, 1, Cin, Cout] as the GEMM operand [k*Cin][ld(Cout)] (downconv), or a decoder deconv with
// the filter [k, 1, Cout, Cin] = [k*Cout][ld(Cin)].  Lin/Lout are the lengths on the wide / narrow side.
struct SgLayer {
  int Cin = 0, Cout = 0, Lin = 0, Lout = 0, k = 0;       // Lin -> Lout for a downconv (Lout = ceil(Lin/2)); a deconv maps Lin -> Lout = 2x
  int tW = -1, tb = -1, ta = -1, tg = -1, tbeta = -1;    // ParamSet indices: filter, bias, PReLU alpha, VBN gamma / beta
  float* Wt[2] = {nullptr, nullptr};                     // parity-class operands of the transposed convolution (launch_prep_tconv)
  int ne[2] = {0, 0};
};

struct SeganModel {
  rsrgan_segan_cfg cfg{};
  int B = 0, Lx = 0, U = 0, Lj = 0, n = 0, Ld = 0;
  ParamSet G, D;
  std::vector<SgLayer> enc, dec, blk;                    // generator encoder / decoder, discriminator blocks
  int t_dense_w = -1, t_dense_b = -1, t_lc = -1, t_fcw = -1, t_fcb = -1;
  // generator activations: z_e[i] = enc_i output before the activation [B*Le(i+1)][C_i]; a_e[i] = after it (input of enc_{i+1});
  // xd[j] = decoder input [B*Ld_in(j)][Cin_j] (concat), zd[j] = dec_j output before the activation
  std::vector<float*> z_e, a_e, xd, zd;
  float *wave = nullptr, *Gy = nullptr, *xin = nullptr, *lab = nullptr, *zbuf = nullptr;
  // discriminator activations for up to 3 passes: h[i] = block i conv output (pre-VBN) [P*B*L(i+1)][C_i], y[i] = after VBN + leaky
  std::vector<float*> dh_, dy_;
  std::vector<float*> coef;                              // [3 passes][8][C_i] VBN coefficients per block
  float *joint = nullptr, *conv_out = nullptr, *logits = nullptr, *dlogits = nullptr, *djoint = nullptr;
  // scratch
  float *pad = nullptr, *t0 = nullptr, *t1 = nullptr, *gA = nullptr, *gB = nullptr, *gC = nullptr, *sums = nullptr, *red = nullptr, *gemm_ws = nullptr;
  size_t pad_floats = 0, t_floats = 0, g_floats = 0, red_floats = 0, gemm_ws_floats = 0;
  float *dyn = nullptr;                                  // device scalars: g_lr, d_lr, l1_lambda
  float *losses = nullptr;                               // [8]
  double scal[4] = {0, 0, 0, 0};
  bool d_grads_ready = false, g_grads_ready = false, g_fwd_valid = false;
  std::vector<void*> allocs;

  int init(const rsrgan_segan_cfg& c, uint64_t seed);
  void destroy();
  template <typename T> T* alloc(size_t n_);
  void refresh_weights(int net, hipStream_t s);
  // primitives (segan.cpp)
  void conv2_fwd(const float* X, int Bn, int L, int Cin, int k, const float* W, int ldw, const float* bias, int Cout, float* Z, hipStream_t s);
  void conv2_wgrad(const float* X, int Bn, int L, int Cin, int k, const float* dZ, int ldz, int Cout, float* dW, int ldw, hipStream_t s);
  void tconv2(const float* S, int Bn, int Ls, int Cs, int Lt, int k, float* const Wt[2], const int ne[2], int Ct, const float* bias, float* T, hipStream_t s);
  void g_forward(const float* x, const float* z, hipStream_t s);
  void d_forward(int P, hipStream_t s);
  void d_backward_pass(int P, int p0, bool wgrads, bool need_dx, hipStream_t s);
  void g_backward_pass(hipStream_t s);
  int d_run(const float* x, const float* labels, const float* z, const float* n_ref, const float* n_real, const float* n_fake, float* out_losses,
            bool want_grads, hipStream_t s);
  int g_run(const float* x, const float* labels, const float* z, const float* n_ref, const float* n_fake, float* out_losses, bool want_grads,
            hipStream_t s);
  int apply(int net, hipStream_t s);
};

}  // namespace rsr
