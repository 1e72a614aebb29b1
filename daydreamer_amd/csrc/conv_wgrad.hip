// Filter gradient of the stride-2 convolutions.
#include "gemm_core.h"

int dd_conv_image_wgrad(const void* big, int big_is_u8, const float* small, int n_img, int hb, int wb,
                        int Cb, int hs, int ws_, int Cs, int k, float* wsp, size_t ws_bytes,
                        int* n_slabs, hipStream_t st, const float* ln_z, const float* ln_stats,
                        const float* ln_gamma, const float* ln_beta);   // conv_image.hip
int dd_ln_partials_reduce(const float* partials, int rows, int C, float* dgamma, float* dbeta, float* dbias,
                          float b, hipStream_t st);                      // rowops.hip

extern "C" int dd_conv2d_s2_wgrad(const void* big, int big_is_u8, const float* small, float* dw,
                                  int n_img, int hb, int wb, int Cb, int hs, int ws_, int Cs, int k,
                                  float in_scale, float beta, float* wsp, size_t ws_bytes,
                                  void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DD_REQUIRE(2 * (hs - 1) + k <= hb && 2 * (ws_ - 1) + k <= wb, "dd_conv2d_s2_wgrad: geometry");
  const int M = k * k * Cb, N = Cs, K = n_img * hs * ws_;
  if (gemm_mode() == 6) {   // image-side layers (3 channels): the dedicated kernel of conv_image.hip
    int n_slabs = 0;
    const int rc = dd_conv_image_wgrad(big, big_is_u8, small, n_img, hb, wb, Cb, hs, ws_, Cs, k, wsp, ws_bytes,
                                       &n_slabs, st, nullptr, nullptr, nullptr, nullptr);
    if (rc == 0) {
      launch_splitk_reduce(wsp, n_slabs, (long)M * N, N, dw, Cs, nullptr, big_is_u8 ? in_scale : 1.f, beta, st);
      DD_CHECK_LAUNCH("dd_conv2d_s2_wgrad(image reduce)");
      return 0;
    }
    if (rc != 1) return rc;
  }
  const int kwc = k * Cb;
  const int vb = aligned16(small) && (Cs % 4 == 0);
  if (big_is_u8) {
    if (kwc % 4 == 0 && vb) {
      ConvWgradA<unsigned char, true> al{(const unsigned char*)big, hs, ws_, hb, wb, Cb, kwc, M, in_scale, 1, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
      return run_mat<false, false>(al, MatRC<true>{small, Cs, Cs, vb}, M, N, K, dw, Cs, nullptr, 1.f, beta, wsp, ws_bytes, st, "dd_conv2d_s2_wgrad");
    }
    ConvWgradA<unsigned char, false> al{(const unsigned char*)big, hs, ws_, hb, wb, Cb, kwc, M, in_scale, 0, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
    return run_mat<false, false>(al, MatRC<false>{small, Cs, Cs, vb}, M, N, K, dw, Cs, nullptr, 1.f, beta, wsp, ws_bytes, st, "dd_conv2d_s2_wgrad");
  }
  const int vec = aligned16(big) && (Cb % 4 == 0) && (kwc % 4 == 0);
  if (vec && vb) {
    ConvWgradA<float, true> al{(const float*)big, hs, ws_, hb, wb, Cb, kwc, M, 1.f, vec, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
    return run_mat<false, false>(al, MatRC<true>{small, Cs, Cs, vb}, M, N, K, dw, Cs, nullptr, 1.f, beta, wsp, ws_bytes, st, "dd_conv2d_s2_wgrad");
  }
  // (the unaligned-row loader of dd_conv2d_s2_down was measured SLOWER here: 595 vs 485 us for the
  // 64x64x3 image layer at k 6 - the filter-gradient tile reads four consecutive patch ROWS per
  // chunk, whose selects cost more than the four scalar loads they replace)
  ConvWgradA<float, false> al{(const float*)big, hs, ws_, hb, wb, Cb, kwc, M, 1.f, vec, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
  return run_mat<false, false>(al, MatRC<false>{small, Cs, Cs, vb}, M, N, K, dw, Cs, nullptr, 1.f, beta, wsp, ws_bytes, st, "dd_conv2d_s2_wgrad");
}

// dd_conv2d_s2_wgrad for an image-side layer whose small side is a Conv2D + LayerNorm + ELU
// (the encoder's first layer, nets.py:291-305): `dout` is the gradient at the layer OUTPUT; the
// LayerNorm + ELU backward (dd_ln_act_bwd's arithmetic, activation recomputed from z) is applied
// while dout's rows are staged, so dz = d loss / d (conv output) is never written to or read from
// HBM.  Also produces what dd_ln_act_bwd would: dgamma, dbeta and dbias (the column sum of dz),
// accumulated when `accumulate`.  Returns 1 - nothing launched - when the geometry is not covered:
// the caller then runs dd_ln_act_bwd + dd_conv2d_s2_wgrad.
extern "C" int dd_conv2d_s2_wgrad_ln(const void* big, int big_is_u8, float in_scale, const float* dout,
                                     const float* z, const float* stats, const float* gamma,
                                     const float* beta_ln, float* dw, float beta, float* dgamma,
                                     float* dbeta, float* dbias, int accumulate, int n_img, int hb, int wb,
                                     int Cb, int hs, int ws_, int Cs, int k, float* wsp, size_t ws_bytes,
                                     void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (gemm_mode() != 6 || 2 * (hs - 1) + k > hb || 2 * (ws_ - 1) + k > wb || !z || !stats || !gamma || !beta_ln ||
      !dgamma || !dbeta || !dbias)
    return 1;
  int n_slabs = 0;
  const int rc = dd_conv_image_wgrad(big, big_is_u8, dout, n_img, hb, wb, Cb, hs, ws_, Cs, k, wsp, ws_bytes, &n_slabs,
                                     st, z, stats, gamma, beta_ln);
  if (rc != 0) return rc;
  const int M = k * k * Cb, N = Cs;
  const float* partials = wsp + (size_t)n_slabs * M * N;
  // (the parameter-gradient rows first: the filter reduce below overwrites nothing of them, but
  // both read the workspace the next launch on this context may reuse)
  const int r2 = dd_ln_partials_reduce(partials, n_slabs, Cs, dgamma, dbeta, dbias, accumulate ? 1.f : 0.f, st);
  if (r2) return r2;
  launch_splitk_reduce(wsp, n_slabs, (long)M * N, N, dw, Cs, nullptr, big_is_u8 ? in_scale : 1.f, beta, st);
  DD_CHECK_LAUNCH("dd_conv2d_s2_wgrad_ln(reduce)");
  return 0;
}
