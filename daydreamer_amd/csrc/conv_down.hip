// Stride-2 VALID convolution as an implicit GEMM (encoder forward, decoder backward-data).
#include "gemm_core.h"

// conv_image.hip: the image-side layer (few `big` channels), rows split once into LDS planes
int dd_conv_image_down(const void* big, int big_is_u8, const float* w, const float* bias, float* small,
                       int n_img, int hb, int wb, int Cb, int hs, int ws_, int Cs, int k, float in_scale,
                       float* wsp, size_t ws_bytes, hipStream_t st, const float* ln_gamma, const float* ln_beta,
                       float* ln_out, float* ln_stats, const float* ln_z, int* n_partials);
int dd_ln_partials_reduce(const float* partials, int rows, int C, float* dgamma, float* dbeta, float* dbias,
                          float b, hipStream_t st);                      // rowops.hip

extern "C" int dd_conv2d_s2_down(const void* big, int big_is_u8, const float* w, const float* bias,
                                 float* small, int n_img, int hb, int wb, int Cb,
                                 int hs, int ws_, int Cs, int k, float in_scale,
                                 float* wsp, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DD_REQUIRE(2 * (hs - 1) + k <= hb && 2 * (ws_ - 1) + k <= wb, "dd_conv2d_s2_down: geometry");
  static const int image_kernel = getenv("DD_DOWN_IMAGE") ? atoi(getenv("DD_DOWN_IMAGE")) : 1;
  if (image_kernel && Cb <= 4 && gemm_mode() == 6) {
    const int rc = dd_conv_image_down(big, big_is_u8, w, bias, small, n_img, hb, wb, Cb, hs, ws_, Cs, k, in_scale, wsp, ws_bytes, st,
                                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc != 1) return rc;   // (1: geometry not covered)
  }
  const int M = n_img * hs * ws_, N = Cs, K = k * k * Cb;
  const int kwc = k * Cb;
  const int vb = aligned16(w) && (Cs % 4 == 0);
  if (big_is_u8) {
    if (kwc % 4 == 0 && vb) {  // a float4 chunk of the patch row = 4 contiguous bytes
      ConvDownA<unsigned char, true> al{(const unsigned char*)big, M, hs, ws_, hb, wb, Cb, kwc, in_scale, 1, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
      return run_mat<true, false>(al, MatRC<true>{w, Cs, Cs, vb}, M, N, K, small, Cs, bias, 1.f, 0.f, wsp, ws_bytes, st, "dd_conv2d_s2_down");
    }
    ConvDownA<unsigned char, false> al{(const unsigned char*)big, M, hs, ws_, hb, wb, Cb, kwc, in_scale, 0, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
    return run_mat<true, false>(al, MatRC<false>{w, Cs, Cs, vb}, M, N, K, small, Cs, bias, 1.f, 0.f, wsp, ws_bytes, st, "dd_conv2d_s2_down");
  }
  const int vec = aligned16(big) && (Cb % 4 == 0) && (kwc % 4 == 0);
  if (vec && vb) {
    ConvDownA<float, true> al{(const float*)big, M, hs, ws_, hb, wb, Cb, kwc, 1.f, vec, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
    return run_mat<true, false>(al, MatRC<true>{w, Cs, Cs, vb}, M, N, K, small, Cs, bias, 1.f, 0.f, wsp, ws_bytes, st, "dd_conv2d_s2_down");
  }
  // few-channel float image (decoder output layer backward): patch rows of kw*Cb floats at
  // 4-byte alignment -> the unaligned-row loader (mode 2) when a row holds at least four
  // floats; the filter operand keeps its own fast path
  const int mode = kwc >= 4 ? 2 : vec;
  ConvDownA<float, false> al{(const float*)big, M, hs, ws_, hb, wb, Cb, kwc, 1.f, mode, FastDiv(hs * ws_), FastDiv(ws_), FastDiv(kwc)};
  if (vb)
    return run_mat<true, false>(al, MatRC<true>{w, Cs, Cs, vb}, M, N, K, small, Cs, bias, 1.f, 0.f, wsp, ws_bytes, st, "dd_conv2d_s2_down");
  return run_mat<true, false>(al, MatRC<false>{w, Cs, Cs, vb}, M, N, K, small, Cs, bias, 1.f, 0.f, wsp, ws_bytes, st, "dd_conv2d_s2_down");
}


// dd_conv2d_s2_down of an image-side layer followed by its LayerNorm + ELU (the encoder's first
// layer, nets.py:291-305 + Norm :585-602) in one pass: writes the pre-norm rows `small`, the
// activations `out` and stats [pixels, 2] = mean / rstd (eps 1e-3) - what dd_conv2d_s2_down +
// dd_ln_act_fwd write, without the second pass over `small`.  Returns 1 with nothing launched
// when the geometry is not covered.
extern "C" int dd_conv2d_s2_down_ln(const void* big, int big_is_u8, const float* w, const float* bias,
                                    const float* gamma, const float* beta_ln, float* small, float* out,
                                    float* stats, int n_img, int hb, int wb, int Cb, int hs, int ws_, int Cs,
                                    int k, float in_scale, float* wsp, size_t ws_bytes, void* stream) {
  if (gemm_mode() != 6 || Cb > 4 || Cs != 64 || !gamma || !beta_ln || !out || !stats ||
      2 * (hs - 1) + k > hb || 2 * (ws_ - 1) + k > wb)
    return 1;
  return dd_conv_image_down(big, big_is_u8, w, bias, small, n_img, hb, wb, Cb, hs, ws_, Cs, k, in_scale, wsp, ws_bytes,
                            (hipStream_t)stream, gamma, beta_ln, out, stats, nullptr, nullptr);
}

// dd_conv2d_s2_down as the DATA GRADIENT of an image-side transposed convolution (the decoder's
// image layer, nets.py:308-327: big = d loss / d image, float) followed by the LayerNorm + ELU
// backward of the layer in front of it (Conv2D transpose + Norm + ELU): the gradient at that
// layer's output exists only in the accumulators; `dz` receives d loss / d (its pre-norm rows),
// and dgamma / dbeta / dbias what dd_ln_act_bwd would produce (accumulated when `accumulate`).
// Returns 1 with nothing computed when the geometry is not covered.
extern "C" int dd_conv2d_s2_down_lnbwd(const float* big, const float* w, const float* z, const float* stats,
                                       const float* gamma, const float* beta_ln, float* dz, float* dgamma,
                                       float* dbeta, float* dbias, int accumulate, int n_img, int hb, int wb,
                                       int Cb, int hs, int ws_, int Cs, int k, float* wsp, size_t ws_bytes,
                                       void* stream) {
  if (gemm_mode() != 6 || Cb > 4 || Cs != 64 || !z || !stats || !gamma || !beta_ln || !dz || !dgamma || !dbeta ||
      !dbias || 2 * (hs - 1) + k > hb || 2 * (ws_ - 1) + k > wb)
    return 1;
  hipStream_t st = (hipStream_t)stream;
  int n_partials = 0;
  const int rc = dd_conv_image_down(big, 0, w, nullptr, dz, n_img, hb, wb, Cb, hs, ws_, Cs, k, 1.f, wsp, ws_bytes, st,
                                    gamma, beta_ln, nullptr, const_cast<float*>(stats), z, &n_partials);
  if (rc != 0) return rc;
  const int OPK = (k * Cb + 7) / 8, KS = (k * OPK + 3) / 4;
  const size_t pbytes = ((size_t)(Cs / 16) * KS * 3 * 1024 + 255) & ~(size_t)255;
  return dd_ln_partials_reduce(reinterpret_cast<const float*>(reinterpret_cast<const char*>(wsp) + pbytes), n_partials,
                               Cs, dgamma, dbeta, dbias, accumulate ? 1.f : 0.f, st);
}
