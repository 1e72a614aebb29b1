// Transposed stride-2 convolution (decoder forward, encoder backward-data).
#include "gemm_core.h"

namespace {

// col2im for the GEMM + col2im form of the transposed conv (few output channels):
// big[n,by,bx,cb] = bias[cb] + sum_{ky,kx : (by-ky),(bx-kx) even, in range} cols[(n,sy,sx),(ky,kx,cb)]
__global__ void k_col2im_s2(const float* __restrict__ cols, const float* __restrict__ bias,
                            float* __restrict__ big, long n_img, int hs, int ws_, int hb, int wb,
                            int Cb, int k) {
  const long total = n_img * hb * wb * Cb;
  const int kkc = k * k * Cb;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    int cb = (int)(id % Cb); long r = id / Cb;
    int bx = (int)(r % wb); r /= wb;
    int by = (int)(r % hb); long n = r / hb;
    float acc = bias ? bias[cb] : 0.f;
    for (int ky = by & 1; ky < k; ky += 2) {
      int sy = (by - ky) >> 1;
      if (by < ky || sy >= hs) continue;
      for (int kx = bx & 1; kx < k; kx += 2) {
        int sx = (bx - kx) >> 1;
        if (bx < kx || sx >= ws_) continue;
        acc += cols[((n * hs + sy) * ws_ + sx) * kkc + (ky * k + kx) * Cb + cb];
      }
    }
    big[id] = acc;
  }
}

}  // namespace

// conv_image.hip: the image-side layer (few output channels) as one parity-form contraction
int dd_convT_image_fwd(const float* small, const float* w, const float* bias, float* big,
                       int n_img, int hs, int ws_, int Cs, int hb, int wb, int Cb, int k,
                       float* wsp, size_t ws_bytes, hipStream_t st);

extern "C" int dd_conv2d_s2_up(const float* small, const float* w, const float* bias, float* big,
                               int n_img, int hs, int ws_, int Cs, int hb, int wb, int Cb, int k,
                               float* wsp, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DD_REQUIRE(2 * (hs - 1) + k <= hb && 2 * (ws_ - 1) + k <= wb, "dd_conv2d_s2_up: geometry");
  // Banded implicit form (gemm_core.h, ConvUpAC): class pixels grouped by their valid taps, no
  // out-of-range tap is multiplied.  Even k: all four parities in one contraction (N = 4*Cb);
  // odd k: one contraction per parity (the parities have different tap counts).
  static const int classed = getenv("DD_UP_CLASSED") ? atoi(getenv("DD_UP_CLASSED")) : 1;
  const bool band_ok = classed && gemm_mode() != 0 && aligned16(small) && aligned16(w) && Cs % 16 == 0 &&
                       hb <= 128 && wb <= 128 && (k % 2 == 0 ? 4 * Cb >= 64 : Cb >= 64);
  if (band_ok) {
    struct Band { int p0, np, t0, nv; };
    auto bands = [](int npix, int nsrc, int nk, Band* out) {   // consecutive pixels with the same valid taps
      int nb = 0;
      for (int j = 0; j < npix; ++j) {
        const int t0 = j - nsrc + 1 > 0 ? j - nsrc + 1 : 0, t1 = j < nk - 1 ? j : nk - 1;
        const int nv = t1 >= t0 ? t1 - t0 + 1 : 0, tt = nv ? t0 : 0;
        if (nb && out[nb - 1].t0 == tt && out[nb - 1].nv == nv) ++out[nb - 1].np;
        else out[nb++] = Band{j, 1, tt, nv};
      }
      return nb;
    };
    const int four = k % 2 == 0;
    bool fits = true;
    for (int par = 0; par < (four ? 1 : 4) && fits; ++par) {
      const int py = four ? 0 : par >> 1, px = four ? 0 : par & 1;
      Band by[64], bx[64];
      fits = bands((hb - py + 1) / 2, hs, (k - py + 1) / 2, by) * bands((wb - px + 1) / 2, ws_, (k - px + 1) / 2, bx) <= UP_MAXCLS;
    }
    if (fits) {
      for (int par = 0; par < (four ? 1 : 4); ++par) {
        const int py = four ? 0 : par >> 1, px = four ? 0 : par & 1;
        const int nj = (hb - py + 1) / 2, ni = (wb - px + 1) / 2;       // class pixels
        const int nky = (k - py + 1) / 2, nkx = (k - px + 1) / 2;       // taps per axis
        if (nj <= 0 || ni <= 0) continue;
        Band by[64], bx[64];
        const int nby = bands(nj, hs, nky, by), nbx = bands(ni, ws_, nkx, bx);
        ConvUpAC al{small, n_img, hs, ws_, Cs, nby * nbx, FastDiv(Cs), {}};
        int tiles = 0, c = 0;
        for (int a = 0; a < nby; ++a)
          for (int b = 0; b < nbx; ++b, ++c) {
            UpCls& u = al.cls[c];
            u.j0 = (unsigned short)by[a].p0; u.njc = (unsigned short)by[a].np;
            u.i0 = (unsigned short)bx[b].p0; u.nic = (unsigned short)bx[b].np;
            u.ty0 = (unsigned char)by[a].t0; u.nvy = (unsigned char)by[a].nv;
            u.tx0 = (unsigned char)bx[b].t0; u.nvx = (unsigned char)bx[b].nv;
            if (!u.nvy || !u.nvx) u.nvy = u.nvx = 0;   // no valid tap: bias only
            u.tile0 = tiles;
            u.d_ji = FastDiv(by[a].np * bx[b].np); u.d_i = FastDiv(bx[b].np);
            tiles += dd_ceil_div(n_img * by[a].np * bx[b].np, 128);
          }
        const int N = four ? 4 * Cb : Cb, K = nky * nkx * Cs;
        EpiConvUp4C ep{big, bias, hb, wb, Cb, four, par, FastDiv(Cb)};
        ConvUpB4C bl{w, Cb, Cs, k, N, four, par, FastDiv(Cs), FastDiv(Cb)};
        const int kps = ((K + BKBIG - 1) / BKBIG) * BKBIG + BKBIG;
        if (N > 64)
          launch_tile<128, 128, true, true>(dim3(tiles * dd_ceil_div(N, 128), 1, 1), st, al, bl, ep, K, kps, tiles);
        else
          launch_tile<128, 64, true, true>(dim3(tiles, 1, 1), st, al, bl, ep, K, kps, tiles);
        DD_CHECK_LAUNCH("dd_conv2d_s2_up");
      }
      return 0;
    }
  }
  static const int image_kernel = getenv("DD_UP_IMAGE") ? atoi(getenv("DD_UP_IMAGE")) : 1;
  if (image_kernel && Cb <= 8 && gemm_mode() == 6) {
    const int rc = dd_convT_image_fwd(small, w, bias, big, n_img, hs, ws_, Cs, hb, wb, Cb, k, wsp, ws_bytes, st);
    if (rc != 1) return rc;   // (1: geometry not covered)
  }
  const int kkc = k * k * Cb;
  const size_t per_img = (size_t)hs * ws_ * kkc * sizeof(float);
  // GEMM + col2im instead of the implicit parity form when (a) there are few output
  // channels (image layer: an MFMA tile would be >90% padding in N), or (b) the
  // column buffer is small enough (<= 1 GiB) that its HBM round trip costs less than
  // the parity form's out-of-range taps (13-55% of its MFMA work at these sizes).
  const size_t cols_bytes = per_img * (size_t)n_img;
  // (c) with an even k the one-launch parity form below is the faster implicit path, so
  // the column buffer only pays up to 512 MiB (measured: 6x6x256 -> 14x14x128 k4, 737 MB of
  // columns: 986 us here vs 864 us implicit; 2x2x512 -> 6x6x256 k4, 118 MB: 351 vs 640 us).
  const bool uni_ok = aligned16(small) && aligned16(w) && (Cs % 4 == 0) && k % 2 == 0 && 4 * Cb >= 64;
  const size_t cols_max = uni_ok ? ((size_t)512 << 20) : ((size_t)1 << 30);
  if (wsp && ws_bytes >= 2 * per_img &&
      (Cb <= 8 || (cols_bytes <= cols_max && cols_bytes <= ws_bytes / 2))) {
    // cols = small[npix,Cs] @ W^T[Cs, k*k*Cb] (dense GEMM), then a gather.
    const int chunk = (int)((ws_bytes / 2) / per_img);  // second half: split-K scratch
    float* cols = wsp;
    float* ws2 = wsp + (ws_bytes / 2) / sizeof(float);
    for (int n0 = 0; n0 < n_img; n0 += chunk) {
      const int nn = (n_img - n0 < chunk) ? (n_img - n0) : chunk;
      const int M = nn * hs * ws_;
      const float* a = small + (size_t)n0 * hs * ws_ * Cs;
      const int vc = aligned16(a) && aligned16(w) && (Cs % 4 == 0);
      int rc = vc ? run_mat<true, true>(MatKC<true>{a, Cs, M, 1}, MatKC<true>{w, Cs, kkc, 1}, M, kkc,
                                        Cs, cols, kkc, nullptr, 1.f, 0.f, ws2, ws_bytes / 2, st,
                                        "dd_conv2d_s2_up(cols)")
                  : run_mat<true, true>(MatKC<false>{a, Cs, M, 0}, MatKC<false>{w, Cs, kkc, 0}, M,
                                        kkc, Cs, cols, kkc, nullptr, 1.f, 0.f, ws2, ws_bytes / 2, st,
                                        "dd_conv2d_s2_up(cols)");
      if (rc) return rc;
      const long total = (long)nn * hb * wb * Cb;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 8192) blocks = 8192;
      k_col2im_s2<<<blocks, 256, 0, st>>>(cols, bias, big + (size_t)n0 * hb * wb * Cb, nn, hs, ws_, hb, wb, Cb, k);
      DD_CHECK_LAUNCH("dd_conv2d_s2_up(col2im)");
    }
    return 0;
  }
  const int vec = aligned16(small) && aligned16(w) && (Cs % 4 == 0);
  if (uni_ok) {  // all parities in one contraction
    const int nj = (hb + 1) / 2, ni = (wb + 1) / 2, nk = k / 2;
    const int M = n_img * nj * ni, N = 4 * Cb, K = nk * nk * Cs;
    EpiConvUp4 ep{big, bias, M, nj, ni, hb, wb, Cb, FastDiv(nj * ni), FastDiv(ni), FastDiv(Cb)};
    ConvUpA<true> al{small, M, nj, ni, hs, ws_, Cs, nk, vec, FastDiv(nj * ni), FastDiv(ni), FastDiv(Cs), FastDiv(nk)};
    ConvUpB4 bl{w, Cb, Cs, k, nk, N, FastDiv(Cs), FastDiv(nk), FastDiv(Cb)};
    const int kps = ((K + BKBIG - 1) / BKBIG) * BKBIG + BKBIG;
    const int tm = dd_ceil_div(M, 128);
    if (N > 64)
      launch_tile<128, 128, true, true>(dim3(tm * dd_ceil_div(N, 128), 1, 1), st, al, bl, ep, K, kps, tm);
    else
      launch_tile<128, 64, true, true>(dim3(tm, 1, 1), st, al, bl, ep, K, kps, tm);
    DD_CHECK_LAUNCH("dd_conv2d_s2_up");
    return 0;
  }
  auto launch = [&](auto fast) -> int {
    constexpr bool FF = decltype(fast)::value;
    using AT = ConvUpA<FF>;
    using BT = ConvUpB<FF>;
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        const int nj = (hb - py + 1) / 2, ni = (wb - px + 1) / 2;  // pixels of this parity
        const int nky = (k - py + 1) / 2, nkx = (k - px + 1) / 2;  // taps of this parity
        if (nj <= 0 || ni <= 0) continue;
        const int M = n_img * nj * ni, N = Cb;
        const int K = (nky > 0 && nkx > 0) ? nky * nkx * Cs : 0;  // K = 0: bias only
        EpiConvUp ep{big, bias, M, nj, ni, hb, wb, Cb, py, px, FastDiv(nj * ni), FastDiv(ni)};
        AT al{small, M, nj, ni, hs, ws_, Cs, nkx > 0 ? nkx : 1, vec, FastDiv(nj * ni), FastDiv(ni), FastDiv(Cs), FastDiv(nkx > 0 ? nkx : 1)};
        BT bl{w, Cb, Cs, k, nkx > 0 ? nkx : 1, py, px, vec, FastDiv(Cs), FastDiv(nkx > 0 ? nkx : 1)};
        const int kps = ((K + BKBIG - 1) / BKBIG) * BKBIG + BKBIG;
        if (M > 64 && N > 64) {
          int tm = dd_ceil_div(M, 128), tn = dd_ceil_div(N, 128);
          launch_tile<128, 128, true, true>(dim3(tm * tn, 1, 1), st, al, bl, ep, K, kps, tm);
        } else if (M > 64) {
          int tm = dd_ceil_div(M, 128), tn = dd_ceil_div(N, 64);
          launch_tile<128, 64, true, true>(dim3(tm * tn, 1, 1), st, al, bl, ep, K, kps, tm);
        } else {
          int tm = dd_ceil_div(M, 64), tn = dd_ceil_div(N, 64);
          launch_tile<64, 64, true, true>(dim3(tm * tn, 1, 1), st, al, bl, ep, K, kps, tm);
        }
        DD_CHECK_LAUNCH("dd_conv2d_s2_up");
      }
    return 0;
  };
  return vec ? launch(std::true_type()) : launch(std::false_type());
}

