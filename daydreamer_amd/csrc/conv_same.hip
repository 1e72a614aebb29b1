// Stride-1 SAME convolution (odd k), its data gradient and its filter gradient as implicit
// GEMMs on the shared contraction loop, plus the 2x2 pooling / repetition of the residual
// encoder and decoder (reference nets.py:330-391: ImageEncoderResnet / ImageDecoderResnet;
// Conv2D with stride 1, pad 'same' nets.py:497-499, 541-547; tf.nn.avg_pool nets.py:343;
// tf.repeat nets.py:378).
//
// Rows of the contraction are the pixels (n, y, x) of the h x w image, k = (ky, kx*C + c) with
// source pixel (y + ky - p, x + kx - p), p = k/2; a source outside the image contributes an
// exact zero (TF 'SAME' for stride 1 and odd k pads p on every side).  With C a multiple of
// four a staged float4 lies inside one tap, so the bounds test is one select per 16 bytes.
#include "gemm_core.h"

namespace {

template <typename T, bool F>
struct ConvSameA {
  const T* x; int npix, h, w, C, kwc, pad; float scale;
  FastDiv d_hw, d_w, d_kwc, d_c;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {  // C % 4 == 0, 16-byte aligned: clamp the address, zero by select
      const int rr = min(r, npix - 1), kk = FULL ? k : min(k, kend - 4);
      int n, rem, y, xx, ky, o;
      d_hw.divmod(rr, n, rem);
      d_w.divmod(rem, y, xx);
      d_kwc.divmod(kk, ky, o);
      const int kx = d_c.div(o), c = o - kx * C;
      int sy = y + ky - pad, sx = xx + kx - pad;
      const bool ok = (FULL || k < kend) && sy >= 0 && sy < h && sx >= 0 && sx < w;
      sy = min(max(sy, 0), h - 1); sx = min(max(sx, 0), w - 1);
      load4_elems(x + (((long)n * h + sy) * w + sx) * C + c, scale, v);
      if (!ok) v[0] = v[1] = v[2] = v[3] = 0.f;
      return;
    }
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (r >= npix) return;
    int n, rem, y, xx;
    d_hw.divmod(r, n, rem);
    d_w.divmod(rem, y, xx);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = k + e;
      if (kk >= kend) continue;
      int ky, o;
      d_kwc.divmod(kk, ky, o);
      const int kx = d_c.div(o), c = o - kx * C;
      const int sy = y + ky - pad, sx = xx + kx - pad;
      if (sy < 0 || sy >= h || sx < 0 || sx >= w) continue;
      v[e] = cvt(x[(((long)n * h + sy) * w + sx) * C + c], scale);
    }
  }
};

// filter gradient: rows r = (ky, kx*C + c), k = pixel (n, y, x)
template <typename T, bool F>
struct ConvSameWgradA {
  const T* x; int h, w, C, kwc, pad, R; float scale;
  FastDiv d_hw, d_w, d_kwc, d_c;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {  // C % 4 == 0 (so R % 4 == 0)
      const int rr = min(r, R - 4), kk = FULL ? k : min(k, kend - 1);
      int n, rem, y, xx, ky, o;
      d_hw.divmod(kk, n, rem);
      d_w.divmod(rem, y, xx);
      d_kwc.divmod(rr, ky, o);
      const int kx = d_c.div(o), c = o - kx * C;
      int sy = y + ky - pad, sx = xx + kx - pad;
      const bool ok = (FULL || k < kend) && sy >= 0 && sy < h && sx >= 0 && sx < w;
      sy = min(max(sy, 0), h - 1); sx = min(max(sx, 0), w - 1);
      load4_elems(x + (((long)n * h + sy) * w + sx) * C + c, scale, v);
      if (!ok) v[0] = v[1] = v[2] = v[3] = 0.f;
      return;
    }
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (k >= kend) return;
    int n, rem, y, xx;
    d_hw.divmod(k, n, rem);
    d_w.divmod(rem, y, xx);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rr = r + e;
      if (rr >= R) continue;
      int ky, o;
      d_kwc.divmod(rr, ky, o);
      const int kx = d_c.div(o), c = o - kx * C;
      const int sy = y + ky - pad, sx = xx + kx - pad;
      if (sy < 0 || sy >= h || sx < 0 || sx >= w) continue;
      v[e] = cvt(x[(((long)n * h + sy) * w + sx) * C + c], scale);
    }
  }
};

// wf[ky][kx][co][ci] = w[k-1-ky][k-1-kx][ci][co]: the data gradient of a SAME convolution is the
// SAME convolution of dy with the filter rotated by 180 degrees and its channel axes swapped
__global__ void __launch_bounds__(256)
k_flip_filter(const float* __restrict__ w, float* __restrict__ wf, int k, int Cin, int Cout) {
  const long total = (long)k * k * Cin * Cout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    long t = i / Cin;
    const int co = (int)(t % Cout);
    t /= Cout;
    const int kx = (int)(t % k), ky = (int)(t / k);
    wf[i] = w[(((long)(k - 1 - ky) * k + (k - 1 - kx)) * Cin + ci) * Cout + co];
  }
}

// y[n][i][j][c] = scale * sum of the 2x2 block of x (avg_pool: scale 0.25; gradient of the
// 2x2 repetition: scale 1).  V floats per thread along the channel axis.
template <int V>
__global__ void __launch_bounds__(256)
k_pool2(const float* __restrict__ x, float* __restrict__ y, long total, int ho, int wo, int C, float scale) {
  const int cv = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % cv) * V;
    long t = i / cv;
    const int j = (int)(t % wo);
    t /= wo;
    const int ii = (int)(t % ho);
    const long n = t / ho;
    const float* p = x + (((n * (2 * ho) + 2 * ii) * (2 * wo)) + 2 * j) * C + c;
    const long rowp = (long)2 * wo * C;
    float o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = 0.f;
    if constexpr (V == 4) {
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + C);
      const float4 d = *reinterpret_cast<const float4*>(p + rowp), f = *reinterpret_cast<const float4*>(p + rowp + C);
      o[0] = ((a.x + b.x) + (d.x + f.x)) * scale; o[1] = ((a.y + b.y) + (d.y + f.y)) * scale;
      o[2] = ((a.z + b.z) + (d.z + f.z)) * scale; o[3] = ((a.w + b.w) + (d.w + f.w)) * scale;
      *reinterpret_cast<float4*>(y + ((n * ho + ii) * wo + j) * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      y[((n * ho + ii) * wo + j) * C + c] = ((p[0] + p[C]) + (p[rowp] + p[rowp + C])) * scale;
    }
  }
}

// y[n][2i+a][2j+b][c] = scale * x[n][i][j][c] (+ beta * y): tf.repeat by 2 on both image axes
// (scale 1) and the gradient of avg_pool (scale 0.25).  One thread per source element group.
template <int V>
__global__ void __launch_bounds__(256)
k_repeat2(const float* __restrict__ x, float* __restrict__ y, long total, int hi, int wi, int C, float scale, float beta) {
  const int cv = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % cv) * V;
    long t = i / cv;
    const int j = (int)(t % wi);
    t /= wi;
    const int ii = (int)(t % hi);
    const long n = t / hi;
    float* q = y + (((n * (2 * hi) + 2 * ii) * (2 * wi)) + 2 * j) * C + c;
    const long rowp = (long)2 * wi * C;
    const float* p = x + ((n * hi + ii) * wi + j) * C + c;
    if constexpr (V == 4) {
      float4 s = *reinterpret_cast<const float4*>(p);
      s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
      float* dst[4] = {q, q + C, q + rowp, q + rowp + C};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float4 r = s;
        if (beta != 0.f) {
          const float4 old = *reinterpret_cast<const float4*>(dst[e]);
          r.x += beta * old.x; r.y += beta * old.y; r.z += beta * old.z; r.w += beta * old.w;
        }
        *reinterpret_cast<float4*>(dst[e]) = r;
      }
    } else {
      const float s = p[0] * scale;
      float* dst[4] = {q, q + C, q + rowp, q + rowp + C};
#pragma unroll
      for (int e = 0; e < 4; ++e) *dst[e] = beta != 0.f ? s + beta * *dst[e] : s;
    }
  }
}

inline int ew_grid(long total) {
  const long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

template <typename T>
int conv_same_run(const T* x, const float* w, const float* bias, float* y, int n_img, int h, int wd,
                  int Cin, int Cout, int k, float in_scale, float alpha, float beta, float* wsp,
                  size_t ws_bytes, hipStream_t st, const char* nm) {
  const int M = n_img * h * wd, N = Cout, K = k * k * Cin, kwc = k * Cin;
  const int vb = aligned16(w) && (Cout % 4 == 0);
  const bool fa = (Cin % 4 == 0) && (sizeof(T) == 1 || aligned16(x));
  const FastDiv dhw(h * wd), dw(wd), dk(kwc), dc(Cin);
  if (fa && vb) {
    ConvSameA<T, true> al{x, M, h, wd, Cin, kwc, k / 2, in_scale, dhw, dw, dk, dc};
    return run_mat<true, false>(al, MatRC<true>{w, Cout, Cout, vb}, M, N, K, y, Cout, bias, alpha, beta, wsp, ws_bytes, st, nm);
  }
  ConvSameA<T, false> al{x, M, h, wd, Cin, kwc, k / 2, in_scale, dhw, dw, dk, dc};
  if (vb)
    return run_mat<true, false>(al, MatRC<true>{w, Cout, Cout, vb}, M, N, K, y, Cout, bias, alpha, beta, wsp, ws_bytes, st, nm);
  return run_mat<true, false>(al, MatRC<false>{w, Cout, Cout, vb}, M, N, K, y, Cout, bias, alpha, beta, wsp, ws_bytes, st, nm);
}

template <typename T>
int conv_same_wgrad_run(const T* x, const float* dy, float* dwt, int n_img, int h, int wd, int Cin,
                        int Cout, int k, float in_scale, float alpha, float beta, float* wsp,
                        size_t ws_bytes, hipStream_t st, const char* nm) {
  const int M = k * k * Cin, N = Cout, K = n_img * h * wd, kwc = k * Cin;
  const int vb = aligned16(dy) && (Cout % 4 == 0);
  const bool fa = (Cin % 4 == 0) && (sizeof(T) == 1 || aligned16(x));
  const FastDiv dhw(h * wd), dw(wd), dk(kwc), dc(Cin);
  if (fa && vb) {
    ConvSameWgradA<T, true> al{x, h, wd, Cin, kwc, k / 2, M, in_scale, dhw, dw, dk, dc};
    return run_mat<false, false>(al, MatRC<true>{dy, Cout, Cout, vb}, M, N, K, dwt, Cout, nullptr, alpha, beta, wsp, ws_bytes, st, nm);
  }
  ConvSameWgradA<T, false> al{x, h, wd, Cin, kwc, k / 2, M, in_scale, dhw, dw, dk, dc};
  if (vb)
    return run_mat<false, false>(al, MatRC<true>{dy, Cout, Cout, vb}, M, N, K, dwt, Cout, nullptr, alpha, beta, wsp, ws_bytes, st, nm);
  return run_mat<false, false>(al, MatRC<false>{dy, Cout, Cout, vb}, M, N, K, dwt, Cout, nullptr, alpha, beta, wsp, ws_bytes, st, nm);
}

}  // namespace

#define DD_SAME_GEOM(nm)                                                                        \
  DD_REQUIRE(k >= 1 && (k & 1) && n_img >= 0 && h > 0 && wd > 0 && Cin > 0 && Cout > 0,          \
             nm ": k must be odd, sizes positive");                                              \
  DD_REQUIRE((long)n_img * h * wd < (1L << 31) && (long)k * k * Cin < (1L << 31), nm ": index range")

extern "C" int dd_conv2d_same(const void* x, int x_is_u8, const float* w, const float* bias, float* y,
                              int n_img, int h, int wd, int Cin, int Cout, int k, float in_scale,
                              float alpha, float beta, float* wsp, size_t ws_bytes, void* stream) {
  DD_SAME_GEOM("dd_conv2d_same");
  hipStream_t st = (hipStream_t)stream;
  if (x_is_u8)
    return conv_same_run((const unsigned char*)x, w, bias, y, n_img, h, wd, Cin, Cout, k, in_scale,
                         alpha, beta, wsp, ws_bytes, st, "dd_conv2d_same");
  return conv_same_run((const float*)x, w, bias, y, n_img, h, wd, Cin, Cout, k, 1.f, alpha, beta,
                       wsp, ws_bytes, st, "dd_conv2d_same");
}

extern "C" int dd_conv2d_same_bwd_data(const float* dy, const float* w, float* dx, int n_img, int h,
                                       int wd, int Cin, int Cout, int k, float alpha, float beta,
                                       float* wsp, size_t ws_bytes, void* stream) {
  DD_SAME_GEOM("dd_conv2d_same_bwd_data");
  hipStream_t st = (hipStream_t)stream;
  // the rotated filter takes the head of the workspace (rounded to 256 B), split-K the rest
  const size_t wf_bytes = (((size_t)k * k * Cin * Cout * sizeof(float)) + 255) & ~(size_t)255;
  DD_REQUIRE(wsp && ws_bytes >= wf_bytes, "dd_conv2d_same_bwd_data: workspace too small for the rotated filter");
  const long total = (long)k * k * Cin * Cout;
  k_flip_filter<<<ew_grid(total), 256, 0, st>>>(w, wsp, k, Cin, Cout);
  DD_CHECK_LAUNCH("dd_conv2d_same_bwd_data(filter)");
  return conv_same_run(dy, wsp, nullptr, dx, n_img, h, wd, Cout, Cin, k, 1.f, alpha, beta,
                       (float*)((char*)wsp + wf_bytes), ws_bytes - wf_bytes, st, "dd_conv2d_same_bwd_data");
}

extern "C" int dd_conv2d_same_wgrad(const void* x, int x_is_u8, const float* dy, float* dwt, int n_img,
                                    int h, int wd, int Cin, int Cout, int k, float in_scale,
                                    float alpha, float beta, float* wsp, size_t ws_bytes, void* stream) {
  DD_SAME_GEOM("dd_conv2d_same_wgrad");
  hipStream_t st = (hipStream_t)stream;
  if (x_is_u8)
    return conv_same_wgrad_run((const unsigned char*)x, dy, dwt, n_img, h, wd, Cin, Cout, k, in_scale,
                               alpha, beta, wsp, ws_bytes, st, "dd_conv2d_same_wgrad");
  return conv_same_wgrad_run((const float*)x, dy, dwt, n_img, h, wd, Cin, Cout, k, 1.f, alpha, beta,
                             wsp, ws_bytes, st, "dd_conv2d_same_wgrad");
}

extern "C" int dd_pool2(const float* x, float* y, long n_img, int ho, int wo, int C, float scale,
                        void* stream) {
  DD_REQUIRE(ho > 0 && wo > 0 && C > 0 && n_img >= 0, "dd_pool2: sizes");
  if (n_img == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (C % 4 == 0 && aligned16(x) && aligned16(y)) {
    const long total = n_img * ho * wo * (C / 4);
    k_pool2<4><<<ew_grid(total), 256, 0, st>>>(x, y, total, ho, wo, C, scale);
  } else {
    const long total = n_img * ho * wo * C;
    k_pool2<1><<<ew_grid(total), 256, 0, st>>>(x, y, total, ho, wo, C, scale);
  }
  DD_CHECK_LAUNCH("dd_pool2");
  return 0;
}

extern "C" int dd_repeat2(const float* x, float* y, long n_img, int hi, int wi, int C, float scale,
                          float beta, void* stream) {
  DD_REQUIRE(hi > 0 && wi > 0 && C > 0 && n_img >= 0, "dd_repeat2: sizes");
  if (n_img == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (C % 4 == 0 && aligned16(x) && aligned16(y)) {
    const long total = n_img * hi * wi * (C / 4);
    k_repeat2<4><<<ew_grid(total), 256, 0, st>>>(x, y, total, hi, wi, C, scale, beta);
  } else {
    const long total = n_img * hi * wi * C;
    k_repeat2<1><<<ew_grid(total), 256, 0, st>>>(x, y, total, hi, wi, C, scale, beta);
  }
  DD_CHECK_LAUNCH("dd_repeat2");
  return 0;
}
