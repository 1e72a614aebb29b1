// Shared core of the contraction kernels: tile loaders, epilogues, the native fp32 and the
// split-bf16 MFMA main loops, split-K and the tile dispatcher.  Included by gemm.hip,
// conv_down.hip, conv_up.hip and conv_wgrad.hip (one translation unit per C-ABI entry point,
// so that the template instantiations compile in parallel).
#pragma once
#include "dd_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "../../include/daydreamer_hip.h"


// arithmetic mode of the main loop (dd_gemm_set_mode), defined in gemm.hip
extern int g_gemm_mode;
// selector of the role-separated loop {on, kmin, kmin of tile-context launches} (dd_gemm_set_ws)
extern int g_ws_select[3];

namespace {

// k-tile of the 128x128 tile (16 measured faster than 32: 109 vs 92 TF at 4096^3)
#ifndef BKBIG
#define BKBIG 16
#endif
// two accumulator chains in the 64x64 tile's loop (k_mfma_gemm_s3 A2)
#ifndef DD_A2_64
#define DD_A2_64 false
#endif

// ---------------------------------------------------------------------------
// Operand loaders.  load4(r, k, kend, v) returns four consecutive elements
// along the operand's contiguous axis: KC = (r, k..k+3), RC = (r..r+3, k).
// ---------------------------------------------------------------------------

// F = true is the branch-free fast path (16-byte aligned, ld, R and K multiples of
// 4): out-of-range rows are clamped (they only feed masked outputs), the K tail is zeroed
// by a select.  Ablation: bounds-check branches in the loaders cost ~15% of GEMM time.
template <bool F>
struct MatKC {  // op(X)[r][k] = p[r*ld + k]
  const float* p; long ld; int R; int vec;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {
      const int rr = min(r, R - 1), kk = FULL ? k : min(k, kend - 4);
      float4 t = *reinterpret_cast<const float4*>(p + (long)rr * ld + kk);
      const bool ok = FULL || k < kend;
      v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f;
      return;
    }
    if (r < R) {
      const float* q = p + (long)r * ld + k;
      if (vec && k + 3 < kend) {
        float4 t = *reinterpret_cast<const float4*>(q);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (k + j < kend) ? q[j] : 0.f;
      }
    } else {
      v[0] = v[1] = v[2] = v[3] = 0.f;
    }
  }
};

template <bool F>
struct MatRC {  // op(X)[r][k] = p[k*ld + r]
  const float* p; long ld; int R; int vec;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {
      const int rr = min(r, R - 4), kk = FULL ? k : min(k, kend - 1);
      float4 t = *reinterpret_cast<const float4*>(p + (long)kk * ld + rr);
      const bool ok = FULL || k < kend;
      v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f;
      return;
    }
    if (k < kend) {
      const float* q = p + (long)k * ld + r;
      if (vec && r + 3 < R) {
        float4 t = *reinterpret_cast<const float4*>(q);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (r + j < R) ? q[j] : 0.f;
      }
    } else {
      v[0] = v[1] = v[2] = v[3] = 0.f;
    }
  }
};

// Division by a launch-constant via multiply-high (valid for dividends < 2^31): the
// gather loaders decode (image, y, x) / (tap, channel) from linear indices every load,
// and a 32-bit integer division costs ~25 VALU instructions on gfx950.
struct FastDiv {
  unsigned mul, shr, d;
  __host__ __device__ FastDiv() : mul(0), shr(0), d(1) {}
  __host__ explicit FastDiv(int dd) {
    d = (unsigned)(dd < 1 ? 1 : dd);
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    shr = l;
  }
  __device__ __forceinline__ int div(int n) const {
    return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr);
  }
  __device__ __forceinline__ void divmod(int n, int& q, int& r) const {
    q = div(n);
    r = n - q * (int)d;
  }
};

// four consecutive elements of a gather operand: one 16-byte load for float, one
// (possibly 2-byte aligned) dword load + byte unpack for uint8 images
typedef unsigned int __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ void load4_elems(const float* p, float, float v[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4_elems(const unsigned char* p, float s, float v[4]) {
  const unsigned w = *reinterpret_cast<const u32_unaligned*>(p);
  v[0] = (float)(w & 255u) * s; v[1] = (float)((w >> 8) & 255u) * s;
  v[2] = (float)((w >> 16) & 255u) * s; v[3] = (float)(w >> 24) * s;
}

// 16 bytes at 4-byte alignment (a patch row of a 3-channel float image starts on any float)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void load4_unaligned(const float* p, float v[4]) {
  const f32x4_u t = *reinterpret_cast<const f32x4_u*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4_unaligned(const unsigned char*, float v[4]) { v[0] = v[1] = v[2] = v[3] = 0.f; }

__device__ __forceinline__ float cvt(float x, float) { return x; }
__device__ __forceinline__ float cvt(unsigned char x, float s) { return (float)x * s; }

// conv "down": rows = output pixels (n,sy,sx), k = (ky, kx*Cb + cb).
template <typename T, bool F>
struct ConvDownA {
  const T* big; int npix, hs, ws, hb, wb, Cb, kwc; float scale; int vec;
  FastDiv d_hw, d_w, d_kwc;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {  // branch-free: clamp the pixel, zero the K tail
      const int rr = min(r, npix - 1), kk = FULL ? k : min(k, kend - 4);
      int n, rem, sy, sx, ky, o;
      d_hw.divmod(rr, n, rem);
      d_w.divmod(rem, sy, sx);
      d_kwc.divmod(kk, ky, o);
      load4_elems(big + (((long)n * hb + 2 * sy + ky) * wb + 2 * sx) * Cb + o, scale, v);
      if (!(FULL || k < kend)) v[0] = v[1] = v[2] = v[3] = 0.f;
      return;
    }
    if (vec == 2) {
      // float image whose patch row (kw*Cb floats) is not a multiple of four (3 channels, k 6):
      // the four k of a chunk are contiguous inside one patch row or straddle two rows - two
      // 16-byte loads at 4-byte alignment (this row at o, the next row at o - kwc) and a
      // per-element select, instead of four scalar loads with their own index arithmetic
      const int rr = min(r, npix - 1), kk = min(k, kend - 1);
      int n, rem, sy, sx, ky, o;
      d_hw.divmod(rr, n, rem);
      d_w.divmod(rem, sy, sx);
      d_kwc.divmod(kk, ky, o);
      const T* row0 = big + (((long)n * hb + 2 * sy + ky) * wb + 2 * sx) * Cb;
      const int ky1 = min(ky + 1, (kend - 1) / kwc);       // stays inside the kernel window
      const T* row1 = big + (((long)n * hb + 2 * sy + ky1) * wb + 2 * sx) * Cb;
      // both loads stay inside their patch rows (no read past the tensor's last pixel)
      float a[4], b[4];
      const int oa = min(o, kwc - 4);
      load4_unaligned(row0 + oa, a);
      load4_unaligned(row1, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ia = o + j - oa, ib = o + j - kwc;
        const float xa = ia == 0 ? a[0] : (ia == 1 ? a[1] : (ia == 2 ? a[2] : a[3]));
        const float xb = ib <= 0 ? b[0] : (ib == 1 ? b[1] : (ib == 2 ? b[2] : b[3]));
        const float x = (o + j < kwc) ? xa : xb;
        v[j] = (r < npix && k + j < kend) ? x * scale : 0.f;
      }
      return;
    }
    if (r >= npix) { v[0] = v[1] = v[2] = v[3] = 0.f; return; }
    int n = r / (hs * ws); int rem = r - n * hs * ws;
    int sy = rem / ws; int sx = rem - sy * ws;
    long base = (((long)n * hb + 2 * sy) * wb + 2 * sx) * Cb;
    long rowpitch = (long)wb * Cb;
    if (vec && k + 3 < kend) {
      int ky = k / kwc; int o = k - ky * kwc;
      const float* q = reinterpret_cast<const float*>(big) + base + ky * rowpitch + o;
      float4 t = *reinterpret_cast<const float4*>(q);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kk = k + j;
        if (kk < kend) {
          int ky = kk / kwc; int o = kk - ky * kwc;
          v[j] = cvt(big[base + ky * rowpitch + o], scale);
        } else {
          v[j] = 0.f;
        }
      }
    }
  }
};

// conv "up", one output-parity class: rows = class pixels (n,j,i), output
// pixel (2j+py, 2i+px); k = (tap=(m,mx), cs) with source pixel (j-m, i-mx).
template <bool F>
struct ConvUpA {
  const float* small; int npix, nj, ni, hs, ws, Cs, nkx; int vec;
  FastDiv d_ji, d_i, d_cs, d_nkx;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {  // branch-free: clamp pixel / tap source, zero by select
      const int rr = min(r, npix - 1), kk = FULL ? k : min(k, kend - 4);
      int n, rem, j, i, tap, c, m, mx;
      d_ji.divmod(rr, n, rem);
      d_i.divmod(rem, j, i);
      d_cs.divmod(kk, tap, c);
      d_nkx.divmod(tap, m, mx);
      int sy = j - m, sx = i - mx;
      const bool ok = (FULL || k < kend) && sy >= 0 && sy < hs && sx >= 0 && sx < ws;
      sy = min(max(sy, 0), hs - 1); sx = min(max(sx, 0), ws - 1);
      float4 t = *reinterpret_cast<const float4*>(small + (((long)n * hs + sy) * ws + sx) * Cs + c);
      v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f;
      return;
    }
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (r >= npix || k >= kend) return;
    int n = r / (nj * ni); int rem = r - n * nj * ni;
    int j = rem / ni; int i = rem - j * ni;
    if (vec) {
      int tap = k / Cs; int c = k - tap * Cs;
      int m = tap / nkx; int mx = tap - m * nkx;
      int sy = j - m, sx = i - mx;
      if (sy < 0 || sy >= hs || sx < 0 || sx >= ws) return;
      const float* q = small + (((long)n * hs + sy) * ws + sx) * Cs + c;
      float4 t = *reinterpret_cast<const float4*>(q);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int kk = k + e;
        if (kk >= kend) continue;
        int tap = kk / Cs; int c = kk - tap * Cs;
        int m = tap / nkx; int mx = tap - m * nkx;
        int sy = j - m, sx = i - mx;
        if (sy < 0 || sy >= hs || sx < 0 || sx >= ws) continue;
        v[e] = small[(((long)n * hs + sy) * ws + sx) * Cs + c];
      }
    }
  }
};

// conv "up" filter operand: B[k=(tap,cs)][n=cb] = W[ky][kx][cb][cs].
template <bool F>
struct ConvUpB {
  const float* w; int Cb, Cs, kw, nkx, py, px; int vec;
  FastDiv d_cs, d_nkx;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {
      const int rr = min(r, Cb - 1), kk = FULL ? k : min(k, kend - 4);
      int tap, c, m, mx;
      d_cs.divmod(kk, tap, c);
      d_nkx.divmod(tap, m, mx);
      int ky = py + 2 * m, kx = px + 2 * mx;
      float4 t = *reinterpret_cast<const float4*>(w + (((long)ky * kw + kx) * Cb + rr) * Cs + c);
      const bool ok = FULL || k < kend;
      v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f;
      return;
    }
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (r >= Cb || k >= kend) return;
    if (vec) {
      int tap = k / Cs; int c = k - tap * Cs;
      int m = tap / nkx; int mx = tap - m * nkx;
      int ky = py + 2 * m, kx = px + 2 * mx;
      const float* q = w + (((long)ky * kw + kx) * Cb + r) * Cs + c;
      float4 t = *reinterpret_cast<const float4*>(q);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int kk = k + e;
        if (kk >= kend) continue;
        int tap = kk / Cs; int c = kk - tap * Cs;
        int m = tap / nkx; int mx = tap - m * nkx;
        int ky = py + 2 * m, kx = px + 2 * mx;
        v[e] = w[(((long)ky * kw + kx) * Cb + r) * Cs + c];
      }
    }
  }
};

// conv "up", all four output parities in one contraction (even k: every parity has the
// same (k/2)^2 taps and, for class pixel (j,i), the same source pixels (j-m, i-mx); only the
// filter taps differ).  The parity becomes part of the column index, n = (py,px,cb):
// B[k=(m,mx,cs)][n] = W[py+2m][px+2mx][cb][cs].  One launch with N = 4*Cb instead of four
// with N = Cb: full 128-wide tiles for 64-channel layers and a quarter of the gather traffic.
struct ConvUpB4 {
  const float* w; int Cb, Cs, kw, nkx, R;
  FastDiv d_cs, d_nkx, d_cb;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    const int rr = min(r, R - 1), kk = FULL ? k : min(k, kend - 4);
    int tap, c, m, mx, q, cb;
    d_cs.divmod(kk, tap, c);
    d_nkx.divmod(tap, m, mx);
    d_cb.divmod(rr, q, cb);
    const int ky = (q >> 1) + 2 * m, kx = (q & 1) + 2 * mx;
    float4 t = *reinterpret_cast<const float4*>(w + (((long)ky * kw + kx) * Cb + cb) * Cs + c);
    const bool ok = FULL || k < kend;
    v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f;
  }
};

// filter gradient: rows r = (ky, kx*Cb + cb) (contiguous in runs of kw*Cb),
// k = small-side pixel (n,sy,sx).
template <typename T, bool F>
struct ConvWgradA {
  const T* big; int hs, ws, hb, wb, Cb, kwc, R; float scale; int vec;
  FastDiv d_hw, d_w, d_kwc;
  template <bool FULL = false>
  __device__ __forceinline__ void load4(int r, int k, int kend, float v[4]) const {
    if constexpr (F) {  // branch-free (R % 4 == 0): clamp, zero the K tail by select
      const int rr = min(r, R - 4), kk = FULL ? k : min(k, kend - 1);
      int n, rem, sy, sx, ky, o;
      d_hw.divmod(kk, n, rem);
      d_w.divmod(rem, sy, sx);
      d_kwc.divmod(rr, ky, o);
      load4_elems(big + (((long)n * hb + 2 * sy + ky) * wb + 2 * sx) * Cb + o, scale, v);
      if (!(FULL || k < kend)) v[0] = v[1] = v[2] = v[3] = 0.f;
      return;
    }
    if (vec == 2) {  // float image, patch row not a multiple of four floats: see ConvDownA
      const int kk = min(k, kend - 1), rr = min(r, R - 1);
      int n, rem, sy, sx, ky, o;
      d_hw.divmod(kk, n, rem);
      d_w.divmod(rem, sy, sx);
      d_kwc.divmod(rr, ky, o);
      const T* row0 = big + (((long)n * hb + 2 * sy + ky) * wb + 2 * sx) * Cb;
      const int ky1 = min(ky + 1, (R - 1) / kwc);
      const T* row1 = big + (((long)n * hb + 2 * sy + ky1) * wb + 2 * sx) * Cb;
      float a[4], b[4];
      const int oa = min(o, kwc - 4);
      load4_unaligned(row0 + oa, a);
      load4_unaligned(row1, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ia = o + j - oa, ib = o + j - kwc;
        const float xa = ia == 0 ? a[0] : (ia == 1 ? a[1] : (ia == 2 ? a[2] : a[3]));
        const float xb = ib <= 0 ? b[0] : (ib == 1 ? b[1] : (ib == 2 ? b[2] : b[3]));
        const float x = (o + j < kwc) ? xa : xb;
        v[j] = (k < kend && r + j < R) ? x * scale : 0.f;
      }
      return;
    }
    if (k >= kend) { v[0] = v[1] = v[2] = v[3] = 0.f; return; }
    int n = k / (hs * ws); int rem = k - n * hs * ws;
    int sy = rem / ws; int sx = rem - sy * ws;
    long base = (((long)n * hb + 2 * sy) * wb + 2 * sx) * Cb;
    long rowpitch = (long)wb * Cb;
    if (vec && r + 3 < R) {
      int ky = r / kwc; int o = r - ky * kwc;
      const float* q = reinterpret_cast<const float*>(big) + base + ky * rowpitch + o;
      float4 t = *reinterpret_cast<const float4*>(q);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int rr = r + j;
        if (rr < R) {
          int ky = rr / kwc; int o = rr - ky * kwc;
          v[j] = cvt(big[base + ky * rowpitch + o], scale);
        } else {
          v[j] = 0.f;
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Epilogues: operator()(m, n, acc).
// ---------------------------------------------------------------------------

struct EpiMat {
  float* C; long ldc; const float* bias; float alpha, beta; int M, N;
  float* slab;  // non-null: split-K partial, raw accumulators
  // beta != 0: the old C values of a 32x32 accumulator block are fetched by one batch of loads
  // before its stores (read_old / put): with the load inside the per-element store loop every
  // load waited for the previous store (possible alias), 64 serial round trips per thread -
  // a beta = 1 GEMM took 55 % longer than the same one with beta = 0 (482 vs 312 us at
  // 40000x1280x512)
  static constexpr bool READS_C = true;
  __device__ __forceinline__ bool wants_old() const { return beta != 0.f && !slab; }
  __device__ __forceinline__ float read_old(int m, int n) const {
    return (m < M && n < N) ? C[(long)m * ldc + n] : 0.f;
  }
  __device__ __forceinline__ void put(int m, int n, float v, float old) const {
    if (m >= M || n >= N) return;
    float r = alpha * v;
    if (bias) r += bias[n];
    r += beta * old;
    C[(long)m * ldc + n] = r;
  }
  __device__ __forceinline__ void operator()(int m, int n, float v) const {
    if (m >= M || n >= N) return;
    if (slab) {
      slab[((long)blockIdx.z * M + m) * N + n] = v;
      return;
    }
    float r = alpha * v;
    if (bias) r += bias[n];
    long i = (long)m * ldc + n;
    if (beta != 0.f) r += beta * C[i];
    C[i] = r;
  }
};

struct EpiConvUp {
  float* big; const float* bias; int npix, nj, ni, hb, wb, Cb, py, px;
  FastDiv d_ji, d_i;

  __device__ __forceinline__ void operator()(int m, int n, float v) const {
    if (m >= npix || n >= Cb) return;
    int img, rem, j, i;
    d_ji.divmod(m, img, rem);
    d_i.divmod(rem, j, i);
    long a = (((long)img * hb + 2 * j + py) * wb + 2 * i + px) * Cb + n;
    big[a] = bias ? v + bias[n] : v;
  }
};

struct EpiConvUp4 {  // column n = (py, px, cb)
  float* big; const float* bias; int npix, nj, ni, hb, wb, Cb;
  FastDiv d_ji, d_i, d_cb;
  __device__ __forceinline__ void operator()(int m, int n, float v) const {
    if (m >= npix || n >= 4 * Cb) return;
    int img, rem, j, i, q, cb;
    d_ji.divmod(m, img, rem);
    d_i.divmod(rem, j, i);
    d_cb.divmod(n, q, cb);
    const int y = 2 * j + (q >> 1), x = 2 * i + (q & 1);
    if (y >= hb || x >= wb) return;
    big[(((long)img * hb + y) * wb + x) * Cb + cb] = bias ? v + bias[cb] : v;
  }
};

// ---- conv "up", all parities, class pixels grouped by tap validity ----------------------
// In the form above every class pixel (j, i) contracts over all (k/2)^2 taps, and the taps
// whose source pixel (j-m, i-mx) lies outside the small image are loaded as zeros: 13-56% of
// the MFMA work at the decoder / encoder-backward sizes (edge pixels have 1..k/2-1 valid taps
// per axis).  The valid taps of a pixel form a rectangle [ty0, ty0+nvy) x [tx0, tx0+nvx) that
// only depends on which edge band j and i lie in, so the class pixels are grouped into the
// (<= 2*(k/2))^2 bands of equal rectangle: each band is its own contraction (rows = the
// band's pixels of all images, K = nvy*nvx*Cs, the band's taps only), all bands in one
// launch (row tile -> band by a table in the kernel arguments).  Skipped products are exact
// zeros, so the result is bit-identical to the unclassed form.
struct UpCls {
  unsigned short j0, njc, i0, nic;
  unsigned char ty0, nvy, tx0, nvx;
  int tile0;          // first row tile of the band
  FastDiv d_ji, d_i;  // band-local row -> (image, jj, ii)
};
constexpr int UP_MAXCLS = 36;
struct UpCtx {        // the band of this workgroup's row tile (wave-uniform)
  int j0, i0, ty0, tx0, nvx, rmul, keff, row0, rows;
  FastDiv d_ji, d_i;
};
struct NoCtx {};
template <class AL, class = void> struct has_tile_ctx : std::false_type {};
template <class AL> struct has_tile_ctx<AL, std::enable_if_t<AL::TILE_CTX>> : std::true_type {};
template <int BM, class AL>
__device__ __forceinline__ auto get_tile_ctx(const AL& al, int tmi) {
  if constexpr (has_tile_ctx<AL>::value) return al.template tile_ctx<BM>(tmi);
  else return NoCtx{};
}
// Tile-context loaders are affine: element (row, k0 + kofs) of the operand lies at
// ptr[row_base(row) + k_off(k0) + kofs] for every k0 that is a multiple of the k-tile - the
// row term is per thread and loop-invariant, the k term is wave-uniform (scalar unit), so a
// staged float4 costs one 64-bit add instead of the ~40 vector instructions of the
// divide-and-clamp gather (which, not the MFMAs, bounded the unclassed form).
// (slot s of the band's tap rectangle -> tap (m, mx); s < 16)
__device__ __forceinline__ void up_tap(const UpCtx& cx, int s, int& m, int& mx) {
  const int q = (s * cx.rmul) >> 16;   // s / nvx
  m = cx.ty0 + q;
  mx = cx.tx0 + s - q * cx.nvx;
}

struct ConvUpAC {
  static constexpr bool TILE_CTX = true;
  const float* small; int n_img, hs, ws, Cs, ncls;
  FastDiv d_cs;
  UpCls cls[UP_MAXCLS];
  template <int BM>
  __device__ __forceinline__ UpCtx tile_ctx(int tmi) const {
    int c = 0;
    for (int q = 1; q < ncls; ++q) c = tmi >= cls[q].tile0 ? q : c;
    const UpCls& u = cls[c];
    UpCtx cx;
    cx.j0 = u.j0; cx.i0 = u.i0; cx.ty0 = u.ty0; cx.tx0 = u.tx0; cx.nvx = u.nvx;
    cx.keff = (int)u.nvy * (int)u.nvx * Cs;
    cx.row0 = (tmi - u.tile0) * BM;
    cx.rows = n_img * (int)u.njc * (int)u.nic;
    cx.rmul = 65536 / (u.nvx ? u.nvx : 1) + 1;
    cx.d_ji = u.d_ji; cx.d_i = u.d_i;
    return cx;
  }
  __device__ __forceinline__ const float* ptr() const { return small; }
  __device__ __forceinline__ long row_base(const UpCtx& cx, int rl) const {
    const int rr = min(rl, cx.rows - 1);   // rows past the band: duplicates, dropped by the epilogue
    int n, rem, jj, ii;
    cx.d_ji.divmod(rr, n, rem);
    cx.d_i.divmod(rem, jj, ii);
    return (((long)n * hs + cx.j0 + jj) * ws + cx.i0 + ii) * Cs;
  }
  __device__ __forceinline__ int k_off(const UpCtx& cx, int k0) const {
    int s, c0, m, mx;
    d_cs.divmod(k0, s, c0);
    up_tap(cx, s, m, mx);
    return c0 - (m * ws + mx) * Cs;   // source pixel (j - m, i - mx): in range by construction
  }
};

// filter operand of the banded form.  four: column n = (py, px, cb), all parities in one
// contraction (even k); else one parity (py, px) per launch, column n = cb (odd k: the
// parities have different tap counts).
struct ConvUpB4C {
  const float* w; int Cb, Cs, kw, R, four, par;   // par = 2*py + px when !four
  FastDiv d_cs, d_cb;
  __device__ __forceinline__ const float* ptr() const { return w; }
  __device__ __forceinline__ long row_base(const UpCtx&, int r) const {
    const int rr = min(r, R - 1);
    int q, cb;
    if (four) d_cb.divmod(rr, q, cb);
    else { q = par; cb = rr; }
    return ((long)((q >> 1) * kw + (q & 1)) * Cb + cb) * Cs;
  }
  __device__ __forceinline__ int k_off(const UpCtx& cx, int k0) const {
    int s, c0, m, mx;
    d_cs.divmod(k0, s, c0);
    up_tap(cx, s, m, mx);
    return c0 + (2 * m * kw + 2 * mx) * Cb * Cs;
  }
};

struct EpiConvUp4C {  // rows band-local; column n = (py, px, cb) or cb (see ConvUpB4C)
  float* big; const float* bias; int hb, wb, Cb, four, par;
  FastDiv d_cb;
  __device__ __forceinline__ void putc(const UpCtx& cx, int rl, int n, float v) const {
    if (rl >= cx.rows || n >= (four ? 4 * Cb : Cb)) return;
    int img, rem, jj, ii, q, cb;
    cx.d_ji.divmod(rl, img, rem);
    cx.d_i.divmod(rem, jj, ii);
    if (four) d_cb.divmod(n, q, cb);
    else { q = par; cb = n; }
    const int y = 2 * (cx.j0 + jj) + (q >> 1), x = 2 * (cx.i0 + ii) + (q & 1);
    if (y >= hb || x >= wb) return;
    big[(((long)img * hb + y) * wb + x) * Cb + cb] = bias ? v + bias[cb] : v;
  }
};

template <class EP, class = void> struct epi_reads_c : std::false_type {};
template <class EP> struct epi_reads_c<EP, std::enable_if_t<EP::READS_C>> : std::true_type {};

// ---------------------------------------------------------------------------
// Operands with a range that is exact in ONE bf16 plane (one-hot latents: the `stoch` columns of
// the feature matrix, nets.py:88-97 / tfutils.py:368-382 - every element 0 or 1).  Their middle
// and low planes are all zero, so of the six plane products only the three with the high plane of
// that operand can be non-zero: the loop issues those three in the same order and leaves out
// three exact `+ 0` updates of the accumulator - the result is bit-identical to the six-product
// loop at half the matrix instructions.  [x0, x1) = the exact columns of the STORED matrix:
// a k range when the operand is k-contiguous (not transposed), a row range of the tile otherwise.
// ---------------------------------------------------------------------------
template <class L>
struct ExactA : L {   // A operand
  int x0, x1;
  static constexpr bool EXACT_A = true;
};
template <class L>
struct ExactB : L {   // B operand (row range only: stored [K, N], columns = output columns)
  int x0, x1;
  static constexpr bool EXACT_B = true;
};
template <class AL, class = void> struct has_exact_a : std::false_type {};
template <class AL> struct has_exact_a<AL, std::enable_if_t<AL::EXACT_A>> : std::true_type {};
template <class BL, class = void> struct has_exact_b : std::false_type {};
template <class BL> struct has_exact_b<BL, std::enable_if_t<BL::EXACT_B>> : std::true_type {};

// ---------------------------------------------------------------------------
// Tile <-> workgroup mapping.
// ---------------------------------------------------------------------------

// XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch placement, used for speed only)
// and every XCD has its own 4 MiB L2.  Each XCD gets a CONTIGUOUS range of the linear tile
// order, in which the column tiles of one row tile are adjacent: the (large, streamed)
// row-operand tile is fetched from HBM once and re-read by the following column tiles from
// that XCD's L2, instead of once per column tile from eight different L2s (PMC, round 1:
// 3.0x the algorithmic bytes per launch).  tiles_m < 0 selects the plain column-major order
// (DD_XCD_SWIZZLE=0, for A/B measurements).
constexpr int TILES_STRIDED = 1 << 30;   // flag in the tiles_m kernel argument
__device__ __forceinline__ bool tile_coords_strided(int tiles_m, int& tm_idx, int& tn_idx);
__device__ __forceinline__ bool tile_coords(int tiles_m_signed, int& tm_idx, int& tn_idx) {
  if (tiles_m_signed > 0 && (tiles_m_signed & TILES_STRIDED))
    return tile_coords_strided(tiles_m_signed & ~TILES_STRIDED, tm_idx, tn_idx);
  const int T = (int)gridDim.x;
  const int tiles_m = tiles_m_signed < 0 ? -tiles_m_signed : tiles_m_signed;
  int tile = (int)blockIdx.x;
  if (tiles_m_signed > 0 && T >= 16) {
    const int per = T >> 3, rem = T & 7;
    const int x = tile & 7, slot = tile >> 3;
    tile = x * per + (x < rem ? x : rem) + slot;
    const int tiles_n = T / tiles_m;
    tm_idx = tile / tiles_n;
    tn_idx = tile - tm_idx * tiles_n;
  } else {
    tm_idx = tile % tiles_m;
    tn_idx = tile / tiles_m;
  }
  return true;
}

// Row tiles of unequal cost (tile-context loaders: the tap bands of ConvUpAC, sorted by band):
// contiguous per-XCD ranges would hand whole bands - all the cheap or all the expensive tiles -
// to one XCD.  Here XCD x takes every 8th row tile (an even mix of the bands), and the column
// tiles of a row tile still run back to back on that XCD.  gridDim.x = ceil8(tiles_m) * tiles_n;
// returns false for the padding workgroups.
__device__ __forceinline__ bool tile_coords_strided(int tiles_m, int& tm_idx, int& tn_idx) {
  const int tm8 = (tiles_m + 7) & ~7;
  const int tiles_n = (int)gridDim.x / tm8;
  const int b = (int)blockIdx.x, x = b & 7, s = b >> 3;
  const int rowslot = s / tiles_n;
  tn_idx = s - rowslot * tiles_n;
  tm_idx = rowslot * 8 + x;
  return tm_idx < tiles_m;
}

inline int xcd_swizzle() {
  static const int on = getenv("DD_XCD_SWIZZLE") ? atoi(getenv("DD_XCD_SWIZZLE")) : 1;
  return on;
}

// ---------------------------------------------------------------------------
// Main loop.
// ---------------------------------------------------------------------------

// ST = global->register prefetch distance in k-tiles.  Few-tile problems (64x64 tiles,
// about one workgroup per CU) are latency-bound on the global loads of the next k-tile;
// keeping ST tiles in flight in registers hides that without needing more workgroups.
template <int BM, int BN, bool AKC, bool BKC, class AL, class BL, class EP, int BK = 16,
          int ST = (BM == 64 && BN == 64) ? 4 : 1>
__global__ void __launch_bounds__(256, 2)
k_mfma_gemm(AL al, BL bl, EP ep, int K, int kps, int tiles_m) {
  // row pitch of the k-major LDS tiles: +1 spreads the scalar transposing stores of a
  // k-contiguous operand over all banks; +4 keeps 16-B alignment for float4 stores.
  constexpr int PA = AKC ? BM + 1 : BM + 4, PB = BKC ? BN + 1 : BN + 4;
  constexpr int KQ = BK / 4;  // float4 chunks along k per row
  __shared__ __attribute__((aligned(16))) float As[2][BK * PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tmi, tni;
  if (!tile_coords(tiles_m, tmi, tni)) return;
  const int m0 = tmi * BM, n0 = tni * BN;
  const int kb = blockIdx.z * kps;
  const int ke = min(K, kb + kps);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;
  static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float ra_[ST][NA][4], rb_[ST][NB][4];

  auto gload_t = [&](int k0, float (&ra)[NA][4], float (&rb)[NB][4], auto full) {
    constexpr bool FULL = decltype(full)::value;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int id = tid + i * 256;
      if (AKC) al.template load4<FULL>(m0 + id / KQ, k0 + (id % KQ) * 4, ke, ra[i]);
      else     al.template load4<FULL>(m0 + (id % (BM / 4)) * 4, k0 + id / (BM / 4), ke, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int id = tid + i * 256;
      if (BKC) bl.template load4<FULL>(n0 + id / KQ, k0 + (id % KQ) * 4, ke, rb[i]);
      else     bl.template load4<FULL>(n0 + (id % (BN / 4)) * 4, k0 + id / (BN / 4), ke, rb[i]);
    }
  };
  // interior k-tiles take the path without any K-tail handling (wave-uniform branch)
  auto gload = [&](int k0, float (&ra)[NA][4], float (&rb)[NB][4]) {
    if (k0 + BK <= ke) gload_t(k0, ra, rb, std::true_type());
    else gload_t(k0, ra, rb, std::false_type());
  };
  auto sstore = [&](int buf, float (&ra)[NA][4], float (&rb)[NB][4]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int id = tid + i * 256;
      if (AKC) {
        int r = id / KQ, kq = (id % KQ) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) As[buf][(kq + j) * PA + r] = ra[i][j];
      } else {
        int k = id / (BM / 4), r = (id % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(&As[buf][k * PA + r]) =
            make_float4(ra[i][0], ra[i][1], ra[i][2], ra[i][3]);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int id = tid + i * 256;
      if (BKC) {
        int r = id / KQ, kq = (id % KQ) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) Bs[buf][(kq + j) * PB + r] = rb[i][j];
      } else {
        int k = id / (BN / 4), r = (id % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(&Bs[buf][k * PB + r]) =
            make_float4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]);
      }
    }
  };

  const int nk = (ke - kb + BK - 1) / BK;
  // prologue: tiles 0..ST-1 in flight, tile 0 staged into LDS
#pragma unroll
  for (int s = 0; s < ST; ++s)
    if (s < nk) gload(kb + s * BK, ra_[s], rb_[s]);
  if (nk > 0) sstore(0, ra_[0], rb_[0]);
  __syncthreads();
  const int lk = lane >> 5, lr = lane & 31;
  for (int t0 = 0; t0 < nk; t0 += ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      const int t = t0 + s;
      if (t < nk) {
        const int buf = t & 1;
#ifndef EXP_NOLOAD  // ablation switches (tools/gemm_exp.py), see DESIGN.md section 5
        // register slot s held tile t (already in LDS): refill it with tile t + ST
        if (t + ST < nk) gload(kb + (t + ST) * BK, ra_[s], rb_[s]);
#endif
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
          float af[TM], bf[TN];
#pragma unroll
          for (int a = 0; a < TM; ++a) af[a] = As[buf][(kk + lk) * PA + wm0 + a * 32 + lr];
#pragma unroll
          for (int b = 0; b < TN; ++b) bf[b] = Bs[buf][(kk + lk) * PB + wn0 + b * 32 + lr];
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
#ifndef EXP_NOSTORE
        if (t + 1 < nk) sstore(buf ^ 1, ra_[(s + 1) % ST], rb_[(s + 1) % ST]);
#endif
#ifndef EXP_NOBARRIER
        __syncthreads();
#endif
      }
    }
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
      {
        const int col = n0 + wn0 + b * 32 + lr;
        bool batched = false;
        if constexpr (epi_reads_c<EP>::value) {
          if (ep.wants_old()) {
            batched = true;
            float oldv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              oldv[r] = ep.read_old(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col);
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ep.put(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r], oldv[r]);
          }
        }
        if (!batched) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ep(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
        }
      }
}

// ---------------------------------------------------------------------------
// Split-bf16 main loop: fp32 operands decomposed exactly into three bf16 terms
// (x = hi + mid + lo, 8 significand bits each, by truncation), products on the bf16
// matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the rate of the fp32 MFMA), fp32
// accumulation.  NP = 6 keeps every cross term down to 2^-16 relative (hi*hi, hi*mid,
// mid*hi, mid*mid, hi*lo, lo*hi): what is dropped is below 2^-22 of a product, i.e.
// fp32-level accuracy at 16/6 the fp32-MFMA peak.  NP = 3 (hi*hi, hi*mid, mid*hi) is an
// experiment switch only (2^-15 relative, reduced precision; never the default).
//
// LDS: three bit-planes per operand, laid out by the operand's contiguous axis so that
// both the stores and the fragment reads are wide and conflict-free:
//   k-contiguous operand   plane[koct][row] = 16-byte slot of 8 consecutive k; a staged
//                          float4 (row, 4 k) is one ds_write_b64, the MFMA operand (row
//                          l&31, k-octet l>>5) one ds_read_b128;
//   row-contiguous operand plane[k][row] bf16; a staged float4 (4 rows, k) is one
//                          ds_write_b64, the MFMA operand two ds_read_b64_tr_b16 (the
//                          gfx950 LDS transpose read: a 16-lane group reads a 4(k) x 16(row)
//                          block, lane i receives row i's four k).
// Strides are padded to 64 mod 256 bytes (koct / k rows land on different bank quarters).
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
#ifdef DD_ABL_NOSPLIT   // ablation (tools/abl_gemm.sh): no split arithmetic (wrong results, timing only)
  h = m = l = __float_as_uint(x);
  return;
#endif
  h = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);          // exact
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));     // exact, <= 8 significant bits
}
// round-to-nearest-even bf16 of an fp32 word, kept in the top half (the single-plane mode
// NP = 1: truncation would bias every product by up to 2^-8)
__device__ __forceinline__ unsigned rne_bf16(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}
// bf16(even) | bf16(odd) << 16 from the top halves of two words
__device__ __forceinline__ unsigned pack_hi(unsigned even, unsigned odd) {
  return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}

template <int BX, bool KC, int BK>
struct PlaneS3 {
  // bytes per k-octet (KC) / per k (RC); the pad staggers bank quarters
#ifndef DD_RCPAD
#define DD_RCPAD 64
#endif
  static constexpr int STR = KC ? BX * 16 + (BK == 16 ? 64 : 32) : BX * 2 + DD_RCPAD;
  static constexpr int BYTES = KC ? (BK / 8) * STR : BK * STR;
  // float4 units staged per tile: KC chunks (row, 4 k), RC chunks (4 rows, k); UNITS / 256 per
  // thread, every thread active (a partially active workgroup puts the loads behind a
  // divergent branch, and the compiler then waits for ALL outstanding loads - vmcnt 3 instead
  // of 9 - before staging: the register prefetch distance collapsed to one k-tile and the
  // 64x64 loop ran at one memory latency per k-tile, with or without the MFMAs)
  static constexpr int UNITS = BX * BK / 4;            // float4s per tile
  static_assert(UNITS % 256 == 0, "tile too small for 256 staging threads");
  static constexpr int N = UNITS / 256;
  static constexpr int ACTIVE = 256;
  // unit u of thread tid -> (row, k) of its first element
  static __device__ __forceinline__ void coord(int tid, int u, int& r, int& k) {
    const int id = tid + u * 256;
    if constexpr (KC) {
      r = id / (BK / 4); k = (id % (BK / 4)) * 4;
    } else {
      k = id / (BX / 4); r = (id % (BX / 4)) * 4;
    }
  }
  // staged float4 -> one 8-byte store per plane
  template <int NPL>
  static __device__ __forceinline__ void store(unsigned char* base, const float (&v)[4], int r, int k) {
    unsigned h[4], m[4], l[4];
    const int o = KC ? (k >> 3) * STR + r * 16 + (k & 4) * 2 : k * STR + r * 2;
    if constexpr (NPL == 1) {  // bf16-input mode: one rounded plane
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = rne_bf16(v[j]);
      *reinterpret_cast<uint2*>(base + o) = make_uint2(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]));
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(v[j], h[j], m[j], l[j]);
    *reinterpret_cast<uint2*>(base + o) = make_uint2(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]));
    *reinterpret_cast<uint2*>(base + BYTES + o) = make_uint2(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]));
    if (NPL == 3)
      *reinterpret_cast<uint2*>(base + 2 * BYTES + o) = make_uint2(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]));
  }
  // MFMA operand of the 32-row block starting at row0, k-step ks (16 k each), this lane
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane, int row0, int ks, int lane) {
    const int lk = lane >> 5;
    if constexpr (KC) {
      uint4 q = *reinterpret_cast<const uint4*>(plane + (2 * ks + lk) * STR + (row0 + (lane & 31)) * 16);
      return __builtin_bit_cast(bf16x8, q);
    } else {
      const int i = lane & 15, g = (lane >> 4) & 1;
      const unsigned char* p = plane + (16 * ks + 8 * lk + (i >> 2)) * STR + (row0 + 16 * g + 4 * (i & 3)) * 2;
      typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
      bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p));
      bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p + 4 * STR));
      uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
      return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
    }
  }
};

// Workgroups per CU the register allocation is held to.  -DDD_OCC3=2 -DDD_RCPAD=32 asks for three
// on the 128x128 tile when at least one operand is k-contiguous (LDS image 50.7 / 53.0 KB,
// <= 168 VGPRs without spills): measured -6..7 % on the NT shapes and -1..3 % on NN in isolation
// (the loop's MFMA half and its staging half barely overlap inside one workgroup,
// profiles/r02_gemm_ablation_128tile.txt), sequential step 40.4 -> 40.2 ms, but the pipelined
// step 33.5 -> 33.9 ms (same box, alternating runs): three resident workgroups leave the other
// stream's kernels less room.  Not the default.
#ifndef DD_OCC3
#define DD_OCC3 0
#endif
#if DD_OCC3 == 2
#define DD_OCC(BM, BN, AKC, BKC) (((BM) == 128 && (BN) == 128 && ((AKC) || (BKC))) ? 3 : 2)
#elif DD_OCC3 == 1
#define DD_OCC(BM, BN, AKC, BKC) (((BM) == 128 && (BN) == 128 && (AKC) && (BKC)) ? 3 : 2)
#else
#define DD_OCC(BM, BN, AKC, BKC) ((BM) >= 256 ? 1 : 2)
#endif

// Raised wave priority around the MFMA block of a k-step (s_setprio): the waves of the two
// resident workgroups then alternate roles - one issues its MFMAs while the other stages -
// instead of both competing for the matrix pipe and then both staging.  DD_PRIO_MASK selects
// the operand layouts it is applied to: bit 0 k-contiguous x k-contiguous, 1 kc x rc, 2 rc x rc,
// 3 rc x kc.
#ifndef DD_PRIO_MASK
#define DD_PRIO_MASK 9
#endif
#define DD_PRIO_ON(AKC, BKC) (((DD_PRIO_MASK) >> (((AKC) ? 0 : 2) + (((AKC) != (BKC)) ? 1 : 0))) & 1)

// A2: a second accumulator per output block for the three small-term products.  A wave of a
// 64x64 tile owns ONE 32x32 block, so its six products per k-step form one dependent MFMA
// chain (each waits for the previous result); two chains of three halve that latency.  The
// partial sums are added once, after the K loop (small terms + large terms).
template <int BM, int BN, bool AKC, bool BKC, class AL, class BL, class EP, int NP = 6, int BK = 16,
          int ST = 1, bool IL = false, bool A2 = false>
__global__ void __launch_bounds__(256, DD_OCC(BM, BN, AKC, BKC))
k_mfma_gemm_s3(AL al, BL bl, EP ep, int K, int kps, int tiles_m) {
  constexpr int NPL = NP == 1 ? 1 : (NP == 3 ? 2 : 3);     // planes kept
  using LA = PlaneS3<BM, AKC, BK>;
  using LB = PlaneS3<BN, BKC, BK>;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][NPL * LA::BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][NPL * LB::BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tmi, tni;
  if (!tile_coords(tiles_m, tmi, tni)) return;
  // loaders with a per-row-tile context (ConvUpAC: the tap band of the tile): rows are
  // context-local and the contraction length is the tile's own
  constexpr bool TC = has_tile_ctx<AL>::value;
  const auto cx = get_tile_ctx<BM>(al, tmi);
  int m0 = tmi * BM;
  const int n0 = tni * BN;
  const int kb = blockIdx.z * kps;
  int ke = min(K, kb + kps);
  if constexpr (TC) { m0 = cx.row0; ke = cx.keff; }
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  constexpr int NA = LA::N, NB = LB::N;
  const bool a_on = LA::ACTIVE >= 256 || tid < LA::ACTIVE;
  const bool b_on = LB::ACTIVE >= 256 || tid < LB::ACTIVE;

  f32x16 acc[TM][TN], acs[A2 ? TM : 1][A2 ? TN : 1];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[a][b][r] = 0.f;
        if (A2) acs[a][b][r] = 0.f;
      }

  float ra_[ST][NA][4], rb_[ST][NB][4];

  // tile-context loaders: per-thread element offsets of the staged units at k0 = 0
  [[maybe_unused]] long abase[TC ? NA : 1], bbase[TC ? NB : 1];
  if constexpr (TC) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      int r, k; LA::coord(tid, u, r, k);
      abase[u] = al.row_base(cx, m0 + r) + k;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      int r, k; LB::coord(tid, u, r, k);
      bbase[u] = bl.row_base(cx, n0 + r) + k;
    }
  }

  auto gload = [&](int k0, float (&ra)[NA][4], float (&rb)[NB][4], auto full) {
    constexpr bool FULL = decltype(full)::value;
    if constexpr (TC) {   // (the tile's contraction length is a whole number of k-tiles)
      const float* pa = al.ptr() + al.k_off(cx, k0);
      const float* pb = bl.ptr() + bl.k_off(cx, k0);
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const float4 t = *reinterpret_cast<const float4*>(pa + abase[u]);
        ra[u][0] = t.x; ra[u][1] = t.y; ra[u][2] = t.z; ra[u][3] = t.w;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const float4 t = *reinterpret_cast<const float4*>(pb + bbase[u]);
        rb[u][0] = t.x; rb[u][1] = t.y; rb[u][2] = t.z; rb[u][3] = t.w;
      }
      return;
    }
    if (a_on) {
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        int r, k; LA::coord(tid, u, r, k);
        if constexpr (!TC) al.template load4<FULL>(m0 + r, k0 + k, ke, ra[u]);
      }
    }
    if (b_on) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        int r, k; LB::coord(tid, u, r, k);
        if constexpr (!TC) bl.template load4<FULL>(n0 + r, k0 + k, ke, rb[u]);
      }
    }
  };
  auto gload_rt = [&](int k0, float (&ra)[NA][4], float (&rb)[NB][4]) {
    if (k0 + BK <= ke) gload(k0, ra, rb, std::true_type());
    else gload(k0, ra, rb, std::false_type());
  };
  // xs = 1 / 2: the staged tile of A / B lies in the operand's exact range - only its high plane is
  // written (the values are bf16-exact: the rounded plane IS the value; the multiply of such a
  // tile reads nothing else), no split arithmetic, a third of that operand's LDS stores
  auto sstore = [&](int buf, float (&ra)[NA][4], float (&rb)[NB][4], auto xs_tag) {
    constexpr int XS = decltype(xs_tag)::value;
    if (a_on) {
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        int r, k; LA::coord(tid, u, r, k);
        LA::template store<(XS == 1 ? 1 : NPL)>(As[buf], ra[u], r, k);
      }
    }
    if (b_on) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        int r, k; LB::coord(tid, u, r, k);
        LB::template store<(XS == 2 ? 1 : NPL)>(Bs[buf], rb[u], r, k);
      }
    }
  };
  // xs = 0: all products; 1 / 2: A / B is exact in its high plane for this k-tile (ExactA /
  // ExactB), only the products with that plane are issued
  auto compute = [&](int buf, auto xs_tag) {
    constexpr int XS = decltype(xs_tag)::value;
    if constexpr (DD_PRIO_ON(AKC, BKC)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int p = 0; p < (XS == 1 ? 1 : NPL); ++p) af[a][p] = LA::frag(As[buf] + p * LA::BYTES, wm0 + a * 32, ks, lane);
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int p = 0; p < (XS == 2 ? 1 : NPL); ++p) bf[b][p] = LB::frag(Bs[buf] + p * LB::BYTES, wn0 + b * 32, ks, lane);
      // product (pa, pb) of the bit-planes, smallest terms first; consecutive MFMAs go
      // to different accumulators
      constexpr int PA_[6] = {NPL - 1, 0, 1, 1, 0, 0}, PB_[6] = {0, NPL - 1, 1, 0, 1, 0};
      // (A2: q = 0..2 go to the small-term accumulator, q = 3..5 to the main one; the order
      // 0,3,1,4,2,5 alternates the two chains)
      constexpr int ORD_[6] = {0, 3, 1, 4, 2, 5};
#pragma unroll
      for (int qi = (NP == 6 ? 0 : (NP == 3 ? 3 : 5)); qi < 6; ++qi)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const int q = (A2 && NP == 6) ? ORD_[qi] : qi;
            if constexpr (XS == 1) { if (PA_[q] != 0) continue; }
            if constexpr (XS == 2) { if (PB_[q] != 0) continue; }
#ifdef DD_ABL_NOMFMA    // ablation: fragment reads kept alive, no matrix instruction
            acc[a][b][q] += (float)af[a][PA_[q]][0] + (float)bf[b][PB_[q]][0];
#elif defined(DD_ABL_ONEMFMA)  // ablation: one product instead of six (dependent-chain length)
            if (q == 5) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA_[q]], bf[b][PB_[q]], acc[a][b], 0, 0, 0);
            else acc[a][b][q] += (float)af[a][PA_[q]][0] + (float)bf[b][PB_[q]][0];
#else
            if (A2 && q < 3)
              acs[A2 ? a : 0][A2 ? b : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  af[a][PA_[q]], bf[b][PB_[q]], acs[A2 ? a : 0][A2 ? b : 0], 0, 0, 0);
            else
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA_[q]], bf[b][PB_[q]], acc[a][b], 0, 0, 0);
#endif
          }
    }
    if constexpr (DD_PRIO_ON(AKC, BKC)) __builtin_amdgcn_s_setprio(0);
  };

  // k-tiles 0..nfull-1 are whole, tile nfull (if any) is the K tail.  ST tiles are in
  // flight in registers (slot = tile % ST), tile t+1 is staged into LDS while tile t is
  // multiplied.  The steady-state loop has exactly one code path (whole-tile prefetch):
  // with the tail variant inside it the register allocator shares load destinations
  // between the variants and the hardware then waits for the prefetch before the MFMAs
  // instead of after them.
  const int nfull = (ke - kb) / BK, nk = (ke - kb + BK - 1) / BK;
  // k-tiles [tx0, tx1) of this workgroup lie inside the exact range of an ExactA / ExactB
  // operand (both multiples of ST; empty without such an operand)
  constexpr int XSEL = (NP == 6 && has_exact_a<AL>::value) ? 1 : ((NP == 6 && has_exact_b<BL>::value) ? 2 : 0);
  int tx0 = 0, tx1 = 0;
  if constexpr (XSEL == 1) {
    if constexpr (AKC) {   // k range
      tx0 = max(0, (al.x0 - kb + BK - 1) / BK);
      tx0 = (tx0 + ST - 1) / ST * ST;
      tx1 = al.x1 > kb ? (min(al.x1, ke) - kb) / BK / ST * ST : 0;
    } else if (m0 >= al.x0 && m0 + BM <= al.x1) {
      tx1 = nk + ST;
    }
  }
  if constexpr (XSEL == 2) {
    static_assert(XSEL != 2 || !BKC, "ExactB: row-contiguous B only");
    if (n0 >= bl.x0 && n0 + BN <= bl.x1) tx1 = nk + ST;
  }
  if (tx1 < tx0) tx1 = tx0;
#pragma unroll
  for (int s_ = 0; s_ < ST; ++s_)
    if (s_ < nk) gload_rt(kb + s_ * BK, ra_[s_], rb_[s_]);
  if (nk > 0) sstore(0, ra_[0], rb_[0], std::integral_constant<int, 0>());
  __syncthreads();
  // one steady-state iteration group (ST k-tiles) with the product set XS
  auto steady = [&](int t0, auto xs_tag) {
    constexpr int XS = decltype(xs_tag)::value;
#pragma unroll
    for (int s_ = 0; s_ < ST; ++s_) {
      const int t = t0 + s_;
      gload(kb + (t + ST) * BK, ra_[s_], rb_[s_], std::true_type());
      // keep the prefetch issued ahead of the MFMA block (the scheduler otherwise sinks
      // the loads next to their first use, behind the MFMAs, and exposes their latency)
      __builtin_amdgcn_sched_barrier(0);
      compute(t & 1, xs_tag);
      if (!IL) __builtin_amdgcn_sched_barrier(0);
      if constexpr (XS != 0) {   // (tile t + 1 of an exact segment is exact too, except behind its last tile)
        if (t + 1 < tx1) sstore((t & 1) ^ 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST], xs_tag);
        else sstore((t & 1) ^ 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST], std::integral_constant<int, 0>());
      } else {
        sstore((t & 1) ^ 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST], std::integral_constant<int, 0>());
      }
      if (IL) {  // tile t+1 arrived an iteration ago: split + stage it under the MFMAs
        constexpr int NMF = TM * TN * (XS ? 3 : NP) * (BK / 16);
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, XS ? 8 : 4, 0);
          if (XS || (i & 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  };
  int t0 = 0;
  if constexpr (XSEL != 0) {
    // (three loops, each with one code path: see the note above about variants inside the loop)
    const int lim = nfull - 2 * ST + 1;      // t0 < lim  <=>  t0 + 2 ST <= nfull
    for (; t0 < min(tx0, lim); t0 += ST) steady(t0, std::integral_constant<int, 0>());
    for (; t0 < min(tx1, lim); t0 += ST) steady(t0, std::integral_constant<int, XSEL>());
  }
  for (; t0 + 2 * ST <= nfull; t0 += ST) steady(t0, std::integral_constant<int, 0>());
  for (; t0 < nk; t0 += ST) {
#pragma unroll
    for (int s_ = 0; s_ < ST; ++s_) {
      const int t = t0 + s_;
      if (t < nk) {
        if (t + ST < nk) gload_rt(kb + (t + ST) * BK, ra_[s_], rb_[s_]);
        if constexpr (XSEL != 0) {
          if (t >= tx0 && t < tx1 && t < nfull) compute(t & 1, std::integral_constant<int, XSEL>());
          else compute(t & 1, std::integral_constant<int, 0>());
        } else {
          compute(t & 1, std::integral_constant<int, 0>());
        }
        if (t + 1 < nk) sstore((t & 1) ^ 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST], std::integral_constant<int, 0>());
        __syncthreads();
      }
    }
  }

  if (A2) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += acs[A2 ? a : 0][A2 ? b : 0][r];
  }
  const int lk = lane >> 5, lr = lane & 31;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
      {
        const int col = n0 + wn0 + b * 32 + lr;
        bool batched = false;
        if constexpr (epi_reads_c<EP>::value) {
          if (ep.wants_old()) {
            batched = true;
            float oldv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              oldv[r] = ep.read_old(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col);
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ep.put(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r], oldv[r]);
          }
        }
        if (!batched) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (TC) ep.putc(cx, m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
            else ep(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
          }
        }
      }
}

// Role-separated form of the split-bf16 loop for the 128x128 tile (round 3: tools/exp_ws,
// profiles/r03_gemm_wave_specialised.txt; shipped in round 4 behind the per-launch selector
// ws_selected below): 512 threads, waves 0-3 only read fragments and issue MFMAs, waves 4-7 only
// load, split and store the next k-tile.  Each role runs its OWN loop with the same number of
// barriers, so the register allocator never sees accumulators and staging registers alive
// together (104 VGPRs: two 8-wave workgroups per CU = two MFMA waves + two staging waves per
// SIMD, the matrix pipe of a SIMD always has a second wave to draw from while the first waits at
// the barrier or for its fragments).  Same LDS image, same product order as
// k_mfma_gemm_s3<128, 128, ..., 6>: results are bit-identical.
template <bool AKC, bool BKC, class AL, class BL, class EP, int ST = 2>
__global__ void __launch_bounds__(512, 4)
k_mfma_gemm_ws(AL al, BL bl, EP ep, int K, int kps, int tiles_m) {
  constexpr int BM = 128, BN = 128, BK = 16, NPL = 3;
  using LA = PlaneS3<BM, AKC, BK>;
  using LB = PlaneS3<BN, BKC, BK>;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][NPL * LA::BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][NPL * LB::BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tmi, tni;
  if (!tile_coords(tiles_m, tmi, tni)) return;
  constexpr bool TC = has_tile_ctx<AL>::value;
  const auto cx = get_tile_ctx<BM>(al, tmi);
  int m0 = tmi * BM;
  const int n0 = tni * BN;
  const int kb = blockIdx.z * kps;
  int ke = min(K, kb + kps);
  if constexpr (TC) { m0 = cx.row0; ke = cx.keff; }
  const int nk = (ke - kb + BK - 1) / BK;
  if (wave >= 4) {   // ---- staging waves
    const int rt = tid & 255;
    constexpr int NA = LA::N, NB = LB::N;
    float ra_[ST][NA][4], rb_[ST][NB][4];
    [[maybe_unused]] long abase[TC ? NA : 1], bbase[TC ? NB : 1];
    if constexpr (TC) {
#pragma unroll
      for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); abase[u] = al.row_base(cx, m0 + r) + k; }
#pragma unroll
      for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bbase[u] = bl.row_base(cx, n0 + r) + k; }
    }
    auto gload = [&](int t, float (&ra)[NA][4], float (&rb)[NB][4]) {
      const int k0 = kb + t * BK;
      if constexpr (TC) {   // (the tile's contraction length is a whole number of k-tiles)
        const float* pa = al.ptr() + al.k_off(cx, k0);
        const float* pb = bl.ptr() + bl.k_off(cx, k0);
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          const float4 q = *reinterpret_cast<const float4*>(pa + abase[u]);
          ra[u][0] = q.x; ra[u][1] = q.y; ra[u][2] = q.z; ra[u][3] = q.w;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const float4 q = *reinterpret_cast<const float4*>(pb + bbase[u]);
          rb[u][0] = q.x; rb[u][1] = q.y; rb[u][2] = q.z; rb[u][3] = q.w;
        }
      } else if (k0 + BK <= ke) {
#pragma unroll
        for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<true>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
        for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<true>(n0 + r, k0 + k, ke, rb[u]); }
      } else {
#pragma unroll
        for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<false>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
        for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<false>(n0 + r, k0 + k, ke, rb[u]); }
      }
    };
    auto sstore = [&](int buf, float (&ra)[NA][4], float (&rb)[NB][4]) {
#pragma unroll
      for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); LA::template store<NPL>(As[buf], ra[u], r, k); }
#pragma unroll
      for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); LB::template store<NPL>(Bs[buf], rb[u], r, k); }
    };
#pragma unroll
    for (int s_ = 0; s_ < ST; ++s_)
      if (s_ < nk) gload(s_, ra_[s_], rb_[s_]);
    if (nk > 0) sstore(0, ra_[0], rb_[0]);
    if (ST < nk) gload(ST, ra_[0], rb_[0]);
    __syncthreads();
    for (int t0 = 0; t0 < nk; t0 += ST) {
#pragma unroll
      for (int s_ = 0; s_ < ST; ++s_) {
        const int t = t0 + s_;
        if (t < nk) {
          if (t + 1 < nk) sstore((t + 1) & 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
          if (t + 1 + ST < nk) gload(t + 1 + ST, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
          __syncthreads();
        }
      }
    }
    return;
  }
  // ---- MFMA waves (2 x 2 wave tiles of 64 x 64)
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int p = 0; p < NPL; ++p) af[a][p] = LA::frag(As[buf] + p * LA::BYTES, wm0 + a * 32, 0, lane);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int p = 0; p < NPL; ++p) bf[b][p] = LB::frag(Bs[buf] + p * LB::BYTES, wn0 + b * 32, 0, lane);
    constexpr int PA_[6] = {NPL - 1, 0, 1, 1, 0, 0}, PB_[6] = {0, NPL - 1, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA_[q]], bf[b][PB_[q]], acc[a][b], 0, 0, 0);
    __syncthreads();
  }
  const int lk = lane >> 5, lr = lane & 31;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn0 + b * 32 + lr;
      bool batched = false;
      if constexpr (epi_reads_c<EP>::value) {
        if (ep.wants_old()) {
          batched = true;
          float oldv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            oldv[r] = ep.read_old(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ep.put(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r], oldv[r]);
        }
      }
      if (!batched) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (TC) ep.putc(cx, m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
          else ep(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
        }
      }
    }
}

// Which launches of the 128x128 tile take the role-separated loop: none by default.
// Round 3 measured it x1.1-1.33 ahead on deep-K shapes in 13-launch timing loops; in steady state
// (round 4, tools/r04/ws_steady.py, profiles/r04_ws_steady.txt: 120 back-to-back launches, last
// 60 averaged) both loops run at the same rate on every shape of the step - 4096^3 717 vs 716 us,
// 422500x128x2304 1487 vs 1503, 40000x512x1280 284 vs 295, 40000x512x512 128 vs 130 - and the
// train step's per-call-site times do not move (profiles/r04_ws_selector.txt: 27.89 vs 27.67 ms).
// The same record shows why: with ZERO operands the identical instruction stream runs 4096^3 in
// 484 us instead of 717 (284 vs 191 TFLOP/s) - the loop is bound by the chip's power-limited
// clock on real data (~1.55 vs ~2.3 GHz), not by its schedule, so a schedule that keeps the matrix
// pipe busier is paid back in clock.  Kept as a selectable variant of an opt-in build (`make WS=1`
// defines DD_BUILD_WS; the default library does not instantiate it) (dd_gemm_set_ws / DD_WS=1;
// launches with >= DD_WS_KMIN contraction elements per split-K slab, DD_WS_KMIN_TC for the banded
// transposed-convolution launches) and covered by the parity tests (bit-identical results).
inline bool ws_selected(int kps, bool tile_ctx) {
  if (g_ws_select[0] < 0) {
    g_ws_select[0] = getenv("DD_WS") ? atoi(getenv("DD_WS")) : 0;
    g_ws_select[1] = getenv("DD_WS_KMIN") ? atoi(getenv("DD_WS_KMIN")) : 1024;
    g_ws_select[2] = getenv("DD_WS_KMIN_TC") ? atoi(getenv("DD_WS_KMIN_TC")) : 1024;
  }
  return g_ws_select[0] && kps >= g_ws_select[tile_ctx ? 2 : 1];
}

// 0 = native fp32 MFMA, 6 = split-bf16 with six products (fp32-level accuracy, default),
// 1 = bf16 inputs (operands rounded to bf16, one product, fp32 accumulation): the opt-in
//     reduced-precision mode `hip.precision: bfloat16`, the counterpart of the reference's
//     tf.precision float16 (tfagent.py:161-168, tfutils.py:164-167), separately toleranced.
inline int gemm_mode() {
  if (g_gemm_mode < 0) {
    const char* e = getenv("DD_GEMM_MODE");
    g_gemm_mode = e ? atoi(e) : 6;
  }
  return g_gemm_mode;
}

// launch the main loop for one tile shape in the selected arithmetic mode.  Measured
// variants of the split-bf16 loop (tools/gemm_modes.py, bench.py): 128-row tiles: prefetch
// distance 2 with the split of tile t+1 interleaved under the MFMAs of tile t (52.6 vs
// 53.2 ms/step for distance 1, no interleave); 64x64 tiles (latency-bound, about one
// workgroup per CU): distance 4 (BK 16) beats BK 32 x distance 2 and the fp32 loop.
template <int BM, int BN, bool AKC, bool BKC, class AL, class BL, class EP>
void launch_tile(dim3 grid, hipStream_t st, AL al, BL bl, EP ep, int K, int kps, int tm) {
  // tile order: 0 plain, 1 contiguous range per XCD, 2 row tiles dealt to the XCDs in turn
  // (always for tile-context loaders: their row tiles are sorted by cost)
  const int order = has_tile_ctx<AL>::value ? 2 : xcd_swizzle();
  if (order == 0) tm = -tm;
  if (order == 2) {
    const int tiles_n = (int)grid.x / tm;
    grid.x = (unsigned)(((tm + 7) & ~7) * tiles_n);
    tm |= TILES_STRIDED;
  }
  // (exact-range operands: six-product mode only, the caller checks the mode)
  constexpr bool XE = has_exact_a<AL>::value || has_exact_b<BL>::value;
  if constexpr (!has_tile_ctx<AL>::value && BM <= 128 && !XE) {   // (tile-context loaders: split loop only, the caller checks the mode)
    if (gemm_mode() == 0) {
      k_mfma_gemm<BM, BN, AKC, BKC, AL, BL, EP><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm);
      return;
    }
  }
  if constexpr (!XE) {
    if (gemm_mode() == 1) {
      if constexpr (BM == 64 && BN == 64)
        k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 1, 16, 4, false><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm);
      else
        k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 1, 16, 2, false><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm);
      return;
    }
  }
  // (64x64 tiles, about one workgroup per CU: prefetch distance 8 measured equal to 4 on every
  // 2500-row and 50-row shape of the step - round 2, profiles/r02_gemm_prefetch_distance.txt -
  // so that loop is not bound by the global-load latency; distance 4 keeps the registers)
  if constexpr (BM == 64 && BN == 64) {
#ifdef DD_EXP64   // experiment build (tools/exp64.sh): loop variants of the 64x64 tile by DD_V64
    static const int v64 = getenv("DD_V64") ? atoi(getenv("DD_V64")) : 0;
    if (v64 == 4) { k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 16, 8, false, false><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm); return; }
    if (v64 == 5) { k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 16, 8, false, true><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm); return; }
    if (v64 == 6) { k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 16, 8, true, true><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm); return; }
    if (v64 == 7) { k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 32, 2, false, false><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm); return; }
    if (v64 == 8) { k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 32, 4, false, false><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm); return; }
    if (v64 == 9) { k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 32, 2, false, true><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm); return; }
#endif
    k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 16, 4, false, DD_A2_64><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm);
  } else {
#ifdef DD_BUILD_WS   // (`make WS=1`: the role-separated loop, measured equal in steady state - not in the default build)
    if constexpr (BM == 128 && BN == 128 && !XE) {
      if (ws_selected(kps, has_tile_ctx<AL>::value)) {
        k_mfma_gemm_ws<AKC, BKC, AL, BL, EP><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tm);
        return;
      }
    }
#endif
    k_mfma_gemm_s3<BM, BN, AKC, BKC, AL, BL, EP, 6, 16, 2, true><<<grid, 256, 0, st>>>(al, bl, ep, K, kps, tm);
  }
}

__global__ void k_splitk_reduce(const float* __restrict__ slab, int S, long MN, int N,
                                float* C, long ldc, const float* bias, float alpha, float beta) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < MN;
       i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slab[(long)z * MN + i];
    long m = i / N; int n = (int)(i - m * N);
    float r = alpha * s;
    if (bias) r += bias[n];
    long o = m * ldc + n;
    if (beta != 0.f) r += beta * C[o];
    C[o] = r;
  }
}

// the same sum (slabs ascending, then alpha, bias, beta * old: bit-identical) four columns
// per thread with 16-byte accesses and the loads of four slabs in flight together;
// N % 4 == 0, ldc % 4 == 0, 16-byte aligned C / bias / slabs, MN < 2^33
__global__ void __launch_bounds__(256)
k_splitk_reduce4(const float* __restrict__ slab, int S, long MN, int N4,
                 float* C, long ldc, const float* bias, float alpha, float beta) {
  const unsigned quads = (unsigned)(MN >> 2);
  for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
    const float* p = slab + 4l * q;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= S; z += 4) {
      const float4 t0 = *reinterpret_cast<const float4*>(p + (z + 0) * MN);
      const float4 t1 = *reinterpret_cast<const float4*>(p + (z + 1) * MN);
      const float4 t2 = *reinterpret_cast<const float4*>(p + (z + 2) * MN);
      const float4 t3 = *reinterpret_cast<const float4*>(p + (z + 3) * MN);
      f4_acc(s, t0); f4_acc(s, t1); f4_acc(s, t2); f4_acc(s, t3);
    }
    for (; z < S; ++z) f4_acc(s, *reinterpret_cast<const float4*>(p + z * MN));
    const unsigned m = q / (unsigned)N4, n = (q - m * (unsigned)N4) * 4u;
    float4 r = make_float4(alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w;
    }
    float4* o = reinterpret_cast<float4*>(C + (long)m * ldc + n);
    if (beta != 0.f) {
      const float4 c = *o;
      r.x += beta * c.x; r.y += beta * c.y; r.z += beta * c.z; r.w += beta * c.w;
    }
    *o = r;
  }
}

inline void launch_splitk_reduce(const float* slab, int S, long MN, int N, float* C, long ldc,
                                 const float* bias, float alpha, float beta, hipStream_t st) {
  const bool v4 = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)slab & 15) == 0 &&
                  ((uintptr_t)bias & 15) == 0 && MN < (1l << 33);
  if (v4) {
    long blocks = (MN / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    k_splitk_reduce4<<<(int)blocks, 256, 0, st>>>(slab, S, MN, N / 4, C, ldc, bias, alpha, beta);
  } else {
    int blocks = (int)((MN + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    k_splitk_reduce<<<blocks, 256, 0, st>>>(slab, S, MN, N, C, ldc, bias, alpha, beta);
  }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Split-K factor: enough workgroups to cover the chip, bounded by K and the
// caller's workspace.
int pick_split(long tiles, int K, long MN, size_t ws_bytes) {
  static const int min_tiles = getenv("DD_SPLIT_MIN_TILES") ? atoi(getenv("DD_SPLIT_MIN_TILES")) : 192;
  if (tiles >= min_tiles || K < 128) return 1;
  // measured (2500x256xK): with >= 100 tiles a short K loop beats split + reduce
  if (tiles >= 100 && K <= 384) return 1;
  static const int kmin = getenv("DD_SPLIT_KMIN") ? atoi(getenv("DD_SPLIT_KMIN")) : 64;
  static const int target = getenv("DD_SPLIT_TARGET") ? atoi(getenv("DD_SPLIT_TARGET")) : 512;
  long s = (target + tiles - 1) / tiles;
  // never more workgroups than the `target` slots (256 CUs x 2 resident workgroups): the few
  // extra ones run as a second round after the first has finished - 18 tiles x 29 slabs = 522
  // workgroups took 1.89 ms where 18 x 28 = 504 take one round (the k = 6 filter gradient)
  static const int floor_split = getenv("DD_SPLIT_FLOOR") ? atoi(getenv("DD_SPLIT_FLOOR")) : 1;
  if (floor_split && s > 1 && s * tiles > target) s = target / tiles;
  long maxs = K / kmin;
  if (s > maxs) s = maxs;
  while (s > 1 && (size_t)s * MN * sizeof(float) > ws_bytes) --s;
  return (int)(s < 1 ? 1 : s);
}

template <bool AKC, bool BKC, class AL, class BL>
int run_mat(AL al, BL bl, int M, int N, int K, float* C, long ldc, const float* bias,
            float alpha, float beta, float* ws, size_t ws_bytes, hipStream_t st,
            const char* name, int* deferred = nullptr) {
  if (M <= 0 || N <= 0) return 0;
  // tiles: 128x128 for large problems, 128x64 for narrow outputs, 64x64 for few rows;
  // mid-size problems drop to smaller tiles until the grid covers the 256 CUs.
  int TMS = (M > 64) ? 128 : 64;
  int TNS = (M > 64 && N > 64) ? 128 : 64;
  // (deep-K problems keep the big tile and get their parallelism from split-K)
  const bool shallow = K <= 1536;
  if (shallow && TMS == 128 && TNS == 128 && (long)dd_ceil_div(M, 128) * dd_ceil_div(N, 128) < 256) TNS = 64;
  // (measured on 2500-row problems: 64x64 wins up to ~500 128x64-tiles for K <= 768,
  // 128x64 wins for deeper K unless it leaves most CUs idle)
  if (shallow && TMS == 128 && TNS == 64 &&
      (long)dd_ceil_div(M, 128) * dd_ceil_div(N, 64) < (K > 768 ? 128 : 512)) TMS = 64;
  {  // experimentation hook: DD_FORCE_TILE=128x128|128x64|64x64
    static const char* force = getenv("DD_FORCE_TILE");
    if (force && (M > 64 || force[strlen(force) - 1] == '!')) {   // ("128x128!": also for few rows)
      if (!strncmp(force, "128x128", 7)) { TMS = 128; TNS = 128; }
      else if (!strncmp(force, "128x64", 6)) { TMS = 128; TNS = 64; }
      else if (!strncmp(force, "64x64", 5)) { TMS = 64; TNS = 64; }
#ifdef DD_EXP_BIG
      else if (!strcmp(force, "256x128") && gemm_mode() == 6 && M > 128 && N > 64) { TMS = 256; TNS = 128; }
#endif
    }
  }
  const int tm = dd_ceil_div(M, TMS), tn = dd_ceil_div(N, TNS);
  const long MN = (long)M * N;
  int S = pick_split((long)tm * tn, K, MN, ws ? ws_bytes : 0);
  int kps = ((dd_ceil_div(K > 0 ? K : 1, S) + BKBIG - 1) / BKBIG) * BKBIG;
  S = K > 0 ? dd_ceil_div(K, kps) : 1;
  EpiMat ep{C, ldc, bias, alpha, beta, M, N, S > 1 ? ws : nullptr};
  dim3 grid(tm * tn, 1, S);
#ifdef DD_EXP_BIG
  if (TMS == 256)
    launch_tile<256, 128, AKC, BKC>(grid, st, al, bl, ep, K, kps, tm);
  else
#endif
  if (TMS == 128 && TNS == 128)
    launch_tile<128, 128, AKC, BKC>(grid, st, al, bl, ep, K, kps, tm);
  else if (TMS == 128)
    launch_tile<128, 64, AKC, BKC>(grid, st, al, bl, ep, K, kps, tm);
  else
    launch_tile<64, 64, AKC, BKC>(grid, st, al, bl, ep, K, kps, tm);
  DD_CHECK_LAUNCH(name);
  if (deferred) *deferred = 0;
  if (S > 1 && deferred) {
    *deferred = S;  // the consumer kernel adds the slabs (PreSum)
    return 0;
  }
  if (S > 1) {
    launch_splitk_reduce(ws, S, MN, N, C, ldc, bias, alpha, beta, st);
    DD_CHECK_LAUNCH("dd_gemm_f32(split-k reduce)");
  }
  return 0;
}

}  // namespace
