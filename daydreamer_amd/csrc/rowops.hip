// Row-wise memory-bound kernels: LayerNorm(+ELU) forward/backward, the
// DreamerV2 GRU-cell epilogue forward/backward, and column reductions.
//
// One 64-lane wavefront owns one row: coalesced loads of the (rows, feature)
// tile, butterfly shuffles for the per-row reductions, no LDS traffic on the
// critical path.  Reference math: Norm nets.py:585-602 (population variance,
// eps 1e-3), ELU alpha 1, RSSM._gru nets.py:149-160.
#include "dd_common.h"
#include <type_traits>
#include "../../include/daydreamer_hip.h"

namespace {

constexpr float LN_EPS = 1e-3f;
constexpr int WPB = 4;  // waves (rows in flight) per 256-thread block

__device__ __forceinline__ float elu_(float y) { return y > 0.f ? y : expm1f(y); }

// ---- LayerNorm + activation forward --------------------------------------
// NPL = cached elements per lane (C <= 64*NPL); NPL = 0 re-reads the row.
template <int NPL>
__global__ void __launch_bounds__(256)
k_ln_act_fwd(const float* __restrict__ z, long ldz, const float* __restrict__ gamma,
             const float* __restrict__ beta, float* __restrict__ out, long ldo,
             float* __restrict__ stats, long lds, int rows, int C, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * WPB + wave; row < rows; row += (long)gridDim.x * WPB) {
    const float* zr = z + row * ldz;
    float x[NPL > 0 ? NPL : 1];
    float s = 0.f;
    if (NPL > 0) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        int c = lane + 64 * i;
        x[i] = c < C ? zr[c] : 0.f;
        s += x[i];
      }
    } else {
      for (int c = lane; c < C; c += 64) s += zr[c];
    }
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    if (NPL > 0) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        int c = lane + 64 * i;
        float d = c < C ? x[i] - mean : 0.f;
        v += d * d;
      }
    } else {
      for (int c = lane; c < C; c += 64) { float d = zr[c] - mean; v += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + LN_EPS);
    float* orow = out + row * ldo;
    if (NPL > 0) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        int c = lane + 64 * i;
        if (c < C) {
          float y = (x[i] - mean) * rstd * gamma[c] + beta[c];
          orow[c] = act ? elu_(y) : y;
        }
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        float y = (zr[c] - mean) * rstd * gamma[c] + beta[c];
        orow[c] = act ? elu_(y) : y;
      }
    }
    if (lane == 0) { stats[row * lds] = mean; stats[row * lds + 1] = rstd; }
  }
}

// ---- LayerNorm + activation backward (data gradient, optional fused
// per-block parameter-gradient partials) ------------------------------------
template <int NPL>
__global__ void __launch_bounds__(256)
k_ln_act_bwd(const float* __restrict__ dout, long ldd, const float* __restrict__ z, long ldz,
             const float* __restrict__ out, long ldo, const float* __restrict__ stats, long lds,
             const float* __restrict__ gamma, const float* __restrict__ beta,
             float* __restrict__ dz, long lddz,
             float* __restrict__ partials, int rows, int C, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[NPL > 0 ? NPL : 1], pb[NPL > 0 ? NPL : 1];
#pragma unroll
  for (int i = 0; i < (NPL > 0 ? NPL : 1); ++i) { pg[i] = 0.f; pb[i] = 0.f; }
  for (long row = (long)blockIdx.x * WPB + wave; row < rows; row += (long)gridDim.x * WPB) {
    const float mean = stats[row * lds], rstd = stats[row * lds + 1];
    const float* dr = dout + row * ldd;
    const float* zr = z + row * ldz;
    const float* orow = out ? out + row * ldo : nullptr;
    float* dzr = dz + row * lddz;
    // ELU'(y) = 1 (y > 0) or elu(y) + 1: from the stored activation, or - out == NULL - from
    // y = LN(z) * gamma + beta recomputed with the forward kernel's own expression (same bits)
    auto dact = [&](int c) {
      const float o = orow ? orow[c] : elu_((zr[c] - mean) * rstd * gamma[c] + beta[c]);
      return o > 0.f ? 1.f : o + 1.f;
    };
    if (NPL > 0) {
      float g[NPL > 0 ? NPL : 1], xh[NPL > 0 ? NPL : 1];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        int c = lane + 64 * i;
        if (c < C) {
          float dy = dr[c];
          if (act) dy *= dact(c);
          xh[i] = (zr[c] - mean) * rstd;
          g[i] = dy * gamma[c];
          pg[i] += dy * xh[i];
          pb[i] += dy;
          s1 += g[i];
          s2 += g[i] * xh[i];
        } else { g[i] = 0.f; xh[i] = 0.f; }
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        int c = lane + 64 * i;
        if (c < C) dzr[c] = rstd * (g[i] - s1 - xh[i] * s2);
      }
    } else {
      float s1 = 0.f, s2 = 0.f;
      for (int c = lane; c < C; c += 64) {
        float dy = dr[c];
        if (act) dy *= dact(c);
        float xh = (zr[c] - mean) * rstd;
        float g = dy * gamma[c];
        s1 += g; s2 += g * xh;
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
      for (int c = lane; c < C; c += 64) {
        float dy = dr[c];
        if (act) dy *= dact(c);
        float xh = (zr[c] - mean) * rstd;
        dzr[c] = rstd * (dy * gamma[c] - s1 - xh * s2);
      }
    }
  }
  if (NPL > 0 && partials) {
    __shared__ float sh[WPB][2][64 * (NPL > 0 ? NPL : 1)];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      sh[wave][0][lane + 64 * i] = pg[i];
      sh[wave][1][lane + 64 * i] = pb[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WPB; ++w) { a += sh[w][0][c]; b += sh[w][1][c]; }
      partials[((long)blockIdx.x * 2) * C + c] = a;
      partials[((long)blockIdx.x * 2 + 1) * C + c] = b;
    }
  }
}


// ---- vectorised LayerNorm kernels: LPR lanes own one row, V float4 per lane,
// 64/LPR rows per wave (C = 64 rows are 16 lanes wide: 4 rows per wavefront).
template <int LPR>
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR, int V>
__global__ void __launch_bounds__(256)
k_ln_act_fwd_v(float* __restrict__ z, long ldz, const float* __restrict__ gamma,
               const float* __restrict__ beta, float* __restrict__ out, long ldo,
               float* __restrict__ stats, long lds, int rows, int C, int act, PreSum ps,
               const float* __restrict__ head_w = nullptr, const float* __restrict__ head_b = nullptr,
               float* __restrict__ head_out = nullptr) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  for (long row0 = ((long)blockIdx.x * WPB + wave) * RPW; row0 < rows;
       row0 += (long)gridDim.x * WPB * RPW) {
    const long row = row0 + sub;
    const bool live = row < rows;
    float4 x[V];
    float s = 0.f;
    float hd = 0.f;   // head_w: the one-unit output layer behind this layer, out . w (+ b), folded in
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = (l + LPR * i) * 4;
      x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && c < C) {
        float4* zp = reinterpret_cast<float4*>(z + row * ldz + c);
        if (ps.S) {
          x[i] = presum4(ps, row, c, ps.beta != 0.f ? *zp : x[i]);
          *zp = x[i];
        } else {
          x[i] = *zp;
        }
      }
      s += x[i].x + x[i].y + x[i].z + x[i].w;
    }
    const float mean = grp_sum<LPR>(s) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = (l + LPR * i) * 4;
      if (c < C) {
        float a = x[i].x - mean, b = x[i].y - mean, d = x[i].z - mean, e = x[i].w - mean;
        v += a * a + b * b + d * d + e * e;
      }
    }
    const float rstd = rsqrtf(grp_sum<LPR>(v) / (float)C + LN_EPS);
    if (live) {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        int c = (l + LPR * i) * 4;
        if (c < C) {
          float4 g = *reinterpret_cast<const float4*>(gamma + c);
          float4 bb = *reinterpret_cast<const float4*>(beta + c);
          float4 y;
          y.x = (x[i].x - mean) * rstd * g.x + bb.x;
          y.y = (x[i].y - mean) * rstd * g.y + bb.y;
          y.z = (x[i].z - mean) * rstd * g.z + bb.z;
          y.w = (x[i].w - mean) * rstd * g.w + bb.w;
          if (act) { y.x = elu_(y.x); y.y = elu_(y.y); y.z = elu_(y.z); y.w = elu_(y.w); }
          *reinterpret_cast<float4*>(out + row * ldo + c) = y;
          if (head_w) {
            const float4 hw = *reinterpret_cast<const float4*>(head_w + c);
            hd += (y.x * hw.x + y.y * hw.y) + (y.z * hw.z + y.w * hw.w);
          }
        }
      }
      if (l == 0) { stats[row * lds] = mean; stats[row * lds + 1] = rstd; }
    }
    if (head_w) {   // (all lanes of the row group take part in the shuffles)
      hd = grp_sum<LPR>(hd);
      if (live && l == 0) head_out[row] = hd + (head_b ? head_b[0] : 0.f);
    }
  }
}

// partials layout: [gridDim.x][3][C] = (dgamma, dbeta, column sum of dz)
template <int LPR, int V>
__global__ void __launch_bounds__(256)
k_ln_act_bwd_v(float* __restrict__ dout, long ldd, const float* __restrict__ z, long ldz,
               const float* __restrict__ out, long ldo, const float* __restrict__ stats, long lds,
               const float* __restrict__ gamma, const float* __restrict__ beta,
               float* __restrict__ dz, long lddz,
               float* __restrict__ partials, int rows, int C, int act, PreSum ps,
               const float* __restrict__ head_dy = nullptr, const float* __restrict__ head_w = nullptr) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  float4 pg[V], pb[V], pz[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    pg[i] = make_float4(0.f, 0.f, 0.f, 0.f); pb[i] = pg[i]; pz[i] = pg[i];
  }
  for (long row0 = ((long)blockIdx.x * WPB + wave) * RPW; row0 < rows;
       row0 += (long)gridDim.x * WPB * RPW) {
    const long row = row0 + sub;
    const bool live = row < rows;
    const float mean = live ? stats[row * lds] : 0.f, rstd = live ? stats[row * lds + 1] : 0.f;
    float4 g[V], xh[V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = (l + LPR * i) * 4;
      g[i] = make_float4(0.f, 0.f, 0.f, 0.f); xh[i] = g[i];
      if (live && c < C) {
        float4* dyp = reinterpret_cast<float4*>(dout + row * ldd + c);
        float4 dy;
        if (head_dy) {   // the gradient of a one-unit output layer: dy[row] * w, never stored
          const float hg = head_dy[row];
          const float4 hw = *reinterpret_cast<const float4*>(head_w + c);
          dy = make_float4(hg * hw.x, hg * hw.y, hg * hw.z, hg * hw.w);
        } else if (ps.S) {
          dy = presum4(ps, row, c, ps.beta != 0.f ? *dyp : make_float4(0.f, 0.f, 0.f, 0.f));
          *dyp = dy;  // the bulk parameter-gradient pass reads it later
        } else {
          dy = *dyp;
        }
        float4 zz = *reinterpret_cast<const float4*>(z + row * ldz + c);
        float4 gm = *reinterpret_cast<const float4*>(gamma + c);
        if (act) {
          float4 o;
          if (out) {
            o = *reinterpret_cast<const float4*>(out + row * ldo + c);
          } else {   // recomputed with the forward kernel's expression: the same bits, one tensor less to read
            const float4 bt = *reinterpret_cast<const float4*>(beta + c);
            o.x = elu_((zz.x - mean) * rstd * gm.x + bt.x); o.y = elu_((zz.y - mean) * rstd * gm.y + bt.y);
            o.z = elu_((zz.z - mean) * rstd * gm.z + bt.z); o.w = elu_((zz.w - mean) * rstd * gm.w + bt.w);
          }
          dy.x *= (o.x > 0.f ? 1.f : o.x + 1.f); dy.y *= (o.y > 0.f ? 1.f : o.y + 1.f);
          dy.z *= (o.z > 0.f ? 1.f : o.z + 1.f); dy.w *= (o.w > 0.f ? 1.f : o.w + 1.f);
        }
        xh[i].x = (zz.x - mean) * rstd; xh[i].y = (zz.y - mean) * rstd;
        xh[i].z = (zz.z - mean) * rstd; xh[i].w = (zz.w - mean) * rstd;
        g[i].x = dy.x * gm.x; g[i].y = dy.y * gm.y; g[i].z = dy.z * gm.z; g[i].w = dy.w * gm.w;
        pg[i].x += dy.x * xh[i].x; pg[i].y += dy.y * xh[i].y; pg[i].z += dy.z * xh[i].z; pg[i].w += dy.w * xh[i].w;
        pb[i].x += dy.x; pb[i].y += dy.y; pb[i].z += dy.z; pb[i].w += dy.w;
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      }
    }
    s1 = grp_sum<LPR>(s1) / (float)C;
    s2 = grp_sum<LPR>(s2) / (float)C;
    if (live) {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        int c = (l + LPR * i) * 4;
        if (c < C) {
          float4 r;
          r.x = rstd * (g[i].x - s1 - xh[i].x * s2); r.y = rstd * (g[i].y - s1 - xh[i].y * s2);
          r.z = rstd * (g[i].z - s1 - xh[i].z * s2); r.w = rstd * (g[i].w - s1 - xh[i].w * s2);
          pz[i].x += r.x; pz[i].y += r.y; pz[i].z += r.z; pz[i].w += r.w;
          *reinterpret_cast<float4*>(dz + row * lddz + c) = r;
        }
      }
    }
  }
  if (partials) {
    // combine the WPB*RPW row groups of this block through LDS
    extern __shared__ __attribute__((aligned(16))) float shv[];  // [WPB*RPW][3][C]
    const int slot = wave * RPW + sub;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = (l + LPR * i) * 4;
      if (c < C) {
        *reinterpret_cast<float4*>(&shv[((long)slot * 3 + 0) * C + c]) = pg[i];
        *reinterpret_cast<float4*>(&shv[((long)slot * 3 + 1) * C + c]) = pb[i];
        *reinterpret_cast<float4*>(&shv[((long)slot * 3 + 2) * C + c]) = pz[i];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * C; e += 256) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < WPB * RPW; ++w) a += shv[(long)w * 3 * C + e];
      partials[(long)blockIdx.x * 3 * C + e] = a;
    }
  }
}

// ---- generic LN parameter gradient: thread per column, 4 row lanes per
// block, grid.y row chunks -> partials[grid.y][2][C] -------------------------
__global__ void __launch_bounds__(256)
k_ln_param_grad(const float* __restrict__ dout, long ldd, const float* __restrict__ z, long ldz,
                const float* __restrict__ out, long ldo, const float* __restrict__ stats, long lds,
                float* __restrict__ partials, int rows, int C, int act,
                const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f, b = 0.f;
  if (c < C) {
    for (long row = (long)blockIdx.y * 4 + rl; row < rows; row += (long)gridDim.y * 4) {
      float dy = dout[row * ldd + c];
      if (act) {   // (out == NULL: recomputed as in k_ln_act_bwd)
        const float o = out ? out[row * ldo + c]
                            : elu_((z[row * ldz + c] - stats[row * lds]) * stats[row * lds + 1] * gamma[c] + beta[c]);
        dy *= (o > 0.f ? 1.f : o + 1.f);
      }
      float xh = (z[row * ldz + c] - stats[row * lds]) * stats[row * lds + 1];
      a += dy * xh;
      b += dy;
    }
  }
  __shared__ float sh[4][2][64];
  sh[rl][0][cl] = a; sh[rl][1][cl] = b;
  __syncthreads();
  if (rl == 0 && c < C) {
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { sa += sh[w][0][cl]; sb += sh[w][1][cl]; }
    partials[((long)blockIdx.y * 2) * C + c] = sa;
    partials[((long)blockIdx.y * 2 + 1) * C + c] = sb;
  }
}

// Several outputs in one launch: out_j[w] = beta*out_j[w] + sum_p partials[p*stride + j*W + w]
// (the fused LayerNorm-backward partials are [P][nout][W]).  Block = 16 columns x 16 partial
// lanes, grid (ceil(W/16), nout): the three separate 64x4 passes it replaces took ~15 us
// each, more than the LayerNorm kernel they follow on mid-size problems.
__global__ void __launch_bounds__(256)
k_col_reduce_n(const float* __restrict__ partials, int P, long stride, int W,
               float* o0, float* o1, float* o2, float beta) {
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int w = blockIdx.x * 16 + cl, j = blockIdx.y;
  float* out = j == 0 ? o0 : (j == 1 ? o1 : o2);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (w < W) {
    const float* q = partials + (long)j * W + w;
    int p = pl;
    for (; p + 48 < P; p += 64) {   // four partial rows per lane in flight
      s0 += q[(long)p * stride]; s1 += q[(long)(p + 16) * stride];
      s2 += q[(long)(p + 32) * stride]; s3 += q[(long)(p + 48) * stride];
    }
    for (; p < P; p += 16) s0 += q[(long)p * stride];
  }
  __shared__ float sh[16][17];
  sh[pl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0 && w < W) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][cl];
    out[w] = (beta != 0.f ? beta * out[w] : 0.f) + t;
  }
}

// First level of a two-level column reduce: block (16 columns x 16 partial lanes), grid
// (ceil(W/16), nout, chunks): inter[(z * nout + j) * W + w] = sum of the partial rows
// [z * rpc, (z + 1) * rpc) of output j.  With thousands of partial rows (the big convolutional
// LayerNorms use 8192 blocks) the one-level pass is a 12-block launch whose threads walk 512 rows
// each: 117 us; two levels are two launches of a few us.
__global__ void __launch_bounds__(256)
k_col_reduce_chunks(const float* __restrict__ partials, int P, long stride, int W, int rpc,
                    float* __restrict__ inter) {
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int w = blockIdx.x * 16 + cl, j = blockIdx.y, nout = gridDim.y;
  const int p0 = blockIdx.z * rpc, p1 = min(P, p0 + rpc);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (w < W) {
    const float* q = partials + (long)j * W + w;
    int p = p0 + pl;
    for (; p + 48 < p1; p += 64) {   // four rows per lane in flight
      s[0] += q[(long)p * stride]; s[1] += q[(long)(p + 16) * stride];
      s[2] += q[(long)(p + 32) * stride]; s[3] += q[(long)(p + 48) * stride];
    }
    for (; p < p1; p += 16) s[0] += q[(long)p * stride];
  }
  __shared__ float sh[16][17];
  sh[pl][cl] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (pl == 0 && w < W) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][cl];
    inter[((long)blockIdx.z * nout + j) * W + w] = t;
  }
}

// column sums: partials[grid.y][C]
__global__ void __launch_bounds__(256)
k_col_sum(const float* __restrict__ x, long ldx, float* __restrict__ partials, long rows, int C) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < C)
    for (long row = (long)blockIdx.y * 4 + rl; row < rows; row += (long)gridDim.y * 4)
      a += x[row * ldx + c];
  __shared__ float sh[4][64];
  sh[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < C)
    partials[(long)blockIdx.y * C + c] = sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl];
}

// ---- GRU cell epilogue -----------------------------------------------------
// z3 [rows, 3D] = concat[deter, x] @ W; LayerNorm over the whole 3D vector,
// then reset = sig(r); cand = tanh(reset * c); update = sig(u - 1);
// h' = update * cand + (1 - update) * h.
__global__ void __launch_bounds__(256)
k_gru_fwd(float* __restrict__ z3, long ldz, const float* __restrict__ gamma,
          const float* __restrict__ beta, const float* __restrict__ h, long ldh,
          float* __restrict__ hn, long ldn, float* __restrict__ stats, long lds, int rows, int D,
          PreSum ps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = 3 * D;
  for (long row = (long)blockIdx.x * WPB + wave; row < rows; row += (long)gridDim.x * WPB) {
    float* zr = z3 + row * ldz;
    float s = 0.f;
    if (ps.S) {
      // (D % 64 == 0, checked by the host: every later read of zr[c] is by the lane that
      // wrote it here)
      // four columns per round, all their slab loads in flight together (a store between
      // the loads of successive columns would serialise one memory latency per column)
      for (int c0 = lane; c0 < C; c0 += 256) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* q = ps.p + row * ps.N + c0;
        for (int z = 0; z < ps.S; ++z)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + 64 * j < C) v[j] += q[z * ps.MN + 64 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = c0 + 64 * j;
          if (c < C) {
            if (ps.beta != 0.f) v[j] += ps.beta * zr[c];
            zr[c] = v[j];
            s += v[j];
          }
        }
      }
    } else {
      for (int c = lane; c < C; c += 64) s += zr[c];
    }
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { float d = zr[c] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + LN_EPS);
    for (int j = lane; j < D; j += 64) {
      float yr = (zr[j] - mean) * rstd * gamma[j] + beta[j];
      float yc = (zr[D + j] - mean) * rstd * gamma[D + j] + beta[D + j];
      float yu = (zr[2 * D + j] - mean) * rstd * gamma[2 * D + j] + beta[2 * D + j];
      float r = sigmoidf_(yr);
      float cand = tanhf(r * yc);
      float u = sigmoidf_(yu - 1.f);
      float hp = h[row * ldh + j];
      hn[row * ldn + j] = u * cand + (1.f - u) * hp;
    }
    if (lane == 0) { stats[row * lds] = mean; stats[row * lds + 1] = rstd; }
  }
}

// Backward: given dh' -> dz3 (through the LayerNorm), dh (direct path,
// (1-update)*dh'), and dy3 (gradient at the LayerNorm output, for the bulk
// parameter-gradient pass).
__global__ void __launch_bounds__(256)
k_gru_bwd(const float* __restrict__ dhn, long lddn, const float* __restrict__ z3, long ldz,
          const float* __restrict__ stats, long lds, const float* __restrict__ gamma,
          const float* __restrict__ beta, const float* __restrict__ h, long ldh,
          float* __restrict__ dz3, long lddz, float* __restrict__ dh, long lddh,
          float* __restrict__ dy3, long lddy, float* __restrict__ zx, long ldzx, int U,
          int rows, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = 3 * D;
  for (long row = (long)blockIdx.x * WPB + wave; row < rows; row += (long)gridDim.x * WPB) {
    const float* zr = z3 + row * ldz;
    const float mean = stats[row * lds], rstd = stats[row * lds + 1];
    float* dyr = dy3 + row * lddy;
    if (zx) for (int j = lane; j < U; j += 64) zx[row * ldzx + j] = 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < D; j += 64) {
      float xr = (zr[j] - mean) * rstd, xc = (zr[D + j] - mean) * rstd,
            xu = (zr[2 * D + j] - mean) * rstd;
      float yr = xr * gamma[j] + beta[j];
      float yc = xc * gamma[D + j] + beta[D + j];
      float yu = xu * gamma[2 * D + j] + beta[2 * D + j];
      float r = sigmoidf_(yr);
      float cand = tanhf(r * yc);
      float u = sigmoidf_(yu - 1.f);
      float hp = h[row * ldh + j];
      float d = dhn[row * lddn + j];
      float du = d * (cand - hp);
      float dc = d * u;
      dh[row * lddh + j] = d * (1.f - u);
      float dpre = dc * (1.f - cand * cand);
      float dyc = dpre * r;
      float dyr_ = dpre * yc * r * (1.f - r);
      float dyu = du * u * (1.f - u);
      dyr[j] = dyr_; dyr[D + j] = dyc; dyr[2 * D + j] = dyu;
      float gr = dyr_ * gamma[j], gc = dyc * gamma[D + j], gu = dyu * gamma[2 * D + j];
      s1 += gr + gc + gu;
      s2 += gr * xr + gc * xc + gu * xu;
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    float* dzr = dz3 + row * lddz;
    // dy3 was written by this same lane for the same columns: plain re-read.
    for (int j = lane; j < D; j += 64) {
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        int c = part * D + j;
        float xh = (zr[c] - mean) * rstd;
        dzr[c] = rstd * (dyr[c] * gamma[c] - s1 - xh * s2);
      }
    }
  }
}

// ---- vectorised GRU cell: the whole 3D row lives in registers (D = 256*Q, Q <= 4) ---------
// Lane l holds float4 chunk i at columns 4*(l + 64*i), i < 3Q; chunk i belongs to gate i / Q
// and covers hidden units j = 4*(l + 64*(i % Q)) .. +3, so the three gates of one hidden unit
// sit in the same lane.  One pass over HBM (the scalar kernel reads the row three times with
// 4-byte loads: 29 us average per launch, < 5 % of the HBM rate).
__device__ __forceinline__ float4 f4_norm(float4 x, float mean, float rstd, float4 g, float4 b) {
  return make_float4((x.x - mean) * rstd * g.x + b.x, (x.y - mean) * rstd * g.y + b.y,
                     (x.z - mean) * rstd * g.z + b.z, (x.w - mean) * rstd * g.w + b.w);
}

// sum over the TPR threads that share a row: a wave (TPR = 64) or the whole workgroup (TPR = 256:
// every thread of the block is in the same loop iteration, so the barriers are uniform)
template <int TPR>
__device__ __forceinline__ float row_sum(float v, float* sh) {
  v = wave_sum(v);
  if constexpr (TPR == 64) return v;
  __syncthreads();   // sh free (the previous sum's readers are done)
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <int Q, int TPR>
__global__ void __launch_bounds__(256)
k_gru_fwd_v(float* __restrict__ z3, long ldz, const float* __restrict__ gamma,
            const float* __restrict__ beta, const float* __restrict__ h, long ldh,
            float* __restrict__ hn, long ldn, float* __restrict__ stats, long lds, int rows,
            PreSum ps) {
  // TPR threads per row (64: a wave per row, D = 256 Q; 256: the workgroup per row, D = 1024 Q)
  constexpr int D = 4 * TPR * Q, C = 3 * D, V = 3 * Q, RPB = 256 / TPR;
  __shared__ float sh[4];
  const int lane = threadIdx.x % TPR, wave = threadIdx.x / TPR;
  for (long row = (long)blockIdx.x * RPB + wave; row < rows; row += (long)gridDim.x * RPB) {
    float* zr = z3 + row * ldz;
    float4 x[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = 4 * (lane + TPR * i);
      float4* zp = reinterpret_cast<float4*>(zr + c);
      if (ps.S) {
        x[i] = presum4(ps, row, c, ps.beta != 0.f ? *zp : make_float4(0.f, 0.f, 0.f, 0.f));
        *zp = x[i];  // the backward pass reads the complete pre-norm row
      } else {
        x[i] = *zp;
      }
      s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    }
    const float mean = row_sum<TPR>(s, sh) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float a = x[i].x - mean, b = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
      v += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(row_sum<TPR>(v, sh) / (float)C + LN_EPS);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = 4 * (lane + TPR * q);
      const float4 yr = f4_norm(x[q], mean, rstd, *reinterpret_cast<const float4*>(gamma + j),
                                *reinterpret_cast<const float4*>(beta + j));
      const float4 yc = f4_norm(x[Q + q], mean, rstd, *reinterpret_cast<const float4*>(gamma + D + j),
                                *reinterpret_cast<const float4*>(beta + D + j));
      const float4 yu = f4_norm(x[2 * Q + q], mean, rstd, *reinterpret_cast<const float4*>(gamma + 2 * D + j),
                                *reinterpret_cast<const float4*>(beta + 2 * D + j));
      const float4 hp = *reinterpret_cast<const float4*>(h + row * ldh + j);
      float4 o;
      { float r = sigmoidf_(yr.x), cand = tanhf(r * yc.x), u = sigmoidf_(yu.x - 1.f); o.x = u * cand + (1.f - u) * hp.x; }
      { float r = sigmoidf_(yr.y), cand = tanhf(r * yc.y), u = sigmoidf_(yu.y - 1.f); o.y = u * cand + (1.f - u) * hp.y; }
      { float r = sigmoidf_(yr.z), cand = tanhf(r * yc.z), u = sigmoidf_(yu.z - 1.f); o.z = u * cand + (1.f - u) * hp.z; }
      { float r = sigmoidf_(yr.w), cand = tanhf(r * yc.w), u = sigmoidf_(yu.w - 1.f); o.w = u * cand + (1.f - u) * hp.w; }
      *reinterpret_cast<float4*>(hn + row * ldn + j) = o;
    }
    if (lane == 0) { stats[row * lds] = mean; stats[row * lds + 1] = rstd; }
  }
}

template <int Q, int TPR>
__global__ void __launch_bounds__(256)
k_gru_bwd_v(const float* __restrict__ dhn, long lddn, const float* __restrict__ z3, long ldz,
            const float* __restrict__ stats, long lds, const float* __restrict__ gamma,
            const float* __restrict__ beta, const float* __restrict__ h, long ldh,
            float* __restrict__ dz3, long lddz, float* __restrict__ dh, long lddh,
            float* __restrict__ dy3, long lddy, float* __restrict__ zx, long ldzx, int U, int rows) {
  constexpr int D = 4 * TPR * Q, C = 3 * D, RPB = 256 / TPR;
  __shared__ float sh[4];
  const int lane = threadIdx.x % TPR, wave = threadIdx.x / TPR;
  for (long row = (long)blockIdx.x * RPB + wave; row < rows; row += (long)gridDim.x * RPB) {
    const float* zr = z3 + row * ldz;
    const float mean = stats[row * lds], rstd = stats[row * lds + 1];
    if (zx) for (int j = 4 * lane; j < U; j += 4 * TPR) *reinterpret_cast<float4*>(zx + row * ldzx + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    float xh[3][Q][4], gy[3][Q][4];   // normalised inputs, gamma * dy per gate (r, c, u)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = 4 * (lane + TPR * q);
      float zz[3][4], gm[3][4], bt[3][4], hp[4], d[4], dyo[3][4], dho[4];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        *reinterpret_cast<float4*>(zz[p]) = *reinterpret_cast<const float4*>(zr + p * D + j);
        *reinterpret_cast<float4*>(gm[p]) = *reinterpret_cast<const float4*>(gamma + p * D + j);
        *reinterpret_cast<float4*>(bt[p]) = *reinterpret_cast<const float4*>(beta + p * D + j);
      }
      *reinterpret_cast<float4*>(hp) = *reinterpret_cast<const float4*>(h + row * ldh + j);
      *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(dhn + row * lddn + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xr = (zz[0][e] - mean) * rstd, xc = (zz[1][e] - mean) * rstd, xu = (zz[2][e] - mean) * rstd;
        const float yr = xr * gm[0][e] + bt[0][e], yc = xc * gm[1][e] + bt[1][e], yu = xu * gm[2][e] + bt[2][e];
        const float r = sigmoidf_(yr), cand = tanhf(r * yc), u = sigmoidf_(yu - 1.f);
        const float du = d[e] * (cand - hp[e]), dc = d[e] * u;
        dho[e] = d[e] * (1.f - u);
        const float dpre = dc * (1.f - cand * cand);
        const float dyc = dpre * r, dyr_ = dpre * yc * r * (1.f - r), dyu = du * u * (1.f - u);
        dyo[0][e] = dyr_; dyo[1][e] = dyc; dyo[2][e] = dyu;
        xh[0][q][e] = xr; xh[1][q][e] = xc; xh[2][q][e] = xu;
        gy[0][q][e] = dyr_ * gm[0][e]; gy[1][q][e] = dyc * gm[1][e]; gy[2][q][e] = dyu * gm[2][e];
        s1 += gy[0][q][e] + gy[1][q][e] + gy[2][q][e];
        s2 += gy[0][q][e] * xr + gy[1][q][e] * xc + gy[2][q][e] * xu;
      }
      *reinterpret_cast<float4*>(dh + row * lddh + j) = *reinterpret_cast<float4*>(dho);
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<float4*>(dy3 + row * lddy + p * D + j) = *reinterpret_cast<float4*>(dyo[p]);
    }
    s1 = row_sum<TPR>(s1, sh) / (float)C;
    s2 = row_sum<TPR>(s2, sh) / (float)C;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = 4 * (lane + TPR * q);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (gy[p][q][e] - s1 - xh[p][q][e] * s2);
        *reinterpret_cast<float4*>(dz3 + row * lddz + p * D + j) = *reinterpret_cast<float4*>(o);
      }
    }
  }
}

template <typename F>
int dispatch_npl(int C, F f) {
  if (C <= 64) return f(std::integral_constant<int, 1>());
  if (C <= 128) return f(std::integral_constant<int, 2>());
  if (C <= 256) return f(std::integral_constant<int, 4>());
  if (C <= 512) return f(std::integral_constant<int, 8>());
  if (C <= 768) return f(std::integral_constant<int, 12>());
  if (C <= 1024) return f(std::integral_constant<int, 16>());
  return f(std::integral_constant<int, 0>());
}

template <typename F>
int dispatch_vec(int C, F f) {
  if (C <= 64) return f(std::integral_constant<int, 16>(), std::integral_constant<int, 1>());
  if (C <= 128) return f(std::integral_constant<int, 32>(), std::integral_constant<int, 1>());
  if (C <= 256) return f(std::integral_constant<int, 64>(), std::integral_constant<int, 1>());
  if (C <= 512) return f(std::integral_constant<int, 64>(), std::integral_constant<int, 2>());
  if (C <= 768) return f(std::integral_constant<int, 64>(), std::integral_constant<int, 3>());
  return f(std::integral_constant<int, 64>(), std::integral_constant<int, 4>());
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

inline int row_blocks(long rows, int cap) {
  long b = (rows + WPB - 1) / WPB;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

namespace {
int ln_act_fwd_impl(float* z, long ldz, const float* gamma, const float* beta,
                    float* out, long ldo, float* stats, long lds, int rows, int C, int act,
                    const float* slabs, int n_slabs, float beta_pre, const float* bias_pre,
                    const float* head_w, const float* head_b, float* head_out, void* stream);
}

extern "C" int dd_ln_act_fwd(float* z, long ldz, const float* gamma, const float* beta,
                             float* out, long ldo, float* stats, long lds, int rows, int C, int act,
                             const float* slabs, int n_slabs, float beta_pre, const float* bias_pre,
                             void* stream) {
  return ln_act_fwd_impl(z, ldz, gamma, beta, out, ldo, stats, lds, rows, C, act, slabs, n_slabs, beta_pre,
                         bias_pre, nullptr, nullptr, nullptr, stream);
}

extern "C" int dd_ln_act_fwd_head(float* z, long ldz, const float* gamma, const float* beta,
                                  float* out, long ldo, float* stats, long lds, int rows, int C, int act,
                                  const float* slabs, int n_slabs, float beta_pre, const float* bias_pre,
                                  const float* head_w, const float* head_b, float* head_out, void* stream) {
  DD_REQUIRE(head_w != nullptr && head_out != nullptr, "dd_ln_act_fwd_head: head kernel and output required");
  return ln_act_fwd_impl(z, ldz, gamma, beta, out, ldo, stats, lds, rows, C, act, slabs, n_slabs, beta_pre,
                         bias_pre, head_w, head_b, head_out, stream);
}

namespace {
int ln_act_fwd_impl(float* z, long ldz, const float* gamma, const float* beta,
                    float* out, long ldo, float* stats, long lds, int rows, int C, int act,
                    const float* slabs, int n_slabs, float beta_pre, const float* bias_pre,
                    const float* head_w, const float* head_b, float* head_out, void* stream) {
  if (rows <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = C % 4 == 0 && C <= 1024 && ldz % 4 == 0 && ldo % 4 == 0 && al16(z) && al16(out) &&
                   al16(gamma) && al16(beta) && al16(slabs) && al16(bias_pre) && al16(head_w);
  DD_REQUIRE(vec || head_w == nullptr, "dd_ln_act_fwd_head: needs the vector path (C % 4 == 0, C <= 1024, 16-byte aligned rows)");
  if (n_slabs > 0 && !vec) {  // the scalar kernels take plain input: finish the sum first
    int rc = dd_splitk_finish(slabs, n_slabs, z, ldz, rows, C, beta_pre, bias_pre, stream);
    if (rc) return rc;
    n_slabs = 0;
  }
  PreSum ps{slabs, n_slabs, (long)rows * C, C, beta_pre, bias_pre};
  if (vec) {
    return dispatch_vec(C, [&](auto lpr, auto v) {
      constexpr int LPR = decltype(lpr)::value, V = decltype(v)::value;
      long groups = (rows + (64 / LPR) - 1) / (64 / LPR);
      int blocks = row_blocks(groups, 1 << 20);
      k_ln_act_fwd_v<LPR, V><<<blocks, 256, 0, st>>>(z, ldz, gamma, beta, out, ldo, stats, lds, rows, C, act, ps,
                                                     head_w, head_b, head_out);
      DD_CHECK_LAUNCH("dd_ln_act_fwd");
      return 0;
    });
  }
  int blocks = row_blocks(rows, 1 << 20);
  return dispatch_npl(C, [&](auto npl) {
    k_ln_act_fwd<decltype(npl)::value><<<blocks, 256, 0, st>>>(z, ldz, gamma, beta, out, ldo, stats, lds, rows, C, act);
    DD_CHECK_LAUNCH("dd_ln_act_fwd");
    return 0;
  });
}
}  // namespace

extern "C" int dd_ln_bwd_parts(int rows, int C) {
  // Number of partial rows dd_ln_act_bwd's fused parameter-gradient pass uses.
  if (C > 1024) {
    long chunks = (rows + 63) / 64;
    return (int)(chunks > 256 ? 256 : (chunks < 1 ? 1 : chunks));
  }
  return row_blocks(rows, 256);
}

namespace {
int ln_act_bwd_impl(float* dout, long ldd, const float* z, long ldz,
                    const float* out, long ldo, const float* stats, long lds, const float* gamma,
                    const float* beta_ln,
                    float* dz, long lddz, float* dgamma, float* dbeta, float* dbias_pre,
                    int accumulate, int rows, int C, int act, float* ws, size_t ws_bytes,
                    const float* slabs, int n_slabs, float beta_pre,
                    const float* head_dy, const float* head_w, void* stream);
}

extern "C" int dd_ln_act_bwd(float* dout, long ldd, const float* z, long ldz,
                             const float* out, long ldo, const float* stats, long lds, const float* gamma,
                             const float* beta_ln,
                             float* dz, long lddz, float* dgamma, float* dbeta, float* dbias_pre,
                             int accumulate, int rows, int C, int act, float* ws, size_t ws_bytes,
                             const float* slabs, int n_slabs, float beta_pre, void* stream) {
  return ln_act_bwd_impl(dout, ldd, z, ldz, out, ldo, stats, lds, gamma, beta_ln, dz, lddz, dgamma, dbeta,
                         dbias_pre, accumulate, rows, C, act, ws, ws_bytes, slabs, n_slabs, beta_pre,
                         nullptr, nullptr, stream);
}

extern "C" int dd_ln_act_bwd_head(const float* head_dy, const float* head_w, const float* z, long ldz,
                                  const float* out, long ldo, const float* stats, long lds, const float* gamma,
                                  const float* beta_ln,
                                  float* dz, long lddz, float* dgamma, float* dbeta, float* dbias_pre,
                                  int accumulate, int rows, int C, int act, float* ws, size_t ws_bytes,
                                  void* stream) {
  DD_REQUIRE(head_dy != nullptr && head_w != nullptr, "dd_ln_act_bwd_head: head gradient and kernel required");
  // (`dout` is never touched: any 16-byte aligned address with a valid leading dimension will do)
  return ln_act_bwd_impl(dz, lddz, z, ldz, out, ldo, stats, lds, gamma, beta_ln, dz, lddz, dgamma, dbeta,
                         dbias_pre, accumulate, rows, C, act, ws, ws_bytes, nullptr, 0, 0.f, head_dy, head_w, stream);
}

namespace {
int ln_act_bwd_impl(float* dout, long ldd, const float* z, long ldz,
                    const float* out, long ldo, const float* stats, long lds, const float* gamma,
                    const float* beta_ln,
                    float* dz, long lddz, float* dgamma, float* dbeta, float* dbias_pre,
                    int accumulate, int rows, int C, int act, float* ws, size_t ws_bytes,
                    const float* slabs, int n_slabs, float beta_pre,
                    const float* head_dy, const float* head_w, void* stream) {
  if (rows <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool want = dgamma != nullptr;
  const float b = accumulate ? 1.f : 0.f;
  DD_REQUIRE(!act || out != nullptr || beta_ln != nullptr,
             "dd_ln_act_bwd: act without `out` needs the LayerNorm offset to recompute it");
  if (!out) ldo = 4;
  const bool vec = C % 4 == 0 && C <= 1024 && ldd % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 &&
                   lddz % 4 == 0 && al16(dout) && al16(z) && al16(out) && al16(dz) && al16(gamma) &&
                   al16(beta_ln) && al16(slabs) && al16(head_w);
  DD_REQUIRE(vec || head_dy == nullptr, "dd_ln_act_bwd_head: needs the vector path (C % 4 == 0, C <= 1024, 16-byte aligned rows)");
  // the deferred sum shares the workspace with the parameter-gradient partials: only
  // the parameter-free vector path consumes it in place
  if (n_slabs > 0 && (!vec || want)) {
    int rc = dd_splitk_finish(slabs, n_slabs, dout, ldd, rows, C, beta_pre, nullptr, stream);
    if (rc) return rc;
    n_slabs = 0;
  }
  PreSum ps{slabs, n_slabs, (long)rows * C, C, beta_pre, nullptr};
  if (vec) {
    return dispatch_vec(C, [&](auto lpr, auto v) {
      constexpr int LPR = decltype(lpr)::value, V = decltype(v)::value;
      constexpr int RPW = 64 / LPR;
      long groups = (rows + RPW - 1) / RPW;
      // (one 256-block wave of the grid leaves 4 waves per CU: too few loads in flight for
      // the big convolutional LayerNorms, 1.7 TB/s; 8 blocks per CU reach the HBM rate)
      // (enough blocks to keep a memory-bound kernel's loads in flight - 256 blocks ran the
      // big convolutional LayerNorms at 1.7 TB/s - but few enough partial rows for the
      // reduce pass; measured: 512 / 8192 beat proportional grids)
      int blocks = want ? row_blocks(groups, groups > 200000 ? 8192 : 512) : row_blocks(groups, 1 << 20);
      size_t shmem = want ? (size_t)WPB * RPW * 3 * C * sizeof(float) : 0;
      if (want) DD_REQUIRE(ws && (size_t)blocks * 3 * C * sizeof(float) <= ws_bytes, "dd_ln_act_bwd: workspace too small");
      k_ln_act_bwd_v<LPR, V><<<blocks, 256, shmem, st>>>(
          dout, ldd, z, ldz, out, ldo, stats, lds, gamma, beta_ln, dz, lddz, want ? ws : nullptr, rows, C, act, ps,
          head_dy, head_w);
      DD_CHECK_LAUNCH("dd_ln_act_bwd");
      if (want) {
        const int nout = dbias_pre ? 3 : 2;
        // (many partial rows: two levels, chunks of 128 rows; the intermediate rows follow the
        // partials in the workspace)
        const int rpc = 128, chunks = (blocks + rpc - 1) / rpc;
        const size_t need = ((size_t)blocks * 3 + (size_t)chunks * 3) * C * sizeof(float);
        if (blocks >= 1024 && need <= ws_bytes) {
          float* inter = ws + (size_t)blocks * 3 * C;
          k_col_reduce_chunks<<<dim3((C + 15) / 16, nout, chunks), 256, 0, st>>>(ws, blocks, 3L * C, C, rpc, inter);
          DD_CHECK_LAUNCH("dd_ln_act_bwd(reduce level 1)");
          k_col_reduce_n<<<dim3((C + 15) / 16, nout), 256, 0, st>>>(
              inter, chunks, (long)nout * C, C, dgamma, dbeta, dbias_pre, b);
        } else {
          k_col_reduce_n<<<dim3((C + 15) / 16, nout), 256, 0, st>>>(
              ws, blocks, 3L * C, C, dgamma, dbeta, dbias_pre, b);
        }
        DD_CHECK_LAUNCH("dd_ln_act_bwd(reduce)");
      }
      return 0;
    });
  }
  const int parts = dd_ln_bwd_parts(rows, C);
  if (want) DD_REQUIRE(ws && (size_t)parts * 2 * C * sizeof(float) <= ws_bytes, "dd_ln_act_bwd: workspace too small");
  const bool fused = want && C <= 1024;
  int blocks = fused ? parts : row_blocks(rows, 1 << 20);
  int rc = dispatch_npl(C, [&](auto npl) {
    k_ln_act_bwd<decltype(npl)::value><<<blocks, 256, 0, st>>>(
        dout, ldd, z, ldz, out, ldo, stats, lds, gamma, beta_ln, dz, lddz, fused ? ws : nullptr, rows, C, act);
    DD_CHECK_LAUNCH("dd_ln_act_bwd");
    return 0;
  });
  if (rc || !want) return rc;
  if (!fused) {
    dim3 grid((C + 63) / 64, parts);
    k_ln_param_grad<<<grid, 256, 0, st>>>(dout, ldd, z, ldz, out, ldo, stats, lds, ws, rows, C, act, gamma, beta_ln);
    DD_CHECK_LAUNCH("dd_ln_act_bwd(param grad)");
  }
  // partials are [parts][2][C]: gamma rows at stride 2C from ws, beta rows from ws + C.
  k_col_reduce_n<<<dim3((C + 15) / 16, 2), 256, 0, st>>>(ws, parts, 2L * C, C, dgamma, dbeta, nullptr, b);
  DD_CHECK_LAUNCH("dd_ln_act_bwd(reduce)");
  if (dbias_pre) return dd_col_sum(dz, lddz, dbias_pre, b, rows, C, ws, ws_bytes, stream);
  return 0;
}
}  // namespace

// partial rows [rows][3][C] (dgamma, dbeta, column sum of dz: the layout of k_ln_act_bwd_v) ->
// dgamma, dbeta, dbias (+ b * old); used by the fused image-side filter gradient (conv_wgrad.hip)
int dd_ln_partials_reduce(const float* partials, int rows, int C, float* dgamma, float* dbeta, float* dbias,
                          float b, hipStream_t st) {
  k_col_reduce_n<<<dim3((C + 15) / 16, 3), 256, 0, st>>>(partials, rows, 3L * C, C, dgamma, dbeta, dbias, b);
  DD_CHECK_LAUNCH("dd_ln_partials_reduce");
  return 0;
}

// Parameter gradients only (bulk pass after a scan): dgamma/dbeta from stored
// dout / out / z / stats of all steps.
extern "C" int dd_ln_param_grad(const float* dout, long ldd, const float* z, long ldz,
                                const float* out, long ldo, const float* stats, long lds,
                                float* dgamma, float* dbeta, int accumulate, int rows, int C,
                                int act, float* ws, size_t ws_bytes, void* stream) {
  if (rows <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  long chunks = (rows + 63) / 64;
  const int parts = (int)(chunks > 256 ? 256 : (chunks < 1 ? 1 : chunks));
  DD_REQUIRE(ws && (size_t)parts * 2 * C * sizeof(float) <= ws_bytes, "dd_ln_param_grad: workspace too small");
  DD_REQUIRE(!act || out != nullptr, "dd_ln_param_grad: act needs the stored activation");
  dim3 grid((C + 63) / 64, parts);
  k_ln_param_grad<<<grid, 256, 0, st>>>(dout, ldd, z, ldz, out, ldo, stats, lds, ws, rows, C, act);
  DD_CHECK_LAUNCH("dd_ln_param_grad");
  const float b = accumulate ? 1.f : 0.f;
  k_col_reduce_n<<<dim3((C + 15) / 16, 2), 256, 0, st>>>(ws, parts, 2L * C, C, dgamma, dbeta, nullptr, b);
  DD_CHECK_LAUNCH("dd_ln_param_grad(reduce)");
  return 0;
}

extern "C" int dd_gru_cell_fwd(float* z3, long ldz, const float* gamma, const float* beta,
                               const float* h, long ldh, float* hn, long ldn, float* stats, long lds,
                               int rows, int D, const float* slabs, int n_slabs, float beta_pre,
                               void* stream) {
  if (rows <= 0) return 0;
  if (n_slabs > 0 && D % 64 != 0) {
    int rc = dd_splitk_finish(slabs, n_slabs, z3, ldz, rows, 3 * D, beta_pre, nullptr, stream);
    if (rc) return rc;
    n_slabs = 0;
  }
  PreSum ps{slabs, n_slabs, (long)rows * 3 * D, 3 * D, beta_pre, nullptr};
  const bool vec = (D == 256 || D == 512 || D == 1024 || D == 2048 || D == 4096) && ldz % 4 == 0 && ldh % 4 == 0 && ldn % 4 == 0 &&
                   al16(z3) && al16(h) && al16(hn) && al16(gamma) && al16(beta) && al16(slabs);
  const int blocks = row_blocks(rows, 1 << 20);
  hipStream_t st = (hipStream_t)stream;
  // (D >= 2048: the row lives in the registers of a whole workgroup - a1_scaled's deter 4096; the
  // scalar kernel read such a row three times with 4-byte loads: 550 us for 2 048 rows, 0.5 TB/s)
  const int blocks1 = rows < (1 << 20) ? rows : (1 << 20);
  if (vec && D == 256) k_gru_fwd_v<1, 64><<<blocks, 256, 0, st>>>(z3, ldz, gamma, beta, h, ldh, hn, ldn, stats, lds, rows, ps);
  else if (vec && D == 512) k_gru_fwd_v<2, 64><<<blocks, 256, 0, st>>>(z3, ldz, gamma, beta, h, ldh, hn, ldn, stats, lds, rows, ps);
  else if (vec && D == 1024) k_gru_fwd_v<4, 64><<<blocks, 256, 0, st>>>(z3, ldz, gamma, beta, h, ldh, hn, ldn, stats, lds, rows, ps);
  else if (vec && D == 2048) k_gru_fwd_v<2, 256><<<blocks1, 256, 0, st>>>(z3, ldz, gamma, beta, h, ldh, hn, ldn, stats, lds, rows, ps);
  else if (vec) k_gru_fwd_v<4, 256><<<blocks1, 256, 0, st>>>(z3, ldz, gamma, beta, h, ldh, hn, ldn, stats, lds, rows, ps);
  else k_gru_fwd<<<blocks, 256, 0, st>>>(z3, ldz, gamma, beta, h, ldh, hn, ldn, stats, lds, rows, D, ps);
  DD_CHECK_LAUNCH("dd_gru_cell_fwd");
  return 0;
}

extern "C" int dd_gru_cell_bwd(const float* dhn, long lddn, const float* z3, long ldz,
                               const float* stats, long lds, const float* gamma, const float* beta,
                               const float* h, long ldh, float* dz3, long lddz,
                               float* dh, long lddh, float* dy3, long lddy,
                               float* zx, long ldzx, int U, int rows, int D, void* stream) {
  if (rows <= 0) return 0;
  const bool vec = (D == 256 || D == 512 || D == 1024 || D == 2048 || D == 4096) && lddn % 4 == 0 && ldz % 4 == 0 && ldh % 4 == 0 &&
                   lddz % 4 == 0 && lddh % 4 == 0 && lddy % 4 == 0 && (zx == nullptr || (ldzx % 4 == 0 && U % 4 == 0 && al16(zx))) &&
                   al16(dhn) && al16(z3) && al16(h) && al16(dz3) && al16(dh) && al16(dy3) && al16(gamma) && al16(beta);
  const int blocks = row_blocks(rows, 1 << 20);
  hipStream_t st = (hipStream_t)stream;
  const int blocks1 = rows < (1 << 20) ? rows : (1 << 20);   // (a workgroup per row: D >= 2048)
  if (vec && D == 256) k_gru_bwd_v<1, 64><<<blocks, 256, 0, st>>>(dhn, lddn, z3, ldz, stats, lds, gamma, beta, h, ldh, dz3, lddz, dh, lddh, dy3, lddy, zx, ldzx, U, rows);
  else if (vec && D == 512) k_gru_bwd_v<2, 64><<<blocks, 256, 0, st>>>(dhn, lddn, z3, ldz, stats, lds, gamma, beta, h, ldh, dz3, lddz, dh, lddh, dy3, lddy, zx, ldzx, U, rows);
  else if (vec && D == 1024) k_gru_bwd_v<4, 64><<<blocks, 256, 0, st>>>(dhn, lddn, z3, ldz, stats, lds, gamma, beta, h, ldh, dz3, lddz, dh, lddh, dy3, lddy, zx, ldzx, U, rows);
  else if (vec && D == 2048) k_gru_bwd_v<2, 256><<<blocks1, 256, 0, st>>>(dhn, lddn, z3, ldz, stats, lds, gamma, beta, h, ldh, dz3, lddz, dh, lddh, dy3, lddy, zx, ldzx, U, rows);
  else if (vec) k_gru_bwd_v<4, 256><<<blocks1, 256, 0, st>>>(dhn, lddn, z3, ldz, stats, lds, gamma, beta, h, ldh, dz3, lddz, dh, lddh, dy3, lddy, zx, ldzx, U, rows);
  else k_gru_bwd<<<blocks, 256, 0, st>>>(dhn, lddn, z3, ldz, stats, lds, gamma, beta, h, ldh, dz3, lddz, dh, lddh, dy3, lddy, zx, ldzx, U, rows, D);
  DD_CHECK_LAUNCH("dd_gru_cell_bwd");
  return 0;
}

// out[c] = beta*out[c] + sum_rows x[row][c]   (bias gradients)
extern "C" int dd_col_sum(const float* x, long ldx, float* out, float beta, long rows, int C,
                          float* ws, size_t ws_bytes, void* stream) {
  if (C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  long chunks = (rows + 255) / 256;
  int parts = (int)(chunks > 512 ? 512 : (chunks < 1 ? 1 : chunks));
  DD_REQUIRE(ws && (size_t)parts * C * sizeof(float) <= ws_bytes, "dd_col_sum: workspace too small");
  dim3 grid((C + 63) / 64, parts);
  k_col_sum<<<grid, 256, 0, st>>>(x, ldx, ws, rows, C);
  DD_CHECK_LAUNCH("dd_col_sum");
  k_col_reduce_n<<<dim3((C + 15) / 16, 1), 256, 0, st>>>(ws, parts, (long)C, C, out, nullptr, nullptr, beta);
  DD_CHECK_LAUNCH("dd_col_sum(reduce)");
  return 0;
}
