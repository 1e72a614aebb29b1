// Learner state kernels: counter-based RNG (Philox4x32-10), the multi-tensor
// optimizer (global-norm clip + weight decay + bias-corrected Adam over one
// flat parameter arena), device-resident scalar controllers (AutoAdapt,
// Normalize), batch statistics, and small data-movement helpers.
//
// Everything that changes from step to step (step counters, bias corrections,
// controller scales) lives in device memory so a captured HIP graph can be
// replayed unchanged.
//
// Reference: Optimizer tfutils.py:180-302, AutoAdapt tfutils.py:414-482,
// Normalize tfutils.py:485-527, RSSM.obs_step reset mask nets.py:100-107.
#include "dd_common.h"
#include "../../include/daydreamer_hip.h"

namespace {

// ---- Philox4x32-10 ----------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Element (o, i, c) of a [outer, inner, cols] tensor draws from
//   counter = (c / 4, global_row, site, step), key = (seed_lo, seed_hi), word c % 4
// with global_row = o * inner_global + inner_offset + i, so the stream does
// not depend on how the batch axis is sharded across GPUs.
// kind 0: uniform [0,1) = (x >> 8) * 2^-24.  kind 1: standard normal
// (Box-Muller on word pairs (0,1) and (2,3)).
__global__ void k_philox(float* __restrict__ out, long outer, long inner, int cols,
                         long inner_global, long inner_offset, uint32_t seed_lo, uint32_t seed_hi,
                         const unsigned long long* __restrict__ step_dev, uint32_t site, int kind) {
  const int cb = (cols + 3) / 4;
  const long total = outer * inner * cb;
  const uint32_t step = (uint32_t)(*step_dev);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    long row = id / cb; int blk = (int)(id - row * cb);
    long o = row / inner, i = row - o * inner;
    long grow = o * inner_global + inner_offset + i;
    uint32_t c[4] = {(uint32_t)blk, (uint32_t)grow, site, step};
    philox4x32(c, seed_lo, seed_hi);
    float v[4];
    if (kind == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
    } else {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        float u1 = ((float)(c[j] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float u2 = (float)(c[j + 1] >> 8) * (1.0f / 16777216.0f);
        float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        v[j] = rad * cs; v[j + 1] = rad * sn;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int col = blk * 4 + j;
      if (col < cols) out[row * cols + col] = v[j];
    }
  }
}

__global__ void k_counter_add(unsigned long long* c, unsigned long long v) { *c += v; }

// ---- statistics --------------------------------------------------------------
// Single block: sums[0]=sum x, sums[1]=sum x^2 (fp64); maxs[0]=max x,
// maxs[1]=max(-x), maxs[2]=max|x|; sums[2]=sum |x|.
__global__ void __launch_bounds__(1024)
k_reduce_stats(const float* __restrict__ x, long n, long stride, double* __restrict__ sums,
               float* __restrict__ maxs) {
  double s = 0.0, q = 0.0, a = 0.0;
  float mx = -INFINITY, mn = -INFINITY, ma = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) {
    float v = x[i * stride];
    s += v; q += (double)v * v; a += fabsf(v);
    mx = fmaxf(mx, v); mn = fmaxf(mn, -v); ma = fmaxf(ma, fabsf(v));
  }
  s = wave_sum_d(s); q = wave_sum_d(q); a = wave_sum_d(a);
  mx = wave_max(mx); mn = wave_max(mn); ma = wave_max(ma);
  __shared__ double shd[3][16];
  __shared__ float shf[3][16];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    shd[0][w] = s; shd[1][w] = q; shd[2][w] = a;
    shf[0][w] = mx; shf[1][w] = mn; shf[2][w] = ma;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0, tq = 0, ta = 0; float tx = -INFINITY, tn = -INFINITY, tm = 0.f;
    for (int i = 0; i < 16; ++i) {
      ts += shd[0][i]; tq += shd[1][i]; ta += shd[2][i];
      tx = fmaxf(tx, shf[0][i]); tn = fmaxf(tn, shf[1][i]); tm = fmaxf(tm, shf[2][i]);
    }
    sums[0] = ts; sums[1] = tq; sums[2] = ta;
    maxs[0] = tx; maxs[1] = tn; maxs[2] = tm;
  }
}

// The same for up to 16 vectors in ONE launch (block b = vector b): the step's metric statistics
// are ~17 independent single-block reductions of 2 500 .. 40 000 floats, each a launch of its
// own latency when issued one by one (13 us each on the step's critical path).
struct StatItems {
  const float* x[16];
  long n[16];
  long stride[16];
  double* sums[16];
  float* maxs[16];
};
__global__ void __launch_bounds__(1024)
k_reduce_stats_multi(StatItems it) {
  const int b = blockIdx.x;
  const float* __restrict__ x = it.x[b];
  const long n = it.n[b], stride = it.stride[b];
  double s = 0.0, q = 0.0, a = 0.0;
  float mx = -INFINITY, mn = -INFINITY, ma = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) {
    float v = x[i * stride];
    s += v; q += (double)v * v; a += fabsf(v);
    mx = fmaxf(mx, v); mn = fmaxf(mn, -v); ma = fmaxf(ma, fabsf(v));
  }
  s = wave_sum_d(s); q = wave_sum_d(q); a = wave_sum_d(a);
  mx = wave_max(mx); mn = wave_max(mn); ma = wave_max(ma);
  __shared__ double shd[3][16];
  __shared__ float shf[3][16];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    shd[0][w] = s; shd[1][w] = q; shd[2][w] = a;
    shf[0][w] = mx; shf[1][w] = mn; shf[2][w] = ma;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0, tq = 0, ta = 0; float tx = -INFINITY, tn = -INFINITY, tm = 0.f;
    for (int i = 0; i < 16; ++i) {
      ts += shd[0][i]; tq += shd[1][i]; ta += shd[2][i];
      tx = fmaxf(tx, shf[0][i]); tn = fmaxf(tn, shf[1][i]); tm = fmaxf(tm, shf[2][i]);
    }
    it.sums[b][0] = ts; it.sums[b][1] = tq; it.sums[b][2] = ta;
    it.maxs[b][0] = tx; it.maxs[b][1] = tn; it.maxs[b][2] = tm;
  }
}

// AutoAdapt 'mult' (impl 0, tfutils.py:460-474) and 'prop' (impl 1, tfutils.py:475-480):
// scale[i] updated from avg_i = sums[i] / count.
__global__ void k_autoadapt(float* __restrict__ scale, const double* __restrict__ sums, int n,
                            double count, float target, float thres, float vel, float lo,
                            float hi, int inverse, int impl) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float avg = (float)(sums[i] / count);
  if (impl == 1) {
    float dir = avg - target;
    if (inverse) dir = -dir;
    scale[i] = fminf(fmaxf(scale[i] + vel * dir, lo), hi);
    return;
  }
  bool below = avg < (1.f / (1.f + thres)) * target;
  bool above = avg > (1.f + thres) * target;
  if (inverse) { bool t = below; below = above; above = t; }
  float s = scale[i];
  float adj = above ? s * (1.f + vel) : (below ? s / (1.f + vel) : s);
  scale[i] = fminf(fmaxf(adj, lo), hi);
}

// Normalize (tfutils.py:498-527): state = {mean, sqrs, step} in fp64.
// The input statistics are those of x; the normaliser sees in_scale * x
// (in_off is not supported upstream of an update).  Writes transform
// (offset, scale) such that y = (in_scale*x - offset) * scale.
// impl: 0 off, 1 mean_std, 2 std.
__global__ void k_normalize_update(double* __restrict__ state, const double* __restrict__ sums,
                                   double count, const float* __restrict__ in_scale_dev,
                                   double decay, double maxv, int impl, int do_update,
                                   float* __restrict__ out_off_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a = in_scale_dev ? (double)*in_scale_dev : 1.0;
  if (do_update) {
    double mean = a * sums[0] / count, sq = a * a * sums[1] / count;
    state[2] += 1.0;
    state[0] = decay * state[0] + (1.0 - decay) * mean;
    state[1] = decay * state[1] + (1.0 - decay) * sq;
  }
  double corr = 1.0 - pow(decay, state[2]);
  double mean = state[0] / corr;
  double var = state[1] / corr - mean * mean;
  double scale;
  if (maxv > 0.0) scale = 1.0 / sqrt(fmax(var, 1.0 / (maxv * maxv)));
  else scale = 1.0 / sqrt(var);
  float off = 0.f, sc = 1.f;
  if (impl == 1) { off = (float)mean; sc = (float)scale; }
  else if (impl == 2) { sc = (float)scale; }
  out_off_scale[0] = off;
  out_off_scale[1] = sc;
}

// dst[i] = a[i] * (b ? b[i] : 1) * c   (tiny scalar glue, n small)
__global__ void k_scalar_mul(float* dst, const float* a, const float* b, float c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = a[i] * (b ? b[i] : 1.f) * c;
}

// y = (accumulate ? y : 0) + alpha * (alpha_dev ? alpha_dev[0] : 1) * x
__global__ void k_axpy(const float* __restrict__ x, float alpha, const float* __restrict__ alpha_dev,
                       float* __restrict__ y, long n, int accumulate) {
  const float a = alpha * (alpha_dev ? alpha_dev[0] : 1.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (accumulate ? y[i] : 0.f) + a * x[i];
}

// tfutils.balance_stats (tfutils.py:395-411) as seven sums: loss*pos, loss*neg, pred*pos,
// (1-pred)*neg, pos, target, mean; pos = target > thres, pred = mean > thres, mean =
// symexp(out) (kind 0) or sigmoid(out) (kind 1).  Two deterministic stages.
__global__ void __launch_bounds__(256)
k_balance_partial(const float* __restrict__ out, const float* __restrict__ target,
                  const float* __restrict__ loss, long n, float thres, int kind,
                  double* __restrict__ partial) {
  double a[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float t = target[i], l = loss[i];
    const float m = kind == 0 ? symexpf_(out[i]) : sigmoidf_(out[i]);
    const float pos = t > thres ? 1.f : 0.f, pr = m > thres ? 1.f : 0.f;
    a[0] += l * pos; a[1] += l * (1.f - pos); a[2] += pr * pos; a[3] += (1.f - pr) * (1.f - pos);
    a[4] += pos; a[5] += t; a[6] += m;
  }
  __shared__ double sh[4][7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double v = wave_sum_d(a[j]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7)
    partial[(long)blockIdx.x * 7 + threadIdx.x] =
        sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ void k_balance_final(const double* __restrict__ partial, int P, double* __restrict__ out7) {
  const int j = threadIdx.x;
  if (j >= 7) return;
  double t = 0.0;
  for (int p = 0; p < P; ++p) t += partial[(long)p * 7 + j];
  out7[j] = t;
}

// ---- optimizer ---------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sumsq_partial(const float* __restrict__ g, long n, double* __restrict__ partial) {
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = g[i];
    s += (double)v * v;
  }
  s = wave_sum_d(s);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// opt_state: [0]=step (as double), [1]=grad norm, [2]=finite flag, [3]=grad scale, [4]=good steps
// (one wave: lane l adds partials l, l+64, ...; fixed butterfly order - a single thread
// walking ~1000 partials took 68 us, three times per step)
// mixed: the reduced-precision mode's loss-scale controller (reference tfutils.py:225-240,
// `self._mixed`): an overflow (non-finite gradient) halves the scale and resets the good-step
// count, 1000 good steps in a row double it, clipped to [1e-4, 1e4]; the update is skipped on
// overflow (k_adam) and the step count does not advance (tfutils.py:255-260).
__global__ void __launch_bounds__(64)
k_norm_finalize(const double* __restrict__ partial, int P, double* __restrict__ st, int mixed) {
  double s = 0.0;
  for (int i = threadIdx.x; i < P; i += 64) s += partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x != 0) return;
  double norm = sqrt(s);
  st[1] = norm;
  bool fin = isfinite(norm);
  st[2] = fin ? 1.0 : 0.0;
  if (fin) st[0] += 1.0;  // tfutils.py:260 (step advances only when applied)
  if (mixed) {
    const double scale = st[3], good = st[4];
    double ns, ng;
    if (!fin) { ns = scale / 2.0; ng = 0.0; }
    else if (good >= 1000.0) { ns = scale * 2.0; ng = 0.0; }
    else { ns = scale; ng = good + 1.0; }
    st[3] = fmin(fmax(ns, 1e-4), 1e4);
    st[4] = ng;
  }
}

// p *= (1 - wd*lr) for i < n_decay, then Adam with bias correction
// (tfutils.py:271-283), gradient scaled by clip / max(norm, clip).
// warmup > 0 (tfutils.py:160-162): lr * clip(step / warmup, 0, 1) with the step count at the time
// of use - the decay runs before the count advances (:254-256 then :260), Adam after it; st[0] is
// the advanced count here (dd_grad_norm advanced it).
__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
       float* __restrict__ v, long n, long n_decay, const double* __restrict__ st,
       float lr, float wd, float eps, float b1, float b2, float clip, float warmup) {
  if (st[2] == 0.0) return;  // non-finite gradient norm: skip, host raises
  const float norm = (float)st[1];
  const float gs = clip > 0.f ? clip / fmaxf(norm, clip) : 1.f;
  const float t = (float)st[0];
  const float c1 = 1.f / (1.f - powf(b1, t)), c2 = 1.f / (1.f - powf(b2, t));
  float lr_wd = lr;
  if (warmup > 0.f) {
    lr_wd = lr * fminf(fmaxf((t - 1.f) / warmup, 0.f), 1.f);
    lr = lr * fminf(fmaxf(t / warmup, 0.f), 1.f);
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = g[i] * gs;
    float pi = p[i];
    if (i < n_decay) pi *= (1.f - wd * lr_wd);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - lr * (mi * c1) / (sqrtf(vi * c2) + eps);
  }
}

// ---- data movement ------------------------------------------------------------
__global__ void k_fill(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_copy2d(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd,
                         long rows, int cols) {
  long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cols; int c = (int)(i - r * cols);
    dst[r * ldd + c] = src[r * lds + c];
  }
}

// out = prev*(1-f) + init*f, f = first[row*fstride]  (init broadcast over rows;
// prev may be null -> treated as zeros)
__global__ void k_reset_mask(const float* __restrict__ prev, long ldp, const float* __restrict__ first,
                             long fstride, const float* __restrict__ init, float* __restrict__ out,
                             long ldo, long rows, int cols) {
  long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cols; int c = (int)(i - r * cols);
    float f = first[r * fstride];
    float pv = prev ? prev[r * ldp + c] : 0.f;
    float iv = init ? init[c] : 0.f;
    out[r * ldo + c] = pv * (1.f - f) + iv * f;
  }
}

// dprev += dout * (1 - f)
__global__ void k_reset_mask_bwd(const float* __restrict__ dout, long ldo, const float* __restrict__ first,
                                 long fstride, float* __restrict__ dprev, long ldp, long rows, int cols) {
  long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cols; int c = (int)(i - r * cols);
    dprev[r * ldp + c] += dout[r * ldo + c] * (1.f - first[r * fstride]);
  }
}

// Two column segments (deter | stoch of the carried state) in one launch.
__global__ void k_reset_mask2(const float* __restrict__ pa, long ldpa, const float* __restrict__ ia,
                              float* __restrict__ oa, long ldoa, int ca,
                              const float* __restrict__ pb, long ldpb, const float* __restrict__ ib,
                              float* __restrict__ ob, long ldob, int cb,
                              const float* __restrict__ first, long fstride, long rows) {
  const int cols = ca + cb;
  long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cols; int c = (int)(i - r * cols);
    float f = first[r * fstride];
    if (c < ca) {
      float pv = pa ? pa[r * ldpa + c] : 0.f, iv = ia ? ia[c] : 0.f;
      oa[r * ldoa + c] = pv * (1.f - f) + iv * f;
    } else {
      c -= ca;
      float pv = pb ? pb[r * ldpb + c] : 0.f, iv = ib ? ib[c] : 0.f;
      ob[r * ldob + c] = pv * (1.f - f) + iv * f;
    }
  }
}

__global__ void k_reset_mask_bwd2(const float* __restrict__ da, long ldda, float* __restrict__ pa, long ldpa, int ca,
                                  const float* __restrict__ db, long lddb, float* __restrict__ pb, long ldpb, int cb,
                                  const float* __restrict__ first, long fstride, long rows) {
  const int cols = ca + cb;
  long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cols; int c = (int)(i - r * cols);
    float g = 1.f - first[r * fstride];
    if (c < ca) pa[r * ldpa + c] += da[r * ldda + c] * g;
    else { c -= ca; pb[r * ldpb + c] += db[r * lddb + c] * g; }
  }
}

// Batch preparation from the wire format: bool (u8) flags -> float.
//   first_f = is_first, cont_f = 1 - is_terminal, act_masked = action*(1-is_first)
__global__ void k_batch_prep(const unsigned char* __restrict__ is_first,
                             const unsigned char* __restrict__ is_terminal,
                             const float* __restrict__ action, float* __restrict__ first_f,
                             float* __restrict__ cont_f, float* __restrict__ act_masked, long ldm,
                             long n, int A) {
  long total = n * (A > 0 ? A : 1);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / (A > 0 ? A : 1); int a = (int)(i - r * (A > 0 ? A : 1));
    float f = is_first[r] ? 1.f : 0.f;
    if (a == 0) { first_f[r] = f; cont_f[r] = is_terminal[r] ? 0.f : 1.f; }
    if (A > 0) act_masked[r * ldm + a] = action[r * A + a] * (1.f - f);
  }
}

// y = tanh(x) ; or dx = dy * (1 - tanh(x)^2) (accumulating)
__global__ void k_tanh_fwd(const float* x, float* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = tanhf(x[i]);
}
__global__ void k_tanh_bwd(const float* x, const float* dy, float* dx, int n, float beta) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float t = tanhf(x[i]); dx[i] = (beta != 0.f ? beta * dx[i] : 0.f) + dy[i] * (1.f - t * t); }
}

inline int gsz(long n, int t = 256, int cap = 4096) {
  long b = (n + t - 1) / t;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int dd_philox(float* out, long outer, long inner, int cols, long inner_global,
                         long inner_offset, unsigned long long seed,
                         const unsigned long long* step_dev, unsigned site, int kind, void* stream) {
  long total = outer * inner * ((cols + 3) / 4);
  if (total <= 0) return 0;
  k_philox<<<gsz(total), 256, 0, (hipStream_t)stream>>>(out, outer, inner, cols, inner_global, inner_offset,
      (uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32), step_dev, site, kind);
  DD_CHECK_LAUNCH("dd_philox");
  return 0;
}

namespace {
// slot: [0] = number of stamps so far, [1 + i % 8] = the i-th stamp (a ring of the last eight)
__global__ void k_stamp(unsigned long long* slot) {
  const unsigned long long i = slot[0];
  slot[1 + (i & 7)] = wall_clock64();
  slot[0] = i + 1;
}
}
// measurement aid (tools/phase_timeline.py): the 100 MHz wall clock at this point of the stream
extern "C" int dd_stamp(unsigned long long* slot, void* stream) {
  k_stamp<<<1, 1, 0, (hipStream_t)stream>>>(slot);
  DD_CHECK_LAUNCH("dd_stamp");
  return 0;
}

extern "C" int dd_counter_add(unsigned long long* counter, unsigned long long v, void* stream) {
  k_counter_add<<<1, 1, 0, (hipStream_t)stream>>>(counter, v);
  DD_CHECK_LAUNCH("dd_counter_add");
  return 0;
}

extern "C" int dd_reduce_stats(const float* x, long n, long stride, double* sums, float* maxs,
                               void* stream) {
  k_reduce_stats<<<1, 1024, 0, (hipStream_t)stream>>>(x, n, stride, sums, maxs);
  DD_CHECK_LAUNCH("dd_reduce_stats");
  return 0;
}

extern "C" int dd_reduce_stats_multi(int count, const float* const* x, const long* n, const long* stride,
                                     double* const* sums, float* const* maxs, void* stream) {
  if (count <= 0) return 0;
  DD_REQUIRE(count <= 16, "dd_reduce_stats_multi: at most 16 vectors per call");
  StatItems it;
  for (int i = 0; i < count; ++i) {
    it.x[i] = x[i]; it.n[i] = n[i]; it.stride[i] = stride[i]; it.sums[i] = sums[i]; it.maxs[i] = maxs[i];
  }
  k_reduce_stats_multi<<<count, 1024, 0, (hipStream_t)stream>>>(it);
  DD_CHECK_LAUNCH("dd_reduce_stats_multi");
  return 0;
}

extern "C" int dd_autoadapt_update(float* scale, const double* sums, int n, double count,
                                   float target, float thres, float vel, float lo, float hi,
                                   int inverse, int impl, void* stream) {
  DD_REQUIRE(impl == 0 || impl == 1, "dd_autoadapt_update: impl must be 0 (mult) or 1 (prop)");
  k_autoadapt<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(scale, sums, n, count, target, thres, vel, lo, hi, inverse, impl);
  DD_CHECK_LAUNCH("dd_autoadapt_update");
  return 0;
}

extern "C" int dd_normalize_update(double* state, const double* sums, double count,
                                   const float* in_scale_dev, double decay, double maxv, int impl,
                                   int do_update, float* out_off_scale, void* stream) {
  k_normalize_update<<<1, 1, 0, (hipStream_t)stream>>>(state, sums, count, in_scale_dev, decay, maxv, impl, do_update, out_off_scale);
  DD_CHECK_LAUNCH("dd_normalize_update");
  return 0;
}

extern "C" int dd_scalar_mul(float* dst, const float* a, const float* b, float c, int n, void* stream) {
  k_scalar_mul<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(dst, a, b, c, n);
  DD_CHECK_LAUNCH("dd_scalar_mul");
  return 0;
}

extern "C" int dd_grad_norm(const float* g, long n, double* opt_state, double* ws, size_t ws_bytes,
                            int mixed, void* stream) {
  int P = gsz(n, 256, 1024);
  DD_REQUIRE(ws && (size_t)P * sizeof(double) <= ws_bytes, "dd_grad_norm: workspace too small");
  k_sumsq_partial<<<P, 256, 0, (hipStream_t)stream>>>(g, n, ws);
  DD_CHECK_LAUNCH("dd_grad_norm");
  k_norm_finalize<<<1, 64, 0, (hipStream_t)stream>>>(ws, P, opt_state, mixed);
  DD_CHECK_LAUNCH("dd_grad_norm(finalize)");
  return 0;
}

extern "C" int dd_adam_step(float* p, const float* g, float* m, float* v, long n, long n_decay,
                            const double* opt_state, float lr, float wd, float eps, float b1,
                            float b2, float clip, float warmup, void* stream) {
  if (n <= 0) return 0;
  k_adam<<<gsz(n, 256, 2048), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, n_decay, opt_state, lr, wd, eps, b1, b2, clip, warmup);
  DD_CHECK_LAUNCH("dd_adam_step");
  return 0;
}

extern "C" int dd_axpy(const float* x, float alpha, const float* alpha_dev, float* y, long n,
                       int accumulate, void* stream) {
  if (n <= 0) return 0;
  k_axpy<<<gsz(n), 256, 0, (hipStream_t)stream>>>(x, alpha, alpha_dev, y, n, accumulate);
  DD_CHECK_LAUNCH("dd_axpy");
  return 0;
}

extern "C" int dd_balance_stats(const float* out, const float* target, const float* loss, long n,
                                float thres, int kind, double* out7, double* ws, size_t ws_bytes,
                                void* stream) {
  if (n <= 0) return 0;
  int P = (int)((n + 2047) / 2048);
  if (P > 256) P = 256;
  DD_REQUIRE(ws && (size_t)P * 7 * sizeof(double) <= ws_bytes, "dd_balance_stats: workspace too small");
  k_balance_partial<<<P, 256, 0, (hipStream_t)stream>>>(out, target, loss, n, thres, kind, ws);
  DD_CHECK_LAUNCH("dd_balance_stats");
  k_balance_final<<<1, 64, 0, (hipStream_t)stream>>>(ws, P, out7);
  DD_CHECK_LAUNCH("dd_balance_stats(final)");
  return 0;
}

extern "C" int dd_fill(float* p, long n, float v, void* stream) {
  if (n <= 0) return 0;
  k_fill<<<gsz(n), 256, 0, (hipStream_t)stream>>>(p, n, v);
  DD_CHECK_LAUNCH("dd_fill");
  return 0;
}

extern "C" int dd_copy2d(const float* src, long lds, float* dst, long ldd, long rows, int cols,
                         void* stream) {
  if (rows * cols <= 0) return 0;
  k_copy2d<<<gsz(rows * cols), 256, 0, (hipStream_t)stream>>>(src, lds, dst, ldd, rows, cols);
  DD_CHECK_LAUNCH("dd_copy2d");
  return 0;
}

// Replay minibatch assembly from an HBM-resident episode ring (byte rows): sample b is
// the T consecutive ring rows starting at starts[b] (a chunk never wraps: episodes are
// stored contiguously).  One workgroup column per sample, 16-byte lanes when the row
// size allows (images, vectors), bytes otherwise (reward, flags).
__global__ void __launch_bounds__(256)
k_replay_gather(const unsigned char* __restrict__ ring, long row_bytes,
                const long long* __restrict__ starts, int T, unsigned char* __restrict__ out,
                int first_flag) {
  const int b = blockIdx.y;
  const long n = (long)T * row_bytes;
  unsigned char* o = out + (long)b * n;
  if (first_flag) {  // is_first of a sampled chunk: 1 at t = 0, else 0 (fixed_length.py:79-80)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
      o[i] = (i < row_bytes) ? 1 : 0;
    return;
  }
  const unsigned char* s = ring + starts[b] * row_bytes;
  if ((row_bytes & 15) == 0 && (((uintptr_t)ring | (uintptr_t)out) & 15) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    uint4* o4 = reinterpret_cast<uint4*>(o);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n / 16; i += (long)gridDim.x * 256)
      o4[i] = s4[i];
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
      o[i] = s[i];
  }
}

extern "C" int dd_replay_gather(const void* ring, long row_bytes, const long long* starts,
                                int B, int T, void* out, int first_flag, void* stream) {
  if (B <= 0 || T <= 0 || row_bytes <= 0) return 0;
  const long n = (long)T * row_bytes;
  long per = (n / 16 + 255) / 256;
  if (per < 1) per = 1;
  if (per > 64) per = 64;
  k_replay_gather<<<dim3((unsigned)per, (unsigned)B), 256, 0, (hipStream_t)stream>>>(
      (const unsigned char*)ring, row_bytes, starts, T, (unsigned char*)out, first_flag);
  DD_CHECK_LAUNCH("dd_replay_gather");
  return 0;
}

extern "C" int dd_reset_mask(const float* prev, long ldp, const float* first, long fstride,
                             const float* init, float* out, long ldo, long rows, int cols,
                             void* stream) {
  if (rows * cols <= 0) return 0;
  k_reset_mask<<<gsz(rows * cols), 256, 0, (hipStream_t)stream>>>(prev, ldp, first, fstride, init, out, ldo, rows, cols);
  DD_CHECK_LAUNCH("dd_reset_mask");
  return 0;
}

extern "C" int dd_reset_mask_bwd(const float* dout, long ldo, const float* first, long fstride,
                                 float* dprev, long ldp, long rows, int cols, void* stream) {
  if (rows * cols <= 0) return 0;
  k_reset_mask_bwd<<<gsz(rows * cols), 256, 0, (hipStream_t)stream>>>(dout, ldo, first, fstride, dprev, ldp, rows, cols);
  DD_CHECK_LAUNCH("dd_reset_mask_bwd");
  return 0;
}

extern "C" int dd_reset_mask2(const float* prev_a, long ldpa, const float* init_a, float* out_a, long ldoa, int cols_a,
                              const float* prev_b, long ldpb, const float* init_b, float* out_b, long ldob, int cols_b,
                              const float* first, long fstride, long rows, void* stream) {
  if (rows * (cols_a + cols_b) <= 0) return 0;
  k_reset_mask2<<<gsz(rows * (cols_a + cols_b)), 256, 0, (hipStream_t)stream>>>(
      prev_a, ldpa, init_a, out_a, ldoa, cols_a, prev_b, ldpb, init_b, out_b, ldob, cols_b, first, fstride, rows);
  DD_CHECK_LAUNCH("dd_reset_mask2");
  return 0;
}

extern "C" int dd_reset_mask_bwd2(const float* dout_a, long ldda, float* dprev_a, long ldpa, int cols_a,
                                  const float* dout_b, long lddb, float* dprev_b, long ldpb, int cols_b,
                                  const float* first, long fstride, long rows, void* stream) {
  if (rows * (cols_a + cols_b) <= 0) return 0;
  k_reset_mask_bwd2<<<gsz(rows * (cols_a + cols_b)), 256, 0, (hipStream_t)stream>>>(
      dout_a, ldda, dprev_a, ldpa, cols_a, dout_b, lddb, dprev_b, ldpb, cols_b, first, fstride, rows);
  DD_CHECK_LAUNCH("dd_reset_mask_bwd2");
  return 0;
}

extern "C" int dd_batch_prep(const unsigned char* is_first, const unsigned char* is_terminal,
                             const float* action, float* first_f, float* cont_f,
                             float* act_masked, long ldm, long n, int A, void* stream) {
  if (n <= 0) return 0;
  k_batch_prep<<<gsz(n * (A > 0 ? A : 1)), 256, 0, (hipStream_t)stream>>>(is_first, is_terminal, action, first_f, cont_f, act_masked, ldm, n, A);
  DD_CHECK_LAUNCH("dd_batch_prep");
  return 0;
}

extern "C" int dd_tanh_fwd(const float* x, float* y, int n, void* stream) {
  k_tanh_fwd<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, y, n);
  DD_CHECK_LAUNCH("dd_tanh_fwd");
  return 0;
}

extern "C" int dd_tanh_bwd(const float* x, const float* dy, float* dx, int n, float beta, void* stream) {
  k_tanh_bwd<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, dy, dx, n, beta);
  DD_CHECK_LAUNCH("dd_tanh_bwd");
  return 0;
}
