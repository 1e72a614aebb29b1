// Discrete stochastic latent: fused softmax + unimix + log + categorical draw
// + one-hot (straight-through forward value), its backward, and the balanced
// categorical KL with both stop-gradient directions in one pass.
//
// Reference: RSSM._stats_layer nets.py:162-171, OneHotDist.sample
// tfutils.py:368-382, RSSM.kl_loss nets.py:178-183 (get_dist nets.py:88-91).
//
// A (row, group) softmax over C <= 64 classes is owned by an LW-lane sub-wave
// (LW = next power of two >= C); all reductions are butterfly shuffles inside
// the sub-wave.  The draw is inverse-CDF: idx = #{c < C-1 : cdf_c <= u*cdf_{C-1}}
// with a Kogge-Stone inclusive scan over the (unimixed) probabilities.
#include "latent_core.h"
#include <math.h>
#include <type_traits>
#include "../../include/daydreamer_hip.h"

namespace {

// mode: 0 sample with u, 1 argmax (OneHotCategorical.mode()).
template <int LW>
__global__ void __launch_bounds__(256)
k_stats_fwd(float* __restrict__ x, long ldx, const float* __restrict__ u, long ldu,
            float* __restrict__ logit, long ldl, float* __restrict__ stoch, long lds,
            int rows, int G, int C, float unimix, int mode, PreSum ps) {
  constexpr int GPW = 64 / LW;  // groups per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane % LW, sub = lane / LW;
  const long total = (long)rows * G;
  const long witems = (total + GPW - 1) / GPW;  // wave work items
  for (long wi = (long)blockIdx.x * 4 + wave; wi < witems; wi += (long)gridDim.x * 4) {
    const long item = wi * GPW + sub;
    const bool live = item < total;
    const long row = live ? item / G : 0;
    const int g = live ? (int)(item - row * G) : 0;
    const bool ok = live && c < C;
    float xv = -INFINITY;
    if (ok) {
      float* xp = x + row * ldx + (long)g * C + c;
      if (ps.S) {
        xv = presum1(ps, row, g * C + c, ps.beta != 0.f ? *xp : 0.f);
        *xp = xv;  // the backward pass reads the raw statistics
      } else {
        xv = *xp;
      }
    }
    float lg;
    int idx;
    const float uu = (mode != 1 && live) ? u[row * ldu + g] : 0.f;
    stats_item<LW>(xv, ok, c, sub, C, unimix, mode, uu, lg, idx);
    if (ok) {
      logit[row * ldl + (long)g * C + c] = lg;
      stoch[row * lds + (long)g * C + c] = (c == idx) ? 1.f : 0.f;
    }
  }
}

// dx from dlogit (nullable) and dstoch (nullable): probs_mixed pm =
// (1-eps)*softmax(x)+eps/C; logit = log pm; straight-through adds d pm.
template <int LW>
__global__ void __launch_bounds__(256)
k_stats_bwd(const float* __restrict__ x, long ldx, const float* __restrict__ dlogit, long ldl,
            const float* __restrict__ dstoch, long lds, float* __restrict__ dx, long lddx,
            int rows, int G, int C, float unimix) {
  constexpr int GPW = 64 / LW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane % LW, sub = lane / LW;
  const long total = (long)rows * G;
  const long witems = (total + GPW - 1) / GPW;
  for (long wi = (long)blockIdx.x * 4 + wave; wi < witems; wi += (long)gridDim.x * 4) {
    const long item = wi * GPW + sub;
    const bool live = item < total;
    const long row = live ? item / G : 0;
    const int g = live ? (int)(item - row * G) : 0;
    const bool ok = live && c < C;
    const long col = (long)g * C + c;
    float xv = ok ? x[row * ldx + col] : -INFINITY;
    float m = sub_max<LW>(xv);
    float e = ok ? dd_exp_det(xv - m) : 0.f;
    float s = sub_sum<LW>(e);
    float p = e / s;
    float pm = dd_unimix_prob(e, s, unimix, C);
    float dpm = 0.f;
    if (ok) {
      if (dstoch) dpm += dstoch[row * lds + col];
      if (dlogit) dpm += dlogit[row * ldl + col] / pm;
    }
    float dp = (1.f - unimix) * dpm;
    float dot = sub_sum<LW>(ok ? dp * p : 0.f);
    if (ok) dx[row * lddx + col] = p * (dp - dot);
  }
}

// One row per block: kl[row] = sum_g KL(post_g || prior_g), plus entropies.
template <int LW>
__global__ void __launch_bounds__(256)
k_kl_fwd(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
         float* __restrict__ kl, float* __restrict__ ent_a, float* __restrict__ ent_b,
         int rows, int G, int C) {
  constexpr int GPW = 64 / LW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane % LW, sub = lane / LW;
  const long row = blockIdx.x;
  if (row >= rows) return;
  float sk = 0.f, sa = 0.f, sb = 0.f;
  for (int g0 = wave * GPW; g0 < G; g0 += 4 * GPW) {
    const int g = g0 + sub;
    const bool ok = g < G && c < C;
    const long col = (long)g * C + c;
    float av = ok ? a[row * lda + col] : -INFINITY;
    float bv = ok ? b[row * ldb + col] : -INFINITY;
    float ma = sub_max<LW>(av), mb = sub_max<LW>(bv);
    float ea = ok ? expf(av - ma) : 0.f, eb = ok ? expf(bv - mb) : 0.f;
    float lsa = logf(sub_sum<LW>(ea)), lsb = logf(sub_sum<LW>(eb));
    if (ok) {
      float la = av - ma - lsa, lb = bv - mb - lsb;
      float pa = expf(la), pb = expf(lb);
      sk += pa * (la - lb);
      sa -= pa * la;
      sb -= pb * lb;
    }
  }
  sk = wave_sum(sk); sa = wave_sum(sa); sb = wave_sum(sb);
  __shared__ float sh[3][4];
  if (lane == 0) { sh[0][wave] = sk; sh[1][wave] = sa; sh[2][wave] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    kl[row] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    ent_a[row] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    ent_b[row] = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
  }
}

// d loss / d post_logit  = coef*(1-bal) * p*((la-lb) - KL_g)
// d loss / d prior_logit = coef*bal     * (q - p)
// coef = coef_host * (coef_dev ? *coef_dev : 1).
template <int LW>
__global__ void __launch_bounds__(256)
k_kl_bwd(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
         const float* __restrict__ coef_dev, float coef_host, float balance,
         float* __restrict__ da, long ldda, float* __restrict__ db, long lddb,
         int rows, int G, int C) {
  constexpr int GPW = 64 / LW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane % LW, sub = lane / LW;
  const float coef = coef_host * (coef_dev ? *coef_dev : 1.f);
  const long total = (long)rows * G;
  const long witems = (total + GPW - 1) / GPW;
  for (long wi = (long)blockIdx.x * 4 + wave; wi < witems; wi += (long)gridDim.x * 4) {
    const long item = wi * GPW + sub;
    const bool live = item < total;
    const long row = live ? item / G : 0;
    const int g = live ? (int)(item - row * G) : 0;
    const bool ok = live && c < C;
    const long col = (long)g * C + c;
    float av = ok ? a[row * lda + col] : -INFINITY;
    float bv = ok ? b[row * ldb + col] : -INFINITY;
    float ma = sub_max<LW>(av), mb = sub_max<LW>(bv);
    float ea = ok ? expf(av - ma) : 0.f, eb = ok ? expf(bv - mb) : 0.f;
    float lsa = logf(sub_sum<LW>(ea)), lsb = logf(sub_sum<LW>(eb));
    float la = ok ? av - ma - lsa : 0.f, lb = ok ? bv - mb - lsb : 0.f;
    float pa = ok ? expf(la) : 0.f, pb = ok ? expf(lb) : 0.f;
    float klg = sub_sum<LW>(pa * (la - lb));
    if (ok) {
      da[row * ldda + col] = coef * (1.f - balance) * pa * ((la - lb) - klg);
      db[row * lddb + col] = coef * balance * (pb - pa);
    }
  }
}

template <typename F>
int dispatch_lw(int C, F f) {
  if (C <= 8) return f(std::integral_constant<int, 8>());
  if (C <= 16) return f(std::integral_constant<int, 16>());
  if (C <= 32) return f(std::integral_constant<int, 32>());
  return f(std::integral_constant<int, 64>());
}

inline int item_blocks(long rows, int G, int LW) {
  long witems = ((long)rows * G + (64 / LW) - 1) / (64 / LW);
  long b = (witems + 3) / 4;
  if (b > 65536) b = 65536;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int dd_stats_sample_fwd(float* x, long ldx, const float* u, long ldu,
                                   float* logit, long ldl, float* stoch, long lds,
                                   int rows, int G, int C, float unimix, int mode,
                                   const float* slabs, int n_slabs, float beta_pre,
                                   const float* bias_pre, void* stream) {
  if (rows <= 0) return 0;
  DD_REQUIRE(C >= 2 && C <= 64, "dd_stats_sample_fwd: classes must be in [2,64]");
  DD_REQUIRE(mode == 1 || u != nullptr, "dd_stats_sample_fwd: noise required");
  PreSum ps{slabs, n_slabs, (long)rows * G * C, G * C, beta_pre, bias_pre};
  return dispatch_lw(C, [&](auto lw) {
    constexpr int LW = decltype(lw)::value;
    k_stats_fwd<LW><<<item_blocks(rows, G, LW), 256, 0, (hipStream_t)stream>>>(
        x, ldx, u, ldu, logit, ldl, stoch, lds, rows, G, C, unimix, mode, ps);
    DD_CHECK_LAUNCH("dd_stats_sample_fwd");
    return 0;
  });
}

// Host twin of k_stats_fwd: the same arithmetic (sampler_core.h) in the same order - the
// butterfly sum of the LW-lane sub-wave, the Kogge-Stone scan - on the host cores, so
// a (statistics, uniform) pair yields the same class index as on the device, bit for bit.
// logit / stoch / index are optional outputs (logit goes through libm's logf: equal to the
// device's only to the last bit or two; the draw never reads it).
extern "C" int dd_onehot_sample_host(const float* x, long ldx, const float* u, long ldu,
                                     float* logit, long ldl, float* stoch, long lds,
                                     int* index, long ldi, int rows, int G, int C,
                                     float unimix, int mode) {
  if (rows <= 0) return 0;
  DD_REQUIRE(C >= 2 && C <= 64, "dd_onehot_sample_host: classes must be in [2,64]");
  DD_REQUIRE(mode == 1 || u != nullptr, "dd_onehot_sample_host: noise required");
  int LW = 8;
  while (LW < C) LW *= 2;
  float e[64], t[64], pm[64], cdf[64];
  for (long row = 0; row < rows; ++row)
    for (int g = 0; g < G; ++g) {
      const float* xp = x + row * ldx + (long)g * C;
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) m = fmaxf(m, xp[c]);
      for (int c = 0; c < LW; ++c) e[c] = c < C ? dd_exp_det(xp[c] - m) : 0.f;
      // sub_sum<LW>: v += shfl_xor(v, o) for o = LW/2 .. 1 (every lane ends with lane 0's value)
      for (int c = 0; c < LW; ++c) t[c] = e[c];
      for (int o = LW / 2; o > 0; o >>= 1) {
        float n[64];
        for (int c = 0; c < LW; ++c) n[c] = t[c] + t[c ^ o];
        for (int c = 0; c < LW; ++c) t[c] = n[c];
      }
      const float s = t[0];
      for (int c = 0; c < LW; ++c) pm[c] = c < C ? dd_unimix_prob(e[c], s, unimix, C) : 0.f;
      int idx = 0;
      if (mode == 1) {
        float best = -1.f;
        for (int c = 0; c < C; ++c) best = fmaxf(best, pm[c]);
        while (pm[idx] != best) ++idx;
      } else {
        for (int c = 0; c < LW; ++c) cdf[c] = pm[c];
        for (int o = 1; o < LW; o <<= 1) {
          float n[64];
          for (int c = 0; c < LW; ++c) n[c] = c >= o ? cdf[c] + cdf[c - o] : cdf[c];
          for (int c = 0; c < LW; ++c) cdf[c] = n[c];
        }
        const float thr = dd_draw_threshold(u[row * ldu + g], cdf[C - 1]);
        for (int c = 0; c < C - 1; ++c) idx += cdf[c] <= thr ? 1 : 0;
      }
      if (index) index[row * ldi + g] = idx;
      for (int c = 0; c < C; ++c) {
        if (logit) logit[row * ldl + (long)g * C + c] = unimix > 0.f ? logf(pm[c]) : (xp[c] - m) - logf(s);
        if (stoch) stoch[row * lds + (long)g * C + c] = c == idx ? 1.f : 0.f;
      }
    }
  return 0;
}

extern "C" int dd_stats_sample_bwd(const float* x, long ldx, const float* dlogit, long ldl,
                                   const float* dstoch, long lds, float* dx, long lddx,
                                   int rows, int G, int C, float unimix, void* stream) {
  if (rows <= 0) return 0;
  DD_REQUIRE(C >= 2 && C <= 64, "dd_stats_sample_bwd: classes must be in [2,64]");
  return dispatch_lw(C, [&](auto lw) {
    constexpr int LW = decltype(lw)::value;
    k_stats_bwd<LW><<<item_blocks(rows, G, LW), 256, 0, (hipStream_t)stream>>>(
        x, ldx, dlogit, ldl, dstoch, lds, dx, lddx, rows, G, C, unimix);
    DD_CHECK_LAUNCH("dd_stats_sample_bwd");
    return 0;
  });
}

extern "C" int dd_cat_kl_fwd(const float* post, long ldp, const float* prior, long ldq,
                             float* kl, float* ent_post, float* ent_prior,
                             int rows, int G, int C, void* stream) {
  if (rows <= 0) return 0;
  DD_REQUIRE(C >= 2 && C <= 64, "dd_cat_kl_fwd: classes must be in [2,64]");
  return dispatch_lw(C, [&](auto lw) {
    constexpr int LW = decltype(lw)::value;
    k_kl_fwd<LW><<<rows, 256, 0, (hipStream_t)stream>>>(post, ldp, prior, ldq, kl, ent_post, ent_prior, rows, G, C);
    DD_CHECK_LAUNCH("dd_cat_kl_fwd");
    return 0;
  });
}

extern "C" int dd_cat_kl_bwd(const float* post, long ldp, const float* prior, long ldq,
                             const float* coef_dev, float coef_host, float balance,
                             float* dpost, long lddp, float* dprior, long lddq,
                             int rows, int G, int C, void* stream) {
  if (rows <= 0) return 0;
  DD_REQUIRE(C >= 2 && C <= 64, "dd_cat_kl_bwd: classes must be in [2,64]");
  return dispatch_lw(C, [&](auto lw) {
    constexpr int LW = decltype(lw)::value;
    k_kl_bwd<LW><<<item_blocks(rows, G, LW), 256, 0, (hipStream_t)stream>>>(
        post, ldp, prior, ldq, coef_dev, coef_host, balance, dpost, lddp, dprior, lddq, rows, G, C);
    DD_CHECK_LAUNCH("dd_cat_kl_bwd");
    return 0;
  });
}
