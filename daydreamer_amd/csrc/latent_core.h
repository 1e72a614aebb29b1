// One (row, group) item of the categorical latent: softmax + unimix + log + draw, owned by an
// LW-lane sub-wave (LW = next power of two >= C).  Shared by k_stats_fwd (latent.hip) and the
// fused observe scan (scan.hip): the same instruction sequence - and therefore the same class
// index as the host twin dd_onehot_sample_host - wherever a latent is drawn.
#pragma once
#include "dd_common.h"
#include "sampler_core.h"

namespace {

template <int LW>
__device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
  for (int o = LW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LW>
__device__ __forceinline__ float sub_max(float v) {
#pragma unroll
  for (int o = LW / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// NI independent items in lock-step (item i: this lane's raw statistic xv[i] of class
// c = lane % LW, ok[i]: c < C and the item is live, uu[i]: the item's uniform): stage by stage
// over the items, so that the shuffle / transcendental latencies of the items overlap.  The
// arithmetic of one item does not depend on NI.  lg[i] = the normalised log-probability,
// idx[i] = the drawn (mode 0) or most likely (mode 1) class (valid on every lane of the sub-wave).
template <int LW, int NI>
__device__ __forceinline__ void stats_items(float (&xv)[NI], const bool (&ok)[NI], int c, int sub, int C,
                                            float unimix, int mode, const float (&uu)[NI],
                                            float (&lg)[NI], int (&idx)[NI]) {
  // (the arithmetic below is sampler_core.h, shared with dd_onehot_sample_host)
  float m[NI], e[NI], s[NI], pm[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) { if (!ok[i]) xv[i] = -INFINITY; m[i] = xv[i]; }
#pragma unroll
  for (int o = LW / 2; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < NI; ++i) m[i] = fmaxf(m[i], __shfl_xor(m[i], o, 64));
#pragma unroll
  for (int i = 0; i < NI; ++i) { e[i] = ok[i] ? dd_exp_det(xv[i] - m[i]) : 0.f; s[i] = e[i]; }
#pragma unroll
  for (int o = LW / 2; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < NI; ++i) s[i] += __shfl_xor(s[i], o, 64);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    pm[i] = ok[i] ? dd_unimix_prob(e[i], s[i], unimix, C) : 0.f;
    lg[i] = unimix > 0.f ? logf(pm[i]) : (xv[i] - m[i]) - logf(s[i]);
  }
  if (mode == 1) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float best = sub_max<LW>(ok[i] ? pm[i] : -1.f);
      unsigned long long b = __ballot(ok[i] && pm[i] == best);
      if constexpr (LW < 64) b = (b >> (sub * LW)) & ((1ull << LW) - 1ull);
      idx[i] = __ffsll((long long)b) - 1;
    }
  } else {
    // inclusive Kogge-Stone scan inside the sub-wave
    float cdf[NI], flag[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) cdf[i] = pm[i];
#pragma unroll
    for (int o = 1; o < LW; o <<= 1)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float t = __shfl_up(cdf[i], o, LW);
        if (c >= o) cdf[i] += t;
      }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float tot = __shfl(cdf[i], C - 1, LW);
      float thr = dd_draw_threshold(uu[i], tot);
      flag[i] = (ok[i] && c < C - 1 && cdf[i] <= thr) ? 1.f : 0.f;
    }
#pragma unroll
    for (int o = LW / 2; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < NI; ++i) flag[i] += __shfl_xor(flag[i], o, 64);
#pragma unroll
    for (int i = 0; i < NI; ++i) idx[i] = (int)flag[i];
  }
}

// One item (k_stats_fwd).
template <int LW>
__device__ __forceinline__ void stats_item(float xv, bool ok, int c, int sub, int C, float unimix,
                                           int mode, float uu, float& lg, int& idx) {
  float x1[1] = {xv}, u1[1] = {uu}, l1[1];
  const bool o1[1] = {ok};
  int i1[1];
  stats_items<LW, 1>(x1, o1, c, sub, C, unimix, mode, u1, l1, i1);
  lg = l1[0];
  idx = i1[0];
}

}  // namespace
