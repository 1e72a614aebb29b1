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

// xv: this lane's raw statistic (class c = lane % LW of the item), ok: c < C and the item is
// live, uu: the item's uniform.  Returns the normalised log-probability in lg and the drawn
// (mode 0) or most likely (mode 1) class in idx (valid on every lane of the sub-wave).
template <int LW>
__device__ __forceinline__ void stats_item(float xv, bool ok, int c, int sub, int C, float unimix,
                                           int mode, float uu, float& lg, int& idx) {
  if (!ok) xv = -INFINITY;
  // (the arithmetic below is sampler_core.h, shared with dd_onehot_sample_host)
  float m = sub_max<LW>(xv);
  float e = ok ? dd_exp_det(xv - m) : 0.f;
  float s = sub_sum<LW>(e);
  float pm = ok ? dd_unimix_prob(e, s, unimix, C) : 0.f;
  lg = unimix > 0.f ? logf(pm) : (xv - m) - logf(s);
  if (mode == 1) {
    float best = sub_max<LW>(ok ? pm : -1.f);
    unsigned long long b = __ballot(ok && pm == best);
    if constexpr (LW < 64) b = (b >> (sub * LW)) & ((1ull << LW) - 1ull);
    idx = __ffsll((long long)b) - 1;
  } else {
    // inclusive Kogge-Stone scan inside the sub-wave
    float cdf = pm;
#pragma unroll
    for (int o = 1; o < LW; o <<= 1) {
      float t = __shfl_up(cdf, o, LW);
      if (c >= o) cdf += t;
    }
    float tot = __shfl(cdf, C - 1, LW);
    float thr = dd_draw_threshold(uu, tot);
    float flag = (ok && c < C - 1 && cdf <= thr) ? 1.f : 0.f;
    idx = (int)sub_sum<LW>(flag);
  }
}

}  // namespace
