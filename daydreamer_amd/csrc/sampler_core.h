// Arithmetic of the categorical latent draw, shared VERBATIM by the gfx950 kernel
// (k_stats_fwd, latent.hip) and its host twin (dd_onehot_sample_host, same file): every
// function here is `__host__ __device__`, uses only IEEE-754 single operations whose result
// does not depend on the target (add, mul, div, floor, compare; no fused multiply-add - the
// library is built with -ffp-contract=off - and no libm transcendental, whose last bit
// differs between the device library and glibc), so that the same (statistics, uniform)
// pair yields the same class index on the MI355X and on the host, bit for bit.
//
// Replaces OneHotDist.sample (reference tfutils.py:368-382: tf.random.categorical on the
// unimixed logits, nets.py:162-171).  The draw is inverse-CDF:
//   idx = #{c < C-1 : cdf_c <= u * cdf_{C-1}},  cdf = inclusive scan of the mixed probs.
#pragma once

#if defined(__HIPCC__)
#define DD_HD __host__ __device__ __forceinline__
#else
#define DD_HD static inline
#endif

// exp(x) for x <= 0 (softmax terms after the max subtraction).  Cody-Waite reduction with a
// 9-bit ln2_hi (n * ln2_hi is exact for |n| < 2^15), degree-5 polynomial (cephes expf
// coefficients), scaling by 2^n through the exponent field.  exp(0) == 1 exactly; below
// -86 the result would leave the normal range and is defined as 0 (no denormals anywhere,
// so flush-to-zero modes cannot matter).  Relative error <= 2^-22.
DD_HD float dd_exp_det(float x) {
  if (!(x >= -86.0f)) return 0.0f;
  const float fn = floorf(1.44269504088896341f * x + 0.5f);
  float r = x - fn * 0.693359375f;
  r = r - fn * -2.12194440e-4f;
  const float z = r * r;
  float p = 1.9875691500e-4f;
  p = p * r + 1.3981999507e-3f;
  p = p * r + 8.3334519073e-3f;
  p = p * r + 4.1665795894e-2f;
  p = p * r + 1.6666665459e-1f;
  p = p * r + 5.0000001201e-1f;
  float y = p * z + r;
  y = y + 1.0f;
  const int n = (int)fn;
  return y * __builtin_bit_cast(float, (unsigned)(n + 127) << 23);
}

// (1 - eps) * softmax + eps / C   (reference nets.py:166-169)
DD_HD float dd_unimix_prob(float e, float s, float unimix, int C) {
  const float p = e / s;
  return (1.0f - unimix) * p + unimix / (float)C;
}

// threshold of the inverse-CDF draw
DD_HD float dd_draw_threshold(float u, float cdf_total) { return u * cdf_total; }
