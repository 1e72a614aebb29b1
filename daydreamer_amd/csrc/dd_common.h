// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DD_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Set by every C-ABI entry point on failure; read with dd_last_error().
void dd_set_error(const char* where, hipError_t e);
void dd_set_error_msg(const char* msg);

#define DD_CHECK_LAUNCH(name)                         \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) {                          \
      dd_set_error(name, e__);                        \
      return (int)e__;                                \
    }                                                 \
  } while (0)

#define DD_REQUIRE(cond, msg)                         \
  do {                                                \
    if (!(cond)) {                                    \
      dd_set_error_msg(msg);                          \
      return -1;                                      \
    }                                                 \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) {
  return 1.0f / (1.0f + expf(-x));
}

__device__ __forceinline__ float symlogf_(float x) {
  return copysignf(log1pf(fabsf(x)), x);
}

__device__ __forceinline__ float symexpf_(float x) {
  return copysignf(expm1f(fabsf(x)), x);
}

// log(sigmoid(x)), stable.
__device__ __forceinline__ float logsigmoidf_(float x) {
  return fminf(x, 0.0f) - log1pf(expf(-fabsf(x)));
}

static inline int dd_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
