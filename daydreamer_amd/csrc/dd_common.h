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

// Deferred split-K sum (dd_gemm_f32 with `deferred`): S partial results [S][rows][N] sit
// in the GEMM's workspace; the consumer kernel adds them in the reduce pass's order
// (slabs ascending, then bias, then beta * old value), writes the total back to its input
// buffer (later kernels read it) and continues.  S = 0: plain input.
struct PreSum {
  const float* p; int S; long MN; int N; float beta; const float* bias;
};
// (the slab loads of four consecutive slabs are issued together - independent addresses - and
// added in ascending order: the sum is bit-identical to the sequential loop, but a consumer
// waits for one memory round trip per four slabs instead of one per slab)
__device__ __forceinline__ float presum1(const PreSum& ps, long row, int c, float old) {
  const float* q = ps.p + row * ps.N + c;
  float s = 0.f;
  int z = 0;
  for (; z + 4 <= ps.S; z += 4) {
    const float t0 = q[(z + 0) * ps.MN], t1 = q[(z + 1) * ps.MN], t2 = q[(z + 2) * ps.MN], t3 = q[(z + 3) * ps.MN];
    s += t0; s += t1; s += t2; s += t3;
  }
  for (; z < ps.S; ++z) s += q[z * ps.MN];
  if (ps.bias) s += ps.bias[c];
  if (ps.beta != 0.f) s += ps.beta * old;
  return s;
}
__device__ __forceinline__ void f4_acc(float4& s, const float4 t) { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
__device__ __forceinline__ float4 presum4(const PreSum& ps, long row, int c, float4 old) {
  const float* q = ps.p + row * ps.N + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int z = 0;
  for (; z + 4 <= ps.S; z += 4) {
    const float4 t0 = *reinterpret_cast<const float4*>(q + (z + 0) * ps.MN);
    const float4 t1 = *reinterpret_cast<const float4*>(q + (z + 1) * ps.MN);
    const float4 t2 = *reinterpret_cast<const float4*>(q + (z + 2) * ps.MN);
    const float4 t3 = *reinterpret_cast<const float4*>(q + (z + 3) * ps.MN);
    f4_acc(s, t0); f4_acc(s, t1); f4_acc(s, t2); f4_acc(s, t3);
  }
  for (; z < ps.S; ++z) f4_acc(s, *reinterpret_cast<const float4*>(q + z * ps.MN));
  if (ps.bias) {
    float4 b = *reinterpret_cast<const float4*>(ps.bias + c);
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
  }
  if (ps.beta != 0.f) { s.x += ps.beta * old.x; s.y += ps.beta * old.y; s.z += ps.beta * old.z; s.w += ps.beta * old.w; }
  return s;
}

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
