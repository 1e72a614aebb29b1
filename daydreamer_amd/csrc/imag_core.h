// Device helpers shared by the fused imagination kernels (imag.hip: continuous actions, deter =
// units = 256, forward and reverse; imag_oh.hip: one-hot actions, deter = units = 512, forward):
// the exact 3-way bf16 split, the streamed-weight contraction of a wave's column tiles against an
// A operand in LDS, the row-wise thread mapping and the serial restatement of the categorical draw.
#pragma once
#include "latent_core.h"
#include <math.h>

// ---- arguments of the fused forward rollout (imag.hip: 16 rows per workgroup; imag32.hip: 32) ----
struct DDImagLayerP {            // a Linear + LayerNorm + ELU layer in one row space
  const char* planes;      // fragment-major bf16 planes [N/16][K/32][3][64][8]
  const float* gamma;
  const float* beta;
  float* z;                // [rows, N] pre-norm
  float* st;               // [rows, 2] mean, rstd
  float* out;              // [rows, N] post-activation
};

struct DDImagArgs {
  int N, H;
  int t0, t1;              // this launch runs the policy of steps t0 .. t1 - 1 and the img_steps of those < H
  float unimix, lo, hi;
  float* traj;             // [H+1, N, F + A]
  const float* u_img;      // [H, N, G]
  const float* eps;        // [H+1, N, A]
  DDImagLayerP actor[4];
  const float* w_actor0;   // actor dense0 kernel [F, AU] fp32 (stoch rows are gathered)
  const char* head_planes; // [2A -> padded][AU]
  const float* head_bias_m;
  const float* head_bias_s;
  float* z_om;             // [M, A]
  float* z_os;             // [M, A]
  DDImagLayerP img_in;           // planes unused
  const float* w_in;       // img_in kernel [S + A, U] fp32
  DDImagLayerP gru;              // gamma / beta over 3D; z = iz3 [H*N, 3D], st = igstats; out unused
  DDImagLayerP img_out[3];
  const char* stats_planes;
  const float* stats_bias;
  float* xs;               // [H*N, S] raw statistics
  unsigned long long* dbg; // optional: time stamps of step 1 on block 0 (100 MHz wall clock)
};

// ---- arguments of the fused reverse pass ----
struct DDImagLayerB {            // backward view of a Linear + LayerNorm + ELU layer
  const char* planes;      // W^T as fragment-major planes [K_fwd/16 tiles][N_fwd/32 k-steps]
  const float* gamma;
  const float* z;
  const float* st;
  const float* out;
};

struct DDImagBwdArgs {
  int N, H;
  float unimix;
  const float* traj;       // [H+1, N, F + A]
  float* dtraj;            // [H+1, N, F + A]
  const float* xs;         // [H*N, S] raw statistics
  const char* stats_planes;  // W_stats^T: K = S, N = U
  DDImagLayerB img_out[3];
  const char* gru_planes;    // W_gru^T: K = 3D, N = D + U
  const float* gru_gamma;
  const float* gru_beta;
  const float* z3;         // [H*N, 3D]
  const float* gstats;     // [H*N, 2]
  DDImagLayerB img_in;           // planes: W_in^T: K = U, N = S + A
  unsigned long long* dbg;
};

// the 32-row form (imag32.hip); returns 0 when it launched the shape, 1 when it does not cover it
int dd_imag32_launch(const DDImagArgs& a, int D, int U, int G, int C, int A, int AU, hipStream_t st);
int dd_imag32_lds_bytes();
int dd_imag32_bwd_launch(const DDImagBwdArgs& a, int D, int U, int G, int C, int A, hipStream_t st);

namespace {

constexpr float LN_EPS = 1e-3f;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;


__device__ __forceinline__ float fexp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ unsigned pack_hi(unsigned even, unsigned odd) {
  return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&pl)[3]) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
  pl[0] = make_uint4(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]), pack_hi(h[4], h[5]), pack_hi(h[6], h[7]));
  pl[1] = make_uint4(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]), pack_hi(m[4], m[5]), pack_hi(m[6], m[7]));
  pl[2] = make_uint4(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]), pack_hi(l[4], l[5]), pack_hi(l[6], l[7]));
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// ---- weight stream -------------------------------------------------------------------------
// This wave's NT column tiles against the A operand in LDS: k-steps of 32, six MFMAs per tile
// and k-step, two k-steps of weight fragments in registers.  `wp`: the wave's first tile.
// TKS: k-steps per tile in the plane cache (> KS when a launch phase covers part of K).
template <int NT, int KS, bool PRE = false, int TKS = KS>
struct Stream {
  static constexpr int TILE_BYTES = TKS * 3 * 1024;
  uint4 bq[2][NT][3];
  // `wp` is wave-uniform (scalar registers): scalar base + one 32-bit lane offset + immediates
  // (per-lane 64-bit addresses per tile would be hoisted out of the time loop and fill the
  // register file, as in scan.hip)
  __device__ __forceinline__ void load(int buf, const char* wp, int ks) {
    const unsigned lane_off = (threadIdx.x & 63u) * 16u;
    const char* base = wp + ks * 3072;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        bq[buf][j][p] = *reinterpret_cast<const uint4*>(base + (j * TILE_BYTES + p * 1024) + lane_off);
  }
  __device__ __forceinline__ void prefetch(const char* wp) {
    if (PRE) { load(0, wp, 0); load(1, wp, 1); }
  }
  __device__ __forceinline__ void step(int buf, const char* abuf, int ks, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
      a[p] = *reinterpret_cast<const bf16x8*>(abuf + ((ks * 3 + p) * 64 + lane) * 16);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bq[buf][j][p]);
      // six cross products, smallest terms first (as k_mfma_gemm_s3)
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc[j], 0, 0, 0);
    }
  }
  // (k-steps 0 and 1 already requested by prefetch(wp))
  __device__ __forceinline__ void run(const char* wp, const char* abuf, f32x4 (&acc)[NT], bool zero = true) {
    static_assert(KS % 2 == 0, "k-steps in pairs");
    if (!PRE) { load(0, wp, 0); load(1, wp, 1); }
    if (zero) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 2) {
      step(0, abuf, ks, acc);
      if (ks + 2 < KS) load(0, wp, ks + 2);
      step(1, abuf, ks + 1, acc);
      if (ks + 3 < KS) load(1, wp, ks + 3);
    }
  }
};


// Thread mapping of the row-wise phases: row = tid >> 4 (a row's 16 threads are 16 consecutive
// lanes of one wave: row reductions are four shuffles, no LDS, no barrier), q = tid & 15, chunks
// of 8 columns k = (q + 16 i) * 8.  Chunk (row, k) is the operand element of k-step k / 32 at
// fragment lane ((k % 32) / 8) * 16 + row.
__device__ __forceinline__ float row16_sum(float s) {
  s += __shfl_xor(s, 8, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  return s;
}
__device__ __forceinline__ void put_operand(char* abuf, int ks0, int row, int q, int i, const float (&o)[8]) {
  uint4 pl[3];
  split8(o, pl);
  const int ks = ks0 + (q >> 2) + 4 * i, fl = (q & 3) * 16 + row;
#pragma unroll
  for (int p = 0; p < 3; ++p)
    *reinterpret_cast<uint4*>(abuf + ((ks * 3 + p) * 64 + fl) * 16) = pl[p];
}
__device__ __forceinline__ float felu_(float y) { return y > 0.f ? y : fexp_(y) - 1.f; }


// One (row, group) item of the categorical draw by ONE thread: the arithmetic and the summation
// trees of latent_core.h's stats_items (butterfly max / sum over xor 16..1, Kogge-Stone inclusive
// scan, inverse-CDF count) restated serially over the C = 32 classes, so the drawn class is the
// one k_stats_fwd and the host twin dd_onehot_sample_host draw, bit for bit - at 1 / 4 of the
// instruction count (no shuffles, no idle lanes), and without the log-probability nobody reads.
__device__ __forceinline__ int draw_item32(const float (&x)[32], float u, float unimix) {
  float m = x[0];
#pragma unroll
  for (int c = 1; c < 32; ++c) m = fmaxf(m, x[c]);
  float e[32], t[16];
#pragma unroll
  for (int c = 0; c < 32; ++c) e[c] = dd_exp_det(x[c] - m);
#pragma unroll
  for (int c = 0; c < 16; ++c) t[c] = e[c] + e[c + 16];
#pragma unroll
  for (int c = 0; c < 8; ++c) t[c] = t[c] + t[c + 8];
#pragma unroll
  for (int c = 0; c < 4; ++c) t[c] = t[c] + t[c + 4];
#pragma unroll
  for (int c = 0; c < 2; ++c) t[c] = t[c] + t[c + 2];
  const float s = t[0] + t[1];
  float cdf[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) cdf[c] = dd_unimix_prob(e[c], s, unimix, 32);
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
#pragma unroll
    for (int c = 31; c >= o; --c) cdf[c] += cdf[c - o];   // (descending: cdf[c - o] is still the previous stage's value)
  }
  const float thr = dd_draw_threshold(u, cdf[31]);
  int idx = 0;
#pragma unroll
  for (int c = 0; c < 31; ++c) idx += cdf[c] <= thr ? 1 : 0;
  return idx;
}

}  // namespace
