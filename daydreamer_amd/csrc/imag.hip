// Fused imagination rollout: WorldModel.imagine (reference agent.py:234-254) as ONE persistent
// launch - the H sequential {policy(state) -> RSSM.img_step (nets.py:119-138) -> sample} steps.
//
// Why: a step of the rollout is ~23 dependent launches on N = B*T rows (2 500 at configs[1]):
// mid-size contractions at 25-60 TFLOP/s, bounded by per-launch fixed cost, H times.  The rows
// of the imagination batch never interact (agent.py:240-247: every start state rolls forward on
// its own), so a block of 16 rows can run all H steps of all layers without any grid-wide
// synchronisation: one workgroup per 16-row block, the only barriers are workgroup barriers.
//
// Per step t and row block (state: deter_t in LDS, the drawn classes of stoch_t in LDS):
//   actor MLP on [deter | stoch]   (nets.MLP nets.py:394-425 + DistLayer 'normal' :461-468)
//       layer 0: the one-hot stoch part is a GATHER of kernel rows (1.0 * w exact), the deter part
//                a K = D contraction; layers 1..: K = units contractions; LayerNorm + ELU between
//       head [mean | std], action = tanh(mean) + ((hi - lo) sigmoid(std) + lo) * eps_t
//   img_in   z1 = gather(stoch rows of W) + action @ W[S:]          (LayerNorm, ELU)  nets.py:124-126
//   GRU      z3 = [deter | x1] @ W_gru, LayerNorm over 3D, gates    nets.py:149-160
//   img_out  3 x (Linear U + LayerNorm + ELU)                        nets.py:131-133
//   img_stats + unimix + categorical draw (latent_core.h)            nets.py:134-137, 162-171
// Every buffer the backward pass and the bulk weight-gradient contractions read (per-layer
// pre-norm z, LayerNorm statistics, post-activation outputs, z3, raw statistics, traj) is written
// exactly where the per-layer launch sequence (learner.imagine_rollout) writes it.
//
// Contractions: v_mfma_f32_16x16x32_bf16 on the exact 3-way bf16 split of both fp32 operands (six
// products, fp32 accumulation - the arithmetic of gemm_core.h).  The 16-row A operand of a layer
// is built once into LDS in fragment order (LayerNorm + ELU + split by the consumer); the four
// waves split the OUTPUT columns, so there is no cross-wave reduction.  Weights stream from
// L2 / Infinity Cache as pre-split planes in fragment order (dd_imag_wprep: every (tile, k-step,
// plane) is one contiguous 1 KB block = one 16-byte load per lane), two k-steps in flight, the
// first two k-steps of the NEXT layer requested before the current layer's epilogue.
#include "latent_core.h"
#include <math.h>
#include "../../include/daydreamer_hip.h"

namespace {

constexpr float LN_EPS = 1e-3f;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct LayerP {            // a Linear + LayerNorm + ELU layer in one row space
  const char* planes;      // fragment-major bf16 planes [N/16][K/32][3][64][8]
  const float* gamma;
  const float* beta;
  float* z;                // [rows, N] pre-norm
  float* st;               // [rows, 2] mean, rstd
  float* out;              // [rows, N] post-activation
};

struct ImagArgs {
  int N, H;
  int t0, t1;              // this launch runs the policy of steps t0 .. t1 - 1 and the img_steps of those < H
  float unimix, lo, hi;
  float* traj;             // [H+1, N, F + A]
  const float* u_img;      // [H, N, G]
  const float* eps;        // [H+1, N, A]
  LayerP actor[4];
  const float* w_actor0;   // actor dense0 kernel [F, AU] fp32 (stoch rows are gathered)
  const char* head_planes; // [2A -> padded][AU]
  const float* head_bias_m;
  const float* head_bias_s;
  float* z_om;             // [M, A]
  float* z_os;             // [M, A]
  LayerP img_in;           // planes unused
  const float* w_in;       // img_in kernel [S + A, U] fp32
  LayerP gru;              // gamma / beta over 3D; z = iz3 [H*N, 3D], st = igstats; out unused
  LayerP img_out[3];
  const char* stats_planes;
  const float* stats_bias;
  float* xs;               // [H*N, S] raw statistics
  unsigned long long* dbg; // optional: time stamps of step 1 on block 0 (100 MHz wall clock)
};

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
template <int NT, int KS, bool PRE = false>
struct Stream {
  static constexpr int TILE_BYTES = KS * 3 * 1024;
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
  __device__ __forceinline__ void run(const char* wp, const char* abuf, f32x4 (&acc)[NT]) {
    static_assert(KS % 2 == 0, "k-steps in pairs");
    if (!PRE) { load(0, wp, 0); load(1, wp, 1); }
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 2) {
      step(0, abuf, ks, acc);
      if (ks + 2 < KS) load(0, wp, ks + 2);
      step(1, abuf, ks + 1, acc);
      if (ks + 3 < KS) load(1, wp, ks + 3);
    }
  }
};

// LDS row strides (floats): stride % 16 == 4 keeps the 16-row accesses of both thread mappings
// (a finished tile's elements, a row's 8-float chunks) off each other's banks
constexpr int ZS = 772;     // z buffer: up to 3 * 256 columns
constexpr int HS = 260;     // deter

// Finished tiles -> z buffer: element (row (lane >> 4) * 4 + r, column lane & 15) of tile j.
template <int NT, bool ADD>
__device__ __forceinline__ void tiles_to_z(const f32x4 (&acc)[NT], float* zb, int col0, const float* bias) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = col0 + j * 16 + (lane & 15);
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* q = zb + ((lane >> 4) * 4 + r) * ZS + col;
      *q = ADD ? (*q + acc[j][r]) : (acc[j][r] + bv);
    }
  }
}

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

// The consumer side of a LayerNorm + ELU layer: statistics of the z rows in LDS, z / statistics
// / output to global, and the next contraction's A operand (three bf16 planes, fragment order,
// k-steps ks0...) into abuf.
// LayerNorm scale / offset of this thread's chunks: requested BEFORE the layer's contraction, so
// that the (L2) latency is hidden behind it.
template <int NC>
struct Affine {
  float gm[NC][8], bt[NC][8];
  __device__ __forceinline__ void load(const float* gamma, const float* beta) {
    const int q = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      ld8(gamma + (q + 16 * i) * 8, gm[i]);
      ld8(beta + (q + 16 * i) * 8, bt[i]);
    }
  }
};

template <int NC>   // NC = columns / 128
__device__ __forceinline__ void norm_layer(const float* zb, const LayerP& L, const Affine<NC>& af, long grow,
                                           bool live, char* abuf, int ks0) {
  constexpr int NCOL = NC * 128;
  const int tid = threadIdx.x, row = tid >> 4, q = tid & 15;
  float v[NC][8];
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    ld8(zb + row * ZS + (q + 16 * i) * 8, v[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) ps += v[i][j];
  }
  const float mean = row16_sum(ps) / (float)NCOL;
  float pv = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) pv += (v[i][j] - mean) * (v[i][j] - mean);
  const float rstd = rsqrtf(row16_sum(pv) / (float)NCOL + LN_EPS);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int k = (q + 16 * i) * 8;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = felu_((v[i][j] - mean) * rstd * af.gm[i][j] + af.bt[i][j]);
    if (live) {
      st8(L.z + grow * NCOL + k, v[i]);
      st8(L.out + grow * NCOL + k, o);
    }
    put_operand(abuf, ks0, row, q, i, o);
  }
  if (live && q == 0) *reinterpret_cast<float2*>(L.st + grow * 2) = make_float2(mean, rstd);
}

// Raw rows (no norm) of an LDS buffer as A operand planes: k-steps ks0 .. ks0 + NC * 4.
template <int NC>
__device__ __forceinline__ void raw_operand(const float* src, int stride, char* abuf, int ks0) {
  const int tid = threadIdx.x, row = tid >> 4, q = tid & 15;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    float v[8];
    ld8(src + row * stride + (q + 16 * i) * 8, v);
    put_operand(abuf, ks0, row, q, i, v);
  }
}

// z[r][col] = sum over groups of W[(g * C + cls[r][g]) * ldw + col] for NCOL columns, fp32 adds in
// group order (the product of the one-hot stoch with W, exactly: 1.0 * w, and 0.0 * w for a group
// without a class).  Thread -> (r = tid >> 4, columns 4 * (tid & 15) + 64 j).  Branch-free and
// unrolled in batches of GB groups: GB * NCOL / 64 independent 16-byte loads in flight per lane.
// extra(r, col, acc): further terms before the store.
template <int NCOL, int G, int C, class Extra>
__device__ __forceinline__ void gather_rows(const float* W, long ldw, const int (*cls)[G], float* zb, Extra extra) {
  constexpr int NJ = NCOL / 64, GB = 16 / NJ, NB = G / GB;   // (two batches of 16 loads per lane in registers)
  const int tid = threadIdx.x, r = tid >> 4, c4 = (tid & 15) * 4;
  float4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 w[2][GB][NJ];
  float m[2][GB];
  auto load = [&](int buf, int g0) {
#pragma unroll
    for (int gb = 0; gb < GB; ++gb) {
      const int c = cls[r][g0 + gb];
      m[buf][gb] = c >= 0 ? 1.f : 0.f;
      const float* wr = W + (long)((g0 + gb) * C + max(c, 0)) * ldw + c4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) w[buf][gb][j] = *reinterpret_cast<const float4*>(wr + 64 * j);
    }
  };
  auto add = [&](int buf) {
#pragma unroll
    for (int gb = 0; gb < GB; ++gb)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[j].x += m[buf][gb] * w[buf][gb][j].x; acc[j].y += m[buf][gb] * w[buf][gb][j].y;
        acc[j].z += m[buf][gb] * w[buf][gb][j].z; acc[j].w += m[buf][gb] * w[buf][gb][j].w;
      }
  };
  // two batches in flight: batch b + 1 is requested before batch b is summed
  load(0, 0);
#pragma unroll 1
  for (int bb = 0; bb < NB; bb += 2) {
    load(1, (bb + 1) * GB);
    add(0);
    if (bb + 2 < NB) load(0, (bb + 2) * GB);
    add(1);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    extra(r, c4 + 64 * j, acc[j]);
    *reinterpret_cast<float4*>(zb + r * ZS + c4 + 64 * j) = acc[j];
  }
}

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

#define TS(i) if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0 && t == a.t0 + 1) a.dbg[i] = wall_clock64()

template <int D, int U, int G, int C, int A, int AU>
__global__ void __launch_bounds__(256, 1)
k_imagine_rollout(ImagArgs a) {
  constexpr int S = G * C, F = D + S, W = F + A;
  constexpr int HT = (2 * A + 15) / 16;            // head column tiles
  static_assert(D == 256 && U == 256 && AU == 512 && C == 32 && G == 32 && A <= 16, "compiled shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* zb = reinterpret_cast<float*>(smem);                       // [16][ZS]
  char* abuf = smem + 16 * ZS * 4;                                  // 16 k-steps x 3 planes x 1 KB
  float* hb = reinterpret_cast<float*>(abuf + 16 * 3 * 1024);       // [16][HS] deter
  float* wact = hb + 16 * HS;                                       // [A][U] action rows of W_in
  int (*cls)[G] = reinterpret_cast<int (*)[G]>(wact + 16 * U);      // [16][G]
  float* actb = reinterpret_cast<float*>(cls + 16);                 // [16][16] action of the step
  float* ebuf = actb + 256;                                         // [16][16] action noise of the step
  float* ubuf = ebuf + 256;                                         // [16][G] uniforms of the step's draws
  // LayerNorm scale / offset of the wide layers (read by every thread every step): actor layer l
  // at par + l * 2 * AU (scale, offset), the GRU's at par + 8 * AU
  float* par = ubuf + 16 * G;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: weight-stream bases stay in SGPRs
  const int N = a.N, H = a.H;
  const long row0 = (long)blockIdx.x * 16;
  // row-wise mapping: row = tid >> 4, q = tid & 15 (tile mapping of finished tiles: rows (lane >> 4) * 4 + r)
  const int gr = tid >> 4, gq = tid & 15;
  const long gg = min(row0 + gr, (long)N - 1);
  const bool glive = row0 + gr < N;

  // ---- prologue: state of step t0 (traj[t0]) -> deter in LDS, classes of the one-hot stoch
  for (int i = tid; i < A * U; i += 256) wact[i] = a.w_in[(long)S * U + i];
  for (int i = tid; i < AU; i += 256) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      par[l * 2 * AU + i] = a.actor[l].gamma[i];
      par[l * 2 * AU + AU + i] = a.actor[l].beta[i];
    }
  }
  for (int i = tid; i < 3 * D; i += 256) {
    par[8 * AU + i] = a.gru.gamma[i];
    par[8 * AU + 3 * D + i] = a.gru.beta[i];
  }
  {
    const float* ts = a.traj + ((long)a.t0 * N) * W;
    const float* t0 = ts + gg * W;
#pragma unroll
    for (int i = 0; i < D / 128; ++i) {
      float v[8];
      ld8(t0 + (gq + 16 * i) * 8, v);
      st8(hb + gr * HS + (gq + 16 * i) * 8, v);
    }
    // (row, group) items: sub-wave of 32 lanes per item, lane = class
    const int c = lane & 31, sub = wave * 2 + (lane >> 5);
    for (int it = sub; it < 16 * G; it += 8) {
      const int r = it & 15, g = it >> 4;
      const long rr = min(row0 + r, (long)N - 1);
      const float x = ts[rr * W + D + g * C + c];
      unsigned long long b = __ballot(x == 1.f);
      b = (b >> ((lane >> 5) * 32)) & 0xFFFFFFFFull;
      if (c == 0) cls[r][g] = b ? __ffsll((long long)b) - 1 : -1;
    }
  }
  __syncthreads();

  Stream<8, 8> sA0;     // actor layer 0, deter part: K = D
  Stream<8, 16> sA;     // actor layers 1..3: K = AU
  Stream<6, 16> sG;     // GRU: K = D + U, 3D columns in two passes of 6 tiles per wave
  Stream<4, 8> sO;      // img_out: K = D or U, U columns
  Stream<8, 8> sS;      // img_stats: K = U, half of the S columns per pass

  for (int t = a.t0; t < a.t1; ++t) {
    const long mrow = (long)t * N + gg;         // this thread's row in the [M, ..] / [H*N, ..] buffers
    float* trow = a.traj + ((long)t * N) * W;

    TS(0);
    // the step's noise: requested now, used after the actor / the img_step (latency hidden)
    {
      const float e = gq < A ? a.eps[mrow * A + gq] : 0.f;
      float2 u2 = make_float2(0.f, 0.f);
      if (t < H) u2 = *reinterpret_cast<const float2*>(a.u_img + mrow * G + gq * 2);
      ebuf[gr * 16 + gq] = e;
      *reinterpret_cast<float2*>(ubuf + gr * G + gq * 2) = u2;
    }
    // ================= actor on [deter_t | stoch_t]
    const char* wp0 = a.actor[0].planes + (long)(wave * 8) * Stream<8, 8>::TILE_BYTES;
    sA0.prefetch(wp0);
    raw_operand<D / 128>(hb, HS, abuf, 0);
    gather_rows<AU, G, C>(a.w_actor0 + (long)D * AU, AU, cls, zb, [](int, int, float4&) {});
    __syncthreads();
    TS(1);
    {
      f32x4 acc[8];
      sA0.run(wp0, abuf, acc);
      const char* wp1 = a.actor[1].planes + (long)(wave * 8) * Stream<8, 16>::TILE_BYTES;
      sA.prefetch(wp1);
      tiles_to_z<8, true>(acc, zb, wave * 128, nullptr);
    }
    __syncthreads();
    TS(2);
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
      // layer l's rows are complete in zb: normalise -> operand of the next contraction
      {
        Affine<AU / 128> afA;
        afA.load(par + l * 2 * AU, par + l * 2 * AU + AU);
        norm_layer<AU / 128>(zb, a.actor[l], afA, mrow, glive, abuf, 0);
      }
      __syncthreads();
      TS(3 + 2 * l);
      if (l < 3) {
        const char* wp = a.actor[l + 1].planes + (long)(wave * 8) * Stream<8, 16>::TILE_BYTES;
        f32x4 acc[8];
        sA.run(wp, abuf, acc);
        if (l < 2) {
          const char* wn = a.actor[l + 2].planes + (long)(wave * 8) * Stream<8, 16>::TILE_BYTES;
          sA.prefetch(wn);
        }
        tiles_to_z<8, false>(acc, zb, wave * 128, nullptr);
        __syncthreads();
        TS(4 + 2 * l);
      }
    }
    // head [mean | std]: the waves split K (4 k-steps each), partial tiles through the z buffer
    {
      uint4 bq[HT][4][3];
#pragma unroll
      for (int j = 0; j < HT; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            bq[j][ks][p] = *reinterpret_cast<const uint4*>(
                a.head_planes + (long)j * (16 * 3072) + (wave * 4 + ks) * 3072 + p * 1024 + (unsigned)lane * 16u);
      f32x4 acc[HT];
#pragma unroll
      for (int j = 0; j < HT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 av[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          av[p] = *reinterpret_cast<const bf16x8*>(abuf + (((wave * 4 + ks) * 3 + p) * 64 + lane) * 16);
#pragma unroll
        for (int j = 0; j < HT; ++j) {
          bf16x8 b[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bq[j][ks][p]);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], b[0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[2], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], b[1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], b[0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[0], acc[j], 0, 0, 0);
        }
      }
      // partial of wave w, head column c -> zb[row][w * 32 + c]
#pragma unroll
      for (int j = 0; j < HT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          zb[((lane >> 4) * 4 + r) * ZS + wave * 32 + j * 16 + (lane & 15)] = acc[j][r];
    }
    __syncthreads();
    TS(10);
    // action of step t
    if (gq < A) {
      const float* zr = zb + gr * ZS;
      const float om = ((zr[gq] + zr[32 + gq]) + zr[64 + gq]) + zr[96 + gq] + a.head_bias_m[gq];
      const float os = ((zr[A + gq] + zr[32 + A + gq]) + zr[64 + A + gq]) + zr[96 + A + gq] + a.head_bias_s[gq];
      const float sd = (a.hi - a.lo) * sigmoidf_(os) + a.lo;
      const float act = tanhf(om) + sd * ebuf[gr * 16 + gq];
      actb[gr * 16 + gq] = act;
      if (glive) {
        a.z_om[mrow * A + gq] = om;
        a.z_os[mrow * A + gq] = os;
        trow[gg * W + F + gq] = act;
      }
    }
    if (t == H) break;
    __syncthreads();
    TS(11);

    // ================= img_step: img_in (gather + action columns), LayerNorm, ELU
    float* tnext = a.traj + ((long)(t + 1) * N) * W;
    const char* wpg = a.gru.planes + (long)(wave * 12) * Stream<6, 16>::TILE_BYTES;
    sG.prefetch(wpg);
    Affine<U / 128> afU;
    afU.load(a.img_in.gamma, a.img_in.beta);
    gather_rows<U, G, C>(a.w_in, U, cls, zb, [&](int r, int col, float4& acc) {
#pragma unroll
      for (int j = 0; j < A; ++j) {
        const float av = actb[r * 16 + j];
        const float4 w = *reinterpret_cast<const float4*>(wact + j * U + col);
        acc.x += av * w.x; acc.y += av * w.y; acc.z += av * w.z; acc.w += av * w.w;
      }
    });
    raw_operand<D / 128>(hb, HS, abuf, 0);                 // [deter_t | x1]: k-steps 0..7 = deter
    __syncthreads();
    TS(12);
    norm_layer<U / 128>(zb, a.img_in, afU, mrow, glive, abuf, D / 32);
    __syncthreads();
    TS(13);
    // ================= GRU contraction, LayerNorm over 3D, gates
#pragma unroll 1
    for (int hp = 0; hp < 2; ++hp) {
      f32x4 acc[6];
      sG.run(wpg + (long)(hp * 6) * Stream<6, 16>::TILE_BYTES, abuf, acc);
      if (hp == 0) sG.prefetch(wpg + (long)6 * Stream<6, 16>::TILE_BYTES);
      else sO.prefetch(a.img_out[0].planes + (long)(wave * 4) * Stream<4, 8>::TILE_BYTES);
      tiles_to_z<6, false>(acc, zb, wave * 192 + hp * 96, nullptr);
    }
    __syncthreads();
    TS(14);
    {
      constexpr int NC = 3 * D / 128, ND = D / 128;   // chunks per thread: [reset | cand | update] x ND
      float v[NC][8];
      Affine<NC> afG;
      afG.load(par + 8 * AU, par + 8 * AU + 3 * D);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        ld8(zb + gr * ZS + (gq + 16 * i) * 8, v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ps += v[i][j];
      }
      const float mean = row16_sum(ps) / (float)(3 * D);
      float pv = 0.f;
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) pv += (v[i][j] - mean) * (v[i][j] - mean);
      const float rstd = rsqrtf(row16_sum(pv) / (float)(3 * D) + LN_EPS);
      if (glive) {
#pragma unroll
        for (int i = 0; i < NC; ++i) st8(a.gru.z + mrow * (3 * D) + (gq + 16 * i) * 8, v[i]);
        if (gq == 0) *reinterpret_cast<float2*>(a.gru.st + mrow * 2) = make_float2(mean, rstd);
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int d = (gq + 16 * i) * 8;
        float hp[8], hn[8];
        ld8(hb + gr * HS + d, hp);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yr = (v[i][j] - mean) * rstd * afG.gm[i][j] + afG.bt[i][j];
          const float yc = (v[i + ND][j] - mean) * rstd * afG.gm[i + ND][j] + afG.bt[i + ND][j];
          const float yu = (v[i + 2 * ND][j] - mean) * rstd * afG.gm[i + 2 * ND][j] + afG.bt[i + 2 * ND][j];
          const float r = sigmoidf_(yr);
          const float cand = tanhf(r * yc);
          const float u = sigmoidf_(yu - 1.f);
          hn[j] = u * cand + (1.f - u) * hp[j];
        }
        st8(hb + gr * HS + d, hn);
        if (glive) st8(tnext + gg * W + d, hn);
        put_operand(abuf, 0, gr, gq, i, hn);
      }
    }
    __syncthreads();
    TS(15);
    // ================= img_out 0..2
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      const char* wp = a.img_out[l].planes + (long)(wave * 4) * Stream<4, 8>::TILE_BYTES;
      afU.load(a.img_out[l].gamma, a.img_out[l].beta);
      f32x4 acc[4];
      sO.run(wp, abuf, acc);
      if (l < 2) {
        const char* wn = a.img_out[l + 1].planes + (long)(wave * 4) * Stream<4, 8>::TILE_BYTES;
        sO.prefetch(wn);
      } else {
        sS.prefetch(a.stats_planes + (long)(wave * 8) * Stream<8, 8>::TILE_BYTES);
      }
      tiles_to_z<4, false>(acc, zb, wave * 64, nullptr);
      __syncthreads();
      TS(16 + 2 * l);
      norm_layer<U / 128>(zb, a.img_out[l], afU, mrow, glive, abuf, 0);
      __syncthreads();
      TS(17 + 2 * l);
    }
    // ================= img_stats + draw: S columns in passes of 512 (16 groups)
    constexpr int NPASS = S / 512;
#pragma unroll 1
    for (int ps_ = 0; ps_ < NPASS; ++ps_) {
      const char* wp = a.stats_planes + (long)(ps_ * 32 + wave * 8) * Stream<8, 8>::TILE_BYTES;
      f32x4 acc[8];
      sS.run(wp, abuf, acc);
      if (ps_ + 1 < NPASS)
        sS.prefetch(a.stats_planes + (long)((ps_ + 1) * 32 + wave * 8) * Stream<8, 8>::TILE_BYTES);
      tiles_to_z<8, false>(acc, zb, wave * 128, a.stats_bias + ps_ * 512);
      __syncthreads();
      TS(22 + 2 * ps_);
      // 16 rows x 16 groups = 256 items, one per thread: row = tid & 15 (a quarter-wave's 16 rows
      // lie 4 banks apart), group = tid >> 4
      {
        const int r = tid & 15, gl = tid >> 4, g = ps_ * 16 + gl;
        float x[32];
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
          const float4 q = *reinterpret_cast<const float4*>(zb + r * ZS + gl * C + c);
          x[c] = q.x; x[c + 1] = q.y; x[c + 2] = q.z; x[c + 3] = q.w;
        }
        const int idx = draw_item32(x, ubuf[r * G + g], a.unimix);
        cls[r][g] = idx;
        if (row0 + r < N) {
          const long gw = row0 + r;
          float* xo = a.xs + ((long)t * N + gw) * S + g * C;
          float* so = tnext + gw * W + D + g * C;
#pragma unroll
          for (int c = 0; c < 32; c += 4) {
            *reinterpret_cast<float4*>(xo + c) = make_float4(x[c], x[c + 1], x[c + 2], x[c + 3]);
            *reinterpret_cast<float4*>(so + c) = make_float4(idx == c ? 1.f : 0.f, idx == c + 1 ? 1.f : 0.f,
                                                             idx == c + 2 ? 1.f : 0.f, idx == c + 3 ? 1.f : 0.f);
          }
        }
      }
      __syncthreads();
      TS(23 + 2 * ps_);
    }
  }
}

// W [K, n] fp32 (row stride ld) -> fragment-major bf16 planes of an [K, Npad] operand, columns
// col0 .. col0 + n (other columns of the destination are left untouched: zero-initialised pads).
__global__ void k_imag_wprep(const float* __restrict__ W, long ld, int K, int n, int col0,
                             char* __restrict__ planes) {
  const int KS = K / 32;
  const long total = (long)((col0 + n + 15) / 16 - col0 / 16) * KS * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long tk = i >> 6;
    const int ks = (int)(tk % KS), tile = col0 / 16 + (int)(tk / KS);
    const int col = tile * 16 + (lane & 15) - col0;
    if (col < 0 || col >= n) continue;
    const int k0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(long)(k0 + j) * ld + col];
    uint4 pl[3];
    split8(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint4*>(planes + (((long)tile * KS + ks) * 3 + p) * 1024 + lane * 16) = pl[p];
  }
}

}  // namespace

extern "C" int dd_imag_wprep(const float* W, long ld, int K, int n, int col0, void* planes, void* stream) {
  DD_REQUIRE(K % 32 == 0 && n >= 1 && col0 >= 0, "dd_imag_wprep: K multiple of 32");
  const long total = (long)((col0 + n + 15) / 16 - col0 / 16) * (K / 32) * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  k_imag_wprep<<<blocks, 256, 0, (hipStream_t)stream>>>(W, ld, K, n, col0, (char*)planes);
  DD_CHECK_LAUNCH("dd_imag_wprep");
  return 0;
}

// compiled shapes (deter, units, groups, classes, action dims, actor units); actor layers = 4,
// prior layers = 3, continuous actions
#define DD_IMAG_SHAPES(X) X(256, 256, 32, 32, 16, 512) X(256, 256, 32, 32, 6, 512)

extern "C" int dd_imagine_rollout_supported(int D, int U, int G, int C, int A, int actor_units,
                                            int actor_layers, int prior_layers, int discrete) {
  if (actor_layers != 4 || prior_layers != 3 || discrete) return 0;
#define X(d, u, g, c, a_, au) if (D == d && U == u && G == g && C == c && A == a_ && actor_units == au) return 1;
  DD_IMAG_SHAPES(X)
#undef X
  return 0;
}

namespace {
constexpr int IMAG_LDS = 16 * ZS * 4 + 16 * 3 * 1024 + 16 * HS * 4 + 16 * 256 * 4 + 16 * 32 * 4 + 256 * 4 + 256 * 4 + 16 * 32 * 4 +
                         (8 * 512 + 6 * 256) * 4;
}

// ptrs (device pointers, in this order):
//   0 traj  1 u_img  2 eps
//   3.. actor layer l = 0..3: planes, gamma, beta, z, stats, out      (6 each -> 3..26)
//   27 actor dense0 kernel (fp32)  28 head planes  29 head bias mean  30 head bias std
//   31 z mean  32 z std
//   33 img_in kernel (fp32)  34 gamma  35 beta  36 z  37 stats  38 out
//   39 gru planes  40 gamma  41 beta  42 z3  43 gstats
//   44.. img_out l = 0..2: planes, gamma, beta, z, stats, out           (6 each -> 44..61)
//   62 stats planes  63 stats bias  64 raw statistics  [65 optional: 32 x u64 time stamps]
extern "C" int dd_imagine_rollout_fwd(int N, int H, int t0, int t1, int D, int U, int G, int C, int A,
                                      int actor_units, float unimix, float lo, float hi,
                                      const void* const* p, int n_ptrs, void* stream) {
  DD_REQUIRE(dd_imagine_rollout_supported(D, U, G, C, A, actor_units, 4, 3, 0), "dd_imagine_rollout_fwd: unsupported shape");
  DD_REQUIRE((n_ptrs == 65 || n_ptrs == 66) && N >= 1 && H >= 1, "dd_imagine_rollout_fwd: 65 pointers (+ optional time-stamp buffer)");
  DD_REQUIRE(0 <= t0 && t0 < t1 && t1 <= H + 1, "dd_imagine_rollout_fwd: 0 <= t0 < t1 <= H + 1");
  ImagArgs a;
  a.N = N; a.H = H; a.t0 = t0; a.t1 = t1; a.unimix = unimix; a.lo = lo; a.hi = hi;
  a.traj = (float*)p[0]; a.u_img = (const float*)p[1]; a.eps = (const float*)p[2];
  auto layer = [&](int i) {
    LayerP L;
    L.planes = (const char*)p[i]; L.gamma = (const float*)p[i + 1]; L.beta = (const float*)p[i + 2];
    L.z = (float*)p[i + 3]; L.st = (float*)p[i + 4]; L.out = (float*)p[i + 5];
    return L;
  };
  for (int l = 0; l < 4; ++l) a.actor[l] = layer(3 + 6 * l);
  a.w_actor0 = (const float*)p[27]; a.head_planes = (const char*)p[28];
  a.head_bias_m = (const float*)p[29]; a.head_bias_s = (const float*)p[30];
  a.z_om = (float*)p[31]; a.z_os = (float*)p[32];
  a.w_in = (const float*)p[33];
  a.img_in.planes = nullptr; a.img_in.gamma = (const float*)p[34]; a.img_in.beta = (const float*)p[35];
  a.img_in.z = (float*)p[36]; a.img_in.st = (float*)p[37]; a.img_in.out = (float*)p[38];
  a.gru.planes = (const char*)p[39]; a.gru.gamma = (const float*)p[40]; a.gru.beta = (const float*)p[41];
  a.gru.z = (float*)p[42]; a.gru.st = (float*)p[43]; a.gru.out = nullptr;
  for (int l = 0; l < 3; ++l) a.img_out[l] = layer(44 + 6 * l);
  a.stats_planes = (const char*)p[62]; a.stats_bias = (const float*)p[63]; a.xs = (float*)p[64];
  a.dbg = n_ptrs == 66 ? (unsigned long long*)p[65] : nullptr;
  const int blocks = (N + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  bool launched = false;
#define X(d, u, g, c, a_, au)                                                                    \
  if (!launched && D == d && U == u && G == g && C == c && A == a_ && actor_units == au) {       \
    static bool attr = false;                                                                    \
    if (!attr) {                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)k_imagine_rollout<d, u, g, c, a_, au>,     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, IMAG_LDS);  \
      if (e != hipSuccess) { dd_set_error("dd_imagine_rollout_fwd(attr)", e); return (int)e; }   \
      attr = true;                                                                               \
    }                                                                                            \
    k_imagine_rollout<d, u, g, c, a_, au><<<blocks, 256, IMAG_LDS, st>>>(a);                     \
    launched = true;                                                                             \
  }
  DD_IMAG_SHAPES(X)
#undef X
  DD_CHECK_LAUNCH("dd_imagine_rollout_fwd");
  return 0;
}
