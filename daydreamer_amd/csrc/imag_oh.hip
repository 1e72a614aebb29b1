// Fused imagination rollout for ONE-HOT action spaces at deter = units = 512 (the xarm / ur5
// blocks of the reference, configs.yaml:245-295; BASELINE configs[2], configs[3]):
// WorldModel.imagine (reference agent.py:234-254) as ONE persistent launch, forward only -
// actor_grad 'reinforce' (agent.py:357-358) does not differentiate the dynamics, so there is no
// reverse kernel to pair it with: the actor's backward pass needs the actor's own activations,
// the drawn actions and their log-probabilities, all of which this launch leaves where the
// per-layer launch sequence (learner.imagine_rollout) leaves them.
//
// Same decomposition as imag.hip: one workgroup (4 waves) per 16-row block through all H steps,
// no grid synchronisation, weights streamed as pre-split fragment-major bf16 planes (dd_imag_wprep),
// six products per fp32 product.  What differs, because nothing fits twice at 512:
//   * the A operand of the GRU contraction (K = deter + units = 1024) is 96 KB of planes, so there
//     is no z buffer of its own in LDS.  Every other contraction has K = 512 and leaves the upper
//     half of the operand buffer free: its finished tiles go there (16 x 512 floats), as in
//     imag.hip.  The GRU's 16 x 1536 tiles go straight to its pre-norm buffer in global memory
//     (which the launch sequence writes anyway), the workgroup barrier orders them, and the gate
//     phase reads its rows back from L2 (96 KB per step against 25 MB of weight planes);
//   * the one-hot stoch part of actor layer 0 and of img_in is gathered INSIDE that row-wise phase
//     (thread = 8-column chunks of one row, the chunks it normalises), so a gathered layer needs no
//     extra pass and no extra barrier; the one-hot action is one more gathered row of img_in;
//   * the policy head is categorical: unimix softmax + inverse-CDF draw over A <= 8 classes, the
//     arithmetic and summation trees of latent_core.h's stats_items restated serially
//     (draw_item_small), so the drawn action is the one k_stats_fwd draws, bit for bit.
// Per-CU stream: 23.6 MB of planes + 2 MB of gathered rows per step (imag.hip streams 12 MB):
// ~200 us per step at the CU's 115-140 GB/s (profiles/r05_stream_probe_per_cu.txt).
#include "imag_core.h"
#include "../../include/daydreamer_hip.h"

namespace {

struct OhLayer {           // a Linear + LayerNorm + ELU layer in one row space
  const char* planes;      // fragment-major bf16 planes [N/16][K/32][3][64][8]
  const float* gamma;
  const float* beta;
  float* z;                // [rows, N] pre-norm
  float* st;               // [rows, 2] mean, rstd
  float* out;              // [rows, N] post-activation
};

struct ImagOhArgs {
  int N, H;
  int t0, t1;              // this launch runs the policy of steps t0 .. t1 - 1 and the img_steps of those < H
  int W;                   // floats per trajectory row (F + A padded to a multiple of four)
  float unimix, act_unimix;
  float* traj;             // [H+1, N, W]
  const float* u_img;      // [H, N, G]
  const float* u_act;      // [H+1, N]
  OhLayer actor[4];
  const float* w_actor0;   // actor dense0 kernel [F, AU] fp32 (stoch rows are gathered)
  const char* head_planes; // [A -> 16][AU]
  const float* head_bias;
  float* z_head;           // [M, A] raw logits of the policy
  float* alogit;           // [M, A] normalised log-probabilities
  OhLayer img_in;          // planes unused
  const float* w_in;       // img_in kernel [S + A, U] fp32
  OhLayer gru;             // gamma / beta over 3D; z = iz3 [H*N, 3D], st = igstats; out unused
  OhLayer img_out[3];
  const char* stats_planes;
  const float* stats_bias;
  float* xs;               // [H*N, S] raw statistics
};

// Finished tiles -> a [rows, ld] buffer in global memory: element (row (lane >> 4) * 4 + r,
// column lane & 15) of tile j (64-byte segments per row; rows past N are not written).
template <int NT>
__device__ __forceinline__ void tiles_to_global(const f32x4 (&acc)[NT], float* zrows, long ld, int col0,
                                                const float* bias, long row0, int N) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = col0 + j * 16 + (lane & 15);
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = (lane >> 4) * 4 + r;
      if (row0 + rr < N) zrows[(long)rr * ld + col] = acc[j][r] + bv;
    }
  }
}

// Finished tiles -> the LDS z buffer (the free upper half of the operand buffer), row stride ZSO.
constexpr int ZSO = 516;    // floats: stride % 16 == 4 keeps tile and row accesses off each other's banks
template <int NT>
__device__ __forceinline__ void tiles_to_zb(const f32x4 (&acc)[NT], float* zb, int col0, const float* bias) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = col0 + j * 16 + (lane & 15);
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) zb[((lane >> 4) * 4 + r) * ZSO + col] = acc[j][r] + bv;
  }
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

// v[i][.] += sum over groups of W[(g * C + cls[row][g]) * ldw + (q + 16 i) * 8 + .] - the product
// of the one-hot stoch with W, exactly (1.0 * w; a group without a class contributes 0.0 * w) -
// for this thread's NC chunks, fp32 adds in group order.  Branch-free, batches of GB groups,
// two batches (2 * GB * NC * 2 sixteen-byte loads per lane) in flight.
template <int NC, int G, int C>
__device__ __forceinline__ void gather_chunks(const float* W, long ldw, const int (*cls)[G], int row, int q,
                                              float (&v)[NC][8]) {
  constexpr int GB = 2, NB = G / GB;
  float4 w[2][GB][NC][2];
  float m[2][GB];
  auto load = [&](int buf, int g0) {
#pragma unroll
    for (int gb = 0; gb < GB; ++gb) {
      const int c = cls[row][g0 + gb];
      m[buf][gb] = c >= 0 ? 1.f : 0.f;
      const float* wr = W + (long)((g0 + gb) * C + max(c, 0)) * ldw + q * 8;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        w[buf][gb][i][0] = *reinterpret_cast<const float4*>(wr + i * 128);
        w[buf][gb][i][1] = *reinterpret_cast<const float4*>(wr + i * 128 + 4);
      }
    }
  };
  auto add = [&](int buf) {
#pragma unroll
    for (int gb = 0; gb < GB; ++gb)
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const float4 a = w[buf][gb][i][0], b = w[buf][gb][i][1];
        const float mm = m[buf][gb];
        v[i][0] += mm * a.x; v[i][1] += mm * a.y; v[i][2] += mm * a.z; v[i][3] += mm * a.w;
        v[i][4] += mm * b.x; v[i][5] += mm * b.y; v[i][6] += mm * b.z; v[i][7] += mm * b.w;
      }
  };
  load(0, 0);
#pragma unroll 1
  for (int bb = 0; bb < NB; bb += 2) {
    load(1, (bb + 1) * GB);
    add(0);
    if (bb + 2 < NB) load(0, (bb + 2) * GB);
    add(1);
  }
}

// The row-wise phase of a LayerNorm + ELU layer: this thread's NC chunks of row `grow` of the
// pre-norm buffer (the contraction's tiles, or zeros for a gathered-only layer) + `pre` (the
// gathered part, summed first) -> statistics over the row's 16 threads, z / statistics / output
// to global, the next contraction's A operand planes (k-steps ks0 ..) into abuf.
template <int NC>
__device__ __forceinline__ void norm_rows(float (&v)[NC][8], const OhLayer& L, long grow, bool live,
                                          char* abuf, int ks0) {
  constexpr int NCOL = NC * 128;
  const int tid = threadIdx.x, row = tid >> 4, q = tid & 15;
  float gm[NC][8], bt[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    ld8(L.gamma + (q + 16 * i) * 8, gm[i]);
    ld8(L.beta + (q + 16 * i) * 8, bt[i]);
  }
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) ps += v[i][j];
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
    for (int j = 0; j < 8; ++j) o[j] = felu_((v[i][j] - mean) * rstd * gm[i][j] + bt[i][j]);
    if (live) {
      st8(L.z + grow * NCOL + k, v[i]);
      st8(L.out + grow * NCOL + k, o);
    }
    put_operand(abuf, ks0, row, q, i, o);
  }
  if (live && q == 0) *reinterpret_cast<float2*>(L.st + grow * 2) = make_float2(mean, rstd);
}

// One (row) item of the policy's categorical draw by ONE thread: stats_items<LW = 8> of
// latent_core.h restated serially over A <= 8 classes (butterfly max / sum over xor 4, 2, 1,
// Kogge-Stone inclusive scan, inverse-CDF count): the class k_stats_fwd draws, bit for bit, and
// its normalised log-probabilities.
template <int A>
__device__ __forceinline__ int draw_item_small(const float (&x)[8], float u, float unimix, float (&lg)[8]) {
  static_assert(A >= 2 && A <= 8, "one sub-wave of 8 lanes");
  float m = x[0];
#pragma unroll
  for (int c = 1; c < A; ++c) m = fmaxf(m, x[c]);
  float e[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) e[c] = c < A ? dd_exp_det(x[c] - m) : 0.f;
  const float t0 = e[0] + e[4], t1 = e[1] + e[5], t2 = e[2] + e[6], t3 = e[3] + e[7];
  const float s = (t0 + t2) + (t1 + t3);
  float cdf[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float pm = c < A ? dd_unimix_prob(e[c], s, unimix, A) : 0.f;
    cdf[c] = pm;
    lg[c] = c < A ? (unimix > 0.f ? logf(pm) : (x[c] - m) - logf(s)) : 0.f;
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
#pragma unroll
    for (int c = 7; c >= o; --c) cdf[c] += cdf[c - o];   // (descending: cdf[c - o] is still the previous stage's value)
  }
  const float thr = dd_draw_threshold(u, cdf[A - 1]);
  int idx = 0;
#pragma unroll
  for (int c = 0; c < A - 1; ++c) idx += cdf[c] <= thr ? 1 : 0;
  return idx;
}

constexpr int HSO = 516;    // LDS row stride of deter (floats): 512 + 4

template <int D, int U, int G, int C, int A, int AU>
__global__ void __launch_bounds__(256, 1)
k_imagine_rollout_oh(ImagOhArgs a) {
  constexpr int S = G * C, F = D + S;
  static_assert(D == 512 && U == 512 && AU == 512 && C == 32 && G == 32 && A <= 8, "compiled shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* abuf = smem;                                                // 32 k-steps x 3 planes x 1 KB
  float* zb = reinterpret_cast<float*>(abuf + 16 * 3 * 1024);       // [16][ZSO]: k-steps 16..31, free while K = 512
  float* hb = reinterpret_cast<float*>(abuf + 32 * 3 * 1024);       // [16][HSO] deter
  int (*cls)[G] = reinterpret_cast<int (*)[G]>(hb + 16 * HSO);      // [16][G]
  float* hz = reinterpret_cast<float*>(cls + 16);                   // [16][64] K-split partials of the head
  float* ubuf = hz + 16 * 64;                                       // [16][G] uniforms of the step's latent draws
  float* uact = ubuf + 16 * G;                                      // [16] uniform of the step's action draw
  int* aidx = reinterpret_cast<int*>(uact + 16);                    // [16] drawn action of the step

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: weight-stream bases stay in SGPRs
  const int N = a.N, H = a.H, W = a.W;
  const long row0 = (long)blockIdx.x * 16;
  const int gr = tid >> 4, gq = tid & 15;
  const long gg = min(row0 + gr, (long)N - 1);
  const bool glive = row0 + gr < N;

  // ---- prologue: state of step t0 (traj[t0]) -> deter in LDS, classes of the one-hot stoch
  {
    const float* ts = a.traj + ((long)a.t0 * N) * W;
    const float* t0 = ts + gg * W;
#pragma unroll
    for (int i = 0; i < D / 128; ++i) {
      float v[8];
      ld8(t0 + (gq + 16 * i) * 8, v);
      st8(hb + gr * HSO + (gq + 16 * i) * 8, v);
    }
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

  using S8 = Stream<8, 16>;     // K = 512, eight column tiles per wave (N = 512 per pass)
  using SGR = Stream<6, 32>;    // GRU: K = D + U = 1024, 3D columns in four passes of 6 tiles per wave
  S8 s8;
  SGR sgr;

  for (int t = a.t0; t < a.t1; ++t) {
    const long mrow = (long)t * N + gg;         // this thread's row in the [M, ..] / [H*N, ..] buffers
    const long mblk = (long)t * N + row0;       // the block's first row there
    float* trow = a.traj + ((long)t * N) * W;

    // the step's noise: requested now, used after the actor / the img_step
    {
      float2 u2 = make_float2(0.f, 0.f);
      if (t < H) u2 = *reinterpret_cast<const float2*>(a.u_img + mrow * G + gq * 2);
      *reinterpret_cast<float2*>(ubuf + gr * G + gq * 2) = u2;
      if (gq == 0) uact[gr] = a.u_act[mrow];
    }
    // ================= actor on [deter_t | stoch_t]
    raw_operand<D / 128>(hb, HSO, abuf, 0);
    __syncthreads();
    {
      f32x4 acc[8];
      s8.run(a.actor[0].planes + (long)(wave * 8) * S8::TILE_BYTES, abuf, acc);
      tiles_to_zb<8>(acc, zb, wave * 128, nullptr);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
      {
        float v[AU / 128][8];
        if (l == 0) {
          // layer 0: the gathered stoch part first (as imag.hip: gather, then the deter contraction)
          float zc[AU / 128][8];
#pragma unroll
          for (int i = 0; i < AU / 128; ++i) {
            ld8(zb + gr * ZSO + (gq + 16 * i) * 8, zc[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
          }
          gather_chunks<AU / 128, G, C>(a.w_actor0 + (long)D * AU, AU, cls, gr, gq, v);
#pragma unroll
          for (int i = 0; i < AU / 128; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] += zc[i][j];
        } else {
#pragma unroll
          for (int i = 0; i < AU / 128; ++i) ld8(zb + gr * ZSO + (gq + 16 * i) * 8, v[i]);
        }
        norm_rows<AU / 128>(v, a.actor[l], mrow, glive, abuf, 0);
      }
      __syncthreads();
      if (l < 3) {
        f32x4 acc[8];
        s8.run(a.actor[l + 1].planes + (long)(wave * 8) * S8::TILE_BYTES, abuf, acc);
        tiles_to_zb<8>(acc, zb, wave * 128, nullptr);
        __syncthreads();
      }
    }
    // policy head: the waves split K (4 k-steps each), partial tiles through hz
    {
      uint4 bq[4][3];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bq[ks][p] = *reinterpret_cast<const uint4*>(
              a.head_planes + (wave * 4 + ks) * 3072 + p * 1024 + (unsigned)lane * 16u);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 av[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          av[p] = *reinterpret_cast<const bf16x8*>(abuf + (((wave * 4 + ks) * 3 + p) * 64 + lane) * 16);
          b[p] = __builtin_bit_cast(bf16x8, bq[ks][p]);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[0], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) hz[((lane >> 4) * 4 + r) * 64 + wave * 16 + (lane & 15)] = acc[r];
    }
    __syncthreads();
    // action of step t: unimix softmax over the A logits, inverse-CDF draw (one thread per row)
    if (gq == 0) {
      const float* zr = hz + gr * 64;
      float x[8], lg[8];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        x[c] = c < A ? ((zr[c] + zr[16 + c]) + zr[32 + c]) + zr[48 + c] + a.head_bias[c] : 0.f;
      const int idx = draw_item_small<A>(x, uact[gr], a.act_unimix, lg);
      aidx[gr] = idx;
      if (glive) {
#pragma unroll
        for (int c = 0; c < A; ++c) {
          a.z_head[mrow * A + c] = x[c];
          a.alogit[mrow * A + c] = lg[c];
          trow[gg * W + F + c] = idx == c ? 1.f : 0.f;
        }
      }
    }
    if (t == H) break;
    __syncthreads();

    // ================= img_step: img_in = gathered stoch rows + the action's row, LayerNorm, ELU
    float* tnext = a.traj + ((long)(t + 1) * N) * W;
    {
      float v[U / 128][8];
      // the one-hot action's row of W_in (rows S .. S + A) starts the sum, the stoch groups follow
#pragma unroll
      for (int i = 0; i < U / 128; ++i) ld8(a.w_in + (long)(S + aidx[gr]) * U + (gq + 16 * i) * 8, v[i]);
      float vs[U / 128][8];
#pragma unroll
      for (int i = 0; i < U / 128; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) vs[i][j] = 0.f;
      gather_chunks<U / 128, G, C>(a.w_in, U, cls, gr, gq, vs);
#pragma unroll
      for (int i = 0; i < U / 128; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = vs[i][j] + v[i][j];
      raw_operand<D / 128>(hb, HSO, abuf, 0);                 // [deter_t | x1]: k-steps 0..15 = deter
      norm_rows<U / 128>(v, a.img_in, mrow, glive, abuf, D / 32);
    }
    __syncthreads();
    // ================= GRU contraction (K = 1024), LayerNorm over 3D, gates
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
      f32x4 acc[6];
      sgr.run(a.gru.planes + (long)(wave * 24 + p * 6) * SGR::TILE_BYTES, abuf, acc);
      tiles_to_global<6>(acc, a.gru.z + mblk * (3 * D), 3 * D, wave * 384 + p * 96, nullptr, row0, N);
    }
    __syncthreads();
    {
      constexpr int NC = 3 * D / 128, ND = D / 128;   // chunks per thread: [reset | cand | update] x ND
      float v[NC][8];
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        ld8(a.gru.z + mrow * (3 * D) + (gq + 16 * i) * 8, v[i]);
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
      if (glive && gq == 0) *reinterpret_cast<float2*>(a.gru.st + mrow * 2) = make_float2(mean, rstd);
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int d = (gq + 16 * i) * 8;
        float hp[8], hn[8], g0[8], g1[8], g2[8], b0[8], b1[8], b2[8];
        ld8(hb + gr * HSO + d, hp);
        ld8(a.gru.gamma + d, g0); ld8(a.gru.gamma + D + d, g1); ld8(a.gru.gamma + 2 * D + d, g2);
        ld8(a.gru.beta + d, b0); ld8(a.gru.beta + D + d, b1); ld8(a.gru.beta + 2 * D + d, b2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yr = (v[i][j] - mean) * rstd * g0[j] + b0[j];
          const float yc = (v[i + ND][j] - mean) * rstd * g1[j] + b1[j];
          const float yu = (v[i + 2 * ND][j] - mean) * rstd * g2[j] + b2[j];
          const float r = sigmoidf_(yr);
          const float cand = tanhf(r * yc);
          const float u = sigmoidf_(yu - 1.f);
          hn[j] = u * cand + (1.f - u) * hp[j];
        }
        st8(hb + gr * HSO + d, hn);
        if (glive) st8(tnext + gg * W + d, hn);
        put_operand(abuf, 0, gr, gq, i, hn);
      }
    }
    __syncthreads();
    // ================= img_out 0..2
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      {
        f32x4 acc[8];
        s8.run(a.img_out[l].planes + (long)(wave * 8) * S8::TILE_BYTES, abuf, acc);
        tiles_to_zb<8>(acc, zb, wave * 128, nullptr);
      }
      __syncthreads();
      {
        float v[U / 128][8];
#pragma unroll
        for (int i = 0; i < U / 128; ++i) ld8(zb + gr * ZSO + (gq + 16 * i) * 8, v[i]);
        norm_rows<U / 128>(v, a.img_out[l], mrow, glive, abuf, 0);
      }
      __syncthreads();
    }
    // ================= img_stats + draw: S columns in passes of 512 (16 groups)
#pragma unroll 1
    for (int ps_ = 0; ps_ < S / 512; ++ps_) {
      {
        f32x4 acc[8];
        s8.run(a.stats_planes + (long)(ps_ * 32 + wave * 8) * S8::TILE_BYTES, abuf, acc);
        tiles_to_zb<8>(acc, zb, wave * 128, a.stats_bias + ps_ * 512);
      }
      __syncthreads();
      // 16 rows x 16 groups = 256 items, one per thread
      {
        const int r = tid & 15, gl = tid >> 4, g = ps_ * 16 + gl;
        float x[32];
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
          const float4 q = *reinterpret_cast<const float4*>(zb + r * ZSO + gl * C + c);
          x[c] = q.x; x[c + 1] = q.y; x[c + 2] = q.z; x[c + 3] = q.w;
        }
        const int idx = draw_item32(x, ubuf[r * G + g], a.unimix);
        cls[r][g] = idx;
        if (row0 + r < N) {
          float* xo = a.xs + ((long)t * N + row0 + r) * S + g * C;
          float* so = tnext + (row0 + r) * W + D + g * C;
#pragma unroll
          for (int c = 0; c < 32; c += 4) {
            *reinterpret_cast<float4*>(xo + c) = make_float4(x[c], x[c + 1], x[c + 2], x[c + 3]);
            *reinterpret_cast<float4*>(so + c) = make_float4(idx == c ? 1.f : 0.f, idx == c + 1 ? 1.f : 0.f,
                                                             idx == c + 2 ? 1.f : 0.f, idx == c + 3 ? 1.f : 0.f);
          }
        }
      }
      __syncthreads();
    }
  }
}

constexpr int IMAG_OH_LDS = 32 * 3 * 1024 + 16 * HSO * 4 + 16 * 32 * 4 + 16 * 64 * 4 + 16 * 32 * 4 + 16 * 4 + 16 * 4;

bool imag_oh_device_ok() {
  int dev = 0, lds = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return true; }
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) {
    (void)hipGetLastError();
    return true;
  }
  return lds >= IMAG_OH_LDS;
}

}  // namespace

// compiled shapes (deter, units, groups, classes, action classes, actor units); actor layers = 4,
// prior layers = 3, one-hot actions
#define DD_IMAG_OH_SHAPES(X) X(512, 512, 32, 32, 6, 512) X(512, 512, 32, 32, 4, 512)

// (used by dd_imagine_rollout_supported, imag.hip)
int imag_oh_supported(int D, int U, int G, int C, int A, int actor_units) {
  if (!imag_oh_device_ok()) return 0;
#define X(d, u, g, c, a_, au) if (D == d && U == u && G == g && C == c && A == a_ && actor_units == au) return 1;
  DD_IMAG_OH_SHAPES(X)
#undef X
  return 0;
}

// ptrs (device pointers, in this order):
//   0 traj  1 u_img  2 u_act
//   3.. actor layer l = 0..3: planes, gamma, beta, z, stats, out      (6 each -> 3..26)
//   27 actor dense0 kernel (fp32)  28 head planes  29 head bias  30 raw logits  31 log-probabilities
//   32 img_in kernel (fp32)  33 gamma  34 beta  35 z  36 stats  37 out
//   38 gru planes  39 gamma  40 beta  41 z3  42 gstats
//   43.. img_out l = 0..2: planes, gamma, beta, z, stats, out           (6 each -> 43..60)
//   61 stats planes  62 stats bias  63 raw statistics
extern "C" int dd_imagine_rollout_oh_fwd(int N, int H, int t0, int t1, int D, int U, int G, int C, int A,
                                         int actor_units, int row_width, float unimix, float actor_unimix,
                                         const void* const* p, int n_ptrs, void* stream) {
  DD_REQUIRE(imag_oh_supported(D, U, G, C, A, actor_units), "dd_imagine_rollout_oh_fwd: unsupported shape");
  DD_REQUIRE(n_ptrs == 64 && N >= 1 && H >= 1, "dd_imagine_rollout_oh_fwd: 64 pointers");
  DD_REQUIRE(0 <= t0 && t0 < t1 && t1 <= H + 1, "dd_imagine_rollout_oh_fwd: 0 <= t0 < t1 <= H + 1");
  DD_REQUIRE(row_width >= D + G * C + A && row_width % 4 == 0, "dd_imagine_rollout_oh_fwd: row width a multiple of four floats");
  ImagOhArgs a;
  a.N = N; a.H = H; a.t0 = t0; a.t1 = t1; a.W = row_width; a.unimix = unimix; a.act_unimix = actor_unimix;
  a.traj = (float*)p[0]; a.u_img = (const float*)p[1]; a.u_act = (const float*)p[2];
  auto layer = [&](int i) {
    OhLayer L;
    L.planes = (const char*)p[i]; L.gamma = (const float*)p[i + 1]; L.beta = (const float*)p[i + 2];
    L.z = (float*)p[i + 3]; L.st = (float*)p[i + 4]; L.out = (float*)p[i + 5];
    return L;
  };
  for (int l = 0; l < 4; ++l) a.actor[l] = layer(3 + 6 * l);
  a.w_actor0 = (const float*)p[27]; a.head_planes = (const char*)p[28]; a.head_bias = (const float*)p[29];
  a.z_head = (float*)p[30]; a.alogit = (float*)p[31];
  a.w_in = (const float*)p[32];
  a.img_in.planes = nullptr; a.img_in.gamma = (const float*)p[33]; a.img_in.beta = (const float*)p[34];
  a.img_in.z = (float*)p[35]; a.img_in.st = (float*)p[36]; a.img_in.out = (float*)p[37];
  a.gru.planes = (const char*)p[38]; a.gru.gamma = (const float*)p[39]; a.gru.beta = (const float*)p[40];
  a.gru.z = (float*)p[41]; a.gru.st = (float*)p[42]; a.gru.out = nullptr;
  for (int l = 0; l < 3; ++l) a.img_out[l] = layer(43 + 6 * l);
  a.stats_planes = (const char*)p[61]; a.stats_bias = (const float*)p[62]; a.xs = (float*)p[63];
  const int blocks = (N + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  bool launched = false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
#define X(d, u, g, c, a_, au)                                                                    \
  if (!launched && D == d && U == u && G == g && C == c && A == a_ && actor_units == au) {       \
    static unsigned long long attr = 0;   /* one bit per device: the attribute is per device */  \
    const unsigned long long bit = 1ull << (dev & 63);                                           \
    if (!(attr & bit)) {                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)k_imagine_rollout_oh<d, u, g, c, a_, au>,  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, IMAG_OH_LDS); \
      if (e != hipSuccess) { dd_set_error("dd_imagine_rollout_oh_fwd(attr)", e); return (int)e; } \
      attr |= bit;                                                                               \
    }                                                                                            \
    k_imagine_rollout_oh<d, u, g, c, a_, au><<<blocks, 256, IMAG_OH_LDS, st>>>(a);               \
    launched = true;                                                                             \
  }
  DD_IMAG_OH_SHAPES(X)
#undef X
  DD_CHECK_LAUNCH("dd_imagine_rollout_oh_fwd");
  return 0;
}
