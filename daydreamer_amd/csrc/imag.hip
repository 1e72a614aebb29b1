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
#include <stdlib.h>
#include "imag_core.h"
#include "../../include/daydreamer_hip.h"

namespace {

using LayerP = DDImagLayerP;     // (imag_core.h: shared with the 32-row form, imag32.hip)
using ImagArgs = DDImagArgs;

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


// ============================================================================================
// Reverse pass: the data gradient of the imagined rollout (actor_grad 'backprop', reference
// agent.py:355-356 - tape.gradient through WorldModel.imagine).  Given d score / d state_t through
// the reward / cont / critic heads in dtraj[t][:, :F] (bulk launches before this one), step
// t = H .. 1 pulls the gradient of state_t back through the draw's straight-through estimator
// and RSSM.img_step (img_stats, img_out 2..0, GRU, img_in) to state_{t-1} and action_{t-1}:
//   dxs   = stats_bwd(xs_{t-1}, dstoch_t)                 softmax + unimix, tfutils.py:376-381
//   ddet += ln_bwd / W^T chain of img_out 2..0 from dxs @ W_stats^T
//   dz3, dh_direct = gru_bwd(ddet_t, z3_{t-1}, h_{t-1});  [dh | dx1] = dz3 @ W_gru^T
//   dz1   = ln_bwd(img_in);  [dstoch_{t-1} | daction_{t-1}] += dz1 @ W_in^T
// Same decomposition as the forward kernel: one workgroup per 16-row block, no grid
// synchronisation, A operands built in LDS by the consumer, transposed weight planes streamed.
// Writes what the per-layer launch sequence (learner._actor_backprop.scan_step) leaves behind:
// dtraj[t][:, :D] = the total gradient of deter_t, dtraj[t-1][:, D:] += the step's contribution.

using LayerB = DDImagLayerB;
using ImagBwdArgs = DDImagBwdArgs;

// LayerNorm + ELU backward of this thread's chunks: dout rows in zb (columns col0 ..), z / out /
// statistics of the forward pass from global (requested by load() BEFORE the contraction that
// produces dout, so their latency hides behind it); dz -> A operand planes (k-steps ks0 ..).
template <int NC>
struct LnBwd {
  float z[NC][8], o[NC][8], gm[NC][8];
  float2 ms;
  __device__ __forceinline__ void load(const LayerB& L, long grow) {
    constexpr int NCOL = NC * 128;
    const int q = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int k = (q + 16 * i) * 8;
      ld8(L.z + grow * NCOL + k, z[i]);
      ld8(L.out + grow * NCOL + k, o[i]);
      ld8(L.gamma + k, gm[i]);
    }
    ms = *reinterpret_cast<const float2*>(L.st + grow * 2);
  }
  __device__ __forceinline__ void run(const float* zb, int col0, char* abuf, int ks0) {
    constexpr int NCOL = NC * 128;
    const int tid = threadIdx.x, row = tid >> 4, q = tid & 15;
    const float mean = ms.x, rstd = ms.y;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float d[8];
      ld8(zb + row * ZS + col0 + (q + 16 * i) * 8, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dy = d[j] * (o[i][j] > 0.f ? 1.f : o[i][j] + 1.f);
        z[i][j] = (z[i][j] - mean) * rstd;          // x hat
        gm[i][j] = dy * gm[i][j];                   // g
        s1 += gm[i][j];
        s2 += gm[i][j] * z[i][j];
      }
    }
    s1 = row16_sum(s1) / (float)NCOL;
    s2 = row16_sum(s2) / (float)NCOL;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float dz[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) dz[j] = rstd * (gm[i][j] - s1 - z[i][j] * s2);
      put_operand(abuf, ks0, row, q, i, dz);
    }
  }
};

#define TSB(i) if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0 && t == a.H - 1) a.dbg[i] = wall_clock64()

template <int D, int U, int G, int C, int A>
__global__ void __launch_bounds__(256, 1)
k_imagine_reverse(ImagBwdArgs a) {
  constexpr int S = G * C, F = D + S, W = F + A;
  static_assert(D == 256 && U == 256 && C == 32 && G == 32 && A <= 16, "compiled shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* zb = reinterpret_cast<float*>(smem);                       // [16][ZS]
  char* abuf = smem + 16 * ZS * 4;                                  // 16 k-steps x 3 planes x 1 KB
  float* dhb = reinterpret_cast<float*>(abuf + 16 * 3 * 1024);      // [16][HS] gradient of deter carried to step t - 1
  float* par = dhb + 16 * HS;                                       // GRU scale [3D], offset [3D]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N, H = a.H;
  const long row0 = (long)blockIdx.x * 16;
  const int gr = tid >> 4, gq = tid & 15;
  const long gg = min(row0 + gr, (long)N - 1);
  const bool glive = row0 + gr < N;

  for (int i = tid; i < 3 * D; i += 256) { par[i] = a.gru_gamma[i]; par[3 * D + i] = a.gru_beta[i]; }
  for (int i = tid; i < 16 * HS; i += 256) dhb[i] = 0.f;
  __syncthreads();

  for (int t = H; t >= 1; --t) {
    const long mrow = (long)(t - 1) * N + gg;          // this thread's row of step t - 1 in the [H*N, ..] buffers
    float* dcur = a.dtraj + ((long)t * N) * W;
    float* dprev = a.dtraj + ((long)(t - 1) * N) * W;
    const float* tprev = a.traj + ((long)(t - 1) * N) * W;
    TSB(0);
    // ================= draw backward + img_stats^T: K = S in two halves of 16 groups
    LnBwd<U / 128> lnb;
    lnb.load(a.img_out[2], mrow);
    f32x4 accs[4];
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      {
        // item = (row r, group g): one thread, serial over the 32 classes
        const int r = tid & 15, gl = tid >> 4, g = h * 16 + gl;
        const long gw = min(row0 + r, (long)N - 1);
        const float* xp = a.xs + ((long)(t - 1) * N + gw) * S + g * C;
        const float* dp_ = dcur + gw * W + D + g * C;
        float x[32], ds[32];
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
          const float4 q = *reinterpret_cast<const float4*>(xp + c);
          const float4 e = *reinterpret_cast<const float4*>(dp_ + c);
          x[c] = q.x; x[c + 1] = q.y; x[c + 2] = q.z; x[c + 3] = q.w;
          ds[c] = e.x; ds[c + 1] = e.y; ds[c + 2] = e.z; ds[c + 3] = e.w;
        }
        float m = x[0];
#pragma unroll
        for (int c = 1; c < 32; ++c) m = fmaxf(m, x[c]);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) { x[c] = fexp_(x[c] - m); sum += x[c]; }
        const float inv = 1.f / sum;
        // pm = (1 - unimix) p + unimix / C; the straight-through sample passes d pm = d stoch
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
          x[c] *= inv;                                  // p
          ds[c] *= (1.f - a.unimix);                    // dp
          dot += ds[c] * x[c];
        }
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
          float dx[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) dx[j] = x[c8 * 8 + j] * (ds[c8 * 8 + j] - dot);
          uint4 pl[3];
          split8(dx, pl);
          // class chunk c8 of group gl is fragment lane c8 * 16 + r of k-step gl
#pragma unroll
          for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(abuf + ((gl * 3 + p) * 64 + c8 * 16 + r) * 16) = pl[p];
        }
      }
      __syncthreads();
      Stream<4, 16, false, 32> sT;
      sT.run(a.stats_planes + (long)(wave * 4) * (32 * 3072) + h * 16 * 3072, abuf, accs, h == 0);
      __syncthreads();
    }
    tiles_to_z<4, false>(accs, zb, wave * 64, nullptr);
    __syncthreads();
    TSB(1);
    // ================= img_out 2 .. 0: LayerNorm / ELU backward, W^T
#pragma unroll 1
    for (int l = 2; l >= 0; --l) {
      lnb.run(zb, 0, abuf, 0);
      __syncthreads();
      if (l > 0) lnb.load(a.img_out[l - 1], mrow);      // the next layer's activations travel during the contraction
      Stream<4, 8> sO;
      f32x4 acc[4];
      sO.run(a.img_out[l].planes + (long)(wave * 4) * Stream<4, 8>::TILE_BYTES, abuf, acc);
      tiles_to_z<4, false>(acc, zb, wave * 64, nullptr);
      __syncthreads();
    }
    TSB(2);
    // ================= GRU backward
    {
      constexpr int NC = 3 * D / 128, ND = D / 128;
      float v[NC][8], dy[NC][8];   // (the normalised z3 is recomputed from v where needed: a third array spills)
#pragma unroll
      for (int i = 0; i < NC; ++i) ld8(a.z3 + mrow * (3 * D) + (gq + 16 * i) * 8, v[i]);
      const float2 ms = *reinterpret_cast<const float2*>(a.gstats + mrow * 2);
      const float mean = ms.x, rstd = ms.y;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int d = (gq + 16 * i) * 8;
        float hp[8], dd[8], carry[8], rec[8], dhd[8];
        ld8(tprev + gg * W + d, hp);
        ld8(dcur + gg * W + d, dd);
        ld8(dhb + gr * HS + d, carry);
        ld8(zb + gr * ZS + d, rec);
#pragma unroll
        for (int j = 0; j < 8; ++j) dd[j] = (dd[j] + carry[j]) + rec[j];     // total gradient of deter_t
        if (glive) st8(dcur + gg * W + d, dd);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xr = (v[i][j] - mean) * rstd, xc = (v[i + ND][j] - mean) * rstd, xu = (v[i + 2 * ND][j] - mean) * rstd;
          const float yr = xr * par[d + j] + par[3 * D + d + j];
          const float yc = xc * par[D + d + j] + par[4 * D + d + j];
          const float yu = xu * par[2 * D + d + j] + par[5 * D + d + j];
          const float r = sigmoidf_(yr);
          const float cand = tanhf(r * yc);
          const float u = sigmoidf_(yu - 1.f);
          const float du = dd[j] * (cand - hp[j]);
          const float dc = dd[j] * u;
          dhd[j] = dd[j] * (1.f - u);
          const float dpre = dc * (1.f - cand * cand);
          const float dyc = dpre * r, dyr = dpre * yc * r * (1.f - r), dyu = du * u * (1.f - u);
          dy[i][j] = dyr * par[d + j]; dy[i + ND][j] = dyc * par[D + d + j]; dy[i + 2 * ND][j] = dyu * par[2 * D + d + j];
          s1 += dy[i][j] + dy[i + ND][j] + dy[i + 2 * ND][j];
          s2 += dy[i][j] * xr + dy[i + ND][j] * xc + dy[i + 2 * ND][j] * xu;
        }
        st8(dhb + gr * HS + d, dhd);          // direct path (1 - update) * dh'; the W^T part is added below
      }
      s1 = row16_sum(s1) / (float)(3 * D);
      s2 = row16_sum(s2) / (float)(3 * D);
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) dy[i][j] = rstd * (dy[i][j] - s1 - (v[i][j] - mean) * rstd * s2);   // dz3
      // [dh | dx1] = dz3 @ W_gru^T, K = 3D in two parts (the operand buffer holds 16 k-steps)
      __syncthreads();                     // zb (rec) and abuf free
#pragma unroll
      for (int i = 0; i < 4; ++i) put_operand(abuf, 0, gr, gq, i, dy[i]);
      __syncthreads();
      lnb.load(a.img_in, mrow);            // img_in's activations travel during the contraction
      // (two groups of 4 tiles per wave: 8 tiles' weight fragments at once do not fit the registers
      // next to the second part of dz3)
      f32x4 acc0[4], acc1[4];
      const char* wg = a.gru_planes + (long)(wave * 8) * (24 * 3072);
      {
        Stream<4, 16, false, 24> sG;
        sG.run(wg, abuf, acc0, true);
        sG.run(wg + (long)4 * (24 * 3072), abuf, acc1, true);
      }
      __syncthreads();
#pragma unroll
      for (int i = 4; i < NC; ++i) put_operand(abuf, -16, gr, gq, i, dy[i]);
      __syncthreads();
      {
        Stream<4, 8, false, 24> sG2;
        sG2.run(wg + 16 * 3072, abuf, acc0, false);
        sG2.run(wg + (long)4 * (24 * 3072) + 16 * 3072, abuf, acc1, false);
      }
      tiles_to_z<4, false>(acc0, zb, wave * 128, nullptr);
      tiles_to_z<4, false>(acc1, zb, wave * 128 + 64, nullptr);
    }
    __syncthreads();
    TSB(3);
    // ================= dh carry, img_in backward
#pragma unroll
    for (int i = 0; i < D / 128; ++i) {
      const int d = (gq + 16 * i) * 8;
      float x[8], y[8];
      ld8(dhb + gr * HS + d, x);
      ld8(zb + gr * ZS + d, y);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += y[j];
      st8(dhb + gr * HS + d, x);
    }
    lnb.run(zb, D, abuf, 0);
    __syncthreads();
    TSB(4);
    // [dstoch_{t-1} | daction_{t-1}] += dz1 @ W_in^T : S columns in two passes of 8 tiles per wave
#pragma unroll 1
    for (int ps_ = 0; ps_ < S / 512; ++ps_) {
      Stream<8, 8> sI;
      f32x4 acc[8];
      sI.run(a.img_in.planes + (long)(ps_ * 32 + wave * 8) * Stream<8, 8>::TILE_BYTES, abuf, acc);
      // through the z buffer: the accumulation into dtraj[t - 1] is a row-wise pass of 32-byte
      // accesses, all loads in flight together (4-byte read-modify-writes straight from the tile
      // layout serialise one memory latency per element)
      __syncthreads();          // zb readers of the previous pass / of ln_bwd are done
      tiles_to_z<8, false>(acc, zb, wave * 128, nullptr);
      __syncthreads();
      if (glive) {
        float o[4][8], d[4][8];
        float* dst = dprev + gg * W + D + ps_ * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i) ld8(dst + (gq + 16 * i) * 8, o[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ld8(zb + gr * ZS + (gq + 16 * i) * 8, d[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[i][j] += d[i][j];
          st8(dst + (gq + 16 * i) * 8, o[i]);
        }
      }
    }
    if (wave == 0) {   // the action columns: one tile (tile S / 16 of the cache)
      uint4 bq[8][3];
      const char* wp = a.img_in.planes + (long)(S / 16) * Stream<8, 8>::TILE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[ks][p] = *reinterpret_cast<const uint4*>(wp + (ks * 3 + p) * 1024 + (unsigned)lane * 16u);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8 av[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          av[p] = *reinterpret_cast<const bf16x8*>(abuf + ((ks * 3 + p) * 64 + lane) * 16);
          b[p] = __builtin_bit_cast(bf16x8, bq[ks][p]);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], b[0], acc, 0, 0, 0);
      }
      const int col = lane & 15;
      if (col < A) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long rw = row0 + (lane >> 4) * 4 + r;
          if (rw < N) dprev[rw * W + F + col] += acc[r];
        }
      }
    }
    __syncthreads();   // the next step reads dtraj[t - 1] (written above) and reuses abuf / zb
    TSB(5);
  }
  // the gradient of the start states' deter (nothing consumes it; written for the launch
  // sequence's dtraj[0], which it leaves complete)
  if (glive) {
#pragma unroll
    for (int i = 0; i < D / 128; ++i) {
      const int d = (gq + 16 * i) * 8;
      float x[8], y[8];
      ld8(a.dtraj + gg * W + d, x);
      ld8(dhb + gr * HS + d, y);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += y[j];
      st8(a.dtraj + gg * W + d, x);
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

namespace {
// The same for the transposed operand: W stored [n, K] (row stride ld), B[k][col] = W[col][k].
__global__ void k_imag_wprep_t(const float* __restrict__ W, long ld, int K, int n, char* __restrict__ planes) {
  const int KS = K / 32;
  const long total = (long)((n + 15) / 16) * KS * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long tk = i >> 6;
    const int ks = (int)(tk % KS), tile = (int)(tk / KS);
    const int col = tile * 16 + (lane & 15);
    const int k0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
    if (col < n) ld8(W + (long)col * ld + k0, v);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    uint4 pl[3];
    split8(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint4*>(planes + (((long)tile * KS + ks) * 3 + p) * 1024 + lane * 16) = pl[p];
  }
}

}  // namespace

extern "C" int dd_imag_wprep_t(const float* W, long ld, int K, int n, void* planes, void* stream) {
  DD_REQUIRE(K % 32 == 0 && n >= 1 && ld % 4 == 0, "dd_imag_wprep_t: K multiple of 32, ld multiple of 4");
  const long total = (long)((n + 15) / 16) * (K / 32) * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  k_imag_wprep_t<<<blocks, 256, 0, (hipStream_t)stream>>>(W, ld, K, n, (char*)planes);
  DD_CHECK_LAUNCH("dd_imag_wprep_t");
  return 0;
}

namespace {
constexpr int IMAG_BWD_LDS = 16 * ZS * 4 + 16 * 3 * 1024 + 16 * HS * 4 + 6 * 256 * 4;
constexpr int IMAG_LDS = 16 * ZS * 4 + 16 * 3 * 1024 + 16 * HS * 4 + 16 * 256 * 4 + 16 * 32 * 4 + 256 * 4 + 256 * 4 + 16 * 32 * 4 +
                         (8 * 512 + 6 * 256) * 4;
int imag_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return dev;
}
// The fused kernels need ~157 KB of dynamic LDS per workgroup, which only a full gfx950 CU grants:
// on a device (or partition) that cannot, the shape query answers "unsupported" and the caller
// keeps the per-layer launch sequence (as scan.hip's scan_device_ok does for its CU count).
// Without a visible device (build / CPU tests) the shape alone decides.
bool imag_device_ok() {
  int dev = 0, lds = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return true; }
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) {
    (void)hipGetLastError();
    return true;
  }
  const int need16 = IMAG_LDS > IMAG_BWD_LDS ? IMAG_LDS : IMAG_BWD_LDS;
  return lds >= need16 && lds >= dd_imag32_lds_bytes();
}
}

// rows per workgroup of the forward rollout: 16 (k_imagine_rollout) or 32 (k_imagine_rollout32)
static int g_imag_rows = -1;
static int imag_rows() {
  if (g_imag_rows < 0) {
    const char* e = getenv("DD_IMAG_ROWS");
    g_imag_rows = (e && atoi(e) == 16) ? 16 : 32;
  }
  return g_imag_rows;
}
extern "C" int dd_imag_set_rows(int rows) {
  const int prev = imag_rows();
  DD_REQUIRE(rows == 16 || rows == 32, "dd_imag_set_rows: 16 or 32");
  g_imag_rows = rows;
  return prev;
}

// ptrs (device pointers, in this order):
//   0 traj  1 dtraj  2 raw statistics  3 img_stats^T planes
//   4.. img_out l = 0..2: W^T planes, gamma, z, stats, out               (5 each -> 4..18)
//   19 gru^T planes  20 gru gamma  21 gru beta  22 z3  23 gstats
//   24 img_in^T planes  25 gamma  26 z  27 stats  28 out   [29 optional: 8 x u64 time stamps]
extern "C" int dd_imagine_rollout_bwd(int N, int H, int D, int U, int G, int C, int A, float unimix,
                                      const void* const* p, int n_ptrs, void* stream) {
  DD_REQUIRE(dd_imagine_rollout_supported(D, U, G, C, A, 512, 4, 3, 0), "dd_imagine_rollout_bwd: unsupported shape");
  DD_REQUIRE((n_ptrs == 29 || n_ptrs == 30) && N >= 1 && H >= 1, "dd_imagine_rollout_bwd: 29 pointers");
  ImagBwdArgs a;
  a.N = N; a.H = H; a.unimix = unimix;
  a.traj = (const float*)p[0]; a.dtraj = (float*)p[1]; a.xs = (const float*)p[2];
  a.stats_planes = (const char*)p[3];
  for (int l = 0; l < 3; ++l) {
    a.img_out[l].planes = (const char*)p[4 + 5 * l]; a.img_out[l].gamma = (const float*)p[5 + 5 * l];
    a.img_out[l].z = (const float*)p[6 + 5 * l]; a.img_out[l].st = (const float*)p[7 + 5 * l];
    a.img_out[l].out = (const float*)p[8 + 5 * l];
  }
  a.gru_planes = (const char*)p[19]; a.gru_gamma = (const float*)p[20]; a.gru_beta = (const float*)p[21];
  a.z3 = (const float*)p[22]; a.gstats = (const float*)p[23];
  a.img_in.planes = (const char*)p[24]; a.img_in.gamma = (const float*)p[25]; a.img_in.z = (const float*)p[26];
  a.img_in.st = (const float*)p[27]; a.img_in.out = (const float*)p[28];
  a.dbg = n_ptrs == 30 ? (unsigned long long*)p[29] : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (imag_rows() == 32) {      // 32 rows per workgroup (imag32.hip)
    const int rc = dd_imag32_bwd_launch(a, D, U, G, C, A, st);
    if (rc == 0) { DD_CHECK_LAUNCH("dd_imagine_rollout_bwd(32 rows)"); return 0; }
    if (rc != 1) return rc;
  }
  const int blocks = (N + 15) / 16;
  bool launched = false;
#define XB(d, u, g, c, a_)                                                                       \
  if (!launched && D == d && U == u && G == g && C == c && A == a_) {                            \
    static unsigned long long attr = 0;   /* one bit per device: the attribute is per device */  \
    const unsigned long long bit = 1ull << (imag_device() & 63);                                 \
    if (!(attr & bit)) {                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)k_imagine_reverse<d, u, g, c, a_>,         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, IMAG_BWD_LDS); \
      if (e != hipSuccess) { dd_set_error("dd_imagine_rollout_bwd(attr)", e); return (int)e; }   \
      attr |= bit;                                                                               \
    }                                                                                            \
    k_imagine_reverse<d, u, g, c, a_><<<blocks, 256, IMAG_BWD_LDS, st>>>(a);                     \
    launched = true;                                                                             \
  }
  XB(256, 256, 32, 32, 16) XB(256, 256, 32, 32, 6)
#undef XB
  DD_CHECK_LAUNCH("dd_imagine_rollout_bwd");
  return 0;
}

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

int imag_oh_supported(int D, int U, int G, int C, int A, int actor_units);   // imag_oh.hip

extern "C" int dd_imagine_rollout_supported(int D, int U, int G, int C, int A, int actor_units,
                                            int actor_layers, int prior_layers, int discrete) {
  if (actor_layers != 4 || prior_layers != 3) return 0;
  if (discrete) return imag_oh_supported(D, U, G, C, A, actor_units);   // forward only (dd_imagine_rollout_oh_fwd)
  if (!imag_device_ok()) return 0;
#define X(d, u, g, c, a_, au) if (D == d && U == u && G == g && C == c && A == a_ && actor_units == au) return 1;
  DD_IMAG_SHAPES(X)
#undef X
  return 0;
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
  hipStream_t st = (hipStream_t)stream;
  if (imag_rows() == 32) {      // 32 rows per workgroup (imag32.hip): half the CUs for the same weight stream
    const int rc = dd_imag32_launch(a, D, U, G, C, A, actor_units, st);
    if (rc == 0) { DD_CHECK_LAUNCH("dd_imagine_rollout_fwd(32 rows)"); return 0; }
    if (rc != 1) return rc;
  }
  const int blocks = (N + 15) / 16;
  bool launched = false;
#define X(d, u, g, c, a_, au)                                                                    \
  if (!launched && D == d && U == u && G == g && C == c && A == a_ && actor_units == au) {       \
    static unsigned long long attr = 0;   /* one bit per device: the attribute is per device */  \
    const unsigned long long bit = 1ull << (imag_device() & 63);                                 \
    if (!(attr & bit)) {                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)k_imagine_rollout<d, u, g, c, a_, au>,     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, IMAG_LDS);  \
      if (e != hipSuccess) { dd_set_error("dd_imagine_rollout_fwd(attr)", e); return (int)e; }   \
      attr |= bit;                                                                               \
    }                                                                                            \
    k_imagine_rollout<d, u, g, c, a_, au><<<blocks, 256, IMAG_LDS, st>>>(a);                     \
    launched = true;                                                                             \
  }
  DD_IMAG_SHAPES(X)
#undef X
  DD_CHECK_LAUNCH("dd_imagine_rollout_fwd");
  return 0;
}
