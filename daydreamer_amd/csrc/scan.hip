// Fused RSSM.observe forward scan: ONE persistent launch for all T steps.
//
// Reference: RSSM.observe nets.py:66-76 over tfutils.scan 50-70, RSSM.obs_step nets.py:99-117
// (reset mask :100-107, img_step :119-130 up to the new deter, obs_out + obs_stats + sample
// :109-116), RSSM._gru nets.py:149-160, Norm nets.py:585-602, OneHotDist.sample tfutils.py:368-382.
//
// Why: a step of the scan is four dependent small GEMMs (B <= 64 rows) with row-wise epilogues.
// As separate launches that is 8 kernels of 4-6 us each per step (fixed cost: one global and
// one LDS round trip, a dependent MFMA chain), 50 x 8 launches per scan.  Here NWG workgroups
// stay resident; every layer of every step is a phase between two grid barriers:
//   P1  z1  = [mask(stoch_{t-1}) | mask(action_t)] @ W_img_in                  (raw, pre-norm)
//   P2  z3  = [mask(deter_{t-1}) | ELU(LN(z1))] @ W_gru                         (raw)
//   P3  zo += GRU(LN(z3), hprev) @ W_obs_out[:D]      (zo holds the hoisted embed part)
//   P4  xq  = ELU(LN(zo)) @ W_obs_stats + b ;  post_logit, stoch_t = sample(xq, u_t)
// The consumer normalises: a phase reads the raw output rows of the previous phase, computes
// the LayerNorm statistics of its own 16 rows and applies norm / ELU / GRU gates while it
// builds its MFMA operand, so no phase needs a reduction across workgroups.
//
// Work decomposition: output tiles of 16 rows x 16 columns (v_mfma_f32_16x16x32_bf16, fp32
// operands split exactly into three bf16 terms, six products, as the big contraction
// kernels); workgroup w owns row block w % 4 in every phase and the column tiles
// w / 4 + (NWG / 4) * j; its four waves split K and sum their partial tiles through LDS in a
// fixed order.  Weights come from a per-step cache of transposed bf16 planes [3][N][Kpad]
// (k_scan_wprep) - L2-resident, one 16-byte load per plane and k-octet, no split arithmetic.
// Every buffer the reverse scan and the bulk weight-gradient contractions read (z1, x1, LN
// statistics, hprev, masked stoch, z3, deter, zo, xo, xq, post_logit, stoch) is written exactly
// as the unfused path writes it.
//
// Grid barrier: one monotonic counter, arrive = release fence + relaxed agent-scope add, wait =
// relaxed agent-scope polling by one lane + acquire fence (MI355X_MICROARCH.md, valid forms);
// every spin is bounded - on a timeout the error word is set and the kernel runs on (garbage
// out, never a hang).  NWG <= CU count, one workgroup per CU, so all workgroups are resident.
#include "latent_core.h"
#include <math.h>
#include <stdlib.h>
#include "../../include/daydreamer_hip.h"

namespace {

constexpr float LN_EPS = 1e-3f;
constexpr int NWG_DEFAULT = 128;   // resident workgroups (column strides x 4 row blocks), DD_SCAN_NWG
constexpr int SPIN_LIMIT = 1 << 22;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct ScanArgs {
  int B, T, D, U, S, G, C, A;      // S = G * C
  int XK, XKp;                     // S + A and its multiple of 32
  int use_carry;
  int nwg;                         // workgroups in the grid (multiple of 4, <= CU count)
  float unimix;
  // inputs
  const float* first;              // [B*T]  is_first as float
  const float* carry;              // [B, D+S] previous posterior (step 0), may be null
  const float* init_deter;         // [D]
  const float* init_stoch;         // [S]
  const float* u_post;             // [T, B, G]
  // weights: bf16 plane caches [3][N][Kp] and fp32 vectors
  const unsigned short *wt1, *wt2, *wt3, *wt4;
  const float* w_in;               // img_in kernel [S+A, U] fp32 (P1 gathers its rows)
  const float *g1, *b1, *gg, *bg, *g3, *b3, *bias4;
  // buffers (rows b*T + t)
  float* xin;        // [N, S+A]
  float* z1;         // [N, U]
  float* st1;        // [N, 2]
  float* gin;        // [N, D+U]   [hprev | x1]
  float* z3;         // [N, 3D]
  float* gst;        // [N, 2]
  float* post;       // [N, D+S]
  float* zo;         // [N, U]
  float* xo;         // [N, U]
  float* st3;        // [N, 2]
  float* xq;         // [N, S]
  float* post_logit; // [N, S]
  int* idx;          // [N, G] drawn class per (row, group) (P1 of the next step gathers by it)
  const int* idx_init;   // [G] classes of the initial stoch (one-hot)
  const int* idx_carry;  // [B, G] classes of the carried stoch (step 0)
  unsigned* ctr;     // [2]: barrier counter, error word
};

__device__ __forceinline__ float elu_(float y) { return y > 0.f ? y : expm1f(y); }

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ unsigned pack_hi(unsigned even, unsigned odd) {
  return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}
// eight fp32 -> three bf16x8 planes (hi, mid, lo)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&pl)[3]) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
  pl[0] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]), pack_hi(h[4], h[5]), pack_hi(h[6], h[7])));
  pl[1] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]), pack_hi(m[4], m[5]), pack_hi(m[6], m[7])));
  pl[2] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]), pack_hi(l[4], l[5]), pack_hi(l[6], l[7])));
}

// eight consecutive floats (32-byte aligned address) as two 16-byte loads
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) {
        __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// LayerNorm statistics of the 16 rows of this workgroup's row block: thread t -> row t >> 4,
// 16 lanes per row.  src rows at stride ld, width W (multiple of 4).  Result in sh[16][2].
__device__ __forceinline__ void block_stats(const float* src, long ld, int W, const long* rowidx,
                                            float (*sh)[2]) {
  const int r = threadIdx.x >> 4, l = threadIdx.x & 15;
  const float* p = src + rowidx[r] * ld;
  float s = 0.f;
  for (int c = l * 4; c < W; c += 64) {
    const float4 x = *reinterpret_cast<const float4*>(p + c);
    s += (x.x + x.y) + (x.z + x.w);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
  const float mean = s / (float)W;
  float v = 0.f;
  for (int c = l * 4; c < W; c += 64) {
    const float4 x = *reinterpret_cast<const float4*>(p + c);
    const float a = x.x - mean, b = x.y - mean, cc = x.z - mean, d = x.w - mean;
    v += (a * a + b * b) + (cc * cc + d * d);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  if (l == 0) { sh[r][0] = mean; sh[r][1] = rsqrtf(v / (float)W + LN_EPS); }
}

// One 16x16 output tile: acc += A(16 rows x K) @ Wt(plane cache rows n0..n0+15).  `afn(k0, v)`
// yields this lane's eight A values (row = lane & 15 of the block, k = k0 .. k0+7).  The four
// waves take k-steps wave, wave+4, ...; the partial tiles are summed through LDS in wave
// order.  Returns the complete tile element of thread t (row (t&63)>>4)*4 + (t>>6), col t&15).
template <class AF>
__device__ __forceinline__ float tile_gemm(AF afn, const unsigned short* wt, int Nn, int Kp, int n0,
                                           float (*red)[256]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = (lane >> 4) * 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long plane = (long)Nn * Kp;
  const unsigned short* wrow = wt + (long)(n0 + (lane & 15)) * Kp + kq;
  // software pipeline: the operands of k-step i+1 are loaded before the MFMAs of k-step i
  float v[8], vn[8];
  uint4 bq[3], bn[3];
  int k0 = wave * 32;
  if (k0 < Kp) {
    afn(k0 + kq, v);
#pragma unroll
    for (int p = 0; p < 3; ++p) bq[p] = *reinterpret_cast<const uint4*>(wrow + p * plane + k0);
  }
  for (; k0 < Kp; k0 += 128) {
    const int k1 = k0 + 128;
    if (k1 < Kp) {
      afn(k1 + kq, vn);
#pragma unroll
      for (int p = 0; p < 3; ++p) bn[p] = *reinterpret_cast<const uint4*>(wrow + p * plane + k1);
    }
    bf16x8 a[3], b[3];
    split8(v, a);
#pragma unroll
    for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bq[p]);
    // six cross products, smallest terms first (as k_mfma_gemm_s3)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    if (k1 < Kp) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = vn[j];
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[p] = bn[p];
    }
  }
  __syncthreads();   // red[] free (previous tile's readers are done)
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r * 64 + lane] = acc[r];
  __syncthreads();
  const int t = threadIdx.x;
  return ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
}

__global__ void __launch_bounds__(256, 1)
k_observe_scan_fwd(ScanArgs a) {
  __shared__ float red[4][256];
  __shared__ float sh_stats[16][2];
  __shared__ long sh_row[16];      // buffer row b*T + t of the block's 16 rows (clamped)
  __shared__ long sh_prev[16];     // row of step t-1
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int NWG = a.nwg;
  const int mblk = wg & 3, nstr = wg >> 2, NSTR = NWG / 4;
  const int D = a.D, U = a.U, S = a.S, C = a.C, F = a.D + a.S, T = a.T;
  const int orow = ((lane >> 4) * 4) + (tid >> 6), ocol = tid & 15;   // element of a finished tile
  const int ob = mblk * 16 + orow;                                    // its batch row
  const bool olive = ob < a.B;
  const int ab = min(mblk * 16 + (lane & 15), a.B - 1);               // batch row of the A operand
  unsigned gen = 0;
  if (a.use_carry & 2) {   // measurement aid: the barriers alone (4 per step), no work
    for (int i = 0; i < 4 * T; ++i) grid_barrier(a.ctr, ++gen * NWG);
    return;
  }
  for (int t = 0; t < T; ++t) {
    if (tid < 16) {
      const int b = min(mblk * 16 + tid, a.B - 1);
      sh_row[tid] = (long)b * T + t;
      sh_prev[tid] = (long)b * T + t - 1;
    }
    __syncthreads();
    const long arow = (long)ab * T + t, aprev = arow - 1;
    const float af = a.first[arow];
    const long oidx = (long)ob * T + t;
    // previous posterior of the A-operand row: carry at t = 0
    const float* pprev = t > 0 ? a.post + aprev * F : ((a.use_carry & 1) ? a.carry + (long)ab * F : nullptr);

    // ---------------- P1: z1 = [mask(stoch) | masked action] @ W_img_in
    // The stoch part of the operand is one-hot per group: its product with W is the SUM of the
    // G selected rows of W (1.0 * w is exact, the zero terms vanish) - a gather instead of a
    // K = S contraction; the A action columns are a short dense product.  fp32 adds in group order.
    {
      // side output: the masked stoch input (the bulk weight gradient reads xin); the
      // workgroups of a row block share it, four floats per thread and trip
      for (int e4 = tid + 256 * nstr; e4 < 16 * S / 4; e4 += 256 * NSTR) {
        const int r = e4 / (S / 4), kk = (e4 - r * (S / 4)) * 4;
        const int b = mblk * 16 + r;
        if (b < a.B) {
          const long row = (long)b * T + t;
          const float f = a.first[row];
          const float* pp = t > 0 ? a.post + (row - 1) * F : ((a.use_carry & 1) ? a.carry + (long)b * F : nullptr);
          const float4 iv = *reinterpret_cast<const float4*>(a.init_stoch + kk);
          const float4 pv = pp ? *reinterpret_cast<const float4*>(pp + D + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(a.xin + row * a.XK + kk) =
              make_float4(pv.x * (1.f - f) + iv.x * f, pv.y * (1.f - f) + iv.y * f,
                          pv.z * (1.f - f) + iv.z * f, pv.w * (1.f - f) + iv.w * f);
        }
      }
      // thread -> (row r = tid >> 4 of the block, column c = tid & 15 of the tile)
      const int r = tid >> 4, c = tid & 15;
      const int b = min(mblk * 16 + r, a.B - 1);
      const long row = (long)b * T + t;
      const float f = a.first[row];
      // class of every group of this row's masked stoch: the previous draw, or the initial
      // state's (is_first), or none (no carry at t = 0: zero vector)
      const int* pidx = t > 0 ? a.idx + (row - 1) * a.G : ((a.use_carry & 1) ? a.idx_carry + (long)b * a.G : nullptr);
      const int* sel = f != 0.f ? a.idx_init : pidx;
      for (int nt = nstr; nt < U / 16; nt += NSTR) {
        const int n = nt * 16 + c;
        float acc = 0.f;
        if (sel) {
          for (int g = 0; g < a.G; ++g) acc += a.w_in[(long)(g * C + sel[g]) * U + n];
        }
        for (int j = 0; j < a.A; ++j) acc += a.xin[row * a.XK + S + j] * a.w_in[(long)(S + j) * U + n];
        if (mblk * 16 + r < a.B) a.z1[row * U + n] = acc;
      }
    }
    grid_barrier(a.ctr, ++gen * NWG);

    // ---------------- P2: z3 = [mask(deter) | ELU(LN(z1))] @ W_gru
    {
      block_stats(a.z1, U, U, sh_row, sh_stats);
      __syncthreads();
      const float mean = sh_stats[lane & 15][0], rstd = sh_stats[lane & 15][1];
      auto afn = [&](int k, float (&v)[8]) {
        if (k < D) {
          float pv[8], iv[8];
          ld8(a.init_deter + k, iv);
          if (pprev) ld8(pprev + k, pv);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (pprev ? pv[j] : 0.f) * (1.f - af) + iv[j] * af;
        } else {
          const int c = k - D;
          float zz[8], gm[8], bt[8];
          ld8(a.z1 + arow * U + c, zz); ld8(a.g1 + c, gm); ld8(a.b1 + c, bt);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = elu_((zz[j] - mean) * rstd * gm[j] + bt[j]);
        }
      };
      // side outputs: hprev, x1 (shared by the 16 workgroups of the row block), statistics
      for (int e = tid + 256 * nstr; e < 16 * (D + U); e += 256 * NSTR) {
        const int r = e / (D + U), kk = e - r * (D + U);
        const int b = mblk * 16 + r;
        if (b < a.B) {
          const long row = (long)b * T + t;
          float x;
          if (kk < D) {
            const float f = a.first[row];
            const float* pp = t > 0 ? a.post + (row - 1) * F : ((a.use_carry & 1) ? a.carry + (long)b * F : nullptr);
            x = (pp ? pp[kk] : 0.f) * (1.f - f) + a.init_deter[kk] * f;
          } else {
            const int c = kk - D;
            x = elu_((a.z1[row * U + c] - sh_stats[r][0]) * sh_stats[r][1] * a.g1[c] + a.b1[c]);
          }
          a.gin[row * (D + U) + kk] = x;
        }
      }
      if (nstr == 0 && tid < 16 && mblk * 16 + tid < a.B) {
        a.st1[sh_row[tid] * 2] = sh_stats[tid][0];
        a.st1[sh_row[tid] * 2 + 1] = sh_stats[tid][1];
      }
      for (int nt = nstr; nt < 3 * D / 16; nt += NSTR) {
        const float r = tile_gemm(afn, a.wt2, 3 * D, D + U, nt * 16, red);
        if (olive) a.z3[oidx * 3 * D + nt * 16 + ocol] = r;
      }
    }
    grid_barrier(a.ctr, ++gen * NWG);

    // ---------------- P3: zo += GRU(LN(z3), hprev) @ W_obs_out[:D]
    {
      block_stats(a.z3, 3 * D, 3 * D, sh_row, sh_stats);
      __syncthreads();
      auto gate = [&](float zr_, float zc_, float zu_, float gr, float gc, float gu, float br,
                      float bc, float bu, float hp, float mean, float rstd) {
        const float yr = (zr_ - mean) * rstd * gr + br;
        const float yc = (zc_ - mean) * rstd * gc + bc;
        const float yu = (zu_ - mean) * rstd * gu + bu;
        const float rr = sigmoidf_(yr), cand = tanhf(rr * yc), uu = sigmoidf_(yu - 1.f);
        return uu * cand + (1.f - uu) * hp;
      };
      auto gru = [&](long row, int j, float mean, float rstd) {
        const float* zr = a.z3 + row * 3 * D;
        return gate(zr[j], zr[D + j], zr[2 * D + j], a.gg[j], a.gg[D + j], a.gg[2 * D + j], a.bg[j],
                    a.bg[D + j], a.bg[2 * D + j], a.gin[row * (D + U) + j], mean, rstd);
      };
      const float mean = sh_stats[lane & 15][0], rstd = sh_stats[lane & 15][1];
      auto afn = [&](int k, float (&v)[8]) {
        float z0[8], z1_[8], z2[8], g0[8], g1_[8], g2[8], b0[8], b1_[8], b2[8], hp[8];
        const float* zr = a.z3 + arow * 3 * D;
        ld8(zr + k, z0); ld8(zr + D + k, z1_); ld8(zr + 2 * D + k, z2);
        ld8(a.gg + k, g0); ld8(a.gg + D + k, g1_); ld8(a.gg + 2 * D + k, g2);
        ld8(a.bg + k, b0); ld8(a.bg + D + k, b1_); ld8(a.bg + 2 * D + k, b2);
        ld8(a.gin + arow * (D + U) + k, hp);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = gate(z0[j], z1_[j], z2[j], g0[j], g1_[j], g2[j], b0[j], b1_[j], b2[j], hp[j], mean, rstd);
      };
      // side outputs: deter_t (shared by the row block's workgroups), GRU statistics
      for (int e = tid + 256 * nstr; e < 16 * D; e += 256 * NSTR) {
        const int r = e / D, j = e - r * D;
        if (mblk * 16 + r < a.B) a.post[sh_row[r] * F + j] = gru(sh_row[r], j, sh_stats[r][0], sh_stats[r][1]);
      }
      if (nstr == 0 && tid < 16 && mblk * 16 + tid < a.B) {
        a.gst[sh_row[tid] * 2] = sh_stats[tid][0];
        a.gst[sh_row[tid] * 2 + 1] = sh_stats[tid][1];
      }
      for (int nt = nstr; nt < U / 16; nt += NSTR) {
        const float r = tile_gemm(afn, a.wt3, U, D, nt * 16, red);
        if (olive) a.zo[oidx * U + nt * 16 + ocol] += r;
      }
    }
    grid_barrier(a.ctr, ++gen * NWG);

    // ---------------- P4: xq = ELU(LN(zo)) @ W_obs_stats + b; sample
    {
      block_stats(a.zo, U, U, sh_row, sh_stats);
      __syncthreads();
      const float mean = sh_stats[lane & 15][0], rstd = sh_stats[lane & 15][1];
      auto afn = [&](int k, float (&v)[8]) {
        float zz[8], gm[8], bt[8];
        ld8(a.zo + arow * U + k, zz); ld8(a.g3 + k, gm); ld8(a.b3 + k, bt);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = elu_((zz[j] - mean) * rstd * gm[j] + bt[j]);
      };
      // side outputs: xo (shared by the row block's workgroups), LayerNorm statistics
      for (int e = tid + 256 * nstr; e < 16 * U; e += 256 * NSTR) {
        const int r = e / U, c = e - r * U;
        if (mblk * 16 + r < a.B)
          a.xo[sh_row[r] * U + c] = elu_((a.zo[sh_row[r] * U + c] - sh_stats[r][0]) * sh_stats[r][1] * a.g3[c] + a.b3[c]);
      }
      if (nstr == 0 && tid < 16 && mblk * 16 + tid < a.B) {
        a.st3[sh_row[tid] * 2] = sh_stats[tid][0];
        a.st3[sh_row[tid] * 2 + 1] = sh_stats[tid][1];
      }
      // units of whole latent groups (C / 16 column tiles each), so that the draw of a
      // (row, group) only reads statistics this workgroup wrote
      const int tpg = C / 16;
      for (int g = nstr; g < a.G; g += NSTR) {
        for (int q = 0; q < tpg; ++q) {
          const int n0 = g * C + q * 16;
          const float r = tile_gemm(afn, a.wt4, S, U, n0, red);
          if (olive) a.xq[oidx * S + n0 + ocol] = r + a.bias4[n0 + ocol];
        }
        __syncthreads();
        // 16 (row, group) items, one LW-lane sub-wave each
        const int LWc = C <= 16 ? 16 : (C <= 32 ? 32 : 64);
        const int per_wave = 64 / LWc, wave = tid >> 6;
        for (int it0 = 0; it0 < 16; it0 += 4 * per_wave) {
          const int sub = lane / LWc, c = lane % LWc;
          const int item = it0 + wave * per_wave + sub;
          const int b = mblk * 16 + item;
          const bool live = item < 16 && b < a.B;
          const bool ok = live && c < C;
          const long row = live ? (long)b * T + t : 0;
          const float xv = ok ? a.xq[row * S + g * C + c] : -INFINITY;
          const float uu = live ? a.u_post[((long)t * a.B + b) * a.G + g] : 0.f;
          float lg;
          int idx;
          if (LWc == 16) stats_item<16>(xv, ok, c, sub, C, a.unimix, 0, uu, lg, idx);
          else if (LWc == 32) stats_item<32>(xv, ok, c, sub, C, a.unimix, 0, uu, lg, idx);
          else stats_item<64>(xv, ok, c, sub, C, a.unimix, 0, uu, lg, idx);
          if (ok) {
            a.post_logit[row * S + g * C + c] = lg;
            a.post[row * F + D + g * C + c] = (c == idx) ? 1.f : 0.f;
            if (c == 0) a.idx[row * a.G + g] = idx;
          }
        }
      }
    }
    grid_barrier(a.ctr, ++gen * NWG);
  }
}

// class index of every one-hot group: idx[r][g] = argmax_c x[r][g*C + c]
__global__ void k_onehot_argmax(const float* __restrict__ x, long ldx, int* __restrict__ idx, int rows,
                                int G, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * G) return;
  const int r = i / G, g = i - r * G;
  const float* p = x + (long)r * ldx + (long)g * C;
  int best = 0;
  for (int c = 1; c < C; ++c) best = p[c] > p[best] ? c : best;
  idx[i] = best;
}

// Weight cache: W [K, N] fp32 (row stride ld) -> three bf16 planes [3][N][Kp], exact 3-way
// split, zero beyond K.
__global__ void k_scan_wprep(const float* __restrict__ W, long ld, int K, int N, int Kp,
                             unsigned short* __restrict__ out) {
  const long total = (long)N * Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
    const float x = k < K ? W[(long)k * ld + n] : 0.f;
    unsigned h, m, l;
    split3(x, h, m, l);
    out[i] = (unsigned short)(h >> 16);
    out[total + i] = (unsigned short)(m >> 16);
    out[2 * total + i] = (unsigned short)(l >> 16);
  }
}

}  // namespace

extern "C" int dd_scan_wprep(const float* W, long ld, int K, int N, int Kp, void* planes, void* stream) {
  DD_REQUIRE(Kp >= K && Kp % 32 == 0 && N % 16 == 0, "dd_scan_wprep: Kp multiple of 32 >= K, N multiple of 16");
  const long total = (long)N * Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  k_scan_wprep<<<blocks, 256, 0, (hipStream_t)stream>>>(W, ld, K, N, Kp, (unsigned short*)planes);
  DD_CHECK_LAUNCH("dd_scan_wprep");
  return 0;
}

extern "C" int dd_observe_scan_supported(int B, int D, int U, int G, int C, int A) {
  return B >= 1 && B <= 64 && D % 32 == 0 && U % 32 == 0 && (C == 16 || C == 32 || C == 64) &&
         G >= 1 && A >= 0;
}

extern "C" int dd_observe_scan_fwd(
    int B, int T, int D, int U, int G, int C, int A, int use_carry, float unimix,
    const float* first, const float* carry, const float* init_deter, const float* init_stoch,
    const float* u_post,
    const void* wt1, const void* wt2, const void* wt3, const void* wt4,
    const float* g1, const float* b1, const float* gg, const float* bg, const float* g3,
    const float* b3, const float* bias4,
    float* xin, float* z1, float* st1, float* gin, float* z3, float* gst, float* post, float* zo,
    float* xo, float* st3, float* xq, float* post_logit, const float* w_in, int* idx_ws,
    unsigned* sync2, void* stream) {
  DD_REQUIRE(dd_observe_scan_supported(B, D, U, G, C, A), "dd_observe_scan_fwd: unsupported shape");
  hipStream_t st = (hipStream_t)stream;
  ScanArgs a;
  a.B = B; a.T = T; a.D = D; a.U = U; a.G = G; a.C = C; a.A = A; a.S = G * C;
  a.XK = a.S + A; a.XKp = (a.XK + 31) / 32 * 32;
  a.use_carry = ((use_carry & 1) && carry != nullptr ? 1 : 0) | (use_carry & 2); a.unimix = unimix;
  a.first = first; a.carry = carry; a.init_deter = init_deter; a.init_stoch = init_stoch;
  a.u_post = u_post;
  a.wt1 = (const unsigned short*)wt1; a.wt2 = (const unsigned short*)wt2;
  a.wt3 = (const unsigned short*)wt3; a.wt4 = (const unsigned short*)wt4;
  a.g1 = g1; a.b1 = b1; a.gg = gg; a.bg = bg; a.g3 = g3; a.b3 = b3; a.bias4 = bias4;
  a.xin = xin; a.z1 = z1; a.st1 = st1; a.gin = gin; a.z3 = z3; a.gst = gst; a.post = post;
  a.zo = zo; a.xo = xo; a.st3 = st3; a.xq = xq; a.post_logit = post_logit; a.ctr = sync2;
  a.w_in = w_in;
  const long N = (long)B * T;
  a.idx = idx_ws; a.idx_carry = idx_ws + N * G; a.idx_init = idx_ws + (N + B) * G;
  static const int nwg_env = getenv("DD_SCAN_NWG") ? atoi(getenv("DD_SCAN_NWG")) : NWG_DEFAULT;
  a.nwg = nwg_env < 4 ? 4 : (nwg_env > 256 ? 256 : nwg_env / 4 * 4);
  if (a.use_carry & 1) {
    k_onehot_argmax<<<(B * G + 255) / 256, 256, 0, st>>>(carry + D, D + a.S, idx_ws + N * G, B, G, C);
    DD_CHECK_LAUNCH("dd_observe_scan_fwd(argmax carry)");
  }
  k_onehot_argmax<<<(G + 255) / 256, 256, 0, st>>>(init_stoch, a.S, idx_ws + (N + B) * G, 1, G, C);
  DD_CHECK_LAUNCH("dd_observe_scan_fwd(argmax init)");
  hipError_t e = hipMemsetAsync(sync2, 0, 2 * sizeof(unsigned), st);
  if (e != hipSuccess) { dd_set_error("dd_observe_scan_fwd(memset)", e); return (int)e; }
  k_observe_scan_fwd<<<a.nwg, 256, 0, st>>>(a);
  DD_CHECK_LAUNCH("dd_observe_scan_fwd");
  return 0;
}
