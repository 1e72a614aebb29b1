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
// Barrier: one monotonic counter per 16-row block (its 16 workgroups are the only producers a
// phase consumes), inter-workgroup outputs stored write-through (st_wt) so that the arrival is a
// drained-stores + relaxed agent-scope add WITHOUT a release fence (Guideline 16, form R1: the fence
// would write back the whole L2's dirty side outputs); together forward scan 1.55 -> 1.26 ms,
// reverse 1.68 -> 1.42 ms (profiles/r04_fused_scan_times.txt, flags 128 / 256 select the old forms).  Wait =
// relaxed agent-scope polling by one lane + acquire fence (MI355X_MICROARCH.md, valid forms);
// every spin is bounded - on a timeout bit 0 of the error word is set and the kernel runs on
// (garbage out, never a hang).  The error word is STICKY: launches reset the counter only, the
// host clears the word after reading it, and Learner.read_metrics raises on a non-zero word
// (like check_numerics, tfutils.py:207,249) - a step that timed out never trains silently.
// NWG <= CU count, one workgroup per CU, so all workgroups become resident as soon as
// concurrent grids of other streams drain (nothing they run waits on this kernel).
#include "latent_core.h"
#include <math.h>
#include <stdlib.h>
#include "../../include/daydreamer_hip.h"

namespace {

constexpr float LN_EPS = 1e-3f;
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

// `prefetch` runs between the arrival and the wait: loads that do not depend on the other
// workgroups' results (the next phase's weight planes) travel while the barrier completes.
// Write-through (sc1) store of a word another workgroup reads in a later phase (Guideline 16, form
// R1): the data goes past the XCD's L2 at once, so the arrival needs NO release fence - which
// would also write back every other dirty line of that L2, i.e. the ~1 MB per step of side outputs
// the backward pass reads after the launch.  Consumers keep the acquire fence + plain loads.
__device__ __forceinline__ void st_wt(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_wt(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `wt`: the phase's inter-workgroup outputs were stored write-through (st_wt): every wave drains
// its stores, then one lane arrives without a release fence.  !wt: plain stores + release fence (the
// protocol of rounds 2-3; flag bit 8, A/B measurements - the st_wt stores are harmless under it).
template <class PF>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned* err, unsigned target, bool wt, PF prefetch) {
  if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!wt) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  prefetch();
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) {
        __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    // (tried: the arrival's returned count so that the last arriver skips the poll, and the bare
    // `buffer_inv sc1` without the `s_waitcnt vmcnt(0)` the builtin puts in front of it, which drains
    // the prefetch above: 1.284 -> 1.262 ms forward, 1.435 -> 1.413 ms reverse - not worth leaving
    // the documented fence form)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned* err, unsigned target, bool wt) {
  grid_barrier(ctr, err, target, wt, [] {});
}

// Activations of the fused scan: every workgroup of a row block rebuilds the phase's whole
// operand, so the transcendental work is on the critical path 16 times over - hardware exp2 /
// reciprocal forms (<= 2 ulp of the libm results the per-layer kernels use).
__device__ __forceinline__ float fexp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float felu_(float x) { return x > 0.f ? x : fexp_(x) - 1.f; }
__device__ __forceinline__ float fsigmoid_(float x) { return __builtin_amdgcn_rcpf(1.f + fexp_(-x)); }
__device__ __forceinline__ float ftanh_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + fexp_(2.f * x)); }

// Sum over a row of the block from the lanes' partial sums: the lanes of a row are the four
// quads (lane & 15 equal) of the four waves.  Fixed order; result on every lane of the row.
__device__ __forceinline__ float row_reduce(float s, float (*ws)[16]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (lane < 16) ws[wave][lane] = s;
  __syncthreads();
  const int l = lane & 15;
  return (ws[0][l] + ws[1][l]) + (ws[2][l] + ws[3][l]);
}

// Up to MAXT output tiles (16 rows x 16 columns each, columns n0[j]) of one phase against the
// same operand rows: the A fragments af[NIT][3] (this wave's k-steps wave*32 + 128*it, three
// bf16 planes) and the weight planes bq of every tile come from the caller (the planes are loaded
// while the grid barrier before the phase completes); 6 MFMAs per k-step and tile; the four
// waves' partial tiles are summed through LDS in wave order.  out[j] = the complete element (row ((t&63)>>4)*4 + (t>>6), col t&15) of tile j.
// Plane layout (k_scan_wprep / k_scan_wprep_rows), FRAGMENT-MAJOR: [column tile n / 16][k-step of
// 128][plane][thread 0..255] x 16 bytes - thread (wave w, lane l) finds the eight k values
// k-step * 128 + w * 32 + (l >> 4) * 8 .. + 7 of column tile * 16 + (l & 15), i.e. its MFMA B fragment.
// One load instruction of a wave reads 1 KB contiguous (8 full cache lines); with [plane][n][k] rows
// it was 16 half lines, and the streamed phases at deter = units = 512 ran at 38 GB/s per CU where
// the L2 -> CU path gives 130-180 with contiguous blocks (tools/probes/stream_probe.hip).
__device__ __forceinline__ long plane_index(int n, int k, int p, int Kp) {
  const int tile = n >> 4, r = n & 15, it = k >> 7, kk = k & 127, w = kk >> 5, q = (kk & 31) >> 3, e = kk & 7;
  return ((((long)(tile * (Kp >> 7) + it) * 3 + p) * 4 + w) * 64 + (q * 16 + r)) * 8 + e;
}

// k-steps it0 .. it0 + NIT - 1 of the column tile that starts at n0
template <int NIT>
__device__ __forceinline__ void load_planes(uint4 (&bq)[NIT][3], const unsigned short* wt, int KP, int n0, int it0 = 0) {
  // wave-uniform base (tile, k-step, plane: scalar registers) + one 32-bit thread offset: per-(tile,
  // plane) 64-bit vector addresses would be hoisted out of the time loop by the compiler and, for
  // every tile of every phase, fill the register file
  const unsigned uoff = __builtin_amdgcn_readfirstlane((unsigned)(((n0 >> 4) * (KP >> 7) + it0) * (3 * 4096)));
  const char* base = reinterpret_cast<const char*>(wt) + uoff;
  const unsigned lane_off = threadIdx.x * 16u;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int p = 0; p < 3; ++p) bq[it][p] = *reinterpret_cast<const uint4*>(base + (it * 3 + p) * 4096 + lane_off);
}

template <int KP, int MAXT>
__device__ __forceinline__ void tiles_gemm(const bf16x8 (&af)[KP / 128][3], const uint4 (&bq)[MAXT][KP / 128][3],
                                           float (*red)[4][256], float (&out)[MAXT]) {
  constexpr int NIT = KP / 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();   // red[] free (the previous phase's readers are done)
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bq[j][it][p]);
      // six cross products, smallest terms first (as k_mfma_gemm_s3)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it][2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it][0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it][1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it][1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it][0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it][0], b[0], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[j][wave][r * 64 + lane] = acc[r];
  }
  __syncthreads();
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < MAXT; ++j) out[j] = ((red[j][0][t] + red[j][1][t]) + red[j][2][t]) + red[j][3][t];
}

// The same with the weight planes streamed in units of NITC k-steps of a tile: two units in
// registers, unit u + 1 is loaded while unit u runs its MFMAs; unit 0 comes from the caller
// (loaded during the grid barrier) - for phases whose tiles' planes together exceed the register
// file (deter = units = 512: the GRU contraction has six 1024-deep tiles per workgroup).
// n0[j]: first column of tile j.
template <int KP, int MAXT, int NITC, int DEPTH = 2>
__device__ __forceinline__ void tiles_gemm_stream(const bf16x8 (&af)[KP / 128][3], uint4 (&bq)[DEPTH][NITC][3],
                                                  const unsigned short* wt,
                                                  const int (&n0)[MAXT], float (*red)[4][256],
                                                  float (&out)[MAXT]) {
  // DEPTH units in registers: unit u + DEPTH - 1 is requested while unit u runs its MFMAs (a unit is
  // 12 KB per wave).  With the fragment-major planes two units reach the L2 -> CU rate (590 KB of
  // Q3's planes in 5.0 us = 118 GB/s per CU); deeper only costs registers.
  constexpr int NIT = KP / 128, H = NIT / NITC, NU = MAXT * H;
  static_assert(NIT % NITC == 0, "k-steps per unit");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();   // red[] free (the previous phase's readers are done)
#pragma unroll
  for (int v = 1; v < DEPTH - 1; ++v)
    if (v < NU) load_planes<NITC>(bq[v], wt, KP, n0[v / H], (v % H) * NITC);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int j = u / H, h = u % H, v = u + DEPTH - 1;
    if (v < NU)
      load_planes<NITC>(bq[v % DEPTH], wt, KP, n0[v / H], (v % H) * NITC);
    __builtin_amdgcn_sched_barrier(0);   // (no further hoisting: DEPTH - 1 units ahead is what the registers hold)
#pragma unroll
    for (int it = 0; it < NITC; ++it) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bq[u % DEPTH][it][p]);
      const int ia = h * NITC + it;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ia][2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ia][0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ia][1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ia][1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ia][0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ia][0], b[0], acc, 0, 0, 0);
    }
    if (h == H - 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[j][wave][r * 64 + lane] = acc[r];
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < MAXT; ++j) out[j] = ((red[j][0][t] + red[j][1][t]) + red[j][2][t]) + red[j][3][t];
}

constexpr int NWG = 64, NSTR = NWG / 4;   // 4 row blocks of 16 batch rows x 16 column strides
// Barrier scope: a phase only consumes what the 16 workgroups of its own row block produced, so each
// row block has its own counter (sync2[576 + 128 m], 128 words apart behind the debug stamps; the
// error word stays at sync2[1]).  Flag bit 7 of use_carry / flags selects the grid-wide counter
// sync2[0] instead (A/B measurements, tools/scan_time.py).
// Workgroup -> (row block, column stride).  Workgroups go to the 8 XCDs round-robin (wg % 8) and
// every XCD has its own 4 MB L2.  Each XCD gets all four row blocks of TWO column strides, so it
// streams 2 / 16 of every weight matrix: at deter = units = 512 that is 2.1 MB of planes per time
// step, which stays in its L2 from step to step; with wg -> (wg & 3, wg >> 2) (flag bit 9) an XCD
// held one row block and 8 strides, 8.6 MB per step - every step's weights came over the fabric
// again (Q3 of the reverse scan: 16 us of its 26).  The 16 workgroups of a row block sit on all 8
// XCDs either way they exchange through write-through stores.
__device__ __forceinline__ void scan_wg_map(int wg, int old_map, int& mblk, int& nstr) {
  if (old_map) { mblk = wg & 3; nstr = wg >> 2; return; }
  const int x = wg & 7, k = wg >> 3;
  mblk = k & 3;
  nstr = x * 2 + (k >> 2);
}

#define RB_CTR(mblk) (bar_rb ? 576 + 128 * (mblk) : 0)
#define RB_N (bar_rb ? NSTR : NWG)
constexpr int cdiv_(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax_(int a, int b) { return a > b ? a : b; }

// The (16 rows x 8 columns) chunk of a phase's operand this lane built, written by the one
// workgroup of the row block that owns it (chunk index modulo the column strides): the side
// outputs the backward pass reads come straight from the registers.
template <int NIT>
__device__ __forceinline__ void store_chunks(const float (&v)[NIT][8], float* rowp, int kq, int nstr, bool live) {
  if (!live) return;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (((it * 16 + (kq >> 3)) % NSTR) == nstr) {
      float* q = rowp + it * 128 + kq;
      *reinterpret_cast<float4*>(q) = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
      *reinterpret_cast<float4*>(q + 4) = make_float4(v[it][4], v[it][5], v[it][6], v[it][7]);
    }
  }
}

// one k-step of store_chunks
__device__ __forceinline__ void store_chunk1(const float (&v)[8], float* rowp, int it, int kq, int nstr, bool live) {
  if (live && ((it * 16 + (kq >> 3)) % NSTR) == nstr) {
    float* q = rowp + it * 128 + kq;
    *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

#define TSW(i)                                                                             \
  if ((a.use_carry & 64) && threadIdx.x == 0 && t == 10)                                    \
  reinterpret_cast<unsigned long long*>(a.ctr + 64)[blockIdx.x * 4 + (i)] = wall_clock64()
#define TS(i)                                                                              \
  if ((a.use_carry & 64) && blockIdx.x == 0 && threadIdx.x == 0 && t == 10)                \
  reinterpret_cast<unsigned long long*>(a.ctr + 2)[i] = wall_clock64()

template <int D, int U, int G, int C, int A>
__global__ void __launch_bounds__(256, 1)
k_observe_scan_fwd(ScanArgs a) {
  constexpr int S = G * C, F = D + S, XK = S + A;
  static_assert(D % 128 == 0 && U % 128 == 0 && C % 16 == 0 && G % 4 == 0, "dims");
  static_assert(U / 16 % NSTR == 0 && (3 * D / 16) % NSTR == 0 && G % NSTR == 0 && S % (NSTR * 64) == 0,
                "every workgroup owns the same number of column tiles");
  // column tiles per workgroup and phase
  constexpr int T1 = U / 16 / NSTR, T2 = 3 * D / 16 / NSTR, T3 = T1;
  constexpr int TPG = C / 16, GPP = G / NSTR, T4 = GPP * TPG;
  constexpr int TMAX = cmax_(cmax_(T2, T3), T4);
  constexpr int NIT2 = (D + U) / 128, NIT3 = D / 128, NIT4 = U / 128;
  __shared__ float red[TMAX][4][256];
  // LayerNorm scales / offsets, the initial state and the stats bias: read every step by every
  // lane, kept in LDS (a global read after each barrier's acquire would go to L2)
  constexpr int O_G1 = 0, O_B1 = U, O_GG = 2 * U, O_BG = O_GG + 3 * D, O_G3 = O_BG + 3 * D,
                O_B3 = O_G3 + U, O_INIT = O_B3 + U, O_BIAS = O_INIT + D, NPAR = O_BIAS + S;
  __shared__ __attribute__((aligned(16))) float par[NPAR];
  __shared__ int cls_init[G];
  __shared__ __attribute__((aligned(16))) float xs[16][T4 * 16 + 4];   // P4: the stats of this workgroup's groups
  __shared__ float ws_a[4][16], ws_b[4][16];   // row_reduce scratch (mean, variance)
  for (int i = threadIdx.x; i < NPAR; i += 256) {
    float v;
    if (i < O_B1) v = a.g1[i];
    else if (i < O_GG) v = a.b1[i - O_B1];
    else if (i < O_BG) v = a.gg[i - O_GG];
    else if (i < O_G3) v = a.bg[i - O_BG];
    else if (i < O_B3) v = a.g3[i - O_G3];
    else if (i < O_INIT) v = a.b3[i - O_B3];
    else if (i < O_BIAS) v = a.init_deter[i - O_INIT];
    else v = a.bias4[i - O_BIAS];
    par[i] = v;
  }
  if (threadIdx.x < G) cls_init[threadIdx.x] = a.idx_init[threadIdx.x];
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int mblk, nstr;
  scan_wg_map(wg, a.use_carry & 512, mblk, nstr);
  const int T = a.T;
  const int orow = ((lane >> 4) * 4) + (tid >> 6), ocol = tid & 15;   // element of a finished tile
  const int ob = mblk * 16 + orow;                                    // its batch row
  const bool olive = ob < a.B;
  const int ab = min(mblk * 16 + (lane & 15), a.B - 1);               // batch row of the A operand
  const bool alive = mblk * 16 + (lane & 15) < a.B;
  const int kq = (lane >> 4) * 8 + wave * 32;                         // this lane's k offset in a k-step
  const bool bar_rb = !(a.use_carry & 128), bar_wt = !(a.use_carry & 256);
  unsigned gen = 0;
  if (a.use_carry & 2) {   // measurement aid: the barriers alone (4 per step), no work
    for (int i = 0; i < 4 * T; ++i) grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt);
    return;
  }
  // P1's dense action columns of W_img_in for this thread's output columns: step-invariant
  const int r1 = tid >> 4, c1 = tid & 15;
  float wa[T1][A];
#pragma unroll
  for (int jt = 0; jt < T1; ++jt)
#pragma unroll
    for (int j = 0; j < A; ++j) wa[jt][j] = a.w_in[(long)(S + j) * U + (nstr + NSTR * jt) * 16 + c1];
  // deter_{t-1} of this lane's operand chunk (every workgroup of the row block computes the whole
  // GRU output in P3): carried in registers from step to step
  float hcur[NIT3][8];
#pragma unroll
  for (int it = 0; it < NIT3; ++it) {
    if (a.use_carry & 1) ld8(a.carry + (long)ab * F + it * 128 + kq, hcur[it]);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) hcur[it][j] = 0.f;
    }
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const long arow = (long)ab * T + t;
    const float af_ = a.first[arow];
    const long oidx = (long)ob * T + t;

    // (use_carry bits 2..5: measurement aid, skip the body of P1..P4; bit 6: time stamps of
    // step 10 on workgroup 0 into ctr[2..], 100 MHz wall clock)
    TS(0);
    // ---------------- P1: z1 = [mask(stoch) | masked action] @ W_img_in
    // The stoch part of the operand is one-hot per group: its product with W is the SUM of the
    // G selected rows of W (1.0 * w is exact, the zero terms vanish) - a gather instead of a
    // K = S contraction; the A action columns are a short dense product.  fp32 adds in group order.
    if (!(a.use_carry & 4)) {
      // thread -> (row r1 = tid >> 4 of the block, column c1 = tid & 15 of the tile)
      const int b = min(mblk * 16 + r1, a.B - 1);
      const bool live = mblk * 16 + r1 < a.B;
      const long row = (long)b * T + t;
      // class of every group of this row's masked stoch: the previous draw, or the initial
      // state's (is_first), or none (-1: no carry at t = 0, zero vector)
      const int* pidx = t > 0 ? a.idx + (row - 1) * G : ((a.use_carry & 1) ? a.idx_carry + (long)b * G : nullptr);
      const float f = a.first[row];
      int cls[G];
#pragma unroll
      for (int g4 = 0; g4 < G; g4 += 4) {
        const int4 q = pidx ? *reinterpret_cast<const int4*>(pidx + g4) : make_int4(-1, -1, -1, -1);
        cls[g4] = q.x; cls[g4 + 1] = q.y; cls[g4 + 2] = q.z; cls[g4 + 3] = q.w;
      }
      float act[A];
#pragma unroll
      for (int j = 0; j < A; ++j) act[j] = a.xin[row * XK + S + j];
      // this workgroup's slice of the masked stoch input (the bulk weight gradient reads xin)
      int xcls[S / (NSTR * 64)];
#pragma unroll
      for (int q = 0; q < S / (NSTR * 64); ++q) {
        const int col = ((q * NSTR + nstr) * 16 + c1) * 4;
        xcls[q] = pidx ? pidx[col / C] : -1;
      }
      if (f != 0.f) {
#pragma unroll
        for (int g = 0; g < G; ++g) cls[g] = cls_init[g];
#pragma unroll
        for (int q = 0; q < S / (NSTR * 64); ++q) xcls[q] = cls_init[(((q * NSTR + nstr) * 16 + c1) * 4) / C];
      }
#pragma unroll
      for (int jt = 0; jt < T1; ++jt) {
        const int n = (nstr + NSTR * jt) * 16 + c1;
        float w[G];
#pragma unroll
        for (int g = 0; g < G; ++g) w[g] = cls[g] >= 0 ? a.w_in[(long)(g * C + cls[g]) * U + n] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) acc += w[g];
#pragma unroll
        for (int j = 0; j < A; ++j) acc += act[j] * wa[jt][j];
        if (live) st_wt(a.z1 + row * U + n, acc);
      }
#pragma unroll
      for (int q = 0; q < S / (NSTR * 64); ++q) {
        const int col = ((q * NSTR + nstr) * 16 + c1) * 4, cc = col % C;
        if (live)
          *reinterpret_cast<float4*>(a.xin + row * XK + col) =
              make_float4(xcls[q] == cc ? 1.f : 0.f, xcls[q] == cc + 1 ? 1.f : 0.f,
                          xcls[q] == cc + 2 ? 1.f : 0.f, xcls[q] == cc + 3 ? 1.f : 0.f);
      }
    }
    TS(1); TSW(0);
    // (weight planes of all T2 tiles in registers if they fit - 48 vector registers per tile and
    // k-step of 128 - else streamed in units of four k-steps, two units in flight)
    constexpr bool ALL2 = T2 * NIT2 <= 12;
    constexpr int NITC2 = NIT2 % 4 == 0 ? 4 : NIT2;
    uint4 bq2[ALL2 ? T2 : 2][ALL2 ? NIT2 : NITC2][3];
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt, [&] {
      if constexpr (ALL2) {
#pragma unroll
        for (int j = 0; j < T2; ++j) load_planes<NIT2>(bq2[j], a.wt2, D + U, (nstr + NSTR * j) * 16);
      } else {
        load_planes<NITC2>(bq2[0], a.wt2, D + U, nstr * 16);
      }
    });
    TS(2);

    // ---------------- P2: z3 = [mask(deter) | ELU(LN(z1))] @ W_gru
    if (!(a.use_carry & 8)) {
      constexpr int KP = D + U, NIT = NIT2;
      // the operand loads go first, the statistics' row loads follow them
      float raw[NIT][8];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = it * 128 + kq;
        if (it * 128 >= D) ld8(a.z1 + arow * U + (k - D), raw[it]);   // (D % 128 == 0: a k-step lies in one part)
      }
      // LayerNorm statistics of z1 from the operand registers (the row block's lanes hold the
      // whole row between them)
      float ps = 0.f;
#pragma unroll
      for (int it = D / 128; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) ps += raw[it][j];
      const float mean = row_reduce(ps, ws_a) / (float)U;
      float pv = 0.f;
#pragma unroll
      for (int it = D / 128; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) pv += (raw[it][j] - mean) * (raw[it][j] - mean);
      const float rstd = rsqrtf(row_reduce(pv, ws_b) / (float)U + LN_EPS);
      TS(3);
      bf16x8 afr[NIT][3];
      float v[NIT][8];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = it * 128 + kq;
        float gm[8], bt[8];
        if (it * 128 < D) {
          ld8(par + O_INIT + k, gm);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[it][j] = hcur[it][j] * (1.f - af_) + gm[j] * af_;
        } else {
          ld8(par + O_G1 + (k - D), gm);
          ld8(par + O_B1 + (k - D), bt);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[it][j] = felu_((raw[it][j] - mean) * rstd * gm[j] + bt[j]);
        }
        split8(v[it], afr[it]);
      }
      // side outputs: [hprev | x1] (the GRU operand) and the LayerNorm statistics
      store_chunks<NIT>(v, a.gin + arow * KP, kq, nstr, alive);
#pragma unroll
      for (int it = 0; it < NIT3; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) hcur[it][j] = v[it][j];   // the masked deter_{t-1}: P3's h
      if (nstr == 0 && tid < 16 && alive) *reinterpret_cast<float2*>(a.st1 + arow * 2) = make_float2(mean, rstd);
      int n0[T2];
#pragma unroll
      for (int j = 0; j < T2; ++j) n0[j] = (nstr + NSTR * j) * 16;
      float out[T2];
      TS(4);
      if constexpr (ALL2) tiles_gemm<KP, T2>(afr, bq2, red, out);
      else tiles_gemm_stream<KP, T2, NITC2>(afr, bq2, a.wt2, n0, red, out);
      TS(5);
#pragma unroll
      for (int j = 0; j < T2; ++j)
        if (olive) st_wt(a.z3 + oidx * 3 * D + n0[j] + ocol, out[j]);
    }
    TS(6); TSW(1);
    uint4 bq3[T3][NIT3][3];
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt, [&] {
#pragma unroll
      for (int j = 0; j < T3; ++j) load_planes<NIT3>(bq3[j], a.wt3, D, (nstr + NSTR * j) * 16);
    });
    TS(7);

    // ---------------- P3: zo += GRU(LN(z3), hprev) @ W_obs_out[:D]
    if (!(a.use_carry & 16)) {
      constexpr int NIT = NIT3;
      float z0[NIT][8], z1_[NIT][8], z2[NIT][8];
      const float* zr = a.z3 + arow * 3 * D;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = it * 128 + kq;
        ld8(zr + k, z0[it]); ld8(zr + D + k, z1_[it]); ld8(zr + 2 * D + k, z2[it]);
      }
      // zo's old value (the embedding part, written before the scan) for the accumulate
      float zold[T3];
#pragma unroll
      for (int j = 0; j < T3; ++j) zold[j] = olive ? a.zo[oidx * U + (nstr + NSTR * j) * 16 + ocol] : 0.f;
      float ps = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) ps += (z0[it][j] + z1_[it][j]) + z2[it][j];
      const float mean = row_reduce(ps, ws_a) / (float)(3 * D);
      float pv = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d0 = z0[it][j] - mean, d1 = z1_[it][j] - mean, d2 = z2[it][j] - mean;
          pv += (d0 * d0 + d1 * d1) + d2 * d2;
        }
      const float rstd = rsqrtf(row_reduce(pv, ws_b) / (float)(3 * D) + LN_EPS);
      TS(8);
      bf16x8 afr[NIT][3];
      float v[NIT][8];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = it * 128 + kq;
        float g0[8], g1_[8], g2[8], b0[8], b1_[8], b2[8];
        ld8(par + O_GG + k, g0); ld8(par + O_GG + D + k, g1_); ld8(par + O_GG + 2 * D + k, g2);
        ld8(par + O_BG + k, b0); ld8(par + O_BG + D + k, b1_); ld8(par + O_BG + 2 * D + k, b2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yr = (z0[it][j] - mean) * rstd * g0[j] + b0[j];
          const float yc = (z1_[it][j] - mean) * rstd * g1_[j] + b1_[j];
          const float yu = (z2[it][j] - mean) * rstd * g2[j] + b2[j];
          const float rr = fsigmoid_(yr), cand = ftanh_(rr * yc), uu = fsigmoid_(yu - 1.f);
          v[it][j] = uu * cand + (1.f - uu) * hcur[it][j];
        }
        split8(v[it], afr[it]);
      }
      // side outputs: deter_t and the GRU's LayerNorm statistics
      store_chunks<NIT>(v, a.post + arow * F, kq, nstr, alive);
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) hcur[it][j] = v[it][j];
      if (nstr == 0 && tid < 16 && alive) *reinterpret_cast<float2*>(a.gst + arow * 2) = make_float2(mean, rstd);
      int n0[T3];
#pragma unroll
      for (int j = 0; j < T3; ++j) n0[j] = (nstr + NSTR * j) * 16;
      float out[T3];
      TS(9);
      tiles_gemm<D, T3>(afr, bq3, red, out);
      TS(10);
#pragma unroll
      for (int j = 0; j < T3; ++j)
        if (olive) st_wt(a.zo + oidx * U + n0[j] + ocol, zold[j] + out[j]);
    }
    TS(11); TSW(2);
    constexpr bool ALL4 = T4 * NIT4 <= 12;
    constexpr int NITC4 = NIT4 % 4 == 0 ? 4 : NIT4;
    uint4 bq4[ALL4 ? T4 : 2][ALL4 ? NIT4 : NITC4][3];
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt, [&] {
      if constexpr (ALL4) {
#pragma unroll
        for (int j = 0; j < T4; ++j)
          load_planes<NIT4>(bq4[j], a.wt4, U, (nstr + NSTR * (j / TPG)) * C + (j % TPG) * 16);
      } else {
        load_planes<NITC4>(bq4[0], a.wt4, U, nstr * C);
      }
    });
    TS(12);

    // ---------------- P4: xq = ELU(LN(zo)) @ W_obs_stats + b; sample
    if (!(a.use_carry & 32)) {
      constexpr int NIT = NIT4;
      float raw[NIT][8];
#pragma unroll
      for (int it = 0; it < NIT; ++it) ld8(a.zo + arow * U + it * 128 + kq, raw[it]);
      // the uniforms of this workgroup's (row, group) items
      constexpr int LWc = C <= 16 ? 16 : (C <= 32 ? 32 : 64);
      constexpr int per_wave = 64 / LWc, NPASS = cdiv_(16, 4 * per_wave);
      float uu[GPP][NPASS];
#pragma unroll
      for (int gi = 0; gi < GPP; ++gi)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int item = ps * 4 * per_wave + wave * per_wave + lane / LWc;
          const int b = mblk * 16 + item;
          uu[gi][ps] = (item < 16 && b < a.B) ? a.u_post[((long)t * a.B + b) * G + nstr + NSTR * gi] : 0.f;
        }
      float ps = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) ps += raw[it][j];
      const float mean = row_reduce(ps, ws_a) / (float)U;
      float pv = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) pv += (raw[it][j] - mean) * (raw[it][j] - mean);
      const float rstd = rsqrtf(row_reduce(pv, ws_b) / (float)U + LN_EPS);
      TS(13);
      bf16x8 afr[NIT][3];
      float v[NIT][8];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = it * 128 + kq;
        float gm[8], bt[8];
        ld8(par + O_G3 + k, gm); ld8(par + O_B3 + k, bt);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] = felu_((raw[it][j] - mean) * rstd * gm[j] + bt[j]);
        split8(v[it], afr[it]);
      }
      // side outputs: xo (the stats layer's operand) and the LayerNorm statistics
      store_chunks<NIT>(v, a.xo + arow * U, kq, nstr, alive);
      if (nstr == 0 && tid < 16 && alive) *reinterpret_cast<float2*>(a.st3 + arow * 2) = make_float2(mean, rstd);
      // whole latent groups (TPG column tiles each) per workgroup, so that the draw of a
      // (row, group) only needs statistics this workgroup computed
      int n0[T4];
#pragma unroll
      for (int j = 0; j < T4; ++j) n0[j] = (nstr + NSTR * (j / TPG)) * C + (j % TPG) * 16;
      float out[T4];
      TS(14);
      if constexpr (ALL4) tiles_gemm<U, T4>(afr, bq4, red, out);
      else tiles_gemm_stream<U, T4, NITC4>(afr, bq4, a.wt4, n0, red, out);
      TS(15);
#pragma unroll
      for (int j = 0; j < T4; ++j) {
        const float x = out[j] + par[O_BIAS + n0[j] + ocol];
        xs[orow][j * 16 + ocol] = x;
        if (olive) a.xq[oidx * S + n0[j] + ocol] = x;
      }
      __syncthreads();
      TS(16);
      // 16 (row, group) items per group, one LWc-lane sub-wave each; this lane's GPP * NPASS
      // items run in lock-step
      {
        constexpr int NI = GPP * NPASS;
        const int sub = lane / LWc, c = lane % LWc;
        float xv[NI], uq[NI], lg[NI];
        bool ok[NI];
        int idx[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int gi = i / NPASS, ps2 = i % NPASS;
          const int item = ps2 * 4 * per_wave + wave * per_wave + sub;
          ok[i] = item < 16 && mblk * 16 + item < a.B && c < C;
          xv[i] = ok[i] ? xs[item & 15][gi * C + c] : -INFINITY;
          uq[i] = uu[gi][ps2];
        }
        stats_items<LWc, NI>(xv, ok, c, sub, C, a.unimix, 0, uq, lg, idx);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int gi = i / NPASS, ps2 = i % NPASS;
          const int item = ps2 * 4 * per_wave + wave * per_wave + sub;
          const int g = nstr + NSTR * gi;
          if (ok[i]) {
            const long row = (long)(mblk * 16 + item) * T + t;
            a.post_logit[row * S + g * C + c] = lg[i];
            a.post[row * F + D + g * C + c] = (c == idx[i]) ? 1.f : 0.f;
            if (c == 0) st_wt(a.idx + row * G + g, idx[i]);
          }
        }
      }
    }
    TS(17); TSW(3);
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt);
    TS(18);
  }
}
#undef TS
#undef TSW

// ===========================================================================================
// Fused reverse scan: the data-gradient backward of the T obs_steps in one persistent launch.
//
// Reference: the gradient tf.GradientTape derives for RSSM.observe (nets.py:66-76, 99-160); the
// launch sequence it replaces is Learner.observe_bwd's loop (stats_bwd, obs_stats dgrad,
// LN-ELU bwd, obs_out dgrad, GRU bwd, GRU dgrad, LN-ELU bwd, img_in dgrad, reset mask).
// Same decomposition as the forward scan (4 row blocks x 16 column strides, consumer-side
// row-wise work, weight planes prefetched during the barrier); per step t = T-1 .. 0:
//   Q1  dxo   = dxq_t @ W_stats^T                                       (K = S)
//   Q2  dzo   = LN-ELU'(dxo; zo, xo);  ddeter_t += dzo @ W_out_h^T      (K = U)
//   Q3  dz3, dy3, dh = GRU'(ddeter_t; z3, hprev);  [dh | dx1] += dz3 @ W_gru^T   (K = 3D)
//       dfeat[t-1, :D] += (1 - first_t) * dh
//   Q4  dz1   = LN-ELU'(dx1; z1, x1);  dxs = dz1 @ W_in_s^T             (K = U)
//       dfeat[t-1, D:] += (1 - first_t) * dxs;  dxq_{t-1} = stats'(xq_{t-1}; dlogit, dfeat[t-1, D:])
// dxq_{T-1} comes from the caller (dd_stats_sample_bwd on the last step's rows).  Every buffer
// the bulk weight-gradient contractions read afterwards (dxq, dxo, dzo, dz3, dy3, [dh | dx1], dz1,
// dxs) is written as the launch sequence writes it.
struct ScanBwdArgs {
  int B, T, flags;
  float unimix;
  const float* first;      // [N]
  // forward activations (rows b*T + t)
  const float *xq, *zo, *xo, *st3, *z3, *gst, *gin, *z1, *st1;
  const float* dlogit;     // [N, S] KL gradient w.r.t. the posterior logits
  // weight planes [3][N][K] (rows = output columns of the backward contraction)
  const unsigned short *w1, *w2, *w3, *w4;
  const float *g3, *gg, *bg, *g1;   // LayerNorm scales (GRU: scale and offset)
  // gradients
  float* dfeat;            // [N, D+S] in/out
  float* dxq;              // [N, S]
  float* dxo;              // [N, U]
  float* dzo;              // [N, U]
  float* dz3;              // [N, 3D]
  float* dy3;              // [N, 3D]
  float* dgin;             // [N, D+U]  [dh | dx1]
  float* dz1;              // [N, U]
  float* dxs;              // [N, S]
  unsigned* ctr;
};

// two row sums at once (see row_reduce)
__device__ __forceinline__ void row_reduce2(float& a, float& b, float (*ws)[2][16]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
  a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
  if (lane < 16) { ws[wave][0][lane] = a; ws[wave][1][lane] = b; }
  __syncthreads();
  const int l = lane & 15;
  a = (ws[0][0][l] + ws[1][0][l]) + (ws[2][0][l] + ws[3][0][l]);
  b = (ws[0][1][l] + ws[1][1][l]) + (ws[2][1][l] + ws[3][1][l]);
}

// sum over the 16 lanes that hold one row of a finished tile (lanes with equal lane >> 4)
__device__ __forceinline__ float tile_row_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 16);
  return v;
}
__device__ __forceinline__ float tile_row_max(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 16));
  return v;
}

// LayerNorm + ELU backward on the operand registers (k_ln_act_bwd): dy at the ELU output ->
// dz at the LayerNorm input, W = row width, NIT k-steps of this lane.
template <int W, int NIT>
__device__ __forceinline__ void ln_elu_bwd(const float (&dy)[NIT][8], const float (&z)[NIT][8],
                                           const float (&o)[NIT][8], float mean, float rstd,
                                           const float* gamma_lds, int kq, float (*ws)[2][16],
                                           float (&dz)[NIT][8]) {
  float g[NIT][8], xh[NIT][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float gm[8];
    ld8(gamma_lds + it * 128 + kq, gm);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = dy[it][j] * (o[it][j] > 0.f ? 1.f : o[it][j] + 1.f);
      xh[it][j] = (z[it][j] - mean) * rstd;
      g[it][j] = d * gm[j];
      s1 += g[it][j];
      s2 += g[it][j] * xh[it][j];
    }
  }
  row_reduce2(s1, s2, ws);
  s1 /= (float)W; s2 /= (float)W;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) dz[it][j] = rstd * (g[it][j] - s1 - xh[it][j] * s2);
}

#define TSB(i)                                                                             \
  if ((a.flags & 64) && blockIdx.x == 0 && threadIdx.x == 0 && t == 10)                     \
  reinterpret_cast<unsigned long long*>(a.ctr + 2)[i] = wall_clock64()

template <int D, int U, int G, int C>
__global__ void __launch_bounds__(256, 1)
k_observe_scan_bwd(ScanBwdArgs a) {
  constexpr int S = G * C, F = D + S;
  static_assert(D % 128 == 0 && U % 128 == 0 && C % 16 == 0, "dims");
  static_assert(U % (16 * NSTR) == 0 && D % (16 * NSTR) == 0 && G % NSTR == 0,
                "whole column tiles per workgroup in Q1 / Q2, whole groups in Q4");
  constexpr int T1 = U / 16 / NSTR, T2 = D / 16 / NSTR;   // Q1 / Q2 column tiles per workgroup
  constexpr int T3 = (D + U) / 16 / NSTR;        // Q3 column tiles per workgroup ([dh | dx1])
  constexpr int TPG = C / 16, GPP = G / NSTR, T4 = GPP * TPG;
  constexpr int TMAX = cmax_(cmax_(T1, T2), cmax_(T3, T4));
  constexpr int NIT1 = S / 128, NIT2 = U / 128, NIT3 = 3 * D / 128, NIT4 = U / 128, ND = D / 128;
  // weight planes of a phase: every tile in registers if they fit (deter = units = 256: all four
  // phases), else streamed in units of four k-steps, two units in flight (tiles_gemm_stream;
  // deter = units = 512: Q1, Q3, Q4 - Q3 alone would be 576 registers)
  constexpr bool ALL1 = T1 * NIT1 <= 12, ALL2 = T2 * NIT2 <= 12, ALL3 = T3 * NIT3 <= 12, ALL4 = T4 * NIT4 <= 12;
  constexpr int NITC1 = NIT1 % 4 == 0 ? 4 : NIT1, NITC3 = NIT3 % 4 == 0 ? 4 : NIT3, NITC4 = NIT4 % 4 == 0 ? 4 : NIT4;
  static_assert(ALL2, "Q2's planes are expected to fit");
  constexpr int DEPTH3 = 2;   // units of Q3's stream in registers (3: 36 spilled registers, 4: 100 - both slower: 1.15 / 1.18 / 1.35 ms at xarm's T = 32)
  __shared__ float red[TMAX][4][256];
  constexpr int O_G3 = 0, O_GG = U, O_BG = O_GG + 3 * D, O_G1 = O_BG + 3 * D, NPAR = O_G1 + U;
  __shared__ __attribute__((aligned(16))) float par[NPAR];
  __shared__ __attribute__((aligned(16))) float dh_lds[16][D + 4];   // Q3: (1 - update) * dhn of the row block
  __shared__ float ws_a[4][2][16], ws_b[4][2][16];
  for (int i = threadIdx.x; i < NPAR; i += 256) {
    float v;
    if (i < O_GG) v = a.g3[i];
    else if (i < O_BG) v = a.gg[i - O_GG];
    else if (i < O_G1) v = a.bg[i - O_BG];
    else v = a.g1[i - O_G1];
    par[i] = v;
  }
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int mblk, nstr;
  scan_wg_map(wg, a.flags & 512, mblk, nstr);
  const int T = a.T;
  const int orow = ((lane >> 4) * 4) + (tid >> 6), ocol = tid & 15;
  const int ob = mblk * 16 + orow;
  const bool olive = ob < a.B;
  const int ab = min(mblk * 16 + (lane & 15), a.B - 1);
  const bool alive = mblk * 16 + (lane & 15) < a.B;
  const int kq = (lane >> 4) * 8 + wave * 32;
  const float um = 1.f - a.unimix;
  const bool bar_rb = !(a.flags & 128), bar_wt = !(a.flags & 256);
  unsigned gen = 0;
  __syncthreads();
  uint4 bq1[ALL1 ? T1 : 2][ALL1 ? NIT1 : NITC1][3];
  auto load_q1 = [&] {
    if constexpr (ALL1) {
#pragma unroll
      for (int j = 0; j < T1; ++j) load_planes<NIT1>(bq1[j], a.w1, S, (nstr + NSTR * j) * 16);
    } else {
      load_planes<NITC1>(bq1[0], a.w1, S, nstr * 16);
    }
  };
  load_q1();
  for (int t = T - 1; t >= 0; --t) {
    const long arow = (long)ab * T + t;
    const long oidx = (long)ob * T + t;
    TSB(0);
    // ---------------- Q1: dxo = dxq_t @ W_stats^T
    {
      float raw[NIT1][8];
#pragma unroll
      for (int it = 0; it < NIT1; ++it) ld8(a.dxq + arow * S + it * 128 + kq, raw[it]);
      bf16x8 afr[NIT1][3];
#pragma unroll
      for (int it = 0; it < NIT1; ++it) split8(raw[it], afr[it]);
      int n0[T1];
#pragma unroll
      for (int j = 0; j < T1; ++j) n0[j] = (nstr + NSTR * j) * 16;
      float out[T1];
      if constexpr (ALL1) tiles_gemm<S, T1>(afr, bq1, red, out);
      else tiles_gemm_stream<S, T1, NITC1>(afr, bq1, a.w1, n0, red, out);
#pragma unroll
      for (int j = 0; j < T1; ++j)
        if (olive) st_wt(a.dxo + oidx * U + n0[j] + ocol, out[j]);
    }
    TSB(1);
    uint4 bq2[T2][NIT2][3];
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt, [&] {
#pragma unroll
      for (int j = 0; j < T2; ++j) load_planes<NIT2>(bq2[j], a.w2, U, (nstr + NSTR * j) * 16);
    });
    TSB(2);

    // ---------------- Q2: dzo = LN-ELU'(dxo);  ddeter_t += dzo @ W_out_h^T
    {
      float dy[NIT2][8], z[NIT2][8], o[NIT2][8], dz[NIT2][8];
#pragma unroll
      for (int it = 0; it < NIT2; ++it) {
        const int k = it * 128 + kq;
        ld8(a.dxo + arow * U + k, dy[it]); ld8(a.zo + arow * U + k, z[it]); ld8(a.xo + arow * U + k, o[it]);
      }
      const float2 st = *reinterpret_cast<const float2*>(a.st3 + arow * 2);
      float dold[T2];
#pragma unroll
      for (int j = 0; j < T2; ++j) dold[j] = olive ? a.dfeat[oidx * F + (nstr + NSTR * j) * 16 + ocol] : 0.f;
      ln_elu_bwd<U, NIT2>(dy, z, o, st.x, st.y, par + O_G3, kq, ws_a, dz);
      store_chunks<NIT2>(dz, a.dzo + arow * U, kq, nstr, alive);
      bf16x8 afr[NIT2][3];
#pragma unroll
      for (int it = 0; it < NIT2; ++it) split8(dz[it], afr[it]);
      float out[T2];
      tiles_gemm<U, T2>(afr, bq2, red, out);
#pragma unroll
      for (int j = 0; j < T2; ++j)
        if (olive) st_wt(a.dfeat + oidx * F + (nstr + NSTR * j) * 16 + ocol, dold[j] + out[j]);
    }
    TSB(3);
    uint4 bq3[ALL3 ? T3 : DEPTH3][ALL3 ? NIT3 : NITC3][3];
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt, [&] {
      if constexpr (ALL3) {
#pragma unroll
        for (int j = 0; j < T3; ++j) load_planes<NIT3>(bq3[j], a.w3, 3 * D, (nstr + NSTR * j) * 16);
      } else {
        load_planes<NITC3>(bq3[0], a.w3, 3 * D, nstr * 16);
      }
    });
    TSB(4);

    // ---------------- Q3: GRU backward;  [dh | dx1] = [(1-u) dhn | 0] + dz3 @ W_gru^T
    {
      float d[ND][8], z0[ND][8], z1_[ND][8], z2[ND][8], hp[ND][8];
      const float* zr = a.z3 + arow * 3 * D;
#pragma unroll
      for (int it = 0; it < ND; ++it) {
        const int k = it * 128 + kq;
        ld8(a.dfeat + arow * F + k, d[it]);
        ld8(zr + k, z0[it]); ld8(zr + D + k, z1_[it]); ld8(zr + 2 * D + k, z2[it]);
        ld8(a.gin + arow * (D + U) + k, hp[it]);
      }
      const float2 st = *reinterpret_cast<const float2*>(a.gst + arow * 2);
      const float mean = st.x, rstd = st.y;
      const float fo = olive ? a.first[oidx] : 1.f;
      float dprev[T3];
#pragma unroll
      for (int j = 0; j < T3; ++j) {
        const int col = (nstr + NSTR * j) * 16 + ocol;
        dprev[j] = (olive && t > 0 && col < D) ? a.dfeat[(oidx - 1) * F + col] : 0.f;
      }
      float xh[NIT3][8], g[NIT3][8];   // index q * ND + it: gate q
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int it = 0; it < ND; ++it) {
        const int k = it * 128 + kq;
        float g0[8], g1_[8], g2[8], b0[8], b1_[8], b2[8], dhd[8], dyr_[8], dyc_[8], dyu_[8];
        ld8(par + O_GG + k, g0); ld8(par + O_GG + D + k, g1_); ld8(par + O_GG + 2 * D + k, g2);
        ld8(par + O_BG + k, b0); ld8(par + O_BG + D + k, b1_); ld8(par + O_BG + 2 * D + k, b2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xr = (z0[it][j] - mean) * rstd, xc = (z1_[it][j] - mean) * rstd,
                      xu = (z2[it][j] - mean) * rstd;
          const float yr = xr * g0[j] + b0[j], yc = xc * g1_[j] + b1_[j], yu = xu * g2[j] + b2[j];
          const float r = fsigmoid_(yr), cand = ftanh_(r * yc), u = fsigmoid_(yu - 1.f);
          const float dd = d[it][j];
          const float du = dd * (cand - hp[it][j]), dc = dd * u;
          dhd[j] = dd * (1.f - u);
          const float dpre = dc * (1.f - cand * cand);
          const float dyc = dpre * r, dyr = dpre * yc * r * (1.f - r), dyu = du * u * (1.f - u);
          dyr_[j] = dyr; dyc_[j] = dyc; dyu_[j] = dyu;
          xh[it][j] = xr; xh[ND + it][j] = xc; xh[2 * ND + it][j] = xu;
          g[it][j] = dyr * g0[j]; g[ND + it][j] = dyc * g1_[j]; g[2 * ND + it][j] = dyu * g2[j];
          s1 += (g[it][j] + g[ND + it][j]) + g[2 * ND + it][j];
          s2 += (g[it][j] * xr + g[ND + it][j] * xc) + g[2 * ND + it][j] * xu;
        }
        // (side outputs leave the registers at once: at deter = 512 the row-wise arrays of this
        // phase are 96 registers each)
        store_chunk1(dyr_, a.dy3 + arow * 3 * D, it, kq, nstr, alive);
        store_chunk1(dyc_, a.dy3 + arow * 3 * D, ND + it, kq, nstr, alive);
        store_chunk1(dyu_, a.dy3 + arow * 3 * D, 2 * ND + it, kq, nstr, alive);
        // the direct path (1 - update) * dhn of the whole row block, for the tile epilogue (the
        // previous step's readers are behind a grid barrier; the contraction's barriers order
        // these writes before this step's reads)
        float* q = &dh_lds[lane & 15][k];
        *reinterpret_cast<float4*>(q) = make_float4(dhd[0], dhd[1], dhd[2], dhd[3]);
        *reinterpret_cast<float4*>(q + 4) = make_float4(dhd[4], dhd[5], dhd[6], dhd[7]);
      }
      TSB(9);
      row_reduce2(s1, s2, ws_b);
      s1 /= (float)(3 * D); s2 /= (float)(3 * D);
      bf16x8 afr[NIT3][3];
#pragma unroll
      for (int i = 0; i < NIT3; ++i) {
        float dz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
        split8(dz, afr[i]);
        store_chunk1(dz, a.dz3 + arow * 3 * D, i, kq, nstr, alive);
      }
      TSB(10);
      float out[T3];
      // (the contraction's barriers order the dh_lds writes before the reads)
      if constexpr (ALL3) {
        tiles_gemm<3 * D, T3>(afr, bq3, red, out);
      } else {
        int n0[T3];
#pragma unroll
        for (int j = 0; j < T3; ++j) n0[j] = (nstr + NSTR * j) * 16;
        tiles_gemm_stream<3 * D, T3, NITC3, DEPTH3>(afr, bq3, a.w3, n0, red, out);
      }
      TSB(11);
#pragma unroll
      for (int j = 0; j < T3; ++j) {
        const int col = (nstr + NSTR * j) * 16 + ocol;
        if (olive) {
          if (col < D) {
            const float dh = dh_lds[orow][col] + out[j];
            st_wt(a.dgin + oidx * (D + U) + col, dh);
            if (t > 0) st_wt(a.dfeat + (oidx - 1) * F + col, dprev[j] + dh * (1.f - fo));
          } else {
            st_wt(a.dgin + oidx * (D + U) + col, out[j]);
          }
        }
      }
    }
    TSB(5);
    uint4 bq4[ALL4 ? T4 : 2][ALL4 ? NIT4 : NITC4][3];
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt, [&] {
      if constexpr (ALL4) {
#pragma unroll
        for (int j = 0; j < T4; ++j)
          load_planes<NIT4>(bq4[j], a.w4, U, (nstr + NSTR * (j / TPG)) * C + (j % TPG) * 16);
      } else {
        load_planes<NITC4>(bq4[0], a.w4, U, nstr * C);
      }
    });
    TSB(6);

    // ---------------- Q4: dz1 = LN-ELU'(dx1);  dxs = dz1 @ W_in_s^T;  mask;  dxq_{t-1}
    {
      float dy[NIT4][8], z[NIT4][8], o[NIT4][8], dz[NIT4][8];
#pragma unroll
      for (int it = 0; it < NIT4; ++it) {
        const int k = it * 128 + kq;
        ld8(a.dgin + arow * (D + U) + D + k, dy[it]); ld8(a.z1 + arow * U + k, z[it]);
        ld8(a.gin + arow * (D + U) + D + k, o[it]);
      }
      const float2 st = *reinterpret_cast<const float2*>(a.st1 + arow * 2);
      // the previous step's stats inputs at this thread's tile elements
      const float fo = olive ? a.first[oidx] : 1.f;
      float xv[T4], dl[T4], dsold[T4];
#pragma unroll
      for (int j = 0; j < T4; ++j) {
        const int col = (nstr + NSTR * (j / TPG)) * C + (j % TPG) * 16 + ocol;
        const bool on = olive && t > 0;
        xv[j] = on ? a.xq[(oidx - 1) * S + col] : 0.f;
        dl[j] = on ? a.dlogit[(oidx - 1) * S + col] : 0.f;
        dsold[j] = on ? a.dfeat[(oidx - 1) * F + D + col] : 0.f;
      }
      ln_elu_bwd<U, NIT4>(dy, z, o, st.x, st.y, par + O_G1, kq, ws_a, dz);
      store_chunks<NIT4>(dz, a.dz1 + arow * U, kq, nstr, alive);
      bf16x8 afr[NIT4][3];
#pragma unroll
      for (int it = 0; it < NIT4; ++it) split8(dz[it], afr[it]);
      float out[T4];
      if constexpr (ALL4) {
        tiles_gemm<U, T4>(afr, bq4, red, out);
      } else {
        int n0[T4];
#pragma unroll
        for (int j = 0; j < T4; ++j) n0[j] = (nstr + NSTR * (j / TPG)) * C + (j % TPG) * 16;
        tiles_gemm_stream<U, T4, NITC4>(afr, bq4, a.w4, n0, red, out);
      }
#pragma unroll
      for (int j = 0; j < T4; ++j) {
        const int col = (nstr + NSTR * (j / TPG)) * C + (j % TPG) * 16 + ocol;
        if (olive) a.dxs[oidx * S + col] = out[j];
      }
      if (t > 0) {
        // stats backward of step t-1 for this workgroup's groups (k_stats_bwd): the softmax of a
        // (row, group) spans the TPG tiles of the group, 16 lanes each
#pragma unroll
        for (int gi = 0; gi < GPP; ++gi) {
          float m = -INFINITY;
#pragma unroll
          for (int q = 0; q < TPG; ++q) m = fmaxf(m, xv[gi * TPG + q]);
          m = tile_row_max(m);
          float e[TPG], sum = 0.f;
#pragma unroll
          for (int q = 0; q < TPG; ++q) { e[q] = dd_exp_det(xv[gi * TPG + q] - m); sum += e[q]; }
          sum = tile_row_sum(sum);
          float pr[TPG], dp[TPG], dot = 0.f, ds[TPG];
#pragma unroll
          for (int q = 0; q < TPG; ++q) {
            const int j = gi * TPG + q;
            pr[q] = e[q] / sum;
            const float pm = dd_unimix_prob(e[q], sum, a.unimix, C);
            ds[q] = dsold[j] + out[j] * (1.f - fo);
            dp[q] = um * (ds[q] + dl[j] / pm);
            dot += dp[q] * pr[q];
          }
          dot = tile_row_sum(dot);
#pragma unroll
          for (int q = 0; q < TPG; ++q) {
            const int j = gi * TPG + q;
            const int col = (nstr + NSTR * gi) * C + q * 16 + ocol;
            if (olive) {
              a.dfeat[(oidx - 1) * F + D + col] = ds[q];
              st_wt(a.dxq + (oidx - 1) * S + col, pr[q] * (dp[q] - dot));
            }
            (void)j;
          }
        }
      }
    }
    TSB(7);
    grid_barrier(a.ctr + RB_CTR(mblk), a.ctr + 1, ++gen * RB_N, bar_wt, load_q1);
    TSB(8);
  }
}
#undef TSB

// class index of every one-hot group: idx[r][g] = argmax_c x[r][g*C + c]
// -1 for an all-zero group (the zero state); a group that is neither one-hot nor zero cannot
// go through the gather form of P1: error word 2.
__global__ void k_onehot_argmax(const float* __restrict__ x, long ldx, int* __restrict__ idx, int rows,
                                int G, int C, unsigned* __restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * G) return;
  const int r = i / G, g = i - r * G;
  const float* p = x + (long)r * ldx + (long)g * C;
  int best = 0, ones = 0, other = 0;
  for (int c = 0; c < C; ++c) {
    best = p[c] > p[best] ? c : best;
    ones += p[c] == 1.f;
    other += p[c] != 1.f && p[c] != 0.f;
  }
  if (other || ones > 1) __hip_atomic_fetch_or(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  idx[i] = ones == 1 ? best : -1;
}

// Weight cache: W [K, N] fp32 (row stride ld) -> three bf16 planes of N x Kp values, exact 3-way
// split, zero beyond K, in the fragment-major order of plane_index (N % 16 == 0, Kp % 128 == 0).
__global__ void k_scan_wprep(const float* __restrict__ W, long ld, int K, int N, int Kp,
                             unsigned short* __restrict__ out) {
  const long total = (long)N * Kp;
  // (consecutive threads: consecutive n of one k - W's rows are read coalesced)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / N), n = (int)(i - (long)k * N);
    const float x = k < K ? W[(long)k * ld + n] : 0.f;
    unsigned h, m, l;
    split3(x, h, m, l);
    out[plane_index(n, k, 0, Kp)] = (unsigned short)(h >> 16);
    out[plane_index(n, k, 1, Kp)] = (unsigned short)(m >> 16);
    out[plane_index(n, k, 2, Kp)] = (unsigned short)(l >> 16);
  }
}

// Weight cache of the reverse scan: W [N, K] fp32 (row stride ld; the backward contraction
// multiplies by W^T, so the cache's columns are W's rows) -> three bf16 planes, same order.
__global__ void k_scan_wprep_rows(const float* __restrict__ W, long ld, int N, int K,
                                  unsigned short* __restrict__ out) {
  const long total = (long)N * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    unsigned h, m, l;
    split3(W[(long)n * ld + k], h, m, l);
    out[plane_index(n, k, 0, K)] = (unsigned short)(h >> 16);
    out[plane_index(n, k, 1, K)] = (unsigned short)(m >> 16);
    out[plane_index(n, k, 2, K)] = (unsigned short)(l >> 16);
  }
}

}  // namespace

extern "C" int dd_scan_wprep_rows(const float* W, long ld, int N, int K, void* planes, void* stream) {
  DD_REQUIRE(N > 0 && K > 0 && K % 128 == 0 && N % 16 == 0, "dd_scan_wprep_rows: N multiple of 16, K multiple of 128");
  const long total = (long)N * K;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  k_scan_wprep_rows<<<blocks, 256, 0, (hipStream_t)stream>>>(W, ld, N, K, (unsigned short*)planes);
  DD_CHECK_LAUNCH("dd_scan_wprep_rows");
  return 0;
}

// Barrier counters back to zero before a scan launch: word 0 (all workgroups) and the four
// row-block counters at 576 + 128 * block.  A KERNEL, not hipMemsetAsync: inside a captured graph
// the memset nodes were the one thing in front of the scan that is not a kernel launch, and
// (docs/LABLOG.md, end of round 6) a replayed scan was seen to run its row-block barriers on
// counters that had not been cleared - barriers that let every workgroup through.
__global__ void k_scan_reset(unsigned* sync2) {
  if (threadIdx.x == 0) sync2[0] = 0;
  sync2[576 + threadIdx.x] = 0;
}
// DD_SCAN_FLAGS: A/B of the barrier protocol (bits 7, 8) and of the workgroup map (bit 9)
static int scan_dbg_flags() {
  static const int f = getenv("DD_SCAN_FLAGS") ? atoi(getenv("DD_SCAN_FLAGS")) & 896 : 0;
  return f;
}
static int reset_counters(unsigned* sync2, hipStream_t st, const char* what) {
  static const int use_memset = getenv("DD_SCAN_MEMSET") ? atoi(getenv("DD_SCAN_MEMSET")) : 0;   // (A/B: the old form)
  if (use_memset) {
    hipError_t e = hipMemsetAsync(sync2, 0, sizeof(unsigned), st);
    if (e == hipSuccess) e = hipMemsetAsync(sync2 + 576, 0, 512 * sizeof(unsigned), st);
    if (e != hipSuccess) { dd_set_error(what, e); return (int)e; }
    return 0;
  }
  k_scan_reset<<<1, 512, 0, st>>>(sync2);
  DD_CHECK_LAUNCH(what);
  return 0;
}

// The grid barrier needs all NWG workgroups resident at once (one per CU): on a device (or a
// compute partition, e.g. CPX mode: 32 CUs) with fewer CUs the persistent kernels would spin
// into their timeout, so such a device reports "unsupported" and the caller keeps the per-layer
// launch sequence.  Without a visible device (build / CPU tests) the shape alone decides.
static bool scan_device_ok() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return true; }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return true; }
  return cus >= NWG;
}

extern "C" int dd_observe_scan_bwd_supported(int B, int D, int U, int G, int C) {
  return B >= 1 && B <= 64 && ((D == 256 && U == 256) || (D == 512 && U == 512)) && G == 32 && C == 32 && scan_device_ok();
}

extern "C" int dd_observe_scan_bwd(
    int B, int T, int D, int U, int G, int C, int flags, float unimix, const float* first,
    const float* xq, const float* zo, const float* xo, const float* st3, const float* z3,
    const float* gst, const float* gin, const float* z1, const float* st1, const float* dlogit,
    const void* w1, const void* w2, const void* w3, const void* w4,
    const float* g3, const float* gg, const float* bg, const float* g1,
    float* dfeat, float* dxq, float* dxo, float* dzo, float* dz3, float* dy3, float* dgin,
    float* dz1, float* dxs, unsigned* sync2, void* stream) {
  DD_REQUIRE(dd_observe_scan_bwd_supported(B, D, U, G, C), "dd_observe_scan_bwd: unsupported shape");
  hipStream_t st = (hipStream_t)stream;
  ScanBwdArgs a;
  a.B = B; a.T = T; a.flags = flags; a.unimix = unimix; a.first = first;
  a.xq = xq; a.zo = zo; a.xo = xo; a.st3 = st3; a.z3 = z3; a.gst = gst; a.gin = gin; a.z1 = z1; a.st1 = st1;
  a.dlogit = dlogit;
  a.w1 = (const unsigned short*)w1; a.w2 = (const unsigned short*)w2;
  a.w3 = (const unsigned short*)w3; a.w4 = (const unsigned short*)w4;
  a.g3 = g3; a.gg = gg; a.bg = bg; a.g1 = g1;
  a.dfeat = dfeat; a.dxq = dxq; a.dxo = dxo; a.dzo = dzo; a.dz3 = dz3; a.dy3 = dy3; a.dgin = dgin;
  a.dz1 = dz1; a.dxs = dxs; a.ctr = sync2;
  a.flags |= scan_dbg_flags();
  if (int rc = reset_counters(sync2, st, "dd_observe_scan_bwd(counters)")) return rc;
  if (D == 256 && U == 256 && G == 32 && C == 32)
    k_observe_scan_bwd<256, 256, 32, 32><<<NWG, 256, 0, st>>>(a);
  else if (D == 512 && U == 512 && G == 32 && C == 32)
    k_observe_scan_bwd<512, 512, 32, 32><<<NWG, 256, 0, st>>>(a);
  else
    DD_REQUIRE(false, "dd_observe_scan_bwd: shape not compiled");
  DD_CHECK_LAUNCH("dd_observe_scan_bwd");
  return 0;
}

extern "C" int dd_scan_wprep(const float* W, long ld, int K, int N, int Kp, void* planes, void* stream) {
  DD_REQUIRE(Kp >= K && Kp % 128 == 0 && N % 16 == 0, "dd_scan_wprep: Kp multiple of 128 >= K, N multiple of 16");
  const long total = (long)N * Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  k_scan_wprep<<<blocks, 256, 0, (hipStream_t)stream>>>(W, ld, K, N, Kp, (unsigned short*)planes);
  DD_CHECK_LAUNCH("dd_scan_wprep");
  return 0;
}

// compiled shapes (deter, units, groups, classes, action dims): loops unroll and the loads of a
// phase issue together only with compile-time extents
#define DD_SCAN_SHAPES(X) X(256, 256, 32, 32, 16) X(256, 256, 32, 32, 6) X(512, 512, 32, 32, 6)

extern "C" int dd_observe_scan_supported(int B, int D, int U, int G, int C, int A) {
  if (B < 1 || B > 64 || !scan_device_ok()) return 0;
#define X(d, u, g, c, a_) if (D == d && U == u && G == g && C == c && A == a_) return 1;
  DD_SCAN_SHAPES(X)
#undef X
  return 0;
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
  a.use_carry = ((use_carry & 1) && carry != nullptr ? 1 : 0) | (use_carry & 1022); a.unimix = unimix;
  a.use_carry |= scan_dbg_flags();
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
  a.nwg = NWG;
  if (int rc = reset_counters(sync2, st, "dd_observe_scan_fwd(counters)")) return rc;
  if (a.use_carry & 1) {
    k_onehot_argmax<<<(B * G + 255) / 256, 256, 0, st>>>(carry + D, D + a.S, idx_ws + N * G, B, G, C, sync2 + 1);
    DD_CHECK_LAUNCH("dd_observe_scan_fwd(argmax carry)");
  }
  k_onehot_argmax<<<(G + 255) / 256, 256, 0, st>>>(init_stoch, a.S, idx_ws + (N + B) * G, 1, G, C, sync2 + 1);
  DD_CHECK_LAUNCH("dd_observe_scan_fwd(argmax init)");
  bool launched = false;
#define X(d, u, g, c, a_) if (!launched && D == d && U == u && G == g && C == c && A == a_) { k_observe_scan_fwd<d, u, g, c, a_><<<a.nwg, 256, 0, st>>>(a); launched = true; }
  DD_SCAN_SHAPES(X)
#undef X
  DD_CHECK_LAUNCH("dd_observe_scan_fwd");
  return 0;
}
