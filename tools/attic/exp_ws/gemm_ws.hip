// EXPERIMENT (not part of libdaydreamer_hip.so; built by tools/exp_ws.sh, measured by
// tools/exp_ws.py): the split-bf16 contraction loop of gemm_core.h with SPECIALISED waves.
//
// The product loop (k_mfma_gemm_s3, 256 threads) makes every wave do everything: global loads,
// the exact 3-way bf16 split (~25 VALU instructions per staged float4), LDS stores, fragment
// reads and MFMAs; rocprofv3 counters put its matrix pipe at 0.63-0.69 busy with 0.41-0.44 of
// the wave cycles stalled on issue (profiles/r01_pmc_mfma_lds.txt, r02_gemm_ablation_128tile.txt:
// "the MFMA half and the staging half of a workgroup barely overlap").  Here a workgroup has
// 512 threads = 8 waves, two per SIMD: waves 0-3 only read fragments and issue MFMAs (2 x 2 wave
// tiles of 64 x 64, as in the product loop), waves 4-7 only load, split and store the next
// k-tile.  One barrier per k-tile; LDS image and arithmetic (six products, smallest terms
// first, fp32 accumulation) are those of the product loop, so results are bit-identical to it.
// One workgroup per CU (50.7 KB of LDS, <= 256 registers per wave).
//
// Entry point: dd_gemm_ws (the plain GEMM forms NN / NT / TN on 16-byte aligned operands whose
// float4 axes are multiples of four), same split-K policy and reduce pass as dd_gemm_f32.
#include "gemm_core.h"

int g_gemm_mode = 6;

namespace {

// (ablation) stage a float4 without the split arithmetic: the truncated high halves go to all three
// planes - what a loader of PRE-SPLIT planes would cost the staging waves, timing only
template <class PL>
__device__ __forceinline__ void store_nosplit(unsigned char* base, const float (&v)[4], int r, int k, bool kc) {
  const int o = kc ? (k >> 3) * PL::STR + r * 16 + (k & 4) * 2 : k * PL::STR + r * 2;
  const uint2 w = make_uint2(pack_hi(__float_as_uint(v[0]), __float_as_uint(v[1])),
                             pack_hi(__float_as_uint(v[2]), __float_as_uint(v[3])));
  *reinterpret_cast<uint2*>(base + o) = w;
  *reinterpret_cast<uint2*>(base + PL::BYTES + o) = w;
  *reinterpret_cast<uint2*>(base + 2 * PL::BYTES + o) = w;
}

// ABL (timing only, wrong results): bit 0 = the staging waves do nothing inside the loop, bit 1 = the compute
// waves read their fragments once and reuse them, bit 2 = no MFMAs (fragment reads kept alive),
// bit 3 = the B operand staged without the split, bit 4 = the A operand staged without the split
template <int BM, int BN, bool AKC, bool BKC, class AL, class BL, class EP, int OCC = 1, int BK = 16, int ST = 2, int ABL = 0>
__global__ void __launch_bounds__(512, 2 * OCC)   // (HIP: second argument = waves per SIMD)
k_gemm_ws(AL al, BL bl, EP ep, int K, int kps, int tiles_m) {
  constexpr int NPL = 3;
  using LA = PlaneS3<BM, AKC, BK>;
  using LB = PlaneS3<BN, BKC, BK>;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][NPL * LA::BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][NPL * LB::BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stager = wave >= 4;
  const int rt = tid & 255;          // thread index inside its role
  const int cw = wave & 3;           // compute wave index (2 x 2)
  int tmi, tni;
  if (!tile_coords(tiles_m, tmi, tni)) return;
  const int m0 = tmi * BM, n0 = tni * BN;
  const int kb = blockIdx.z * kps;
  const int ke = min(K, kb + kps);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  const int wm0 = (cw >> 1) * WM, wn0 = (cw & 1) * WN;
  constexpr int NA = LA::N, NB = LB::N;
  const int nk = (ke - kb + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float ra_[ST][NA][4], rb_[ST][NB][4];

  auto gload = [&](int t, float (&ra)[NA][4], float (&rb)[NB][4]) {
    const int k0 = kb + t * BK;
    if (k0 + BK <= ke) {
#pragma unroll
      for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<true>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
      for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<true>(n0 + r, k0 + k, ke, rb[u]); }
    } else {
#pragma unroll
      for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<false>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
      for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<false>(n0 + r, k0 + k, ke, rb[u]); }
    }
  };
  auto sstore = [&](int buf, float (&ra)[NA][4], float (&rb)[NB][4]) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      int r, k; LA::coord(rt, u, r, k);
      if constexpr (ABL & 16) store_nosplit<LA>(As[buf], ra[u], r, k, AKC);
      else LA::template store<NPL>(As[buf], ra[u], r, k);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      int r, k; LB::coord(rt, u, r, k);
      if constexpr (ABL & 8) store_nosplit<LB>(Bs[buf], rb[u], r, k, BKC);
      else LB::template store<NPL>(Bs[buf], rb[u], r, k);
    }
  };
  bf16x8 af0[TM][NPL], bf0[TN][NPL];
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[TM][NPL], bf[TN][NPL];
      if constexpr (ABL & 2) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int p = 0; p < NPL; ++p) af[a][p] = af0[a][p];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int p = 0; p < NPL; ++p) bf[b][p] = bf0[b][p];
      } else {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int p = 0; p < NPL; ++p) af[a][p] = LA::frag(As[buf] + p * LA::BYTES, wm0 + a * 32, ks, lane);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int p = 0; p < NPL; ++p) bf[b][p] = LB::frag(Bs[buf] + p * LB::BYTES, wn0 + b * 32, ks, lane);
      }
      constexpr int PA_[6] = {NPL - 1, 0, 1, 1, 0, 0}, PB_[6] = {0, NPL - 1, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            if constexpr (ABL & 4) acc[a][b][q] += (float)af[a][PA_[q]][0] + (float)bf[b][PB_[q]][0];
            else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA_[q]], bf[b][PB_[q]], acc[a][b], 0, 0, 0);
          }
    }
  };

  // prologue (stagers): tiles 0 .. ST-1 into registers, tile 0 into LDS buffer 0, then the
  // freed slot 0 takes tile ST
  if (stager) {
#pragma unroll
    for (int s_ = 0; s_ < ST; ++s_)
      if (s_ < nk) gload(s_, ra_[s_], rb_[s_]);
    if (nk > 0) sstore(0, ra_[0], rb_[0]);
    if (ST < nk) gload(ST, ra_[0], rb_[0]);
  }
  __syncthreads();
  if constexpr (ABL & 2) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int p = 0; p < NPL; ++p) af0[a][p] = LA::frag(As[0] + p * LA::BYTES, wm0 + a * 32, 0, lane);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int p = 0; p < NPL; ++p) bf0[b][p] = LB::frag(Bs[0] + p * LB::BYTES, wn0 + b * 32, 0, lane);
  }
  // iteration t: compute waves multiply buffer t & 1; stagers put tile t + 1 (register slot
  // (t + 1) % ST) into buffer (t + 1) & 1 - last read in iteration t - 1, before the barrier
  // that ended it - and then request tile t + 1 + ST into the slot just emptied
  for (int t0 = 0; t0 < nk; t0 += ST) {
#pragma unroll
    for (int s_ = 0; s_ < ST; ++s_) {
      const int t = t0 + s_;
      if (t < nk) {
        if (stager) {
          if constexpr (!(ABL & 1)) {
            if (t + 1 < nk) sstore((t + 1) & 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
            if (t + 1 + ST < nk) gload(t + 1 + ST, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
          }
        } else {
          compute(t & 1);
        }
        __syncthreads();
      }
    }
  }
  if (stager) return;

  const int lk = lane >> 5, lr = lane & 31;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn0 + b * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ep(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
    }
}

// Role-separated form: each role runs its OWN loop (same number of barriers), so the register
// allocator sees either the accumulators + fragments or the staging registers, never both - the
// kernel fits 128 registers and two workgroups (two MFMA waves + two staging waves per SIMD) share a CU.
// PRIO: 1 = the MFMA waves run at raised priority, 2 = the staging waves do
template <int BM, int BN, bool AKC, bool BKC, class AL, class BL, class EP, int ST = 2, int PRIO = 0>
__global__ void __launch_bounds__(512, 4)
k_gemm_ws2(AL al, BL bl, EP ep, int K, int kps, int tiles_m) {
  constexpr int BK = 16, NPL = 3;
  using LA = PlaneS3<BM, AKC, BK>;
  using LB = PlaneS3<BN, BKC, BK>;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][NPL * LA::BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][NPL * LB::BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tmi, tni;
  if (!tile_coords(tiles_m, tmi, tni)) return;
  const int m0 = tmi * BM, n0 = tni * BN;
  const int kb = blockIdx.z * kps;
  const int ke = min(K, kb + kps);
  const int nk = (ke - kb + BK - 1) / BK;
  if (wave >= 4) {   // ---- staging waves
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(2);
    const int rt = tid & 255;
    constexpr int NA = LA::N, NB = LB::N;
    float ra_[ST][NA][4], rb_[ST][NB][4];
    auto gload = [&](int t, float (&ra)[NA][4], float (&rb)[NB][4]) {
      const int k0 = kb + t * BK;
      if (k0 + BK <= ke) {
#pragma unroll
        for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<true>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
        for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<true>(n0 + r, k0 + k, ke, rb[u]); }
      } else {
#pragma unroll
        for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<false>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
        for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<false>(n0 + r, k0 + k, ke, rb[u]); }
      }
    };
    auto sstore = [&](int buf, float (&ra)[NA][4], float (&rb)[NB][4]) {
#pragma unroll
      for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); LA::template store<NPL>(As[buf], ra[u], r, k); }
#pragma unroll
      for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); LB::template store<NPL>(Bs[buf], rb[u], r, k); }
    };
#pragma unroll
    for (int s_ = 0; s_ < ST; ++s_)
      if (s_ < nk) gload(s_, ra_[s_], rb_[s_]);
    if (nk > 0) sstore(0, ra_[0], rb_[0]);
    if (ST < nk) gload(ST, ra_[0], rb_[0]);
    __syncthreads();
    for (int t0 = 0; t0 < nk; t0 += ST) {
#pragma unroll
      for (int s_ = 0; s_ < ST; ++s_) {
        const int t = t0 + s_;
        if (t < nk) {
          if (t + 1 < nk) sstore((t + 1) & 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
          if (t + 1 + ST < nk) gload(t + 1 + ST, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
          __syncthreads();
        }
      }
    }
    return;
  }
  // ---- MFMA waves
  if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(2);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int p = 0; p < NPL; ++p) af[a][p] = LA::frag(As[buf] + p * LA::BYTES, wm0 + a * 32, 0, lane);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int p = 0; p < NPL; ++p) bf[b][p] = LB::frag(Bs[buf] + p * LB::BYTES, wn0 + b * 32, 0, lane);
    constexpr int PA_[6] = {NPL - 1, 0, 1, 1, 0, 0}, PB_[6] = {0, NPL - 1, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA_[q]], bf[b][PB_[q]], acc[a][b], 0, 0, 0);
    __syncthreads();
  }
  const int lk = lane >> 5, lr = lane & 31;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn0 + b * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ep(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
    }
}

// Persistent form: one workgroup per CU walks a contiguous range of tiles (column tiles of a row
// tile back to back).  After the last k-tile of a tile the compute waves write its results while
// the staging waves are already loading and staging the first k-tiles of the next one.
template <int BM, int BN, bool AKC, bool BKC, class AL, class BL, class EP, int BK = 16, int ST = 2>
__global__ void __launch_bounds__(512, 2)
k_gemm_ws_p(AL al, BL bl, EP ep, int K, int tiles_m, int tiles_n, int per) {
  constexpr int NPL = 3;
  using LA = PlaneS3<BM, AKC, BK>;
  using LB = PlaneS3<BN, BKC, BK>;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][NPL * LA::BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][NPL * LB::BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stager = wave >= 4;
  const int rt = tid & 255, cw = wave & 3;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  const int wm0 = (cw >> 1) * WM, wn0 = (cw & 1) * WN;
  constexpr int NA = LA::N, NB = LB::N;
  const int ke = K, nk = (K + BK - 1) / BK;
  const int ntiles = tiles_m * tiles_n;
  const int tbeg = (int)blockIdx.x * per, tend = min(ntiles, tbeg + per);
  float ra_[ST][NA][4], rb_[ST][NB][4];
  const int lk = lane >> 5, lr = lane & 31;

  for (int tile = tbeg; tile < tend; ++tile) {
    const int tmi = tile / tiles_n, tni = tile - tmi * tiles_n;
    const int m0 = tmi * BM, n0 = tni * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto gload = [&](int t, float (&ra)[NA][4], float (&rb)[NB][4]) {
      const int k0 = t * BK;
      if (k0 + BK <= ke) {
#pragma unroll
        for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<true>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
        for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<true>(n0 + r, k0 + k, ke, rb[u]); }
      } else {
#pragma unroll
        for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); al.template load4<false>(m0 + r, k0 + k, ke, ra[u]); }
#pragma unroll
        for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); bl.template load4<false>(n0 + r, k0 + k, ke, rb[u]); }
      }
    };
    auto sstore = [&](int buf, float (&ra)[NA][4], float (&rb)[NB][4]) {
#pragma unroll
      for (int u = 0; u < NA; ++u) { int r, k; LA::coord(rt, u, r, k); LA::template store<NPL>(As[buf], ra[u], r, k); }
#pragma unroll
      for (int u = 0; u < NB; ++u) { int r, k; LB::coord(rt, u, r, k); LB::template store<NPL>(Bs[buf], rb[u], r, k); }
    };
    auto compute = [&](int buf) {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int p = 0; p < NPL; ++p) af[a][p] = LA::frag(As[buf] + p * LA::BYTES, wm0 + a * 32, ks, lane);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int p = 0; p < NPL; ++p) bf[b][p] = LB::frag(Bs[buf] + p * LB::BYTES, wn0 + b * 32, ks, lane);
        constexpr int PA_[6] = {NPL - 1, 0, 1, 1, 0, 0}, PB_[6] = {0, NPL - 1, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA_[q]], bf[b][PB_[q]], acc[a][b], 0, 0, 0);
      }
    };

    if (stager) {
#pragma unroll
      for (int s_ = 0; s_ < ST; ++s_)
        if (s_ < nk) gload(s_, ra_[s_], rb_[s_]);
      if (nk > 0) sstore(0, ra_[0], rb_[0]);
      if (ST < nk) gload(ST, ra_[0], rb_[0]);
    }
    __syncthreads();
    for (int t0 = 0; t0 < nk; t0 += ST) {
#pragma unroll
      for (int s_ = 0; s_ < ST; ++s_) {
        const int t = t0 + s_;
        if (t < nk) {
          if (stager) {
            if (t + 1 < nk) sstore((t + 1) & 1, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
            if (t + 1 + ST < nk) gload(t + 1 + ST, ra_[(s_ + 1) % ST], rb_[(s_ + 1) % ST]);
          } else {
            compute(t & 1);
          }
          __syncthreads();
        }
      }
    }
    if (!stager) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = n0 + wn0 + b * 32 + lr;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ep(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col, acc[a][b][r]);
        }
    }
  }
}

template <bool AKC, bool BKC, class AL, class BL>
int run_ws(AL al, BL bl, int M, int N, int K, float* C, long ldc, float* ws, size_t ws_bytes, hipStream_t st) {
  const int tm = dd_ceil_div(M, 128), tn = dd_ceil_div(N, 128);
  const long MN = (long)M * N;
  // one workgroup per CU: 256 slots (DD_WS_TARGET), same one-round rule as pick_split
  static const int target = getenv("DD_WS_TARGET") ? atoi(getenv("DD_WS_TARGET")) : 256;
  long tiles = (long)tm * tn;
  long s = 1;
  if (tiles < target * 3 / 4 && K >= 128) {
    s = target / tiles;
    if (s > K / 64) s = K / 64;
    while (s > 1 && (size_t)s * MN * sizeof(float) > ws_bytes) --s;
    if (s < 1) s = 1;
  }
  int S = (int)s;
  int kps = ((dd_ceil_div(K, S) + 31) / 32) * 32;
  S = dd_ceil_div(K, kps);
  static const int persist = getenv("DD_WS_PERSIST") ? atoi(getenv("DD_WS_PERSIST")) : 0;
  if (persist && S == 1) {
    static const int cus = getenv("DD_WS_CUS") ? atoi(getenv("DD_WS_CUS")) : 256;
    const int nt = tm * tn, per = dd_ceil_div(nt, cus), G = dd_ceil_div(nt, per);
    EpiMat ep{C, ldc, nullptr, 1.f, 0.f, M, N, nullptr};
    static const int bkp = getenv("DD_WS_BK") ? atoi(getenv("DD_WS_BK")) : 16;
    static const int stp = getenv("DD_WS_ST") ? atoi(getenv("DD_WS_ST")) : 2;   // k-tiles in flight in registers
    if (bkp == 32)
      k_gemm_ws_p<128, 128, AKC, BKC, AL, BL, EpiMat, 32><<<G, 512, 0, st>>>(al, bl, ep, K, tm, tn, per);
    else if (stp == 4)
      k_gemm_ws_p<128, 128, AKC, BKC, AL, BL, EpiMat, 16, 4><<<G, 512, 0, st>>>(al, bl, ep, K, tm, tn, per);
    else
      k_gemm_ws_p<128, 128, AKC, BKC, AL, BL, EpiMat, 16><<<G, 512, 0, st>>>(al, bl, ep, K, tm, tn, per);
    DD_CHECK_LAUNCH("dd_gemm_ws(persistent)");
    return 0;
  }
  EpiMat ep{C, ldc, nullptr, 1.f, 0.f, M, N, S > 1 ? ws : nullptr};
  dim3 grid(tm * tn, 1, S);
  int tma = tm;
  if (xcd_swizzle() == 0) tma = -tm;
  // DD_WS_BK=32: two k-steps per barrier (101 KB of LDS); DD_WS_OCC=2: held to 128 registers per
  // wave for two workgroups per CU (spills ~420 registers: kept only to be measured)
  static const int occ = getenv("DD_WS_OCC") ? atoi(getenv("DD_WS_OCC")) : 1;
  static const int bk = getenv("DD_WS_BK") ? atoi(getenv("DD_WS_BK")) : 16;
  if (getenv("DD_WS_ROLES") && atoi(getenv("DD_WS_ROLES")) == 1)   // role-separated loops, two workgroups per CU
  {
    const int v = getenv("DD_WS_V2") ? atoi(getenv("DD_WS_V2")) : 0;   // 1: MFMA waves prioritised, 2: staging waves, 3: four k-tiles in flight
    if (v == 1) k_gemm_ws2<128, 128, AKC, BKC, AL, BL, EpiMat, 2, 1><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else if (v == 2) k_gemm_ws2<128, 128, AKC, BKC, AL, BL, EpiMat, 2, 2><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else if (v == 3) k_gemm_ws2<128, 128, AKC, BKC, AL, BL, EpiMat, 4, 0><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else k_gemm_ws2<128, 128, AKC, BKC, AL, BL, EpiMat><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
  }
  else if (occ == 2)
    k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 2, 16><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
  else if (getenv("DD_WS_ABL")) {   // ablations of the BK = 32 loop (timing only)
    const int abl = atoi(getenv("DD_WS_ABL"));
    if (abl == 1) k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 2, 1><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else if (abl == 2) k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 2, 2><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else if (abl == 3) k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 2, 3><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else if (abl == 8) k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 2, 8><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else if (abl == 24) k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 2, 24><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else if (abl == 5) k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 2, 5><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
    else k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 2, 4><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
  }
  else if (bk == 32 && getenv("DD_WS_ST") && atoi(getenv("DD_WS_ST")) == 4)
    k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32, 4><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
  else if (bk == 32)
    k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 32><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
  else if (getenv("DD_WS_ST") && atoi(getenv("DD_WS_ST")) == 4)
    k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 16, 4><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
  else
    k_gemm_ws<128, 128, AKC, BKC, AL, BL, EpiMat, 1, 16><<<grid, 512, 0, st>>>(al, bl, ep, K, kps, tma);
  DD_CHECK_LAUNCH("dd_gemm_ws");
  if (S > 1) {
    launch_splitk_reduce(ws, S, MN, N, C, ldc, nullptr, 1.f, 0.f, st);
    DD_CHECK_LAUNCH("dd_gemm_ws(split-k reduce)");
  }
  return 0;
}

}  // namespace

extern "C" int dd_gemm_ws(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb,
                          long ldc, int transA, int transB, float* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DD_REQUIRE(aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0 && M % 4 == 0 && N % 4 == 0 &&
             K % 4 == 0 && M >= 4 && N >= 4 && K >= 4, "dd_gemm_ws: aligned operands, sizes multiples of four");
  if (!transA && !transB)
    return run_ws<true, false>(MatKC<true>{A, lda, M, 1}, MatRC<true>{B, ldb, N, 1}, M, N, K, C, ldc, ws, ws_bytes, st);
  if (!transA && transB)
    return run_ws<true, true>(MatKC<true>{A, lda, M, 1}, MatKC<true>{B, ldb, N, 1}, M, N, K, C, ldc, ws, ws_bytes, st);
  if (transA && !transB)
    return run_ws<false, false>(MatRC<true>{A, lda, M, 1}, MatRC<true>{B, ldb, N, 1}, M, N, K, C, ldc, ws, ws_bytes, st);
  DD_REQUIRE(false, "dd_gemm_ws: TT not instantiated");
}

// the product loop on the same operands, for the comparison in the same process
extern "C" int dd_gemm_ref(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb,
                           long ldc, int transA, int transB, float* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!transA && !transB)
    return run_mat<true, false>(MatKC<true>{A, lda, M, 1}, MatRC<true>{B, ldb, N, 1}, M, N, K, C, ldc, nullptr, 1.f, 0.f, ws, ws_bytes, st, "dd_gemm_ref");
  if (!transA && transB)
    return run_mat<true, true>(MatKC<true>{A, lda, M, 1}, MatKC<true>{B, ldb, N, 1}, M, N, K, C, ldc, nullptr, 1.f, 0.f, ws, ws_bytes, st, "dd_gemm_ref");
  if (transA && !transB)
    return run_mat<false, false>(MatRC<true>{A, lda, M, 1}, MatRC<true>{B, ldb, N, 1}, M, N, K, C, ldc, nullptr, 1.f, 0.f, ws, ws_bytes, st, "dd_gemm_ref");
  DD_REQUIRE(false, "dd_gemm_ref: TT not instantiated");
}
