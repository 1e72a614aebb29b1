// Image-side layers of the decoder / encoder: convolutions whose "big" tensor has a handful of
// channels (the RGB / depth image).  Reference: ImageDecoderSimple's last Conv2D (transposed,
// k = 6, stride 2, VALID; nets.py:308-327, Conv2D nets.py:495-554).
//
// Why a dedicated kernel: as a GEMM the image side is the N (or K) dimension and is 3-6 wide, an
// MFMA tile would be >90 % padding, so the generic path was GEMM (N = k*k*Cb) + col2im: 0.96 +
// 0.51 ms at configs[1] for 699 MB of algorithmic traffic (0.73 TB/s).  Here the FOUR output
// parities are the column dimension: an output block (2 x 2 pixels x Cb channels) is one row of
//   out[n, 2i+py, 2j+px, cb] = bias[cb] + sum_{a, b < k/2} sum_ci in[n, i-a, j-b, ci] * W[2a+py, 2b+px, cb, ci]
// i.e. a contraction with M = n * ceil(hb/2) * ceil(wb/2) rows, N = 4 * Cb <= 16 columns (75 % of a
// 16-wide MFMA tile at Cb = 3) and K = (k/2)^2 * Cs.  A workgroup owns 4 block rows x 32 block
// columns of one image; per 32-channel chunk of Cs it stages the input patch (with its
// (k/2 - 1)-pixel halo) ONCE as three bf16 planes in LDS (exact 3-way split, as every
// contraction of this library), every tap's A fragment is a 16-byte LDS read per lane and plane
// (pixel stride 80 B: the 16 pixels of a fragment fall on distinct banks), the tap weights come
// from a pre-split fragment-major plane cache.  v_mfma_f32_16x16x32_bf16, six products.
#include "dd_common.h"
#include <stdlib.h>
#include "../../include/daydreamer_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

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

constexpr int BI = 4, BJ = 32;      // output blocks (2 x 2 pixels) per workgroup: rows x columns
constexpr int PSTRIDE = 80;         // bytes per staged pixel and plane (32 bf16 + pad)

// Weight cache of the parity form: B[(chunk, a, b, ci32)][(py, px, cb)] = W[2a+py, 2b+px, cb, chunk*32+ci32]
// as fragment-major bf16 planes [NT tiles][K/32 k-steps][3][64 lanes][8]; columns >= 4 * Cb are zero.
__global__ void k_convT_image_wprep(const float* __restrict__ w, int k, int Cb, int Cs, int NT,
                                    char* __restrict__ planes) {
  const int T = k / 2, KS = T * T * (Cs / 32);
  const long total = (long)NT * KS * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long tk = i >> 6;
    const int ks = (int)(tk % KS), tile = (int)(tk / KS);
    const int n = tile * 16 + (lane & 15);
    const int chunk = ks / (T * T), ab = ks % (T * T), a = ab / T, b = ab % T;
    float v[8];
    if (n < 4 * Cb) {
      const int par = n / Cb, cb = n - par * Cb, py = par >> 1, px = par & 1;
      const float* src = w + ((long)((2 * a + py) * k + (2 * b + px)) * Cb + cb) * Cs + chunk * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[j];
    } else {
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

// Persistent, two wave groups: a workgroup is 8 waves on one CU; waves 0-3 and waves 4-7 each own
// half of the workgroup's contiguous tile range, a patch buffer of their own, and alternate
// between STAGING a (tile, channel chunk) unit (global -> registers was requested PD units ahead;
// split once, three planes into LDS) and MULTIPLYING it, half a period apart: while one group
// stages (VALU, LDS writes, memory requests) the other runs its MFMAs.  The tap weights of all
// chunks stay in LDS for the lifetime of the workgroup and are shared by both groups
// (re-fetching a chunk's 27 KB per unit has every workgroup of the chip hammer the same few L2
// lines: 575 vs 293 us for the staging alone).
template <int T, int NT, int NCH>
__global__ void __launch_bounds__(512, 1)
k_convT_image(const float* __restrict__ small, const char* __restrict__ planes,
              const float* __restrict__ bias, float* __restrict__ big, int hs, int ws_, int Cs,
              int hb, int wb, int Cb, int tiles_j, int tiles_i, int n_tiles, int dbg) {
  constexpr int PI = BI + T - 1, PJ = BJ + T - 1, NPIX = PI * PJ;
  constexpr int PLANE = NPIX * PSTRIDE;              // bytes per staged plane
  constexpr int TT = T * T, PD = 2, KS = TT * NCH;
  __shared__ __attribute__((aligned(16))) char patches[2][3 * PLANE];
  __shared__ __attribute__((aligned(16))) char bl[NCH * TT * NT * 3 * 1024];
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* patch = patches[grp];
  // the tap weights: [chunk][tap][tile][plane][lane] 16-byte fragments
  for (int it = threadIdx.x; it < NCH * TT * NT * 3 * 64; it += 512) {
    const int l = it & 63, r = it >> 6;            // r = ((chunk * TT + tap) * NT + tile) * 3 + plane
    const int p = r % 3, tt = r / 3, tl = tt % NT, ks = tt / NT;
    *reinterpret_cast<uint4*>(bl + (long)r * 1024 + l * 16) =
        *reinterpret_cast<const uint4*>(planes + (((long)tl * KS + ks) * 3 + p) * 1024 + l * 16);
  }
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int wg_lo = blockIdx.x * per, wg_hi = min(n_tiles, wg_lo + per);
  const int half = (max(0, wg_hi - wg_lo) + 1) / 2;
  const int tile_lo = grp == 0 ? wg_lo : wg_lo + half;
  const int tile_hi = grp == 0 ? min(wg_hi, wg_lo + half) : wg_hi;
  const int n_units = max(0, tile_hi - tile_lo) * NCH;
  const int max_units = half * NCH;                  // group 0's count: the longer of the two

  // this thread's patch items (32 bytes each): pixel (pi, pj) of the patch, 8-channel group q
  constexpr int NPI = (NPIX * 4 + 255) / 256;
  int ioff[NPI], ipi[NPI], ipj[NPI], lofs[NPI];
#pragma unroll
  for (int u = 0; u < NPI; ++u) {
    const int it = tid + 256 * u;
    const int pix = min(it >> 2, NPIX - 1), q = it & 3;
    ipi[u] = it < NPIX * 4 ? pix / PJ : -100000;
    ipj[u] = pix % PJ;
    ioff[u] = ((pix / PJ) * ws_ + pix % PJ) * Cs + q * 8;
    lofs[u] = pix * PSTRIDE + q * 16;
  }
  float4 pre[PD][NPI][2];
  auto request = [&](int slot, int unit) {
    const int tile = tile_lo + unit / NCH, c = unit % NCH;
    const int tj = tile % tiles_j, ti = (tile / tiles_j) % tiles_i;
    const long img = tile / (tiles_j * tiles_i);
    const int y0 = ti * BI - (T - 1), x0 = tj * BJ - (T - 1);
    const float* base = small + img * (long)hs * ws_ * Cs + ((long)y0 * ws_ + x0) * Cs + c * 32;
#pragma unroll
    for (int u = 0; u < NPI; ++u) {
      const int y = y0 + ipi[u], x = x0 + ipj[u];
      pre[slot][u][0] = pre[slot][u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y >= 0 && y < hs && x >= 0 && x < ws_) {
        pre[slot][u][0] = *reinterpret_cast<const float4*>(base + ioff[u]);
        pre[slot][u][1] = *reinterpret_cast<const float4*>(base + ioff[u] + 4);
      }
    }
  };
  // this wave's two M tiles: block row wave, the two halves of the 32 block columns
  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int slot, int unit) {     // registers of `slot` -> the group's patch; re-arm the slot
    if (unit >= n_units) return;
#pragma unroll
    for (int u = 0; u < NPI; ++u) {
      if (ipi[u] >= 0 && !((dbg & 8) && pre[slot][u][0].x != 12345.f)) {
        const float v[8] = {pre[slot][u][0].x, pre[slot][u][0].y, pre[slot][u][0].z, pre[slot][u][0].w,
                            pre[slot][u][1].x, pre[slot][u][1].y, pre[slot][u][1].z, pre[slot][u][1].w};
        uint4 pl[3];
        split8(v, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(patch + p * PLANE + lofs[u]) = pl[p];
      }
    }
    if (unit + PD < n_units) request(slot, unit + PD);
  };
  auto multiply = [&](int unit) {            // the group's patch x the chunk's tap weights (+ the tile's epilogue)
    if (unit < 0 || unit >= n_units) return;
    const int c = unit % NCH;
    if (!(dbg & 2)) {
      const int r16 = lane & 15, kq = lane >> 4;
      const char* blc = bl + (long)c * (TT * NT * 3 * 1024);
#pragma unroll
      for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) {
          bf16x8 bf[NT][3];
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              bf[t][p] = *reinterpret_cast<const bf16x8*>(blc + (((a * T + b) * NT + t) * 3 + p) * 1024 + lane * 16);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            // row r16 of the tile = block column m * 16 + r16; its tap pixel in the patch
            const int pix = (wave + (T - 1) - a) * PJ + (m * 16 + r16 + (T - 1) - b);
            bf16x8 af[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
              af[p] = *reinterpret_cast<const bf16x8*>(patch + p * PLANE + pix * PSTRIDE + kq * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[t][0], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[t][2], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[t][1], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[t][0], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[t][1], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[t][0], acc[m][t], 0, 0, 0);
            }
          }
        }
    }
    if (c == NCH - 1) {
      // epilogue of the tile: element (row (lane >> 4) * 4 + r, column lane & 15) of tile (m, t)
      const int tile = tile_lo + unit / NCH;
      const int tj = tile % tiles_j, ti = (tile / tiles_j) % tiles_i;
      const long img = tile / (tiles_j * tiles_i);
      const int i = ti * BI + wave, j0 = tj * BJ;
      float* dst = big + img * (long)hb * wb * Cb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = t * 16 + (lane & 15);
        const int par = n / Cb, cb = n - par * Cb, py = par >> 1, px = par & 1;
        const int y = 2 * i + py;
        const bool ok = n < 4 * Cb && y < hb && !(dbg & 4);
        const float bv = (ok && bias) ? bias[cb] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = j0 + m * 16 + (lane >> 4) * 4 + r, x = 2 * j + px;
            if (ok && x < wb) dst[((long)y * wb + x) * Cb + cb] = acc[m][t][r] + bv;
          }
          acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };

#pragma unroll
  for (int s_ = 0; s_ < PD; ++s_)
    if (s_ < n_units) request(s_, s_);
  __syncthreads();
  // group 0: stage(u), multiply(u); group 1: multiply(u - 1), stage(u) - half a period apart
  for (int u0 = 0; u0 < max_units + 1; u0 += PD) {
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) {
      const int unit = u0 + s_;
      if (grp == 0) stage(s_, unit); else multiply(unit - 1);
      __syncthreads();
      if (grp == 0) multiply(unit); else stage(s_, unit);
      __syncthreads();
    }
  }
}

}  // namespace

// Decoder image layer, forward.  Returns 1 when the geometry is not covered (the caller then takes
// the generic path), 0 on success, an error code otherwise.  Workspace: the weight planes
// (NT * (k/2)^2 * Cs / 32 * 3 KB).
int dd_convT_image_fwd(const float* small, const float* w, const float* bias, float* big,
                       int n_img, int hs, int ws_, int Cs, int hb, int wb, int Cb, int k,
                       float* wsp, size_t ws_bytes, hipStream_t st) {
  const int T = k / 2, NT = (4 * Cb + 15) / 16, NCH = Cs / 32;
  if (k % 2 || T < 1 || T > 3 || Cs % 32 || NT > 2 || Cb < 1) return 1;
  if (!(NCH == 1 || NCH == 2) || (NT == 2 && T == 3 && NCH == 2)) return 1;   // (all tap weights resident in LDS)
  if ((((uintptr_t)small | (uintptr_t)w) & 15) != 0) return 1;
  if ((long)hs * ws_ * Cs * 4 > (1L << 30)) return 1;      // (32-bit patch offsets inside an image)
  const size_t pbytes = (size_t)NT * T * T * NCH * 3 * 1024;
  if (!wsp || ws_bytes < pbytes) return 1;
  char* planes = reinterpret_cast<char*>(wsp);
  const long total = (long)NT * T * T * NCH * 64;
  k_convT_image_wprep<<<(int)((total + 255) / 256), 256, 0, st>>>(w, k, Cb, Cs, NT, planes);
  DD_CHECK_LAUNCH("dd_conv2d_s2_up(image wprep)");
  const int dbg = getenv("DD_IMG_DBG") ? atoi(getenv("DD_IMG_DBG")) : 0;   // measurement aid: 2 no MFMAs, 4 no stores
  const int nbi = (hb + 1) / 2, nbj = (wb + 1) / 2;
  const int tj = (nbj + BJ - 1) / BJ, ti = (nbi + BI - 1) / BI;
  const long nt_ = (long)tj * ti * n_img;
  if (nt_ > (1 << 30)) return 1;
  const int n_tiles = (int)nt_;
  const int grid = n_tiles < 256 ? n_tiles : 256;    // one persistent workgroup per CU
#define LAUNCH(T_, NT_, NCH_) k_convT_image<T_, NT_, NCH_><<<grid, 512, 0, st>>>(small, planes, bias, big, hs, ws_, Cs, hb, wb, Cb, tj, ti, n_tiles, dbg)
#define PICK(NCH_)                                   \
  if (T == 3 && NT == 1) LAUNCH(3, 1, NCH_);         \
  else if (T == 2 && NT == 1) LAUNCH(2, 1, NCH_);    \
  else if (T == 2 && NT == 2) LAUNCH(2, 2, NCH_);    \
  else if (T == 1 && NT == 1) LAUNCH(1, 1, NCH_);    \
  else if (T == 1 && NT == 2) LAUNCH(1, 2, NCH_);    \
  else return 1;
  if (NCH == 1) {
    if (T == 3 && NT == 2) LAUNCH(3, 2, 1);
    else { PICK(1) }
  } else { PICK(2) }
#undef PICK
#undef LAUNCH
  DD_CHECK_LAUNCH("dd_conv2d_s2_up(image)");
  return 0;
}
