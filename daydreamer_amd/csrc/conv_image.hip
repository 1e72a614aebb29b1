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
#include <type_traits>
#include "../../include/daydreamer_hip.h"

// Measurement hooks of tools/conv_image_probe.py / conv_down_probe.py (2 no MFMAs, 4 no stores, 8 no
// staging - wrong results by design): only in a `make IMGDBG=1` build, where the DD_IMG_DBG
// environment variable selects them; the default library compiles the branches away.
#ifdef DD_BUILD_IMGDBG
#define IMG_DBG(bit) ((dbg & (bit)) != 0)
#else
#define IMG_DBG(bit) false
#endif

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
      if (ipi[u] >= 0 && !(IMG_DBG(8) && pre[slot][u][0].x != 12345.f)) {
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
    if (!IMG_DBG(2)) {
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
            // (tap weights as the ROW operand: the result tile is [parity x channel][block], a lane
            // holds four consecutive outputs of one 2 x 2 block - neighbours in memory)
            for (int t = 0; t < NT; ++t) {
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][0], af[2], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][2], af[0], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][1], af[1], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][0], af[1], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][1], af[0], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][0], af[0], acc[m][t], 0, 0, 0);
            }
          }
        }
    }
    if (c == NCH - 1) {
      // epilogue of the tile: elements (rows (lane >> 4) * 4 + r = outputs n, column lane & 15 = block) of tile (m, t);
      // output n = (py * 2 + px) * Cb + cb of block (i, j) lives at big[2i + py, 2j + px, cb]: the 2 * Cb outputs
      // of a parity row are contiguous, and (Cb odd or even) pairs (n, n + 1) with n even never straddle rows
      // when 2 * Cb is even - stored as 8-byte pairs where both are live
      const int tile = tile_lo + unit / NCH;
      const int tj = tile % tiles_j, ti = (tile / tiles_j) % tiles_i;
      const long img = tile / (tiles_j * tiles_i);
      const int i = ti * BI + wave, j0 = tj * BJ;
      float* dst = big + img * (long)hb * wb * Cb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int j = j0 + m * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const int n = t * 16 + (lane >> 4) * 4 + r;            // even; n + 1 is in the same parity row (2 * Cb even)
            const int py = n / (2 * Cb), w2 = n - py * 2 * Cb;      // w2 = px * Cb + cb, even
            const int y = 2 * i + py;
            const bool ok = n < 4 * Cb && y < hb && !IMG_DBG(4);
            const int x0 = 2 * j;                                    // the block's first output column
            if (ok && x0 * Cb + w2 + 1 < wb * Cb) {
              const float b0 = bias ? bias[w2 % Cb] : 0.f, b1 = bias ? bias[(w2 + 1) % Cb] : 0.f;
              *reinterpret_cast<float2*>(dst + ((long)y * wb + x0) * Cb + w2) =
                  make_float2(acc[m][t][r] + b0, acc[m][t][r + 1] + b1);
            } else if (ok && x0 * Cb + w2 < wb * Cb) {
              dst[((long)y * wb + x0) * Cb + w2] = acc[m][t][r] + (bias ? bias[w2 % Cb] : 0.f);
            }
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
  if ((((uintptr_t)small | (uintptr_t)w) & 15) != 0 || ((uintptr_t)big & 7) != 0 || (wb * Cb) % 2) return 1;
  if ((long)hs * ws_ * Cs * 4 > (1L << 30)) return 1;      // (32-bit patch offsets inside an image)
  const size_t pbytes = (size_t)NT * T * T * NCH * 3 * 1024;
  if (!wsp || ws_bytes < pbytes) return 1;
  char* planes = reinterpret_cast<char*>(wsp);
  const long total = (long)NT * T * T * NCH * 64;
  k_convT_image_wprep<<<(int)((total + 255) / 256), 256, 0, st>>>(w, k, Cb, Cs, NT, planes);
  DD_CHECK_LAUNCH("dd_conv2d_s2_up(image wprep)");
#ifdef DD_BUILD_IMGDBG
  const int dbg = getenv("DD_IMG_DBG") ? atoi(getenv("DD_IMG_DBG")) : 0;   // measurement aid: 2 no MFMAs, 4 no stores
#else
  const int dbg = 0;
#endif
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

namespace {
// ============================================================================================
// Image-side stride-2 VALID convolution: small[n,i,j,co] = in_scale * sum_{ky,kx,cb}
// big[n,2i+ky,2j+kx,cb] * W[ky,kx,cb,co] (+ bias) for a `big` tensor with a handful of channels:
// the encoder's first layer on the uint8 image (nets.py:291-305, Conv2D :547, `/255` of
// agent.py:129-130 fused) and the data gradient of the decoder's image layer.
//
// As a contraction K = k*k*Cb is tiny (48 / 108) and its rows are strided 4-byte (or 1-byte)
// gathers; the generic implicit GEMM ran these two call sites at 45-65 TFLOP/s (1.4-2 TB/s of their
// ~0.65 GB).  Here a workgroup owns 4 output rows x 32 output columns of one image: the k + 6
// image rows it needs are split ONCE into three bf16 planes in LDS (14 KB); K is laid out as
// k row segments of OPK octets (k*Cb values + zero-weight padding), so an A fragment is 8
// consecutive values of one image row (four 4-byte LDS reads per plane: the octet starts at an
// arbitrary 4-byte offset); the filter is a pre-split fragment-major plane cache resident in LDS.
// Persistent workgroups, the next tile's rows requested while the current one multiplies.
constexpr int DI = 4, DJ = 32;          // output rows x columns per workgroup tile

// B[(ky, slot)][co] = W[ky, kx, cb, co] for slot = kx*Cb + cb < k*Cb, else 0; K padded to k-steps of 32.
__global__ void k_conv_image_down_wprep(const float* __restrict__ w, int k, int Cb, int Cs, int OPK, int KS,
                                        char* __restrict__ planes) {
  const int NT = Cs / 16;
  const long total = (long)NT * KS * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long tk = i >> 6;
    const int ks = (int)(tk % KS), tile = (int)(tk / KS);
    const int co = tile * 16 + (lane & 15);
    const int o = ks * 4 + (lane >> 4), ky = o / OPK, oc = o - ky * OPK;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int slot = oc * 8 + j;
      v[j] = (ky < k && slot < k * Cb) ? w[((long)ky * k * Cb + slot) * Cs + co] : 0.f;
    }
    uint4 pl[3];
    split8(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint4*>(planes + (((long)tile * KS + ks) * 3 + p) * 1024 + lane * 16) = pl[p];
  }
}

// LNF: the layer's LayerNorm + ELU (nets.py:585-602, 510-513) in the epilogue - a pixel's 64
// channels sit in the 16 accumulator registers of four lanes (lane, lane ^ 16, lane ^ 32, lane ^ 48),
// so its statistics are two shuffles; writes the pre-norm rows (small), the activations (ln.out)
// and [pixels, 2] mean / rstd - what dd_ln_act_fwd would from a second pass over `small`.
// LNB (EPI = 2): the contraction's result is the gradient at the OUTPUT of the small-side layer's
// LayerNorm + ELU (the decoder's last hidden layer behind the image layer, nets.py:308-327): its
// backward (k_ln_act_bwd_v's expressions, activation recomputed from z) runs in the epilogue, `small`
// receives dz, and the LayerNorm scale / offset and bias gradients leave as one partial row [3][64]
// per workgroup - the gradient at the layer output is never written to or read from HBM.
struct DownLn {
  const float* gamma; const float* beta; float* out; float* stats;
  const float* z; float* partials;
};

template <int KS, int NT, typename TB, int EPI>
__global__ void __launch_bounds__(256, 2)
k_conv_image_down(const TB* __restrict__ big, const char* __restrict__ planes, const float* __restrict__ bias,
                  float* __restrict__ small, int hb, int wb, int Cb, int hs, int ws_, int Cs, int k, int OPK,
                  float in_scale, int tiles_j, int tiles_i, int n_tiles, int dbg, DownLn ln) {
  constexpr bool LNF = EPI == 1, LNB = EPI == 2;
  [[maybe_unused]] float pgs[LNB ? 16 : 1], pbs[LNB ? 16 : 1], pzs[LNB ? 16 : 1];
  if constexpr (LNB) {
#pragma unroll
    for (int q_ = 0; q_ < 16; ++q_) pgs[q_] = pbs[q_] = pzs[q_] = 0.f;
  }
  constexpr int EPV = sizeof(TB) == 1 ? 16 : 4;        // elements per 16-byte vector
  constexpr int RMAX = 12, RSMAX = 264;                // staged rows (k + 6 <= 12), row stride (elements, wb*Cb + 8 <= 264): 19 KB
  __shared__ __attribute__((aligned(16))) unsigned short patch[3][RMAX * RSMAX];
  __shared__ __attribute__((aligned(16))) char bl[NT * KS * 3 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int it = tid; it < NT * KS * 3 * 64; it += 256)
    *reinterpret_cast<uint4*>(bl + (long)it * 16) = *reinterpret_cast<const uint4*>(planes + (long)it * 16);
  const int rowlen = wb * Cb, RS = rowlen + 8, nrows = k + 2 * (DI - 1);
  const int vec_per_row = (rowlen + EPV - 1) / EPV, nvec = nrows * vec_per_row;
  constexpr int NV = 3;                                 // 16-byte vectors of a tile's rows per thread (<= 768 vectors)
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int tile_lo = blockIdx.x * per, tile_hi = min(n_tiles, tile_lo + per);
  uint4 pre[NV];
  auto request = [&](int tile) {
    const int ti = (tile / tiles_j) % tiles_i;
    const long img = tile / (tiles_j * tiles_i);
    const TB* base = big + (img * hb + 2 * ti * DI) * (long)rowlen;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int id = tid + 256 * v;
      const int r = id / vec_per_row, c = (id - r * vec_per_row) * EPV;
      const bool ok = id < nvec && 2 * ti * DI + r < hb;
      pre[v] = ok ? *reinterpret_cast<const uint4*>(base + (long)r * rowlen + c) : make_uint4(0, 0, 0, 0);
    }
  };
  if (tile_lo < tile_hi) request(tile_lo);
  __syncthreads();
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    // ---- stage this tile's image rows: split once, three planes
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int id = tid + 256 * v;
      if (id < nvec && !(IMG_DBG(8) && pre[v].x != 0x12345u)) {
        const int r = id / vec_per_row, c = (id - r * vec_per_row) * EPV;
        float f[EPV];
        if constexpr (sizeof(TB) == 1) {
          const unsigned w4[4] = {pre[v].x, pre[v].y, pre[v].z, pre[v].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f[4 * q] = (float)(w4[q] & 255u); f[4 * q + 1] = (float)((w4[q] >> 8) & 255u);
            f[4 * q + 2] = (float)((w4[q] >> 16) & 255u); f[4 * q + 3] = (float)(w4[q] >> 24);
          }
        } else {
          f[0] = __uint_as_float(pre[v].x); f[1] = __uint_as_float(pre[v].y);
          f[2] = __uint_as_float(pre[v].z); f[3] = __uint_as_float(pre[v].w);
        }
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
          unsigned h0, m0, l0, h1, m1, l1;
          split3(f[e], h0, m0, l0);
          split3(f[e + 1], h1, m1, l1);
          const int o = r * RS + c + e;
          *reinterpret_cast<unsigned*>(&patch[0][o]) = pack_hi(h0, h1);
          if constexpr (sizeof(TB) != 1) {
            *reinterpret_cast<unsigned*>(&patch[1][o]) = pack_hi(m0, m1);
            *reinterpret_cast<unsigned*>(&patch[2][o]) = pack_hi(l0, l1);
          }
        }
      }
    }
    // (the 8 pad elements behind a row are read by the last octet of the last pixels: finite
    // garbage times zero weights; they are zeroed once so that no NaN pattern can sit there)
    if (tile == tile_lo) {
      for (int it = tid; it < nrows * 8; it += 256) {
        const int r = it >> 3, e = it & 7;
#pragma unroll
        for (int p = 0; p < 3; ++p) patch[p][r * RS + rowlen + e] = 0;
      }
    }
    __syncthreads();
    if (tile + 1 < tile_hi) request(tile + 1);
    const int tj = tile % tiles_j, ti = (tile / tiles_j) % tiles_i;
    const long img = tile / (tiles_j * tiles_i);
    const int i = ti * DI + wave;                       // this wave's output row
    // ---- MFMAs: this wave's two M tiles (16 output columns each) x NT column tiles
    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, q = lane >> 4;
    if (!IMG_DBG(2))
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const int o = s_ * 4 + q;
      int ky = o / OPK;
      const int oc = o - ky * OPK;
      ky = min(ky, k - 1);                              // (padding octets: zero weights, any valid address)
      bf16x8 bf[NT][3];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[t][p] = *reinterpret_cast<const bf16x8*>(bl + ((t * KS + s_) * 3 + p) * 1024 + lane * 16);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int j = min(tj * DJ + m * 16 + r16, ws_ - 1);
        const int e0 = (2 * wave + ky) * RS + 2 * j * Cb + oc * 8;      // even element index (RS, Cb*2j, oc*8 even)
        bf16x8 af[3];
#pragma unroll
        for (int p = 0; p < (sizeof(TB) == 1 ? 1 : 3); ++p) {
          const unsigned* src = reinterpret_cast<const unsigned*>(&patch[p][e0]);
          af[p] = __builtin_bit_cast(bf16x8, make_uint4(src[0], src[1], src[2], src[3]));
        }
#pragma unroll
        // (filter fragment as the ROW operand: the result tile is [channel][pixel], so a lane ends
        // up with four consecutive channels of one pixel = one 16-byte store)
        for (int t = 0; t < NT; ++t) {
          // (a uint8 image - 0..255, the `/255` is in_scale in the epilogue - is exact in its high
          // plane: the three products with its all-zero lower planes are left out, bit-identical)
          if constexpr (sizeof(TB) != 1) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][0], af[2], acc[m][t], 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][2], af[0], acc[m][t], 0, 0, 0);
          if constexpr (sizeof(TB) != 1) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][1], af[1], acc[m][t], 0, 0, 0);
          if constexpr (sizeof(TB) != 1) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][0], af[1], acc[m][t], 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][1], af[0], acc[m][t], 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][0], af[0], acc[m][t], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: elements (rows (lane >> 4) * 4 + r = channels, column lane & 15 = output column) of tile (m, t)
    if constexpr (LNB) {
      static_assert(!LNB || NT == 4, "LayerNorm epilogue: 64 channels");
      if (i < hs) {
        const long prow = (img * hs + i) * (long)ws_;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int j = tj * DJ + m * 16 + (lane & 15);
          const bool live = j < ws_;
          const long pr = prow + min(j, ws_ - 1);
          const float2 ms = *reinterpret_cast<const float2*>(ln.stats + pr * 2);
          const float mean = ms.x, rstd = ms.y;
          float xh[NT][4], g[NT][4], dy[NT][4];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int co = t * 16 + (lane >> 4) * 4;
            const float4 zz = *reinterpret_cast<const float4*>(ln.z + pr * Cs + co);
            const float4 gm = *reinterpret_cast<const float4*>(ln.gamma + co);
            const float4 bt = *reinterpret_cast<const float4*>(ln.beta + co);
            const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, gmv[4] = {gm.x, gm.y, gm.z, gm.w};
            const float btv[4] = {bt.x, bt.y, bt.z, bt.w}, bvv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float d = live ? in_scale * acc[m][t][r] + bvv[r] : 0.f;
              xh[t][r] = (zv[r] - mean) * rstd;
              const float y = xh[t][r] * gmv[r] + btv[r];
              dy[t][r] = d * (y > 0.f ? 1.f : __builtin_amdgcn_exp2f(y * 1.44269504088896341f));
              g[t][r] = dy[t][r] * gmv[r];
              s1 += g[t][r];
              s2 += g[t][r] * xh[t][r];
            }
          }
          s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          s1 /= 64.f; s2 /= 64.f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int co = t * 16 + (lane >> 4) * 4;
            float dz[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dz[r] = rstd * (g[t][r] - s1 - xh[t][r] * s2);
              pgs[t * 4 + r] += dy[t][r] * xh[t][r];
              pbs[t * 4 + r] += dy[t][r];
              pzs[t * 4 + r] += live ? dz[r] : 0.f;
            }
            if (live) *reinterpret_cast<float4*>(small + (prow + j) * Cs + co) = make_float4(dz[0], dz[1], dz[2], dz[3]);
          }
        }
      }
    } else
    if constexpr (LNF) {
      static_assert(!LNF || NT == 4, "LayerNorm epilogue: 64 channels");
      if (i < hs) {
        const long prow = (img * hs + i) * (long)ws_;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int j = tj * DJ + m * 16 + (lane & 15);
          float zv[NT][4];
          float ps = 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int co = t * 16 + (lane >> 4) * 4;
            const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            zv[t][0] = in_scale * acc[m][t][0] + bv.x; zv[t][1] = in_scale * acc[m][t][1] + bv.y;
            zv[t][2] = in_scale * acc[m][t][2] + bv.z; zv[t][3] = in_scale * acc[m][t][3] + bv.w;
            ps += (zv[t][0] + zv[t][1]) + (zv[t][2] + zv[t][3]);
          }
          ps += __shfl_xor(ps, 16, 64);
          ps += __shfl_xor(ps, 32, 64);
          const float mean = ps / 64.f;
          float pv = 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) pv += (zv[t][r] - mean) * (zv[t][r] - mean);
          pv += __shfl_xor(pv, 16, 64);
          pv += __shfl_xor(pv, 32, 64);
          const float rstd = rsqrtf(pv / 64.f + 1e-3f);
          if (j < ws_) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int co = t * 16 + (lane >> 4) * 4;
              const float4 gm = *reinterpret_cast<const float4*>(ln.gamma + co);
              const float4 bt = *reinterpret_cast<const float4*>(ln.beta + co);
              const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
              float o[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float y = (zv[t][r] - mean) * rstd * gmv[r] + btv[r];
                // (hardware exponential: within 1e-7 absolute of expm1f, a tenth of its instructions)
                o[r] = y > 0.f ? y : __builtin_amdgcn_exp2f(y * 1.44269504088896341f) - 1.f;
              }
              *reinterpret_cast<float4*>(small + (prow + j) * Cs + co) = make_float4(zv[t][0], zv[t][1], zv[t][2], zv[t][3]);
              *reinterpret_cast<float4*>(ln.out + (prow + j) * Cs + co) = make_float4(o[0], o[1], o[2], o[3]);
            }
            if ((lane >> 4) == 0) *reinterpret_cast<float2*>(ln.stats + (prow + j) * 2) = make_float2(mean, rstd);
          }
        }
      }
    } else
    if (i < hs && !IMG_DBG(4)) {
      float* dst = small + ((img * hs + i) * (long)ws_) * Cs;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int co = t * 16 + (lane >> 4) * 4;
        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int j = tj * DJ + m * 16 + (lane & 15);
          if (j < ws_)
            *reinterpret_cast<float4*>(dst + (long)j * Cs + co) =
                make_float4(in_scale * acc[m][t][0] + bv.x, in_scale * acc[m][t][1] + bv.y,
                            in_scale * acc[m][t][2] + bv.z, in_scale * acc[m][t][3] + bv.w);
        }
      }
    }
    __syncthreads();     // the patch is free for the next tile
  }
  if constexpr (LNB) {
    // lanes with equal lane >> 4 hold the sums of the same 16 channels (t * 16 + (lane >> 4) * 4 + r)
    // over their pixels: add the 16 lanes, then the four waves through LDS -> [3][64] per workgroup
    float* red = reinterpret_cast<float*>(&patch[0][0]);       // [4 waves][3][64]
#pragma unroll
    for (int q_ = 0; q_ < 16; ++q_) {
#pragma unroll
      for (int o_ = 1; o_ < 16; o_ <<= 1) {
        pgs[q_] += __shfl_xor(pgs[q_], o_, 64); pbs[q_] += __shfl_xor(pbs[q_], o_, 64); pzs[q_] += __shfl_xor(pzs[q_], o_, 64);
      }
    }
    if ((lane & 15) == 0) {
#pragma unroll
      for (int q_ = 0; q_ < 16; ++q_) {
        const int co = (q_ >> 2) * 16 + (lane >> 4) * 4 + (q_ & 3);
        red[wave * 192 + co] = pgs[q_]; red[wave * 192 + 64 + co] = pbs[q_]; red[wave * 192 + 128 + co] = pzs[q_];
      }
    }
    __syncthreads();
    if (tid < 192) ln.partials[(long)blockIdx.x * 192 + tid] = (red[tid] + red[192 + tid]) + (red[384 + tid] + red[576 + tid]);
  }
}

}  // namespace

// Returns 1 when the geometry is not covered (the caller then takes the generic path).
int dd_conv_image_down(const void* big, int big_is_u8, const float* w, const float* bias, float* small,
                       int n_img, int hb, int wb, int Cb, int hs, int ws_, int Cs, int k, float in_scale,
                       float* wsp, size_t ws_bytes, hipStream_t st, const float* ln_gamma, const float* ln_beta,
                       float* ln_out, float* ln_stats, const float* ln_z, int* n_partials) {
  const int OPK = (k * Cb + 7) / 8, KS = (k * OPK + 3) / 4, NT = Cs / 16;
  const int epv = big_is_u8 ? 16 : 4, rowlen = wb * Cb;
  if (Cb < 1 || Cb > 4 || Cs != 64 || k < 2 || k + 2 * (DI - 1) > 12 || rowlen + 8 > 264) return 1;
  if (!(KS == 2 || KS == 3 || KS == 5) || rowlen % epv || (rowlen % 2) || ((uintptr_t)big & 15) || ((uintptr_t)w & 3)) return 1;
  if (((uintptr_t)small & 15) || (bias && ((uintptr_t)bias & 15))) return 1;
  if ((k + 2 * (DI - 1)) * ((rowlen + epv - 1) / epv) > 768) return 1;
  if (2 * (ws_ - 1) * Cb + OPK * 8 > rowlen + 8) return 1;          // the last octet stays inside the padded row
  const size_t pbytes = (size_t)NT * KS * 3 * 1024;
  if (!wsp || ws_bytes < pbytes) return 1;
  char* planes = reinterpret_cast<char*>(wsp);
  const long total = (long)NT * KS * 64;
  k_conv_image_down_wprep<<<(int)((total + 255) / 256), 256, 0, st>>>(w, k, Cb, Cs, OPK, KS, planes);
  DD_CHECK_LAUNCH("dd_conv2d_s2_down(image wprep)");
  const int tj = (ws_ + DJ - 1) / DJ, ti = (hs + DI - 1) / DI;
  const long nt_ = (long)tj * ti * n_img;
  if (nt_ > (1 << 30)) return 1;
  const int n_tiles = (int)nt_;
  const int grid = n_tiles < 512 ? n_tiles : 512;
#ifdef DD_BUILD_IMGDBG
  static const int dbg = getenv("DD_IMG_DBG") ? atoi(getenv("DD_IMG_DBG")) : 0;   // measurement aid: 2 no MFMAs, 4 no stores, 8 no staging
#else
  const int dbg = 0;
#endif
  // (LayerNorm backward: the partial rows [grid][3][64] follow the weight planes in the workspace)
  float* partials = reinterpret_cast<float*>(planes + ((pbytes + 255) & ~(size_t)255));
  if (ln_z && ws_bytes < ((pbytes + 255) & ~(size_t)255) + (size_t)grid * 192 * sizeof(float)) return 1;
  const DownLn ln{ln_gamma, ln_beta, ln_out, ln_stats, ln_z, partials};
  if ((ln_out || ln_z) && ((((uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)ln_out | (uintptr_t)ln_z) & 15) || ((uintptr_t)ln_stats & 7))) return 1;
  const int epi = ln_z ? 2 : (ln_out ? 1 : 0);
  if (n_partials) *n_partials = grid;
#define LDX(KS_, T_, L_) k_conv_image_down<KS_, 4, T_, L_><<<grid, 256, 0, st>>>(                     \
        (const T_*)big, planes, bias, small, hb, wb, Cb, hs, ws_, Cs, k, OPK, in_scale, tj, ti, n_tiles, dbg, ln)
#define LD(KS_)                                                                                        \
  if (KS == KS_) {                                                                                     \
    if (big_is_u8) { if (epi == 1) LDX(KS_, unsigned char, 1); else if (epi == 0) LDX(KS_, unsigned char, 0); else return 1; } \
    else { if (epi == 2) LDX(KS_, float, 2); else if (epi == 1) LDX(KS_, float, 1); else LDX(KS_, float, 0); } \
  }
  LD(2) LD(3) LD(5)
#undef LD
#undef LDX
  DD_CHECK_LAUNCH("dd_conv2d_s2_down(image)");
  return 0;
}

namespace {
// ============================================================================================
// Image-side filter gradient: dw[(ky, kx, cb), co] = sum over images n and output pixels (i, j) of
// big[n, 2i+ky, 2j+kx, cb] * small[n, i, j, co] for a `big` tensor with a handful of channels - the
// encoder's first layer on the uint8 image (nets.py:291-305, Conv2D :547 under GradientTape,
// tfutils.py:214) and the decoder's image layer (nets.py:308-327).
//
// As a contraction M = k*k*Cb is 48 / 108 rows, N = 64, K = n*hs*ws = 2.3 M: the generic implicit
// GEMM gathers a 4-byte (1-byte) element per (tap, pixel) straight from global memory for every
// k-tile and ran the two call sites at 41 / 54 TFLOP/s (364 / 577 us for 0.65 / 0.70 GB).  Here
// the contraction step is ONE output row (<= 32 pixels) of one image:
//   * the k image rows it touches sit in a rolling 8-row LDS window, split once into bf16 planes
//     and de-interleaved by column parity / channel and stored once per tap shift s = kx / 2, so
//     the 8 consecutive output pixels of a tap (image columns 2j + kx) are 16 contiguous, aligned
//     bytes: the A fragment of v_mfma_f32_16x16x32_bf16 is one ds_read_b128 per plane;
//   * the 64-channel row is staged [pixel][channel] as three planes and read through
//     ds_read_tr16_b64 (the transposing read of gemm_core.h's row-contiguous operands);
//   * a uint8 image is exact in ONE bf16 plane (0..255): three products instead of six, `/255`
//     applied once to the sum;
//   * each workgroup keeps its [taps x 64] partial sum in registers over its rows (work item =
//     half an image, so that consecutive rows reuse the window) and writes one slab; the slabs
//     are added in workgroup order by the split-K reduce pass (deterministic).
constexpr int WG_RUN = 40;      // bf16 elements per (parity, channel, shift) run: 32 + zero pad
constexpr int WG_RW = 8;        // image rows in the window (k + 2 new ones <= 8)
constexpr int WG_GSTR = 192;    // bytes per pixel of a staged 64-channel plane (128 + pad)
#ifndef WG_IL_VALU
#define WG_IL_VALU 6            // VALU instructions of the staging issued behind each matrix instruction
#endif

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_w;

// LNB: `G` is the gradient at the OUTPUT of the small-side layer's LayerNorm + ELU (nets.py:585-602,
// 510-513); its backward (the arithmetic of k_ln_act_bwd_v, rowops.hip) runs on the staged row -
// a pixel's 64 channels are 16 consecutive lanes - so dz is never written to or read from HBM;
// the LayerNorm parameter gradients and the bias gradient (column sums of dy * xhat, dy, dz) are
// kept per thread over the workgroup's rows and leave as one partial row [3][64] per workgroup.
struct WgradLn {
  const float* z; const float* stats; const float* gamma; const float* beta; float* partials;
};

template <int K, int CB, typename TI, bool LNB>
__global__ void __launch_bounds__(256, (K == 6 && CB == 4 && sizeof(TI) == 4) ? 1 : 2)   // (that window + row buffers: 83 KB of LDS)
k_conv_image_wgrad(const TI* __restrict__ img, const float* __restrict__ G, float* __restrict__ slabs,
                   int hb, int wb, int hs, int ws_, int n_items, int HR, int per, int dbg, WgradLn ln) {
  constexpr bool U8 = sizeof(TI) == 1;
  constexpr int NP = U8 ? 1 : 3;                  // planes of the image operand
  constexpr int NS = K / 2, NTAP = K * K * CB, MT = (NTAP + 15) / 16;
  constexpr int ROWEL = 2 * CB * NS * WG_RUN;     // elements per window row and plane
  constexpr int EPV = U8 ? 4 : 2;                 // image elements per thread-vector (4 bytes / float2: the staging
                                                  // of a row's 192 elements is spread over 48 / 96 threads)
  __shared__ __attribute__((aligned(16))) unsigned short iw[NP][WG_RW][ROWEL];
  __shared__ __attribute__((aligned(16))) unsigned char gp[2][3][32 * WG_GSTR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rowlen = wb * CB, VPR = rowlen / EPV;           // image row in elements / vectors
  constexpr int NIV = U8 ? (K > 4 ? 2 : 1) : (K > 4 ? 3 : 2);  // image vectors per thread and output row (K rows x 48 / 96 vectors)

  for (int it = tid; it < NP * WG_RW * ROWEL / 2; it += 256) reinterpret_cast<unsigned*>(&iw[0][0][0])[it] = 0u;

  // this lane's tap of M tile m (A operand row lane & 15): window-row offset ky and the byte
  // offset of its run (parity, channel, shift) + this lane's 8-pixel chunk
  int tky[MT], toff[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int tap = min(m * 16 + (lane & 15), NTAP - 1);
    const int ky = tap / (K * CB), rem = tap - ky * (K * CB), kx = rem / CB, cb = rem - kx * CB;
    tky[m] = ky;
    toff[m] = ((((kx & 1) * CB + cb) * NS + (kx >> 1)) * WG_RUN + (lane >> 4) * 8) * 2;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int item_lo = blockIdx.x * per, item_hi = min(n_items, item_lo + per);
  // cursor over the output rows of this workgroup's items: (item, i), first = first row of its item
  struct Cur { int item, i, first, ok; };
  auto start = [&](int item) {
    Cur c; c.item = item; c.i = (item & 1) * HR; c.first = 1; c.ok = item < item_hi;
    return c;
  };
  auto next = [&](Cur c) {
    Cur d = c;
    const int i_end = min(hs, ((c.item & 1) + 1) * HR);
    if (c.i + 1 < i_end) { d.i = c.i + 1; d.first = 0; }
    else d = start(c.item + 1);
    return d;
  };
  uint4 gv[4][2];
  [[maybe_unused]] uint4 gz[LNB ? 4 : 1][2];
  [[maybe_unused]] float2 gs[LNB ? 4 : 1][2];
  [[maybe_unused]] float4 ln_g = make_float4(0.f, 0.f, 0.f, 0.f), ln_b = ln_g, pg = ln_g, pb = ln_g, pz = ln_g;
  if constexpr (LNB) {
    ln_g = *reinterpret_cast<const float4*>(ln.gamma + (tid & 15) * 4);
    ln_b = *reinterpret_cast<const float4*>(ln.beta + (tid & 15) * 4);
  }
  typename std::conditional<U8, unsigned, uint2>::type iv[4][NIV];
  auto load = [&](int set, Cur c) {
    if (!c.ok || IMG_DBG(8)) return;
    const int n = c.item >> 1;
    const float* grow = G + (((long)n * hs + c.i) * ws_) * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int id = tid + 256 * u, pix = id >> 4, c4 = (id & 15) * 4;
      gv[set][u] = pix < ws_ ? *reinterpret_cast<const uint4*>(grow + pix * 64 + c4) : make_uint4(0, 0, 0, 0);
      if constexpr (LNB) {
        const long prow = ((long)n * hs + c.i) * ws_ + pix;
        gz[set][u] = pix < ws_ ? *reinterpret_cast<const uint4*>(ln.z + prow * 64 + c4) : make_uint4(0, 0, 0, 0);
        gs[set][u] = pix < ws_ ? *reinterpret_cast<const float2*>(ln.stats + prow * 2) : make_float2(0.f, 0.f);
      }
    }
    const int r0 = c.first ? 2 * c.i : 2 * c.i + K - 2, nr = c.first ? K : 2;
    const TI* ibase = img + ((long)n * hb + r0) * rowlen;
#pragma unroll
    for (int u = 0; u < NIV; ++u) {
      const int id = tid + 256 * u, rr = id / VPR, v = id - rr * VPR;
      const bool okv = rr < nr && r0 + rr < hb;
      if constexpr (U8) iv[set][u] = okv ? *reinterpret_cast<const unsigned*>(ibase + (long)rr * rowlen + v * EPV) : 0u;
      else iv[set][u] = okv ? *reinterpret_cast<const uint2*>(ibase + (long)rr * rowlen + v * EPV) : make_uint2(0, 0);
    }
  };
  auto stage = [&](int set, Cur c, int buf) {
    if (!c.ok) return;
    // ---- the 64-channel row: [pixel][channel] planes
    if (!IMG_DBG(16))
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int id = tid + 256 * u, pix = id >> 4, c4 = (id & 15) * 4;
      unsigned h[4], m[4], l[4];
      float gvf[4] = {__uint_as_float(gv[set][u].x), __uint_as_float(gv[set][u].y),
                      __uint_as_float(gv[set][u].z), __uint_as_float(gv[set][u].w)};
      if constexpr (LNB) {   // dz = LayerNorm + ELU backward of this pixel (k_ln_act_bwd_v's expressions)
        const float zz[4] = {__uint_as_float(gz[set][u].x), __uint_as_float(gz[set][u].y),
                             __uint_as_float(gz[set][u].z), __uint_as_float(gz[set][u].w)};
        const float gm[4] = {ln_g.x, ln_g.y, ln_g.z, ln_g.w}, bt[4] = {ln_b.x, ln_b.y, ln_b.z, ln_b.w};
        const float mean = gs[set][u].x, rstd = gs[set][u].y;
        float xh[4], g[4], dy[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[j] = (zz[j] - mean) * rstd;
          const float y = xh[j] * gm[j] + bt[j];
          // ELU'(y) = 1 (y > 0) or exp(y): k_ln_act_bwd_v forms it as expm1f(y) + 1; the hardware
          // exponential agrees with that to a few ulp at a tenth of the instructions
          dy[j] = gvf[j] * (y > 0.f ? 1.f : __builtin_amdgcn_exp2f(y * 1.44269504088896341f));
          g[j] = dy[j] * gm[j];
        }
        s1 = (g[0] + g[1]) + (g[2] + g[3]);
        s2 = (g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3]);
#pragma unroll
        for (int o_ = 1; o_ < 16; o_ <<= 1) { s1 += __shfl_xor(s1, o_, 64); s2 += __shfl_xor(s2, o_, 64); }
        s1 /= 64.f; s2 /= 64.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) gvf[j] = rstd * (g[j] - s1 - xh[j] * s2);
        pg.x += dy[0] * xh[0]; pg.y += dy[1] * xh[1]; pg.z += dy[2] * xh[2]; pg.w += dy[3] * xh[3];
        pb.x += dy[0]; pb.y += dy[1]; pb.z += dy[2]; pb.w += dy[3];
        pz.x += gvf[0]; pz.y += gvf[1]; pz.z += gvf[2]; pz.w += gvf[3];
      }
      split3(gvf[0], h[0], m[0], l[0]);
      split3(gvf[1], h[1], m[1], l[1]);
      split3(gvf[2], h[2], m[2], l[2]);
      split3(gvf[3], h[3], m[3], l[3]);
      const int o = pix * WG_GSTR + c4 * 2;
      *reinterpret_cast<uint2*>(&gp[buf][0][o]) = make_uint2(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]));
      *reinterpret_cast<uint2*>(&gp[buf][1][o]) = make_uint2(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]));
      *reinterpret_cast<uint2*>(&gp[buf][2][o]) = make_uint2(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]));
    }
    if (IMG_DBG(4)) return;
    // ---- the new image rows: planes, de-interleaved, one copy per tap shift
    const int r0 = c.first ? 2 * c.i : 2 * c.i + K - 2, nr = c.first ? K : 2;
#pragma unroll
    for (int u = 0; u < NIV; ++u) {
      const int id = tid + 256 * u, rr = id / VPR, v = id - rr * VPR;
      if (rr < nr) {
        const int slot = (r0 + rr) & (WG_RW - 1);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          float f;
          if constexpr (U8) f = (float)((iv[set][u] >> (8 * e)) & 255u);
          else f = __uint_as_float(e == 0 ? iv[set][u].x : iv[set][u].y);
          unsigned pl[3];
          split3(f, pl[0], pl[1], pl[2]);
          const int idx = v * EPV + e, x = idx / CB, cb = idx - x * CB;
          const int run0 = (((x & 1) * CB + cb) * NS) * WG_RUN, mm = x >> 1;
#pragma unroll
          for (int s_ = 0; s_ < NS; ++s_)
            if (mm - s_ >= 0) {
#pragma unroll
              for (int p = 0; p < NP; ++p) iw[p][slot][run0 + s_ * WG_RUN + mm - s_] = (unsigned short)(pl[p] >> 16);
            }
        }
      }
    }
  };
  auto multiply = [&](Cur c, int buf) {
    if (!c.ok || IMG_DBG(2)) return;
    typedef __attribute__((address_space(3))) bf16x4_w* lds_p;
    bf16x8 bf[3];
    {
      const int i16 = lane & 15, g4 = lane >> 4;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned char* q = &gp[buf][p][(g4 * 8 + (i16 >> 2)) * WG_GSTR + (wave * 16 + 4 * (i16 & 3)) * 2];
        const bf16x4_w lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(q));
        const bf16x4_w hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(q + 4 * WG_GSTR));
        const uint2 a_ = __builtin_bit_cast(uint2, lo), b_ = __builtin_bit_cast(uint2, hi);
        bf[p] = __builtin_bit_cast(bf16x8, make_uint4(a_.x, a_.y, b_.x, b_.y));
      }
    }
    // all A fragments first, then the products plane pair by plane pair over the M tiles:
    // consecutive matrix instructions go to different accumulators (a tile's six products
    // back to back are one dependent chain)
    bf16x8 af[MT][NP];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int slot = (2 * c.i + tky[m]) & (WG_RW - 1);
#pragma unroll
      for (int p = 0; p < NP; ++p)
        af[m][p] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(&iw[p][slot][0]) + toff[m]);
    }
    // (smallest terms first, as every split contraction of the library; uint8: the image plane is
    // exact, the products with its zero lower planes are left out)
    constexpr int PA_[6] = {NP - 1, 0, NP > 1 ? 1 : 0, NP > 1 ? 1 : 0, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if (U8 && !(q == 1 || q == 4 || q == 5)) continue;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][PA_[q]], bf[PB_[q]], acc[m], 0, 0, 0);
    }
  };

  Cur c0 = start(item_lo), c1 = next(c0), c2 = next(c1), c3 = next(c2);
  load(0, c0);
  load(1, c1);
  load(2, c2);
  __syncthreads();            // (the window's zero fill)
  stage(0, c0, 0);
  __syncthreads();
  // (a role-separated form - four waves staging, four multiplying, one barrier per row - was
  // measured SLOWER, 567 / 331 us: with half the waves issuing loads the stream fell to 1.8 TB/s)
  // Inside an item the multiply of row r and the staging of row r + 1 touch different LDS (other
  // buffer, other window rows): one block, the matrix instructions interleaved with the staging's
  // VALU / LDS work by the scheduler hints (they ran back to back, each about a third of the time).
  // A new item rewrites window rows the multiply reads: there the staging waits for it.
#define WG_STEP(P)                                                                         \
  load(((P) + 3) & 3, c3);                                                                 \
  if (c1.first) {                                                                          \
    multiply(c0, (P) & 1);                                                                 \
    __syncthreads();                                                                       \
    stage(((P) + 1) & 3, c1, ((P) + 1) & 1);                                               \
  } else {                                                                                 \
    multiply(c0, (P) & 1);                                                                 \
    stage(((P) + 1) & 3, c1, ((P) + 1) & 1);                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < MT * (U8 ? 3 : 6); ++i_) {                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
      __builtin_amdgcn_sched_group_barrier(0x002, WG_IL_VALU, 0);                          \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                   \
    }                                                                                      \
  }                                                                                        \
  __syncthreads();                                                                         \
  c0 = c1; c1 = c2; c2 = c3; c3 = next(c3);
  while (c0.ok) {
    WG_STEP(0)
    if (!c0.ok) break;
    WG_STEP(1)
    if (!c0.ok) break;
    WG_STEP(2)
    if (!c0.ok) break;
    WG_STEP(3)
  }
#undef WG_STEP
  if constexpr (LNB) {
    // the threads tid & 15 == q hold the sums of channels 4 q .. 4 q + 3 over their pixels: add the
    // 16 pixel groups (tid >> 4) through LDS -> one partial row [3][64] of this workgroup
    float* red = reinterpret_cast<float*>(&gp[0][0][0]);       // [16][3][64]  (the row buffers are free: barrier above)
    float* r0 = red + (tid >> 4) * 192 + (tid & 15) * 4;
    *reinterpret_cast<float4*>(r0) = pg;
    *reinterpret_cast<float4*>(r0 + 64) = pb;
    *reinterpret_cast<float4*>(r0 + 128) = pz;
    __syncthreads();
    if (tid < 192) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[i * 192 + tid];
      ln.partials[(long)blockIdx.x * 192 + tid] = t;
    }
  }
  // ---- this workgroup's slab: D element (row (lane >> 4) * 4 + r, column lane & 15) of tile m
  float* slab = slabs + (long)blockIdx.x * NTAP * 64;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tap = m * 16 + (lane >> 4) * 4 + r;
      if (tap < NTAP) slab[tap * 64 + wave * 16 + (lane & 15)] = acc[m][r];
    }
}

}  // namespace

// Image-side filter gradient.  Returns 1 when the geometry is not covered (the caller then takes the
// generic path), else 0 with *n_slabs partial sums [n_slabs][k*k*Cb][64] at the start of the
// workspace: the caller adds them (split-K reduce pass; uint8: times in_scale).  With ln_z != NULL
// `small` is the gradient at the output of the small-side LayerNorm + ELU, whose backward is applied
// while the rows are staged (k_conv_image_wgrad LNB); the [n_slabs][3][64] partial rows of the
// LayerNorm scale / offset gradients and the bias gradient follow the slabs in the workspace.
int dd_conv_image_wgrad(const void* big, int big_is_u8, const float* small, int n_img, int hb, int wb,
                        int Cb, int hs, int ws_, int Cs, int k, float* wsp, size_t ws_bytes,
                        int* n_slabs, hipStream_t st, const float* ln_z, const float* ln_stats,
                        const float* ln_gamma, const float* ln_beta) {
  static const int off = getenv("DD_IMG_WGRAD_OFF") ? atoi(getenv("DD_IMG_WGRAD_OFF")) : 0;
  if (off || Cs != 64 || !(k == 4 || k == 6) || !(Cb == 3 || Cb == 4) || wb > 64 || ws_ > 32 || ws_ < 1 || hs < 2 || n_img < 1) return 1;
  const int rowlen = wb * Cb;
  if (rowlen % 4 || (((uintptr_t)big | (uintptr_t)small) & 15)) return 1;
  if (rowlen > 256) return 1;                                          // (image vectors per thread: K rows x <= 64 / 128)
  if (2 * (hs - 1) + k > hb || 2 * (ws_ - 1) + k > wb) return 1;
  if (ln_z && ((((uintptr_t)ln_z | (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) || ((uintptr_t)ln_stats & 7))) return 1;
  const int HR = (hs + 1) / 2;
  const long items_l = 2l * n_img;
  if (items_l > (1 << 30)) return 1;
  const int n_items = (int)items_l;
  int grid = n_items < 512 ? n_items : 512;
  const int per = (n_items + grid - 1) / grid;
  grid = (n_items + per - 1) / per;                                   // every workgroup has at least one item
#ifdef DD_BUILD_IMGDBG
  static const int dbg = getenv("DD_IMG_DBG") ? atoi(getenv("DD_IMG_DBG")) : 0;   // 2 no MFMAs, 4 no image staging, 16 no row staging, 8 no loads
#else
  const int dbg = 0;
#endif
  const size_t need = (size_t)grid * (k * k * Cb * 64 + (ln_z ? 192 : 0)) * sizeof(float);
  if (!wsp || ws_bytes < need) return 1;
  WgradLn ln{ln_z, ln_stats, ln_gamma, ln_beta, wsp + (size_t)grid * k * k * Cb * 64};
#define LW(K_, T_, L_) { if (Cb == 3) k_conv_image_wgrad<K_, 3, T_, L_><<<grid, 256, 0, st>>>((const T_*)big, small, wsp, hb, wb, hs, ws_, n_items, HR, per, dbg, ln); \
                         else k_conv_image_wgrad<K_, 4, T_, L_><<<grid, 256, 0, st>>>((const T_*)big, small, wsp, hb, wb, hs, ws_, n_items, HR, per, dbg, ln); }
#define LWL(K_, T_) { if (ln_z) LW(K_, T_, true) else LW(K_, T_, false) }
  if (big_is_u8) { if (k == 4) LWL(4, unsigned char) else LWL(6, unsigned char) }
  else { if (k == 4) LWL(4, float) else LWL(6, float) }
#undef LWL
#undef LW
  DD_CHECK_LAUNCH("dd_conv2d_s2_wgrad(image)");
  *n_slabs = grid;
  return 0;
}
