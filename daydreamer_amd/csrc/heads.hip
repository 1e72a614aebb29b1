// Loss heads and imagination-time scalar paths: reconstruction / reward /
// continue log-prob losses (fused forward value + gradient), the Normal policy
// head, the lambda-return scan with discount weights, and the critic / actor
// loss seeds.
//
// Reference: MSEDist / SymlogDist tfutils.py:305-356, Bernoulli head
// nets.py:469-471, Normal head nets.py:461-468, WorldModel.imagine
// agent.py:256-259 (cont, weight), VFunction.target 'gve' agent.py:434-440,
// VFunction.train agent.py:398-417, ImagActorCritic.loss agent.py:351-381.
#include "dd_common.h"
#include "../../include/daydreamer_hip.h"

namespace {

// loss[row] = sum over pixels and channels [c0,c1) of (sigmoid(z) - x/255)^2 ;
// dz = coef * 2 (s - x) s (1 - s) on those channels (others untouched).
__global__ void __launch_bounds__(256)
k_image_loss(const float* __restrict__ z, const unsigned char* __restrict__ img,
             float* __restrict__ loss, float* __restrict__ dz, long P, int ctot, int c0, int c1,
             float coef) {
  const long row = blockIdx.x;
  const float* zr = z + row * P;
  const unsigned char* ir = img + row * P;
  float* dr = dz + row * P;
  float acc = 0.f;
  const bool all = (c0 == 0 && c1 == ctot);
  for (long p = threadIdx.x; p < P; p += 256) {
    if (!all) { int c = (int)(p % ctot); if (c < c0 || c >= c1) continue; }
    float s = sigmoidf_(zr[p]);
    float d = s - (float)ir[p] * (1.f / 255.f);
    acc += d * d;
    dr[p] = coef * 2.f * d * s * (1.f - s);
  }
  acc = wave_sum(acc);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[row] = sh[0] + sh[1] + sh[2] + sh[3];
}

// Agent.report's video grids on the device (tfutils.video_grid tfutils.py:390-392 of
// WorldModel.report agent.py:266-282 / Greedy.report behaviors.py:32-46): one thread per element of
// the model section, out[t][sec * H + h][b * W + w][c - c0]; with the uint8 truth image three
// sections (truth / 255 | model = sigmoid(z) | error = (model - truth + 1) / 2), else the model.
__global__ void __launch_bounds__(256)
k_video_grid(const float* __restrict__ z, const unsigned char* __restrict__ img, float* __restrict__ out,
             int nb, int nt, int H, int W, int ctot, int c0, int c1, long zsb, long zst) {
  const int cn = c1 - c0;
  const long total = (long)nb * nt * H * W * cn;
  const int secs = img ? 3 : 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i;
    const int c = (int)(r % cn); r /= cn;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int b = (int)(r % nb);
    const int t = (int)(r / nb);
    const long src = ((b * zsb + t * zst) * H * W + (long)h * W + w) * ctot + c0 + c;
    const float m = sigmoidf_(z[src]);
    const long row = (long)nb * W * cn;                       // floats per output row
    float* o = out + (((long)t * secs * H + h) * row) + ((long)b * W + w) * cn + c;
    if (img) {
      const float tr = (float)img[src] * (1.f / 255.f);
      o[0] = tr;
      o[(long)H * row] = m;
      o[2l * H * row] = (m - tr + 1.f) * 0.5f;
    } else {
      o[0] = m;
    }
  }
}

// wave per row: loss[row] = sum_d (pred - target)^2 ; dpred = coef*2*(pred-target)
__global__ void __launch_bounds__(256)
k_mse_loss(const float* __restrict__ pred, long ldp, const float* __restrict__ tgt, long ldt,
           float* __restrict__ loss, float* __restrict__ dpred, long lddp, int rows, int D,
           float coef) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    float e = pred[row * ldp + d] - tgt[row * ldt + d];
    acc += e * e;
    dpred[row * lddp + d] = coef * 2.f * e;
  }
  acc = wave_sum(acc);
  if (lane == 0) loss[row] = acc;
}

// kind 0: symlog MSE (pred - symlog(t))^2 ; kind 1: Bernoulli(logits) NLL.
__global__ void k_scalar_loss(const float* __restrict__ pred, const float* __restrict__ tgt,
                              float* __restrict__ loss, float* __restrict__ dpred, long n,
                              float coef, int kind) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p = pred[i], t = tgt[i];
  if (kind == 0) {
    float e = p - symlogf_(t);
    loss[i] = e * e;
    dpred[i] = coef * 2.f * e;
  } else {
    loss[i] = -(t * logsigmoidf_(p) + (1.f - t) * logsigmoidf_(-p));
    dpred[i] = coef * (sigmoidf_(p) - t);
  }
}

// action = tanh(o_mean) + ((hi-lo)*sigmoid(o_std)+lo) * eps   (eps may be null -> mode)
__global__ void k_normal_head_fwd(const float* __restrict__ om, long ldm,
                                  const float* __restrict__ os, long ldsd,
                                  const float* __restrict__ eps, long lde,
                                  float* __restrict__ act, long lda, int rows, int A,
                                  float lo, float hi) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * A) return;
  long r = i / A; int a = (int)(i - r * A);
  float mean = tanhf(om[r * ldm + a]);
  float v = mean;
  if (eps) {
    float std = (hi - lo) * sigmoidf_(os[r * ldsd + a]) + lo;
    v += std * eps[r * lde + a];
  }
  act[r * lda + a] = v;
}

// tfutils.action_noise (reference tfutils.py:85-93) on a policy's action rows, in place:
//   continuous: a <- clip(a + amount * eps, -1, 1)          (Normal(a, amount).sample())
//   discrete:   probs = amount / A + (1 - amount) * a (a one-hot), a <- one_hot(draw) with the
//               inverse-CDF rule of the latent sampler (idx = #{c < A-1 : cdf_c <= u * cdf_{A-1}},
//               sequential fp32 sum).  One thread per row (rows = number of environments).
__global__ void k_action_noise(float* __restrict__ act, long lda, const float* __restrict__ noise,
                               long ldn, int rows, int A, float amount, int discrete) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float* a = act + (long)r * lda;
  if (!discrete) {
    for (int j = 0; j < A; ++j) a[j] = fminf(fmaxf(a[j] + amount * noise[(long)r * ldn + j], -1.f), 1.f);
    return;
  }
  float tot = 0.f;
  for (int j = 0; j < A; ++j) tot += amount / (float)A + (1.f - amount) * a[j];
  const float thr = noise[(long)r * ldn] * tot;
  float cdf = 0.f;
  int idx = 0;
  for (int j = 0; j < A - 1; ++j) {
    cdf += amount / (float)A + (1.f - amount) * a[j];
    idx += cdf <= thr ? 1 : 0;
  }
  for (int j = 0; j < A; ++j) a[j] = j == idx ? 1.f : 0.f;
}

// Backward of the sampled action and of the (normalised) entropy bonus.
// rows_ent: rows [0, rows_ent) carry the entropy term weighted by w[row].
// d loss/d log(std_a) = -scale_a * w * ent_coef, ent_coef = 1/(count*(hi_ent-lo_ent)).
// ent_row[row] = w[row] * sum_a scale_a * (-ent_norm_a)  (weighted loss value of the bonus).
__global__ void __launch_bounds__(256)
k_normal_head_bwd(const float* __restrict__ om, long ldm, const float* __restrict__ os, long ldsd,
                  const float* __restrict__ eps, long lde, const float* __restrict__ dact, long ldda,
                  const float* __restrict__ w, const float* __restrict__ scale,
                  float* __restrict__ dom, long lddm, float* __restrict__ dos, long lddsd,
                  float* __restrict__ ent_row, int rows, int rows_ent, int A,
                  float lo, float hi, float ent_coef, float ent_lo, float ent_div) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float er = 0.f;
  for (int a = lane; a < A; a += 64) {
    float o = om[r * ldm + a], s = os[r * ldsd + a];
    float mean = tanhf(o);
    float sg = sigmoidf_(s);
    float std = (hi - lo) * sg + lo;
    float da = dact ? dact[r * ldda + a] : 0.f;
    float dstd = da * eps[r * lde + a];
    if (r < rows_ent) {
      float sc = scale[a];
      dstd += -sc * w[r] * ent_coef / std;
      er += sc * -((logf(std) - ent_lo) / ent_div);
    }
    dom[r * lddm + a] = da * (1.f - mean * mean);
    dos[r * lddsd + a] = dstd * (hi - lo) * sg * (1.f - sg);
  }
  er = wave_sum(er);
  if (lane == 0 && ent_row) ent_row[r] = (r < rows_ent) ? w[r] * er : 0.f;
}

// Per action dimension: sum and sum of squares (fp64) of the normalised entropy over
// the first `rows` rows.  Two deterministic stages: thread (a, l) of a block walks rows
// l, l+L, ... of the block's chunk (a row's A values are one coalesced segment), lanes are
// combined through LDS in fixed order into partial[block][2][A]; one small block then adds
// the partials in block order.
__global__ void __launch_bounds__(256)
k_actent_partial(const float* __restrict__ os, long ldsd, int rows, int A, int rows_per_block,
                 float lo, float hi, float ent_lo, float ent_div, double* __restrict__ partial) {
  extern __shared__ double shd[];  // [L][2][A]
  const int L = 256 / A;
  const int a = threadIdx.x % A, l = threadIdx.x / A;
  double s = 0.0, q = 0.0;
  if (l < L) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (long r = r0 + l; r < r1; r += L) {
      float std = (hi - lo) * sigmoidf_(os[r * ldsd + a]) + lo;
      float e = (logf(std) - ent_lo) / ent_div;
      s += e; q += (double)e * e;
    }
    shd[(l * 2 + 0) * A + a] = s;
    shd[(l * 2 + 1) * A + a] = q;
  }
  __syncthreads();
  if (threadIdx.x < 2 * A) {
    double t = 0.0;
    for (int i = 0; i < L; ++i) t += shd[i * 2 * A + threadIdx.x];
    partial[(long)blockIdx.x * 2 * A + threadIdx.x] = t;
  }
}

__global__ void k_actent_final(const double* __restrict__ partial, int P, int A, double* __restrict__ out) {
  const int j = threadIdx.x;
  if (j >= 2 * A) return;
  double t = 0.0;
  for (int p = 0; p < P; ++p) t += partial[(long)p * 2 * A + j];
  out[j] = t;
}

// Thread per imagined trajectory (column n), sequential over the horizon.
__global__ void k_imag_returns_fwd(const float* __restrict__ rew_raw, const float* __restrict__ val_raw,
                                   const float* __restrict__ cont_raw, const float* __restrict__ first_cont,
                                   float* __restrict__ reward, float* __restrict__ value,
                                   float* __restrict__ cont, float* __restrict__ weight,
                                   float* __restrict__ ret, int H, long N, float gamma, float lam,
                                   int gae) {
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float wprod = 1.f;
  for (int t = 0; t <= H; ++t) {
    float c = t == 0 ? first_cont[n] : sigmoidf_(cont_raw[t * N + n]);
    if (cont) cont[t * N + n] = c;
    wprod *= gamma * c;
    if (weight) weight[t * N + n] = wprod / gamma;
    value[t * N + n] = symexpf_(val_raw[t * N + n]);
    if (t > 0) reward[(t - 1) * N + n] = symexpf_(rew_raw[t * N + n]);
  }
  if (gae) {  // VFunction.target 'gae' (agent.py:428-433): the same return, summed as advantages
    float adv = 0.f;
    for (int t = H - 1; t >= 0; --t) {
      float c = sigmoidf_(cont_raw[(t + 1) * N + n]);
      float d = c * gamma;
      float r = symexpf_(rew_raw[(t + 1) * N + n]);
      float delta = r + d * value[(t + 1) * N + n] - value[t * N + n];
      adv = delta + d * lam * adv;
      ret[t * N + n] = adv + value[t * N + n];
    }
    return;
  }
  float R = value[H * N + n];
  for (int t = H - 1; t >= 0; --t) {
    float c = sigmoidf_(cont_raw[(t + 1) * N + n]);
    float d = c * gamma;
    float r = symexpf_(rew_raw[(t + 1) * N + n]);
    R = r + d * ((1.f - lam) * value[(t + 1) * N + n] + lam * R);
    ret[t * N + n] = R;
  }
}

// dret [H,N], dbase [H,N] (gradient w.r.t. value[:-1] used as baseline).
__global__ void k_imag_returns_bwd(const float* __restrict__ dret, const float* __restrict__ dbase,
                                   const float* __restrict__ rew_raw, const float* __restrict__ val_raw,
                                   const float* __restrict__ cont_raw, const float* __restrict__ value,
                                   const float* __restrict__ ret, float* __restrict__ d_rew_raw,
                                   float* __restrict__ d_val_raw, float* __restrict__ d_cont_raw,
                                   int H, long N, float gamma, float lam) {
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  d_rew_raw[n] = 0.f;
  d_cont_raw[n] = 0.f;
  float carry = 0.f;   // gradient flowing into R_t from R_{t-1}
  float dv = 0.f;      // gradient accumulated for value_t from step t-1
  for (int t = 0; t < H; ++t) {
    float g = dret[t * N + n] + carry;
    float c = sigmoidf_(cont_raw[(t + 1) * N + n]);
    float d = c * gamma;
    float vnext = value[(t + 1) * N + n];
    float Rnext = (t + 1 < H) ? ret[(t + 1) * N + n] : vnext;
    float A = (1.f - lam) * vnext + lam * Rnext;
    // value_t: baseline grad + what step t-1 pushed into it
    float dvt = dv + (dbase ? dbase[t * N + n] : 0.f);
    d_val_raw[t * N + n] = dvt * expf(fabsf(val_raw[t * N + n]));
    d_rew_raw[(t + 1) * N + n] = g * expf(fabsf(rew_raw[(t + 1) * N + n]));
    d_cont_raw[(t + 1) * N + n] = g * A * gamma * c * (1.f - c);
    dv = g * d * (1.f - lam);
    carry = g * d * lam;
  }
  d_val_raw[H * N + n] = (dv + carry) * expf(fabsf(val_raw[H * N + n]));
}

// critic: loss_i = w_i * (out_i - symlog(ret_i))^2 ; dout = coef * 2 w (out - symlog ret)
__global__ void k_critic_loss(const float* __restrict__ out, const float* __restrict__ ret,
                              const float* __restrict__ w, float* __restrict__ loss,
                              float* __restrict__ dout, long n, float coef) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float e = out[i] - symlogf_(ret[i]);
  loss[i] = w[i] * e * e;
  dout[i] = coef * 2.f * w[i] * e;
}

// actor: score = ((ret - base) * sc[0] - sc[1]) * sc[2]  (normaliser chain folded
// into three device scalars), loss_i = w * (-score + ent_row);
// dret = -w * coef * sc[0]*sc[2], dbase = -dret.
__global__ void k_actor_seed(const float* __restrict__ ret, const float* __restrict__ base,
                             const float* __restrict__ w, const float* __restrict__ ent_row,
                             const float* __restrict__ sc, float* __restrict__ loss,
                             float* __restrict__ dret, float* __restrict__ dbase, long n, float coef) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float score = ((ret[i] - base[i]) * sc[0] - sc[1]) * sc[2];
  loss[i] = w[i] * (-score + (ent_row ? ent_row[i] : 0.f));
  float g = -w[i] * coef * sc[0] * sc[2];
  dret[i] = g;
  dbase[i] = -g;
}

__global__ void k_sub(const float* __restrict__ a, const float* __restrict__ b,
                      float* __restrict__ o, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] - b[i];
}

// o = symexp(x): the mean of a SymlogDist head (tfutils.py:345-349), for the critic's own
// prediction on the imagined states (metrics imag_critic_mean / _std, agent.py:411-412)
__global__ void k_symexp(const float* __restrict__ x, float* __restrict__ o, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = symexpf_(x[i]);
}

// ---- one-hot (categorical) policy head, REINFORCE (agent.py:357-358, 372-377) ----
// wave per row; logit [rows, A] are normalised log-probs (unimix already applied).
// ent_out[row] = H(row) / ent_div.
__global__ void __launch_bounds__(256)
k_onehot_entropy(const float* __restrict__ logit, long ldl, float* __restrict__ ent_out,
                 int rows, int A, float ent_div) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float m = -INFINITY;
  for (int a = lane; a < A; a += 64) m = fmaxf(m, logit[r * ldl + a]);
  m = wave_max(m);
  float s = 0.f;
  for (int a = lane; a < A; a += 64) s += expf(logit[r * ldl + a] - m);
  const float lse = m + logf(wave_sum(s));
  float h = 0.f;
  for (int a = lane; a < A; a += 64) {
    float ll = logit[r * ldl + a] - lse;
    h -= expf(ll) * ll;
  }
  h = wave_sum(h);
  if (lane == 0) ent_out[r] = h / ent_div;
}

// rows < rows_grad: score = ((ret - base)*sc[0] - sc[1])*sc[2];
//   loss_pg[row]  = w * (-logp(action) * score)
//   loss_ent[row] = w * (scale * -H/ent_div)
//   dlogit_c = coef * w * ( -score*(onehot_c - p_c) + scale * p_c*(ll_c + H)/ent_div )
// rows >= rows_grad: dlogit = 0.
__global__ void __launch_bounds__(256)
k_onehot_policy_grad(const float* __restrict__ logit, long ldl, const float* __restrict__ action,
                     long lda, const float* __restrict__ ret, const float* __restrict__ base,
                     const float* __restrict__ w, const float* __restrict__ sc,
                     const float* __restrict__ scale, float* __restrict__ dlogit, long lddl,
                     float* __restrict__ loss_pg, float* __restrict__ loss_ent, int rows,
                     int rows_grad, int A, float coef, float ent_div) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  if (r >= rows_grad) {
    for (int a = lane; a < A; a += 64) dlogit[r * lddl + a] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int a = lane; a < A; a += 64) m = fmaxf(m, logit[r * ldl + a]);
  m = wave_max(m);
  float s = 0.f;
  for (int a = lane; a < A; a += 64) s += expf(logit[r * ldl + a] - m);
  const float lse = m + logf(wave_sum(s));
  float h = 0.f, lp = 0.f;
  for (int a = lane; a < A; a += 64) {
    float ll = logit[r * ldl + a] - lse;
    h -= expf(ll) * ll;
    lp += action[r * lda + a] * ll;
  }
  h = wave_sum(h);
  lp = wave_sum(lp);
  const float score = ((ret[r] - base[r]) * sc[0] - sc[1]) * sc[2];
  const float wr = w[r], es = scale[0];
  for (int a = lane; a < A; a += 64) {
    float ll = logit[r * ldl + a] - lse;
    float p = expf(ll);
    float g = -score * (action[r * lda + a] - p) + es * p * (ll + h) / ent_div;
    dlogit[r * lddl + a] = coef * wr * g;
  }
  if (lane == 0) {
    loss_pg[r] = wr * (-lp * score);
    loss_ent[r] = wr * (es * -(h / ent_div));
  }
}

inline int nblk(long n, int t = 256) { return (int)((n + t - 1) / t); }

}  // namespace

extern "C" int dd_image_loss(const float* z, const unsigned char* img, float* loss, float* dz,
                             int rows, long P, int ctot, int c0, int c1, float coef, void* stream) {
  if (rows <= 0) return 0;
  k_image_loss<<<rows, 256, 0, (hipStream_t)stream>>>(z, img, loss, dz, P, ctot, c0, c1, coef);
  DD_CHECK_LAUNCH("dd_image_loss");
  return 0;
}

extern "C" int dd_video_grid(const float* z, const unsigned char* img, float* out, int nb, int nt,
                             int H, int W, int ctot, int c0, int c1, long zsb, long zst, void* stream) {
  if (nb <= 0 || nt <= 0) return 0;
  DD_REQUIRE(0 <= c0 && c0 < c1 && c1 <= ctot && H >= 1 && W >= 1, "dd_video_grid: channel range");
  const long total = (long)nb * nt * H * W * (c1 - c0);
  k_video_grid<<<nblk(total > (1l << 22) ? (1l << 22) : total), 256, 0, (hipStream_t)stream>>>(
      z, img, out, nb, nt, H, W, ctot, c0, c1, zsb, zst);
  DD_CHECK_LAUNCH("dd_video_grid");
  return 0;
}

extern "C" int dd_mse_loss(const float* pred, long ldp, const float* tgt, long ldt, float* loss,
                           float* dpred, long lddp, int rows, int D, float coef, void* stream) {
  if (rows <= 0) return 0;
  k_mse_loss<<<nblk(rows, 4), 256, 0, (hipStream_t)stream>>>(pred, ldp, tgt, ldt, loss, dpred, lddp, rows, D, coef);
  DD_CHECK_LAUNCH("dd_mse_loss");
  return 0;
}

extern "C" int dd_scalar_loss(const float* pred, const float* tgt, float* loss, float* dpred,
                              long n, float coef, int kind, void* stream) {
  if (n <= 0) return 0;
  k_scalar_loss<<<nblk(n), 256, 0, (hipStream_t)stream>>>(pred, tgt, loss, dpred, n, coef, kind);
  DD_CHECK_LAUNCH("dd_scalar_loss");
  return 0;
}

extern "C" int dd_normal_head_fwd(const float* om, long ldm, const float* os, long ldsd,
                                  const float* eps, long lde, float* act, long lda,
                                  int rows, int A, float lo, float hi, void* stream) {
  if (rows <= 0) return 0;
  k_normal_head_fwd<<<nblk((long)rows * A), 256, 0, (hipStream_t)stream>>>(om, ldm, os, ldsd, eps, lde, act, lda, rows, A, lo, hi);
  DD_CHECK_LAUNCH("dd_normal_head_fwd");
  return 0;
}

extern "C" int dd_action_noise(float* act, long lda, const float* noise, long ldn, int rows, int A,
                               float amount, int discrete, void* stream) {
  if (rows <= 0 || amount == 0.f) return 0;
  DD_REQUIRE(noise != nullptr && A >= 1, "dd_action_noise: noise required");
  k_action_noise<<<(rows + 63) / 64, 64, 0, (hipStream_t)stream>>>(act, lda, noise, ldn, rows, A, amount, discrete);
  DD_CHECK_LAUNCH("dd_action_noise");
  return 0;
}

extern "C" int dd_normal_head_bwd(const float* om, long ldm, const float* os, long ldsd,
                                  const float* eps, long lde, const float* dact, long ldda,
                                  const float* w, const float* scale,
                                  float* dom, long lddm, float* dos, long lddsd, float* ent_row,
                                  int rows, int rows_ent, int A, float lo, float hi,
                                  float ent_coef, float ent_lo, float ent_div, void* stream) {
  if (rows <= 0) return 0;
  k_normal_head_bwd<<<nblk(rows, 4), 256, 0, (hipStream_t)stream>>>(
      om, ldm, os, ldsd, eps, lde, dact, ldda, w, scale, dom, lddm, dos, lddsd, ent_row,
      rows, rows_ent, A, lo, hi, ent_coef, ent_lo, ent_div);
  DD_CHECK_LAUNCH("dd_normal_head_bwd");
  return 0;
}

extern "C" int dd_actent_stats(const float* os, long ldsd, int rows, int A, float lo, float hi,
                               float ent_lo, float ent_div, double* out, double* ws,
                               size_t ws_bytes, void* stream) {
  DD_REQUIRE(A >= 1 && A <= 128, "dd_actent_stats: 1 <= A <= 128");
  const int L = 256 / A;
  int P = (rows + L * 16 - 1) / (L * 16);
  if (P > 256) P = 256;
  if (P < 1) P = 1;
  DD_REQUIRE(ws && (size_t)P * 2 * A * sizeof(double) <= ws_bytes, "dd_actent_stats: workspace too small");
  const int rpb = (rows + P - 1) / P;
  k_actent_partial<<<P, 256, (size_t)L * 2 * A * sizeof(double), (hipStream_t)stream>>>(
      os, ldsd, rows, A, rpb, lo, hi, ent_lo, ent_div, ws);
  DD_CHECK_LAUNCH("dd_actent_stats");
  k_actent_final<<<1, 2 * A <= 64 ? 64 : 256, 0, (hipStream_t)stream>>>(ws, P, A, out);
  DD_CHECK_LAUNCH("dd_actent_stats(final)");
  return 0;
}

extern "C" int dd_imag_returns_fwd(const float* rew_raw, const float* val_raw, const float* cont_raw,
                                   const float* first_cont, float* reward, float* value,
                                   float* cont, float* weight, float* ret, int H, long N,
                                   float gamma, float lam, int gae, void* stream) {
  if (N <= 0) return 0;
  k_imag_returns_fwd<<<nblk(N), 256, 0, (hipStream_t)stream>>>(rew_raw, val_raw, cont_raw, first_cont, reward, value, cont, weight, ret, H, N, gamma, lam, gae);
  DD_CHECK_LAUNCH("dd_imag_returns_fwd");
  return 0;
}

extern "C" int dd_imag_returns_bwd(const float* dret, const float* dbase, const float* rew_raw,
                                   const float* val_raw, const float* cont_raw, const float* value,
                                   const float* ret, float* d_rew_raw, float* d_val_raw,
                                   float* d_cont_raw, int H, long N, float gamma, float lam,
                                   void* stream) {
  if (N <= 0) return 0;
  k_imag_returns_bwd<<<nblk(N), 256, 0, (hipStream_t)stream>>>(dret, dbase, rew_raw, val_raw, cont_raw, value, ret, d_rew_raw, d_val_raw, d_cont_raw, H, N, gamma, lam);
  DD_CHECK_LAUNCH("dd_imag_returns_bwd");
  return 0;
}

extern "C" int dd_critic_loss(const float* out, const float* ret, const float* w, float* loss,
                              float* dout, long n, float coef, void* stream) {
  if (n <= 0) return 0;
  k_critic_loss<<<nblk(n), 256, 0, (hipStream_t)stream>>>(out, ret, w, loss, dout, n, coef);
  DD_CHECK_LAUNCH("dd_critic_loss");
  return 0;
}

extern "C" int dd_actor_seed(const float* ret, const float* base, const float* w,
                             const float* ent_row, const float* sc, float* loss, float* dret,
                             float* dbase, long n, float coef, void* stream) {
  if (n <= 0) return 0;
  k_actor_seed<<<nblk(n), 256, 0, (hipStream_t)stream>>>(ret, base, w, ent_row, sc, loss, dret, dbase, n, coef);
  DD_CHECK_LAUNCH("dd_actor_seed");
  return 0;
}

extern "C" int dd_sub(const float* a, const float* b, float* o, long n, void* stream) {
  if (n <= 0) return 0;
  k_sub<<<nblk(n), 256, 0, (hipStream_t)stream>>>(a, b, o, n);
  DD_CHECK_LAUNCH("dd_sub");
  return 0;
}

extern "C" int dd_symexp(const float* x, float* o, long n, void* stream) {
  if (n <= 0) return 0;
  k_symexp<<<nblk(n), 256, 0, (hipStream_t)stream>>>(x, o, n);
  DD_CHECK_LAUNCH("dd_symexp");
  return 0;
}

extern "C" int dd_onehot_entropy(const float* logit, long ldl, float* ent_out, int rows, int A,
                                 float ent_div, void* stream) {
  if (rows <= 0) return 0;
  k_onehot_entropy<<<nblk(rows, 4), 256, 0, (hipStream_t)stream>>>(logit, ldl, ent_out, rows, A, ent_div);
  DD_CHECK_LAUNCH("dd_onehot_entropy");
  return 0;
}

extern "C" int dd_onehot_policy_grad(const float* logit, long ldl, const float* action, long lda,
                                     const float* ret, const float* base, const float* w,
                                     const float* sc, const float* scale, float* dlogit, long lddl,
                                     float* loss_pg, float* loss_ent, int rows, int rows_grad,
                                     int A, float coef, float ent_div, void* stream) {
  if (rows <= 0) return 0;
  k_onehot_policy_grad<<<nblk(rows, 4), 256, 0, (hipStream_t)stream>>>(
      logit, ldl, action, lda, ret, base, w, sc, scale, dlogit, lddl, loss_pg, loss_ent, rows,
      rows_grad, A, coef, ent_div);
  DD_CHECK_LAUNCH("dd_onehot_policy_grad");
  return 0;
}
