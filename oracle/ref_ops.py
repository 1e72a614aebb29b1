"""CPU ORACLE for the individual kernels (test infrastructure, not product
code).  `RefOps` restates, in plain PyTorch, the contract of every entry point
of include/daydreamer_hip.h with the same method signatures as
`daydreamer_amd.hipops.HipOps`.  Uses:
  * `tests/` (-m gpu): per-kernel parity, HIP vs this file on the same inputs;
  * `tests/` (not gpu): drive the learner's host logic (manual backward,
    buffer layout) on CPU against the autograd oracle `oracle/dreamer_ref.py`.
The product never imports this module.
"""

import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-3


def _symlog(x):
  return torch.sign(x) * torch.log1p(torch.abs(x))


def _symexp(x):
  return torch.sign(x) * torch.expm1(torch.abs(x))


def philox4x32(c, k0, k1):
  """Philox4x32-10 on uint64-held 32-bit lanes; c: [n,4] uint64 array."""
  M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
  W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
  mask = np.uint64(0xFFFFFFFF)
  c = c.copy()
  k0 = np.uint64(k0)
  k1 = np.uint64(k1)
  for _ in range(10):
    p0 = M0 * c[:, 0]
    p1 = M1 * c[:, 2]
    hi0, lo0 = p0 >> np.uint64(32), p0 & mask
    hi1, lo1 = p1 >> np.uint64(32), p1 & mask
    n0 = hi1 ^ c[:, 1] ^ k0
    n2 = hi0 ^ c[:, 3] ^ k1
    c = np.stack([n0, lo1, n2, lo0], 1)
    k0 = (k0 + W0) & mask
    k1 = (k1 + W1) & mask
  return c


def philox_field(outer, inner, cols, inner_global, inner_offset, seed, step,
                 site, kind):
  """Noise tensor [outer, inner, cols] exactly as dd_philox defines it."""
  cb = (cols + 3) // 4
  o, i, b = np.meshgrid(np.arange(outer), np.arange(inner), np.arange(cb),
                        indexing='ij')
  grow = (o * inner_global + inner_offset + i).reshape(-1)
  ctr = np.stack([b.reshape(-1), grow, np.full(grow.shape, site),
                  np.full(grow.shape, step & 0xFFFFFFFF)], 1).astype(np.uint64)
  ctr &= np.uint64(0xFFFFFFFF)
  x = philox4x32(ctr, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
  if kind == 0:
    v = (x >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
  else:
    u1 = ((x[:, 0::2] >> np.uint64(8)).astype(np.float32) + np.float32(0.5)
          ) * np.float32(1.0 / 16777216.0)
    u2 = (x[:, 1::2] >> np.uint64(8)).astype(np.float32) * np.float32(
        1.0 / 16777216.0)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1))
    ang = np.float32(6.283185307179586) * u2
    v = np.empty(x.shape, np.float32)
    v[:, 0::2] = rad * np.cos(ang)
    v[:, 1::2] = rad * np.sin(ang)
  v = v.reshape(outer, inner, cb * 4)[:, :, :cols]
  return np.ascontiguousarray(v)


def exp_det(x):
  """csrc/sampler_core.h dd_exp_det restated in numpy float32 (every operation rounds to
  float32 exactly as the device / host code does: no fused multiply-add anywhere)."""
  f = np.float32
  x = np.asarray(x, np.float32)
  with np.errstate(invalid='ignore', over='ignore'):
    live = x >= f(-86.0)
    xs = np.where(live, x, f(0.0)).astype(np.float32)
    fn = np.floor(f(1.44269504088896341) * xs + f(0.5)).astype(np.float32)
    r = (xs - fn * f(0.693359375)).astype(np.float32)
    r = (r - fn * f(-2.12194440e-4)).astype(np.float32)
    z = r * r
    p = np.full_like(r, f(1.9875691500e-4))
    for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1,
              5.0000001201e-1):
      p = p * r + f(c)
    y = (p * z + r) + f(1.0)
    scale = ((fn.astype(np.int64) + 127).astype(np.uint32) << np.uint32(23)).view(np.float32)
    out = (y * scale).astype(np.float32)
  return np.where(live, out, f(0.0)).astype(np.float32)


def sample_twin_np(x, u, G, C, unimix, mode=0):
  """The categorical draw of k_stats_fwd / dd_onehot_sample_host restated in numpy float32,
  operation for operation (deterministic exp, the LW-lane butterfly sum, the Kogge-Stone
  scan): x [rows, G*C], u [rows, G] -> (index int64 [rows, G], mixed probs [rows, G, C]).
  Replaces tf.random.categorical (reference tfutils.py:374)."""
  f = np.float32
  x = np.asarray(x, np.float32)
  rows = x.shape[0]
  xv = x.reshape(rows, G, C)
  LW = 8
  while LW < C:
    LW *= 2
  m = xv.max(-1, keepdims=True)
  e = np.zeros((rows, G, LW), np.float32)
  e[..., :C] = exp_det(xv - m)
  t = e.copy()
  lanes = np.arange(LW)
  o = LW // 2
  while o > 0:
    t = (t + t[..., lanes ^ o]).astype(np.float32)
    o //= 2
  s = t[..., :1]
  pm = np.zeros_like(e)
  pm[..., :C] = (f(1.0) - f(unimix)) * (e[..., :C] / s) + f(unimix) / f(C)
  pm = pm.astype(np.float32)
  if mode == 1:
    return pm[..., :C].argmax(-1), pm[..., :C]
  cdf = pm.copy()
  o = 1
  while o < LW:
    n = cdf.copy()
    n[..., o:] = cdf[..., o:] + cdf[..., :-o]
    cdf = n.astype(np.float32)
    o *= 2
  thr = (np.asarray(u, np.float32) * cdf[..., C - 1])[..., None]
  idx = (cdf[..., :C - 1] <= thr).sum(-1)
  return idx, pm[..., :C]


class RefOps:

  name = 'ref'

  def __init__(self, device='cpu', dtype=torch.float32):
    self.device = torch.device(device)

  # ---- contractions ---------------------------------------------------------

  def gemm(self, A, B, C, ta=False, tb=False, alpha=1.0, beta=0.0, bias=None, defer=False):
    # (defer: the HIP path may leave split-K partial sums for the consumer; here C is
    # always complete and None is returned, so consumers get pre=None)
    a = A.t() if ta else A
    b = B.t() if tb else B
    r = alpha * (a @ b)
    if bias is not None:
      r = r + bias
    if beta != 0.0:
      r = r + beta * C
    C.copy_(r)

  @staticmethod
  def _big(big, in_scale, dtype):
    if big.dtype == torch.uint8:
      return big.to(dtype) * torch.tensor(in_scale, dtype=dtype)
    return big

  def conv_down(self, big, w, bias, small, k, in_scale=1.0):
    x = self._big(big, in_scale, w.dtype).permute(0, 3, 1, 2)
    y = F.conv2d(x, w.permute(3, 2, 0, 1), stride=2).permute(0, 2, 3, 1)
    hs, ws = small.shape[1:3]
    y = y[:, :hs, :ws]
    small.copy_(y + bias if bias is not None else y)

  def conv_up(self, small, w, bias, big, k):
    y = F.conv_transpose2d(small.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1),
                           stride=2).permute(0, 2, 3, 1)
    hb, wb = big.shape[1:3]
    out = torch.zeros_like(big)
    out[:, :y.shape[1], :y.shape[2]] = y[:, :hb, :wb]
    big.copy_(out + bias if bias is not None else out)

  def conv_wgrad(self, big, small, dw, k, in_scale=1.0, beta=0.0):
    x = self._big(big, in_scale, small.dtype)
    n, hs, ws, cs = small.shape
    g = torch.zeros_like(dw)
    for ky in range(k):
      for kx in range(k):
        patch = x[:, ky:ky + 2 * hs:2, kx:kx + 2 * ws:2, :]
        g[ky, kx] = torch.einsum('nhwb,nhws->bs', patch, small)
    dw.copy_(g + beta * dw if beta != 0.0 else g)

  # stride-1 SAME convolutions (odd k) + 2x2 pooling / repetition: include/daydreamer_hip.h
  # dd_conv2d_same*, dd_pool2, dd_repeat2 (reference nets.py:330-391)

  def conv_same(self, x, w, bias, y, k, in_scale=1.0, alpha=1.0, beta=0.0):
    xx = self._big(x, in_scale, w.dtype).permute(0, 3, 1, 2)
    r = alpha * F.conv2d(xx, w.permute(3, 2, 0, 1), padding=k // 2).permute(0, 2, 3, 1)
    if bias is not None:
      r = r + bias
    y.copy_(r + beta * y if beta != 0.0 else r)

  def conv_same_bwd(self, dy, w, dx, k, alpha=1.0, beta=0.0):
    # dx[n,y,x,ci] = sum_{ky,kx,co} dy[n, y-ky+p, x-kx+p, co] * w[ky,kx,ci,co]
    n, h, wd, cout = dy.shape
    p = k // 2
    dyp = F.pad(dy, (0, 0, p, p, p, p))
    r = torch.zeros_like(dx)
    for ky in range(k):
      for kx in range(k):
        sl = dyp[:, 2 * p - ky:2 * p - ky + h, 2 * p - kx:2 * p - kx + wd, :]
        r += torch.einsum('nhwo,io->nhwi', sl, w[ky, kx])
    r = alpha * r
    dx.copy_(r + beta * dx if beta != 0.0 else r)

  def conv_same_wgrad(self, x, dy, dw, k, in_scale=1.0, alpha=1.0, beta=0.0):
    xx = self._big(x, in_scale, dy.dtype)
    n, h, wd, cin = xx.shape
    p = k // 2
    xp = F.pad(xx, (0, 0, p, p, p, p))
    g = torch.zeros_like(dw)
    for ky in range(k):
      for kx in range(k):
        g[ky, kx] = torch.einsum('nhwi,nhwo->io', xp[:, ky:ky + h, kx:kx + wd, :], dy)
    g = alpha * g
    dw.copy_(g + beta * dw if beta != 0.0 else g)

  def pool2(self, x, y, scale=0.25):
    y.copy_(((x[:, 0::2, 0::2] + x[:, 0::2, 1::2]) + (x[:, 1::2, 0::2] + x[:, 1::2, 1::2])) * scale)

  def repeat2(self, x, y, scale=1.0, beta=0.0):
    r = (x * scale).repeat_interleave(2, 1).repeat_interleave(2, 2)
    y.copy_(r + beta * y if beta != 0.0 else r)

  # ---- LayerNorm / GRU ----------------------------------------------------------

  def ln_act_fwd(self, z, gamma, beta, out, stats, act=True, pre=None, head=None):
    assert pre is None
    mean = z.mean(-1, keepdim=True)
    var = ((z - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + LN_EPS)
    y = (z - mean) * rstd * gamma + beta
    out.copy_(F.elu(y) if act else y)
    stats[:, 0] = mean[:, 0]
    stats[:, 1] = rstd[:, 0]
    if head is not None:   # dd_ln_act_fwd_head: the one-unit output layer behind this layer
      w, b, yh = head
      yh.copy_(out @ w.reshape(-1, 1) + (b if b is not None else 0.0))

  def ln_act_bwd_head(self, head_dy, head_w, z, out, stats, gamma, dz, dgamma=None, dbeta=None,
                      accumulate=False, act=True, dbias_pre=None, beta=None):
    """dd_ln_act_bwd_head: the layer's output gradient is head_dy x head_w."""
    dout = head_dy.reshape(-1, 1) * head_w.reshape(1, -1)
    self.ln_act_bwd(dout, z, out, stats, gamma, dz, dgamma, dbeta, accumulate, act, dbias_pre, beta=beta)

  def _ln_dy(self, dout, out, act):
    if act:
      return dout * torch.where(out > 0, torch.ones_like(out), out + 1.0)
    return dout

  def ln_act_bwd(self, dout, z, out, stats, gamma, dz, dgamma=None,
                 dbeta=None, accumulate=False, act=True, dbias_pre=None, pre=None, beta=None):
    assert pre is None   # (beta: the device recomputes `out` from z; the value is the same)
    mean, rstd = stats[:, :1], stats[:, 1:2]
    dy = self._ln_dy(dout, out, act)
    xh = (z - mean) * rstd
    g = dy * gamma
    s1 = g.mean(-1, keepdim=True)
    s2 = (g * xh).mean(-1, keepdim=True)
    res = rstd * (g - s1 - xh * s2)
    if dgamma is not None:
      dg, db = (dy * xh).sum(0), dy.sum(0)
      dgamma.copy_(dgamma + dg if accumulate else dg)
      dbeta.copy_(dbeta + db if accumulate else db)
      if dbias_pre is not None:
        dp = res.sum(0)
        dbias_pre.copy_(dbias_pre + dp if accumulate else dp)
    dz.copy_(res)

  def ln_param_grad(self, dout, z, out, stats, dgamma, dbeta,
                    accumulate=False, act=True):
    mean, rstd = stats[:, :1], stats[:, 1:2]
    dy = self._ln_dy(dout, out, act)
    xh = (z - mean) * rstd
    dg, db = (dy * xh).sum(0), dy.sum(0)
    dgamma.copy_(dgamma + dg if accumulate else dg)
    dbeta.copy_(dbeta + db if accumulate else db)

  def col_sum(self, x, out, beta=0.0):
    r = x.sum(0)
    out.copy_(beta * out + r if beta != 0.0 else r)

  def gru_fwd(self, z3, gamma, beta, h, hn, stats, pre=None):
    assert pre is None
    D = h.shape[1]
    mean = z3.mean(-1, keepdim=True)
    var = ((z3 - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + LN_EPS)
    y = (z3 - mean) * rstd * gamma + beta
    r = torch.sigmoid(y[:, :D])
    c = torch.tanh(r * y[:, D:2 * D])
    u = torch.sigmoid(y[:, 2 * D:] - 1)
    hn.copy_(u * c + (1 - u) * h)
    stats[:, 0] = mean[:, 0]
    stats[:, 1] = rstd[:, 0]

  def gru_bwd(self, dhn, z3, stats, gamma, beta, h, dz3, dh, dy3, zero=None):
    D = h.shape[1]
    mean, rstd = stats[:, :1], stats[:, 1:2]
    xh = (z3 - mean) * rstd
    y = xh * gamma + beta
    r = torch.sigmoid(y[:, :D])
    yc = y[:, D:2 * D]
    c = torch.tanh(r * yc)
    u = torch.sigmoid(y[:, 2 * D:] - 1)
    du = dhn * (c - h)
    dc = dhn * u
    dhv = dhn * (1 - u)
    dpre = dc * (1 - c * c)
    dy = torch.cat([dpre * yc * r * (1 - r), dpre * r, du * u * (1 - u)], 1)
    g = dy * gamma
    s1 = g.mean(-1, keepdim=True)
    s2 = (g * xh).mean(-1, keepdim=True)
    res = rstd * (g - s1 - xh * s2)
    dh.copy_(dhv)
    if zero is not None:
      zero.zero_()
    dy3.copy_(dy)
    dz3.copy_(res)

  # ---- categorical latent ----------------------------------------------------------

  @staticmethod
  def _kogge_stone(p):
    """Inclusive scan over the last axis in the device's summation order
    (LW = next power of two >= C lanes, Kogge-Stone)."""
    C = p.shape[-1]
    LW = 8
    while LW < C:
      LW *= 2
    x = torch.zeros(p.shape[:-1] + (LW,), dtype=p.dtype)
    x[..., :C] = p
    o = 1
    while o < LW:
      y = x.clone()
      y[..., o:] = x[..., o:] + x[..., :-o]
      x = y
      o *= 2
    return x[..., :C]

  def stats_fwd(self, x, u, logit, stoch, G, C, unimix, mode=0, pre=None):
    assert pre is None
    rows = x.shape[0]
    if x.dtype == torch.float32:
      # float32: the device's own arithmetic (bit-exact draw), see sample_twin_np
      idx, pm = sample_twin_np(x.detach().numpy(), None if u is None else u.detach().numpy(),
                               G, C, unimix, mode)
      pm = torch.from_numpy(pm)
      if unimix > 0:
        lg = torch.log(pm)
      else:
        xv = x.reshape(rows, G, C)
        m = xv.max(-1, keepdim=True).values
        lg = (xv - m) - torch.log(torch.from_numpy(exp_det((xv - m).numpy())).sum(-1, keepdim=True))
      logit.copy_(lg.reshape(rows, G * C))
      stoch.copy_(F.one_hot(torch.from_numpy(idx), C).to(x.dtype).reshape(rows, G * C))
      return
    xv = x.reshape(rows, G, C)
    m = xv.max(-1, keepdim=True).values
    e = torch.exp(xv - m)
    s = e.sum(-1, keepdim=True)
    p = e / s
    pm = (1 - unimix) * p + unimix / C
    lg = torch.log(pm) if unimix > 0 else (xv - m) - torch.log(s)
    if mode == 1:
      idx = torch.argmax(pm, -1)
    else:
      cdf = self._kogge_stone(pm)
      thr = (u * cdf[..., -1])[..., None]
      idx = (cdf[..., :-1] <= thr).sum(-1)
    logit.copy_(lg.reshape(rows, G * C))
    stoch.copy_(F.one_hot(idx, C).to(x.dtype).reshape(rows, G * C))

  def stats_bwd(self, x, dlogit, dstoch, dx, G, C, unimix):
    rows = x.shape[0]
    p = torch.softmax(x.reshape(rows, G, C), -1)
    pm = (1 - unimix) * p + unimix / C
    dpm = torch.zeros_like(p)
    if dstoch is not None:
      dpm = dpm + dstoch.reshape(rows, G, C)
    if dlogit is not None:
      dpm = dpm + dlogit.reshape(rows, G, C) / pm
    dp = (1 - unimix) * dpm
    dot = (dp * p).sum(-1, keepdim=True)
    dx.copy_((p * (dp - dot)).reshape(rows, G * C))

  def kl_fwd(self, post, prior, kl, ent_post, ent_prior, G, C):
    rows = post.shape[0]
    la = torch.log_softmax(post.reshape(rows, G, C), -1)
    lb = torch.log_softmax(prior.reshape(rows, G, C), -1)
    pa, pb = torch.exp(la), torch.exp(lb)
    kl.copy_((pa * (la - lb)).sum((-1, -2)))
    ent_post.copy_(-(pa * la).sum((-1, -2)))
    ent_prior.copy_(-(pb * lb).sum((-1, -2)))

  def kl_bwd(self, post, prior, coef_dev, coef_host, balance, dpost, dprior,
             G, C):
    rows = post.shape[0]
    coef = coef_host * (float(coef_dev.reshape(-1)[0]) if coef_dev is not None
                        else 1.0)
    la = torch.log_softmax(post.reshape(rows, G, C), -1)
    lb = torch.log_softmax(prior.reshape(rows, G, C), -1)
    pa, pb = torch.exp(la), torch.exp(lb)
    klg = (pa * (la - lb)).sum(-1, keepdim=True)
    dpost.copy_((coef * (1 - balance) * pa * ((la - lb) - klg)
                 ).reshape(rows, G * C))
    dprior.copy_((coef * balance * (pb - pa)).reshape(rows, G * C))

  # ---- losses / imagination scalars ---------------------------------------------

  def image_loss(self, z, img, loss, dz, coef, c0=0, c1=None):
    c1 = z.shape[-1] if c1 is None else c1
    zz = z[..., c0:c1]
    rows = z.shape[0]
    s = torch.sigmoid(zz)
    d = s - img[..., c0:c1].to(z.dtype) * torch.tensor(1.0 / 255.0, dtype=z.dtype)
    loss.copy_((d * d).reshape(rows, -1).sum(-1))
    dz[..., c0:c1] = coef * 2 * d * s * (1 - s)

  def video_grid(self, z, img, out, nb, nt, c0, c1, zsb, zst):
    """dd_video_grid: tfutils.video_grid (tfutils.py:390-392) of [truth | model | error]
    (agent.py:276-281) or of the model alone (behaviors.py:44-45)."""
    H, W, ctot = z.shape[-3:]
    idx = (torch.arange(nb)[:, None] * zsb + torch.arange(nt)[None, :] * zst).reshape(-1)
    m = torch.sigmoid(z.reshape(-1, H, W, ctot)[idx][..., c0:c1]).reshape(nb, nt, H, W, c1 - c0)
    secs = [m]
    if img is not None:
      tr = img.reshape(-1, H, W, ctot)[idx][..., c0:c1].reshape(nb, nt, H, W, c1 - c0).to(z.dtype) / 255.0
      secs = [tr, m, (m - tr + 1) / 2]
    video = torch.cat(secs, 2)                                  # [nb, nt, secs * H, W, c]
    out.copy_(video.permute(1, 2, 0, 3, 4).reshape(out.shape))

  def mse_loss(self, pred, tgt, loss, dpred, coef):
    e = pred - tgt
    loss.copy_((e * e).sum(-1))
    dpred.copy_(coef * 2 * e)

  def scalar_loss(self, pred, tgt, loss, dpred, coef, kind):
    if kind == 0:
      e = pred - _symlog(tgt)
      loss.copy_(e * e)
      dpred.copy_(coef * 2 * e)
    else:
      loss.copy_(-(tgt * F.logsigmoid(pred) + (1 - tgt) * F.logsigmoid(-pred)))
      dpred.copy_(coef * (torch.sigmoid(pred) - tgt))

  def normal_head_fwd(self, om, os, eps, act, lo, hi):
    v = torch.tanh(om)
    if eps is not None:
      v = v + ((hi - lo) * torch.sigmoid(os) + lo) * eps
    act.copy_(v)

  def action_noise(self, act, noise, amount, discrete):
    """tfutils.action_noise, reference tfutils.py:85-93 (in place)."""
    if amount == 0:
      return
    if not discrete:
      act.copy_(torch.clamp(act + amount * noise, -1.0, 1.0))
      return
    A = act.shape[1]
    probs = amount / A + (1 - amount) * act
    cdf = torch.zeros_like(probs)
    run = torch.zeros_like(probs[:, 0])
    for j in range(A):  # the kernel's sequential fp32 sum
      run = run + probs[:, j]
      cdf[:, j] = run
    thr = (noise[:, 0] * cdf[:, -1])[:, None]
    idx = (cdf[:, :-1] <= thr).sum(-1)
    act.copy_(F.one_hot(idx, A).to(act.dtype))

  def normal_head_bwd(self, om, os, eps, dact, w, scale, dom, dos, ent_row,
                      rows_ent, lo, hi, ent_coef, ent_lo, ent_div):
    rows = om.shape[0]
    mean = torch.tanh(om)
    sg = torch.sigmoid(os)
    std = (hi - lo) * sg + lo
    da = dact if dact is not None else torch.zeros_like(om)
    dstd = da * eps
    live = (torch.arange(rows, device=om.device) < rows_ent).to(om.dtype)
    wv = torch.zeros(rows, dtype=om.dtype, device=om.device)
    wv[:rows_ent] = w.reshape(-1)[:rows_ent]
    dstd = dstd + (-scale[None] * (wv * live)[:, None] * ent_coef / std)
    er = (scale[None] * -((torch.log(std) - ent_lo) / ent_div)).sum(-1) * live * wv
    dom.copy_(da * (1 - mean * mean))
    dos.copy_(dstd * (hi - lo) * sg * (1 - sg))
    if ent_row is not None:
      ent_row.copy_(er)

  def actent_stats(self, os, rows, lo, hi, ent_lo, ent_div, out):
    A = os.shape[1]
    std = (hi - lo) * torch.sigmoid(os[:rows]) + lo
    e = ((torch.log(std) - ent_lo) / ent_div)
    out[:A] = e.double().sum(0)
    out[A:2 * A] = (e.double() ** 2).sum(0)

  def imag_returns_fwd(self, rew_raw, val_raw, cont_raw, first_cont, reward,
                       value, cont, weight, ret, H, N, gamma, lam, impl='gve'):
    rr = rew_raw.reshape(H + 1, N)
    vr = val_raw.reshape(H + 1, N)
    cr = cont_raw.reshape(H + 1, N)
    c = torch.cat([first_cont.reshape(1, N), torch.sigmoid(cr[1:])], 0)
    if cont is not None:
      cont.copy_(c.reshape(cont.shape))
    if weight is not None:
      weight.copy_((torch.cumprod(gamma * c, 0) / gamma).reshape(weight.shape))
    v = _symexp(vr)
    r = _symexp(rr[1:])
    value.copy_(v.reshape(value.shape))
    reward.copy_(r.reshape(reward.shape))
    d = c[1:] * gamma
    outs = []
    if impl == 'gae':  # agent.py:428-433
      adv = torch.zeros_like(v[0])
      for t in reversed(range(H)):
        adv = (r[t] + d[t] * v[t + 1] - v[t]) + d[t] * lam * adv
        outs.append(adv + v[t])
    else:
      R = v[H]
      for t in reversed(range(H)):
        R = r[t] + d[t] * ((1 - lam) * v[t + 1] + lam * R)
        outs.append(R)
    ret.copy_(torch.stack(list(reversed(outs)), 0).reshape(ret.shape))

  def imag_returns_bwd(self, dret, dbase, rew_raw, val_raw, cont_raw, value,
                       ret, d_rew_raw, d_val_raw, d_cont_raw, H, N, gamma, lam):
    # Autograd restatement of the forward recurrence.
    rr = rew_raw.reshape(H + 1, N).detach().clone().requires_grad_(True)
    vr = val_raw.reshape(H + 1, N).detach().clone().requires_grad_(True)
    cr = cont_raw.reshape(H + 1, N).detach().clone().requires_grad_(True)
    v = _symexp(vr)
    r = _symexp(rr[1:])
    d = torch.sigmoid(cr[1:]) * gamma
    R = v[H]
    outs = []
    for t in reversed(range(H)):
      R = r[t] + d[t] * ((1 - lam) * v[t + 1] + lam * R)
      outs.append(R)
    rets = torch.stack(list(reversed(outs)), 0)
    obj = (rets * dret.reshape(H, N)).sum()
    if dbase is not None:
      obj = obj + (v[:-1] * dbase.reshape(H, N)).sum()
    g = torch.autograd.grad(obj, [rr, vr, cr])
    d_rew_raw.copy_(g[0].reshape(d_rew_raw.shape))
    d_val_raw.copy_(g[1].reshape(d_val_raw.shape))
    d_cont_raw.copy_(g[2].reshape(d_cont_raw.shape))

  def critic_loss(self, out, ret, w, loss, dout, coef):
    e = out.reshape(-1) - _symlog(ret.reshape(-1))
    wv = w.reshape(-1)[:e.numel()]
    loss.copy_((wv * e * e).reshape(loss.shape))
    dout.copy_((coef * 2 * wv * e).reshape(dout.shape))

  def actor_seed(self, ret, base, w, ent_row, sc, loss, dret, dbase, coef):
    n = ret.numel()
    wv = w.reshape(-1)[:n]
    score = ((ret.reshape(-1) - base.reshape(-1)[:n]) * sc[0] - sc[1]) * sc[2]
    er = ent_row.reshape(-1)[:n] if ent_row is not None else 0.0
    loss.copy_((wv * (-score + er)).reshape(loss.shape))
    g = -wv * coef * sc[0] * sc[2]
    dret.copy_(g.reshape(dret.shape))
    dbase.copy_((-g).reshape(dbase.shape))

  def onehot_entropy(self, logit, ent_out, ent_div):
    ll = torch.log_softmax(logit, -1)
    ent_out.copy_(-(torch.exp(ll) * ll).sum(-1) / ent_div)

  def onehot_policy_grad(self, logit, action, ret, base, w, sc, scale, dlogit,
                         loss_pg, loss_ent, rows_grad, coef, ent_div):
    n = rows_grad
    ll = torch.log_softmax(logit[:n], -1)
    p = torch.exp(ll)
    h = -(p * ll).sum(-1, keepdim=True)
    lp = (action[:n] * ll).sum(-1)
    score = ((ret.reshape(-1)[:n] - base.reshape(-1)[:n]) * sc[0] - sc[1]) * sc[2]
    wv = w.reshape(-1)[:n]
    es = scale.reshape(-1)[0]
    g = -score[:, None] * (action[:n] - p) + es * p * (ll + h) / ent_div
    out = torch.zeros_like(logit)
    out[:n] = coef * wv[:, None] * g
    dlogit.copy_(out)
    loss_pg[:n] = wv * (-lp * score)
    loss_ent[:n] = wv * (es * -(h[:, 0] / ent_div))

  def symexp(self, x, o):
    o.copy_(_symexp(x.reshape(-1)).reshape(o.shape))

  def sub(self, a, b, o):
    o.copy_((a.reshape(-1)[:o.numel()] - b.reshape(-1)[:o.numel()]
             ).reshape(o.shape))

  # ---- learner state -----------------------------------------------------------------

  def philox(self, out, outer, inner, cols, inner_global, inner_offset, seed,
             step_dev, site, kind):
    v = philox_field(outer, inner, cols, inner_global, inner_offset, seed,
                     int(step_dev.reshape(-1)[0]), site, kind)
    out.copy_(torch.from_numpy(v).reshape(out.shape))

  def counter_add(self, counter, v=1):
    counter += v

  def reduce_stats(self, x, sums, maxs):
    xd = x.double()
    sums[0] = xd.sum()
    sums[1] = (xd * xd).sum()
    sums[2] = xd.abs().sum()
    maxs[0] = x.max()
    maxs[1] = (-x).max()
    maxs[2] = x.abs().max()

  def reduce_stats_multi(self, items):
    for x, sums, maxs in items:
      self.reduce_stats(x, sums, maxs)

  def autoadapt_update(self, scale, sums, count, target, thres, vel, lo, hi,
                       inverse, impl='mult'):
    n = scale.numel()
    avg = (sums[:n] / count).float()
    if impl == 'prop':  # tfutils.py:475-480
      d = avg - np.float32(target)
      d = -d if inverse else d
      scale.copy_(torch.clamp(scale.float() + np.float32(vel) * d, np.float32(lo), np.float32(hi)))
      return
    below = avg < np.float32(1.0 / (1.0 + thres)) * np.float32(target)
    above = avg > np.float32(1.0 + thres) * np.float32(target)
    if inverse:
      below, above = above, below
    s32 = scale.float()  # the controller state is float32 (tfutils.py:430)
    adj = torch.where(above, s32 * np.float32(1 + vel),
                      torch.where(below, s32 / np.float32(1 + vel), s32))
    scale.copy_(torch.clamp(adj, np.float32(lo), np.float32(hi)))

  def normalize_update(self, state, sums, count, in_scale_dev, decay, maxv,
                       impl, do_update, out):
    a = float(in_scale_dev.reshape(-1)[0]) if in_scale_dev is not None else 1.0
    if do_update:
      mean = a * float(sums[0]) / count
      sq = a * a * float(sums[1]) / count
      state[2] += 1.0
      state[0] = decay * float(state[0]) + (1 - decay) * mean
      state[1] = decay * float(state[1]) + (1 - decay) * sq
    corr = 1.0 - decay ** float(state[2])
    mean = float(state[0]) / corr
    var = float(state[1]) / corr - mean * mean
    if maxv > 0:
      scale = 1.0 / math.sqrt(max(var, 1.0 / maxv ** 2))
    else:
      scale = 1.0 / math.sqrt(var)
    off, sc = 0.0, 1.0
    if impl == 1:
      off, sc = mean, scale
    elif impl == 2:
      sc = scale
    out[0] = off
    out[1] = sc

  def scalar_mul(self, dst, a, b, c):
    dst.copy_(a * (b if b is not None else 1.0) * c)

  def axpy(self, x, alpha, alpha_dev, y, accumulate=True):
    a = alpha * (alpha_dev.reshape(-1)[0] if alpha_dev is not None else 1.0)
    y.copy_((y if accumulate else 0.0) + a * x)

  def balance_stats(self, out, target, loss, thres, kind, out7):
    m = _symexp(out) if kind == 0 else torch.sigmoid(out)
    pos = (target > thres).to(loss.dtype)
    pr = (m > thres).to(loss.dtype)
    vals = [loss * pos, loss * (1 - pos), pr * pos, (1 - pr) * (1 - pos), pos, target, m]
    out7.copy_(torch.stack([v.double().sum() for v in vals]))

  def grad_norm(self, g, opt_state, mixed=False):
    norm = math.sqrt(float((g.double() ** 2).sum()))
    opt_state[1] = norm
    fin = math.isfinite(norm)
    opt_state[2] = 1.0 if fin else 0.0
    if fin:
      opt_state[0] += 1.0
    if mixed:  # loss-scale controller, tfutils.py:225-240
      scale, good = float(opt_state[3]), float(opt_state[4])
      if not fin:
        scale, good = scale / 2, 0.0
      elif good >= 1000:
        scale, good = scale * 2, 0.0
      else:
        good += 1.0
      opt_state[3] = min(max(scale, 1e-4), 1e4)
      opt_state[4] = good

  def adam_step(self, p, g, m, v, n_decay, opt_state, lr, wd, eps, b1, b2,
                clip, warmup=0):
    if float(opt_state[2]) == 0.0:
      return
    norm = float(opt_state[1])
    gs = clip / max(norm, clip) if clip > 0 else 1.0
    t = float(opt_state[0])
    lr_wd = lr
    if warmup:   # tfutils.py:160-162: the decay sees the step count before the increment
      lr_wd = lr * min(max((t - 1.0) / warmup, 0.0), 1.0)
      lr = lr * min(max(t / warmup, 0.0), 1.0)
    gi = g * gs
    p[:n_decay] *= (1 - wd * lr_wd)
    m.copy_(b1 * m + (1 - b1) * gi)
    v.copy_(b2 * v + (1 - b2) * gi * gi)
    c1 = 1.0 / (1.0 - b1 ** t)
    c2 = 1.0 / (1.0 - b2 ** t)
    p.sub_(lr * (m * c1) / (torch.sqrt(v * c2) + eps))

  def fill(self, t, v=0.0):
    t.fill_(v)

  def copy2d(self, src, dst):
    dst.copy_(src)

  def replay_gather(self, ring, starts, out, first_flag=False):
    B, T = out.shape[:2]
    if first_flag:
      out.zero_()
      out[:, 0] = 1
      return
    idx = starts.reshape(B, 1) + torch.arange(T, device=starts.device).reshape(1, T)
    out.copy_(ring[idx.reshape(-1)].reshape(out.shape))

  def reset_mask(self, prev, first, init, out):
    f = first.reshape(-1, 1)
    pv = prev if prev is not None else torch.zeros_like(out)
    iv = init.reshape(1, -1) if init is not None else 0.0
    out.copy_(pv * (1 - f) + iv * f)

  def reset_mask2(self, prev_a, init_a, out_a, prev_b, init_b, out_b, first):
    self.reset_mask(prev_a, first, init_a, out_a)
    self.reset_mask(prev_b, first, init_b, out_b)

  def reset_mask_bwd2(self, dout_a, dprev_a, dout_b, dprev_b, first):
    self.reset_mask_bwd(dout_a, first, dprev_a)
    self.reset_mask_bwd(dout_b, first, dprev_b)

  def reset_mask_bwd(self, dout, first, dprev):
    dprev.add_(dout * (1 - first.reshape(-1, 1)))

  def batch_prep(self, is_first, is_terminal, action, first_f, cont_f,
                 act_masked):
    f = is_first.reshape(-1).float()
    first_f.copy_(f.reshape(first_f.shape))
    cont_f.copy_((1.0 - is_terminal.reshape(-1).float()).reshape(cont_f.shape))
    A = action.shape[-1]
    act_masked.copy_(action.reshape(-1, A) * (1 - f)[:, None])

  def tanh_fwd(self, x, y):
    y.copy_(torch.tanh(x))

  def tanh_bwd(self, x, dy, dx, beta=0.0):
    t = torch.tanh(x)
    r = dy * (1 - t * t)
    dx.copy_(beta * dx + r if beta != 0.0 else r)
