"""CPU ORACLE (test infrastructure, not product code).

A function-for-function restatement, in PyTorch-CPU with autograd, of the
DreamerV2+ learner step of danijar/daydreamer.  Only `tests/`, `bench.py`'s
`cpu_baseline` leg and `__graft_entry__.smoke()` may import this module; the
product package `daydreamer_amd` never does.

PARITY: the reference holds no golden vectors / known-answer tests for this path (all its agent
tests are timing smoke tests on zeros; sampling uses seed=None, weight init is unseeded), and
TensorFlow / TFP / sonnet are not installable here.  This restatement follows the reference
source line by line - every function cites the file:line it restates (paths relative to
/root/reference/embodied/agents/dreamerv2plus/) - and is PINNED TO THE REFERENCE'S OWN SOURCES
EXECUTED ON A TENSORFLOW STAND-IN: tests/golden/make_reference_golden.py imports agent.py,
nets.py, tfutils.py, tfagent.py and behaviors.py unmodified and runs two Agent.train calls on
oracle/tf_on_torch.py (the tf / tfd / snt primitives those files reach, on torch, float64);
tests/test_reference_golden.py holds this module to every metric, drawn class, per-parameter
gradient, updated parameter and controller state of those runs at 1e-9 (4 cases).  Not pinned
by that: TensorFlow's own numerics under the primitives, i.e. the documented semantics below.

TF/TFP semantics assumed from documentation (falsify these if TF is available).  Each one is
pinned WITHOUT PyTorch by a loop-form numpy / scalar restatement written from the documented
formula; the test is named after the arrow (tests/test_oracle_pins.py unless a file is given):
  * tf.nn.conv2d: NHWC, filter [kh,kw,in,out], cross-correlation, VALID; stride 1 'SAME' pads
    floor(k/2) zeros per side for odd k          -> test_oracle_independent.py::test_conv2d_matches_direct_loops,
                                                    ::test_same_conv_pool_repeat_and_residual_block_match_direct_loops
  * tf.nn.conv2d_transpose: filter [kh,kw,out,in], == input-gradient of conv2d
    (no kernel flip), VALID output = stride*in + k - stride
                                                 -> test_oracle_independent.py::test_conv2d_transpose_matches_direct_loops_and_is_the_adjoint
  * tf.nn.avg_pool 2x2 / stride 2, tf.repeat     -> test_oracle_independent.py::test_same_conv_pool_repeat_and_residual_block_match_direct_loops
  * tf.nn.moments: population variance; batch_normalization(x,m,v,off,scale,eps)
    = (x-m)*rsqrt(v+eps)*scale+off               -> test_layer_norm_is_population_moments_and_batch_normalization
  * tf.nn.elu alpha=1                            -> test_elu
  * symlog / symexp (tfutils.py:77-82)           -> test_symlog_symexp
  * tfd.kl_divergence(OneHotCategorical(a), OneHotCategorical(b))
    = sum softmax(a)*(log_softmax(a)-log_softmax(b));
    OneHotCategorical.entropy = -sum p log p     -> test_categorical_kl_and_entropy
  * .mode = one_hot(argmax), no grad; OneHotDist.sample straight-through (tfutils.py:368-382)
                                                 -> test_onehot_mode_and_straight_through_sample
  * Bernoulli(logits=l).log_prob(x) = x*logsigmoid(l)+(1-x)*logsigmoid(-l);
    .mean() = sigmoid(l)                         -> test_bernoulli_log_prob_and_mean
  * Normal.entropy = 0.5*log(2*pi*e*sigma^2); Normal.sample is reparameterised
                                                 -> test_normal_entropy_and_reparameterised_sample
  * tf.math.cumprod inclusive along axis 0 (agent.py:258)
                                                 -> test_discount_weights_are_cumprod_over_time
  * tf.linalg.global_norm; tf.clip_by_global_norm: g * clip / max(norm, clip); the reference's own
    Adam (tfutils.py:271-283) and the order clip -> decay -> Adam (:205-266)
                                                 -> test_global_norm_clip_and_literal_adam,
                                                    test_optimizer_order_clip_then_decay_then_adam
  * reduce_std / .std() population               -> (through AutoAdapt's metrics) test_autoadapt_*
and the reference's own control logic restated from its lines: the lambda-return recurrences
(agent.py:422-442) -> test_lambda_return_recurrences; AutoAdapt update-before-use (tfutils.py:
440-482) -> test_autoadapt_mult_updates_before_use, test_autoadapt_prop_and_fixed; Normalize's
bias-corrected float64 EMA (:498-527) -> test_normalize_bias_correction; balance_stats (:395-411)
-> test_balance_stats.  The autograd gradients are checked against float64 finite differences
in tests/test_oracle_independent.py.

Determinism contract: weights are an explicit name->tensor dict and every
stochastic site takes explicit noise (uniforms for the categorical latents,
standard normals for the actions), so `train` is a pure function.
The categorical sampler is inverse-CDF on the (unimixed) class probabilities;
the reference's `tf.random.categorical(seed=None)` is non-deterministic, any
exact categorical sampler is in-distribution.
"""

import math
import re

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-3  # nets.py:600


# ----------------------------------------------------------------------------
# tfutils.py
# ----------------------------------------------------------------------------

def symlog(x):  # tfutils.py:77-78
  return torch.sign(x) * torch.log(1 + torch.abs(x))


def symexp(x):  # tfutils.py:81-82
  return torch.sign(x) * (torch.exp(torch.abs(x)) - 1)


SAMPLE_TOL = [1e-5]  # near-boundary tolerance for forced draws (tests may widen)
# bookkeeping of forced draws: how many were compared / adopted from the device (tests print
# and bound the fraction; reset with SAMPLE_STATS.update(draws=0, adopted=0))
SAMPLE_STATS = dict(draws=0, adopted=0, max_gap=0.0)


def sample_onehot(probs, u, forced=None, tol=None):
  """Inverse-CDF categorical draw (replaces tf.random.categorical,
  tfutils.py:374).  probs [..., C], u [...] in [0,1).  Returns int64 indices.

  idx = #{c < C-1 : cdf_c <= u * cdf_{C-1}}.
  If `forced` is given (indices drawn by the device path from the same u), it
  is adopted wherever u lies within `tol` of a CDF boundary (float reassociation
  can legitimately flip such draws) and must otherwise agree exactly."""
  tol = SAMPLE_TOL[0] if tol is None else tol
  cdf = torch.cumsum(probs.detach(), -1)
  thr = (u * cdf[..., -1])[..., None]
  idx = (cdf[..., :-1] <= thr).sum(-1)
  if forced is not None:
    forced = torch.as_tensor(forced, dtype=torch.int64)
    differ = idx != forced
    SAMPLE_STATS['draws'] += int(differ.numel())
    SAMPLE_STATS['adopted'] += int(differ.sum())
    if differ.any():
      gap = torch.abs(cdf - thr).min(-1).values
      SAMPLE_STATS['max_gap'] = max(SAMPLE_STATS['max_gap'], float(gap[differ].max()))
      near = gap < tol
      bad = differ & ~near
      assert not bad.any(), (
          f'{int(bad.sum())} forced samples disagree outside tolerance '
          f'(max gap {float(gap[differ].max()):.3e}, tol {tol:.1e})')
      idx = torch.where(differ, forced, idx)
  return idx


def onehot_straight_through(logit, u, forced=None):
  """OneHotDist.sample, tfutils.py:368-382: draw, one-hot, then add
  probs - stop_gradient(probs) where probs = softmax(logit)."""
  probs = torch.softmax(logit, -1)
  idx = sample_onehot(probs, u, forced)
  sample = F.one_hot(idx, logit.shape[-1]).to(logit.dtype)
  return sample + probs - probs.detach(), idx


def onehot_mode(logit):
  """tfd.OneHotCategorical.mode(): one_hot(argmax), no gradient."""
  idx = torch.argmax(logit, -1)
  return F.one_hot(idx, logit.shape[-1]).to(logit.dtype)


def video_grid(video):  # tfutils.py:390-392
  B, T, H, W, C = video.shape
  return video.permute(1, 2, 0, 3, 4).reshape(T, H, B * W, C)


def categorical_kl(a, b):
  """tfd.kl_divergence(Independent(OneHotCategorical(a),1), ...(b)),
  nets.py:181-182 via get_dist nets.py:88-91: sum over classes then groups."""
  la = torch.log_softmax(a, -1)
  lb = torch.log_softmax(b, -1)
  return (torch.exp(la) * (la - lb)).sum(-1).sum(-1)


def categorical_entropy(a):
  la = torch.log_softmax(a, -1)
  return -(torch.exp(la) * la).sum(-1).sum(-1)


def bernoulli_log_prob(logit, x):
  """tfd.Bernoulli(logits=l).log_prob(x), nets.py:469-471 (the `cont` head):
  x * log sigmoid(l) + (1 - x) * log sigmoid(-l)."""
  return x * F.logsigmoid(logit) + (1 - x) * F.logsigmoid(-logit)


def normal_entropy(std):
  """tfd.Normal(mean, std).entropy() = 0.5 * log(2 pi e) + log(std)  (agent.py:361-371;
  `std` may be a python float for the minent / maxent bounds of nets.py:466-467)."""
  if not torch.is_tensor(std):
    return 0.5 * math.log(2 * math.pi * math.e) + math.log(std)
  return 0.5 * math.log(2 * math.pi * math.e) + torch.log(std)


def discount_weights(cont, discount):
  """agent.py:258: tf.math.cumprod(discount * cont) / discount along the time axis."""
  return torch.cumprod(discount * cont, 0) / discount


def lambda_return(reward, value, disc, lam, impl='gve'):
  """agent.py:422-442 (VFunction.target): reward [H], value [H+1], disc [H] (all with trailing
  batch axes).  'gve': ret_t = r_t + disc_t * ((1 - lam) * v_{t+1} + lam * ret_{t+1}), bootstrap
  ret_H = v_H; 'gae': adv_t = delta_t + disc_t * lam * adv_{t+1}, returns adv + v.  Returns
  (target [H], baseline value[:-1])."""
  if impl == 'gae':  # :428-433
    advs = [torch.zeros_like(value[0])]
    deltas = reward + disc * value[1:] - value[:-1]
    for t in reversed(range(len(disc))):
      advs.append(deltas[t] + disc[t] * lam * advs[-1])
    adv = torch.stack(list(reversed(advs))[:-1])
    return adv + value[:-1], value[:-1]
  assert impl == 'gve', impl  # :434-440
  vals = [value[-1]]
  interm = reward + disc * value[1:] * (1 - lam)
  for t in reversed(range(len(disc))):
    vals.append(interm[t] + disc[t] * lam * vals[-1])
  ret = torch.stack(list(reversed(vals))[:-1])
  return ret, value[:-1]


def global_norm(grads):
  """tf.linalg.global_norm, tfutils.py:243: sqrt(sum over tensors of sum of squares)."""
  return torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)


def clip_by_global_norm(grads, clip, norm):
  """tf.clip_by_global_norm(grads, clip, use_norm=norm), tfutils.py:244-245:
  g * clip / max(norm, clip)."""
  return [g * clip / torch.clamp(norm, min=clip) for g in grads]


def adam_update(p, g, m, v, t, lr, eps, b1=0.9, b2=0.999):
  """tfutils.py:271-283 (_apply_adam), one tensor: returns the new (m, v); p is updated in place."""
  m = b1 * m + (1. - b1) * g
  v = b2 * v + (1. - b2) * g * g
  m_hat = m / (1. - b1 ** t)
  v_hat = v / (1. - b2 ** t)
  p.sub_(lr * m_hat / (torch.sqrt(v_hat) + eps))
  return m, v


class AutoAdapt:
  """tfutils.py:414-482 ('fixed', 'mult' and 'prop' impls)."""

  def __init__(self, shape, impl, scale, target, min, max, vel=0.1,
               thres=0.1, inverse=False, dtype=torch.float32):
    """dtype: the scale is a float32 variable in the reference (:430-432); the comparison with the
    reference's sources run in float64 (tests/test_reference_golden.py) keeps it in float64."""
    self.shape = tuple(shape)
    self.impl, self.target, self.min, self.max = impl, target, min, max
    self.vel, self.thres, self.inverse = vel, thres, inverse
    self.dtype = dtype
    if impl == 'fixed':
      self.scale = torch.tensor(float(scale), dtype=dtype)
    elif impl in ('mult', 'prop'):
      self.scale = torch.ones(self.shape, dtype=dtype)  # :430-432
    else:
      raise NotImplementedError(impl)

  def __call__(self, reg, update=True):  # :440-447
    update and self.update(reg)
    scale = self.scale.to(reg.dtype)
    loss = scale * (-reg if self.inverse else reg)
    metrics = {
        'mean': reg.mean(), 'std': reg.std(unbiased=False),
        'scale_mean': scale.mean(),
        'scale_std': scale.std(unbiased=False) if scale.numel() > 1
        else torch.zeros(())}
    return loss, metrics

  def update(self, reg):  # :460-474
    if self.impl == 'fixed':
      return
    dims = list(range(reg.dim() - len(self.shape)))
    avg = reg.detach().mean(dims).to(self.dtype)
    if self.impl == 'prop':  # :475-480
      direction = avg - self.target
      if self.inverse:
        direction = -direction
      self.scale = torch.clamp(self.scale + self.vel * direction, self.min, self.max)
      return
    below = avg < (1 / (1 + self.thres)) * self.target
    above = avg > (1 + self.thres) * self.target
    if self.inverse:
      below, above = above, below
    inside = ~below & ~above
    adjusted = (
        above.to(self.dtype) * self.scale * (1 + self.vel) +
        below.to(self.dtype) * self.scale / (1 + self.vel) +
        inside.to(self.dtype) * self.scale)
    self.scale = torch.clamp(adjusted, self.min, self.max)


class Normalize:
  """tfutils.py:485-527; state is float64 (:494-496)."""

  def __init__(self, impl='mean_std', decay=0.99, max=1e8, vareps=0.0,
               stdeps=0.0):
    self.impl, self.decay, self.max = impl, decay, max
    self.vareps, self.stdeps = vareps, stdeps
    self.mean = torch.zeros((), dtype=torch.float64)
    self.sqrs = torch.zeros((), dtype=torch.float64)
    self.step = 0

  def __call__(self, values, update=True):
    update and self.update(values)
    return self.transform(values)

  def update(self, values):  # :502-507
    x = values.detach().double()
    m = self.decay
    self.step += 1
    self.mean = m * self.mean + (1 - m) * x.mean()
    self.sqrs = m * self.sqrs + (1 - m) * (x ** 2).mean()

  def transform(self, values):  # :509-527
    correction = 1 - self.decay ** float(self.step)
    mean = self.mean / correction
    var = (self.sqrs / correction) - mean ** 2
    if self.max > 0.0:
      scale = torch.rsqrt(
          torch.clamp(var, min=1 / self.max ** 2 + self.vareps) + self.stdeps)
    else:
      scale = torch.rsqrt(var + self.vareps) + self.stdeps
    if self.impl == 'off':
      pass
    elif self.impl == 'mean_std':
      values = values - mean.to(values.dtype)
      values = values * scale.to(values.dtype)
    elif self.impl == 'std':
      values = values * scale.to(values.dtype)
    else:
      raise NotImplementedError(self.impl)
    return values


class Optimizer:
  """tfutils.py:143-302: global-norm clip, weight decay on names matching
  `wd_pattern` applied before Adam, hand-coded bias-corrected Adam."""

  def __init__(self, name, lr, opt='adam', eps=1e-5, clip=0.0, warmup=0,
               wd=0.0, wd_pattern='kernel'):
    assert opt == 'adam'
    self.name, self.lr, self.eps, self.clip = name, lr, eps, clip
    self.warmup = warmup
    self.wd, self.wd_pattern = wd, wd_pattern
    self.step = 0
    self.m, self.v = {}, {}

  def _lr(self):
    """:160-162: lr * clip(step / warmup, 0, 1) with the step count AT THE TIME OF THE CALL - the
    decay (:254-256, before the increment) sees the old count, Adam (:260-261, after it) the new."""
    if not self.warmup:
      return self.lr
    return self.lr * min(max(self.step / self.warmup, 0.0), 1.0)

  def __call__(self, loss, params, names, world_grads=None):
    """params: dict name->leaf tensor; names: which of them to train.
    Returns metrics and the raw gradients (for parity tests)."""
    metrics = {}
    names = sorted(names)  # :189
    plist = [params[n] for n in names]
    if not torch.isfinite(loss):  # check_numerics :207
      raise FloatingPointError(self.name + '_loss')
    metrics[f'{self.name}_loss'] = loss.detach()
    grads = torch.autograd.grad(loss, plist, retain_graph=True,
                                allow_unused=True)
    for n, g in zip(names, grads):
      if g is None:  # :215-218
        raise RuntimeError(
            f'{self.name} optimizer found no gradient for {n}.')
    if world_grads is not None:  # :221-223 all_reduce('mean')
      grads = world_grads(grads)
    raw = {n: g.detach().clone() for n, g in zip(names, grads)}
    norm = global_norm(grads)  # :243
    if self.clip:  # :244-245
      grads = clip_by_global_norm(grads, self.clip, norm)
    if not torch.isfinite(norm):  # :249
      raise FloatingPointError(self.name + '_norm')
    metrics[f'{self.name}_grad_norm'] = norm
    with torch.no_grad():
      if self.wd:  # :254-256, 285-301
        for n, p in zip(names, plist):
          if re.search(self.wd_pattern, self.name + '/' + n):
            p.mul_(1 - self.wd * self._lr())
      self.step += 1  # :260
      t = float(self.step)
      for n, p, g in zip(names, plist, grads):  # :271-283
        if n not in self.m:
          self.m[n] = torch.zeros_like(p)
          self.v[n] = torch.zeros_like(p)
        self.m[n], self.v[n] = adam_update(p, g, self.m[n], self.v[n], t, self._lr(), self.eps)
    metrics[f'{self.name}_grad_steps'] = torch.tensor(self.step)
    return metrics, raw


def balance_stats(mean, logp_fn, target, thres):  # tfutils.py:395-411
  pos = (target > thres).to(target.dtype)
  neg = (target <= thres).to(target.dtype)
  pred = (mean > thres).to(target.dtype)
  loss = -logp_fn(target)
  return dict(
      pos_loss=(loss * pos).sum() / pos.sum(),
      neg_loss=(loss * neg).sum() / neg.sum(),
      pos_acc=(pred * pos).sum() / pos.sum(),
      neg_acc=((1 - pred) * neg).sum() / neg.sum(),
      rate=pos.mean(), avg=target.mean(), pred=mean.mean())


# ----------------------------------------------------------------------------
# nets.py
# ----------------------------------------------------------------------------

def get_act(name):  # nets.py:629-643
  return {'none': lambda x: x, 'elu': F.elu, 'relu': F.relu,
          'tanh': torch.tanh}[name]


def layer_norm(x, scale, bias):  # nets.py:594-600
  mean = x.mean(-1, keepdim=True)
  var = ((x - mean) ** 2).mean(-1, keepdim=True)
  return (x - mean) * torch.rsqrt(var + LN_EPS) * scale + bias


def linear(p, name, x, act='none', norm='none', **_unused):
  """nets.py:557-582.  Bias exists only when norm == 'none' (:563)."""
  x = x @ p[f'{name}/kernel']
  if norm == 'none':
    x = x + p[f'{name}/bias']
  else:
    x = layer_norm(x, p[f'{name}/norm/scale'], p[f'{name}/norm/bias'])
  return get_act(act)(x)


def conv2d(p, name, x, transp, act='none', norm='none'):
  """nets.py:495-554, stride 2, VALID, bias kept even with LayerNorm
  (:548-553).  x is NHWC."""
  k = p[f'{name}/kernel']
  xn = x.permute(0, 3, 1, 2)
  if transp:  # filter [kh,kw,out,in]  (:523, :539)
    y = F.conv_transpose2d(xn, k.permute(3, 2, 0, 1), stride=2)
  else:       # filter [kh,kw,in,out]  (:541, :547)
    y = F.conv2d(xn, k.permute(3, 2, 0, 1), stride=2)
  y = y.permute(0, 2, 3, 1) + p[f'{name}/bias']
  if norm != 'none':
    y = layer_norm(y, p[f'{name}/norm/scale'], p[f'{name}/norm/bias'])
  return get_act(act)(y)


def conv2d_same(p, name, x, act='none', norm='none', preact=False, bias=True):
  """nets.py:495-554 with Conv2D's defaults stride 1, pad 'same' (odd kernel: (k-1)/2 zeros on
  every side).  preact (:510-513): norm and activation act on the INPUT, then the layer."""
  if preact:
    if norm != 'none':
      x = layer_norm(x, p[f'{name}/norm/scale'], p[f'{name}/norm/bias'])
    x = get_act(act)(x)
  k = p[f'{name}/kernel']
  y = F.conv2d(x.permute(0, 3, 1, 2), k.permute(3, 2, 0, 1), padding=k.shape[0] // 2)
  y = y.permute(0, 2, 3, 1)
  if bias:
    y = y + p[f'{name}/bias']
  if not preact:
    if norm != 'none':
      y = layer_norm(y, p[f'{name}/norm/scale'], p[f'{name}/norm/bias'])
    y = get_act(act)(y)
  return y


def res_block(p, name, depth, x, **kw):  # nets.py:351-358, 384-391
  skip = x
  if skip.shape[-1] != depth:
    skip = conv2d_same(p, f'{name}s', skip, bias=False)
  x = conv2d_same(p, f'{name}a', x, preact=True, **kw)
  x = conv2d_same(p, f'{name}b', x, preact=True, **kw)
  return skip + 0.1 * x


def encoder_resnet(p, name, image, depth, blocks, **kw):  # nets.py:337-349
  stages = int(np.log2(image.shape[-2])) - 2
  x = conv2d_same(p, f'{name}/in', image)
  for i in range(stages):
    # tf.nn.avg_pool(x, [2, 2], [2, 2], 'SAME') on even sides: plain 2x2 means
    x = F.avg_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    for j in range(blocks):
      x = res_block(p, f'{name}/s{i}b{j}', depth, x, **kw)
    depth *= 2
  x = x.reshape(x.shape[0], -1)
  return linear(p, f'{name}/out', x)


def decoder_resnet(p, name, feat, shape, depth, blocks, **kw):  # nets.py:370-382
  stages = int(np.log2(shape[0])) - 2
  depth = 2 ** stages * depth
  x = linear(p, f'{name}/in', feat)
  x = x.reshape(-1, 4, 4, depth)
  for i in range(stages):
    for j in range(blocks):
      x = res_block(p, f'{name}/s{i}b{j}', depth, x, **kw)
    x = x.repeat_interleave(2, 1).repeat_interleave(2, 2)  # tf.repeat(tf.repeat(x, 2, 1), 2, 2)
    depth //= 2
  return torch.sigmoid(conv2d_same(p, f'{name}/out', x))


def mlp_trunk(p, name, x, layers, act, norm):  # nets.py:408-414
  for i in range(layers):
    x = linear(p, f'{name}/dense{i}', x, act, norm)
  return x


class RSSM:
  """nets.py:11-183 (discrete latent, 'learned2' initial, gru_layers 1)."""

  def __init__(self, p, deter, stoch, classes, units, act, norm, unimix,
               prior_layers, post_layers, initial='learned2', gru_layers=1,
               **_unused):
    assert initial == 'learned2' and gru_layers == 1 and post_layers == 1
    self.p = p
    self.deter, self.stoch, self.classes = deter, stoch, classes
    self.kw = dict(act=act, norm=norm)
    self.unimix = unimix
    self.prior_layers = prior_layers

  def initial(self, bs):  # nets.py:28-62 ('learned2')
    deter = torch.tanh(self.p['rssm/initial_deter'])[None].repeat(bs, 1)
    dt = deter.dtype
    return dict(
        deter=deter,
        logit=torch.zeros(bs, self.stoch, self.classes, dtype=dt),
        stoch=self.get_stoch(deter))

  def get_stoch(self, deter):  # nets.py:140-147
    x = deter
    for i in range(self.prior_layers):
      x = linear(self.p, f'rssm/img_out_{i}', x, **self.kw)
    return onehot_mode(self._stats_layer('rssm/img_stats', x))

  def _stats_layer(self, name, x):  # nets.py:162-171
    x = linear(self.p, name, x)  # default kwargs: bias, no norm, no act
    logit = x.reshape(x.shape[:-1] + (self.stoch, self.classes))
    if self.unimix:
      probs = torch.softmax(logit, -1)
      uniform = torch.ones_like(probs) / probs.shape[-1]
      probs = (1 - self.unimix) * probs + self.unimix * uniform
      logit = torch.log(probs)
    return logit

  def _gru(self, x, deter):  # nets.py:149-160
    x = torch.cat([deter, x], -1)
    x = linear(self.p, 'rssm/gru_out', x, act='none', norm=self.kw['norm'])
    reset, cand, update = torch.split(x, self.deter, -1)
    reset = torch.sigmoid(reset)
    cand = torch.tanh(reset * cand)
    update = torch.sigmoid(update - 1)
    return update * cand + (1 - update) * deter

  def img_step(self, prev, action, u, forced=None):  # nets.py:119-138
    stoch = prev['stoch'].reshape(prev['stoch'].shape[0], -1)
    x = torch.cat([stoch, action], -1)
    x = linear(self.p, 'rssm/img_in', x, **self.kw)
    deter = self._gru(x, prev['deter'])
    x = deter
    for i in range(self.prior_layers):
      x = linear(self.p, f'rssm/img_out_{i}', x, **self.kw)
    logit = self._stats_layer('rssm/img_stats', x)
    stoch, idx = onehot_straight_through(logit, u, forced)
    return dict(stoch=stoch, deter=deter, logit=logit), idx

  def obs_step(self, prev, action, embed, is_first, u_prior, u_post,
               forced_prior=None, forced_post=None):  # nets.py:99-117
    f = is_first.to(embed.dtype)
    mask = lambda x: x * (1.0 - f).reshape((-1,) + (1,) * (x.dim() - 1))
    prev = {k: mask(v) for k, v in prev.items()}  # :102-104
    action = mask(action)
    init = self.initial(len(f))  # :105-107
    prev = {k: v + init[k] * f.reshape((-1,) + (1,) * (v.dim() - 1))
            for k, v in prev.items()}
    prior, idx_prior = self.img_step(prev, action, u_prior, forced_prior)
    x = torch.cat([prior['deter'], embed], -1)  # :109
    x = linear(self.p, 'rssm/obs_out', x, **self.kw)
    logit = self._stats_layer('rssm/obs_stats', x)
    stoch, idx_post = onehot_straight_through(logit, u_post, forced_post)
    post = dict(stoch=stoch, deter=prior['deter'], logit=logit)
    return post, prior, idx_prior, idx_post

  def observe(self, embed, action, is_first, state, u_prior, u_post,
              forced=None):
    """nets.py:66-76 with tfutils.scan (static unroll) tfutils.py:50-70.
    embed [B,T,E]; noise [T,B,G]."""
    B, T = action.shape[:2]
    if state is None:
      state = self.initial(B)
    posts, priors, idxs = [], [], dict(prior=[], post=[])
    prev = state
    for t in range(T):
      fp = None if forced is None else forced['obs_prior'][t]
      fq = None if forced is None else forced['obs_post'][t]
      post, prior, ip, iq = self.obs_step(
          prev, action[:, t], embed[:, t], is_first[:, t],
          u_prior[t], u_post[t], fp, fq)
      posts.append(post)
      priors.append(prior)
      idxs['prior'].append(ip)
      idxs['post'].append(iq)
      prev = post
    stack = lambda seq: {k: torch.stack([s[k] for s in seq], 1)
                         for k in seq[0]}
    idxs = {k: torch.stack(v, 0) for k, v in idxs.items()}
    return stack(posts), stack(priors), idxs

  def kl_loss(self, post, prior, balance=0.8):  # nets.py:178-183
    lhs = categorical_kl(post['logit'].detach(), prior['logit'])
    rhs = categorical_kl(post['logit'], prior['logit'].detach())
    return balance * lhs + (1 - balance) * rhs


def feat_of(state):  # nets.Input(['deter','stoch']) nets.py:612-626
  stoch = state['stoch']
  return torch.cat(
      [state['deter'], stoch.reshape(stoch.shape[:-2] + (-1,))], -1)


# ----------------------------------------------------------------------------
# agent.py
# ----------------------------------------------------------------------------

class RefAgent:
  """Restates Agent / WorldModel / ImagActorCritic / VFunction (agent.py) for
  the Greedy behaviour with a V-function critic and a continuous 'normal'
  actor trained by backprop through the imagined rollout."""

  def __init__(self, cfg, obs_shapes, act_dim, params, dtype=torch.float64,
               act_discrete=False, ctrl_dtype=torch.float32):
    """cfg: nested dict as in configs.yaml; obs_shapes: name->shape tuple of
    the observation space (without batch dims); params: name->array.
    act_discrete: one-hot action space -> 'onehot' actor trained by REINFORCE
    (actor_dist_disc / actor_grad_disc, agent.py:295-300)."""
    self.cfg = cfg
    self.dtype = dtype
    self.act_dim = act_dim
    self.discrete = act_discrete
    self.p = {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(
        not k.startswith('critic_target/')) for k, v in params.items()}
    enc, dec = cfg['encoder'], cfg['decoder']
    excl = ('is_first', 'is_last')
    shapes = {k: tuple(v) for k, v in obs_shapes.items()
              if not k.startswith('log_')}
    s_enc = {k: v for k, v in shapes.items() if k not in excl}
    # nets.py:192-199
    self.enc_cnn = [k for k, v in s_enc.items()
                    if re.match(enc['cnn_keys'], k) and len(v) == 3]
    self.enc_mlp = [k for k, v in s_enc.items()
                    if re.match(enc['mlp_keys'], k) and len(v) in (0, 1)]
    self.enc_shapes = s_enc
    excl = ('is_first', 'is_last', 'is_terminal', 'reward')  # nets.py:241
    s_dec = {k: v for k, v in shapes.items() if k not in excl}
    self.dec_cnn = {k: v for k, v in s_dec.items()
                    if re.match(dec['cnn_keys'], k) and len(v) == 3}
    self.dec_mlp = {k: v for k, v in s_dec.items()
                    if re.match(dec['mlp_keys'], k) and len(v) == 1}
    self.rssm = RSSM(self.p, **cfg['rssm'])
    self.wmkl = AutoAdapt((), **cfg['wmkl'], inverse=False, dtype=ctrl_dtype)  # agent.py:155
    self.model_opt = Optimizer('model', **cfg['model_opt'])
    self.actor_opt = Optimizer('actor', **cfg['actor_opt'])
    self.critic_opt = Optimizer('critic', **cfg['critic_opt'])
    self.advnorm = Normalize(**cfg['advnorm'])      # agent.py:301
    self.retnorm = Normalize(**cfg['retnorm'])      # agent.py:302-303
    self.scorenorm = Normalize(**cfg['scorenorm'])  # agent.py:304-305
    # agent.py:306-308: per-dimension scale for continuous, scalar for discrete
    self.actent = AutoAdapt(() if act_discrete else (act_dim,), **cfg['actent'],
                            inverse=True, dtype=ctrl_dtype)
    self.slow_updates = -1  # agent.py:393
    self.last = {}

  # -- networks ---------------------------------------------------------------

  def encoder(self, data):  # nets.py:212-232 (+ Simple CNN :298-305)
    enc = self.cfg['encoder']
    kw = dict(act=enc['act'], norm=enc['norm'])
    lead = data['is_first'].shape
    outs = []
    if self.enc_cnn:
      x = torch.cat([data[k].reshape((-1,) + data[k].shape[len(lead):])
                     for k in self.enc_cnn], -1)
      if enc['cnn'] == 'resnet':  # nets.py:206-207
        x = encoder_resnet(self.p, 'enc/cnn', x, enc['cnn_depth'], enc['cnn_blocks'], **kw)
      else:
        for i, _ in enumerate(enc['cnn_kernels']):
          x = conv2d(self.p, f'enc/cnn/conv{i}', x, False, **kw)
      outs.append(x.reshape(x.shape[0], -1))
    if self.enc_mlp:
      xs = []
      for k in self.enc_mlp:
        v = data[k].reshape((-1,) + data[k].shape[len(lead):])
        xs.append(v[..., None] if len(self.enc_shapes[k]) == 0 else v)
      x = torch.cat(xs, -1)
      outs.append(mlp_trunk(self.p, 'enc/mlp', x, enc['mlp_layers'], **kw))
    out = torch.cat(outs, -1)
    return out.reshape(lead + out.shape[1:])

  def decoder(self, feat):
    """nets.py:267-280 (+ ImageDecoderSimple :316-327).  Returns dict of
    means: images through sigmoid, vectors linear (MSEDist)."""
    dec = self.cfg['decoder']
    kw = dict(act=dec['act'], norm=dec['norm'])
    lead = feat.shape[:-1]
    flat = feat.reshape(-1, feat.shape[-1])
    means = {}
    if self.dec_cnn and dec['cnn'] == 'resnet':  # nets.py:255-256
      shapes = list(self.dec_cnn.values())
      merged = tuple(shapes[0][:-1]) + (sum(v[-1] for v in shapes),)
      x = decoder_resnet(self.p, 'dec/cnn', flat, merged, dec['cnn_depth'], dec['cnn_blocks'], **kw)
    elif self.dec_cnn:
      x = flat.reshape(-1, 1, 1, flat.shape[-1])
      for i, _ in enumerate(dec['cnn_kernels'][:-1]):
        x = conv2d(self.p, f'dec/cnn/conv{i}', x, True, **kw)
      x = torch.sigmoid(conv2d(self.p, 'dec/cnn/out', x, True))
    if self.dec_cnn:
      x = x.reshape(lead + x.shape[1:])
      chans = [v[-1] for v in self.dec_cnn.values()]
      for k, m in zip(self.dec_cnn, torch.split(x, chans, -1)):
        means[k] = m
    if self.dec_mlp:
      x = mlp_trunk(self.p, 'dec/mlp', flat, dec['mlp_layers'], **kw)
      for k, shape in self.dec_mlp.items():
        m = linear(self.p, f'dec/mlp/dist_{k}/out', x)
        means[k] = m.reshape(lead + tuple(shape))
    return means

  def head(self, name, feat, cfgkey):
    """MLP(()) + DistLayer, nets.py:408-425, 447-452: returns raw `out`."""
    c = self.cfg[cfgkey]
    lead = feat.shape[:-1]
    x = mlp_trunk(self.p, name, feat.reshape(-1, feat.shape[-1]),
                  c['layers'], c['act'], c['norm'])
    out = linear(self.p, f'{name}/dist_out/out', x)
    return out.reshape(lead)

  def actor(self, feat):
    """nets.py:461-468 'normal': mean=tanh(out), std=(hi-lo)*sigmoid(s)+lo."""
    c = self.cfg['actor']
    lead = feat.shape[:-1]
    x = mlp_trunk(self.p, 'actor', feat.reshape(-1, feat.shape[-1]),
                  c['layers'], c['act'], c['norm'])
    out = linear(self.p, 'actor/dist_out/out', x)
    if self.discrete:  # nets.py:480-491 'onehot' with unimix
      logit = out.reshape(lead + (self.act_dim,))
      if c['unimix']:
        probs = torch.softmax(logit, -1)
        probs = (1 - c['unimix']) * probs + c['unimix'] / self.act_dim
        logit = torch.log(probs)
      return logit, None
    std = linear(self.p, 'actor/dist_out/std', x)
    lo, hi = c['minstd'], c['maxstd']
    mean = torch.tanh(out).reshape(lead + (self.act_dim,))
    std = ((hi - lo) * torch.sigmoid(std) + lo).reshape(
        lead + (self.act_dim,))
    return mean, std

  # -- agent ------------------------------------------------------------------

  def preprocess(self, data):  # agent.py:123-139
    obs = {}
    for key, value in data.items():
      if key.startswith('log_') or key in ('key',):
        continue
      value = np.asarray(value)
      if value.ndim > 3 and value.dtype == np.uint8:
        obs[key] = torch.tensor(value, dtype=self.dtype) / 255.0
      else:
        obs[key] = torch.tensor(value.astype(np.float64), dtype=self.dtype)
    assert self.cfg['transform_rewards'] in ('off', False)
    obs['cont'] = 1.0 - obs['is_terminal']
    return obs

  def wm_loss(self, data, state, noise, forced=None, update=True):  # agent.py:165-212
    # (update=False: the AutoAdapt scale is read but not adapted - Agent.report never
    # applies gradients or keeps variable updates; see RefAgent.report)
    cfg = self.cfg
    metrics = {}
    embed = self.encoder(data)
    post, prior, idxs = self.rssm.observe(
        embed, data['action'], data['is_first'], state,
        noise['u_obs_prior'], noise['u_obs_post'], forced)
    feat = feat_of(post)
    feat_const = feat.detach()
    gh = cfg['grad_heads']
    losses = {}
    kl = self.rssm.kl_loss(post, prior, cfg['wmkl_balance'])
    kl, mets = self.wmkl(kl, update=update)
    losses['kl'] = kl
    metrics.update({f'wmkl_{k}': v for k, v in mets.items()})
    means = self.decoder(feat if 'decoder' in gh else feat_const)
    for key, mean in means.items():  # MSEDist.log_prob tfutils.py:320-329
      ndim = mean.dim() - 2
      dist = (mean - data[key]) ** 2
      losses[key] = dist.sum(tuple(range(-ndim, 0)))
    rew = self.head('reward', feat if 'reward' in gh else feat_const,
                    'reward_head')
    losses['reward'] = (rew - symlog(data['reward'])) ** 2  # :347-356
    cont = self.head('cont', feat if 'cont' in gh else feat_const,
                     'cont_head')
    x = data['cont']
    losses['cont'] = -bernoulli_log_prob(cont, x)
    metrics.update({f'{k}_loss_mean': v.mean() for k, v in losses.items()})
    metrics.update(
        {f'{k}_loss_std': v.std(unbiased=False) for k, v in losses.items()})
    model_loss = sum(
        v * cfg['loss_scales'].get(k, 1.0) for k, v in losses.items())
    metrics['prior_ent_mean'] = categorical_entropy(prior['logit']).mean()
    metrics['post_ent_mean'] = categorical_entropy(post['logit']).mean()
    metrics['prior_ent_min'] = categorical_entropy(prior['logit']).min()
    metrics['post_ent_min'] = categorical_entropy(post['logit']).min()
    metrics['model_loss_mean'] = model_loss.mean()
    metrics['model_loss_std'] = model_loss.std(unbiased=False)
    # agent.py:204-209 (tf.debug_nans False)
    for k, v in balance_stats(symexp(rew), lambda t: -((rew - symlog(t)) ** 2),
                              data['reward'], 0.1).items():
      metrics[f'reward_{k}'] = v
    for k, v in balance_stats(
        torch.sigmoid(cont),
        lambda t: bernoulli_log_prob(cont, t),
        data['cont'], 0.5).items():
      metrics[f'cont_{k}'] = v
    last_state = {k: v[:, -1].detach() for k, v in post.items()}
    out = dict(embed=embed, post=post, prior=prior, idxs=idxs,
               losses=losses)
    return model_loss.mean(), last_state, out, metrics

  def imagine(self, start, first_cont, noise, horizon, forced=None):
    """agent.py:234-261 with the policy of ImagActorCritic.train
    (agent.py:319-320): actor(sg(latent)).sample()."""
    sg = lambda s: {k: v.detach() for k, v in s.items()}
    def policy(state, eps, t):
      mean, std = self.actor(feat_of(sg(state)))
      if self.discrete:  # OneHotDist.sample (straight-through), tfutils.py:368-382
        f = None if forced is None else forced['act'][t]
        a, _ = onehot_straight_through(mean, noise['u_act'][t], f)
        return a, (mean, mean)
      return mean + std * eps, (mean, std)
    eps_all = noise.get('eps_act', [None] * (horizon + 1))
    states, actions, dists, idxs = [start], [], [], []
    action, d = policy(start, eps_all[0], 0)
    actions.append(action)
    dists.append(d)
    state = start
    for t in range(horizon):
      f = None if forced is None else forced['img'][t]
      state, idx = self.rssm.img_step(state, action, noise['u_img'][t], f)
      action, d = policy(state, eps_all[t + 1], t + 1)
      states.append(state)
      actions.append(action)
      dists.append(d)
      idxs.append(idx)
    traj = {k: torch.stack([s[k] for s in states], 0) for k in start}
    traj['action'] = torch.stack(actions, 0)
    cont = torch.sigmoid(self.head('cont', feat_of(traj), 'cont_head'))
    traj['cont'] = torch.cat([first_cont[None], cont[1:]], 0)  # :256-257
    disc = self.cfg['discount']
    traj['weight'] = discount_weights(traj['cont'], disc)  # :258
    traj['idx'] = torch.stack(idxs, 0) if idxs else None
    traj['policy_mean'] = torch.stack([d[0] for d in dists], 0)
    traj['policy_std'] = torch.stack([d[1] for d in dists], 0)
    return traj

  def critic_target(self, traj, reward, prefix, impl='gve'):  # agent.py:422-442
    cfg = self.cfg
    disc = traj['cont'][1:] * cfg['discount']
    value = symexp(self.head(prefix, feat_of(traj), 'critic'))
    return lambda_return(reward, value, disc, cfg['return_lambda'], impl)

  def update_slow(self):  # agent.py:444-454
    cfg = self.cfg
    if not cfg['slow_target']:
      return
    initialize = (self.slow_updates == -1)
    if initialize or self.slow_updates >= cfg['slow_target_update']:
      self.slow_updates = 0
      mix = 1.0 if initialize else cfg['slow_target_fraction']
      with torch.no_grad():
        for k in self.p:
          if k.startswith('critic/'):
            d = self.p['critic_target/' + k[len('critic/'):]]
            d.copy_(mix * self.p[k] + (1 - mix) * d)
    self.slow_updates += 1

  def train(self, data, noise, state=None, forced=None, world_grads=None):
    """agent.py:67-93.  Returns outs, state, metrics; raw gradients and
    intermediate tensors are kept in self.last for the parity tests."""
    cfg = self.cfg
    noise = {k: torch.tensor(np.asarray(v), dtype=self.dtype)
             for k, v in noise.items()}
    metrics = {}
    data = self.preprocess(data)
    # ---- WorldModel.train, agent.py:157-163
    wm_names = [k for k in self.p if k.split('/')[0] in
                ('enc', 'rssm', 'dec', 'reward', 'cont')]
    model_loss, state, wm_out, mets = self.wm_loss(data, state, noise, forced)
    metrics.update(mets)
    mets, g_model = self.model_opt(model_loss, self.p, wm_names, world_grads)
    metrics.update(mets)
    # ---- flatten context, agent.py:82-83
    post = wm_out['post']
    start = {k: v.detach().reshape((-1,) + v.shape[2:])
             for k, v in post.items()}
    first_cont = (1.0 - data['is_terminal']).reshape(-1)
    # ---- ImagActorCritic.train, agent.py:317-324 (uses UPDATED wm weights)
    H = cfg['imag_horizon']
    traj = self.imagine(start, first_cont, noise, H, forced)
    # ---- VFunction.train, agent.py:398-417
    rewfn = lambda tr: symexp(self.head('reward', feat_of(tr),
                                        'reward_head'))[1:]
    reward = rewfn(traj)
    tprefix = 'critic_target' if cfg['slow_target'] else 'critic'
    target = self.critic_target(traj, reward, tprefix, cfg['critic_return'])[0].detach()
    tr_in = {k: v[:-1].detach() for k, v in traj.items()
             if k in ('deter', 'stoch')}
    out = self.head('critic', feat_of(tr_in), 'critic')
    logp = -(out - symlog(target)) ** 2
    critic_loss = -(logp * traj['weight'][:-1].detach()).mean()
    critic_names = [k for k in self.p if k.startswith('critic/')]
    mets, g_critic = self.critic_opt(critic_loss, self.p, critic_names,
                                     world_grads)
    metrics.update({f'extr_{k}': v for k, v in mets.items()})
    cm = symexp(out.detach())
    metrics.update({
        'extr_critic_loss': critic_loss.detach(),
        'extr_imag_reward_mean': reward.mean().detach(),
        'extr_imag_reward_std': reward.std(unbiased=False).detach(),
        'extr_imag_critic_mean': cm.mean(),
        'extr_imag_critic_std': cm.std(unbiased=False),
        'extr_imag_return_mean': target.mean(),
        'extr_imag_return_std': target.std(unbiased=False)})
    self.update_slow()
    # ---- actor update, agent.py:326-349 (sees the post-update slow critic)
    ret, baseline = self.critic_target(traj, rewfn(traj), tprefix, cfg['actor_return'])
    ret = self.retnorm(ret)
    baseline = self.retnorm(baseline, update=False)
    score = self.scorenorm(ret - baseline)
    metrics['extr_score_mean'] = score.mean().detach()
    metrics['extr_score_std'] = score.std(unbiased=False).detach()
    metrics['extr_score_mag'] = score.abs().mean().detach()
    metrics['extr_score_max'] = score.abs().max().detach()
    score = self.advnorm(score * 1.0)
    # ---- ImagActorCritic.loss, agent.py:351-381
    mean, std = self.actor(feat_of(
        {k: traj[k].detach() for k in ('deter', 'stoch')}))
    ca = cfg['actor']
    if self.discrete:  # 'reinforce', agent.py:357-358; entropy :372-377
      logp = (traj['action'].detach() * torch.log_softmax(mean, -1)).sum(-1)
      loss = -logp[:-1] * score.detach()
      ll = torch.log_softmax(mean, -1)
      ent = -(torch.exp(ll) * ll).sum(-1)[:-1]
      if cfg['actent_norm']:  # minent 0, maxent log(A), nets.py:489-490
        ent = (ent - 0.0) / (math.log(self.act_dim) - 0.0)
      ent_loss, mets = self.actent(ent)
      metrics.update({f'actent_{k}': v.detach() for k, v in mets.items()})
      loss = loss + ent_loss
    else:  # 'backprop', agent.py:355-356; entropy :361-371
      loss = -score
      ent = normal_entropy(std)[:-1]
      if cfg['actent_norm']:  # :364-367, minent/maxent nets.py:466-467
        lo = normal_entropy(ca['minstd'])
        hi = normal_entropy(ca['maxstd'])
        ent = (ent - lo) / (hi - lo)
      ent_loss, mets = self.actent(ent)
      metrics.update({f'actent_{k}': v.detach() for k, v in mets.items()})
      loss = loss + ent_loss.sum(-1)
    loss = loss * traj['weight'].detach()[:-1]
    actor_loss = loss.mean()
    actor_names = [k for k in self.p if k.startswith('actor/')]
    mets, g_actor = self.actor_opt(actor_loss, self.p, actor_names,
                                   world_grads)
    metrics.update(mets)
    self.last = dict(
        grads={**g_model, **g_critic, **g_actor}, wm=wm_out, traj=traj,
        target=target, reward=reward, score=score, model_loss=model_loss,
        critic_loss=critic_loss, actor_loss=actor_loss)
    metrics = {k: (v.detach() if torch.is_tensor(v) else torch.tensor(v))
               for k, v in metrics.items()}
    return {}, state, metrics

  def policy(self, obs, state, noise, mode='train', forced=None):  # agent.py:42-65
    """noise: u_prior / u_post [n, G] uniforms of the obs_step, eps [n, A] (continuous) or
    u_act [n] (discrete) for the action sample, act_noise ([n, A] normals / [n] uniforms) for
    tfutils.action_noise when expl_noise / eval_noise is set.  forced: optional device draws
    {'post': [n, G], 'act': [n], 'act_noise': [n]} adopted near CDF edges (sample_onehot)."""
    obs = self.preprocess(obs)
    forced = forced or {}
    n = len(obs['is_first'])
    if state is None:
      latent = self.rssm.initial(n)
      action = torch.zeros(n, self.act_dim, dtype=self.dtype)
    else:
      latent, action = state
    lead1 = {k: v[:, None] for k, v in obs.items()}
    embed = self.encoder(lead1)[:, 0]
    u = torch.tensor(np.asarray(noise['u_post']), dtype=self.dtype)
    up = torch.tensor(np.asarray(noise['u_prior']), dtype=self.dtype)
    latent, _, _, _ = self.rssm.obs_step(
        latent, action, embed, obs['is_first'], up, u, None, forced.get('post'))
    latent = {k: v.detach() for k, v in latent.items()}
    mean, std = self.actor(feat_of(latent))
    amount = self.cfg['eval_noise'] if mode == 'eval' else self.cfg['expl_noise']
    if self.discrete:
      if mode == 'eval':
        action = onehot_mode(mean)
      else:
        ua = torch.tensor(np.asarray(noise['u_act']), dtype=self.dtype)
        action, _ = onehot_straight_through(mean, ua, forced.get('act'))
      if amount:  # tfutils.action_noise, tfutils.py:89-91
        un = torch.tensor(np.asarray(noise['act_noise']), dtype=self.dtype)
        probs = amount / action.shape[-1] + (1 - amount) * action.detach()
        idx = sample_onehot(probs, un, forced.get('act_noise'))
        action = F.one_hot(idx, action.shape[-1]).to(self.dtype)
    else:
      if mode == 'eval':
        action = mean  # Normal.mode()
      else:
        eps = torch.tensor(np.asarray(noise['eps']), dtype=self.dtype)
        action = mean + std * eps
      if amount:  # tfutils.py:92-93
        en = torch.tensor(np.asarray(noise['act_noise']), dtype=self.dtype)
        action = torch.clamp(action + amount * en, -1.0, 1.0)
    return {'action': action.detach()}, (latent, action.detach())

  def report(self, data, noise, forced=None):
    """Agent.report (agent.py:95-106) = WorldModel.report (agent.py:266-282) + Greedy.report
    under the `task_` prefix (behaviors.py:32-46).  noise: u_obs_prior / u_obs_post [T,B,G]
    for the observe pass (reused by the reference's second observe over [:6, :5]: same draws),
    u_openl [T-5, 6, G] for RSSM.imagine's prior samples, and for the policy rollout
    u_img [H, 6, G] with eps_act [H+1, 6, A] (or u_act [H+1, 6]).  forced: optional device
    draws {obs_prior, obs_post, openl, img, act}."""
    cfg = self.cfg
    forced = forced or {}
    noise = {k: torch.tensor(np.asarray(v), dtype=self.dtype) for k, v in noise.items()}
    with torch.no_grad():
      data = self.preprocess(data)
      _, _, wm_out, mets = self.wm_loss(data, None, noise, forced or None, update=False)
      report = {k: v for k, v in mets.items()}
      post = wm_out['post']
      n, ctx = min(6, data['is_first'].shape[0]), 5
      T = data['is_first'].shape[1]
      if not self.dec_cnn or T <= ctx:
        return report
      context = {k: v[:n, :ctx] for k, v in post.items()}
      start = {k: v[:, -1] for k, v in context.items()}
      recon = self.decoder(feat_of(context))
      # RSSM.imagine, nets.py:78-86: prior rollout with the recorded actions
      state, states = start, []
      for i, t in enumerate(range(ctx, T)):
        f = forced.get('openl')
        state, _ = self.rssm.img_step(state, data['action'][:n, t], noise['u_openl'][i],
                                      None if f is None else f[i])
        states.append(state)
      prior = {k: torch.stack([s[k] for s in states], 1) for k in start}
      openl = self.decoder(feat_of(prior))
      for key in self.dec_cnn:
        truth = data[key][:n]
        model = torch.cat([recon[key][:, :ctx], openl[key]], 1)
        error = (model - truth + 1) / 2
        report[f'openl_{key}'] = video_grid(torch.cat([truth, model, error], 2))
      # Greedy.report, behaviors.py:32-46: imagined rollout of the policy
      H = cfg['imag_horizon']
      first_cont = 1.0 - data['is_terminal'][:n, ctx - 1]
      traj = self.imagine(start, first_cont, noise, H, forced if forced else None)
      dists = self.decoder(feat_of({k: traj[k] for k in ('deter', 'stoch')}))
      for key in self.dec_cnn:
        report[f'task_imag_{key}'] = video_grid(dists[key].permute(1, 0, 2, 3, 4))
    return report

  def export_params(self):
    return {k: v.detach().numpy().copy() for k, v in self.p.items()}
