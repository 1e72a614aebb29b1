"""Every TF / TFP semantics assumption of the CPU oracle (oracle/dreamer_ref.py header), pinned
WITHOUT PyTorch: each is restated here as scalar Python / numpy loops written from the formula
the TensorFlow or TensorFlow-Probability documentation gives (or, for the reference's own
hand-written code - Adam, AutoAdapt, Normalize, the lambda-return - from the reference lines
cited), and compared with the oracle function the learner parity tests rely on.

This does not pin the oracle against the reference's OUTPUTS (impossible here: no TensorFlow, no
golden vectors in the reference; the oracle header and DESIGN.md say "parity unpinned"); it removes
the possibility that the oracle and the HIP path agree with each other through a shared PyTorch
idiom that differs from the documented TF behaviour.  tf.nn.conv2d / conv2d_transpose / SAME
convolution / pooling / repetition and the autograd-vs-finite-difference checks live in
tests/test_oracle_independent.py.
"""

import math

import numpy as np
import torch

from oracle import dreamer_ref as R

RNG = np.random.RandomState(7)
T64 = lambda a: torch.tensor(np.asarray(a, np.float64))


def close(a, b, tol=1e-12):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  assert a.shape == b.shape, (a.shape, b.shape)
  assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()


# ---- tf.nn.moments + tf.nn.batch_normalization (nets.py:594-600), tf.nn.elu -----------------

def test_layer_norm_is_population_moments_and_batch_normalization():
  """tf.nn.moments(x, -1): mean = sum(x) / n, variance = sum((x - mean)^2) / n  (population,
  not the n - 1 sample form); tf.nn.batch_normalization(x, mean, var, offset, scale, eps) =
  (x - mean) * scale / sqrt(var + eps) + offset, eps = 1e-3 (nets.py:600)."""
  x = RNG.randn(5, 13) * 3 + 1
  scale, offset = RNG.randn(13), RNG.randn(13)
  want = np.zeros_like(x)
  for r in range(x.shape[0]):
    n = x.shape[1]
    mean = sum(x[r, c] for c in range(n)) / n
    var = sum((x[r, c] - mean) ** 2 for c in range(n)) / n
    for c in range(n):
      want[r, c] = (x[r, c] - mean) * scale[c] / math.sqrt(var + 1e-3) + offset[c]
  close(R.layer_norm(T64(x), T64(scale), T64(offset)).numpy(), want)
  # the sample-variance form must NOT match (the check can tell the two apart)
  sample_var = x.var(-1, ddof=1, keepdims=True)
  other = (x - x.mean(-1, keepdims=True)) / np.sqrt(sample_var + 1e-3) * scale + offset
  assert np.abs(other - want).max() > 1e-3


def test_elu():
  """tf.nn.elu: x for x > 0, exp(x) - 1 otherwise (alpha = 1)."""
  x = np.concatenate([RNG.randn(40) * 3, [0.0, -0.0, 1e-9, -1e-9, 30.0, -30.0]])
  want = np.array([v if v > 0 else math.expm1(v) for v in x])
  close(R.get_act('elu')(T64(x)).numpy(), want)


# ---- symlog / symexp (tfutils.py:77-82) ------------------------------------------------------

def test_symlog_symexp():
  x = np.concatenate([RNG.randn(50) * 20, [0.0, 1.0, -1.0, 1e-12, -1e6]])
  sign = lambda v: float(v > 0) - float(v < 0)
  close(R.symlog(T64(x)).numpy(), [sign(v) * math.log(1 + abs(v)) for v in x])
  y = np.clip(x, -20, 20)
  close(R.symexp(T64(y)).numpy(), [sign(v) * (math.exp(abs(v)) - 1) for v in y], 1e-11)
  close(R.symexp(R.symlog(T64(x))).numpy(), x, 1e-9)


# ---- OneHotCategorical: KL, entropy, mode (nets.py:88-91, 178-183) -----------------------------

def _softmax_loop(logits):
  m = max(logits)
  e = [math.exp(v - m) for v in logits]
  z = sum(e)
  return [v / z for v in e]


def test_categorical_kl_and_entropy():
  """tfd.kl_divergence(OneHotCategorical(logits=a), OneHotCategorical(logits=b)) =
  sum_c p_c (log p_c - log q_c) with p = softmax(a), q = softmax(b); Independent(..., 1) sums it
  over the groups; entropy = - sum_c p_c log p_c."""
  a, b = RNG.randn(4, 3, 6) * 2, RNG.randn(4, 3, 6) * 2
  kl, ent = np.zeros(4), np.zeros(4)
  for r in range(4):
    for g in range(3):
      p, q = _softmax_loop(list(a[r, g])), _softmax_loop(list(b[r, g]))
      kl[r] += sum(pc * (math.log(pc) - math.log(qc)) for pc, qc in zip(p, q))
      ent[r] -= sum(pc * math.log(pc) for pc in p)
  close(R.categorical_kl(T64(a), T64(b)).numpy(), kl)
  close(R.categorical_entropy(T64(a)).numpy(), ent)
  assert (kl >= 0).all()
  close(R.categorical_kl(T64(a), T64(a)).numpy(), np.zeros(4))


def test_onehot_mode_and_straight_through_sample():
  """.mode() = one_hot(argmax); OneHotDist.sample (tfutils.py:368-382) = one_hot(draw) +
  probs - stop_gradient(probs): the VALUE is the one-hot draw, the GRADIENT that of softmax."""
  logit = RNG.randn(7, 5)
  mode = R.onehot_mode(T64(logit)).numpy()
  for r in range(7):
    k = max(range(5), key=lambda c: logit[r, c])
    assert mode[r].tolist() == [1.0 if c == k else 0.0 for c in range(5)]
  u = RNG.rand(7)
  lt = T64(logit).requires_grad_(True)
  sample, idx = R.onehot_straight_through(lt, T64(u))
  # inverse-CDF draw restated: first class whose cumulative probability exceeds u * total
  for r in range(7):
    p = _softmax_loop(list(logit[r]))
    cdf, k = 0.0, 4
    tot = sum(p)
    acc = []
    for c in range(5):
      cdf += p[c]
      acc.append(cdf)
    k = sum(1 for c in range(4) if acc[c] <= u[r] * acc[4])
    assert int(idx[r]) == k
    close(sample[r].detach().numpy(), [1.0 if c == k else 0.0 for c in range(5)], 1e-15)
  w = RNG.randn(7, 5)
  (sample * T64(w)).sum().backward()
  want = np.zeros((7, 5))
  for r in range(7):   # d/dlogit_j of sum_c w_c softmax_c = p_j (w_j - sum_c w_c p_c)
    p = _softmax_loop(list(logit[r]))
    dot = sum(w[r, c] * p[c] for c in range(5))
    want[r] = [p[j] * (w[r, j] - dot) for j in range(5)]
  close(lt.grad.numpy(), want, 1e-12)


# ---- Bernoulli, Normal -------------------------------------------------------------------------

def test_bernoulli_log_prob_and_mean():
  """tfd.Bernoulli(logits=l): P(1) = sigmoid(l) = 1 / (1 + exp(-l)); log_prob(x) =
  log(P(1)^x * P(0)^(1-x)) (also for the soft targets x in (0, 1) the reference feeds it);
  .mean() = P(1)."""
  l = np.concatenate([RNG.randn(30) * 4, [0.0, 20.0, -20.0]])
  x = np.concatenate([RNG.randint(0, 2, 30).astype(np.float64), [0.3, 1.0, 0.0]])
  want = []
  for lv, xv in zip(l, x):
    p1 = 1.0 / (1.0 + math.exp(-lv))
    want.append(xv * math.log(p1) + (1 - xv) * math.log(1.0 - p1))
  close(R.bernoulli_log_prob(T64(l), T64(x)).numpy(), want, 1e-9)
  close(torch.sigmoid(T64(l)).numpy(), [1.0 / (1.0 + math.exp(-v)) for v in l])


def test_normal_entropy_and_reparameterised_sample():
  """tfd.Normal(mu, sigma).entropy() = 0.5 * log(2 pi e sigma^2); .sample() = mu + sigma * eps
  with eps ~ N(0, 1) (reparameterised: d sample / d mu = 1, d sample / d sigma = eps) - the
  oracle takes eps explicitly (RefAgent.imagine: mean + std * eps)."""
  std = np.abs(RNG.randn(20)) + 0.1
  close(R.normal_entropy(T64(std)).numpy(), [0.5 * math.log(2 * math.pi * math.e * s * s) for s in std])
  assert abs(R.normal_entropy(0.1) - 0.5 * math.log(2 * math.pi * math.e * 0.01)) < 1e-15
  # numerical check of the closed form: - integral of pdf * log pdf
  s = 0.7
  xs = np.linspace(-12 * s, 12 * s, 200001)
  pdf = np.exp(-xs ** 2 / (2 * s * s)) / math.sqrt(2 * math.pi * s * s)
  trap = getattr(np, 'trapezoid', None) or np.trapz
  num = -trap(pdf * np.log(pdf + 1e-300), xs)
  assert abs(num - float(R.normal_entropy(T64([s]))[0])) < 1e-8


# ---- trajectory weights and the lambda-return (agent.py:256-259, 422-442) ---------------------

def test_discount_weights_are_cumprod_over_time():
  """tf.math.cumprod(discount * cont) / discount along axis 0 (inclusive product)."""
  cont = RNG.rand(6, 4)
  disc = 0.997
  want = np.zeros_like(cont)
  for n in range(4):
    prod = 1.0
    for t in range(6):
      prod *= disc * cont[t, n]
      want[t, n] = prod / disc
  close(R.discount_weights(T64(cont), disc).numpy(), want)


def test_lambda_return_recurrences():
  """VFunction.target, agent.py:422-442, restated per trajectory with scalar recursions."""
  H, N, lam = 7, 5, 0.95
  reward, value, disc = RNG.randn(H, N), RNG.randn(H + 1, N), RNG.rand(H, N)
  gve, gae = np.zeros((H, N)), np.zeros((H, N))
  for n in range(N):
    nxt = value[H, n]                       # vals = [value[-1]]
    for t in reversed(range(H)):
      nxt = reward[t, n] + disc[t, n] * value[t + 1, n] * (1 - lam) + disc[t, n] * lam * nxt
      gve[t, n] = nxt
    adv = 0.0                               # advs = [zeros]
    for t in reversed(range(H)):
      delta = reward[t, n] + disc[t, n] * value[t + 1, n] - value[t, n]
      adv = delta + disc[t, n] * lam * adv
      gae[t, n] = adv + value[t, n]
  for impl, want in (('gve', gve), ('gae', gae)):
    ret, base = R.lambda_return(T64(reward), T64(value), T64(disc), lam, impl)
    close(ret.numpy(), want)
    close(base.numpy(), value[:-1])
  # lambda = 1: both are the discounted Monte-Carlo return bootstrapped with value[H]
  ret1, _ = R.lambda_return(T64(reward), T64(value), T64(disc), 1.0, 'gve')
  ret2, _ = R.lambda_return(T64(reward), T64(value), T64(disc), 1.0, 'gae')
  close(ret1.numpy(), ret2.numpy(), 1e-12)


# ---- Optimizer: global norm, clipping, the literal Adam, weight decay (tfutils.py:143-302) ----

def test_global_norm_clip_and_literal_adam():
  shapes = [(3, 4), (5,), (2, 2, 2)]
  grads = [RNG.randn(*s) for s in shapes]
  norm = math.sqrt(sum(float(v) ** 2 for g in grads for v in g.reshape(-1)))
  assert abs(float(R.global_norm([T64(g) for g in grads])) - norm) < 1e-12
  for clip in (0.5 * norm, 2.0 * norm):   # clipping active / inactive
    got = R.clip_by_global_norm([T64(g) for g in grads], clip, T64(norm))
    # tf.clip_by_global_norm doc: t_list[i] * clip_norm / max(global_norm, clip_norm)
    for g, q in zip(grads, got):
      close(q.numpy(), g * clip / max(norm, clip))
  # _apply_adam, tfutils.py:271-283, three steps on one tensor, scalar loops
  lr, eps, b1, b2 = 3e-4, 1e-5, 0.9, 0.999
  p0 = RNG.randn(6)
  gs = [RNG.randn(6) for _ in range(3)]
  p, m, v = list(p0), [0.0] * 6, [0.0] * 6
  for t, g in enumerate(gs, 1):
    for i in range(6):
      m[i] = b1 * m[i] + (1. - b1) * g[i]
      v[i] = b2 * v[i] + (1. - b2) * g[i] * g[i]
      m_hat = m[i] / (1. - b1 ** t)
      v_hat = v[i] / (1. - b2 ** t)
      p[i] -= lr * m_hat / (math.sqrt(v_hat) + eps)
  pt, mt, vt = T64(p0).clone(), torch.zeros(6, dtype=torch.float64), torch.zeros(6, dtype=torch.float64)
  for t, g in enumerate(gs, 1):
    mt, vt = R.adam_update(pt, T64(g), mt, vt, float(t), lr, eps)
  close(pt.numpy(), p)
  close(mt.numpy(), m)
  close(vt.numpy(), v)


def test_optimizer_order_clip_then_decay_then_adam():
  """Optimizer.__call__ (tfutils.py:205-266) on a quadratic: gradient -> global norm -> clip ->
  weight decay on names matching the pattern ((1 - wd * lr) * param, BEFORE Adam) -> step += 1 ->
  Adam with the clipped gradient."""
  lr, eps, wd, clip = 1e-2, 1e-5, 0.1, 0.3
  w0, b0 = RNG.randn(4), RNG.randn(3)
  params = {'dense/kernel': T64(w0).clone().requires_grad_(True),
            'dense/bias': T64(b0).clone().requires_grad_(True)}
  opt = R.Optimizer('model', lr, eps=eps, clip=clip, wd=wd, wd_pattern='kernel')
  loss = (params['dense/kernel'] ** 2).sum() + (3 * params['dense/bias']).sum()
  mets, raw = opt(loss, params, list(params))
  gk, gb = 2 * w0, np.full(3, 3.0)
  norm = math.sqrt(float((gk ** 2).sum() + (gb ** 2).sum()))
  assert abs(float(mets['model_grad_norm']) - norm) < 1e-12
  scale = clip / max(norm, clip)
  for name, p_old, g, decayed in (('dense/kernel', w0, gk, True), ('dense/bias', b0, gb, False)):
    want = []
    for i in range(len(p_old)):
      p = p_old[i] * (1 - wd * lr) if decayed else p_old[i]
      gi = g[i] * scale
      m = 0.1 * gi
      v = 0.001 * gi * gi
      p -= lr * (m / (1 - 0.9)) / (math.sqrt(v / (1 - 0.999)) + eps)
      want.append(p)
    close(params[name].detach().numpy(), want)
    close(raw[name].numpy(), g)               # the reported gradient is the UNCLIPPED one
  assert int(mets['model_grad_steps']) == 1


# ---- AutoAdapt (tfutils.py:414-482) and Normalize (tfutils.py:485-527) -----------------------

def test_autoadapt_mult_updates_before_use():
  """__call__ runs update(reg) FIRST and scales the loss with the UPDATED scale (:440-442);
  'mult': scale *= (1 + vel) above (1 + thres) * target, /= (1 + vel) below target / (1 + thres),
  clipped to [min, max]; `inverse` swaps the directions and negates the regulariser."""
  for inverse in (False, True):
    aa = R.AutoAdapt((), 'mult', 1.0, target=0.5, min=1e-3, max=2.0, vel=0.1, thres=0.1, inverse=inverse)
    scale = 1.0
    for avg in (0.9, 0.9, 0.52, 0.1, 0.1, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9):
      reg = T64(np.full((3, 2), avg))
      below, above = avg < 0.5 / 1.1, avg > 0.5 * 1.1
      if inverse:
        below, above = above, below
      if above:
        scale = scale * 1.1
      elif below:
        scale = scale / 1.1
      scale = min(max(scale, 1e-3), 2.0)
      loss, mets = aa(reg)
      close(loss.numpy(), np.full((3, 2), scale * (-avg if inverse else avg)), 1e-6)
      assert abs(float(mets['scale_mean']) - scale) < 1e-6 and abs(float(mets['mean']) - avg) < 1e-12


def test_autoadapt_prop_and_fixed():
  aa = R.AutoAdapt((), 'prop', 1.0, target=0.5, min=0.0, max=3.0, vel=0.1)
  scale = 1.0
  for avg in (0.9, 0.2, 5.0, 30.0):
    scale = min(max(scale + 0.1 * (avg - 0.5), 0.0), 3.0)
    loss, _ = aa(T64([avg]))
    assert abs(float(loss[0]) - scale * avg) < 1e-5
  fx = R.AutoAdapt((), 'fixed', 0.25, target=0.5, min=0.0, max=3.0)
  for avg in (0.9, 0.2):
    assert abs(float(fx(T64([avg]))[0][0]) - 0.25 * avg) < 1e-7


def test_normalize_bias_correction():
  """Normalize 'mean_std' (:498-527): EMA of mean and of squares in float64, both divided by
  1 - decay^step before use; scale = rsqrt(max(var, 1 / max^2 + vareps) + stdeps); update(values)
  runs before transform(values); update=False transforms with the state as it is."""
  decay, maxv = 0.99, 1e8
  nz = R.Normalize('mean_std', decay, maxv)
  mean = sqrs = 0.0
  for step in range(1, 6):
    x = RNG.randn(4, 3) * (1 + step) + step
    mean = decay * mean + (1 - decay) * float(x.mean())
    sqrs = decay * sqrs + (1 - decay) * float((x ** 2).mean())
    corr = 1 - decay ** step
    mu = mean / corr
    var = sqrs / corr - mu ** 2
    scale = 1.0 / math.sqrt(max(var, 1 / maxv ** 2))
    close(nz(T64(x)).numpy(), (x - mu) * scale, 1e-10)
  y = RNG.randn(5)
  close(nz(T64(y), update=False).numpy(), (y - mu) * scale, 1e-10)
  # first step: mean / (1 - decay) = the batch mean itself, variance = the batch's population variance
  nz1 = R.Normalize('mean_std', decay, maxv)
  x = RNG.randn(50) * 2 + 3
  close(nz1(T64(x)).numpy(), (x - x.mean()) / x.std(), 1e-9)
  sd = R.Normalize('std', decay, maxv)
  close(sd(T64(x)).numpy(), x / math.sqrt((x ** 2).mean() - x.mean() ** 2), 1e-9)


def test_balance_stats():
  """tfutils.py:395-411: positive / negative split of a target at a threshold."""
  target = np.array([0.0, 0.2, 0.05, 1.0, 0.0, 0.5])
  mean = np.array([0.3, 0.0, 0.2, 0.9, 0.05, 0.02])
  lp = -(mean - target) ** 2
  out = R.balance_stats(T64(mean), lambda t: T64(lp), T64(target), 0.1)
  pos = [i for i in range(6) if target[i] > 0.1]
  neg = [i for i in range(6) if target[i] <= 0.1]
  assert abs(float(out['pos_loss']) - sum(-lp[i] for i in pos) / len(pos)) < 1e-12
  assert abs(float(out['neg_loss']) - sum(-lp[i] for i in neg) / len(neg)) < 1e-12
  assert abs(float(out['pos_acc']) - sum(mean[i] > 0.1 for i in pos) / len(pos)) < 1e-12
  assert abs(float(out['neg_acc']) - sum(mean[i] <= 0.1 for i in neg) / len(neg)) < 1e-12
  assert abs(float(out['rate']) - len(pos) / 6) < 1e-12
