"""Independent cross-checks of the CPU oracle (oracle/dreamer_ref.py).

The oracle is parity-unpinned (the reference holds no golden vectors for this path and
TensorFlow cannot run here), so its weakest links are the TF semantics it assumes from the
documentation (dreamer_ref.py header).  Two of them are pinned here without PyTorch:

  * tf.nn.conv2d / tf.nn.conv2d_transpose as the reference calls them (nets.py:539, :547;
    NHWC, stride 2, VALID; filters [kh,kw,in,out] and [kh,kw,out,in]) restated as direct numpy
    loops from the TF documentation's definitions - conv2d as the cross-correlation sum,
    conv2d_transpose as "the gradient of conv2d with respect to its input" - against the
    oracle's torch calls (F.conv2d / F.conv_transpose2d on permuted filters);
  * the oracle's gradients against float64 central finite differences of its own losses, on
    every parameter group whose gradient is a true derivative of the sampled loss value
    (decoder, reward / cont heads, critic).  The encoder / RSSM / actor gradients pass through
    the straight-through estimator of the categorical draw (tfutils.py:376-381), which by
    construction is NOT the derivative of the sampled value, so finite differences cannot
    check them; their building blocks are autograd of plain torch ops.
  * the residual nets (`cnn: resnet`, nets.py:330-391): tf.nn.conv2d with stride 1 and 'SAME'
    (odd kernel: floor(k/2) zeros on every side - TF pads total k-1, the smaller half first,
    equal halves for odd k), tf.nn.avg_pool 2x2 / stride 2 on even sides, tf.repeat on both
    image axes, and one whole pre-activation residual block, as direct numpy loops.
"""

import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'tests' / 'golden'))

from oracle import dreamer_ref
import make_golden as mg


def conv2d_loops(x, f, stride=2):
  """TF doc, tf.nn.conv2d (NHWC, VALID): output[b, i, j, k] =
  sum_{di, dj, q} input[b, stride * i + di, stride * j + dj, q] * filter[di, dj, q, k]."""
  n, h, w, cin = x.shape
  kh, kw, _, cout = f.shape
  ho, wo = (h - kh) // stride + 1, (w - kw) // stride + 1
  y = np.zeros((n, ho, wo, cout))
  for b in range(n):
    for i in range(ho):
      for j in range(wo):
        for di in range(kh):
          for dj in range(kw):
            y[b, i, j] += x[b, stride * i + di, stride * j + dj] @ f[di, dj]
  return y


def conv2d_transpose_loops(y, f, stride=2):
  """TF doc, tf.nn.conv2d_transpose: "the transpose (gradient) of conv2d" with `filter`
  [kh, kw, output_channels, in_channels]: the input gradient of a conv2d whose filter is `filter`
  read as [kh, kw, in = output_channels, out = in_channels].  Every input pixel scatters
  input[b, i, j, :] @ filter[di, dj, o, :] to output[b, stride * i + di, stride * j + dj, o];
  VALID output size stride * (in - 1) + k  (= 2 * in + k - 2, nets.py:530-533)."""
  n, h, w, cin = y.shape
  kh, kw, cout, _ = f.shape
  out = np.zeros((n, stride * (h - 1) + kh, stride * (w - 1) + kw, cout))
  for b in range(n):
    for i in range(h):
      for j in range(w):
        for di in range(kh):
          for dj in range(kw):
            out[b, stride * i + di, stride * j + dj] += f[di, dj] @ y[b, i, j]
  return out


def oracle_conv(x, f, transp):
  p = {'c/kernel': torch.tensor(f), 'c/bias': torch.zeros(f.shape[2] if transp else f.shape[3], dtype=torch.float64)}
  return dreamer_ref.conv2d(p, 'c', torch.tensor(x), transp).numpy()


def test_conv2d_matches_direct_loops():
  rng = np.random.RandomState(0)
  for (h, k, cin, cout) in ((10, 4, 3, 5), (9, 4, 2, 3), (6, 3, 4, 2)):
    x = rng.randn(2, h, h, cin)
    f = rng.randn(k, k, cin, cout)
    assert np.abs(oracle_conv(x, f, False) - conv2d_loops(x, f)).max() < 1e-12


def test_conv2d_transpose_matches_direct_loops_and_is_the_adjoint():
  rng = np.random.RandomState(1)
  for (h, k, cin, cout) in ((4, 5, 3, 2), (3, 6, 2, 3), (5, 4, 4, 1), (1, 5, 6, 4)):
    y = rng.randn(2, h, h, cin)
    f = rng.randn(k, k, cout, cin)               # [kh, kw, out, in]  (nets.py:523)
    up = conv2d_transpose_loops(y, f)
    assert up.shape[1] == 2 * h + k - 2          # nets.py:530-533
    assert np.abs(oracle_conv(y, f, True) - up).max() < 1e-12
    # adjoint identity: <conv2d(x, f), y> == <x, conv2d_transpose(y, f)> for f read as [kh,kw,in,out]
    x = rng.randn(*up.shape)
    lhs = (conv2d_loops(x, f) * y).sum()
    rhs = (x * up).sum()
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def conv2d_same_loops(x, f):
  """tf.nn.conv2d, stride 1, 'SAME' (NHWC): output[b, i, j, k] = sum_{di, dj, q}
  input[b, i + di - pt, j + dj - pl, q] * filter[di, dj, q, k] with the input zero outside the
  image; TF: pad_total = k - 1, pad_top = pad_left = pad_total // 2."""
  n, h, w, cin = x.shape
  kh, kw, _, cout = f.shape
  pt, pl = (kh - 1) // 2, (kw - 1) // 2
  y = np.zeros((n, h, w, cout))
  for b in range(n):
    for i in range(h):
      for j in range(w):
        for di in range(kh):
          for dj in range(kw):
            si, sj = i + di - pt, j + dj - pl
            if 0 <= si < h and 0 <= sj < w:
              y[b, i, j] += x[b, si, sj] @ f[di, dj]
  return y


def _ln_elu(x, scale, bias):   # Norm nets.py:594-600 (eps 1e-3, population variance) + tf.nn.elu
  m = x.mean(-1, keepdims=True)
  v = ((x - m) ** 2).mean(-1, keepdims=True)
  y = (x - m) / np.sqrt(v + 1e-3) * scale + bias
  return np.where(y > 0, y, np.exp(np.minimum(y, 0)) - 1)


def test_same_conv_pool_repeat_and_residual_block_match_direct_loops():
  rng = np.random.RandomState(5)
  t = lambda a: torch.tensor(a)
  for (h, k, cin, cout) in ((5, 3, 3, 4), (4, 1, 4, 2), (6, 3, 2, 2)):
    x, f, b = rng.randn(2, h, h, cin), rng.randn(k, k, cin, cout), rng.randn(cout)
    p = {'c/kernel': t(f), 'c/bias': t(b)}
    got = dreamer_ref.conv2d_same(p, 'c', t(x)).numpy()
    assert np.abs(got - (conv2d_same_loops(x, f) + b)).max() < 1e-12
  # one residual block with a 1x1 skip (nets.py:351-358): channels 3 -> 4 on 4x4 pixels
  cin, d, h = 3, 4, 4
  P = {'b/a/kernel': rng.randn(3, 3, cin, d), 'b/a/bias': rng.randn(d),
       'b/a/norm/scale': rng.rand(cin) + 0.5, 'b/a/norm/bias': rng.randn(cin),
       'b/b/kernel': rng.randn(3, 3, d, d), 'b/b/bias': rng.randn(d),
       'b/b/norm/scale': rng.rand(d) + 0.5, 'b/b/norm/bias': rng.randn(d),
       'b/s/kernel': rng.randn(1, 1, cin, d)}
  P = {k.replace('b/a', 'ba').replace('b/b', 'bb').replace('b/s', 'bs'): v for k, v in P.items()}
  x = rng.randn(2, h, h, cin)
  skip = conv2d_same_loops(x, P['bs/kernel'])
  y = conv2d_same_loops(_ln_elu(x, P['ba/norm/scale'], P['ba/norm/bias']), P['ba/kernel']) + P['ba/bias']
  y = conv2d_same_loops(_ln_elu(y, P['bb/norm/scale'], P['bb/norm/bias']), P['bb/kernel']) + P['bb/bias']
  want = skip + 0.1 * y
  got = dreamer_ref.res_block({k: t(v) for k, v in P.items()}, 'b', d, t(x), act='elu', norm='layer').numpy()
  assert np.abs(got - want).max() < 1e-11
  # encoder / decoder plumbing: avg_pool 2x2 (even sides: no padding) and tf.repeat by 2 on both axes
  img = rng.randn(2, 8, 8, 3)
  pooled = np.zeros((2, 4, 4, 3))
  rep = np.zeros((2, 16, 16, 3))
  for i in range(4):
    for j in range(4):
      pooled[:, i, j] = img[:, 2 * i:2 * i + 2, 2 * j:2 * j + 2].mean((1, 2))
  for i in range(16):
    for j in range(16):
      rep[:, i, j] = img[:, i // 2, j // 2]
  import torch.nn.functional as F
  tp = F.avg_pool2d(t(img).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).numpy()
  tr = t(img).repeat_interleave(2, 1).repeat_interleave(2, 2).numpy()
  assert np.abs(tp - pooled).max() < 1e-14 and np.array_equal(tr, rep)


def _losses(params, which):
  plain, sp, shapes, _, data, B, T = mg.build()
  H = plain['imag_horizon']
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64)
  noise = mg.golden_noise(B, T, H, sp.groups, sp.act_dim, 1)
  ag.train(data, noise, None)
  return float(ag.last[which].detach()), ag.last['grads']


def test_oracle_gradients_match_finite_differences():
  """Central differences (fp64, h = 1e-5) of model_loss w.r.t. decoder / reward / cont
  parameters and of critic_loss w.r.t. critic parameters against the oracle's autograd
  gradients, on the tiny golden problem."""
  _, _, _, params, _, _, _ = mg.build()
  params = {k: np.asarray(v, np.float64) for k, v in params.items()}
  _, grads = _losses(params, 'model_loss')
  rng = np.random.RandomState(3)
  picks = [('dec/cnn/out/kernel', 'model_loss'), ('dec/cnn/conv0/kernel', 'model_loss'),
           ('dec/cnn/conv1/norm/scale', 'model_loss'), ('dec/cnn/out/bias', 'model_loss'),
           ('reward/dense0/kernel', 'model_loss'), ('reward/dist_out/out/kernel', 'model_loss'),
           ('cont/dist_out/out/bias', 'model_loss'), ('cont/dense1/norm/bias', 'model_loss'),
           ('critic/dense0/kernel', 'critic_loss'), ('critic/dist_out/out/kernel', 'critic_loss'),
           ('critic/dense1/norm/scale', 'critic_loss')]
  h = 1e-5   # (model_loss ~ 2e3: rounding of the difference ~ 1e-16 * 2e3 / h = 2e-8)
  for name, which in picks:
    assert name in params, (name, sorted(params)[:5])
    g = grads[name].numpy()
    # the entry with the largest gradient magnitude (a meaningful relative error) and a random one
    flat = [int(np.abs(g).argmax()), int(rng.randint(g.size))]
    for i in flat:
      idx = np.unravel_index(i, g.shape)
      vals = []
      for sgn in (+1, -1):
        q = dict(params)
        w = params[name].copy()
        w[idx] += sgn * h
        q[name] = w
        vals.append(_losses(q, which)[0])
      fd = (vals[0] - vals[1]) / (2 * h)
      assert abs(fd - g[idx]) <= 1e-4 * abs(g[idx]) + 2e-7, (name, idx, fd, g[idx])


def _resnet_problem():
  import helpers
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=2, replay_chunk=3, imag_horizon=2)
  cfg = cfg.update({'encoder.cnn': 'resnet', 'decoder.cnn': 'resnet', 'encoder.cnn_depth': 2,
                    'decoder.cnn_depth': 2, 'encoder.cnn_blocks': 1, 'decoder.cnn_blocks': 1})
  return helpers.make_problem(cfg, image=16, vector=5, action=3, terminals=0.2)


def test_oracle_resnet_decoder_gradients_match_finite_differences():
  """The residual decoder's restatement (dreamer_ref.decoder_resnet: Linear, pre-activation
  blocks with and without the 1x1 skip, repetition, output convolution) against central
  differences of model_loss, float64, on a 16x16 image (two stages)."""
  plain, sp, shapes, params, data, B, T = _resnet_problem()
  H = plain['imag_horizon']
  params = {k: np.asarray(v, np.float64) for k, v in params.items()}
  noise = mg.golden_noise(B, T, H, sp.groups, sp.act_dim, 1)

  def loss(p):
    ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, p, torch.float64)
    ag.train(data, noise, None)
    return float(ag.last['model_loss'].detach()), ag.last['grads']

  _, grads = loss(params)
  rng = np.random.RandomState(4)
  h = 1e-5
  for name in ('dec/cnn/in/kernel', 'dec/cnn/in/bias', 'dec/cnn/s0b0a/kernel', 'dec/cnn/s0b0a/norm/scale',
               'dec/cnn/s0b0b/bias', 'dec/cnn/s1b0s/kernel', 'dec/cnn/s1b0b/kernel',
               'dec/cnn/s1b0a/norm/bias', 'dec/cnn/out/kernel', 'dec/cnn/out/bias'):
    assert name in params, (name, [k for k in params if k.startswith('dec/cnn')])
    g = grads[name].numpy()
    for i in (int(np.abs(g).argmax()), int(rng.randint(g.size))):
      idx = np.unravel_index(i, g.shape)
      vals = []
      for sgn in (+1, -1):
        q = dict(params)
        w = params[name].copy()
        w[idx] += sgn * h
        q[name] = w
        vals.append(loss(q)[0])
      fd = (vals[0] - vals[1]) / (2 * h)
      assert abs(fd - g[idx]) <= 1e-4 * abs(g[idx]) + 2e-7, (name, idx, fd, g[idx])
