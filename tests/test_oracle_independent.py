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
