"""The TensorFlow stand-in (oracle/tf_on_torch.py) on its own: the library semantics it
substitutes under the reference's sources, each against a loop-form / closed-form statement of
the TensorFlow documentation, and the fixtures' provenance (regenerated from the reference
checkout when it is present and compared with the committed files).

Nothing here calls `install()`: the functions are used directly, so the test process never gets a
fake `tensorflow` module (the generator script runs in its own process).
"""

import math
import pathlib
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import tf_on_torch as tft

HERE = pathlib.Path(__file__).parent
REF = pathlib.Path('/root/reference')


def T(a):
  return tft.tensor_of(np.asarray(a))


def test_shape_is_a_tensorshape_as_far_as_the_sources_need():
  x = T(np.zeros((2, 3, 4), np.float32))
  assert x.dtype == torch.float64                      # every float is the one float type
  assert x.shape[:-1] + [5, 6] == (2, 3, 5, 6)         # nets.py:164: TensorShape + list
  assert x.shape[-2:] == (3, 4) and x.shape[-2:] == [3, 4] and not (x.shape == (2, 3))
  assert isinstance(x.shape[1:], tft.Shape) and x.shape[0] == 2 and len(x.shape) == 3
  assert int(np.prod(x.shape[1:])) == 12 and (7,) + x.shape == (7, 2, 3, 4)


def test_tensors_are_immutable_and_variables_assign():
  a = T([1.0, 2.0])
  b = a
  a += 1.0                                             # tf: rebinds; torch would modify b as well
  assert b.numpy().tolist() == [1.0, 2.0] and a.numpy().tolist() == [2.0, 3.0]
  v = tft.Variable(np.array([1.0, 2.0], np.float32), name='v')
  assert v.trainable and v.requires_grad and v.name == 'v:0'
  y = (v * v).sum()
  (g,) = tft.GradientTape().gradient(y, [v])
  assert g.numpy().tolist() == [2.0, 4.0]
  v.assign_sub(T([0.5, 0.5]))
  assert v.numpy().tolist() == [0.5, 1.5]
  v.reset()
  assert v.numpy().tolist() == [1.0, 2.0]
  step = tft.Variable(0, trainable=False, dtype=torch.int64)
  step.assign_add(1)
  assert int(step.numpy()) == 1 and not step.requires_grad
  assert type(v * 2) is tft.Tensor                     # results of ops on variables are tensors


def test_reductions():
  x = np.arange(24, dtype=np.float64).reshape(2, 3, 4) ** 1.5
  t = T(x)
  assert np.allclose(tft.reduce_mean(t, [0, 1]).numpy(), x.mean((0, 1)))
  assert np.allclose(tft.reduce_std(t).numpy(), x.std())                      # population
  assert np.allclose(tft.reduce_variance(t, -1).numpy(), x.var(-1))
  assert tft.reduce_mean(t, []).shape == (2, 3, 4)                            # axis=[]: nothing reduced
  assert float(tft.reduce_std(T(3.0)).numpy()) == 0.0                        # of a scalar
  assert float(tft.reduce_mean(T(3.0)).numpy()) == 3.0
  assert np.allclose(tft.reduce_sum([t, t], 0).numpy(), 2 * x)                # a list of tensors
  assert np.allclose(tft.reduce_max(t, (-1, -2)).numpy(), x.max((-1, -2)))
  assert np.allclose(tft.reduce_prod(T(x[:, :, :2] + 1), [1, 2]).numpy(), (x[:, :, :2] + 1).prod((1, 2)))
  assert bool(tft.reduce_all([T(True), T(True)]).numpy()) and not bool(tft.reduce_all([T(True), T(False)]).numpy())
  # bound as methods, as the reference does (tfutils.py:24-39)
  class X(tft.Tensor):
    pass
  X.mean = tft.reduce_mean
  assert np.allclose(t.as_subclass(X).mean([0]).numpy(), x.mean(0))


def test_conv2d_and_transpose_against_direct_loops():
  rng = np.random.RandomState(0)
  x, w = rng.randn(2, 7, 7, 3), rng.randn(4, 4, 3, 5)
  y = tft.conv2d(T(x), T(w), 2, 'VALID').numpy()
  want = np.zeros((2, 2, 2, 5))
  for i in range(2):
    for j in range(2):
      want[:, i, j] = np.einsum('nabc,abco->no', x[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4], w)
  assert np.allclose(y, want)
  # SAME, stride 1, odd filter: floor(k/2) zeros per side
  w3 = rng.randn(3, 3, 3, 2)
  y = tft.conv2d(T(x), T(w3), 1, 'SAME').numpy()
  xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
  want = np.zeros((2, 7, 7, 2))
  for i in range(7):
    for j in range(7):
      want[:, i, j] = np.einsum('nabc,abco->no', xp[:, i:i + 3, j:j + 3], w3)
  assert np.allclose(y, want)
  # transpose: filter [kh, kw, out, in], scatter form, no flip; VALID output = stride * (in - 1) + k
  s, wt = rng.randn(2, 3, 3, 5), rng.randn(4, 4, 6, 5)
  y = tft.conv2d_transpose(T(s), T(wt), (2, 8, 8, 6), 2, 'VALID').numpy()
  want = np.zeros((2, 8, 8, 6))
  for i in range(3):
    for j in range(3):
      want[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4] += np.einsum('nc,aboc->nabo', s[:, i, j], wt)
  assert np.allclose(y, want)
  with pytest.raises(AssertionError):
    tft.conv2d_transpose(T(s), T(wt), (2, 9, 9, 6), 2, 'VALID')


def test_moments_batch_normalization_pool_repeat_split():
  rng = np.random.RandomState(1)
  x = rng.randn(3, 6)
  m, v = tft.moments(T(x), -1, keepdims=True)
  assert np.allclose(m.numpy(), x.mean(-1, keepdims=True)) and np.allclose(v.numpy(), x.var(-1, keepdims=True))
  g, b = rng.randn(6), rng.randn(6)
  y = tft.batch_normalization(T(x), m, v, T(b), T(g), 1e-3).numpy()
  assert np.allclose(y, (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-3) * g + b)
  img = rng.randn(1, 4, 4, 2)
  p = tft.avg_pool(T(img), [2, 2], [2, 2], 'SAME').numpy()
  assert np.allclose(p[0, 1, 0], img[0, 2:4, 0:2].mean((0, 1)))
  r = tft.repeat(tft.repeat(T(img), 2, 1), 2, 2).numpy()
  assert r.shape == (1, 8, 8, 2) and np.array_equal(r[0, 3, 5], img[0, 1, 2])
  a, c = tft.split(T(x), [2, 4], -1)
  assert a.shape == (3, 2) and c.shape == (3, 4)
  assert [t.shape for t in tft.split(T(x), 3, -1)] == [(3, 2)] * 3


def test_clip_by_global_norm_and_scan_and_nest():
  g = [T([3.0, 0.0]), T([[0.0, 4.0]])]
  clipped, norm = tft.clip_by_global_norm(g, 2.5)
  assert float(norm.numpy()) == 5.0 and np.allclose(clipped[0].numpy(), [1.5, 0.0]) and np.allclose(clipped[1].numpy(), [[0.0, 2.0]])
  same, _ = tft.clip_by_global_norm(g, 100.0, norm)
  assert np.allclose(same[1].numpy(), [[0.0, 4.0]])                           # below the clip: unchanged
  out = tft.scan(lambda acc, x: {'s': acc['s'] + x, 'p': acc['p'] * x}, T([1.0, 2.0, 3.0]),
                 {'s': T(0.0), 'p': T(1.0)})
  assert out['s'].numpy().tolist() == [1.0, 3.0, 6.0] and out['p'].numpy().tolist() == [1.0, 2.0, 6.0]
  s = {'b': (1, [2, 3]), 'a': 0}
  assert tft.nest_flatten(s) == [0, 1, 2, 3]                                  # dict keys sorted, as tf.nest
  assert tft.nest_pack(s, [10, 11, 12, 13]) == {'b': (11, [12, 13]), 'a': 10}
  assert tft.nest_map(lambda x, y: x + y, s, s) == {'b': (2, [4, 6]), 'a': 0}
  with pytest.raises(AssertionError):
    tft.nest_assert_same({'a': 1}, {'b': 1})


def test_distributions_against_closed_forms():
  rng = np.random.RandomState(2)
  la, lb = rng.randn(3, 4, 5), rng.randn(3, 4, 5)
  pa = np.exp(la) / np.exp(la).sum(-1, keepdims=True)
  pb = np.exp(lb) / np.exp(lb).sum(-1, keepdims=True)
  A, B = tft.Independent(tft.OneHotCategorical(T(la)), 1), tft.Independent(tft.OneHotCategorical(T(lb)), 1)
  assert np.allclose(tft.kl_divergence(A, B).numpy(), (pa * (np.log(pa) - np.log(pb))).sum((-1, -2)))
  assert np.allclose(A.entropy().numpy(), -(pa * np.log(pa)).sum((-1, -2)))
  assert A.batch_shape == (3,) and A.event_shape == (4, 5)
  mode = A.mode().numpy()
  assert np.array_equal(mode.argmax(-1), la.argmax(-1)) and (mode.sum(-1) == 1).all()
  onehot = np.eye(5)[rng.randint(0, 5, (3, 4))]
  assert np.allclose(A.log_prob(T(onehot)).numpy(), (onehot * np.log(pa)).sum((-1, -2)))
  from_probs = tft.OneHotCategorical(probs=T(pa))
  assert np.allclose(from_probs.logits_parameter().numpy(), np.log(pa))
  mu, sd, x = rng.randn(2, 3), rng.rand(2, 3) + 0.5, rng.randn(2, 3)
  N = tft.Independent(tft.Normal(T(mu), T(sd)), 1)
  assert np.allclose(N.log_prob(T(x)).numpy(),
                     (-0.5 * ((x - mu) / sd) ** 2 - np.log(sd) - 0.5 * math.log(2 * math.pi)).sum(-1))
  assert np.allclose(N.distribution.entropy().numpy(), 0.5 * np.log(2 * math.pi * math.e * sd ** 2))
  assert np.allclose(float(tft.Normal(0.0, 0.1).entropy().numpy()), 0.5 * math.log(2 * math.pi * math.e * 0.01))
  lg, t = rng.randn(4), np.array([1.0, 0.0, 1.0, 0.0])
  Bn = tft.Bernoulli(T(lg))
  sig = 1 / (1 + np.exp(-lg))
  assert np.allclose(Bn.log_prob(T(t)).numpy(), t * np.log(sig) + (1 - t) * np.log(1 - sig))
  assert np.allclose(Bn.mean().numpy(), sig)
  assert np.allclose(tft.Independent(Bn, 0).log_prob(T(t)).numpy(), Bn.log_prob(T(t)).numpy())


def test_injected_randomness():
  tft.FEED.load([('uniform', 'u', np.array([[0.05, 0.95]])), ('normal', 'e', np.array([[1.0, -2.0]]))])
  tft.FEED.draws.clear()
  probs = np.array([[0.1, 0.2, 0.7], [0.1, 0.2, 0.7]])
  idx = tft.random_categorical(T(np.log(probs)), 1).numpy()
  assert idx.tolist() == [[0], [2]]                    # inverse CDF: cdf = .1 .3 1.
  s = tft.Normal(T([[10.0, 20.0]]), T([[2.0, 3.0]])).sample().numpy()
  assert s.tolist() == [[12.0, 14.0]]                  # loc + scale * eps
  assert [n for n, _ in tft.FEED.draws] == ['u', 'e'] and not tft.FEED.items
  with pytest.raises(AssertionError):
    tft.random_categorical(T(np.log(probs)), 1)        # nothing left to draw from
  tft.FEED.load([('normal', 'e', np.zeros(2))])
  with pytest.raises(AssertionError):
    tft.random_categorical(T(np.log(probs)), 1)        # a uniform is asked for, a normal is queued
  tft.FEED.items.clear()


def test_module_scopes_name_variables_uniquely():
  class Base(tft.SntModule):
    def __new__(cls, name, *args):          # as tfutils.Module.__new__ does (tfutils.py:98-107):
      obj = super().__new__(cls)            # the module is named before its __init__ runs
      tft.SntModule.__init__(obj, name=name)
      return obj
  class Leaf(Base):
    def __init__(self, name):
      pass
    def build(self):
      self.k = tft.Variable(np.zeros(2, np.float32), name='kernel')
      return self.k
  class Tree(Base):
    def __init__(self, name):
      self.a, self.b = Leaf('Leaf'), Leaf('Leaf1')
      self.extra = {'z': tft.Variable(1.0, trainable=False, name='count')}
  t = Tree('Tree')
  ka, kb = t.a.build(), t.b.build()
  assert ka.name == 'Tree/Leaf/kernel:0' and kb.name == 'Tree/Leaf1/kernel:0'
  assert [v.name for v in t.variables] == ['Tree/Leaf/kernel:0', 'Tree/Leaf1/kernel:0', 'Tree/count:0']
  assert [v.name for v in t.trainable_variables] == ['Tree/Leaf/kernel:0', 'Tree/Leaf1/kernel:0']
  assert tft.SCOPE == ['']


@pytest.mark.skipif(not REF.exists(), reason='reference checkout not present')
@pytest.mark.parametrize('what', ['options_onehot', 'policy_debug', 'report_onehot'])
def test_committed_fixtures_are_what_the_reference_checkout_produces(what, tmp_path):
  """Provenance: run the committed generator on the reference checkout (own process: it installs
  the stand-in modules) and compare every array with the committed fixture."""
  name = f'reference_{what}.npz'
  code = (
      'import importlib.util, pathlib, sys\n'
      f'spec = importlib.util.spec_from_file_location("mrg", r"{HERE / "golden" / "make_reference_golden.py"}")\n'
      'm = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n'
      f'm.HERE = pathlib.Path(r"{tmp_path}")\n'
      f'what = "{what}"\n'
      'if what.startswith("policy_"): m.generate_policy(what[7:])\n'
      'elif what.startswith("report_"): m.generate_report(what[7:])\n'
      'else: m.generate(what, verbose=False)\n')
  res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stderr[-2000:]
  new, old = np.load(tmp_path / name), np.load(HERE / 'golden' / name)
  assert sorted(new.files) == sorted(old.files)
  for k in old.files:
    assert new[k].shape == old[k].shape, k
    assert np.allclose(new[k], old[k], rtol=1e-12, atol=1e-14, equal_nan=True), k


@pytest.mark.skipif(not REF.exists(), reason='reference checkout not present')
def test_reference_q_critics_do_not_construct_as_shipped():
  """DESIGN.md section 9 / SURVEY 8(f4): QFunction / TwinQFunction are not built because the
  reference cannot construct them from its own config: its `qfunction` block does not even apply
  to `defaults` (it sets `pengs_qlambda`, a key `defaults` does not have -> KeyError), and without
  that key the critics trip their own assertions (agent.py:459-461: backprop for both action types
  and an actor that takes the action as input, which no block sets).  Executed, not read."""
  code = (
      'import importlib.util, sys, traceback\n'
      f'spec = importlib.util.spec_from_file_location("mrg", r"{HERE / "golden" / "make_reference_golden.py"}")\n'
      'm = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n'
      'import numpy as np\n'
      'embodied, ref = m.reference_modules()\n'
      'cfg = embodied.Config(ref.Agent.configs["defaults"])\n'
      'block = dict(ref.Agent.configs["qfunction"])\n'
      'try:\n'
      '  cfg.update(block)\n'
      '  print("block APPLIED")\n'
      'except KeyError as e:\n'
      '  print("block KeyError", e)\n'
      'block.pop("pengs_qlambda")\n'
      'for kind in ("qfunction", "qtwin"):\n'
      '  c = cfg.update(block).update({"critic_type": kind, "tf.platform": "cpu", "tf.jit": False})\n'
      '  obs = {"vector": embodied.Space(np.float32, (4,)), "reward": embodied.Space(np.float32), '
      '"is_first": embodied.Space(bool), "is_last": embodied.Space(bool), "is_terminal": embodied.Space(bool)}\n'
      '  act = {"action": embodied.Space(np.float32, (3,), -1.0, 1.0)}\n'
      '  try:\n'
      '    ref.Agent(obs, act, embodied.Counter(), c)\n'
      '    print(kind, "CONSTRUCTED")\n'
      '  except AssertionError as e:\n'
      '    tb = traceback.extract_tb(e.__traceback__)[-1]\n'
      '    print(kind, "AssertionError", tb.filename.split("/")[-1], tb.line)\n')
  res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stderr[-2000:]
  out = res.stdout.splitlines()
  assert any(l.startswith('block KeyError') and 'pengs_qlambda' in l for l in out), out
  lines = [l for l in out if l.startswith(('qfunction', 'qtwin'))]
  assert len(lines) == 2 and all('AssertionError agent.py assert config.actor_grad_' in l for l in lines), lines
