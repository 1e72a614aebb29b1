"""Host twin of the categorical latent draw (dd_onehot_sample_host, compiled from the same
source as the gfx950 kernel: csrc/sampler_core.h) against its numpy restatement
(oracle/ref_ops.sample_twin_np) and committed golden vectors: bit-exact indices, including
uniforms sitting exactly on CDF edges.  No GPU needed (the twin is host code in the C-ABI
library); the device-vs-twin equality is tests/test_hip_ops.py::test_stats_sample."""

import pathlib

import numpy as np
import pytest
import torch

from daydreamer_amd import hipops
from oracle import ref_ops

GOLD = pathlib.Path(__file__).parent / 'golden' / 'sampler_twin.npz'


def problem(rows, G, C, seed):
  g = torch.Generator().manual_seed(seed)
  x = torch.randn(rows, G * C, generator=g) * 2
  u = torch.rand(rows, G, generator=torch.Generator().manual_seed(seed + 1))
  return x, u


@pytest.mark.parametrize('rows,G,C,um', [
    (50, 32, 32, 0.01), (300, 32, 32, 0.0), (96, 8, 8, 0.01), (10, 64, 64, 0.01),
    (500, 1, 6, 0.1), (40, 4, 20, 0.01), (1, 1, 2, 0.0)])
def test_twin_equals_restatement(rows, G, C, um):
  x, u = problem(rows, G, C, 3)
  idx, stoch, logit = hipops.onehot_sample_host(x, u, G, C, um, 0)
  idx_n, pm = ref_ops.sample_twin_np(x.numpy(), u.numpy(), G, C, um)
  assert np.array_equal(idx.numpy(), idx_n)
  assert np.array_equal(stoch.reshape(rows, G, C).argmax(-1).numpy(), idx_n)
  assert float(stoch.sum()) == rows * G
  if um > 0:
    assert np.allclose(logit.reshape(rows, G, C).numpy(), np.log(pm), rtol=0, atol=2e-6)
  a, _, _ = hipops.onehot_sample_host(x, None, G, C, um, 1)
  assert np.array_equal(a.numpy(), ref_ops.sample_twin_np(x.numpy(), None, G, C, um, 1)[0])


def test_twin_on_cdf_edges():
  rows, G, C, um = 32, 32, 32, 0.01
  x, _ = problem(rows, G, C, 7)
  _, pm = ref_ops.sample_twin_np(x.numpy(), np.zeros((rows, G), np.float32), G, C, um)
  cdf = np.cumsum(pm.astype(np.float64), -1)
  for k in range(3):
    e = np.take_along_axis(cdf, np.random.RandomState(k).randint(0, C - 1, (rows, G, 1)), -1)[..., 0]
    u = (e / cdf[..., -1]).astype(np.float32)
    if k != 1:
      u = np.nextafter(u, np.float32(k - 1), dtype=np.float32)
    u = torch.from_numpy(np.clip(u, 0, np.float32(1) - np.float32(2 ** -24)))
    idx, _, _ = hipops.onehot_sample_host(x, u, G, C, um, 0)
    assert np.array_equal(idx.numpy(), ref_ops.sample_twin_np(x.numpy(), u.numpy(), G, C, um)[0])


def test_exp_det_properties():
  x = np.linspace(-86, 0, 400001).astype(np.float32)
  e = ref_ops.exp_det(x)
  assert ref_ops.exp_det(np.float32([0.0]))[0] == 1.0
  assert np.abs(e / np.exp(x.astype(np.float64)) - 1).max() < 2.0 ** -22
  assert (ref_ops.exp_det(np.float32([-86.5, -1e30, -np.inf])) == 0).all()
  assert (np.diff(e) >= 0).all()   # monotone on the grid


def test_golden_vectors():
  """Committed indices (tests/golden/make_sampler_golden.py): guards both the twin and the
  restatement against a silent change of the shared arithmetic."""
  gold = np.load(GOLD)
  for key in ('a', 'b', 'c'):
    G, C, um = (int(gold[f'{key}_G']), int(gold[f'{key}_C']), float(gold[f'{key}_um']))
    x, u = torch.from_numpy(gold[f'{key}_x']), torch.from_numpy(gold[f'{key}_u'])
    idx, _, _ = hipops.onehot_sample_host(x, u, G, C, um, 0)
    assert np.array_equal(idx.numpy(), gold[f'{key}_idx'])
    assert np.array_equal(ref_ops.sample_twin_np(x.numpy(), u.numpy(), G, C, um)[0], gold[f'{key}_idx'])
