"""Generates tests/golden/sampler_twin.npz: (statistics, uniforms) -> class indices of the
categorical latent draw, from the numpy restatement oracle/ref_ops.sample_twin_np."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import ref_ops  # noqa: E402

out = {}
for key, (rows, G, C, um) in dict(a=(16, 32, 32, 0.01), b=(6, 64, 64, 0.01), c=(64, 1, 6, 0.1)).items():
  rng = np.random.RandomState(ord(key))
  x = (rng.randn(rows, G * C) * 2).astype(np.float32)
  u = rng.rand(rows, G).astype(np.float32)
  idx, _ = ref_ops.sample_twin_np(x, u, G, C, um)
  out.update({f'{key}_x': x, f'{key}_u': u, f'{key}_idx': idx.astype(np.int32), f'{key}_G': G,
              f'{key}_C': C, f'{key}_um': um})
np.savez_compressed(ROOT / 'tests' / 'golden' / 'sampler_twin.npz', **out)
print('written')
