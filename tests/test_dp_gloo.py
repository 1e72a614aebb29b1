"""Data parallelism: 2 ranks (one process each, gloo on CPU), each on its shard
of the global batch with summed gradients / batch statistics, must reproduce the
single-process run on the whole batch (the learner's noise is keyed by global
row, so even the sampled latents are identical)."""

import socket
import tempfile

import numpy as np
import torch
import torch.multiprocessing as mp

import helpers
import dp_worker
from daydreamer_amd import learner as LM


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


import pytest


@pytest.mark.parametrize('overlap', [False, True])
def test_two_ranks_equal_one(overlap):
  """overlap: the all-reduce of the decoder / head gradient range issued before the encoder's
  backward pass, the rest in front of the optimizer (Learner.allreduce_early) - same sums."""
  from oracle import ref_ops
  steps = 2
  with tempfile.TemporaryDirectory() as d:
    mp.spawn(dp_worker.run, args=(2, free_port(), d, steps, overlap), nprocs=2, join=True)
    got = dict(np.load(f'{d}/dp.npz'))
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=4, imag_horizon=3)
  plain, sp, shapes, params, data, B, T = helpers.make_problem(
      cfg, image=64, vector=5, action=3, terminals=0.15)
  L = LM.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params, noise_seed=5,
                 dtype=torch.float64)
  for i in range(steps):
    L.upload(data)
    L.train_step_device(use_carry=(i > 0))
    mets = L.read_metrics()
  for name, v in L.export_params().items():
    assert helpers.rel_err(got[name], v) < 1e-9, name
  for k in ('model_loss', 'actor_loss', 'extr_critic_loss', 'model_grad_norm',
            'actent_scale_mean', 'wmkl_scale_mean', 'extr_score_std'):
    assert abs(float(got[f'metric/{k}']) - float(mets[k])) <= 1e-6 * max(1, abs(float(mets[k]))), k


@pytest.mark.parametrize('case', ['debug', 'xarm'])
def test_two_ranks_reproduce_the_reference_run(case):
  """North star, data-parallel row: N ranks on their rows of the global batch == the reference on
  the whole batch.  Directly against a run of the reference's own sources
  (tests/golden/reference_*.npz, tests/test_reference_golden.py): 2 ranks x 1 row, gloo, float64,
  early all-reduce on - every metric both sides hold and every parameter after each of two steps."""
  import pathlib
  gold = np.load(pathlib.Path(__file__).parent / 'golden' / f'reference_{case}.npz')
  with tempfile.TemporaryDirectory() as d:
    mp.spawn(dp_worker.run_reference_case, args=(2, free_port(), d, case), nprocs=2, join=True)
    got = dict(np.load(f'{d}/dp_ref.npz'))
  n = 0
  for k, v in got.items():
    if k not in gold.files:
      continue
    ref = gold[k]
    if '/metric/' in k:
      if np.isnan(ref):
        assert np.isnan(v), k
      else:
        assert abs(float(v) - float(ref)) <= 1e-6 * max(abs(float(ref)), 1e-2), (k, float(v), float(ref))
    else:
      assert np.abs(v - ref).max() <= 1e-9 * max(ref[1], 1e-30), k
    n += 1
  assert n >= 2 * (50 + 100), n
