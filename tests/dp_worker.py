"""Worker for the 2-process data-parallel test (gloo on CPU)."""

import os
import sys
import pathlib

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))


def run(rank, world, port, outdir, steps, overlap=False):
  import numpy as np
  import torch
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(2)
  import helpers
  from daydreamer_amd import agent as agent_mod, learner as LM
  from oracle import ref_ops
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=4, imag_horizon=3)
  plain, sp, shapes, params, data, B, T = helpers.make_problem(
      cfg, image=64, vector=5, action=3, terminals=0.15)
  per = B // world
  shard = {k: v[rank * per:(rank + 1) * per] for k, v in data.items()}
  L = LM.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', per, T, params=params, rank=rank,
                 world=world, comm=agent_mod.DistComm(), noise_seed=5, dtype=torch.float64,
                 dp_overlap=overlap)
  assert L.dp_overlap == bool(overlap) and (not overlap or L.early_range() is not None)
  for i in range(steps):
    L.upload(shard)
    L.train_step_device(use_carry=(i > 0))
    mets = L.read_metrics()
  if rank == 0:
    np.savez(os.path.join(outdir, 'dp.npz'), **L.export_params(),
             **{f'metric/{k}': v for k, v in mets.items()})
  dist.barrier()
  dist.destroy_process_group()


def run_reference_case(rank, world, port, outdir, case):
  """Two learner steps on this rank's rows of a reference-run problem
  (tests/golden/make_reference_golden.py); rank 0 saves metrics and parameter digests."""
  import importlib.util
  import numpy as np
  import torch
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(2)
  from daydreamer_amd import agent as agent_mod, learner as LM
  from oracle import ref_ops
  spec = importlib.util.spec_from_file_location(
      'make_reference_golden', ROOT / 'tests' / 'golden' / 'make_reference_golden.py')
  mrg = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mrg)
  base, (plain, sp, shapes, params, data, B, T) = mrg.build(case)
  per = B // world
  shard = {k: v[rank * per:(rank + 1) * per] for k, v in data.items()}
  L = LM.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', per, T, params=params, rank=rank, world=world,
                 comm=agent_mod.DistComm(), noise_seed=mrg.mg.NOISE_SEED, dtype=torch.float64,
                 dp_overlap=True)
  out = {}
  for step in (1, 2):
    L.upload(shard)
    L.train_step_device(use_carry=(step > 1))
    for k, v in L.read_metrics().items():
      out[f's{step}/metric/{k}'] = np.float64(v)
    for k, v in L.export_params().items():
      v = np.asarray(v, np.float64)
      out[f's{step}/paramsum/{k}'] = np.array([v.sum(), np.abs(v).sum()])
  if rank == 0:
    np.savez(os.path.join(outdir, 'dp_ref.npz'), **out)
  dist.barrier()
  dist.destroy_process_group()
