"""Worker of tests/test_dp_gpu.py: one data-parallel rank on the HIP kernels.  With
DD_DP_DISTINCT=1 (set by the test when the box has at least as many GPUs as ranks) rank r runs on
GPU r, as the driver launches bench.py; otherwise all ranks share the one visible GPU (LOCAL_RANK
forced to 0).  Collectives run on device tensors through the backend in DD_DIST_BACKEND (gloo by
default; nccl = RCCL, which needs one GPU per rank).  DD_DP_BATCH: the global batch (default 6).

  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \\
      tests/dp_gpu_worker.py OUTDIR
"""
import os
import sys
import pathlib

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
DISTINCT = os.environ.get('DD_DP_DISTINCT') == '1'
if not DISTINCT:
  os.environ['LOCAL_RANK'] = '0'

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from daydreamer_amd import agent as agent_mod, synthetic  # noqa: E402
import helpers  # noqa: E402


def main():
  outdir = sys.argv[1]
  local = int(os.environ['LOCAL_RANK'])
  torch.cuda.set_device(local)
  backend = os.environ.get('DD_DIST_BACKEND', 'gloo')
  if backend == 'nccl':
    dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
  else:
    dist.init_process_group(backend)
  rank, world = dist.get_rank(), dist.get_world_size()
  BG = int(os.environ.get('DD_DP_BATCH', 6))
  dev = f'cuda:{local}'
  print(f'rank {rank}/{world}: device {dev} of {torch.cuda.device_count()}, backend {backend}', flush=True)
  # DD_DP_CONFIG=xarm: the configs[2] family (image + depth + five proprio keys, one-hot 6-way action,
  # REINFORCE, deter = units = 512 at full width) instead of the a1_vision debug block
  if os.environ.get('DD_DP_CONFIG') == 'xarm':
    cfg = helpers.make_config(('xarm',), batch_size=BG, replay_chunk=8, imag_horizon=4)
    obs, act = synthetic.config_spaces('xarm')
  else:
    cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=BG, replay_chunk=8, imag_horizon=4)
    obs, act = synthetic.make_spaces(64, 5, 3)
  batches = [synthetic.make_batch(obs, act, BG, 8, seed=s, smooth_images=True, terminals=0.1)
             for s in range(3)]
  res = {}
  # DD_DP_TUNE=1: the pipelined agent re-measures its stream pair (Pipeline.tune: every ordered
  # pair of the pool, real train steps of the same sequence the sequential agent runs); every
  # rank must then run the pair rank 0 chose
  tune = os.environ.get('DD_DP_TUNE') == '1'
  steps = 6 + (12 * agent_mod.Pipeline.TRIAL + 1 if tune else 0)
  for mode in (False, True):
    ag = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': mode}))
    assert ag.world == world and ag.rank == rank and ag.ops.name == 'hip'
    box = dict(state=None, i=0, m=None)
    def step():
      i = box['i']
      batch = batches[i % 3]
      if i >= 3:  # rank-sharded minibatches, as a sharded Agent.dataset yields them
        per = BG // world
        batch = agent_mod.ShardedBatch({k: v[rank * per:(rank + 1) * per] for k, v in batch.items()})
      _, box['state'], box['m'] = ag.train(batch, box['state'])
      box['i'] = i + 1
    while box['i'] < steps:
      if tune and mode and box['i'] == 3:
        ag._pipe.tune(step, force=True)
      else:
        step()
    assert box['i'] == steps, box['i']
    m = box['m']
    last = ag.flush()
    res[mode] = (ag.save(), last if mode else m)
  if os.environ.get('DD_DP_TUNE') == '1':
    best = agent_mod.Pipeline.BEST
    assert len(best) == 1, best
    pick = torch.tensor(list(best.values())[0], dtype=torch.int64, device=dev)
    picks = [torch.zeros_like(pick) for _ in range(world)]
    dist.all_gather(picks, pick)
    assert all(torch.equal(p, picks[0]) for p in picks), picks
    print(f'rank {rank}: stream pair {tuple(int(x) for x in pick)} on every rank', flush=True)
  a, b = res[False][0], res[True][0]
  bad = [k for k in a if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True)]
  ma, mb = res[False][1], res[True][1]
  badm = [k for k in ma if not np.array_equal(ma[k], mb[k], equal_nan=True)]
  print(f'rank {rank}/{world}: pipelined vs sequential: {len(bad)} arrays / {len(badm)} metrics '
        f'differ {bad[:3]} {badm[:3]}', flush=True)
  if rank == 0:
    np.savez(os.path.join(outdir, 'dp_gpu.npz'), **{f'p/{k}': np.asarray(v) for k, v in a.items()},
             **{f'm/{k}': v for k, v in ma.items()})
  dist.barrier()
  dist.destroy_process_group()
  sys.exit(1 if bad or badm else 0)


if __name__ == '__main__':
  main()
