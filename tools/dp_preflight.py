#!/usr/bin/env python
"""Pre-flight of the data-parallel plumbing on a multi-GPU node, BEFORE a real run:

  python tools/dp_preflight.py --gpus 8        (launches its own ranks, one per GPU)

Every collective pattern the learner uses is exercised on its own, in the order a first train
step would reach it, each under a watchdog so that a hang names the step it hung in instead of
stalling a benchmark (the first multi-rank RCCL run of this code base is also the first run of
these patterns - VERDICT r3, weak #8):

  1. process group: backend nccl (= RCCL) with device_id, HSA_ENABLE_IPC_MODE_LEGACY as exported,
     per-rank device ids printed;
  2. all-reduce (sum) of the three gradient arenas at their configs[1] sizes (19.3 M / 1.46 M /
     1.45 M floats) issued on the library-owned `plan` stream (torch ExternalStream over a
     dd_stream_create handle), and of the decoder / head range on the `comm` stream while the `plan`
     stream computes (Learner.allreduce_early), results checked;
  3. the three communicators of the two-stream pipeline (default group + 2 x dist.new_group) with
     collectives in flight concurrently on `pipe0` / `pipe1` / `read`;
  4. the stream-pair broadcast of the pipeline tuning (int64 pair from rank 0);
  5. graph segment -> host-issued collective -> graph segment on one stream (GraphPlan.cut), replayed;
  6. one real Agent.train step of a tiny configuration on every rank (default schedule, then
     hip.pipeline), parameters equal across ranks afterwards.

Exit code 0 and a PASS line per step on rank 0, or the failing step's name and the exception.
"""
import argparse
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class StepTimeout(Exception):
  pass


def guarded(name, seconds, fn, rank):
  def on_alarm(signum, frame):
    raise StepTimeout(f'step "{name}" did not finish within {seconds} s on rank {rank}')
  signal.signal(signal.SIGALRM, on_alarm)
  signal.alarm(seconds)
  t0 = time.time()
  try:
    fn()
  except Exception as e:  # noqa: BLE001
    print(f'[preflight rank {rank}] FAIL {name}: {type(e).__name__}: {e}', flush=True)
    os._exit(3)   # (a hung collective cannot be unwound: leave at once, the launcher reaps the rest)
  finally:
    signal.alarm(0)
  if rank == 0:
    print(f'[preflight] PASS {name} ({time.time() - t0:.1f} s)', flush=True)


def worker(args):
  import numpy as np
  import torch
  import torch.distributed as dist
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  local = int(os.environ.get('LOCAL_RANK', 0))
  ndev = torch.cuda.device_count()
  shared = ndev < world
  backend = args.backend or ('gloo' if shared else 'nccl')
  local = local % max(ndev, 1)
  os.environ['LOCAL_RANK'] = str(local)
  dev = torch.device(f'cuda:{local}')
  torch.cuda.set_device(dev)
  print(f'[preflight rank {rank}/{world}] device {dev} of {ndev} visible, backend {backend}, '
        f'HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}'
        f'{" (ranks SHARE devices: gloo plumbing check only)" if shared else ""}', flush=True)
  T = args.timeout

  def init():
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev)
    else:
      dist.init_process_group(backend)
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    assert float(t) == world, float(t)
  guarded('1 process group + first all-reduce', T, init, rank)

  from daydreamer_amd import graphs
  plan_s, comm_s = graphs.stream(dev, 'plan'), graphs.stream(dev, 'comm')

  def arenas():
    for n in (19_326_108, 1_461_000, 1_450_000):
      g = torch.full((n,), float(rank + 1), device=dev)
      torch.cuda.synchronize()
      with torch.cuda.stream(plan_s):
        g.mul_(2.0)
        dist.all_reduce(g)
      torch.cuda.current_stream().wait_stream(plan_s)
      want = 2.0 * world * (world + 1) / 2
      assert float(g[0]) == want and float(g[-1]) == want, (n, float(g[0]), want)
    # early range on the comm stream next to compute on the plan stream
    g = torch.full((19_326_108,), float(rank + 1), device=dev)
    a = torch.randn(2048, 2048, device=dev)
    torch.cuda.synchronize()
    comm_s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm_s):
      dist.all_reduce(g[5_651_456:19_299_072])
    with torch.cuda.stream(plan_s):
      for _ in range(10):
        a = a @ a * 1e-3
      dist.all_reduce(g[:5_651_456])
      dist.all_reduce(g[19_299_072:])
      plan_s.wait_stream(comm_s)
    torch.cuda.synchronize()
    want = world * (world + 1) / 2
    assert float(g.min()) == want == float(g.max()), (float(g.min()), float(g.max()), want)
  guarded('2 gradient-arena all-reduces on the library-owned plan / comm streams', T, arenas, rank)

  groups = {}

  def comms():
    groups['b'] = dist.new_group()
    groups['m'] = dist.new_group()
    streams = [graphs.stream(dev, r) for r in ('pipe0', 'pipe1', 'read')]
    ts = [torch.full((1 << 20,), float(rank + 1 + i), device=dev) for i in range(3)]
    torch.cuda.synchronize()
    for _ in range(3):
      for s, t, grp in zip(streams, ts, (None, groups['b'], groups['m'])):
        with torch.cuda.stream(s):
          dist.all_reduce(t, group=grp)
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in ts)
  guarded('3 three communicators, collectives in flight on pipe0 / pipe1 / read', T, comms, rank)

  def bcast():
    pick = torch.tensor([2, 1] if rank == 0 else [0, 0], dtype=torch.int64, device=dev)
    dist.broadcast(pick, src=0)
    assert pick.tolist() == [2, 1], pick.tolist()
  guarded('4 stream-pair broadcast', T, bcast, rank)

  def graph_cut():
    x = torch.zeros(1 << 16, device=dev)
    plan = graphs.GraphPlan(dev)
    def body():
      x.add_(1.0)
      plan.cut(lambda: dist.all_reduce(x))
      x.mul_(0.5)
    plan.capture(body)
    for _ in range(3):
      plan.replay()
    torch.cuda.synchronize()
    v = 0.0
    for _ in range(3):
      v = (v + 1.0) * world * 0.5
    assert abs(float(x[0]) - v) < 1e-4 * max(1.0, v), (float(x[0]), v)
  guarded('5 graph segment -> collective -> graph segment, replayed', T, graph_cut, rank)

  def train():
    import helpers
    from daydreamer_amd import agent as agent_mod, synthetic
    B = 2 * world
    cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=B, replay_chunk=6, imag_horizon=3)
    obs, act = synthetic.make_spaces(64, 5, 3)
    batch = synthetic.make_batch(obs, act, B, 6, seed=0, smooth_images=True)
    os.environ['DD_PIPE_TUNE'] = '0'
    for pipe in (False, True):
      ag = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': pipe}))
      state = None
      for _ in range(4):
        _, state, mets = ag.train(batch, state)
      ag.flush()
      flat = ag.groups['model'].flat
      lo, hi = flat.clone(), flat.clone()
      dist.all_reduce(lo, op=dist.ReduceOp.MIN)
      dist.all_reduce(hi, op=dist.ReduceOp.MAX)
      assert torch.equal(lo, hi), f'parameters differ across ranks (pipeline={pipe})'
      assert np.isfinite(float(mets['model_loss']))
      del ag
  guarded('6 Agent.train on every rank (default schedule, then hip.pipeline)', 4 * T, train, rank)

  dist.barrier()
  dist.destroy_process_group()
  if rank == 0:
    print('[preflight] all steps passed', flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=2)
  ap.add_argument('--backend', default='', help='nccl (default when every rank has its own GPU) | gloo')
  ap.add_argument('--timeout', type=int, default=120, help='seconds per step')
  args = ap.parse_args()
  if 'WORLD_SIZE' in os.environ:
    worker(args)
    return
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == '__main__':
  main()
