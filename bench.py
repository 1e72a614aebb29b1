#!/usr/bin/env python
"""Benchmark of the DreamerV2+ learner step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): imagined env-steps/sec (learner) = B*T*H per
`Agent.train` call / steady-state wall time, on the a1 config with the 64x64
camera (`a1_vision`: batch 50, seq 50, horizon 15, 16-dim action), synthetic
data, random-init weights, fp32 (exact-f32 MFMA).  Every rank (one process
per GPU) trains on its own batch-50 shard of a global batch 50*N with the
gradients summed over RCCL: weak scaling.  Inputs are resident in HBM when
the timed region starts; the PCIe-inclusive rate is reported separately.

One JSON line on rank 0, with `roofline` (dominant kernel: the fp32 MFMA
contraction kernel `k_mfma_gemm`, timed live with HIP events on its launch
stream) and `cpu_baseline` (the oracle restatement of the reference graph on
the host cores, bounded sample).
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from daydreamer_amd import agent as agent_mod
from daydreamer_amd import config as config_mod
from daydreamer_amd import synthetic

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def make_config(name):
  cfgs = config_mod.load_configs()
  return config_mod.Config(cfgs['defaults']).update(cfgs[name])


def cpu_baseline(cfg, frac_batch=5, T=50, threads=16):
  """The CPU restatement of the reference graph (oracle/dreamer_ref.py, fp32,
  PyTorch-CPU with all host cores) on a bounded sample: batch `frac_batch` of
  the workload's 50, same seq / horizon / networks.  16 threads: measured on
  the MI355X host (256 cores) 16 threads beat 32 / 64 / 256 at every batch size
  tried (5, 25, 50) because the graph is dominated by small sequential ops."""
  from oracle import dreamer_ref
  from daydreamer_amd import spec as spec_mod
  threads = min(threads or os.cpu_count(), os.cpu_count())
  torch.set_num_threads(threads)
  plain = config_mod.to_plain(cfg)
  obs, act = synthetic.make_spaces(64, 16, 16)
  shapes = {k: v.shape for k, v in obs.items()}
  sp = spec_mod.build_spec(plain, shapes, 16)
  params = spec_mod.init_params(sp, 0)
  data = synthetic.make_batch(obs, act, frac_batch, T, seed=0)
  ag = dreamer_ref.RefAgent(plain, shapes, 16, params, torch.float32)
  H, N, G = plain['imag_horizon'], frac_batch * T, sp.groups
  rng = np.random.default_rng(0)
  noise = dict(u_obs_prior=rng.random((T, frac_batch, G)),
               u_obs_post=rng.random((T, frac_batch, G)),
               u_img=rng.random((H, N, G)),
               eps_act=rng.standard_normal((H + 1, N, 16)))
  _, state, _ = ag.train(data, noise)          # warm-up (allocator, oneDNN)
  t0 = time.perf_counter()
  reps = 2
  for _ in range(reps):
    _, state, _ = ag.train(data, noise, state)
  dt = (time.perf_counter() - t0) / reps
  return dict(
      value=frac_batch * T * H / dt, unit='imagined_env_steps/s', cores=threads,
      kind='port',
      sample=(f'oracle/dreamer_ref.py fp32 on PyTorch-CPU, batch {frac_batch} x '
              f'seq {T} x horizon {H} (1/{50 // frac_batch} of the workload batch), '
              f'{reps} steps after 1 warm-up, {dt:.2f} s/step'))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--config', default='a1_vision')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--pipeline', type=int, default=1,
                  help='hip.pipeline (two-stream pipeline of consecutive steps; single GPU)')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', 1))
  if world == 1 and not os.environ.get('DD_BENCH_CHILD') and os.environ.get('DD_FORCE_DIST') != '1':
    # Single GPU: run the measurement in a child process.  The one-time stream-pair
    # selection of the pipeline re-captures HIP graphs; should the runtime die in it, the
    # measurement is repeated with the default pair instead of losing the bench line.
    import subprocess
    for tune in ('1', '0') if os.environ.get('DD_PIPE_TUNE', '1') == '1' else ('0',):
      env = dict(os.environ, DD_BENCH_CHILD='1', DD_PIPE_TUNE=tune)
      r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                         stdout=subprocess.PIPE, text=True)
      lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
      if r.returncode == 0 and lines:
        print(lines[-1])
        return
      sys.stderr.write(f'bench child (DD_PIPE_TUNE={tune}) failed with code {r.returncode}\n')
    sys.exit(1)
  rank = int(os.environ.get('RANK', 0))
  local = int(os.environ.get('LOCAL_RANK', 0))
  ndev = torch.cuda.device_count()
  local = local % max(ndev, 1)  # (testing aid: several ranks may share one GPU under gloo)
  os.environ['LOCAL_RANK'] = str(local)
  torch.cuda.set_device(local)
  if world > 1 or os.environ.get('DD_FORCE_DIST') == '1':
    import torch.distributed as dist
    backend = os.environ.get('DD_DIST_BACKEND', 'nccl')  # nccl == RCCL on ROCm
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
    else:
      dist.init_process_group(backend)
  assert world == args.gpus, (world, args.gpus)

  cfg = make_config(args.config).update({'hip.pipeline': bool(args.pipeline)})
  plain = config_mod.to_plain(cfg)
  B, T, H = plain['batch_size'], plain['replay_chunk'], plain['imag_horizon']
  obs, act = synthetic.make_spaces(64, 16, 16)
  agent = agent_mod.Agent(obs, act, None, cfg)
  data = synthetic.make_batch(obs, act, B * world, T, seed=0)

  def barrier():
    if world > 1:
      import torch.distributed as dist
      dist.barrier()
    torch.cuda.synchronize()

  # ---- warm-up: first call builds + runs eagerly, second captures the graphs
  state = None
  warm = max(args.warmup, 3)
  for _ in range(warm):
    _, state, mets = agent.train(data, state)
  L = agent.learner
  plan = agent._plan

  pipelined = isinstance(plan, agent_mod.Pipeline)

  def resident_step():
    if pipelined:  # enqueue step k, read the metrics of step k-1
      return plan.step()
    plan.replay()
    return L.read_metrics()

  if pipelined:  # finish the one-time stream-pair selection (real train steps, untimed)
    plan.tune(resident_step)

  # ---- timed region: inputs already resident in HBM
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    mets = resident_step() or mets
  mets = agent.flush() or mets  # (pipeline: the last step's behaviour phase is inside the timed region)
  barrier()
  dt = time.perf_counter() - t0
  if world > 1:
    import torch.distributed as dist
    tmax = torch.tensor([dt], dtype=torch.float64, device=f'cuda:{local}')
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
  ms = 1e3 * dt / args.steps
  value = B * world * T * H / (dt / args.steps)

  # ---- PCIe-inclusive (host numpy batch -> Agent.train -> numpy metrics)
  barrier()
  t0 = time.perf_counter()
  n_incl = max(3, args.steps // 4)
  for _ in range(n_incl):
    _, state, mets = agent.train(data, state)
  mets = agent.flush() or mets
  barrier()
  dt_incl = (time.perf_counter() - t0) / n_incl

  # ---- replay-inclusive: minibatches gathered in HBM from a DeviceReplay
  # (embodied.Replay API) -> Agent.train; no host copy of the batch
  dt_replay = None
  if world == 1:
    from daydreamer_amd import replay as replay_mod
    rep = replay_mod.DeviceReplay(chunk=T, capacity=50_000)
    eps = synthetic.make_batch(obs, act, 16, 4 * T, seed=1)
    for e in range(16):
      rep.add_traj({**{k: v[e] for k, v in eps.items()},
                    'is_last': np.arange(4 * T) == 4 * T - 1})
    ds = agent.dataset(rep.dataset)
    for _ in range(2):
      _, state, mets = agent.train(next(ds), state)
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_incl):
      _, state, mets = agent.train(next(ds), state)
    mets = agent.flush() or mets
    barrier()
    dt_replay = (time.perf_counter() - t0) / n_incl

  # ---- live roofline of the dominant kernel: events around every contraction
  # launch of one eager step on the launch stream.
  roof = None
  step_flops = None
  if rank == 0:
    # every launch context of the step: main, side stream (deferred weight gradients),
    # behaviour phase
    all_ops = [o for o in {id(o): o for o in (L.ops_a, L.ops2, L.ops_b) if o is not None}.values()]
    shared = []
    for o in all_ops:
      o.trace = shared
    L.plan_backup, L.plan = L.plan, __import__('daydreamer_amd.graphs', fromlist=['EagerPlan']).EagerPlan()
    torch.cuda.synchronize()
    L.train_step_device(True)
    torch.cuda.synchronize()
    trace = shared
    for o in all_ops:
      o.trace = None
    L.plan = L.plan_backup
    tot_f = sum(f for _, f, _, _ in trace)
    tot_t = sum(e0.elapsed_time(e1) for _, _, e0, e1 in trace) * 1e-3
    by = {}
    for lab, f, e0, e1 in trace:
      d = by.setdefault(lab.split(' ')[0], [0, 0.0, 0.0])
      d[0] += 1
      d[1] += f
      d[2] += e0.elapsed_time(e1) * 1e-3
    step_flops = tot_f
    # algorithmic bytes (operands read once, result written once), from the labels
    alg_bytes = sum(int(lab.rsplit(' B', 1)[1]) for lab, _, _, _ in trace)
    pmc = None
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_hbm_traffic.json')
    if os.path.exists(pmc_path):
      pmc = json.load(open(pmc_path))
    ach = tot_f / tot_t / 1e12
    roof = dict(
        bound='mfma',
        kernel='k_mfma_gemm_s3<*> (fp32 GEMM + implicit-GEMM conv on the bf16 matrix pipe: exact 3-way bf16 split, 6 products, fp32 accumulate; incl. split-K reduce)',
        achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
        frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
        peak_note='157.3 = dense fp32-input MFMA peak (the dtype of the path); the split-bf16 '
                  'loop is bounded by the bf16 MFMA rate / 6 = 397 TFLOP/s (2382 measured '
                  'peak, MI355X_MICROARCH.md), frac of that = %.4f' % (ach / (2382.0 / 6)),
        traffic=None if pmc is None else pmc['bytes_per_launch'],
        traffic_source=None if pmc is None else pmc['source'],
        algorithmic_bytes_per_launch=round(alg_bytes / len(trace)),
        launches=len(trace), avg_launch_us=round(1e6 * tot_t / len(trace), 2),
        flops_per_launch=tot_f / len(trace),
        kernel_time_ms_per_step=round(1e3 * tot_t, 2),
        by_class={k: dict(launches=v[0], tflop=round(v[1] / 1e12, 4),
                          ms=round(1e3 * v[2], 3),
                          tflops=round(v[1] / max(v[2], 1e-9) / 1e12, 2))
                  for k, v in by.items()})
  # all ranks must keep participating in the collectives of that extra step
  if world > 1 and rank != 0:
    L.plan_backup, L.plan = L.plan, __import__('daydreamer_amd.graphs', fromlist=['EagerPlan']).EagerPlan()
    L.train_step_device(True)
    L.plan = L.plan_backup
  barrier()

  base = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    base = cpu_baseline(cfg)

  if rank == 0:
    out = dict(
        metric='imagined env-steps/sec (learner)', value=round(value, 1),
        unit='imagined_env_steps/s', n_gpus=world, steps=args.steps,
        warmup=warm, ms_per_step=round(ms, 3), higher_is_better=True,
        scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
        config=dict(
            workload=('BASELINE configs[1]: a1 config + 64x64x3 image + 16-dim '
                      'proprio, 16-dim action, batch 50 x seq 50 x horizon 15 '
                      'per GPU, rssm deter 256 / stoch 32x32, one full '
                      'Agent.train step (world model + critic + actor updates)'),
            global_batch=B * world, seq_len=T, horizon=H,
            parallelism=f'dp{world}', hip_graphs=plan.n_graphs,
            pipeline=('two-stream: behaviour phase of step k overlaps world-model phase of step k+1'
                      if pipelined else 'off')),
        pcie_inclusive=dict(value=round(B * world * T * H / dt_incl, 1),
                            ms_per_step=round(1e3 * dt_incl, 3)),
        replay_inclusive=None if dt_replay is None else dict(
            value=round(B * T * H / dt_replay, 1), ms_per_step=round(1e3 * dt_replay, 3),
            note='DeviceReplay.sample_batch (dd_replay_gather from the HBM episode ring) + '
                 'Agent.train per step, numpy metrics out'),
        step_algorithmic_tflop=None if step_flops is None else round(step_flops / 1e12, 4),
        step_mfma_frac=None if step_flops is None else round(
            step_flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        losses=dict(model_loss=float(mets['model_loss']),
                    actor_loss=float(mets['actor_loss']),
                    critic_loss=float(mets['extr_critic_loss'])),
        roofline=roof, cpu_baseline=base)
    print(json.dumps(out))
  if world > 1 or os.environ.get('DD_FORCE_DIST') == '1':
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
