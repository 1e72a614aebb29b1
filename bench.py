#!/usr/bin/env python
"""Benchmark of the DreamerV2+ learner step on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launches its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): imagined env-steps/sec (learner) = B*T*H per
`Agent.train` call / steady-state wall time of the call - host numpy minibatch in (the PCIe
upload is inside the timed region, as it is inside TFAgent.train), all three optimizers
stepped, numpy metrics out.  Workload: BASELINE configs[1] by default (`a1_vision`: a1 block
with the 64x64 camera, batch 50 x seq 50 x horizon 15, 16-dim action), synthetic data,
random-init weights, fp32 arithmetic.  Other workloads: --config xarm | ur5_multicam |
a1_scaled | a1 (spaces: daydreamer_amd/synthetic.config_spaces).

Scaling over N GPUs (one process per GPU, gradients + controller statistics summed over RCCL):
  --scaling auto    (default) strong where the workload's global batch divides by N - the metric
                    is defined at a fixed global batch: configs[1]'s batch 50 over
                    2 GPUs is 25 + 25, BASELINE configs[2]'s own split - weak where it does not
                    (50 does not divide by 4 or 8: batch 50 per GPU); the JSON line's `scaling`
                    and `metric` say which one ran
  --scaling strong  the global batch is the config's (or --batch) and every rank takes B/N rows,
                    e.g.  --config a1_scaled --scaling strong  (batch 256) or --batch 48
  --scaling weak    every rank trains on its own batch-B shard of a global batch B*N.
On a box with fewer GPUs than ranks the ranks share devices and the collectives fall back to
gloo on device tensors (a plumbing check, not a measurement; the JSON line says so).

`value` is measured on the SHIPPED DEFAULT schedule (`hip.pipeline: auto`): on one GPU the
two-stream pipeline of consecutive steps on its measured stream pair - every call's own metrics,
read from the device when looked at, at the latest inside the next call; the timed region ends
with a drain, so every step's metrics are fetched inside it - under data parallelism the
sequential schedule.  The other schedule is reported next to it (`sequential_default` /
`pipelined`, single GPU); --pipeline 0 | 1 forces one.

One JSON line on rank 0 with `roofline` (dominant kernel family: the MFMA contraction kernels,
timed live with HIP events on their launch streams; `peak` is the roof of the instruction stream
the kernels issue - the dense bf16 MFMA peak / 6 products per fp32 product - and the fp32-MFMA
nominal figure is given beside it) and `cpu_baseline` (the oracle restatement of the reference
graph on the host cores, full workload batch).  `roofline.traffic` is measured in this invocation
(two rocprofv3 --pmc passes over a short child run) when rocprofv3 is on the box (--pmc auto),
else it is the committed record of the same kernel sources and is labelled so.
"""

import argparse
import json
import os
import subprocess
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (measured 2382-2495)
# the contraction kernels compute an fp32 product as six v_mfma_f32_32x32x16_bf16 products of the
# exact 3-way bf16 split, so the roof of the instructions they issue is the bf16 peak / 6
PEAK_SPLIT6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6


def kernel_sources_sha():
  """Hash of the HIP sources: the PMC traffic record (profiles/pmc_hbm_traffic.json, collected by
  a separate rocprofv3 --pmc run, tools/pmc_bench.sh) is only quoted for the kernels it measured."""
  import hashlib
  h = hashlib.sha256()
  src = os.path.join(ROOT, 'daydreamer_amd', 'csrc')
  for name in sorted(os.listdir(src)):
    if name.endswith(('.hip', '.h')):
      h.update(name.encode())
      h.update(open(os.path.join(src, name), 'rb').read())
  return h.hexdigest()[:16]


def make_config(name):
  cfgs = config_mod.load_configs()
  return config_mod.Config(cfgs['defaults']).update(cfgs[name])


def cpu_baseline(cfg, name, batch, T, threads=16, reps=5):
  """The CPU restatement of the reference graph AS WRITTEN (oracle/dreamer_ref.py, fp32,
  PyTorch-CPU) on the workload's full per-GPU batch: 1 warm-up + `reps` timed train steps.
  16 threads: measured on the MI355X host (256 cores) 16 threads beat 32 / 64 / 256 at batch
  5, 25 and 50 because the graph is dominated by small sequential ops.
  Eager only: SURVEY 8(d) also asks for a torch.compile'd variant with the faster one reported, but
  inductor's CPU backend needs 52 s here to compile f(x, w) = elu(layer_norm(x @ w)) alone (one g++
  invocation per fused kernel), and the train step is ~1 500 such operations unrolled over the T = 50
  observe scan and the H = 15 rollout, under autograd, with Python-side controller state
  (AutoAdapt, Normalize, the hand-written Adam): not compilable inside a benchmark that has to
  finish in minutes (docs/LABLOG.md, round 6)."""
  from oracle import dreamer_ref
  from daydreamer_amd import spec as spec_mod
  threads = min(threads or os.cpu_count(), os.cpu_count())
  torch.set_num_threads(threads)
  plain = config_mod.to_plain(cfg)
  obs, act = synthetic.config_spaces(name)
  shapes = {k: v.shape for k, v in obs.items()}
  A = act['action'].shape[0]
  disc = bool(getattr(act['action'], 'discrete', False))
  sp = spec_mod.build_spec(plain, shapes, A, disc)
  params = spec_mod.init_params(sp, 0)
  data = synthetic.make_batch(obs, act, batch, T, seed=0)
  ag = dreamer_ref.RefAgent(plain, shapes, A, params, torch.float32, act_discrete=disc)
  H, N, G = plain['imag_horizon'], batch * T, sp.groups
  rng = np.random.default_rng(0)
  noise = dict(u_obs_prior=rng.random((T, batch, G)), u_obs_post=rng.random((T, batch, G)),
               u_img=rng.random((H, N, G)), eps_act=rng.standard_normal((H + 1, N, A)),
               u_act=rng.random((H + 1, N)))
  _, state, _ = ag.train(data, noise)          # warm-up (allocator, oneDNN)
  t0 = time.perf_counter()
  for _ in range(reps):
    _, state, _ = ag.train(data, noise, state)
  dt = (time.perf_counter() - t0) / reps
  return dict(
      value=batch * T * H / dt, unit='imagined_env_steps/s', cores=threads, threads=threads,
      host_cores=os.cpu_count(), kind='port',
      sample=(f'oracle/dreamer_ref.py (reference graph as written, fp32, PyTorch-CPU, {threads} '
              f'threads) on the full workload batch {batch} x seq {T} x horizon {H}: '
              f'{reps} timed train steps after 1 warm-up, {dt:.2f} s/step'))


def pmc_live(argv_tail, timeout=170):
  """HBM traffic of the contraction kernels, measured now: two rocprofv3 --pmc passes (FETCH_SIZE
  and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots') over a short
  child run of this script.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes,
  same guide, section HBM).  Returns the record dict or None (no rocprofv3, failure, timeout)."""
  import collections
  import csv
  import glob
  import re
  import shutil
  import tempfile
  exe = shutil.which('rocprofv3')
  if exe is None:
    return None
  per = {}
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    out = tempfile.mkdtemp(prefix=f'dd_pmc_{counter}_', dir='/tmp')
    cmd = [exe, '--pmc', counter, '--kernel-trace', '-d', out, '-o', 'b', '--output-format', 'csv', '--',
           sys.executable, os.path.abspath(__file__), '--child', '--steps', '2', '--warmup', '3',
           '--no-cpu-baseline', '--pmc', 'off'] + argv_tail
    env = dict(os.environ, TMPDIR='/tmp', PYTHONPATH=ROOT, DD_PIPE_TUNE='0')   # (the counters only need the kernels: skip the 48 stream-pair trial steps)
    try:
      subprocess.run(cmd, env=env, cwd='/tmp', timeout=timeout, check=True,
                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
      files = glob.glob(os.path.join(out, '**', '*counter_collection*.csv'), recursive=True)
      agg = collections.defaultdict(lambda: [0, 0.0])
      for r in csv.DictReader(open(files[0])):
        if r['Counter_Name'] != counter:
          continue
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        key = 'k_mfma_gemm' if 'k_mfma_gemm' in name else 'other'
        agg[key][0] += 1
        agg[key][1] += float(r['Counter_Value'])
      per[counter] = agg['k_mfma_gemm']
    except Exception as e:  # noqa: BLE001 - any failure falls back to the committed record
      print(f'[bench] live PMC pass {counter} failed: {type(e).__name__}: {e}', file=sys.stderr)
      return None
    finally:
      shutil.rmtree(out, ignore_errors=True)
  (nf, vf), (nw, vw) = per['FETCH_SIZE'], per['WRITE_SIZE']
  if not nf or not nw:
    return None
  return dict(
      source=('measured in this invocation: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) '
              'over a child run of bench.py (--steps 2 --warmup 3, default schedule); FETCH_SIZE doubled '
              '(gfx950 counts 128-B requests at 64 B); L2-miss side, Infinity-Cache hits included'),
      kernel='k_mfma_gemm_s3<*> + k_mfma_gemm_ws<*>', dispatches=nf,
      fetch_kib_per_launch_reported=round(vf / nf, 2), write_kib_per_launch=round(vw / nw, 2),
      bytes_per_launch=int(1024 * (2 * vf / nf + vw / nw)), kernel_sources_sha=kernel_sources_sha())


def self_launch(args):
  """`python bench.py --gpus N` with N > 1 and no launcher environment: re-exec under
  torch.distributed.run, one rank per GPU (the driver's own multi-GPU form is that command)."""
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
         str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
         os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--config', default='a1_vision')
  ap.add_argument('--batch', type=int, default=0, help='override the config batch size')
  ap.add_argument('--length', type=int, default=0, help='override the config sequence length')
  ap.add_argument('--horizon', type=int, default=0, help='override the config imagination horizon')
  ap.add_argument('--scaling', choices=('auto', 'weak', 'strong'), default='auto')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cnn', choices=('simple', 'resnet'), default='simple',
                  help='image encoder / decoder family (reference encoder.cnn / decoder.cnn); the BASELINE '
                       'configs use simple')
  ap.add_argument('--precision', choices=('float32', 'bfloat16'), default='float32',
                  help='hip.precision; bfloat16 is the opt-in reduced-precision mode (not the parity mode)')
  ap.add_argument('--pipeline', type=int, default=-1,
                  help='hip.pipeline for the headline value: -1 (default) = the shipped default `auto` (pipeline on '
                       'one GPU, sequential under data parallelism), 0 / 1 force the sequential / pipelined schedule; '
                       'on one GPU the other schedule is measured too and reported beside it')
  ap.add_argument('--pmc', choices=('auto', 'on', 'off'), default='auto',
                  help='HBM traffic of the contraction kernels: auto/on = two rocprofv3 --pmc passes over a '
                       'short child run of this script (single GPU, rocprofv3 on PATH); off = committed record')
  ap.add_argument('--child', action='store_true', help=argparse.SUPPRESS)  # the --pmc child run
  args = ap.parse_args()

  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    self_launch(args)
  world = int(os.environ.get('WORLD_SIZE', 1))
  rank = int(os.environ.get('RANK', 0))
  local = int(os.environ.get('LOCAL_RANK', 0))
  assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
  ndev = torch.cuda.device_count()
  shared_devices = ndev < world
  local = local % max(ndev, 1)
  os.environ['LOCAL_RANK'] = str(local)
  torch.cuda.set_device(local)
  backend = None
  if world > 1 or os.environ.get('DD_FORCE_DIST') == '1':
    import torch.distributed as dist
    if 'RANK' not in os.environ:   # DD_FORCE_DIST=1 without a launcher: a one-rank group of our own
      import socket
      s_ = socket.socket()
      s_.bind(('127.0.0.1', 0))
      os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(s_.getsockname()[1]))
      s_.close()
    # nccl == RCCL on ROCm; RCCL refuses two ranks on one device -> gloo on device tensors
    backend = os.environ.get('DD_DIST_BACKEND', 'gloo' if shared_devices else 'nccl')
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
    else:
      dist.init_process_group(backend)

  if backend is not None or world > 1:
    print(f'[bench rank {rank}/{world}] device cuda:{local} of {ndev} visible, backend '
          f'{backend}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}',
          file=sys.stderr, flush=True)

  cfg = make_config(args.config).update({'hip.precision': args.precision})
  if args.pipeline >= 0:
    cfg = cfg.update({'hip.pipeline': bool(args.pipeline)})
  if args.cnn != 'simple':
    cfg = cfg.update({'encoder.cnn': args.cnn, 'decoder.cnn': args.cnn})
  if args.batch:
    cfg = cfg.update({'batch_size': args.batch})
  if args.length:
    cfg = cfg.update({'replay_chunk': args.length})
  if args.horizon:
    cfg = cfg.update({'imag_horizon': args.horizon})
  plain = config_mod.to_plain(cfg)
  T, H = plain['replay_chunk'], plain['imag_horizon']
  scaling = args.scaling
  if scaling == 'auto':
    scaling = 'strong' if plain['batch_size'] % world == 0 else 'weak'
  if scaling == 'strong':
    Bg = plain['batch_size']
    assert Bg % world == 0, f'global batch {Bg} does not divide over {world} GPUs (use --batch)'
  else:
    Bg = plain['batch_size'] * world
  B = Bg // world
  obs, act = synthetic.config_spaces(args.config)
  agent = agent_mod.Agent(obs, act, None, cfg)
  # every rank holds its own rows of the global batch (what a rank-sharded Agent.dataset yields)
  mine = synthetic.make_batch(obs, act, B, T, seed=rank)
  data = agent_mod.ShardedBatch(mine) if world > 1 else mine

  def barrier():
    if world > 1:
      import torch.distributed as dist
      dist.barrier()
    torch.cuda.synchronize()

  def timed(fn, n):
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
      fn()
    agent.flush()  # (pipeline: the last step's behaviour phase is inside the timed region)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
      import torch.distributed as dist
      tmax = torch.tensor([dt], dtype=torch.float64, device=f'cuda:{local}')
      dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
      dt = float(tmax[0])
    return dt / n

  # ---- warm-up: first call builds + runs eagerly, second captures the graphs
  box = dict(state=None, mets=None)
  def train_call():
    _, box['state'], box['mets'] = agent.train(data, box['state'])
  for _ in range(max(args.warmup, 3)):
    train_call()
  # a pipelined agent measures its stream pair inside its first 48 pipelined train calls (real
  # train steps): finish that now, untimed - the timed region is the steady state of what ships
  box['state'] = agent.tune_pipeline(data, box['state'])
  L, plan = agent.learner, agent._plan
  pipelined = isinstance(plan, agent_mod.Pipeline)

  # ---- THE metric: Agent.train, host minibatch in, numpy metrics out, exactly K calls
  dt = timed(train_call, args.steps)
  mets = agent.flush() or box['mets']
  ms = 1e3 * dt
  value = Bg * T * H / dt
  if args.child:   # the --pmc child: the counters only need the kernels to have run
    return

  # ---- the same with inputs already resident in HBM (no upload, metrics still read)
  def resident_call():
    if pipelined:
      box['mets'] = plan.step() or box['mets']
    else:
      plan.replay()
      box['mets'] = L.read_metrics()
  n_extra = max(3, args.steps // 2)
  dt_res = timed(resident_call, n_extra)

  # ---- the other schedule next to the headline one (single GPU): with the default headline
  # (hip.pipeline off) the opt-in pipeline, with --pipeline 1 the shipped default
  # (N > 1: the shipped default is the sequential schedule; the pipelined one - hip.pipeline: true,
  # one communicator per phase - is measured beside it so that a scaling curve does not only exist
  # for the slower schedule.  DD_BENCH_DP_PIPELINE=0 skips it.)
  def measure_other():
    other = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': not pipelined}))
    obox = dict(state=None)
    def other_call():
      _, obox["state"], _ = other.train(data, obox["state"])
    for _ in range(3):
      other_call()
    obox["state"] = other.tune_pipeline(data, obox["state"])
    def timed_other(n):
      barrier()
      t0 = time.perf_counter()
      for _ in range(n):
        other_call()
      other.flush()
      barrier()
      return (time.perf_counter() - t0) / n
    d = timed_other(args.steps)
    other.flush()
    del other
    return d
  dt_other, other_hung = None, False
  if world == 1:
    dt_other = measure_other()

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
    def replay_call():
      _, box['state'], box['mets'] = agent.train(next(ds), box['state'])
    for _ in range(2):
      replay_call()
    dt_replay = timed(replay_call, n_extra)

  # ---- live roofline of the dominant kernel family: events around every contraction
  # launch of one sequential eager step, on the launch streams.
  roof = None
  step_flops = None
  if True:  # every rank runs the traced step (it contains the collectives); rank 0 reports
    from daydreamer_amd import graphs
    agent.flush()
    all_ops = list({id(o): o for o in (L.ops_a, L.ops2, L.ops_b) if o is not None}.values())
    trace = []
    for o in all_ops:
      o.trace = trace
    keep, L.plan = L.plan, graphs.EagerPlan()
    # strictly one launch at a time: no side-stream branches while the launches are timed
    side = (L.ops2, L.ops_b2)
    L.ops2 = L.ops_b2 = None
    torch.cuda.synchronize()
    L.upload(agent._shard(data))
    L.train_step_device(True)
    torch.cuda.synchronize()
    for o in all_ops:
      o.trace = None
    L.plan = keep
    L.ops2, L.ops_b2 = side
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
    pmc, pmc_note = None, None
    if rank == 0 and world == 1 and args.pmc != 'off' and args.config == 'a1_vision':
      tail = []
      if args.cnn != 'simple':
        tail += ['--cnn', args.cnn]
      if args.batch:
        tail += ['--batch', str(args.batch)]
      if args.length:
        tail += ['--length', str(args.length)]
      if args.horizon:
        tail += ['--horizon', str(args.horizon)]
      torch.cuda.synchronize()
      pmc = pmc_live(tail)
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_hbm_traffic.json')
    if pmc is None and os.path.exists(pmc_path) and args.config == 'a1_vision':
      pmc = json.load(open(pmc_path))
      if pmc.get('kernel_sources_sha') != kernel_sources_sha():
        pmc_note = ('committed record profiles/pmc_hbm_traffic.json was measured on other kernel sources '
                    f'({pmc.get("kernel_sources_sha")}): not quoted')
        pmc = None
      else:
        pmc['source'] = 'committed record (profiles/pmc_hbm_traffic.json, same kernel sources): ' + pmc.get('source', '')
    ach = tot_f / tot_t / 1e12
    kinds = {k: dict(launches=v[0], tflops=round(v[1] / v[2] / 1e12, 1), ms=round(1e3 * v[2], 3),
                     frac=round(v[1] / v[2] / 1e12 / PEAK_SPLIT6_TFLOPS, 4)) for k, v in by.items()}
    roof = dict(
        bound='mfma',
        kernel=('k_mfma_gemm_s3<*> / k_mfma_gemm_ws<*> + k_imagine_rollout<*> / k_imagine_reverse<*> (fp32 GEMM, '
                'implicit-GEMM conv and the fused imagination rollout on the bf16 matrix pipe: exact 3-way bf16 '
                'split, 6 products of v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate; incl. split-K reduce)'),
        achieved=round(ach, 2), peak=round(PEAK_SPLIT6_TFLOPS, 1), unit='TFLOP/s',
        frac=round(ach / PEAK_SPLIT6_TFLOPS, 4),
        peak_note=('peak = dense bf16 MFMA peak 2500 TFLOP/s (MI355X_MICROARCH.md) / 6 bf16 products per fp32 product: '
                   'the roof of the instructions the kernels issue; achieved counts ALGORITHMIC fp32 flop (2*M*N*K)'),
        fp32_mfma_nominal=dict(peak=PEAK_F32_MFMA_TFLOPS, frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                               note='v_mfma_f32_32x32x2_f32 dense peak: what a native-fp32 MFMA loop is bounded by '
                                    '(kinds above 1.0 of it are possible on the bf16 pipe)'),
        traffic=None if pmc is None else pmc['bytes_per_launch'],
        traffic_source=pmc_note if pmc is None else pmc.get('source'),
        traffic_dispatches=None if pmc is None else pmc.get('dispatches'),
        algorithmic_bytes_per_launch=round(alg_bytes / max(len(trace), 1)),
        launches_per_step=len(trace),
        avg_launch_us=round(1e6 * tot_t / max(len(trace), 1), 2),
        kernel_time_ms=round(1e3 * tot_t, 3),
        by_kind=kinds)

  # ---- N > 1: the pipelined schedule (hip.pipeline: true under data parallelism, one communicator
  # per phase) beside the shipped sequential default - LAST and under a watchdog: its collectives
  # have never run on RCCL with more than one rank, and a hang there must not cost the headline line
  if world > 1 and os.environ.get('DD_BENCH_DP_PIPELINE', '1') == '1':
    import threading
    res = {}
    def run_other():
      try:
        torch.cuda.set_device(local)   # (the current device is per thread: HIP launches of this thread go to this rank's GPU)
        res['dt'] = measure_other()
      except Exception as e:   # noqa: BLE001 - reported, the headline stands
        res['err'] = repr(e)
    th = threading.Thread(target=run_other, daemon=True)
    th.start()
    th.join(float(os.environ.get('DD_BENCH_DP_PIPELINE_TIMEOUT', 240)))
    other_hung = th.is_alive()
    dt_other = res.get('dt')
    other_note = 'timed out (watchdog)' if other_hung else res.get('err')
  else:
    other_note = None
  dt_seq = dt_other if pipelined else None
  dt_pipe = dt_other if not pipelined else None

  if rank == 0:
    base = None
    if world == 1 and not args.no_cpu_baseline:
      base = cpu_baseline(cfg, args.config, B, T)
    def rate(d, b=Bg):
      return None if d is None else dict(value=round(b * T * H / d, 1), ms_per_step=round(1e3 * d, 3))
    out = dict(
        metric='imagined env-steps/sec (learner)' + (
            f' [weak scaling: batch {B} per GPU, global batch {Bg}]' if scaling == 'weak' and world > 1 else ''),
        value=round(value, 1), unit='imagined_env_steps/s', n_gpus=world,
        steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3),
        higher_is_better=True, scaling=scaling, vs_baseline=None,
        dtype='f32' if args.precision == 'float32' else 'bf16 inputs, f32 accumulate (opt-in reduced precision)',
        data='synthetic',
        config=dict(
            workload=(f'{args.config}{" with cnn: " + args.cnn if args.cnn != "simple" else ""} '
                      f'(BASELINE.json configs): batch {Bg} ({B} per GPU) x seq {T} x '
                      f'horizon {H}; Agent.train with host minibatch in (PCIe upload timed), all three '
                      'optimizers stepped, numpy metrics out'),
            global_batch=Bg, per_gpu_batch=B, seq_len=T, horizon=H, parallelism=f'dp{world}',
            scaling_note=('--scaling auto: strong (fixed global batch, the metric\'s definition) where the batch '
                          'divides by the GPU count, weak (batch per GPU fixed) where it does not - configs[1] batch 50 '
                          'does not divide by 4 or 8; --config a1_scaled (batch 256) or --batch 48 divide by 1, 2, 4, 8'),
            collectives=None if backend is None else dict(
                backend='RCCL (nccl)' if backend == 'nccl' else backend, world_size=world,
                ranks_share_devices=shared_devices),
            hip_graphs=plan.n_graphs,
            schedule=(('shipped default (hip.pipeline: auto -> on for one process)' if args.pipeline < 0 else 'hip.pipeline: true')
                      + ': behaviour phase of step k next to the world-model phase of step k+1 on the measured stream pair '
                      + f'{plan.pair}, bit-identical parameters, each call\'s own metrics read when looked at, at the '
                      'latest inside the next call (all inside the timed region: it ends with a drain)'
                      if pipelined else
                      ('shipped default (hip.pipeline: auto -> off under data parallelism)' if args.pipeline < 0 and world > 1
                       else 'hip.pipeline: false') + ': sequential schedule, train() returns this call\'s metrics as host values')),
        resident=rate(dt_res),
        pipelined=(None if other_note is None else dict(value=None, note=f'hip.pipeline: true not measured: {other_note}')) if dt_pipe is None else dict(
            **rate(dt_pipe), note='hip.pipeline: true - bit-identical parameters, each call\'s own metrics returned lazily (LazyMetrics)'),
        sequential_default=None if dt_seq is None else dict(
            **rate(dt_seq), note='hip.pipeline: false - the sequential schedule (the shipped default until round 4): train() returns host values'),
        replay_inclusive=None if dt_replay is None else dict(
            **rate(dt_replay), note='DeviceReplay.sample_batch (dd_replay_gather from the HBM '
            'episode ring) + Agent.train per step, numpy metrics out'),
        step_algorithmic_tflop=None if step_flops is None else round(step_flops / 1e12, 4),
        step_mfma_frac=None if step_flops is None else round(
            step_flops / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        losses=dict(model_loss=float(mets['model_loss']), actor_loss=float(mets['actor_loss']),
                    critic_loss=float(mets['extr_critic_loss'])),
        roofline=roof, cpu_baseline=base)
    print(json.dumps(out), flush=True)
  if other_hung:       # (a collective of the watchdogged measurement never returned: no orderly teardown)
    sys.stdout.flush()
    os._exit(0)
  if backend is not None:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
