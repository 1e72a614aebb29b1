#!/usr/bin/env python
"""Where Agent.report's time goes at the reference's TEST_CONFIG: wall time per call against the
sum of kernel time and the number of launches (torch profiler, device activity)."""
import sys, pathlib, time
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools'))
import bench_test_config as btc
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic
cfgs = config_mod.load_configs()
cfg = config_mod.Config(cfgs['defaults']).update(btc.TEST_CONFIG)
obs, act = btc.spaces()
ag = agent_mod.Agent(obs, act, None, cfg)
batch = synthetic.make_batch(obs, act, 8, 8, seed=0, terminals=0.05, smooth_images=True)
batch['step'] = np.tile(np.arange(8, dtype=np.int32), (8, 1))
st = None
for _ in range(3):
  _, st, _ = ag.train(batch, st)
for _ in range(2):
  ag.report(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
n = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
  t0 = time.perf_counter()
  for _ in range(n):
    rep = ag.report(batch)
  torch.cuda.synchronize()
  wall = (time.perf_counter() - t0) / n * 1e3
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
kern = sum(e.device_time for e in ev) / n / 1e3
print(f'report: wall {wall:.2f} ms per call; device activity {kern:.2f} ms in {len(ev) / n:.0f} kernels / copies per call')
tab = {}
for e in ev:
  t = tab.setdefault(e.name[:60], [0, 0.0]); t[0] += 1; t[1] += e.device_time
for name, (c, t) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:12]:
  print(f'  {t / n / 1e3:7.3f} ms  n={c / n:6.1f}  {name}')
