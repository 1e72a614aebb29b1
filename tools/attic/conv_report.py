"""Per-launch time / TFLOP/s of every convolution (and the ten slowest GEMMs) of one eager
BASELINE-config step (HIP events around each C-ABI call, all launch contexts)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic, graphs
cfgs = agent_mod.Agent.configs
cfg = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision']).update({'hip.pipeline': False})
obs, act = synthetic.make_spaces(64, 16, 16)
ag = agent_mod.Agent(obs, act, None, cfg)
data = synthetic.make_batch(obs, act, 50, 50, seed=0)
state = None
for _ in range(2):
  _, state, _ = ag.train(data, state)
L = ag.learner
shared = []
for o in (L.ops_a, L.ops2, L.ops_b):
  o.trace = shared
L.plan = graphs.EagerPlan()
torch.cuda.synchronize()
L.train_step_device(True)
torch.cuda.synchronize()
rows = [(lab, f, e0.elapsed_time(e1)) for lab, f, e0, e1 in shared]
for lab, f, ms in rows:
  if lab.startswith('conv'):
    print(f'{lab:60s} {ms*1e3:8.1f} us {f/ms/1e9:7.1f} TF')
print('--- slowest GEMMs')
for lab, f, ms in sorted((r for r in rows if r[0].startswith('gemm')), key=lambda r: -r[2])[:12]:
  print(f'{lab:60s} {ms*1e3:8.1f} us {f/ms/1e9:7.1f} TF')

print('--- GEMMs by shape (count, total us, avg us, TF)')
import collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for lab, f, ms in rows:
  if lab.startswith('gemm'):
    k = ' '.join(lab.split(' ')[:4])
    agg[k][0] += 1; agg[k][1] += ms; agg[k][2] += f
for k, (n, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
  print(f'{k:40s} x{n:4d} {ms*1e3:9.1f} us  avg {ms*1e3/n:7.1f} us {f/ms/1e9:7.1f} TF')
