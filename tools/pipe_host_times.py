"""Host-side time of the parts of a pipelined Agent.train call (steady state): upload, the
enqueue of the three phase plans, the wait for the metrics of the step before the previous one."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic, learner as LM
cfgs = config_mod.load_configs()
base = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision'])
obs, act = synthetic.config_spaces('a1_vision')
data = synthetic.make_batch(obs, act, base.batch_size, base.replay_chunk, seed=0)
ag = agent_mod.Agent(obs, act, None, base)
state = None
for _ in range(6):
  _, state, m = ag.train(data, state)
P = ag._pipe
acc = {}
def wrap(obj, name, label):
  f = getattr(obj, name)
  def g(*a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    return r
  setattr(obj, name, g)
wrap(ag.learner, 'upload', 'upload')
for pl, nm in ((P.pa1, 'enqueue A1'), (P.pa2, 'enqueue A2'), (P.pb, 'enqueue B')):
  wrap(pl, 'replay_on', nm)
wrap(P, '_read', 'wait for + fetch metrics of step k-2')
wrap(P, '_publish', 'publish')
n = 30
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
  _, state, m = ag.train(data, state)
t1 = time.perf_counter()
ag.flush(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{n} calls: host loop {1e3 * (t1 - t0) / n:.2f} ms per call, incl. drain {1e3 * (t2 - t0) / n:.2f}')
for k, v in acc.items():
  print(f'  {k:40s} {1e3 * v / n:7.3f} ms per call')
