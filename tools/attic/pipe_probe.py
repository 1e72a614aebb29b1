"""Durations of the pipelined phases while they overlap: A1 (world-model forward + backward on
stream 1) and B (behaviour on stream 2) of the steady state, against the step period and the
phases' stand-alone times.  python tools/pipe_probe.py"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from bench import make_config
from daydreamer_amd import agent as agent_mod, synthetic, config as config_mod

cfg = make_config('a1_vision').update({'hip.pipeline': True})
plain = config_mod.to_plain(cfg)
obs, act = synthetic.config_spaces('a1_vision')
agent = agent_mod.Agent(obs, act, None, cfg)
data = synthetic.make_batch(obs, act, plain['batch_size'], plain['replay_chunk'], seed=0)
state = None
for _ in range(45):   # includes the stream-pair selection
  _, state, _ = agent.train(data, state)
P = agent._pipe if hasattr(agent, '_pipe') else agent.pipeline
ev = []
def wrap(plan, tag):
  orig = plan.replay_on
  def f(stream, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    r = orig(stream, *a, **k)
    e1.record(stream)
    ev.append((tag, e0, e1))
    return r
  plan.replay_on = f
wrap(P.pa1, 'A1'); wrap(P.pa2, 'A2'); wrap(P.pb, 'B')
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
t0.record()
N = 20
for _ in range(N):
  _, state, _ = agent.train(data, state)
agent.flush()
t1.record()
torch.cuda.synchronize()
print(f'period {t0.elapsed_time(t1) / N:.2f} ms')
base = ev[0][1]
for tag in ('A1', 'A2', 'B'):
  d = [e0.elapsed_time(e1) for t, e0, e1 in ev if t == tag]
  print(f'{tag}: mean {np.mean(d[4:]):.2f} ms over {len(d)} replays')
print('timeline of steps 10..12 (ms since first event): tag start end')
for t, e0, e1 in ev:
  s = base.elapsed_time(e0)
  if 10 * 33 < s < 13.2 * 33:
    print(f'  {t:2s} {s:8.2f} {base.elapsed_time(e1):8.2f}')
