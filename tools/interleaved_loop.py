"""Train step time when every train call follows a policy call (the in-process loop of the
reference's run/train.py) under hip.pipeline auto / true / false at configs[1]."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic
cfgs = config_mod.load_configs()
base = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision'])
obs, act = synthetic.config_spaces('a1_vision')
data = synthetic.make_batch(obs, act, 50, 50, seed=0)
o = {k: v[:1, 0] for k, v in data.items() if k not in ('action', 'reset')}
for mode in ('auto', True, False):
  ag = agent_mod.Agent(obs, act, None, base if mode == 'auto' else base.update({'hip.pipeline': mode}))
  state, pst = None, None
  def loop(n):
    global state, pst
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      _, pst = ag.policy(o, pst, 'train')
      _, state, m = ag.train(data, state)
    ag.flush()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
  loop(8)
  print(f'hip.pipeline {mode}: policy + train {loop(20):.2f} ms per iteration (plan: {type(ag._plan).__name__})')
  def loop2(n):
    global state
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      _, state, m = ag.train(data, state)
    ag.flush()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
  loop2(60)
  print(f'hip.pipeline {mode}: train only     {loop2(20):.2f} ms per iteration (plan: {type(ag._plan).__name__})')
  del ag
