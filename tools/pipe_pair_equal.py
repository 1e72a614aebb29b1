"""Pipelined == sequential, bit for bit, on a given stream pair (DD_PIPE_PAIR=a,b) and config:
  python tools/pipe_pair_equal.py xarm 6"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, torch
import helpers
from daydreamer_amd import agent as agent_mod, synthetic
name, BG = sys.argv[1], int(sys.argv[2])
if name == 'xarm':
  cfg = helpers.make_config(('xarm',), batch_size=BG, replay_chunk=8, imag_horizon=4)
  obs, act = synthetic.config_spaces('xarm')
else:
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=BG, replay_chunk=8, imag_horizon=4)
  obs, act = synthetic.make_spaces(64, 5, 3)
batches = [synthetic.make_batch(obs, act, BG, 8, seed=s, smooth_images=True, terminals=0.1) for s in range(3)]
res = {}
for mode in (False, True):
  ag = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': mode}))
  state = None
  for i in range(int(os.environ.get('STEPS', 8))):
    _, state, m = ag.train(batches[i % 3], state)
  ag.flush()
  res[mode] = ag.save()
  if mode: print('pair', ag._pipe.pair)
a, b = res[False], res[True]
bad = [k for k in a if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True)]
print(f'{name} pair {os.environ.get("DD_PIPE_PAIR")} tune {os.environ.get("DD_PIPE_TUNE")}: {len(bad)} of {len(a)} arrays differ {bad[:4]}')
