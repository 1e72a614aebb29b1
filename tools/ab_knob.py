"""A/B of one `hip.<knob>` on the same box, alternating runs: step time of configs[1] under the
sequential and the shipped default schedule with the knob on / off.
  python tools/ab_knob.py fold_heads [config]
  python tools/ab_knob.py imag_rows=32,16 [config]      (two values instead of on / off)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic
knob = sys.argv[1]
vals = (True, False)
if '=' in knob:
  knob, vs = knob.split('=')
  vals = tuple(int(v) if v.lstrip('-').isdigit() else v for v in vs.split(','))
name = sys.argv[2] if len(sys.argv) > 2 else 'a1_vision'
cfgs = config_mod.load_configs()
base = config_mod.Config(cfgs['defaults']).update(cfgs[name])
obs, act = synthetic.config_spaces(name)
B, T = base.batch_size, base.replay_chunk
data = synthetic.make_batch(obs, act, B, T, seed=0)
def run(cfg, n=30):
  ag = agent_mod.Agent(obs, act, None, cfg)
  state = None
  for _ in range(4):
    _, state, m = ag.train(data, state)
  state = ag.tune_pipeline(data, state)
  ag.flush(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    _, state, m = ag.train(data, state)
  ag.flush(); torch.cuda.synchronize()
  dt = 1e3 * (time.perf_counter() - t0) / n
  del ag
  return dt
for rep in range(3):
  for pipe in (False, 'auto'):
    res = []
    for on in vals:
      cfg = base.update({f'hip.{knob}': on})
      if pipe is False:
        cfg = cfg.update({'hip.pipeline': False})
      res.append(run(cfg))
    print(f'rep {rep} schedule {"sequential" if pipe is False else "default"}: {knob} {vals[0]} {res[0]:.2f} ms  {vals[1]} {res[1]:.2f} ms', flush=True)
