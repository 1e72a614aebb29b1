"""Repro attempt: pipeline stream-pair selection (graph re-capture) fed by DeviceReplay
minibatches vs host minibatches.  usage: tune_repro.py device|host"""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import helpers
from daydreamer_amd import agent as agent_mod, replay as replay_mod, synthetic
mode = sys.argv[1]
cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=6, replay_chunk=8, imag_horizon=4)
obs, act = synthetic.make_spaces(64, 5, 3)
rep = replay_mod.DeviceReplay(chunk=8, capacity=2000)
for e in range(6):
  ep = synthetic.make_batch(obs, act, 1, 40, seed=e, terminals=0.0, smooth_images=True)
  rep.add_traj({**{k: v[0] for k, v in ep.items()}, 'is_last': np.arange(40) == 39})
ag = agent_mod.Agent(obs, act, None, cfg)
ds = ag.dataset(rep.dataset)
def batch():
  b = next(ds)
  return b if mode == 'device' else {k: v.cpu().numpy() for k, v in b.items()}
state = None
for i in range(3):
  _, state, m = ag.train(batch(), state)
print('pipelined, tuning...', flush=True)
box = [state]
def run():
  _, box[0], _ = ag.train(batch(), box[0])
ag._pipe.tune(run)
print(mode, 'tuned ok', ag._pipe.periods and min(ag._pipe.periods.values()), flush=True)
