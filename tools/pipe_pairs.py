"""Stream-pair measurement of the two-stream pipeline (agent.Pipeline): period of every ordered pair
of the pool at the bench workload, and the wall time of the measurement itself.
  python tools/pipe_pairs.py [config] [batch] [length]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['DD_PIPE_TUNE'] = '1'
import torch  # noqa: E402
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'a1_vision'
cfgs = config_mod.load_configs()
cfg = config_mod.Config(cfgs['defaults']).update(cfgs[name]).update({'hip.pipeline': True})
if len(sys.argv) > 2:
  cfg = cfg.update({'batch_size': int(sys.argv[2])})
if len(sys.argv) > 3:
  cfg = cfg.update({'replay_chunk': int(sys.argv[3])})
plain = config_mod.to_plain(cfg)
B, T = plain['batch_size'], plain['replay_chunk']
obs, act = synthetic.config_spaces(name)
ag = agent_mod.Agent(obs, act, None, cfg)
data = synthetic.make_batch(obs, act, B, T, seed=0)
state = None
for _ in range(2):
  _, state, _ = ag.train(data, state)
torch.cuda.synchronize()
t0 = time.perf_counter()
state = ag.tune_pipeline(data, state, force=True)
ag.flush()
torch.cuda.synchronize()
print(f'tuning wall time {time.perf_counter() - t0:.2f} s')
pipe = ag._pipe
for pair, ms in sorted(pipe.periods.items(), key=lambda kv: kv[1]):
  print(f'pair {pair}: {ms:.2f} ms')
print('selected', agent_mod.Pipeline.BEST)
for rep in range(3):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(20):
    _, state, mets = ag.train(data, state)
  ag.flush()
  torch.cuda.synchronize()
  print(f'steady {1e3 * (time.perf_counter() - t0) / 20:.2f} ms per step')

# ---- every ordered pair at steady state (3 warm + 12 timed steps each, wall clock around a drained pipeline)
print('steady state per pair:')
for a, b in pipe.cands:
  ag.flush()
  torch.cuda.synchronize()
  pipe._use_pair(a, b)
  for _ in range(3):
    _, state, mets = ag.train(data, state)
  ag.flush()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(12):
    _, state, mets = ag.train(data, state)
  ag.flush()
  torch.cuda.synchronize()
  print(f'pair {(a, b)}: {1e3 * (time.perf_counter() - t0) / 12:.2f} ms per step')
