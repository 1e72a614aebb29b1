"""Where the phases of consecutive pipelined train steps lie in time, WITHOUT a profiler (rocprofv3's
host overhead moves the world-model phase of the next step behind the rollout): DD_STAMPS=1 makes the
learner capture a one-thread kernel that stores the device's 100 MHz wall clock at the phase
boundaries.  Prints, for the steady state, each boundary relative to the start of the behaviour phase."""
import os, sys
os.environ['DD_STAMPS'] = '1'
os.environ.setdefault('DD_PIPE_TUNE', '0')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic
cfgs = config_mod.load_configs()
base = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision'])
for kv in sys.argv[1:]:
  k, v = kv.split('=')
  base = base.update({k: int(v) if v.lstrip('-').isdigit() else v})
obs, act = synthetic.config_spaces('a1_vision')
data = synthetic.make_batch(obs, act, base.batch_size, base.replay_chunk, seed=0)
ag = agent_mod.Agent(obs, act, None, base)
state = None
for _ in range(8):
  _, state, m = ag.train(data, state)
names = {0: 'A1 start (world-model forward)', 1: 'world-model forward done', 2: 'A1 done (gradients)', 3: 'A2 done (optimizer)',
         4: 'B start', 5: 'rollout start', 6: 'rollout done', 7: 'imagination + critic done', 8: 'B done (actor)'}
for _ in range(12):
  _, state, m = ag.train(data, state)
ag.flush(); torch.cuda.synchronize()
st = ag.learner.stamps.cpu().numpy().reshape(16, 9)
def stamp(slot, back):      # the stamp `back` steps before the last one
  n = int(st[slot, 0])
  return int(st[slot, 1 + (n - 1 - back) % 8])
# steady state: step k = the one before the last two; its behaviour phase runs next to A1 of step k + 1
for back in (3, 2):
  t0 = stamp(3, back)       # A2 done of step k
  print(f'step {back} before the last (ms after its optimizer step; period {(stamp(3, back - 1) - t0) / 1e5:.3f}):')
  for k in (4, 5, 6, 7, 8):
    print(f'  B(k)    {names[k]:34s} {(stamp(k, back) - t0) / 1e5:8.3f}')
  for k in (0, 1, 2, 3):
    print(f'  A(k+1)  {names[k]:34s} {(stamp(k, back - 1) - t0) / 1e5:8.3f}')
