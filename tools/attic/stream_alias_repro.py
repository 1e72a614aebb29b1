"""Root-cause probe of the round-2 SIGSEGV: torch.cuda.Stream() hands out streams of a 32-entry
round-robin pool, so the prefetch thread's copy stream can BE the stream a GraphPlan captures on.
This script forces that aliasing (pre-fix code paths) and shows what happens.
usage: stream_alias_repro.py alias|noalias"""
import faulthandler, itertools, os, sys, threading
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers
from daydreamer_amd import agent as agent_mod, graphs, synthetic

mode = sys.argv[1]
shared = torch.cuda.Stream('cuda:0')
orig_stream = torch.cuda.Stream
main = threading.main_thread()
def patched(*a, **k):
  if mode == 'alias' and threading.current_thread() is not main:
    print('batcher thread gets the shared stream', flush=True)
    return shared
  return orig_stream(*a, **k)
torch.cuda.Stream = patched
orig_init = graphs.GraphPlan.__init__
def plan_init(self, device):
  orig_init(self, device)
  if mode == 'alias':
    self.stream = shared
graphs.GraphPlan.__init__ = plan_init

obs, act = synthetic.make_spaces(64, 5, 3)
for rep in range(6):
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6, imag_horizon=3)
  ag = agent_mod.Agent(obs, act, None, cfg)
  def gen():
    for s in itertools.count():
      ep = synthetic.make_batch(obs, act, 1, 6, seed=s % 5, smooth_images=True)
      yield {k: v[0] for k, v in ep.items()}
  ds = iter(ag.dataset(gen))
  state = None
  for i in range(8):
    batch = next(ds)
    _, state, mets = ag.train(batch, state)
    print(rep, i, float(mets['model_loss']), flush=True)
print('REPRO_DONE', flush=True)
os._exit(0)
