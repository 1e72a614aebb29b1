"""How much does the GPU gain from running two independent learner steps concurrently
(upper bound for overlapping step k's imagination phase with step k+1's world-model
phase)?  Two Agents, two host threads, each replaying its own HIP graphs on its own
stream; compare with one agent alone."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic

cfgs = agent_mod.Agent.configs
cfg = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision'])
obs, act = synthetic.make_spaces(64, 16, 16)
plain = config_mod.to_plain(cfg)
B, T = plain['batch_size'], plain['replay_chunk']
K = 20


lock = threading.Lock()


def worker(idx, out, start, go):
  try:
    _worker(idx, out, start, go)
  except Exception:
    import traceback
    traceback.print_exc()
    start.abort()


def _worker(idx, out, start, go):
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    with lock:  # build + capture one learner at a time
      ag = agent_mod.Agent(obs, act, None, cfg)
      data = synthetic.make_batch(obs, act, B, T, seed=idx)
      state = None
      for _ in range(3):
        _, state, _ = ag.train(data, state)
      torch.cuda.synchronize()
    start.wait()
    go[idx].wait()
    t0 = time.perf_counter()
    for _ in range(K):
      ag._plan.replay()
    s.synchronize()
    out[idx] = (time.perf_counter() - t0) / K * 1e3


for mode in ('alone', 'pair'):
  n = 1 if mode == 'alone' else 2
  out = [None] * n
  start = threading.Barrier(n)
  go = [threading.Event() for _ in range(n)]
  th = [threading.Thread(target=worker, args=(i, out, start, go)) for i in range(n)]
  [t.start() for t in th]
  go[0].set()
  if n == 2:
    time.sleep(0.025)  # offset the second learner by about half a step
    go[1].set()
  [t.join() for t in th]
  print(mode, 'ms per step per agent', out, '-> effective ms/step', max(out) / n, flush=True)
