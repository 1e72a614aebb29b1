"""Soak of the shipped default (`hip.pipeline: auto`) at configs[1]: a learner loop that mixes what
the reference's loops do around `Agent.train` - minibatches from a DeviceReplay dataset and from a
host generator through the prefetch thread, bursts of `policy` calls (the schedule switches to the
sequential plan and back), `report` at a log interval, metrics collected per call and aggregated
late (run/train.py:77-85), `save` / `load` round trips - and watches what must stay flat: device
memory, live graph executables, finite metrics, wall time per step.
  python tools/soak.py [steps]"""
import collections, itertools, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from daydreamer_amd import agent as agent_mod, config as config_mod, graphs, replay as replay_mod, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
cfgs = config_mod.load_configs()
cfg = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision'])
obs, act = synthetic.config_spaces('a1_vision')
B, T = cfg.batch_size, cfg.replay_chunk
ag = agent_mod.Agent(obs, act, None, cfg)
rep = replay_mod.DeviceReplay(chunk=T, capacity=60_000)
eps = synthetic.make_batch(obs, act, 16, 4 * T, seed=1, smooth_images=True, terminals=0.01)
for e in range(16):
  rep.add_traj({**{k: v[e] for k, v in eps.items()}, 'is_last': np.arange(4 * T) == 4 * T - 1})
ds_dev = ag.dataset(rep.dataset)
pool = [{k: v[e, j * T:(j + 1) * T] for k, v in eps.items()} for e in range(16) for j in range(4)]
def gen():   # (cheap: a generator that synthesises images would starve the learner thread of the GIL)
  for s in itertools.count(np.random.randint(64)):
    yield pool[s % 64]
ds_host = iter(ag.dataset(gen))
o = {k: v[:1, 0] for k, v in eps.items() if k not in ('action', 'reset')}
state, pst = None, None
coll = collections.defaultdict(list)
kinds = collections.Counter()
mem0 = None
t0 = time.perf_counter()
tlast, slast = t0, 0
for i in range(steps):
  burst = (i // 100) % 3 == 2            # every third block of 100 steps: act before every train call
  if burst:
    _, pst = ag.policy(o, pst, 'train')
  batch = next(ds_dev) if (i // 50) % 2 == 0 else next(ds_host)
  _, state, m = ag.train(batch, state)
  kinds[type(m).__name__] += 1
  for k, v in m.items():
    coll[k].append(v)
  if i % 150 == 149:
    r = ag.report(next(ds_dev))
    assert np.isfinite(r['model_loss_mean']) and r['openl_image'].shape[0] == T
  if i % 400 == 399:
    ck = ag.save()
    ag.load(ck)
  if i % 100 == 99:
    agg = {k: float(np.nanmean(v, dtype=np.float64)) for k, v in coll.items()}
    coll.clear()
    assert all(np.isfinite(agg[k]) for k in ('model_loss', 'actor_loss', 'extr_critic_loss', 'model_grad_norm')), agg
    torch.cuda.synchronize()
    mem = torch.cuda.memory_allocated() / 2**30
    if i + 1 == 200:   # (the first report builds its own learner: a second set of activation buffers, once)
      mem0 = mem
    now = time.perf_counter()
    print(f'step {i + 1:5d}  {1e3 * (now - tlast) / (i + 1 - slast):6.2f} ms/step  model_loss {agg["model_loss"]:9.3f}  '
          f'mem {mem:6.2f} GiB  live graphs {graphs.n_live_graphs():3d}  plan {type(ag._plan).__name__}  {dict(kinds)}', flush=True)
    tlast, slast = now, i + 1
    kinds.clear()
    assert mem0 is None or mem < mem0 + 0.5, (mem, mem0)
ag.flush()
print(f'SOAK_OK {steps} steps in {time.perf_counter() - t0:.1f} s')
