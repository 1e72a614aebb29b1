"""2-rank data-parallel check of the two-stream pipeline on ONE GPU (gloo collectives on
device tensors): parameters after n steps must equal the sequential 2-rank schedule.
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_pipeline_check.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['LOCAL_RANK'] = '0'  # both ranks on the one visible GPU
import numpy as np
import torch
import torch.distributed as dist
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic

dist.init_process_group(os.environ.get('DD_DIST_BACKEND', 'gloo'))
rank = dist.get_rank()
cfgs = agent_mod.Agent.configs
cfg = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision']).update(cfgs['debug'])
cfg = cfg.update({'batch_size': 6, 'replay_chunk': 8, 'imag_horizon': 4})
obs, act = synthetic.make_spaces(64, 5, 3)
batches = [synthetic.make_batch(obs, act, 6, 8, seed=s, smooth_images=True) for s in range(3)]
res = {}
for mode in (False, True):
  ag = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': mode}))
  state = None
  for i in range(7):
    _, state, m = ag.train(batches[i % 3], state)
  last = ag.flush()
  res[mode] = (ag.save(), last if mode else m)
a, b = res[False][0], res[True][0]
bad = [k for k in a if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True)]
ma, mb = res[False][1], res[True][1]
badm = [k for k in ma if not np.array_equal(ma[k], mb[k], equal_nan=True)]
print(f'rank {rank}: {len(a)} arrays, mismatching {bad[:5]}, metrics mismatching {badm[:5]}, '
      f'model_loss {float(ma["model_loss"]):.4f} / {float(mb["model_loss"]):.4f}', flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad or badm else 0)
