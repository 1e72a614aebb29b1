"""Per-call-site timing of the contraction kernels in one eager train step."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import helpers
from daydreamer_amd import learner as LM, hipops
# usage: trace_shapes.py [top_n] [config batch length]   (default: a1_vision at its own batch / length)
if len(sys.argv) > 2:
  name, B, T = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
  plain, sp, shapes, params, data = helpers.make_named_problem(name, B, T, terminals=0.0)
else:
  cfg = helpers.make_config(('a1_vision',))
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, image=64, vector=16, action=16, terminals=0.0, smooth=False)
ops = hipops.HipOps('cuda:0')
L = LM.Learner(sp, ops, 'cuda:0', B, T, params=params)
L.upload(data)
for i in range(2):
  L.train_step_device(use_carry=i > 0)
torch.cuda.synchronize()
ops.trace = []
L.train_step_device(True)
torch.cuda.synchronize()
by = {}
for lab, f, e0, e1 in ops.trace:
  d = by.setdefault(lab, [0, 0.0, 0.0]); d[0] += 1; d[1] += f; d[2] += e0.elapsed_time(e1)
tot = sum(v[2] for v in by.values())
print('total traced ms', tot)
for lab, v in sorted(by.items(), key=lambda kv: -kv[1][2])[:int(sys.argv[1]) if len(sys.argv) > 1 else 45]:
  print(f'{v[2]:8.3f} ms n={v[0]:4d} {v[1]/v[2]/1e9:7.1f} TF  avg {1e3*v[2]/v[0]:8.1f} us  {lab}')
