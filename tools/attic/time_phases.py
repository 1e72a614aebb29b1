import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, numpy as np
import helpers
from daydreamer_amd import learner as LM, hipops
cfg = helpers.make_config(('a1_vision',))
plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, image=64, vector=16, action=16, terminals=0.0, smooth=False)
ops = hipops.HipOps('cuda:0')
L = LM.Learner(sp, ops, 'cuda:0', B, T, params=params)
print('buffers GB', L._nbytes/1e9)
L.upload(data)
for i in range(3):
  t=time.time(); L.train_step_device(use_carry=i>0); torch.cuda.synchronize(); print('eager step', time.time()-t)
# phase timing
def tm(f,*a):
  torch.cuda.synchronize(); t=time.time(); f(*a); torch.cuda.synchronize(); return time.time()-t
print('prep', tm(L.phase_prep)); print('wm_fwd', tm(L.phase_wm_fwd, True)); print('wm_bwd', tm(L.phase_wm_bwd)); print('imagine', tm(L.phase_imagine)); print('actor', tm(L.phase_actor))
print(L.read_metrics()['model_loss'])
# graph replay
plan = L.capture()
print('graphs', plan.n_graphs, [k for k,_ in plan.items])
for i in range(3):
  torch.cuda.synchronize(); t=time.time(); plan.replay(); torch.cuda.synchronize(); print('graph step', time.time()-t)
t=time.time()
for i in range(10): plan.replay()
torch.cuda.synchronize(); print('graph avg', (time.time()-t)/10)
m = L.read_metrics(); print(m['model_loss'], m['actor_loss'], m['model_grad_steps'])
