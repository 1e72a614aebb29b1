"""Run ONE phase of the BASELINE configs[1] train step repeatedly (after two full warm-up
steps), for `rocprofv3 --kernel-trace --stats`: the per-kernel device time of the observe
scan, its reverse, the imagination rollout or the actor phase alone.

  rocprofv3 --kernel-trace --stats -d out -- python tools/scan_profile.py observe_fwd 20
"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import helpers
from daydreamer_amd import learner as LM, hipops, synthetic, config as config_mod, spec as spec_mod

phase, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = helpers.make_config(('a1_vision',))
plain = config_mod.to_plain(cfg)
obs, act = synthetic.config_spaces('a1_vision')
shapes = {k: v.shape for k, v in obs.items()}
sp = spec_mod.build_spec(plain, shapes, 16, False)
B, T = plain['batch_size'], plain['replay_chunk']
data = synthetic.make_batch(obs, act, B, T, seed=0)
L = LM.Learner(sp, hipops.HipOps('cuda:0'), 'cuda:0', B, T, params=spec_mod.init_params(sp, 0))
L.upload(data)
for i in range(2):
  L.train_step_device(use_carry=i > 0)
torch.cuda.synchronize()
fn = {'observe_fwd': lambda: L.observe_fwd(True), 'observe_bwd': L.observe_bwd,
      'imagine_rollout': L.imagine_rollout, 'actor': L.phase_actor,
      'imagine': L.phase_imagine}[phase]
if phase in ('imagine_rollout', 'actor', 'imagine'):
  L.ops, L._in_b = L.ops_b, True
for _ in range(reps):
  fn()
torch.cuda.synchronize()
print('done', phase, reps)
