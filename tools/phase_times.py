"""Graph-replay time of every phase of one sequential train step (BASELINE configs[1] by
default): each phase is captured into its own HIP graph and replayed, so the numbers are
device time without host launch cost - the ground truth for where a step's 40-50 ms go
(observe scan forward / backward, imagination rollout / its backward, encoder, decoder, heads).

  python tools/phase_times.py [config] [batch] [length]
"""
import sys
import json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import helpers
from daydreamer_amd import learner as LM, hipops, graphs, synthetic, config as config_mod, spec as spec_mod

name = sys.argv[1] if len(sys.argv) > 1 else 'a1_vision'
cfg = helpers.make_config((name,))
if len(sys.argv) > 2:
  cfg = cfg.update({'batch_size': int(sys.argv[2])})
if len(sys.argv) > 3:
  cfg = cfg.update({'replay_chunk': int(sys.argv[3])})
plain = config_mod.to_plain(cfg)
obs, act = synthetic.config_spaces(name)
shapes = {k: v.shape for k, v in obs.items()}
A = act['action'].shape[0]
sp = spec_mod.build_spec(plain, shapes, A, bool(getattr(act['action'], 'discrete', False)))
B, T = plain['batch_size'], plain['replay_chunk']
data = synthetic.make_batch(obs, act, B, T, seed=0)
ops = hipops.HipOps('cuda:0')
# side launch contexts as the Agent creates them in its default (sequential) mode
L = LM.Learner(sp, ops, 'cuda:0', B, T, params=spec_mod.init_params(sp, 0),
               ops2=hipops.HipOps('cuda:0', ws_bytes=1024 << 20),
               ops_b2=hipops.HipOps('cuda:0', ws_bytes=1024 << 20))
L.upload(data)
for i in range(2):
  L.train_step_device(use_carry=i > 0)
torch.cuda.synchronize()


def timed(label, fn, reps=5):
  plan = graphs.GraphPlan('cuda:0')
  keep, L.plan = L.plan, plan
  plan.capture(fn)
  L.plan = keep
  for _ in range(2):
    plan.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    plan.replay()
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  print(f'{label:28s} {ms:8.3f} ms', flush=True)
  return ms


b = L.b
feat = b['post']
out = {}
out['prep'] = timed('phase_prep', L.phase_prep)
out['encoder_fwd'] = timed('encoder_fwd', L.encoder_fwd)
out['initial_fwd'] = timed('initial_fwd', L.initial_fwd)
out['observe_fwd'] = timed('observe_fwd (T steps)', lambda: L.observe_fwd(True))
out['decoder_fwd'] = timed('decoder_fwd', lambda: L.decoder_fwd(feat))
out['wm_fwd'] = timed('phase_wm_fwd (all)', lambda: L.phase_wm_fwd(True))
out['observe_bwd'] = timed('observe_bwd (T steps)', L.observe_bwd)
out['encoder_bwd'] = timed('encoder_bwd', L.encoder_bwd)
out['wm_bwd'] = timed('phase_wm_bwd (all)', L.phase_wm_bwd)
out['wm_opt'] = timed('phase_wm_opt', L.phase_wm_opt)
L.ops, L._in_b = L.ops_b, True
out['prep_b'] = timed('phase_prep_b', L.phase_prep_b)
out['imagine_rollout'] = timed('imagine_rollout (H steps)', L.imagine_rollout)
out['imagine'] = timed('phase_imagine (all)', L.phase_imagine)
out['actor'] = timed('phase_actor (all)', L.phase_actor)
L.ops, L._in_b = L.ops_a, False
out['step'] = timed('train_step_device', lambda: L.train_step_device(True))
print(json.dumps({k: round(v, 3) for k, v in out.items()}))
