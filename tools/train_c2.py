"""Trains the BASELINE config on a small fixed synthetic replay (8 different
batches, cycled) for a few hundred steps; prints the metric trajectory."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfgs = config_mod.load_configs()
cfg = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision'])
obs, act = synthetic.make_spaces(64, 16, 16)
ag = agent_mod.Agent(obs, act, None, cfg)
batches = [synthetic.make_batch(obs, act, 50, 50, seed=s, smooth_images=True, terminals=0.01) for s in range(8)]
state = ag.tune_pipeline(batches[0])  # finish the stream-pair trial steps (real train steps, untimed)
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
  _, state, m = ag.train(batches[i % 8], state)
  if i % 25 == 0 or i == steps - 1:
    torch.cuda.synchronize()
    print(json.dumps({'step': i, 'wall_s': round(time.time() - t0, 2),
                      **{k: round(float(m[k]), 4) for k in ('model_loss', 'image_loss_mean', 'kl_loss_mean', 'reward_loss_mean',
                         'extr_critic_loss', 'actor_loss', 'model_grad_norm', 'actor_grad_norm', 'wmkl_scale_mean',
                         'actent_mean', 'actent_scale_mean', 'extr_imag_return_mean')}}), flush=True)
