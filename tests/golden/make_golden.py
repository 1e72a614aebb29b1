"""Generates tests/golden/debug_step.npz: one learner step of the float64 CPU
oracle (oracle/dreamer_ref.py) on a tiny seeded problem, with the noise the
device RNG defines (oracle/ref_ops.philox_field = dd_philox restated).

The reference repository holds no golden vectors for this path (SURVEY.md
section 4 / 8c) and TensorFlow cannot run here, so these vectors pin OUR oracle
(regression) and the HIP path against it; they do not pin the oracle against
the reference.  Run:  python tests/golden/make_golden.py
"""

import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

import helpers  # noqa: E402
from daydreamer_amd import learner as LM  # noqa: E402
from oracle import dreamer_ref, ref_ops  # noqa: E402

NOISE_SEED = 11
PROBLEM = dict(image=64, vector=5, action=3, terminals=0.2)
CONFIG = dict(blocks=('a1_vision', 'debug'), batch_size=2, replay_chunk=3,
              imag_horizon=2)


def golden_noise(B, T, H, G, A, step):
  N = B * T
  f = ref_ops.philox_field
  return dict(
      # prior noise is generated batch-major (row = b*T + t)
      u_obs_prior=np.ascontiguousarray(f(B, T, G, T, 0, NOISE_SEED, step, LM.SITE_OBS_PRIOR, 0).transpose(1, 0, 2)),
      u_obs_post=f(T, B, G, B, 0, NOISE_SEED, step, LM.SITE_OBS_POST, 0),
      u_img=f(H, N, G, N, 0, NOISE_SEED, step, LM.SITE_IMG, 0),
      eps_act=f(H + 1, N, A, N, 0, NOISE_SEED, step, LM.SITE_ACT, 1))


def build():
  c = dict(CONFIG)
  cfg = helpers.make_config(c.pop('blocks'), **c)
  return helpers.make_problem(cfg, **PROBLEM)


def main():
  plain, sp, shapes, params, data, B, T = build()
  H = plain['imag_horizon']
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64)
  out = {}
  state = None
  for step in (1, 2):
    noise = golden_noise(B, T, H, sp.groups, sp.act_dim, step)
    _, state, mets = ag.train(data, noise, state)
    for k, v in mets.items():
      out[f's{step}/metric/{k}'] = np.float64(v)
    for k, g in ag.last['grads'].items():
      g = g.numpy()
      out[f's{step}/gradsum/{k}'] = np.array([g.sum(), np.abs(g).sum()])
    out[f's{step}/idx_post'] = ag.last['wm']['idxs']['post'].numpy()
    out[f's{step}/idx_prior'] = ag.last['wm']['idxs']['prior'].numpy()
    out[f's{step}/idx_img'] = ag.last['traj']['idx'].numpy()
    for k in ('rssm/initial_deter', 'actor/dist_out/std/kernel',
              'reward/dist_out/out/kernel', 'rssm/obs_stats/bias'):
      out[f's{step}/grad/{k}'] = ag.last['grads'][k].numpy()
  for k in ('rssm/img_in/norm/scale', 'critic/dense0/kernel',
            'critic_target/dense0/kernel'):
    out[f'final/param/{k}'] = ag.export_params()[k]
  path = pathlib.Path(__file__).parent / 'debug_step.npz'
  np.savez_compressed(path, **out)
  print('wrote', path, path.stat().st_size, 'bytes')


if __name__ == '__main__':
  main()
