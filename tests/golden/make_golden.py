"""Generates tests/golden/debug_step.npz (continuous actions, actor by backprop, simple CNN),
onehot_step.npz (one-hot actions, actor by REINFORCE) and resnet_step.npz (residual encoder /
decoder): two learner steps of the float64 CPU oracle (oracle/dreamer_ref.py) on a tiny seeded
problem, with the noise the device RNG defines (oracle/ref_ops.philox_field = dd_philox restated).

The reference repository holds no golden vectors for this path (SURVEY.md
section 4 / 8c) and TensorFlow cannot run here, so these vectors pin OUR oracle
(regression) and the HIP path against it; they do not pin the oracle against
the reference.  Run:  python tests/golden/make_golden.py [debug|onehot|resnet ...]
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
RESNET = {'encoder.cnn': 'resnet', 'decoder.cnn': 'resnet', 'encoder.cnn_depth': 4,
          'decoder.cnn_depth': 4, 'encoder.cnn_blocks': 1, 'decoder.cnn_blocks': 1}
# name -> (problem overrides, config overrides)
CASES = dict(
    debug=(dict(), dict()),
    onehot=(dict(action=4, discrete=True), dict()),
    resnet=(dict(), RESNET))


def golden_noise(B, T, H, G, A, step):
  N = B * T
  f = ref_ops.philox_field
  return dict(
      u_act=f(H + 1, N, 1, N, 0, NOISE_SEED, step, LM.SITE_ACT, 0)[..., 0],
      # prior noise is generated batch-major (row = b*T + t)
      u_obs_prior=np.ascontiguousarray(f(B, T, G, T, 0, NOISE_SEED, step, LM.SITE_OBS_PRIOR, 0).transpose(1, 0, 2)),
      u_obs_post=f(T, B, G, B, 0, NOISE_SEED, step, LM.SITE_OBS_POST, 0),
      u_img=f(H, N, G, N, 0, NOISE_SEED, step, LM.SITE_IMG, 0),
      eps_act=f(H + 1, N, A, N, 0, NOISE_SEED, step, LM.SITE_ACT, 1))


def build(case='debug'):
  c = dict(CONFIG)
  pover, cover = CASES[case]
  cfg = helpers.make_config(c.pop('blocks'), **c)
  if cover:
    cfg = cfg.update(cover)
  return helpers.make_problem(cfg, **{**PROBLEM, **pover})


def make_ref(case, plain, sp, shapes, params):
  return dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                              act_discrete=bool(CASES[case][0].get('discrete', False)))


def grad_keys(case):
  keys = ['rssm/initial_deter', 'reward/dist_out/out/kernel', 'rssm/obs_stats/bias']
  if case != 'onehot':
    keys.append('actor/dist_out/std/kernel')
  else:
    keys.append('actor/dist_out/out/kernel')
  return keys


def generate(case):
  plain, sp, shapes, params, data, B, T = build(case)
  H = plain['imag_horizon']
  ag = make_ref(case, plain, sp, shapes, params)
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
    if case == 'onehot':
      out[f's{step}/idx_act'] = ag.last['traj']['action'].argmax(-1).numpy()
    for k in grad_keys(case):
      out[f's{step}/grad/{k}'] = ag.last['grads'][k].numpy()
  for k in ('rssm/img_in/norm/scale', 'critic/dense0/kernel',
            'critic_target/dense0/kernel'):
    out[f'final/param/{k}'] = ag.export_params()[k]
  path = pathlib.Path(__file__).parent / f'{case}_step.npz'
  np.savez_compressed(path, **out)
  print('wrote', path, path.stat().st_size, 'bytes')


if __name__ == '__main__':
  for case_ in (sys.argv[1:] or list(CASES)):
    generate(case_)
