"""Shared builders for the parity tests."""

import numpy as np
import torch

from daydreamer_amd import config, spec, synthetic


def make_config(blocks=('a1_vision', 'debug'), **overrides):
  cfgs = config.load_configs()
  cfg = config.Config(cfgs['defaults'])
  for b in blocks:
    cfg = cfg.update(cfgs[b])
  if overrides:
    cfg = cfg.update(overrides)
  return cfg


def make_problem(cfg, image=64, vector=5, action=3, batch=None, length=None,
                 seed=0, terminals=0.1, smooth=True, discrete=False, cameras=1):
  plain = config.to_plain(cfg)
  obs, act = synthetic.make_spaces(image, vector, action)
  for i in range(1, cameras):  # further cameras: `image2`, ... (matched by cnn_keys 'image')
    obs[f'image{i + 1}'] = synthetic.Space(np.uint8, (image, image, 3))
  if cameras > 1:  # keep the reference's key order (images first)
    obs = {**{k: v for k, v in obs.items() if k.startswith('image')},
           **{k: v for k, v in obs.items() if not k.startswith('image')}}
  shapes = {k: v.shape for k, v in obs.items()}
  sp = spec.build_spec(plain, shapes, action, discrete)
  params = spec.init_params(sp, seed)
  # non-trivial norm / bias parameters so their gradients are exercised
  rng = np.random.RandomState(seed + 1)
  for p in sp.params:
    if p.init == 'ones':
      params[p.name] = (1 + 0.1 * rng.randn(*p.shape)).astype(np.float32)
    elif p.init == 'zeros':
      params[p.name] = (0.1 * rng.randn(*p.shape)).astype(np.float32)
  B = batch or plain['batch_size']
  T = length or plain['replay_chunk']
  data = synthetic.make_batch(obs, act, B, T, seed=seed + 2,
                              terminals=terminals, smooth_images=smooth)
  if discrete:  # one-hot actions as embodied.wrappers.OneHotAction delivers them
    idx = np.random.RandomState(seed + 3).randint(0, action, (B, T))
    data['action'] = np.eye(action, dtype=np.float32)[idx]
  return plain, sp, shapes, params, data, B, T


def make_named_problem(name, batch, length, seed=0, terminals=0.01, horizon=None, **overrides):
  """A BASELINE.json workload by config block name (synthetic.config_spaces) with its own
  networks at full width; batch / length / horizon are the per-GPU shard under test."""
  cfg = make_config((name,), **overrides)
  if horizon is not None:
    cfg = cfg.update({'imag_horizon': horizon})
  plain = config.to_plain(cfg)
  obs, act = synthetic.config_spaces(name)
  shapes = {k: v.shape for k, v in obs.items()}
  adim = act['action'].shape[0]
  discrete = bool(getattr(act['action'], 'discrete', False))
  sp = spec.build_spec(plain, shapes, adim, discrete)
  params = spec.init_params(sp, seed)
  rng = np.random.RandomState(seed + 1)
  for p in sp.params:
    if p.init == 'ones':
      params[p.name] = (1 + 0.1 * rng.randn(*p.shape)).astype(np.float32)
    elif p.init == 'zeros':
      params[p.name] = (0.1 * rng.randn(*p.shape)).astype(np.float32)
  data = synthetic.make_batch(obs, act, batch, length, seed=seed + 2, terminals=terminals,
                              smooth_images=True)
  return plain, sp, shapes, params, data


def forced_from_learner(L):
  """Sample indices the learner drew, in the oracle's layout."""
  b = L.b
  B, T, N, H, D, F, G, C = L.B, L.T, L.N, L.H, L.D, L.F, L.G, L.C
  post = b['post'].view(B, T, F)[:, :, D:].reshape(B, T, G, C)
  prior = b['prior_stoch'].view(B, T, G, C)
  img = b['traj'][1:, :, D:F].reshape(H, N, G, C)
  extra = {}
  if L.discrete:
    extra['act'] = b['traj'][:, :, F:F + L.A].argmax(-1).cpu()
  return dict(
      **extra,
      obs_post=post.argmax(-1).permute(1, 0, 2).cpu(),
      obs_prior=prior.argmax(-1).permute(1, 0, 2).cpu(),
      img=img.argmax(-1).cpu())


def noise_from_learner(L):
  b = L.b
  extra = {}
  if L.discrete:
    extra['u_act'] = b['u_act'][..., 0].cpu().numpy()
  return dict(
      **extra,
      u_obs_prior=b['u_prior'].permute(1, 0, 2).cpu().numpy(), u_obs_post=b['u_post'].cpu().numpy(),
      u_img=b['u_img'].cpu().numpy(), eps_act=b['eps'].cpu().numpy())


def rel_err(a, b):
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def metrics_finite(mets):
  """Every metric finite, except the balance statistics that are NaN by definition when a
  minibatch has no positives / negatives (tfutils.py:396-398)."""
  legal_nan = ('_pos_loss', '_neg_loss', '_pos_acc', '_neg_acc')
  return all(np.isfinite(v) or k.endswith(legal_nan) for k, v in mets.items())
