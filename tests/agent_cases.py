"""Shared bodies of the Agent.policy / Agent.report / checkpoint tests: run on CPU through the
kernel restatements in float64 (tests/test_agent_api.py, host logic) and on the MI355X through
the HIP kernels (tests/test_agent_gpu.py), both against oracle/dreamer_ref.RefAgent."""

import numpy as np
import torch

from daydreamer_amd import agent as agent_mod, synthetic
from oracle import dreamer_ref
import helpers


def make_agent(cfg, obs, act, backend):
  """backend: None -> HIP product path; a RefOps instance -> CPU float64 host-logic path."""
  if backend is None:
    return agent_mod.Agent(obs, act, None, cfg)
  return agent_mod.Agent(obs, act, None, cfg, _ops=backend, _device='cpu', _dtype=torch.float64)


def ref_agent(ag):
  shapes = {k: tuple(v.shape) for k, v in ag.obs_space.items()}
  ag._ensure_params()
  return dreamer_ref.RefAgent(ag.cfg, shapes, ag.act_dim, ag.learner.export_params(),
                              torch.float64, act_discrete=ag.act_discrete)


def spaces(discrete, image=64, vector=5, action=3):
  obs, act = synthetic.make_spaces(image, vector, action)
  if discrete:
    act['action'].discrete = True
  return obs, act


def with_cnn(cfg, cnn):
  """cnn = 'resnet': the residual encoder / decoder (reference nets.py:330-391) at a small depth."""
  if cnn == 'simple':
    return cfg
  return cfg.update({'encoder.cnn': cnn, 'decoder.cnn': cnn, 'encoder.cnn_depth': 4,
                     'decoder.cnn_depth': 4, 'encoder.cnn_blocks': 1, 'decoder.cnn_blocks': 1})


def policy_parity(backend, discrete, tol, noise_amount=0.0, cnn='simple'):
  """Three consecutive policy calls (initial state, carried state, reset by is_first) in the
  'train' and 'eval' modes against RefAgent.policy with the learner's own noise."""
  dreamer_ref.SAMPLE_TOL[0] = tol['sample']
  cfg = helpers.make_config(('a1_vision', 'debug'))
  cfg = cfg.update({'expl_noise': noise_amount, 'eval_noise': noise_amount / 2})
  cfg = with_cnn(cfg, cnn)
  A = 4 if discrete else 3
  obs_space, act_space = spaces(discrete, action=A)
  ag = make_agent(cfg, obs_space, act_space, backend)
  ref = ref_agent(ag)
  n = 3
  rng = np.random.RandomState(0)
  st, rst = None, None
  for i, mode in enumerate(('train', 'eval', 'train', 'explore')):
    obs = {'image': rng.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8),
           'vector': rng.randn(n, 5).astype(np.float32),
           'reward': rng.randn(n).astype(np.float32),
           'is_first': np.array([i == 0, i == 2, False]),
           'is_last': np.zeros(n, bool), 'is_terminal': np.zeros(n, bool)}
    out, st = ag.policy(obs, st, mode)
    P = ag._policies[n]
    b = P.b
    G, C, D, F = P.G, P.C, P.D, P.F
    noise = dict(u_prior=b['u_prior'][:, 0].cpu().numpy(), u_post=b['u_post'][0].cpu().numpy(),
                 eps=b['eps'][0].cpu().numpy())
    forced = dict(post=b['post'][:, D:].reshape(n, G, C).argmax(-1).cpu())
    amount = cfg['eval_noise'] if mode == 'eval' else cfg['expl_noise']
    if discrete:
      noise['u_act'] = b['u_act'][0, :, 0].cpu().numpy()
      if amount:
        noise['act_noise'] = b['act_noise'][:, 0].cpu().numpy()
        forced['act_noise'] = torch.as_tensor(out['action']).argmax(-1)
      elif mode != 'eval':
        forced['act'] = torch.as_tensor(out['action']).argmax(-1)
    elif amount:
      noise['act_noise'] = b['act_noise'].cpu().numpy()
    rout, rst = ref.policy(obs, rst, noise, mode, forced)
    a, o = out['action'].astype(np.float64), rout['action'].numpy()
    assert a.shape == (n, A) and out['action'].dtype == np.float32
    assert np.abs(a - o).max() <= tol['action'], (i, mode, np.abs(a - o).max())
    lat = st.latent.cpu().numpy()
    assert np.abs(lat[:, :D] - rst[0]['deter'].numpy()).max() <= tol['latent'], (i, mode)
    assert np.array_equal(lat[:, D:].reshape(n, G, C).argmax(-1),
                          rst[0]['stoch'].detach().reshape(n, G, C).argmax(-1).numpy())
    if discrete:
      assert np.array_equal(a.sum(-1), np.ones(n))
    else:
      assert np.abs(a).max() <= (1.0 if amount else 10.0)
  return ag


def report_parity(backend, discrete, tol, device_batch=False, cameras=1, cnn='simple', calls=1):
  """Agent.report against RefAgent.report: world-model loss metrics, open-loop grids and the
  Greedy behaviour's imagined-rollout grids; nothing in the agent's state may change.
  calls: consecutive reports (GPU: the first runs eagerly, the second also captures the launch
  sequence into a HIP graph, later ones replay it); the last one is compared."""
  dreamer_ref.SAMPLE_TOL[0] = tol['sample']
  dreamer_ref.SAMPLE_STATS.update(draws=0, adopted=0)
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=7, replay_chunk=8, imag_horizon=3)
  cfg = with_cnn(cfg, cnn)
  A = 4 if discrete else 3
  obs_space, act_space = spaces(discrete, action=A)
  for i in range(1, cameras):
    obs_space[f'image{i + 1}'] = synthetic.Space(np.uint8, (64, 64, 3))
  ag = make_agent(cfg, obs_space, act_space, backend)
  ref = ref_agent(ag)
  B, T, H = 7, 8, 3
  data = synthetic.make_batch(obs_space, act_space, B, T, seed=1, smooth_images=True, terminals=0.1)
  if discrete:
    idx = np.random.RandomState(3).randint(0, A, (B, T))
    data['action'] = np.eye(A, dtype=np.float32)[idx]
  before = ag.save()
  feed = data
  if device_batch:  # what Agent.dataset yields on the GPU
    feed = {k: torch.from_numpy(v).to(ag.device) for k, v in data.items()}
  for _ in range(calls):
    rep = ag.report(feed)
  after = ag.save()
  for k in before:
    assert np.array_equal(np.asarray(before[k]), np.asarray(after[k])), k
  built = ag._policies[('report', B, T)]
  assert (built['plan'] is not None) == (backend is None and calls >= 2)   # replayed from a HIP graph
  R = built['R']
  roll, _ = built['imag']
  G, C, D, F = R.G, R.C, R.D, R.F
  ctx, n = 5, 6
  u_img = R.b['u_img'].reshape(-1, G)
  noise = dict(
      u_obs_prior=R.b['u_prior'].permute(1, 0, 2).cpu().numpy(), u_obs_post=R.b['u_post'].cpu().numpy(),
      u_openl=torch.stack([u_img[i * B:i * B + n] for i in range(T - ctx)]).cpu().numpy(),
      u_img=roll.b['u_img'].cpu().numpy(), eps_act=roll.b['eps'].cpu().numpy())
  tr = R.b['traj'].view(-1, R.TW)
  forced = dict(
      obs_post=R.b['post'].view(B, T, F)[:, :, D:].reshape(B, T, G, C).argmax(-1).permute(1, 0, 2).cpu(),
      obs_prior=R.b['prior_stoch'].view(B, T, G, C).argmax(-1).permute(1, 0, 2).cpu(),
      openl=torch.stack([tr[(i + 1) * B:(i + 1) * B + n, D:F].reshape(n, G, C).argmax(-1)
                         for i in range(T - ctx)]).cpu(),
      img=roll.b['traj'][1:, :, D:F].reshape(H, n, G, C).argmax(-1).cpu())
  if discrete:
    noise['u_act'] = roll.b['u_act'][..., 0].cpu().numpy()
    forced['act'] = roll.b['traj'][:, :, F:F + R.A].argmax(-1).cpu()
  want = ref.report(data, noise, forced)
  keys = [f'openl_{k}' for k in ag.spec.dec_cnn_keys] + [f'task_imag_{k}' for k in ag.spec.dec_cnn_keys]
  for k in keys:
    assert rep[k].dtype == np.float32 and rep[k].shape == tuple(want[k].shape), (k, rep[k].shape)
    err = np.abs(rep[k].astype(np.float64) - want[k].numpy()).max()
    assert err <= tol['video'], (k, err)
  assert rep['openl_image'].shape == (T, 3 * 64, 6 * 64, 3)
  assert rep['task_imag_image'].shape == (H + 1, 64, 6 * 64, 3)
  for k, v in want.items():
    if k in keys:
      continue
    assert k in rep, k
    a, o = float(rep[k]), float(v)
    if np.isnan(o):
      assert np.isnan(a), k
    else:
      assert abs(a - o) <= tol['metric'] * max(1.0, abs(o)), (k, a, o)
  return dreamer_ref.SAMPLE_STATS['adopted'], dreamer_ref.SAMPLE_STATS['draws']


def load_before_train_keeps_controller_state(backend):
  """load() on a fresh agent (bootstrap learner), then train(): AutoAdapt scales,
  Normalize moments, the slow-critic counter and the noise step must be the checkpoint's."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6, imag_horizon=3)
  obs, act = spaces(False)
  data = synthetic.make_batch(obs, act, 4, 6, seed=2, smooth_images=True)
  a = make_agent(cfg, obs, act, backend)
  state = None
  for _ in range(3):
    _, state, _ = a.train(data, state)
  a.flush()
  ckpt = a.save()
  # continue the original for one step (from the initial recurrent state)
  _, _, m_a = a.train(data, None)
  a.flush()
  b = make_agent(cfg, obs, act, backend)
  b.load(ckpt)                      # before any train(): goes to the bootstrap learner
  b.policy({k: v[:, 0] for k, v in data.items() if k not in ('action', 'reset')})
  _, _, m_b = b.train(data, None)
  b.flush()
  sa, sb = a.save(), b.save()
  assert int(sb['state/slow_updates']) == int(sa['state/slow_updates']) == 4
  for k in sa:
    assert np.array_equal(np.asarray(sa[k]), np.asarray(sb[k]), equal_nan=True), k
