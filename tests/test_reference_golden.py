"""The oracle against THE REFERENCE'S OWN SOURCES, executed.

tests/golden/reference_*.npz were produced by importing the reference's agent.py / nets.py /
tfutils.py / tfagent.py / behaviors.py unmodified from /root/reference and running two consecutive
`Agent.train` calls on a TensorFlow stand-in (oracle/tf_on_torch.py: every tf / tfd / sonnet
primitive the sources reach, written on torch from the TensorFlow documentation; float64; draws by
inverse CDF from injected uniforms) - tests/golden/make_reference_golden.py, run in the container
that holds the reference.  So every line the reference WROTE is in the loop here: module wiring,
scan, stop-gradients, the straight-through sample, KL balance, loss scales, lambda-returns (gve and
gae), Normalize, AutoAdapt, the hand-written Adam with clip / weight decay, the slow
critic and the order of the three updates; what is not is the library underneath the primitives
(pinned separately, without torch, in tests/test_oracle_pins.py and test_oracle_independent.py).

Cases: continuous actions with the actor trained by backprop (`debug`), one-hot actions with
REINFORCE (`onehot`), residual encoder / decoder (`resnet`), and `decay` = weight decay on kernels,
a gradient clip that bites and GAE returns.  The float64 oracle
(oracle/dreamer_ref.RefAgent) must reproduce, from the same initial parameters, data and noise:
every metric the reference returned (1e-9), the classes it drew (exactly), the gradient its tape
handed to each optimizer (1e-9 of the gradient's |sum|), the parameters after each of the two steps
and the controller state (AutoAdapt scales, Normalize moments, slow-critic counter).  The HIP path
is held to the oracle elsewhere (tests/test_golden.py, test_learner_gpu.py), and to these vectors
directly in test_hip_path_matches_reference_run.
"""

import importlib.util
import pathlib

import numpy as np
import pytest
import torch

HERE = pathlib.Path(__file__).parent
_spec = importlib.util.spec_from_file_location('make_reference_golden', HERE / 'golden' / 'make_reference_golden.py')
mrg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mrg)
mg = mrg.mg
from oracle import dreamer_ref  # noqa: E402

CASES = list(mrg.CASES)
GOLDS = {c: np.load(HERE / 'golden' / f'reference_{c}.npz') for c in CASES}
# metrics of the reference's mixed-precision bookkeeping do not exist in float32 mode
SKIP = ('_grad_scale', '_grad_overflow')


def _oracle(case):
  base, (plain, sp, shapes, params, data, B, T) = mrg.build(case)
  discrete = mrg.spaces_of(base)[2]
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                            act_discrete=discrete, ctrl_dtype=torch.float64)
  return ag, plain, sp, data, B, T, discrete


def _close(a, b, tol, what):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  assert a.shape == b.shape, (what, a.shape, b.shape)
  if np.isnan(b).any():
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    a, b = np.nan_to_num(a), np.nan_to_num(b)
  err = np.abs(a - b).max() if a.size else 0.0
  assert err <= tol * max(1.0, np.abs(b).max() if b.size else 0.0), (what, float(err))


@pytest.mark.parametrize('case', CASES)
def test_oracle_reproduces_the_reference_run(case):
  GOLD = GOLDS[case]
  ag, plain, sp, data, B, T, discrete = _oracle(case)
  H, G, A = plain['imag_horizon'], sp.groups, sp.act_dim
  dreamer_ref.SAMPLE_STATS.update(draws=0, adopted=0, max_gap=0.0)
  state, n_metrics, n_grads = None, 0, 0
  for step in (1, 2):
    noise = mg.golden_noise(B, T, H, G, A, step)
    forced = dict(obs_prior=GOLD[f's{step}/idx_prior'], obs_post=GOLD[f's{step}/idx_post'],
                  img=GOLD[f's{step}/idx_img'])
    if discrete:
      forced['act'] = GOLD[f's{step}/idx_act']
    _, state, mets = ag.train(data, noise, state, forced=forced)
    # ---- the classes the reference drew
    assert np.array_equal(ag.last['wm']['idxs']['prior'].numpy(), GOLD[f's{step}/idx_prior'])
    assert np.array_equal(ag.last['wm']['idxs']['post'].numpy(), GOLD[f's{step}/idx_post'])
    assert np.array_equal(ag.last['traj']['idx'].numpy(), GOLD[f's{step}/idx_img'])
    if discrete:
      assert np.array_equal(ag.last['traj']['action'].argmax(-1).numpy(), GOLD[f's{step}/idx_act'])
    # ---- every metric the reference returned
    ref_keys = [k[len(f's{step}/metric/'):] for k in GOLD.files if k.startswith(f's{step}/metric/')]
    ref_keys = [k for k in ref_keys if not k.endswith(SKIP)]
    missing = [k for k in ref_keys if k not in mets]
    assert not missing, f'metrics the reference returns and the oracle does not: {missing}'
    extra = [k for k in mets if k not in ref_keys]
    assert not extra, f'metrics the oracle returns and the reference does not: {extra}'
    for k in ref_keys:
      _close(float(mets[k]), float(GOLD[f's{step}/metric/{k}']), 1e-9, (step, 'metric', k))
      n_metrics += 1
    # ---- the gradient the reference's tape handed to each optimizer
    grads = ag.last['grads']
    gkeys = [k[len(f's{step}/gradsum/'):] for k in GOLD.files if k.startswith(f's{step}/gradsum/')]
    assert sorted(gkeys) == sorted(grads), sorted(set(gkeys) ^ set(grads))
    for name in gkeys:
      ref = GOLD[f's{step}/gradsum/{name}']
      g = grads[name].numpy()
      assert abs(g.sum() - ref[0]) <= 1e-9 * max(ref[1], 1e-30), (step, 'grad sum', name)
      assert abs(np.abs(g).sum() - ref[1]) <= 1e-9 * max(ref[1], 1e-30), (step, 'grad |sum|', name)
      n_grads += 1
    for k in [k for k in GOLD.files if k.startswith(f's{step}/grad/')]:
      name = k[len(f's{step}/grad/'):]
      _close(grads[name].numpy(), GOLD[k], 1e-9, (step, 'grad', name))
    # ---- parameters after the step (Adam, clip, decay, slow critic)
    now = ag.export_params()
    for k in [k for k in GOLD.files if k.startswith(f's{step}/paramsum/')]:
      name = k[len(f's{step}/paramsum/'):]
      p = np.asarray(now[name], np.float64)
      assert abs(p.sum() - GOLD[k][0]) <= 1e-10 * max(GOLD[k][1], 1e-30), (step, 'param sum', name)
      assert abs(np.abs(p).sum() - GOLD[k][1]) <= 1e-10 * max(GOLD[k][1], 1e-30), (step, 'param |sum|', name)
    for k in [k for k in GOLD.files if k.startswith(f's{step}/param/')]:
      _close(now[k[len(f's{step}/param/'):]], GOLD[k], 1e-11, (step, k))
    # ---- the carried state and the controllers
    for k in ('deter', 'stoch', 'logit'):
      v = state[k].detach().numpy()
      if v.size > mrg.FULL_MAX:
        v = v.reshape(-1)[::max(1, v.size // mrg.FULL_MAX)]
      _close(v, GOLD[f's{step}/state/{k}'], 1e-9, (step, 'state', k))
    _close(ag.wmkl.scale.numpy(), GOLD[f's{step}/ctrl/wmkl_scale'], 1e-12, (step, 'wmkl scale'))
    _close(ag.actent.scale.numpy(), GOLD[f's{step}/ctrl/actent_scale'], 1e-12, (step, 'actent scale'))
    for nm in ('advnorm', 'retnorm', 'scorenorm'):
      n = getattr(ag, nm)
      _close([float(n.mean), float(n.sqrs), float(n.step)], GOLD[f's{step}/ctrl/{nm}'], 1e-9, (step, nm))
    assert int(ag.slow_updates) == int(GOLD[f's{step}/ctrl/slow_updates']), step
  assert n_metrics >= 2 * 55 and n_grads >= 2 * 100, (n_metrics, n_grads)
  # the draws are the oracle's own (same inverse-CDF rule on the same float64 probabilities):
  # nothing had to be adopted from the reference run
  assert dreamer_ref.SAMPLE_STATS['adopted'] == 0, dreamer_ref.SAMPLE_STATS


def _learner(case, ops, device, dtype):
  from daydreamer_amd import learner as LM
  base, (plain, sp, shapes, params, data, B, T) = mrg.build(case)
  L = LM.Learner(sp, ops, device, B, T, params=params, noise_seed=mg.NOISE_SEED, dtype=dtype)
  return base, L, data


PRODUCT_CASES = [c for c in CASES if c not in mrg.ORACLE_ONLY]


def test_product_rejects_what_only_the_oracle_restates():
  from daydreamer_amd import learner as LM
  from oracle import ref_ops
  for case in mrg.ORACLE_ONLY:
    base, (plain, sp, shapes, params, data, B, T) = mrg.build(case)
    with pytest.raises(AssertionError):
      LM.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params)


@pytest.mark.parametrize('case', PRODUCT_CASES)
def test_learner_host_logic_matches_the_reference_run(case):
  """The product's learner (its phase orchestration, hoisting, deferred weight gradients, flat
  optimizer arenas, device RNG) on the CPU restatement of the kernels in float64, against the
  reference run: same draws, metrics and per-parameter gradients."""
  from oracle import ref_ops
  from test_golden import check_learner
  base, L, data = _learner(case, ref_ops.RefOps('cpu'), 'cpu', torch.float64)
  check_learner(L, data, 1e-6, 1e-6, True, base, gold=GOLDS[case], all_metrics=True)


@pytest.mark.gpu
@pytest.mark.parametrize('case', PRODUCT_CASES)
def test_hip_path_matches_reference_run(hip, case):
  """The HIP kernels (fp32 contractions on the split-bf16 matrix pipe) against the reference run:
  every class drawn as the reference's sources drew it, metrics to 1e-3, gradients to 1e-3 of
  their |sum|, over two learner steps."""
  from test_golden import check_learner
  base, L, data = _learner(case, hip, 'cuda:0', torch.float32)
  check_learner(L, data, 1e-3, 1e-3, False, base, gold=GOLDS[case], all_metrics=True)


@pytest.mark.parametrize('case', ('debug', 'onehot'))
def test_oracle_policy_reproduces_the_reference_run(case):
  """reference Agent.policy (agent.py:42-65) called four times with the carried state - sampled,
  explore, mode, sampled action, each followed by tfutils.action_noise (expl_noise 0.2 /
  eval_noise 0.1) - against RefAgent.policy: actions and the carried latent at 1e-9."""
  GOLD = np.load(HERE / 'golden' / f'reference_policy_{case}.npz')
  base, (plain, sp, shapes, params, data, B, T) = mrg.build(
      case, batch=3, length=len(mrg.POLICY_MODES), extra=mrg.POLICY_EXTRA)
  discrete = mrg.spaces_of(base)[2]
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                            act_discrete=discrete, ctrl_dtype=torch.float64)
  state = None
  for t, mode in enumerate(mrg.POLICY_MODES):
    obs = {k: v[:, t] for k, v in data.items() if k not in ('action', 'reset')}
    noise = mrg.policy_noise(B, sp.groups, sp.act_dim, discrete, t)
    outs, state = ag.policy(obs, state, noise, mode)
    _close(outs['action'].numpy(), GOLD[f'c{t}/action'], 1e-9, (t, mode, 'action'))
    for k in ('deter', 'stoch', 'logit'):
      _close(state[0][k].numpy(), GOLD[f'c{t}/latent/{k}'], 1e-9, (t, mode, k))
    assert np.array_equal(state[0]['stoch'].argmax(-1).numpy(), GOLD[f'c{t}/idx_post'])
  acts = np.stack([GOLD[f'c{t}/action'] for t in range(len(mrg.POLICY_MODES))])
  assert np.abs(acts).max() <= 1.0
  if discrete:
    assert ((acts == 0) | (acts == 1)).all() and (acts.sum(-1) == 1).all()
    assert len({tuple(a.argmax(-1)) for a in acts}) > 1
  else:
    assert len(np.unique(acts.round(6))) > 4


@pytest.mark.parametrize('case', ('debug', 'onehot'))
def test_oracle_report_reproduces_the_reference_run(case):
  """reference Agent.report (agent.py:95-106, 266-282, behaviors.py:32-46) against
  RefAgent.report: the loss metrics without update, the reconstruction / open-loop video and the
  imagined-rollout video (per-frame sums and a strided sample of every grid) at 1e-9."""
  GOLD = np.load(HERE / 'golden' / f'reference_report_{case}.npz')
  base, (plain, sp, shapes, params, data, B, T) = mrg.build(case, **mrg.REPORT_SHAPE)
  discrete = mrg.spaces_of(base)[2]
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                            act_discrete=discrete, ctrl_dtype=torch.float64)
  noise = mrg.report_noise(B, T, plain['imag_horizon'], sp.groups, sp.act_dim, min(6, B))
  rep = ag.report({k: v for k, v in data.items() if k != 'reset'}, noise)
  ref_metrics = [k[len('metric/'):] for k in GOLD.files if k.startswith('metric/')]
  ref_videos = sorted({k.split('/')[1] for k in GOLD.files if k.startswith('video/')})
  assert sorted(rep) == sorted(ref_metrics + ref_videos), sorted(set(rep) ^ set(ref_metrics + ref_videos))
  assert ref_videos == ['openl_image', 'task_imag_image'] and len(ref_metrics) >= 30
  for k in ref_metrics:
    _close(float(rep[k]), float(GOLD[f'metric/{k}']), 1e-9, ('metric', k))
  for k in ref_videos:
    want = {kk: GOLD[f'video/{k}/{kk}'] for kk in ('shape', 'sums', 'abssums', 'sample')}
    got = mrg.video_digest(rep[k].numpy())
    assert np.array_equal(got['shape'], want['shape']), (k, got['shape'], want['shape'])
    for kk in ('sums', 'abssums', 'sample'):
      _close(got[kk], want[kk], 1e-9, (k, kk))


@pytest.mark.parametrize('case', ('debug', 'onehot'))
def test_agent_train_and_report_return_the_reference_key_sets(case):
  """The product's Agent (CPU restatement of the kernels through the test-only backend seam):
  `train` returns exactly the metric names the reference's train returned (minus its float16
  loss-scale bookkeeping), `report` exactly the reference's report keys."""
  import helpers
  from daydreamer_amd import agent as agent_mod
  from oracle import ref_ops
  base, (plain, sp, shapes, params, data, B, T) = mrg.build(case, **mrg.REPORT_SHAPE)
  obs, act, _ = mrg.spaces_of(base)
  c = dict(mg.CONFIG)
  cfg = helpers.make_config(c.pop('blocks'), **c)
  ag = agent_mod.Agent(obs, act, None, cfg, _ops=ref_ops.RefOps('cpu'), _device='cpu', _dtype=torch.float64)
  batch = {k: v for k, v in data.items() if k != 'reset'}
  _, _, mets = ag.train(batch, None)
  G = GOLDS[case]
  want = {k[len('s1/metric/'):] for k in G.files if k.startswith('s1/metric/') and not k.endswith(SKIP)}
  assert set(mets) == want, sorted(set(mets) ^ want)
  rep = ag.report(batch)
  R = np.load(HERE / 'golden' / f'reference_report_{case}.npz')
  want = {k[len('metric/'):] for k in R.files if k.startswith('metric/')} | \
         {k.split('/')[1] for k in R.files if k.startswith('video/')}
  assert set(rep) == want, sorted(set(rep) ^ want)
  for k in ('openl_image', 'task_imag_image'):
    assert tuple(np.asarray(rep[k]).shape) == tuple(R[f'video/{k}/shape']), k


@pytest.mark.parametrize('case', ('debug', 'xarm'))
def test_agent_returns_the_reference_priorities(case):
  """Prioritized replay (agent.py:89-93): a batch that carries replay keys comes back as
  outs = {key, priority}, priority = the per-step loss `config.priority` names (reward_loss).
  The product's Agent (CPU restatement of the kernels, float64, the device RNG's noise) against
  the priorities the reference's train returned in its two calls."""
  import helpers
  from daydreamer_amd import agent as agent_mod, config as config_mod
  from oracle import ref_ops
  base, (plain, sp, shapes, params, data, B, T) = mrg.build(case)
  obs, act, _ = mrg.spaces_of(base)
  if base.startswith('named:'):
    cfg = helpers.make_config((base[6:],), imag_horizon=mg.CONFIG['imag_horizon'])
  else:
    c = dict(mg.CONFIG)
    cfg = helpers.make_config(c.pop('blocks'), **c)
  cfg = cfg.update({'hip.noise_seed': mg.NOISE_SEED, 'batch_size': B, 'replay_chunk': T})
  ag = agent_mod.Agent(obs, act, None, cfg, _ops=ref_ops.RefOps('cpu'), _device='cpu', _dtype=torch.float64)
  for g in ag.groups.values():   # (this package's deterministic initial values, as in the reference run)
    g.load(params)
  keys = np.arange(B * T, dtype=np.uint64).reshape(B, T)
  batch = {**{k: v for k, v in data.items() if k != 'reset'}, 'key': keys}
  state = None
  for step in (1, 2):
    outs, state, mets = ag.train(batch, state)
    assert np.array_equal(outs['key'], keys)
    _close(outs['priority'], GOLDS[case][f's{step}/priority'], 1e-6, (step, 'priority'))
    _close(float(mets['reward_loss_mean']), float(np.mean(GOLDS[case][f's{step}/priority'])), 1e-6, step)


def test_fixtures_cover_what_the_cases_claim():
  """decay: the clip bites, decayed kernels shrink;
  onehot: REINFORCE case has discrete action draws; gae / gve returns differ."""
  d, g = GOLDS['decay'], GOLDS['debug']
  assert float(d['s1/metric/model_grad_norm']) > 5.0            # model_opt.clip = 5 is active
  assert float(d['s1/metric/model_grad_norm']) == pytest.approx(float(g['s1/metric/model_grad_norm']), rel=1e-12)
  assert abs(float(d['s1/paramsum/rssm/img_in/kernel'][1]) - float(g['s1/paramsum/rssm/img_in/kernel'][1])) > 1e-9
  assert float(d['s1/metric/extr_imag_return_mean']) != float(g['s1/metric/extr_imag_return_mean']) or \
      float(d['s1/metric/extr_score_mean']) != float(g['s1/metric/extr_score_mean'])
  assert 's1/idx_act' in GOLDS['onehot'].files and 's1/idx_act' not in g.files
  assert int(g['s2/ctrl/slow_updates']) == 2 and float(g['s2/ctrl/advnorm'][2]) == 2.0
