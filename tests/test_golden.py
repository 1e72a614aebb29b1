"""Golden vectors (tests/golden/{debug,onehot,resnet}_step.npz, made by make_golden.py from the
float64 oracle: continuous actions / actor by backprop, one-hot actions / actor by REINFORCE,
residual encoder / decoder): the oracle must reproduce them, the learner's host logic must
match them on CPU, and the HIP path must match them on the GPU."""

import importlib.util
import pathlib

import numpy as np
import pytest
import torch

import helpers
from daydreamer_amd import learner as LM

HERE = pathlib.Path(__file__).parent
spec_ = importlib.util.spec_from_file_location('make_golden', HERE / 'golden' / 'make_golden.py')
mg = importlib.util.module_from_spec(spec_)
spec_.loader.exec_module(mg)
GOLDS = {case: np.load(HERE / 'golden' / f'{case}_step.npz') for case in mg.CASES}
CASES = list(mg.CASES)

KEYS = ('model_loss', 'image_loss_mean', 'vector_loss_mean', 'kl_loss_mean',
        'reward_loss_mean', 'cont_loss_mean', 'extr_critic_loss', 'actor_loss',
        'model_grad_norm', 'extr_critic_grad_norm', 'actor_grad_norm',
        'actent_mean', 'prior_ent_mean', 'post_ent_mean')


@pytest.mark.parametrize('case', CASES)
def test_oracle_reproduces_golden(case):
  GOLD = GOLDS[case]
  plain, sp, shapes, params, data, B, T = mg.build(case)
  H = plain['imag_horizon']
  ag = mg.make_ref(case, plain, sp, shapes, params)
  state = None
  for step in (1, 2):
    noise = mg.golden_noise(B, T, H, sp.groups, sp.act_dim, step)
    _, state, mets = ag.train(data, noise, state)
    for k in KEYS:
      g = float(GOLD[f's{step}/metric/{k}'])
      assert abs(float(mets[k]) - g) <= 1e-9 * max(1, abs(g)), (step, k)
    assert np.array_equal(ag.last['wm']['idxs']['post'].numpy(), GOLD[f's{step}/idx_post'])
    assert np.array_equal(ag.last['traj']['idx'].numpy(), GOLD[f's{step}/idx_img'])
    if case == 'onehot':
      assert np.array_equal(ag.last['traj']['action'].argmax(-1).numpy(), GOLD[f's{step}/idx_act'])


def check_learner(L, data, mtol, gtol, exact_idx, case='debug', gold=None, all_metrics=False):
  """gold: another fixture with the same layout (tests/test_reference_golden.py: the vectors made
  by running the reference's own sources); all_metrics: every metric the fixture and the learner
  both hold instead of KEYS."""
  GOLD = GOLDS[case] if gold is None else gold
  discrete = L.discrete
  B, T, N, H, D, F, G, C = L.B, L.T, L.N, L.H, L.D, L.F, L.G, L.C
  for step in (1, 2):
    L.upload(data)
    L.train_step_device(use_carry=(step > 1))
    mets = L.read_metrics()
    f = helpers.forced_from_learner(L)
    sites = [('obs_post', 'idx_post'), ('obs_prior', 'idx_prior'), ('img', 'idx_img')]
    if discrete:
      sites.append(('act', 'idx_act'))
    for nm, key in sites:
      same = (f[nm].numpy() == GOLD[f's{step}/{key}'])
      if exact_idx:
        assert same.all(), (step, nm)
      else:
        assert same.mean() == 1.0, f'step {step}: {nm} draws differ from golden ({same.mean():.4f} equal)'
    keys = KEYS
    if all_metrics:
      keys = [k[len(f's{step}/metric/'):] for k in GOLD.files if k.startswith(f's{step}/metric/')]
      missing = [k for k in keys if k not in mets and not k.endswith(('_grad_scale', '_grad_overflow'))]
      assert not missing, f'metrics the reference returns and the learner does not: {missing}'
      keys = [k for k in keys if k in mets]
      assert len(keys) >= 40, len(keys)
    for k in keys:
      g = float(GOLD[f's{step}/metric/{k}'])
      if np.isnan(g):
        assert np.isnan(float(mets[k])), (step, k)
        continue
      assert abs(float(mets[k]) - g) <= mtol * max(abs(g), 1e-2), (step, k, float(mets[k]), g)
    grads = L.export_grads()
    for name, g in grads.items():
      ref = GOLD[f's{step}/gradsum/{name}']
      assert abs(g.astype(np.float64).sum() - ref[0]) <= gtol * max(ref[1], 1e-12), (step, name)
    for k in mg.grad_keys('onehot' if discrete else 'debug'):
      if f's{step}/grad/{k}' in GOLD.files:
        assert helpers.rel_err(grads[k], GOLD[f's{step}/grad/{k}']) < gtol * 10, (step, k)


@pytest.mark.parametrize('case', CASES)
def test_learner_host_logic_matches_golden(case):
  from oracle import ref_ops
  plain, sp, shapes, params, data, B, T = mg.build(case)
  L = LM.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params,
                 noise_seed=mg.NOISE_SEED, dtype=torch.float64)
  check_learner(L, data, 1e-6, 1e-6, True, case)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_hip_matches_golden(hip, case):
  plain, sp, shapes, params, data, B, T = mg.build(case)
  L = LM.Learner(sp, hip, 'cuda:0', B, T, params=params, noise_seed=mg.NOISE_SEED)
  check_learner(L, data, 1e-3, 1e-3, False, case)
