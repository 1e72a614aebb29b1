#!/usr/bin/env python
"""One-off / on-demand fuzz of the option space: random combinations of the learner's option
branches, each run (a) through the reference's own sources on the TensorFlow stand-in
(tests/golden/make_reference_golden.py machinery, needs /root/reference), (b) through the float64
oracle and (c) through the product's learner on the CPU restatement of the kernels; two train steps,
every metric / gradient sum / parameter sum compared.  Not part of the test suite (a combination
takes ~5 s); the fixed cases of tests/test_reference_golden.py are the regression net.

  python tools/fuzz_reference.py [--policy-report] [--steps N] [n_combinations] [seed]
  python tools/fuzz_reference.py --emit DIR n seed     (here: also keep the fixtures + cases.json)
  python tools/fuzz_reference.py --hip DIR             (on the MI355X: the HIP path against them)
"""
import contextlib
import importlib.util
import io
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
spec = importlib.util.spec_from_file_location('mrg', ROOT / 'tests/golden/make_reference_golden.py')
mrg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mrg)
mg, tft = mrg.mg, mrg.tft
from oracle import dreamer_ref, ref_ops  # noqa: E402
from daydreamer_amd import learner as LM  # noqa: E402


def draw(rng):
  pick = lambda *xs: xs[rng.randint(len(xs))]
  over = {
      'critic_return': pick('gve', 'gae'), 'actor_return': pick('gve', 'gae'),
      'slow_target': pick(True, True, False), 'slow_target_update': pick(1, 2, 100),
      'slow_target_fraction': pick(1.0, 0.5, 0.1),
      'wmkl.impl': pick('mult', 'prop', 'fixed'), 'actent.impl': pick('mult', 'prop', 'fixed'),
      'retnorm.impl': pick('off', 'std', 'mean_std'), 'scorenorm.impl': pick('off', 'std'),
      'advnorm.impl': pick('off', 'std', 'mean_std'), 'actent_norm': pick(True, False),
      'rssm.unimix': pick(0.0, 0.01, 0.1), 'actor.unimix': pick(0.0, 0.01, 0.1),
      'wmkl_balance': pick(0.8, 0.5, 1.0), 'discount': pick(0.998, 0.9), 'return_lambda': pick(0.95, 0.5, 1.0),
      'loss_scales.kl': pick(1.0, 0.3), 'loss_scales.cont': pick(1.0, 5.0),
      'grad_heads': pick(['decoder', 'reward', 'cont'], ['decoder'], ['reward', 'cont']),
      'model_opt.clip': pick(100.0, 3.0), 'model_opt.wd': pick(0.0, 1e-2), 'model_opt.wd_pattern': 'kernel',
      'actor_opt.eps': pick(1e-6, 1e-3), 'critic_opt.lr': pick(1e-4, 1e-2),
      'wmkl.target': pick(3.5, 1.0), 'actent.target': pick(0.5, 0.1), 'actor.minstd': pick(0.03, 0.1),
      'rssm.prior_layers': pick(3, 1),
      'batch_size': pick(2, 3, 4), 'replay_chunk': pick(3, 5), 'imag_horizon': pick(1, 2, 4),
      'rssm.stoch': pick(8, 4), 'rssm.classes': pick(8, 16),
      'model_opt.warmup': pick(0, 0, 3), 'actor_opt.warmup': pick(0, 2), 'critic_opt.warmup': pick(0, 10),
  }
  return pick('debug', 'onehot'), over


def run(base, over, idx, emit=None):
  name = f'fuzz{idx}'
  mrg.CASES[name] = (base, over)
  out = {}
  keep = mrg.HERE
  import tempfile
  with tempfile.TemporaryDirectory() as d:
    mrg.HERE = pathlib.Path(emit or d)
    with contextlib.redirect_stdout(io.StringIO()):
      gold = mrg.generate(name, verbose=False)
    mrg.HERE = keep
  problems = []
  b, (plain, sp, shapes, params, data, B, T) = mrg.build(name)
  discrete = mrg.spaces_of(b)[2]
  H, G, A = plain['imag_horizon'], sp.groups, sp.act_dim
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64, act_discrete=discrete,
                            ctrl_dtype=torch.float64)
  try:
    L = LM.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params, noise_seed=mg.NOISE_SEED,
                   dtype=torch.float64)
  except AssertionError as e:
    L = None
    problems.append(f'learner rejects: {e}')
  state = None
  for step in mrg.STEPS:
    noise = mg.golden_noise(B, T, H, G, A, step)
    forced = dict(obs_prior=gold[f's{step}/idx_prior'], obs_post=gold[f's{step}/idx_post'], img=gold[f's{step}/idx_img'])
    if discrete:
      forced['act'] = gold[f's{step}/idx_act']
    _, state, mets = ag.train(data, noise, state, forced=forced)
    lm = None
    if L is not None:
      L.upload(data)
      L.train_step_device(use_carry=(step > 1))
      lm = L.read_metrics()
    for k in [k for k in gold if k.startswith(f's{step}/metric/')]:
      m = k.split('/metric/')[1]
      if m.endswith(('_grad_scale', '_grad_overflow')):
        continue
      r = float(gold[k])
      # (the learner reports float32 metric slabs also in float64 mode: 1e-6 relative, 1e-7 absolute)
      for who, got, tol in (('oracle', mets, 1e-9), ('learner', lm, 1e-6)):
        if got is None:
          continue
        if m not in got:
          problems.append(f'{who} misses {m}')
          continue
        v = float(got[m])
        if np.isnan(r) != np.isnan(v) or (not np.isnan(r) and abs(v - r) > tol * max(abs(r), 0.1 if who == 'learner' else 1.0)):
          problems.append(f's{step} {who} {m}: {v} vs {r}')
    now = ag.export_params()
    lp = L.export_params() if L is not None else None
    for k in [k for k in gold if k.startswith(f's{step}/paramsum/')]:
      n = k.split('/paramsum/')[1]
      ref = gold[k]
      for who, got, tol in (('oracle', now, 1e-10), ('learner', lp, 1e-8)):
        if got is None:
          continue
        p = np.asarray(got[n], np.float64)
        if abs(p.sum() - ref[0]) > tol * max(ref[1], 1e-30) or abs(np.abs(p).sum() - ref[1]) > tol * max(ref[1], 1e-30):
          problems.append(f's{step} {who} param {n}')
  return problems


def run_policy_report(base, over, idx):
  """Agent.policy x4 and Agent.report of the reference's sources against the oracle (1e-9)."""
  import tempfile
  name = f'fuzzpr{idx}'
  over = {k: v for k, v in over.items() if k not in ('batch_size', 'replay_chunk')}
  mrg.CASES[name] = (base, over)
  problems = []
  def close(a, b, what, tol=1e-9):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape or np.abs(a - b).max() > tol * max(1.0, np.abs(b).max()):
      problems.append(what)
  keep = mrg.HERE
  with tempfile.TemporaryDirectory() as d:
    mrg.HERE = pathlib.Path(d)
    try:
      with contextlib.redirect_stdout(io.StringIO()):
        mrg.generate_policy(name)
        mrg.generate_report(name)
      pol = dict(np.load(pathlib.Path(d) / f'reference_policy_{name}.npz'))
      rep_gold = dict(np.load(pathlib.Path(d) / f'reference_report_{name}.npz'))
    finally:
      mrg.HERE = keep
  discrete = mrg.spaces_of(base)[2]
  mk = lambda plain, shapes, sp, params: dreamer_ref.RefAgent(
      plain, shapes, sp.act_dim, params, torch.float64, act_discrete=discrete, ctrl_dtype=torch.float64)
  b, (plain, sp, shapes, params, data, B, T) = mrg.build(name, batch=3, length=len(mrg.POLICY_MODES), extra=mrg.POLICY_EXTRA)
  ag, state = mk(plain, shapes, sp, params), None
  for t, mode in enumerate(mrg.POLICY_MODES):
    obs = {k: v[:, t] for k, v in data.items() if k not in ('action', 'reset')}
    outs, state = ag.policy(obs, state, mrg.policy_noise(B, sp.groups, sp.act_dim, discrete, t), mode)
    close(outs['action'].numpy(), pol[f'c{t}/action'], f'policy call {t} ({mode}) action')
    for k in ('deter', 'stoch', 'logit'):
      close(state[0][k].numpy(), pol[f'c{t}/latent/{k}'], f'policy call {t} latent {k}')
  b, (plain, sp, shapes, params, data, B, T) = mrg.build(name, **mrg.REPORT_SHAPE)
  ag = mk(plain, shapes, sp, params)
  noise = mrg.report_noise(B, T, plain['imag_horizon'], sp.groups, sp.act_dim, min(6, B))
  rep = ag.report({k: v for k, v in data.items() if k != 'reset'}, noise)
  for k, v in rep_gold.items():
    if k.startswith('metric/'):
      m = k[len('metric/'):]
      if m not in rep:
        problems.append(f'report misses {m}')
      else:
        close(float(rep[m]), float(v), f'report metric {m}')
  for vid in sorted({k.split('/')[1] for k in rep_gold if k.startswith('video/')}):
    got = mrg.video_digest(rep[vid].numpy())
    for kk in ('sums', 'abssums', 'sample'):
      close(got[kk], rep_gold[f'video/{vid}/{kk}'], f'report video {vid} {kk}')
  return problems


def hip(directory):
  """The HIP path against emitted fixtures (no reference checkout needed)."""
  import json
  from daydreamer_amd import hipops
  from test_golden import check_learner
  ops = hipops.HipOps('cuda:0')
  cases = json.load(open(pathlib.Path(directory) / 'cases.json'))
  bad = 0
  for name, base, over in cases:
    mrg.CASES[name] = (base, over)
    b, (plain, sp, shapes, params, data, B, T) = mrg.build(name)
    gold = np.load(pathlib.Path(directory) / f'reference_{name}.npz')
    try:
      L = LM.Learner(sp, ops, 'cuda:0', B, T, params=params, noise_seed=mg.NOISE_SEED)
      check_learner(L, data, 1e-3, 1e-3, False, b, gold=gold, all_metrics=True)
      print(f'[{name}] {base} ok', flush=True)
    except AssertionError as e:
      bad += 1
      print(f'[{name}] {base} FAILED {str(e)[:300]}\n    options: {over}', flush=True)
  print(f'{len(cases) - bad} of {len(cases)} combinations: HIP path == reference sources (metrics 1e-3, gradients 1e-3)')


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == '--hip':
    hip(sys.argv[2])
    sys.exit(0)
  emit = None
  policy_report = '--policy-report' in sys.argv    # also Agent.policy x4 and Agent.report (oracle)
  if policy_report:
    sys.argv.remove('--policy-report')
  if len(sys.argv) > 2 and sys.argv[1] == '--steps':    # more train calls per combination
    mrg.STEPS = tuple(range(1, int(sys.argv[2]) + 1))
    del sys.argv[1:3]
  if len(sys.argv) > 1 and sys.argv[1] == '--emit':
    emit = sys.argv[2]
    pathlib.Path(emit).mkdir(parents=True, exist_ok=True)
    del sys.argv[1:3]
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
  rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
  bad = 0
  cases = []
  for i in range(n):
    base, over = draw(rng)
    cases.append((f'fuzz{i}', base, over))
    try:
      problems = run(base, over, i, emit)
      if policy_report:
        problems += run_policy_report(base, over, i)
    except Exception as e:  # noqa: BLE001
      problems = [f'EXCEPTION {type(e).__name__}: {str(e)[:300]}']
    status = 'ok' if not problems else f'{len(problems)} PROBLEMS'
    print(f'[{i}] {base} {status}', flush=True)
    if problems:
      bad += 1
      print('    options:', over)
      for p in problems[:12]:
        print('    ', p)
  print(f'{n - bad} of {n} combinations agree (reference sources == oracle == learner host logic)')
  if emit:
    import json
    json.dump(cases, open(pathlib.Path(emit) / 'cases.json', 'w'))
