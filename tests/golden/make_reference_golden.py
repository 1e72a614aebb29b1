"""Generates tests/golden/reference_{debug,onehot,resnet,decay}.npz by EXECUTING THE REFERENCE'S OWN
SOURCES - /root/reference/embodied/agents/dreamerv2plus/{agent,nets,tfutils,tfagent,behaviors}.py,
imported unmodified from where they lie - on the tf-on-torch stand-in of oracle/tf_on_torch.py
(TensorFlow / TFP / sonnet cannot be installed in this container; that module's header says exactly
what is the reference's code and what is substituted library).  Two consecutive `Agent.train`
calls of the reference agent on the tiny seeded problems of make_golden.py, in float64, with

  * this package's deterministic initial parameters assigned into the reference's variables (the
    reference initialises from unseeded np.random; the module tree is walked and every trainable
    variable must be matched by name and shape - so the fixture need not store ~3 M weights),
  * the categorical / normal draws taken from the device RNG's uniforms / normals
    (make_golden.golden_noise) by inverse CDF, in the order the reference's code asks for them,

and recorded: every metric the reference returns, the gradient the reference's tape hands to each
of its three optimizers (sum and |sum| per parameter, a few in full), the drawn classes, the
parameters after each step (sum / |sum| per parameter, a few in full) and the optimizer / controller
variables.  tests/test_reference_golden.py holds the float64 oracle (and through it the HIP path)
to these numbers.  Needs /root/reference: run here, commit the .npz.

  python tests/golden/make_reference_golden.py [case ...]
"""

import importlib.util
import io
import contextlib
import pathlib
import sys

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parents[1]
REF = pathlib.Path('/root/reference')
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

_spec = importlib.util.spec_from_file_location('make_golden', HERE / 'make_golden.py')
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)

from daydreamer_amd import synthetic  # noqa: E402
from oracle import tf_on_torch as tft  # noqa: E402

# case -> (make_golden case it shares problem and noise with, extra config overrides)
CASES = dict(
    debug=('debug', {}),
    onehot=('onehot', {}),
    resnet=('resnet', {}),
    # weight decay on kernels, gradient clipping that bites, GAE returns
    # (the debug block's `.*\.wd: 0.0` is a prefix match: it also turns wd_pattern into '0.0')
    decay=('debug', {'model_opt.wd': 1e-2, 'actor_opt.wd': 1e-2, 'critic_opt.wd': 1e-2,
                     'model_opt.wd_pattern': 'kernel', 'actor_opt.wd_pattern': 'kernel',
                     'critic_opt.wd_pattern': 'kernel', 'model_opt.clip': 5.0, 'critic_return': 'gae',
                     'actor_return': 'gae'}),
    # option branches: no slow critic, AutoAdapt 'prop' / 'fixed', score normalisation, other
    # balance / discount / lambda / loss scales, reward and cont heads behind a stop-gradient
    options=('debug', {'slow_target': False, 'wmkl.impl': 'prop', 'actent.impl': 'fixed',
                       'scorenorm.impl': 'std', 'retnorm.impl': 'std', 'wmkl_balance': 0.5,
                       'discount': 0.95, 'return_lambda': 0.8, 'loss_scales.kl': 0.5,
                       'loss_scales.reward': 2.0, 'grad_heads': ['decoder'], 'rssm.unimix': 0.05}),
    # one-hot actions: soft slow-critic updates every step, AutoAdapt the other way round,
    # unnormalised entropy, advantage normalisation off
    options_onehot=('onehot', {'slow_target_update': 1, 'slow_target_fraction': 0.5,
                               'wmkl.impl': 'fixed', 'actent.impl': 'prop', 'actent_norm': False,
                               'advnorm.impl': 'off', 'actor.unimix': 0.1}),
    # learning-rate warm-up (with decay): the decay sees the step count before the increment, Adam
    # the one after it
    warmup=('debug', {'model_opt.warmup': 4, 'actor_opt.warmup': 3, 'critic_opt.warmup': 1,
                      'model_opt.wd': 1e-2, 'model_opt.wd_pattern': 'kernel', 'actor_opt.wd': 1e-2,
                      'actor_opt.wd_pattern': 'kernel'}),
    # BASELINE.json workloads at their full network widths (batch 2 x 3, horizon 2): proprio only;
    # image + depth + five proprio keys with a one-hot 6-way action; two 128 x 128 cameras
    a1=('named:a1', {}),
    xarm=('named:xarm', {}),
    ur5_multicam=('named:ur5_multicam', {}))
FULL_GRADS = ('rssm/initial_deter', 'rssm/obs_stats/bias', 'reward/dist_out/out/kernel',
              'rssm/gru_out/norm/scale', 'critic/dist_out/out/kernel', 'actor/dist_out/out/kernel',
              'actor/dist_out/std/kernel')
ORACLE_ONLY = ()   # cases with options the product rejects at construction (none at present)
STEPS = (1, 2)      # train calls recorded per case (tools/fuzz_reference.py --steps N runs more)
FULL_MAX = 1024   # arrays stored in full up to this size (the carried state: strided beyond it)
FULL_PARAMS = ('rssm/img_in/norm/scale', 'critic/dist_out/out/kernel', 'critic_target/dist_out/out/kernel',
               'rssm/initial_deter', 'actor/dense0/norm/bias')


def _flat(d, prefix=''):
  out = {}
  for k, v in d.items():
    if isinstance(v, dict):
      out.update(_flat(v, f'{prefix}{k}.'))
    else:
      out[prefix + k] = v
  return out


def reference_modules():
  tft.install()
  sys.argv[0] = str(REF / 'embodied/agents/dreamerv2plus/train.py')   # (agent.py reads configs.yaml next to it)
  for p in (str(REF), str(REF / 'embodied/agents')):
    if p not in sys.path:
      sys.path.insert(0, p)
  import embodied
  import dreamerv2plus.agent as ref_agent
  return embodied, ref_agent


def reference_config(embodied, ref_agent, plain):
  """The reference's `defaults` with every key this package's config also has set to our value."""
  config = embodied.Config(ref_agent.Agent.configs['defaults'])
  ours = {k: v for k, v in _flat({k: v for k, v in plain.items() if k != 'hip'}).items()}
  known = config.flat
  missing = sorted(k for k in ours if k not in known)
  assert not missing, f'keys of our config the reference does not have: {missing}'
  config = config.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in ours.items()})
  return config.update({'tf.platform': 'cpu', 'tf.precision': 'float32', 'tf.jit': False,
                        'expl_behavior': 'None', 'task_behavior': 'Greedy'})


def variable_map(agent):
  """our parameter name -> the reference's tf.Variable, by walking its module tree."""
  out = {}
  def collect(module, prefix):
    for name, child in module._modules.items():
      if isinstance(child, tft.Variable):
        assert f'{prefix}/{name}' not in out
        out[f'{prefix}/{name}'] = child
      else:
        collect(child, f'{prefix}/{name}')
    norm = module.__dict__.get('_norm')
    if isinstance(norm, tft.SntModule):
      collect(norm, f'{prefix}/norm')
  wm = agent.wm
  for net, prefix in ((wm.encoder, 'enc'), (wm.heads['decoder'], 'dec')):
    for attr in ('_cnn', '_mlp'):
      if attr in net.__dict__:
        collect(net.__dict__[attr], f'{prefix}/{attr[1:]}')
  collect(wm.rssm, 'rssm')
  collect(wm.heads['reward'], 'reward')
  collect(wm.heads['cont'], 'cont')
  ac = agent.task_behavior.ac
  collect(ac.actor, 'actor')
  critic = ac.critics['extr']
  collect(critic.net, 'critic')
  if critic.target_net is not critic.net:     # (slow_target: False -> target_net IS net, agent.py:395-396)
    collect(critic.target_net, 'critic_target')
  return out


def feed_items(noise, T, H, discrete):
  items = []
  for t in range(T):
    items.append(('uniform', f'obs_prior/{t}', noise['u_obs_prior'][t]))
    items.append(('uniform', f'obs_post/{t}', noise['u_obs_post'][t]))
  act = (lambda t: ('uniform', f'act/{t}', noise['u_act'][t])) if discrete else \
        (lambda t: ('normal', f'act/{t}', noise['eps_act'][t]))
  items.append(act(0))
  for h in range(H):
    items.append(('uniform', f'img/{h}', noise['u_img'][h]))
    items.append(act(h + 1))
  return items


def spaces_of(base):
  """(observation spaces, action spaces, discrete?) of a make_golden case or a `named:` workload."""
  if base.startswith('named:'):
    obs, act = synthetic.config_spaces(base[6:])
    return obs, act, bool(getattr(act['action'], 'discrete', False))
  pover = mg.CASES[base][0]
  obs, act = synthetic.make_spaces(mg.PROBLEM['image'], mg.PROBLEM['vector'],
                                   pover.get('action', mg.PROBLEM['action']))
  return obs, act, bool(pover.get('discrete', False))


def build(case, batch=None, length=None, extra=None):
  base, over = CASES[case]
  import helpers
  if base.startswith('named:'):   # a BASELINE.json workload at its full network widths, tiny batch
    B, T = batch or mg.CONFIG['batch_size'], length or mg.CONFIG['replay_chunk']
    plain, sp, shapes, params, data = helpers.make_named_problem(
        base[6:], B, T, terminals=0.2, horizon=mg.CONFIG['imag_horizon'], **{**over, **(extra or {})})
    return base, (plain, sp, shapes, params, data, B, T)
  c = dict(mg.CONFIG)
  pover, cover = mg.CASES[base]
  cfg = helpers.make_config(c.pop('blocks'), **c)
  for o in (cover, over, extra):
    if o:
      cfg = cfg.update(o)
  return base, helpers.make_problem(cfg, batch=batch, length=length, **{**mg.PROBLEM, **pover})


def setup(case, problem, verbose=False):
  """The reference agent for `case` with every variable created, all state reset and this
  package's initial parameters assigned.  Returns (agent, variable map, batch, quiet context)."""
  base, (plain, sp, shapes, params, data, B, T) = problem
  H, G, A = plain['imag_horizon'], sp.groups, sp.act_dim
  obs, act, discrete = spaces_of(base)
  embodied, ref_agent = reference_modules()
  config = reference_config(embodied, ref_agent, plain)
  obs_space = {k: embodied.Space(v.dtype, v.shape) for k, v in obs.items()}
  act_space = {'action': embodied.Space(np.float32, (A,), -1.0 if not discrete else 0.0, 1.0)}
  act_space['action'].discrete = discrete
  batch = {k: v for k, v in data.items() if k in obs_space or k == 'action'}

  tft.VARIABLES.clear()
  np.random.seed(0)
  quiet = (lambda: contextlib.redirect_stdout(io.StringIO())) if not verbose else contextlib.nullcontext
  with quiet():
    agent = ref_agent.Agent(obs_space, act_space, embodied.Counter(), config)
    # one throw-away call creates every variable (the reference builds them lazily) ...
    tft.FEED.load(feed_items(mg.golden_noise(B, T, H, G, A, 99), T, H, discrete))
    agent.train(batch, None)
  assert not tft.FEED.items
  # ... then every variable goes back to its initial value (optimizer steps and moments,
  # AutoAdapt scales, Normalize moments, the slow critic's counter) and the trainable ones get
  # this package's deterministic initial parameters
  for v in tft.VARIABLES:
    v.reset()
  vmap = variable_map(agent.agent)
  trainable = {id(v) for v in tft.VARIABLES if v.trainable and v.dtype.is_floating_point
               and not v.name.split('/')[-1].startswith(('m_', 'v_'))}
  mapped = {id(v) for v in vmap.values()}
  stray = [v.name for v in tft.VARIABLES if id(v) in trainable - mapped
           and not any(s in v.name for s in ('AutoAdapt', 'Normalize', 'Optimizer'))]
  assert not stray, f'reference variables without a counterpart: {stray}'
  used = {k for k in params if plain['slow_target'] or not k.startswith('critic_target/')}
  assert set(vmap) == used, (sorted(set(vmap) ^ used))
  for name, var in vmap.items():
    assert tuple(var.shape) == tuple(params[name].shape), (name, tuple(var.shape), params[name].shape)
    var.assign(params[name])
  return agent, vmap, batch, quiet, discrete


def generate(case, verbose=True):
  problem = build(case)
  base, (plain, sp, shapes, params, data, B, T) = problem
  H, G, A = plain['imag_horizon'], sp.groups, sp.act_dim
  agent, vmap, batch, quiet, discrete = setup(case, problem, verbose)
  names = {var.name: name for name, var in vmap.items()}
  # replay keys ride along (agent.py:89-93: outs = {key, priority = criteria[config.priority]})
  batch = {**batch, 'key': np.arange(B * T, dtype=np.uint64).reshape(B, T)}
  out = {}
  state = None
  for step in STEPS:
    noise = mg.golden_noise(B, T, H, G, A, step)
    tft.FEED.load(feed_items(noise, T, H, discrete))
    tft.FEED.draws.clear()
    tft.GradientTape.LOG.clear()
    with quiet():
      outs, state, mets = agent.train(batch, state)
    assert np.array_equal(outs['key'], batch['key'])
    out[f's{step}/priority'] = np.array(outs['priority'], np.float64)
    # (a tf.GradientTape only records inside its `with` block: the state handed to the next call
    # is a constant there; torch autograd is always on, so cut it here)
    state = tft.nest_map(tft.stop_gradient, state)
    assert not tft.FEED.items and len(tft.GradientTape.LOG) == 3
    for k, v in mets.items():
      out[f's{step}/metric/{k}'] = np.float64(v)
    draws = dict(tft.FEED.draws)
    out[f's{step}/idx_prior'] = np.stack([draws[f'obs_prior/{t}'].reshape(B, G) for t in range(T)])
    out[f's{step}/idx_post'] = np.stack([draws[f'obs_post/{t}'].reshape(B, G) for t in range(T)])
    out[f's{step}/idx_img'] = np.stack([draws[f'img/{h}'].reshape(B * T, G) for h in range(H)])
    if discrete:
      out[f's{step}/idx_act'] = np.stack([draws[f'act/{t}'].reshape(B * T) for t in range(H + 1)])
    for log in tft.GradientTape.LOG:     # model, critic, actor - in the order the reference updates
      for vname, g in log.items():
        out[f's{step}/gradsum/{names[vname]}'] = np.array([g.sum(), np.abs(g).sum()])
        if names[vname] in FULL_GRADS and g.size <= FULL_MAX:
          out[f's{step}/grad/{names[vname]}'] = g
    for name, var in vmap.items():
      p = var.numpy()
      out[f's{step}/paramsum/{name}'] = np.array([p.sum(), np.abs(p).sum()])
      if name in FULL_PARAMS and p.size <= FULL_MAX:
        out[f's{step}/param/{name}'] = p.copy()
    for k, v in state.items():
      v = np.array(v.numpy() if hasattr(v, 'numpy') else v)
      out[f's{step}/state/{k}'] = v if v.size <= FULL_MAX else v.reshape(-1)[::max(1, v.size // FULL_MAX)]
    ac = agent.agent.task_behavior.ac
    # (np.array: .numpy() of a variable is a live view of its storage)
    out[f's{step}/ctrl/wmkl_scale'] = np.array(agent.agent.wm.wmkl.scale().numpy())
    out[f's{step}/ctrl/actent_scale'] = np.array(ac.actent.scale().numpy())
    for nm, norm in (('advnorm', ac.advnorm), ('retnorm', ac.retnorms['extr']),
                     ('scorenorm', ac.scorenorms['extr'])):
      out[f's{step}/ctrl/{nm}'] = np.array([norm._mean.numpy(), norm._sqrs.numpy(), norm._step.numpy()],
                                           np.float64)
    upd = getattr(ac.critics['extr'], 'updates', None)
    out[f's{step}/ctrl/slow_updates'] = np.array(-1 if upd is None else upd.numpy())
  path = HERE / f'reference_{case}.npz'
  np.savez_compressed(path, **out)
  print('wrote', path, path.stat().st_size, 'bytes;', len(out), 'arrays; model_loss',
        float(out['s1/metric/model_loss_mean']), float(out['s2/metric/model_loss_mean']))
  return out


# ------------------------------------------------------------------ Agent.policy / Agent.report

POLICY_MODES = ('train', 'explore', 'eval', 'train')
POLICY_EXTRA = {'expl_noise': 0.2, 'eval_noise': 0.1}


def policy_noise(n, G, A, discrete, call):
  rng = np.random.RandomState(1000 + call)
  noise = dict(u_prior=rng.rand(n, G), u_post=rng.rand(n, G))
  if discrete:
    noise.update(u_act=rng.rand(n), act_noise=rng.rand(n))
  else:
    noise.update(eps=rng.randn(n, A), act_noise=rng.randn(n, A))
  return noise


def generate_policy(case, verbose=False):
  """reference Agent.policy (agent.py:42-65) called four times with the carried state: sampled /
  explore / mode / sampled actions, then tfutils.action_noise with expl_noise / eval_noise."""
  problem = build(case, batch=3, length=len(POLICY_MODES), extra=POLICY_EXTRA)
  base, (plain, sp, shapes, params, data, B, T) = problem
  G, A = sp.groups, sp.act_dim
  agent, vmap, batch, quiet, discrete = setup(case, problem, verbose)
  out, state = {}, None
  for t, mode in enumerate(POLICY_MODES):
    obs = {k: v[:, t] for k, v in batch.items() if k != 'action'}
    noise = policy_noise(B, G, A, discrete, t)
    items = [('uniform', 'prior', noise['u_prior']), ('uniform', 'post', noise['u_post'])]
    kind = 'uniform' if discrete else 'normal'
    if mode != 'eval':
      items.append((kind, 'act', noise['u_act'] if discrete else noise['eps']))
    items.append((kind, 'act_noise', noise['act_noise']))
    tft.FEED.load(items)
    tft.FEED.draws.clear()
    with quiet():
      outs, state = agent.policy(obs, state, mode)
    assert not tft.FEED.items
    latent = state[0]
    out[f'c{t}/action'] = np.array(outs['action'])
    for k in ('deter', 'stoch', 'logit'):
      out[f'c{t}/latent/{k}'] = np.array(latent[k].numpy())
    out[f'c{t}/idx_post'] = dict(tft.FEED.draws)['post'].reshape(B, G)
  path = HERE / f'reference_policy_{case}.npz'
  np.savez_compressed(path, **out)
  print('wrote', path, path.stat().st_size, 'bytes;', len(out), 'arrays')


REPORT_SHAPE = dict(batch=3, length=7)
VIDEO_STRIDE = (1, 11, 13, 1)


def report_noise(B, T, H, G, A, n):
  rng = np.random.RandomState(2000)
  return dict(u_obs_prior=rng.rand(T, B, G), u_obs_post=rng.rand(T, B, G), u_openl=rng.rand(T - 5, n, G),
              u_img=rng.rand(H, n, G), eps_act=rng.randn(H + 1, n, A), u_act=rng.rand(H + 1, n))


def video_digest(v):
  """Per-frame sum and |sum| plus a strided sample of a [T, H, W, C] video grid."""
  v = np.asarray(v, np.float64)
  s = VIDEO_STRIDE
  return dict(shape=np.array(v.shape), sums=v.sum((1, 2, 3)), abssums=np.abs(v).sum((1, 2, 3)),
              sample=v[::s[0], ::s[1], ::s[2], ::s[3]].copy())


def generate_report(case, verbose=False):
  """reference Agent.report (agent.py:95-106): WorldModel.report (loss metrics without update,
  reconstruction + open-loop video) and Greedy.report (imagined rollout video).  The reference
  re-observes the first five steps twice with fresh samples; the same uniforms are fed again."""
  problem = build(case, **REPORT_SHAPE)
  base, (plain, sp, shapes, params, data, B, T) = problem
  H, G, A = plain['imag_horizon'], sp.groups, sp.act_dim
  agent, vmap, batch, quiet, discrete = setup(case, problem, verbose)
  n, ctx = min(6, B), 5
  noise = report_noise(B, T, H, G, A, n)
  observe = lambda steps, rows: [x for t in range(steps) for x in (
      ('uniform', f'obs_prior/{t}', noise['u_obs_prior'][t][:rows]),
      ('uniform', f'obs_post/{t}', noise['u_obs_post'][t][:rows]))]
  act = (lambda t: ('uniform', f'act/{t}', noise['u_act'][t])) if discrete else \
        (lambda t: ('normal', f'act/{t}', noise['eps_act'][t]))
  items = observe(T, B) + observe(ctx, n) + [('uniform', f'openl/{i}', noise['u_openl'][i]) for i in range(T - ctx)]
  items += observe(ctx, n) + [act(0)]
  for h in range(H):
    items += [('uniform', f'img/{h}', noise['u_img'][h]), act(h + 1)]
  tft.FEED.load(items)
  with quiet():
    rep = agent.report(batch)
  assert not tft.FEED.items
  out = {}
  for k, v in rep.items():
    v = np.asarray(v)
    if v.ndim == 4:
      for kk, vv in video_digest(v).items():
        out[f'video/{k}/{kk}'] = vv
    else:
      out[f'metric/{k}'] = np.float64(v)
  path = HERE / f'reference_report_{case}.npz'
  np.savez_compressed(path, **out)
  print('wrote', path, path.stat().st_size, 'bytes;', len(out), 'arrays;', sorted(k for k in out if k.startswith('video'))[:8])


if __name__ == '__main__':
  assert REF.exists(), 'needs the reference checkout at /root/reference'
  want = sys.argv[1:]
  for case_ in CASES:
    if not want or case_ in want:
      generate(case_, verbose=False)
  for case_ in ('debug', 'onehot'):
    if not want or f'policy_{case_}' in want:
      generate_policy(case_)
    if not want or f'report_{case_}' in want:
      generate_report(case_)
