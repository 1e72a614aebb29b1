"""Host-logic parity on CPU: the learner's explicit forward / hand-derived
backward (driven through the kernel restatements of oracle/ref_ops.py, in
float64) against the autograd oracle oracle/dreamer_ref.py on the same
minibatch, weights and noise: losses, every parameter gradient, every updated
parameter, controller state, two consecutive steps."""

import numpy as np
import pytest
import torch

from daydreamer_amd import learner as learner_mod
from oracle import dreamer_ref, ref_ops
import helpers


def run_pair(cfg, steps=2, side=False, **kw):
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, **kw)
  ops = ref_ops.RefOps('cpu')
  L = learner_mod.Learner(sp, ops, 'cpu', B, T, params=params, noise_seed=7,
                          dtype=torch.float64,
                          ops2=ref_ops.RefOps('cpu') if side else None)
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                            act_discrete=sp.act_discrete)
  state = None
  out = []
  for i in range(steps):
    p_before = L.export_params()
    L.upload(data)
    L.train_step_device(use_carry=(i > 0))
    mets = L.read_metrics()
    noise = helpers.noise_from_learner(L)
    forced = helpers.forced_from_learner(L)
    _, state, omets = ag.train(data, noise, state, forced)
    out.append((L, ag, mets, omets))
    grads = L.export_grads()
    for name, g in ag.last['grads'].items():
      err = helpers.rel_err(grads[name], g.numpy())
      assert err < 1e-6, f'step {i} grad {name}: rel err {err:.3e}'
    newp = L.export_params()
    for name, v in ag.export_params().items():
      err = helpers.rel_err(newp[name], v)
      assert err < 1e-7, f'step {i} param {name}: rel err {err:.3e}'
    for k in ('model_loss', 'image_loss_mean', 'vector_loss_mean', 'kl_loss_mean',
              'reward_loss_mean', 'cont_loss_mean', 'extr_critic_loss', 'actor_loss',
              'model_grad_norm', 'extr_critic_grad_norm', 'actor_grad_norm',
              'wmkl_scale_mean', 'actent_mean', 'actent_scale_mean',
              'extr_score_mean', 'extr_score_std', 'extr_score_mag', 'extr_score_max',
              'prior_ent_mean', 'post_ent_mean', 'prior_ent_min', 'post_ent_min',
              'extr_imag_reward_mean', 'extr_imag_return_std', 'kl_loss_std',
              'reward_loss_std', 'actent_std', 'model_loss_std', 'reward_pos_loss',
              'reward_neg_loss', 'reward_pos_acc', 'reward_neg_acc', 'reward_rate', 'reward_avg',
              'reward_pred', 'cont_pos_loss', 'cont_neg_loss', 'cont_pos_acc', 'cont_neg_acc',
              'cont_rate', 'cont_avg', 'cont_pred'):
      if k not in omets:
        continue
      a, o = float(mets[k]), float(omets[k])
      if np.isnan(o):
        assert np.isnan(a), f'step {i} metric {k}: {a} vs nan'
        continue
      assert abs(a - o) <= 2e-6 * max(1.0, abs(o)), f'step {i} metric {k}: {a} vs {o}'
  return out


def test_learner_matches_oracle_vision():
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=3,
                            replay_chunk=5, imag_horizon=4)
  run_pair(cfg, steps=2, image=64, vector=5, action=3, terminals=0.15)


def test_learner_matches_oracle_resnet():
  """`cnn: resnet` (reference nets.py:330-391): residual encoder and decoder, hand-derived
  backward against the oracle's autograd: every gradient, every updated parameter."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=2, replay_chunk=3, imag_horizon=2)
  cfg = cfg.update({'encoder.cnn': 'resnet', 'decoder.cnn': 'resnet', 'encoder.cnn_depth': 4,
                    'decoder.cnn_depth': 4, 'encoder.cnn_blocks': 2, 'decoder.cnn_blocks': 2})
  out = run_pair(cfg, steps=2, image=32, vector=5, action=3, terminals=0.15)
  L = out[0][0]
  names = {p.name for p in L.spec.params}
  # stage 1 of either net changes the channel count in its first block only: one 1x1 skip each
  assert 'enc/cnn/s1b0s/kernel' in names and 'enc/cnn/s1b1s/kernel' not in names
  assert 'enc/cnn/s0b0s/kernel' not in names and 'dec/cnn/s0b0s/kernel' not in names
  assert 'dec/cnn/s1b0s/kernel' in names and 'dec/cnn/in/bias' in names
  assert L.spec.embed == 1024 + 512


def test_learner_deferred_weight_grads():
  """Same check with the weight-gradient contractions queued for the side
  launch context (the GPU runs them on a second stream)."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=2,
                            replay_chunk=4, imag_horizon=2)
  run_pair(cfg, steps=1, side=True, image=64, vector=5, action=3, terminals=0.1)


def test_learner_matches_oracle_proprio():
  """BASELINE configs[0] shape family: a1 block, proprio only (vector 7, A 6)."""
  cfg = helpers.make_config(('a1', 'debug'), batch_size=4, replay_chunk=6,
                            imag_horizon=3)
  run_pair(cfg, steps=2, image=0, vector=7, action=6, terminals=0.1)


def test_learner_matches_oracle_discrete():
  """xarm / ur5 style: one-hot action space -> 'onehot' actor with REINFORCE."""
  cfg = helpers.make_config(('xarm', 'debug'), batch_size=3, replay_chunk=5,
                            imag_horizon=4)
  cfg = cfg.update({'encoder.mlp_keys': 'vector', 'decoder.mlp_keys': 'vector',
                    'encoder.cnn_keys': 'image', 'decoder.cnn_keys': 'image'})
  run_pair(cfg, steps=2, image=64, vector=5, action=6, terminals=0.15, discrete=True)


def test_learner_matches_oracle_image_and_depth():
  """xarm observation layout: image (3 ch) + depth (1 ch) concatenated on the
  channel axis in encoder and decoder, separate per-key losses."""
  import numpy as np
  from daydreamer_amd import config, spec, synthetic
  cfg = helpers.make_config(('xarm', 'debug'), batch_size=2, replay_chunk=4,
                            imag_horizon=2)
  cfg = cfg.update({'encoder.mlp_keys': 'vector', 'decoder.mlp_keys': 'vector'})
  plain = config.to_plain(cfg)
  obs, act = synthetic.make_spaces(64, 5, 6)
  obs = {'image': obs['image'], 'depth': synthetic.Space(np.uint8, (64, 64, 1)),
         **{k: v for k, v in obs.items() if k != 'image'}}
  shapes = {k: v.shape for k, v in obs.items()}
  sp = spec.build_spec(plain, shapes, 6, True)
  params = spec.init_params(sp, 0)
  data = synthetic.make_batch(obs, act, 2, 4, seed=3, smooth_images=False)
  data['action'] = np.eye(6, dtype=np.float32)[np.random.RandomState(0).randint(0, 6, (2, 4))]
  L = learner_mod.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', 2, 4, params=params,
                          dtype=torch.float64)
  ag = dreamer_ref.RefAgent(plain, shapes, 6, params, torch.float64, act_discrete=True)
  L.upload(data)
  L.train_step_device(use_carry=False)
  mets = L.read_metrics()
  _, _, omets = ag.train(data, helpers.noise_from_learner(L), None,
                         helpers.forced_from_learner(L))
  for k in ('image_loss_mean', 'depth_loss_mean', 'model_loss', 'actor_loss'):
    assert abs(float(mets[k]) - float(omets[k])) <= 2e-6 * max(1, abs(float(omets[k]))), k
  grads = L.export_grads()
  for name, g in ag.last['grads'].items():
    assert helpers.rel_err(grads[name], g.numpy()) < 1e-6, name


def test_weight_decay_and_is_first_midsequence():
  cfg = helpers.make_config(('a1', 'debug'), batch_size=3, replay_chunk=6,
                            imag_horizon=2)
  cfg = cfg.update({'model_opt.wd': 1e-2, 'actor_opt.wd': 1e-2, 'critic_opt.wd': 1e-2})
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, image=0, vector=7, action=6)
  data['is_first'][1, 3] = True
  data['is_first'][2, 1] = True
  ops = ref_ops.RefOps('cpu')
  L = learner_mod.Learner(sp, ops, 'cpu', B, T, params=params, dtype=torch.float64)
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64)
  L.upload(data)
  L.train_step_device(use_carry=False)
  ag.train(data, helpers.noise_from_learner(L), None, helpers.forced_from_learner(L))
  grads, newp = L.export_grads(), L.export_params()
  for name, g in ag.last['grads'].items():
    assert helpers.rel_err(grads[name], g.numpy()) < 1e-6, name
  for name, v in ag.export_params().items():
    assert helpers.rel_err(newp[name], v) < 1e-7, name


def test_loss_scale_option_keeps_the_mixed_optimizer_contract():
  """hip.loss_scale: true (the reference's float16 optimizer contract, tfutils.py:164-167,
  225-240, 246-260): `*_grad_scale` / `*_grad_overflow` metrics, the loss-scale controller state,
  and an update that is skipped - not an exception - when a gradient is not finite.  Without
  the option BOTH precisions raise, as the reference does for float32 and bfloat16
  (`self._mixed` is float16 only; check_numerics, tfutils.py:249).  Host logic on the CPU
  restatement of the kernels."""
  cfg = helpers.make_config(('a1', 'debug'), batch_size=3, replay_chunk=4, imag_horizon=2)
  cfg = cfg.update({'hip.precision': 'bfloat16', 'hip.loss_scale': True})
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, image=0, vector=7, action=6)
  L = learner_mod.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params)
  assert L.mixed
  L.upload(data)
  L.train_step_device(use_carry=False)
  mets = L.read_metrics()
  for pre in ('model', 'extr_critic', 'actor'):
    assert float(mets[f'{pre}_grad_scale']) == 1e4 and float(mets[f'{pre}_grad_overflow']) == 0.0
    assert float(mets[f'{pre}_grad_steps']) == 1.0
  assert float(L.groups['actor'].opt_state[4]) == 1.0          # good steps
  # an overflowing actor gradient: no exception, the step is skipped, the scale halves
  before = L.groups['actor'].flat.clone()
  keep = L.opt_step
  def poisoned(name, cfgkey):
    if name == 'actor':
      L.groups['actor'].gflat[5] = float('inf')
    keep(name, cfgkey)
  L.opt_step = poisoned
  L.train_step_device(use_carry=True)
  mets = L.read_metrics()
  assert float(mets['actor_grad_overflow']) == 1.0 and np.isnan(float(mets['actor_grad_norm']))
  assert float(mets['actor_grad_scale']) == 5e3 and float(mets['actor_grad_steps']) == 1.0
  assert float(mets['model_grad_overflow']) == 0.0 and float(mets['model_grad_steps']) == 2.0
  assert torch.equal(L.groups['actor'].flat, before)
  # without the option the same gradient raises in both precisions (check_numerics, tfutils.py:249)
  for prec in ('float32', 'bfloat16'):
    plain2 = dict(plain, hip=dict(plain.get('hip', {}), precision=prec, loss_scale=False))
    L2 = learner_mod.Learner(type(sp)(**{**sp.__dict__, 'cfg': plain2}), ref_ops.RefOps('cpu'), 'cpu', B, T, params=params)
    assert not L2.mixed
    L2.upload(data)
    keep2 = L2.opt_step
    def poisoned2(name, cfgkey, L2=L2, keep2=keep2):
      if name == 'actor':
        L2.groups['actor'].gflat[5] = float('inf')
      keep2(name, cfgkey)
    L2.opt_step = poisoned2
    L2.train_step_device(use_carry=False)
    with pytest.raises(FloatingPointError):
      L2.read_metrics()


def test_report_open_loop_grid():
  """Agent.report: loss metrics + the open-loop video grid layout of
  WorldModel.report (reference agent.py:276-281, tfutils.video_grid)."""
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=7, replay_chunk=8,
                            imag_horizon=3)
  obs, act = synthetic.make_spaces(64, 5, 3)
  ag = agent_mod.Agent(obs, act, None, cfg, _ops=ref_ops.RefOps('cpu'), _device='cpu')
  data = synthetic.make_batch(obs, act, 7, 8, seed=1, smooth_images=True)
  before = ag.save()
  rep = ag.report(data)
  after = ag.save()
  for k in before:  # report must not change any state
    assert np.array_equal(np.asarray(before[k]), np.asarray(after[k])), k
  v = rep['openl_image']
  assert v.shape == (8, 3 * 64, 6 * 64, 3) and v.dtype == np.float32
  truth = data['image'][:6].astype(np.float32) / 255.0          # [6,T,H,W,C]
  grid_truth = truth.transpose(1, 2, 0, 3, 4).reshape(8, 64, 6 * 64, 3)
  assert np.allclose(v[:, :64], grid_truth, atol=1e-6)
  model, error = v[:, 64:128], v[:, 128:]
  assert model.min() > 0 and model.max() < 1
  assert np.allclose(error, (model - grid_truth + 1) / 2, atol=1e-6)
  assert np.isfinite(rep['image_loss_mean'])


def test_learner_matches_oracle_multicam_128():
  """BASELINE configs[3] geometry: two 128x128 cameras concatenated on channels, five
  transposed-conv decoder layers (kernels 5,5,6,6,2 -> 128x128), per-camera image losses."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=2, replay_chunk=3,
                            imag_horizon=2)
  cfg = cfg.update({'decoder.cnn_kernels': [5, 5, 6, 6, 2]})
  run_pair(cfg, steps=1, image=128, cameras=2, vector=5, action=3, terminals=0.1)


def test_metric_snapshots_and_arena_alignment():
  """read_metrics(host=...) - the path the pipelined agent uses with per-phase snapshots -
  equals the live read-out; parameter arenas keep every tensor on a 16-byte boundary, the
  padding stays exactly zero through optimizer steps, and the behaviour phase's metric
  slots are tracked."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=2, replay_chunk=4, imag_horizon=3)
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, image=64, vector=5, action=3)
  L = learner_mod.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params, noise_seed=3,
                          dtype=torch.float64)
  for i in range(2):
    L.upload(data)
    L.train_step_device(use_carry=(i > 0))
  live = L.read_metrics()
  host = {k: v.cpu().numpy().copy() for k, v in L.metric_tensors().items()}
  snap = L.read_metrics(host)
  assert live.keys() == snap.keys()
  for k in live:
    assert np.array_equal(live[k], snap[k], equal_nan=True), k
  names = [L.stat_names[k] for k in sorted(L.stat_b_slots)]
  assert 'critic_loss' in names and 'kl_loss' not in names and 'image_loss' not in names
  for g in L.groups.values():
    used = torch.zeros(g.n, dtype=torch.bool)
    for p in g.specs:
      off = g.offset[p.name]
      assert off % 4 == 0, (p.name, off)
      assert not used[off:off + p.size].any()
      used[off:off + p.size] = True
    assert g.n % 4 == 0 and g.n_decay % 4 == 0
    pad = ~used
    assert float(g.flat[pad].abs().sum()) == 0.0
    if hasattr(g, 'm'):
      assert float(g.m[pad].abs().sum()) == 0.0 and float(g.gflat[pad].abs().sum()) == 0.0


@pytest.mark.parametrize('name', ['xarm', 'ur5_multicam'])
def test_named_workload_plumbing(name):
  """The BASELINE robot workloads as synthetic.config_spaces defines them (several image keys,
  five proprio keys concatenated for the encoder and decoded per key, one-hot actions) through
  the host logic at a tiny batch with shrunk networks; the full-width networks run on the GPU
  (tests/test_learner_gpu.py::test_full_size_*)."""
  plain, sp, shapes, params, data = helpers.make_named_problem(
      name, 2, 3, horizon=2, **{'rssm.deter': 32, 'rssm.units': 32, 'rssm.stoch': 4, 'rssm.classes': 4,
                                'encoder.cnn_depth': 4, 'decoder.cnn_depth': 4, '.*\\.layers': 1,
                                '.*\\.units': 16, 'encoder.mlp_layers': 1, 'decoder.mlp_layers': 1})
  L = learner_mod.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', 2, 3, params=params, dtype=torch.float64)
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                            act_discrete=sp.act_discrete)
  L.upload(data)
  L.train_step_device(use_carry=False)
  mets = L.read_metrics()
  _, _, omets = ag.train(data, helpers.noise_from_learner(L), None, helpers.forced_from_learner(L))
  keys = ['model_loss', 'actor_loss', 'extr_critic_loss'] + [
      f'{k}_loss_mean' for k in list(sp.dec_cnn_keys) + list(sp.dec_mlp_keys)]
  assert len(sp.dec_mlp_keys) == 5 and len(sp.dec_cnn_keys) == 2
  for k in keys:
    assert abs(float(mets[k]) - float(omets[k])) <= 2e-6 * max(1, abs(float(omets[k]))), k
  grads = L.export_grads()
  for n, g in ag.last['grads'].items():
    assert helpers.rel_err(grads[n], g.numpy()) < 1e-6, n


@pytest.mark.parametrize('over', [
    {'actor_return': 'gae', 'critic_return': 'gae'},
    {'slow_target': False},
    {'wmkl.impl': 'prop', 'actent.impl': 'fixed'},
    {'wmkl.impl': 'fixed', 'actent.impl': 'prop', 'scorenorm.impl': 'std'}])
def test_learner_option_variants(over):
  """Options of the reference beyond the defaults: VFunction.target 'gae' (agent.py:428-433),
  slow_target: False (target_net = net, agent.py:395-396), AutoAdapt 'prop' / 'fixed'
  (tfutils.py:427-432, 475-480): two steps against the autograd oracle."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=3, replay_chunk=4, imag_horizon=3)
  cfg = cfg.update(over)
  run_pair(cfg, steps=2, image=64, vector=5, action=3, terminals=0.1)


def test_chunked_heads_equal_bulk():
  """phase_imagine with a side context evaluates the reward / cont / target-critic heads per
  chunk of finished time rows (overlapping the rollout on the GPU): same results as one bulk
  evaluation after the rollout."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=3, replay_chunk=4, imag_horizon=6)
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, image=64, vector=5, action=3)
  outs = []
  for side in (None, ref_ops.RefOps('cpu')):
    L = learner_mod.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params, noise_seed=7,
                            dtype=torch.float64, ops_b2=side)
    for i in range(2):
      L.upload(data)
      L.train_step_device(use_carry=(i > 0))
    outs.append((L.export_params(), L.read_metrics()))
  for k, v in outs[0][0].items():
    assert np.array_equal(v, outs[1][0][k]), k
  for k, v in outs[0][1].items():
    assert np.array_equal(v, outs[1][1][k], equal_nan=True), k


def test_early_allreduce_range_of_the_model_arena():
  """Learner.early_range (data parallel, default schedule): the part of the model gradient arena
  that is all-reduced before the encoder's backward pass is exactly the contiguous run of decoder /
  reward / cont tensors - with weight decay (a1_vision: decayed kernels first) their kernels,
  71 % of the arena - and the two ranges around it cover everything else exactly once."""
  cfg = helpers.make_config(('a1_vision',), batch_size=2, replay_chunk=3, imag_horizon=2)
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, image=64, vector=16, action=16)
  L = learner_mod.Learner(sp, ref_ops.RefOps('cpu'), 'cpu', B, T, params=params)
  g = L.groups['model']
  lo, hi = L.early_range()
  assert lo % 4 == 0 and hi % 4 == 0 and 0 < lo < hi <= g.gflat.numel()
  inside = [p for p in g.specs if lo <= g.offset[p.name] < hi]
  assert inside and all(p.name.split('/')[0] in ('dec', 'reward', 'cont') for p in inside)
  assert all(g.offset[p.name] + p.size <= hi for p in inside)
  mods = {p.name.split('/')[0] for p in inside}
  assert mods == {'dec', 'reward', 'cont'}
  # every decayed tensor of those modules is inside (the kernels), the range is most of the arena
  for p in g.specs:
    if p.decay and p.name.split('/')[0] in mods:
      assert lo <= g.offset[p.name] < hi, p.name
  assert (hi - lo) / g.gflat.numel() > 0.6
