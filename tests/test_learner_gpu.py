"""End-to-end parity on the MI355X: one and two `Learner.train_step_device`
calls through the HIP kernels (float32) against the float64 autograd oracle on
the same minibatch, weights and noise.  Tolerance: losses / grad norms 1e-3
relative (north_star), gradients 2e-3 of their max (fp32 vs fp64 through
deep LayerNorm stacks)."""

import numpy as np
import pytest
import torch

from daydreamer_amd import learner as learner_mod
import helpers

pytestmark = pytest.mark.gpu

LOSS_KEYS = ('model_loss', 'image_loss_mean', 'vector_loss_mean', 'kl_loss_mean',
             'reward_loss_mean', 'cont_loss_mean', 'extr_critic_loss', 'actor_loss',
             'model_grad_norm', 'extr_critic_grad_norm', 'actor_grad_norm',
             'actent_mean', 'extr_score_std', 'prior_ent_mean', 'post_ent_mean',
             'extr_imag_reward_mean', 'extr_imag_return_mean')


SAMPLE_TOL = 1e-6       # a forced draw may differ from the oracle's only this close to a CDF edge
ADOPTED_FRAC = 1e-5     # and at most this fraction of all draws may do so (measured: 3 of 1.36 M at
                        # configs[1] full size, edge gaps <= 2.4e-8; profiles/r02_pytest_gpu.log)


def check_adopted(what):
  """The discrete latent draws are bit-exact given identical fp32 statistics (test_hip_ops:
  device == host twin == restatement).  Against the float64 oracle the statistics differ in the
  last bits, so a draw whose uniform lies within float noise of a CDF edge can legitimately
  land on the other side: those are adopted from the device, counted, printed and bounded."""
  from oracle import dreamer_ref
  st = dreamer_ref.SAMPLE_STATS
  frac = st['adopted'] / max(st['draws'], 1)
  print(f'{what}: {st["adopted"]} of {st["draws"]} draws adopted from the device '
        f'({frac:.2e}), largest CDF-edge gap {st["max_gap"]:.2e}')
  assert frac <= ADOPTED_FRAC or st['adopted'] <= 2, (what, st)


def reset_adopted():
  from oracle import dreamer_ref
  dreamer_ref.SAMPLE_TOL[0] = SAMPLE_TOL
  dreamer_ref.SAMPLE_STATS.update(draws=0, adopted=0, max_gap=0.0)


def run(hip, cfg, steps, gtol=2e-3, kw_side=None, **kw):
  from oracle import dreamer_ref
  reset_adopted()
  plain, sp, shapes, params, data, B, T = helpers.make_problem(cfg, **kw)
  L = learner_mod.Learner(sp, hip, 'cuda:0', B, T, params=params, noise_seed=3,
                          ops2=kw_side)
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                            act_discrete=sp.act_discrete)
  state = None
  for i in range(steps):
    L.upload(data)
    L.train_step_device(use_carry=(i > 0))
    torch.cuda.synchronize()
    mets = L.read_metrics()
    _, state, omets = ag.train(data, helpers.noise_from_learner(L), state,
                               helpers.forced_from_learner(L))
    for k in LOSS_KEYS:
      if k in omets:
        a, o = float(mets[k]), float(omets[k])
        assert abs(a - o) <= 1e-3 * max(abs(o), 1e-2), f'step {i} {k}: {a} vs {o}'
    grads = L.export_grads()
    worst = max((helpers.rel_err(grads[n], g.numpy()), n)
                for n, g in ag.last['grads'].items())
    assert worst[0] < gtol, f'step {i} worst grad {worst}'
    newp = L.export_params()
    worstp = max((helpers.rel_err(newp[n], v), n) for n, v in ag.export_params().items())
    # Adam's first steps are sign-like: a 1e-6 gradient difference can flip lr-sized updates
    assert worstp[0] < 5e-3, f'step {i} worst param {worstp}'
  check_adopted(f'B{B} T{T}')
  return L


def test_e2e_debug_vision(hip):
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6,
                            imag_horizon=4)
  run(hip, cfg, 2, image=64, vector=5, action=3, terminals=0.1)


def test_e2e_side_stream(hip):
  """Weight-gradient contractions on the side HIP stream (second launch context
  with its own workspace), overlapping the reverse scan: same parity bar."""
  from daydreamer_amd import hipops
  side = hipops.HipOps('cuda:0', ws_bytes=256 << 20)
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6,
                            imag_horizon=4)
  run(hip, cfg, 2, kw_side=side, image=64, vector=5, action=3, terminals=0.1)


def test_e2e_split_forward(hip):
  """hip.split_fwd (opt-in): the world-model forward as two batch halves - the fused observe scan
  of one half next to the encoder / decoder of the other on the side stream, per-half noise
  tensors keyed by global row - against the float64 oracle at the same parity bar (odd batch:
  halves of 3 and 2 rows), image + vector keys."""
  from daydreamer_amd import hipops
  side = hipops.HipOps('cuda:0', ws_bytes=256 << 20)
  cfg = helpers.make_config(('a1_vision',), batch_size=5, replay_chunk=6, imag_horizon=3)
  cfg = cfg.update({'hip.split_fwd': True})
  L = run(hip, cfg, 2, kw_side=side, image=64, vector=16, action=16, terminals=0.1)
  assert L.split_fwd and L.halves == [(0, 3), (3, 5)]


def test_e2e_a1_proprio(hip):
  """BASELINE configs[0]: a1 block, proprio only, batch 16 x seq 16, horizon 5."""
  cfg = helpers.make_config(('a1',), batch_size=16, replay_chunk=16, imag_horizon=5)
  run(hip, cfg, 1, image=0, vector=7, action=6, terminals=0.05)


def test_e2e_vision_small_units(hip):
  """a1_vision geometry (64x64 image, 32x32 latent, A=16) with a short batch."""
  cfg = helpers.make_config(('a1_vision',), batch_size=4, replay_chunk=5, imag_horizon=3)
  run(hip, cfg, 1, image=64, vector=16, action=16, terminals=0.0)


def test_e2e_xarm_discrete(hip):
  """BASELINE configs[2] family: xarm block (deter 512), 64x64 image + proprio,
  one-hot 6-way action -> 'onehot' actor trained by REINFORCE."""
  cfg = helpers.make_config(('xarm',), batch_size=4, replay_chunk=5, imag_horizon=3)
  cfg = cfg.update({'encoder.mlp_keys': 'vector', 'decoder.mlp_keys': 'vector',
                    'encoder.cnn_keys': 'image', 'decoder.cnn_keys': 'image'})
  run(hip, cfg, 2, image=64, vector=20, action=6, terminals=0.05, discrete=True)


def test_e2e_scaled_latent(hip):
  """BASELINE configs[4] network family (a1_scaled: deter 4096, stoch 64x64, horizon
  20) at a tiny batch: exercises the generic (C > 1024) LayerNorm / GRU paths and the
  64-class latent kernels against the oracle."""
  cfg = helpers.make_config(('a1_scaled',), batch_size=2, replay_chunk=3, imag_horizon=2)
  cfg = cfg.update({'rssm.deter': 1536})  # keeps the fp64 oracle quick; still > 1024 LN
  run(hip, cfg, 1, image=64, vector=16, action=16, terminals=0.0)


def test_e2e_multicam_128(hip):
  """BASELINE configs[3] geometry (ur5 multi-camera): two 128x128 cameras on channels,
  decoder kernels 5,5,6,6,2 -> 128x128 (k = 2 transposed conv, 6-channel uint8 input)."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=3, replay_chunk=4, imag_horizon=3)
  cfg = cfg.update({'decoder.cnn_kernels': [5, 5, 6, 6, 2]})
  run(hip, cfg, 1, image=128, cameras=2, vector=5, action=3, terminals=0.1)


def test_e2e_resnet(hip):
  """`encoder.cnn: resnet` / `decoder.cnn: resnet` (reference nets.py:330-391): stride-1 SAME
  3x3 convolutions with pre-activation, residual blocks, 2x2 pooling / 2x repetition, on a
  64x64 image (four stages each way) against the oracle: losses, every gradient, every
  updated parameter, two steps."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=3, replay_chunk=4, imag_horizon=3)
  cfg = cfg.update({'encoder.cnn': 'resnet', 'decoder.cnn': 'resnet', 'encoder.cnn_depth': 8,
                    'decoder.cnn_depth': 8})
  L = run(hip, cfg, 2, image=64, vector=5, action=3, terminals=0.1)
  assert len(L.spec.enc_res.stages) == 4 and L.spec.dec_res.feat_c == 128


def test_e2e_resnet_multicam_128(hip):
  """BASELINE configs[3] geometry through the residual nets - the reference's own way to a
  128x128 image (five stages each way), two cameras on the channel axis."""
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=2, replay_chunk=3, imag_horizon=2)
  cfg = cfg.update({'encoder.cnn': 'resnet', 'decoder.cnn': 'resnet', 'encoder.cnn_depth': 4,
                    'decoder.cnn_depth': 4, 'encoder.cnn_blocks': 1, 'decoder.cnn_blocks': 1})
  L = run(hip, cfg, 1, image=128, cameras=2, vector=5, action=3, terminals=0.1)
  assert len(L.spec.enc_res.stages) == 5 and L.spec.image_c == 6


def test_fused_observe_scan_equals_launch_sequence(hip):
  """csrc/scan.hip: RSSM.observe forward as ONE persistent launch (grid barriers between the
  four layers of a step) against the per-layer launch sequence on the same inputs, at the full
  configs[1] size and on a ragged batch (B = 21: partly filled row block): every buffer the
  backward pass reads must agree to float reassociation, the drawn latents exactly - except in
  sequences where a draw sat on a CDF edge and flipped (the two paths sum the small contractions
  in different orders); those are counted and must be rare.  The barrier's error word stays 0."""
  for (name, B, T, first_mid) in (('a1_vision', 50, 50, False), ('a1_vision', 21, 7, True), ('xarm', 25, 50, False),
                                  ('a1_vision', 64, 3, False), ('a1_vision', 1, 2, False)):
    cfg = helpers.make_config((name,), batch_size=B, replay_chunk=T)
    if name == 'xarm':   # deter = units = 512, one-hot 6-way actions: the streamed-plane variant of the kernel
      cfg = cfg.update({'encoder.mlp_keys': 'vector', 'decoder.mlp_keys': 'vector',
                        'encoder.cnn_keys': 'image', 'decoder.cnn_keys': 'image'})
      plain, sp, shapes, params, data, _, _ = helpers.make_problem(
          cfg, image=64, vector=20, action=6, terminals=0.02, smooth=True, discrete=True)
    else:
      plain, sp, shapes, params, data, _, _ = helpers.make_problem(
          cfg, image=64, vector=16, action=16, terminals=0.02, smooth=True)
    if first_mid:
      data['is_first'][3, 4] = True
      data['is_first'][20, 2] = True
    Ls = []
    for fused in (True, False):
      plain2 = dict(plain, hip=dict(plain.get('hip', {}), fused_scan=fused))
      sp2 = type(sp)(**{**sp.__dict__, 'cfg': plain2})
      L = learner_mod.Learner(sp2, hip, 'cuda:0', B, T, params=params, noise_seed=5)
      assert L.fused_scan == fused
      L.upload(data)
      L.reset_carry()
      L.phase_prep()
      L.encoder_fwd()
      L.initial_fwd()
      L.observe_fwd(True)
      torch.cuda.synchronize()
      Ls.append(L)
    A, Bq = Ls
    assert int(A.scan_sync[1]) == 0, 'a grid-barrier spin timed out'
    G, C, D = A.G, A.C, A.D
    ia = A.b['post'][:, D:].view(B, T, G, C).argmax(-1)
    ib = Bq.b['post'][:, D:].view(B, T, G, C).argmax(-1)
    assert torch.equal(A.b['post'][:, D:].sum(-1), torch.full((B * T,), float(G), device='cuda'))
    same = (ia == ib).all(-1).all(-1)          # sequences with identical draws throughout
    print(f'fused scan {name} B{B} T{T}: {int((~same).sum())} of {B} sequences contain a flipped draw')
    assert int((~same).sum()) <= max(1, B // 25)
    rows = same.repeat_interleave(T)
    def cmp(x, y, what, tol=2e-5):
      x, y = x[rows].double(), y[rows].double()
      err = float((x - y).abs().max() / (y.abs().max() + 1e-30))
      assert err < tol, (what, err)
    cmp(A.b['post'], Bq.b['post'], 'post')
    cmp(A.b['post_logit'], Bq.b['post_logit'], 'post_logit')
    cmp(A.b['xin'], Bq.b['xin'], 'xin')
    cmp(A.b['gin'], Bq.b['gin'], 'gin')
    cmp(A.b['z3'], Bq.b['z3'], 'z3')
    cmp(A.b['gstats'], Bq.b['gstats'], 'gstats')
    cmp(A.a_img_in.z, Bq.a_img_in.z, 'z1')
    cmp(A.a_img_in.stats, Bq.a_img_in.stats, 'st1')
    cmp(A.a_obs_out.z, Bq.a_obs_out.z, 'zo')
    cmp(A.a_obs_out.out, Bq.a_obs_out.out, 'xo')
    cmp(A.a_obs_out.stats, Bq.a_obs_out.stats, 'st3')
    cmp(A.a_obs_stats.z, Bq.a_obs_stats.z, 'xq')
    cmp(A.b['prior_logit'], Bq.b['prior_logit'], 'prior_logit')
    del Ls, A, Bq
    torch.cuda.empty_cache()


def _record_flips(what, B, T, H, adim, flipped, N):
  """Rows whose discrete draws differ between the fused kernel and the launch sequence, kept next
  to the test log (gpurun_out/ travels back from the GPU box; copied into profiles/)."""
  import os, pathlib
  out = pathlib.Path(os.environ.get('GRAFT_REPO_ROOT', pathlib.Path(__file__).resolve().parents[1])) / 'gpurun_out'
  try:
    out.mkdir(exist_ok=True)
    with open(out / 'fused_rollout_flipped_rows.txt', 'a') as f:
      f.write(f'{what} B{B} T{T} H{H} A{adim}: {flipped} of {N} rows contain a flipped draw\n')
  except OSError:
    pass


@pytest.mark.parametrize('rows', [16, 32])
def test_fused_imagination_rollout_equals_launch_sequence(hip, rows):
  """(rows: of the imagination batch per workgroup - csrc/imag.hip 16, csrc/imag32.hip 32.)
  csrc/imag.hip: WorldModel.imagine (H img_steps + H + 1 policy evaluations) as ONE persistent
  launch against the per-layer launch sequence, inside a whole train step on the same minibatch
  and weights, at the full configs[1] size, on a ragged row count (N = 21 * 7 = 147: a partly
  filled 16-row block) and for the 6-dim action space.  Rows whose latent draws all agree must
  agree in every buffer the backward pass reads to float reassociation; a draw that sat on a CDF
  edge flips the rest of that row's trajectory (the one-hot inputs are gathered instead of
  multiplied, so the small contractions sum in a different order) - such rows are counted and
  must be rare.  The losses of the step agree to the tolerance of the oracle parity tests."""
  for (name, B, T, H, adim) in (('a1_vision', 50, 50, 15, 16), ('a1_vision', 21, 7, 5, 16),
                                ('a1_vision', 2, 3, 2, 6)):
    cfg = helpers.make_config((name,), batch_size=B, replay_chunk=T, imag_horizon=H)
    plain, sp, shapes, params, data, _, _ = helpers.make_problem(
        cfg, image=64, vector=16, action=adim, terminals=0.02, smooth=True)
    Ls, mets = [], []
    for fused in (True, False):
      plain2 = dict(plain, hip=dict(plain.get('hip', {}), fused_imag=fused, imag_rows=rows))
      sp2 = type(sp)(**{**sp.__dict__, 'cfg': plain2})
      L = learner_mod.Learner(sp2, hip, 'cuda:0', B, T, params=params, noise_seed=7)
      assert L.fused_imag == fused
      L.upload(data)
      L.train_step_device(use_carry=False)
      torch.cuda.synchronize()
      mets.append(L.read_metrics())
      Ls.append(L)
    A, Bq = Ls
    N, G, C, D, F = A.N, A.G, A.C, A.D, A.F
    sa = A.b['traj'][:, :, D:F].reshape(H + 1, N, G, C)
    sb = Bq.b['traj'][:, :, D:F].reshape(H + 1, N, G, C)
    assert torch.equal(sa.sum(-1), torch.ones_like(sa.sum(-1)))          # one-hot everywhere
    same = (sa.argmax(-1) == sb.argmax(-1)).all(-1).all(0)                # rows with identical draws throughout
    flipped = int((~same).sum())
    print(f'fused imagination {name} B{B} T{T} H{H} A{adim}: {flipped} of {N} rows contain a flipped draw')
    _record_flips(f'fused imagination (continuous, {rows} rows per workgroup)', B, T, H, adim, flipped, N)
    assert flipped <= max(1, N // 500)
    def cmp(x, y, what, rows_per_t, tol=5e-5):
      x = x.reshape(rows_per_t, N, -1)[:, same].double()
      y = y.reshape(rows_per_t, N, -1)[:, same].double()
      err = float((x - y).abs().max() / (y.abs().max() + 1e-30))
      assert err < tol, (what, err)
    W = F + A.A
    cmp(A.b['traj'][..., :W], Bq.b['traj'][..., :W], 'traj', H + 1)
    la, lb = A.acts_im['actor'], Bq.acts_im['actor']
    for i in range(len(la[0])):
      cmp(la[0][i].z, lb[0][i].z, f'actor{i}.z', H + 1)
      cmp(la[0][i].stats, lb[0][i].stats, f'actor{i}.stats', H + 1)
      cmp(la[0][i].out, lb[0][i].out, f'actor{i}.out', H + 1)
    for i in range(2):
      cmp(la[1][i].z, lb[1][i].z, f'actor head {i}', H + 1)
    cmp(A.ai_img_in.z, Bq.ai_img_in.z, 'img_in.z', H)
    cmp(A.ai_img_in.stats, Bq.ai_img_in.stats, 'img_in.stats', H)
    cmp(A.ai_img_in.out, Bq.ai_img_in.out, 'img_in.out', H)
    cmp(A.b['iz3'], Bq.b['iz3'], 'z3', H)
    cmp(A.b['igstats'], Bq.b['igstats'], 'gstats', H)
    for i in range(A.n_prior):
      cmp(A.ai_img_out[i].z, Bq.ai_img_out[i].z, f'img_out{i}.z', H)
      cmp(A.ai_img_out[i].stats, Bq.ai_img_out[i].stats, f'img_out{i}.stats', H)
      cmp(A.ai_img_out[i].out, Bq.ai_img_out[i].out, f'img_out{i}.out', H)
    cmp(A.ai_img_stats.z, Bq.ai_img_stats.z, 'img_stats', H)
    # (a flipped row's trajectory differs from its flipped draw on: the batch means move by at most
    # that row's share - the tolerance grows by 4 x the flipped share, 1e-3 with no flip)
    mtol = 1e-3 + 4.0 * flipped / N
    for k in ('model_loss', 'extr_critic_loss', 'actor_loss', 'actor_grad_norm', 'extr_critic_grad_norm'):
      a_, b_ = float(mets[0][k]), float(mets[1][k])
      assert abs(a_ - b_) <= mtol * max(abs(b_), 1e-2), (k, a_, b_, flipped)
    del Ls, A, Bq
    torch.cuda.empty_cache()


def test_fused_onehot_rollout_equals_launch_sequence(hip):
  """csrc/imag_oh.hip: the one-hot / REINFORCE rollout at deter = units = 512 (xarm / ur5 blocks) as
  ONE persistent launch, forward only, against the per-layer launch sequence inside a whole train
  step on the same minibatch and weights: at the xarm shard's row count (25 x 50), on a ragged row
  count and with the 4-class instantiation.  Rows whose latent AND action draws all agree must
  agree in every buffer the actor's backward pass and the heads read; the losses and the actor /
  critic gradient norms of the step agree to the tolerance of the oracle parity tests."""
  for (B, T, H, adim) in ((25, 50, 15, 6), (7, 5, 4, 6), (3, 4, 3, 4)):
    plain, sp, shapes, params, data = helpers.make_named_problem('xarm', B, T, horizon=H)
    if adim != 6:   # the 4-class instantiation: same networks, another action space
      from daydreamer_amd import spec as spec_mod, synthetic
      obs, act = synthetic.config_spaces('xarm')
      act['action'] = synthetic.Space(np.float32, (adim,), 0, 1)
      act['action'].discrete = True
      sp = spec_mod.build_spec(plain, shapes, adim, True)
      params = spec_mod.init_params(sp, 0)
      data = synthetic.make_batch(obs, act, B, T, seed=2, terminals=0.01, smooth_images=True)
      idx = np.random.RandomState(3).randint(0, adim, (B, T))
      data['action'] = np.eye(adim, dtype=np.float32)[idx]
    Ls, mets = [], []
    for fused in (True, False):
      plain2 = dict(plain, hip=dict(plain.get('hip', {}), fused_imag=fused))
      sp2 = type(sp)(**{**sp.__dict__, 'cfg': plain2})
      L = learner_mod.Learner(sp2, hip, 'cuda:0', B, T, params=params, noise_seed=7)
      assert L.discrete and L.fused_imag == fused and not L.fused_imag_bwd
      L.upload(data)
      L.train_step_device(use_carry=False)
      torch.cuda.synchronize()
      mets.append(L.read_metrics())
      Ls.append(L)
    A, Bq = Ls
    N, G, C, D, F, Ad = A.N, A.G, A.C, A.D, A.F, A.A
    assert A.TW == Bq.TW == (F + Ad + 3) // 4 * 4
    ta, tb = A.b['traj'], Bq.b['traj']
    sa, sb = ta[:, :, D:F].reshape(H + 1, N, G, C), tb[:, :, D:F].reshape(H + 1, N, G, C)
    aa, ab = ta[:, :, F:F + Ad], tb[:, :, F:F + Ad]
    assert torch.equal(sa.sum(-1), torch.ones_like(sa.sum(-1))) and torch.equal(aa.sum(-1), torch.ones_like(aa.sum(-1)))
    assert float(ta[:, :, F + Ad:].abs().max()) == 0.0 if A.TW > F + Ad else True      # the padding stays zero
    same = ((sa.argmax(-1) == sb.argmax(-1)).all(-1) & (aa.argmax(-1) == ab.argmax(-1))).all(0)
    flipped = int((~same).sum())
    print(f'fused one-hot rollout B{B} T{T} H{H} A{adim}: {flipped} of {N} rows contain a flipped draw')
    _record_flips('fused one-hot rollout', B, T, H, adim, flipped, N)
    assert flipped <= max(1, N // 500)
    def cmp(x, y, what, rows_per_t, tol=5e-5):
      x = x.reshape(rows_per_t, N, -1)[:, same].double()
      y = y.reshape(rows_per_t, N, -1)[:, same].double()
      err = float((x - y).abs().max() / (y.abs().max() + 1e-30))
      assert err < tol, (what, err)
    cmp(ta, tb, 'traj', H + 1)
    la, lb = A.acts_im['actor'], Bq.acts_im['actor']
    for i in range(len(la[0])):
      cmp(la[0][i].z, lb[0][i].z, f'actor{i}.z', H + 1)
      cmp(la[0][i].stats, lb[0][i].stats, f'actor{i}.stats', H + 1)
      cmp(la[0][i].out, lb[0][i].out, f'actor{i}.out', H + 1)
    cmp(la[1][0].z, lb[1][0].z, 'actor logits', H + 1)
    cmp(A.b['alogit'], Bq.b['alogit'], 'actor log-probabilities', H + 1)
    cmp(A.b['iz3'], Bq.b['iz3'], 'z3', H)
    cmp(A.ai_img_stats.z, Bq.ai_img_stats.z, 'img_stats', H)
    mtol = 1e-3 + 4.0 * flipped / N
    for k in ('model_loss', 'extr_critic_loss', 'actor_loss', 'actor_grad_norm', 'extr_critic_grad_norm', 'actent_mean'):
      a_, b_ = float(mets[0][k]), float(mets[1][k])
      assert abs(a_ - b_) <= mtol * max(abs(b_), 1e-2), (k, a_, b_, flipped)
    del Ls, A, Bq
    torch.cuda.empty_cache()


@pytest.mark.parametrize('rows', [16, 32])
def test_fused_imagination_reverse_equals_launch_sequence(hip, rows):
  """(rows: of the imagination batch per workgroup - csrc/imag.hip 16, csrc/imag32.hip 32.)
  csrc/imag.hip k_imagine_reverse: the data gradient of the imagined rollout (steps H .. 1 of
  draw / img_stats / img_out / GRU / img_in backward) as ONE persistent launch against the
  per-layer launch sequence, inside a whole train step with the same (fused) forward rollout:
  the gradient of every imagined state and action (dtraj) and the actor's parameter gradients
  must agree to float reassociation."""
  for (B, T, H, adim) in ((50, 50, 15, 16), (21, 7, 5, 16), (2, 3, 2, 6)):
    cfg = helpers.make_config(('a1_vision',), batch_size=B, replay_chunk=T, imag_horizon=H)
    plain, sp, shapes, params, data, _, _ = helpers.make_problem(
        cfg, image=64, vector=16, action=adim, terminals=0.02, smooth=True)
    Ls = []
    for fused in (True, False):
      plain2 = dict(plain, hip=dict(plain.get('hip', {}), fused_imag_bwd=fused, imag_rows=rows))
      sp2 = type(sp)(**{**sp.__dict__, 'cfg': plain2})
      L = learner_mod.Learner(sp2, hip, 'cuda:0', B, T, params=params, noise_seed=7)
      assert L.fused_imag and L.fused_imag_bwd == fused
      L.upload(data)
      L.train_step_device(use_carry=False)
      torch.cuda.synchronize()
      Ls.append(L)
    A, Bq = Ls
    assert torch.equal(A.b['traj'], Bq.b['traj'])             # same forward
    def cmp(x, y, what, tol):
      x, y = x.double(), y.double()
      err = float((x - y).abs().max() / (y.abs().max() + 1e-30))
      assert err < tol, (what, err)
    F, D = A.F, A.D
    da, db = A.b['dtraj'], Bq.b['dtraj']
    cmp(da[:, :, F:], db[:, :, F:], 'd action', 2e-5)
    cmp(da[:, :, :D], db[:, :, :D], 'd deter', 2e-5)
    cmp(da[:, :, D:F], db[:, :, D:F], 'd stoch', 2e-5)
    ga, gb = A.export_grads(), Bq.export_grads()
    for name in ga:
      if name.startswith('actor/'):
        assert helpers.rel_err(ga[name], gb[name]) < 5e-5, name
    del Ls, A, Bq
    torch.cuda.empty_cache()


def test_fused_reverse_scan_equals_launch_sequence(hip):
  """csrc/scan.hip k_observe_scan_bwd: the data gradient of the T obs_steps as ONE persistent
  launch against the per-layer launch sequence, on the same forward state and the same incoming
  gradients (random dfeat / KL gradients), at the full configs[1] size, on a ragged batch and at
  deter = units = 512 (the xarm / ur5 blocks; weight planes streamed):
  every buffer the bulk weight-gradient contractions read and the resulting parameter
  gradients agree to float reassociation (and the hardware exp2 / reciprocal forms of the gates)."""
  for (name, B, T, first_mid) in (('a1_vision', 50, 50, False), ('a1_vision', 21, 7, True), ('a1_vision', 64, 3, False),
                                  ('a1_vision', 1, 2, False), ('xarm', 25, 50, False), ('xarm', 21, 7, True)):
    cfg = helpers.make_config((name,), batch_size=B, replay_chunk=T)
    if name == 'xarm':   # deter = units = 512: the variant of the kernel that streams its weight planes
      cfg = cfg.update({'encoder.mlp_keys': 'vector', 'decoder.mlp_keys': 'vector',
                        'encoder.cnn_keys': 'image', 'decoder.cnn_keys': 'image'})
      plain, sp, shapes, params, data, _, _ = helpers.make_problem(
          cfg, image=64, vector=20, action=6, terminals=0.02, smooth=True, discrete=True)
    else:
      plain, sp, shapes, params, data, _, _ = helpers.make_problem(
          cfg, image=64, vector=16, action=16, terminals=0.02, smooth=True)
    if first_mid:
      data['is_first'][3, 4] = True
      data['is_first'][20, 2] = True
    L = learner_mod.Learner(sp, hip, 'cuda:0', B, T, params=params, noise_seed=5)
    assert L.fused_scan and L.fused_scan_bwd
    L.upload(data)
    L.reset_carry()
    L.phase_prep()
    L.encoder_fwd()
    L.initial_fwd()
    L.observe_fwd(True)
    gen = torch.Generator(device='cuda').manual_seed(3)
    rnd = lambda t, s: torch.randn(t.shape, generator=gen, device='cuda') * s
    seeds = {k: rnd(L.b[k], s) for k, s in (('dfeat', 1e-2), ('dpost_logit', 1e-3), ('dprior_logit', 1e-3))}
    names = ('dfeat', 'dz3', 'dy3', 'dgin', 'dxin_s')
    acts = (('Aq.dout', lambda l: l.a_obs_stats.dout), ('Ao.dout', lambda l: l.a_obs_out.dout),
            ('Ao.dz', lambda l: l.a_obs_out.dz), ('Ai.dz', lambda l: l.a_img_in.dz))
    got = []
    for fused in (True, False):
      L.fused_scan_bwd = fused
      for k, v in seeds.items():
        L.b[k].copy_(v)
      L.groups['model'].gflat.zero_()
      L.observe_bwd()
      torch.cuda.synchronize()
      out = {k: L.b[k].clone() for k in names}
      out.update({k: f(L).clone() for k, f in acts})
      out['grads'] = L.groups['model'].gflat.clone()
      got.append(out)
    assert int(L.scan_sync[1]) == 0, 'a grid-barrier spin timed out'
    for k in got[0]:
      x, y = got[0][k].double(), got[1][k].double()
      err = float((x - y).abs().max() / (y.abs().max() + 1e-30))
      print(f'reverse scan {name} B{B} T{T} {k}: {err:.2e}')
      assert err < 2e-5, (k, err)
    del L, got
    torch.cuda.empty_cache()


def test_full_size_properties(hip):
  """BASELINE configs[1] at full size (batch 50 x seq 50 x horizon 15): the oracle
  is too slow here, so check size-independent properties instead: finite losses,
  one-hot latents, weights in [0,1] and non-increasing, gradient-norm
  consistency between the flat arena and the per-tensor views."""
  cfg = helpers.make_config(('a1_vision',))
  plain, sp, shapes, params, data, B, T = helpers.make_problem(
      cfg, image=64, vector=16, action=16, terminals=0.01, smooth=False)
  L = learner_mod.Learner(sp, hip, 'cuda:0', B, T, params=params)
  L.upload(data)
  L.train_step_device(use_carry=False)
  torch.cuda.synchronize()
  mets = L.read_metrics()
  for k, v in mets.items():
    assert np.isfinite(v), k
  b = L.b
  st = b['post'][:, L.D:].view(L.N, L.G, L.C)
  assert torch.equal(st.sum(-1), torch.ones_like(st.sum(-1)))
  tr = b['traj'][:, :, L.D:L.F].reshape(L.H + 1, L.N, L.G, L.C)
  assert torch.equal(tr.sum(-1), torch.ones_like(tr.sum(-1)))
  w = b['i_weight'].view(L.H + 1, L.N)
  assert float(w.min()) >= 0 and float(w.max()) <= 1.0
  assert bool((w[1:] <= w[:-1] + 1e-6).all())
  g = L.groups['model']
  n1 = float(torch.sqrt((g.gflat.double() ** 2).sum()))
  assert abs(n1 - float(mets['model_grad_norm'])) <= 1e-4 * n1
  # lambda-return fixed point: with lambda = 0 ... covered in test_hip_ops; here check
  # ret_H-1 = r + d * v_H exactly as the recurrence defines it
  H, N = L.H, L.N
  r = b['i_reward'].view(H, N)[H - 1]
  v = b['i_value'].view(H + 1, N)[H]
  d = b['i_cont'].view(H + 1, N)[H] * plain['discount']
  ret = b['i_ret'].view(H, N)[H - 1]
  assert torch.allclose(ret, r + d * v, rtol=1e-5, atol=1e-5)


def test_full_size_parity_vs_oracle(hip):
  """BASELINE configs[1] at FULL size (batch 50 x seq 50 x horizon 15, 64x64 image,
  deter 256, 32x32 latent, 16-dim action): one complete train step on the HIP path
  against the float64 oracle on the same minibatch, weights and noise (~1 min of
  host time for the oracle).  Tolerances as north_star: losses / grad norms 1e-3
  relative; gradients 5e-3 of their max at this depth and row count."""
  import torch
  from oracle import dreamer_ref
  torch.set_num_threads(16)
  reset_adopted()
  cfg = helpers.make_config(('a1_vision',))
  plain, sp, shapes, params, data, B, T = helpers.make_problem(
      cfg, image=64, vector=16, action=16, terminals=0.01, smooth=True)
  assert (B, T, plain['imag_horizon']) == (50, 50, 15)
  L = learner_mod.Learner(sp, hip, 'cuda:0', B, T, params=params, noise_seed=9)
  L.upload(data)
  L.train_step_device(use_carry=False)
  torch.cuda.synchronize()
  mets = L.read_metrics()
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64)
  _, _, omets = ag.train(data, helpers.noise_from_learner(L), None,
                         helpers.forced_from_learner(L))
  for k in LOSS_KEYS:
    if k in omets:
      a, o = float(mets[k]), float(omets[k])
      assert abs(a - o) <= 1e-3 * max(abs(o), 1e-2), f'{k}: {a} vs {o}'
  grads = L.export_grads()
  worst = max((helpers.rel_err(grads[n], g.numpy()), n)
              for n, g in ag.last['grads'].items())
  assert worst[0] < 5e-3, f'worst grad {worst}'
  print('full-size parity: model_loss', float(mets['model_loss']), 'vs', float(omets['model_loss']),
        'worst grad rel err', worst)
  check_adopted('configs[1] B50 T50 H15')


def full_size(hip, name, B, T, H=None, threads=32, gtol=5e-3, **overrides):
  """One complete train step of a BASELINE workload's per-GPU shard - its own networks at
  full width - on the HIP path against the float64 oracle (same minibatch, weights, noise)."""
  import time
  from oracle import dreamer_ref
  torch.set_num_threads(threads)
  reset_adopted()
  plain, sp, shapes, params, data = helpers.make_named_problem(name, B, T, horizon=H, **overrides)
  L = learner_mod.Learner(sp, hip, 'cuda:0', B, T, params=params, noise_seed=11)
  L.upload(data)
  L.train_step_device(use_carry=False)
  torch.cuda.synchronize()
  mets = L.read_metrics()
  t0 = time.time()
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64,
                            act_discrete=sp.act_discrete)
  _, _, omets = ag.train(data, helpers.noise_from_learner(L), None,
                         helpers.forced_from_learner(L))
  for k in LOSS_KEYS + tuple(f'{key}_loss_mean' for key in list(sp.dec_cnn_keys) + list(sp.dec_mlp_keys)):
    if k in omets:
      a, o = float(mets[k]), float(omets[k])
      assert abs(a - o) <= 1e-3 * max(abs(o), 1e-2), f'{name} {k}: {a} vs {o}'
  grads = L.export_grads()
  worst = max((helpers.rel_err(grads[n], g.numpy()), n) for n, g in ag.last['grads'].items())
  print(f'{name} B{B} T{T} H{L.H} (deter {sp.deter}, units {sp.units}, stoch {sp.groups}x{sp.classes}): '
        f'model_loss {float(mets["model_loss"]):.4f} vs {float(omets["model_loss"]):.4f}, '
        f'worst grad rel err {worst}, oracle {time.time() - t0:.0f} s')
  assert worst[0] < gtol, f'{name} worst grad {worst}'
  check_adopted(f'{name} B{B} T{T} H{L.H}')
  del L, ag
  torch.cuda.empty_cache()


def test_full_size_xarm_shard(hip):
  """BASELINE configs[2] (xarm, 2 GPUs data parallel): one GPU's shard at real size -
  batch 25 x seq 50 x horizon 15, image + depth (4 channels) + 20 proprio dims in five keys,
  deter = units = 512, one-hot 6-way action trained by REINFORCE."""
  full_size(hip, 'xarm', 25, 50, batch_size=50, replay_chunk=50)


def test_full_size_ur5_multicam_shard(hip):
  """BASELINE configs[3] (ur5, two 128x128 cameras, 4 GPUs): one GPU's shard at real size -
  batch 16 x seq 64 x horizon 15 with the FULL networks (deter = units = 512, cnn depth 64,
  decoder kernels 5,5,6,6,2), one-hot 6-way action."""
  full_size(hip, 'ur5_multicam', 16, 64)


def test_full_size_a1_scaled_shard(hip):
  """BASELINE configs[4] (a1 scaled, 8 GPUs): one GPU's shard at REAL size - batch 32 x seq 64 x
  horizon 20 with the scaled networks themselves (deter 4096, stoch 64 x 64; 147 M parameters,
  2.9 M latent draws): ~2.7 min, of which the float64 oracle takes 2.6 (DD_A1_SCALED_T
  shortens the sequence for a quicker run; the network is never shrunk)."""
  import os
  T = int(os.environ.get('DD_A1_SCALED_T', 64))
  full_size(hip, 'a1_scaled', 32, T, replay_chunk=T)


def test_bfloat16_mode_separately_toleranced(hip):
  """hip.precision: bfloat16 - the opt-in counterpart of the reference's tf.precision float16
  (tfagent.py:161-168): contraction operands rounded to bf16, fp32 accumulation / storage /
  optimizer.  NOT the parity mode: tolerances are its own (losses 3e-2 relative, gradients
  0.15 of their max against the float64 oracle; latent draws adopted within 5e-2 of a CDF
  edge, at most 10 % of them); the float32 default keeps 1e-3 / 2e-3 / 1e-6."""
  from oracle import dreamer_ref
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6, imag_horizon=4)
  plain, sp, shapes, params, data, B, T = helpers.make_problem(
      cfg, image=64, vector=5, action=3, terminals=0.1)
  prev = hip.set_gemm_mode(1)
  try:
    reset_adopted()
    dreamer_ref.SAMPLE_TOL[0] = 5e-2
    L = learner_mod.Learner(sp, hip, 'cuda:0', B, T, params=params, noise_seed=3)
    L.upload(data)
    L.train_step_device(use_carry=False)
    torch.cuda.synchronize()
    mets = L.read_metrics()
  finally:
    hip.set_gemm_mode(prev)
  ag = dreamer_ref.RefAgent(plain, shapes, sp.act_dim, params, torch.float64)
  _, _, omets = ag.train(data, helpers.noise_from_learner(L), None, helpers.forced_from_learner(L))
  worst_loss = max((abs(float(mets[k]) - float(omets[k])) / max(abs(float(omets[k])), 1e-2), k)
                   for k in LOSS_KEYS if k in omets)
  grads = L.export_grads()
  worst = max((helpers.rel_err(grads[n], g.numpy()), n) for n, g in ag.last['grads'].items())
  st = dreamer_ref.SAMPLE_STATS
  print(f'bfloat16 mode: worst loss rel err {worst_loss}, worst grad rel err {worst}, '
        f'{st["adopted"]} of {st["draws"]} draws adopted (largest gap {st["max_gap"]:.2e})')
  assert worst_loss[0] < 3e-2 and worst[0] < 0.15, (worst_loss, worst)
  assert st['adopted'] <= 0.1 * st['draws']
  # and through the Agent API: the config key selects the mode, training still converges
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=8, replay_chunk=8, imag_horizon=5)
  cfg = cfg.update({'model_opt.lr': 1e-3, 'hip.precision': 'bfloat16'})
  obs, act = synthetic.make_spaces(64, 5, 3)
  try:
    ag2 = agent_mod.Agent(obs, act, None, cfg)
    batch = synthetic.make_batch(obs, act, 8, 8, seed=4, smooth_images=True)
    state, first = None, None
    for i in range(30):
      _, state, m = ag2.train(batch, state)
      assert helpers.metrics_finite(m), i
      first = first if first is not None else float(m['model_loss'])
    assert float(m['model_loss']) < 0.9 * first
  finally:
    hip.set_gemm_mode(6)


def test_training_reduces_loss_through_agent(hip):
  """Through the public Agent API with HIP-graph replay: 40 train steps on a
  fixed small batch must reduce the world-model loss, keep every metric finite,
  advance all three optimizers, and policy() must keep working while training."""
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=8, replay_chunk=8,
                            imag_horizon=5)
  cfg = cfg.update({'model_opt.lr': 1e-3})
  obs, act = synthetic.make_spaces(64, 5, 3)
  ag = agent_mod.Agent(obs, act, None, cfg)
  data = synthetic.make_batch(obs, act, 8, 8, seed=4, smooth_images=True)
  state, first, last = None, None, None
  for i in range(40):
    _, state, mets = ag.train(data, state)
    assert helpers.metrics_finite(mets), i
    first = first if first is not None else float(mets['model_loss'])
    last = float(mets['model_loss'])
  mets = ag.flush() or mets
  last = float(mets['model_loss'])
  assert ag._plan is not None and ag._plan.n_graphs >= 2   # replayed from HIP graphs
  assert last < 0.9 * first, (first, last)
  assert float(mets['model_grad_steps']) == 40 and float(mets['actor_grad_steps']) == 40
  o = {'image': np.zeros((1, 64, 64, 3), np.uint8), 'vector': np.zeros((1, 5), np.float32),
       'reward': np.zeros(1, np.float32), 'is_first': np.array([True]),
       'is_last': np.zeros(1, bool), 'is_terminal': np.zeros(1, bool)}
  a, st = ag.policy(o, None, 'train')
  a2, st = ag.policy({**o, 'is_first': np.array([False])}, st, 'eval')
  assert a['action'].shape == (1, 3) and np.isfinite(a2['action']).all()


def test_training_from_device_replay(hip):
  """embodied.Replay on the GPU: episodes in an HBM ring, Agent.dataset(replay.dataset)
  yields device minibatches (dd_replay_gather), Agent.train consumes them without a
  host copy.  The first step equals the step on the same minibatch passed as numpy."""
  import numpy as np
  from daydreamer_amd import agent as agent_mod, replay as replay_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=6, replay_chunk=8, imag_horizon=4)
  obs, act = synthetic.make_spaces(64, 5, 3)
  rep = replay_mod.DeviceReplay(chunk=8, capacity=2000)
  for e in range(6):
    ep = synthetic.make_batch(obs, act, 1, 40, seed=e, terminals=0.0, smooth_images=True)
    for t in range(40):
      rep.add({**{k: v[0, t] for k, v in ep.items()}, 'is_first': t == 0, 'is_last': t == 39})
  assert rep.stats == {'replay_steps': 240, 'replay_trajs': 6}
  ag, ag2 = agent_mod.Agent(obs, act, None, cfg), agent_mod.Agent(obs, act, None, cfg)
  ds = ag.dataset(rep.dataset)
  batch = next(ds)
  assert batch['image'].is_cuda and batch['image'].dtype == torch.uint8
  host = {k: v.cpu().numpy() for k, v in batch.items()}
  _, state, m1 = ag.train(batch)
  _, _, m2 = ag2.train(host)
  for k in m1:
    assert np.array_equal(m1[k], m2[k], equal_nan=True), k
  for i in range(6):  # graph replay on fresh device minibatches
    _, state, mets = ag.train(next(ds), state)
    assert helpers.metrics_finite(mets), i
  assert ag._plan is not None


def test_pipelined_steps_equal_sequential(hip):
  """hip.pipeline: step k's behaviour phase (imagination, critic, actor) runs on a second
  stream next to step k+1's world-model phase.  Same arithmetic in the same order inside
  each phase => parameters, optimizer moments and controller state after n steps are
  bit-identical to the sequential schedule, and every call returns its own metrics (lazily: a
  LazyMetrics that is fetched when looked at, or when the next call has been enqueued)."""
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=6, replay_chunk=8, imag_horizon=4)
  obs, act = synthetic.make_spaces(64, 5, 3)
  batches = [synthetic.make_batch(obs, act, 6, 8, seed=s, smooth_images=True) for s in range(4)]
  runs = {}
  for mode in (False, True):
    ag = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': mode}))
    state, mets = None, []
    for i in range(9):
      # (call 5 starts from the initial state again: reset_carry next to work in flight)
      _, state, m = ag.train(batches[i % 4], None if i == 5 else state)
      if mode and i >= 1:
        assert isinstance(m, agent_mod.LazyMetrics) and not m.resolved    # train() did not wait for the step
        if len(mets) >= 2 and isinstance(mets[-1], agent_mod.LazyMetrics):
          assert mets[-1].resolved                                         # fetched when this call was enqueued
        if i == 3:
          assert float(m['model_loss']) == float(m['model_loss_mean']) and m.resolved   # looked at right away
      mets.append(m)
    last = ag.flush()
    if mode:
      assert isinstance(ag._plan, agent_mod.Pipeline) and last is not None
      assert dict(last) == dict(mets[-1])
      mets = [dict(m) for m in mets]
    runs[mode] = (ag.save(), mets)
  (sa, ma), (sb, mb) = runs[False], runs[True]
  assert sa.keys() == sb.keys()
  for k in sa:
    assert np.array_equal(np.asarray(sa[k]), np.asarray(sb[k]), equal_nan=True), k
  assert len(ma) == len(mb)
  for i, (x, y) in enumerate(zip(ma, mb)):
    for k in x:
      assert np.array_equal(x[k], y[k], equal_nan=True), (i, k)


def test_dataset_stages_minibatches_on_device(hip):
  """Agent.dataset over an ordinary host generator (reference Prefetch role): the prefetch
  thread stacks into pinned buffers and uploads on its own stream; train() gets device
  tensors with the wire dtypes and the same values as the host stack."""
  import itertools
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6, imag_horizon=3)
  obs, act = synthetic.make_spaces(64, 5, 3)
  ag = agent_mod.Agent(obs, act, None, cfg)
  def gen():
    for s in itertools.count():
      ep = synthetic.make_batch(obs, act, 1, 6, seed=s % 5, smooth_images=True)
      yield {k: v[0] for k, v in ep.items()}
  ds = iter(ag.dataset(gen))
  state = None
  for i in range(8):
    batch = next(ds)
    assert batch['image'].is_cuda and batch['image'].dtype == torch.uint8
    assert batch['is_first'].dtype == torch.bool and batch['reward'].dtype == torch.float32
    if i == 0:
      ref = [next(gen()) for _ in range(1)][0]   # every generator starts with seed 0
      for k, v in ref.items():
        assert np.array_equal(batch[k][0].cpu().numpy(), v), k
    _, state, mets = ag.train(batch, state)
    assert helpers.metrics_finite(mets)


@pytest.mark.parametrize('pipeline', [False, True])
def test_non_finite_gradient_raises_and_skips_the_update(hip, pipeline):
  """The check_numerics contract (reference tfutils.py:207,249; SURVEY 8b "Errors") on the HIP path,
  in both schedules: a non-finite weight makes every gradient norm non-finite; dd_grad_norm's
  device-side flag makes dd_adam_step skip the update (parameters, Adam moments and step counts
  untouched) and the host raises FloatingPointError - inside the train call (sequential schedule)
  or when the call's metrics are looked at, at the latest inside the NEXT train call (pipelined
  schedule).  Afterwards the agent still checkpoints and trains on."""
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6, imag_horizon=3)
  cfg = cfg.update({'hip.pipeline': pipeline})
  obs, act = synthetic.make_spaces(64, 5, 3)
  ag = agent_mod.Agent(obs, act, None, cfg)
  data = synthetic.make_batch(obs, act, 4, 6, seed=3, smooth_images=True)
  state = None
  for _ in range(3):
    _, state, mets = ag.train(data, state)
  ag.flush()
  assert isinstance(ag._plan, agent_mod.Pipeline) == pipeline
  before = ag.save()
  w = ag.groups['model'].p['reward/dense0/kernel']
  keep = w[0, 0].clone()
  w[0, 0] = float('nan')
  torch.cuda.synchronize()
  if pipeline:
    _, state, m1 = ag.train(data, None)          # enqueued; nobody looks at its metrics
    assert isinstance(m1, agent_mod.LazyMetrics)
    with pytest.raises(FloatingPointError):
      ag.train(data, None)                       # ... so it surfaces while the next call is enqueued
    with pytest.raises(FloatingPointError):
      float(m1['model_loss'])                    # (and again for whoever looks at that call's metrics)
    with pytest.raises(FloatingPointError):
      ag.flush()                                 # the step enqueued by the raising call is non-finite too
    assert ag.flush() is None                    # raised once; the pipeline is drained
  else:
    with pytest.raises(FloatingPointError):
      ag.train(data, None)
  after = ag.save()                              # (checkpointing still works)
  assert before.keys() == after.keys()
  for k in before:
    a, b = np.asarray(before[k]), np.asarray(after[k])
    if k == 'params/reward/dense0/kernel':
      assert np.isnan(b[0, 0]) and np.array_equal(a.reshape(-1)[1:], b.reshape(-1)[1:])
    elif k != 'state/noise_step':
      # parameters, Adam moments, optimizer step counts, the slow critic: untouched.  (The
      # controllers that are updated before the optimizer - AutoAdapt, Normalize - hold what
      # the non-finite batch statistics made of them, as in the reference's graph.)
      if k.startswith(('params/', 'opt/')):
        assert np.array_equal(a, b, equal_nan=True), k
  # repair the weight and the controllers, train on from the initial state
  w[0, 0] = keep
  ag.load(before)
  for _ in range(3):
    _, state, mets = ag.train(data, None if _ == 0 else state)
    assert helpers.metrics_finite(mets)
  ag.flush()


def test_stream_pair_measurement_keeps_parameters_bit_identical(hip):
  """The pipelined agent's stream-pair measurement (agent.Pipeline: 12 ordered pairs x TRIAL real
  train steps inside the first pipelined calls, each pair with its own captured graphs): the steps
  it runs are ordinary train steps - parameters, moments and metrics after it equal the
  sequential schedule's, bit for bit - and the selected pair is cached for the process."""
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6, imag_horizon=3)
  obs, act = synthetic.make_spaces(64, 5, 3)
  batches = [synthetic.make_batch(obs, act, 4, 6, seed=s, smooth_images=True) for s in range(3)]
  n = 2 + 12 * agent_mod.Pipeline.TRIAL + 1 + 3
  runs = {}
  keep = dict(agent_mod.Pipeline.BEST)
  try:
    for mode in (False, True):
      ag = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': mode}))
      state = None
      for i in range(n):
        if mode and i == 2:
          assert ag._pipe is not None
          ag._pipe.k_tune, ag._pipe.periods, ag._pipe.ticks, ag._pipe.tuned = 0, {}, [], False   # (DD_PIPE_TUNE=0 in the tests)
        _, state, m = ag.train(batches[i % 3], state)
      last = ag.flush() or m
      if mode:
        pipe = ag._pipe
        assert pipe.tuned and len(pipe.periods) == 12 and all(p > 0 for p in pipe.periods.values())
        assert agent_mod.Pipeline.BEST[pipe.key] == pipe.pair == min(pipe.periods, key=pipe.periods.get)
        assert list(pipe.plans) == [pipe.pair]            # the losing pairs' graphs are retired
      runs[mode] = (ag.save(), dict(last))
  finally:
    agent_mod.Pipeline.BEST.clear()
    agent_mod.Pipeline.BEST.update(keep)
  (sa, ma), (sb, mb) = runs[False], runs[True]
  for k in sa:
    assert np.array_equal(np.asarray(sa[k]), np.asarray(sb[k]), equal_nan=True), k
  for k in ma:
    assert np.array_equal(ma[k], mb[k], equal_nan=True), k


def test_auto_schedule_follows_the_callers_loop(hip):
  """hip.pipeline: auto (the default).  Train calls in a row (the learner process of
  run/learning.py) run the two-stream pipeline; once every train call follows a policy call
  (run/train.py: act, then train - each policy call drains the pipeline, which then is the
  sequential step without its in-step overlaps) the sequential plan is replayed instead, and
  the pipeline again when the train calls come in a row again.  Parameters are those of the
  sequential schedule throughout, bit for bit."""
  import numpy as np
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=6, imag_horizon=3)
  obs, act = synthetic.make_spaces(64, 5, 3)
  batches = [synthetic.make_batch(obs, act, 4, 6, seed=s, smooth_images=True) for s in range(3)]
  o = {k: v[:, 0] for k, v in batches[0].items() if k not in ('action', 'reset')}
  K = agent_mod.Agent.ADAPT
  pattern = [False] * (K + 2) + [True] * (K + 3) + [False] * (K + 2)     # policy call before the train call?
  runs = {}
  for mode in ('auto', False):
    ag = agent_mod.Agent(obs, act, None, cfg if mode == 'auto' else cfg.update({'hip.pipeline': False}))
    state, pst, kinds = None, None, []
    for i, act_first in enumerate(pattern):
      if act_first:
        _, pst = ag.policy(o, pst, 'train')
      _, state, m = ag.train(batches[i % 3], state)
      kinds.append(isinstance(m, agent_mod.LazyMetrics))
    last = dict(ag.flush() or m)
    if mode == 'auto':
      a, b = K + 2, 2 * K + 5
      assert kinds[0] is False and all(kinds[1:a])                 # eager first call, then pipelined
      assert all(kinds[a:a + K - 1]) and not any(kinds[a + K - 1:b])   # K calls behind a policy call: sequential plan
      assert ag._seq_plan is not None and ag._pipe is not None
      assert not any(kinds[b:b + K - 1]) and all(kinds[b + K - 1:])    # K calls in a row again: pipelined
      assert ag._plan is ag._pipe
    else:
      assert not any(kinds)
    runs[mode] = (ag.save(), last)
  (sa, ma), (sb, mb) = runs['auto'], runs[False]
  for k in sa:
    assert np.array_equal(np.asarray(sa[k]), np.asarray(sb[k]), equal_nan=True), k
  for k in ma:
    assert np.array_equal(ma[k], mb[k], equal_nan=True), k
