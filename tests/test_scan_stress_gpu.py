"""Inter-workgroup hand-off of the persistent observe-scan kernels under UNEVEN load.

The scan kernels exchange every phase's output between the 16 workgroups of a row block inside one
launch: write-through (sc1) stores, drained, a relaxed arrival on the row block's counter, and on
the consumer side one relaxed poll + one agent-scope acquire + plain loads (Guideline 16 form R1 of
the CDNA4 guide).  A protocol mistake shows as STALE reads, and - per the guide - mostly under
uneven load with L1-warm consumers, which an idle-chip parity test can miss.  So: the kernels are
deterministic; run the forward and the reverse scan many times while another stream keeps the chip
unevenly busy with contractions of varying size, and require every output buffer to be
bit-identical to the first (quiet) run, and to the run with the rounds-2/3 protocol (grid-wide
counter + release fence at every arrival, flags 128 | 256).
"""

import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu

FWD_OUT = ('xin', 'gin', 'z3', 'gstats', 'post', 'post_logit')
BWD_OUT = ('dfeat', 'dz3', 'dy3', 'dgin', 'dxin_s')


def _learner(hip):
  from daydreamer_amd import learner as LM
  cfg = helpers.make_config(('a1_vision',), batch_size=50, replay_chunk=20, imag_horizon=2)
  plain, sp, shapes, params, data, B, T = helpers.make_problem(
      cfg, image=64, vector=16, action=16, terminals=0.1)
  L = LM.Learner(sp, hip, 'cuda:0', B, T, params=params, noise_seed=7)
  assert L.fused_scan and L.fused_scan_bwd
  L.upload(data)
  L.train_step_device(use_carry=False)
  torch.cuda.synchronize()
  return L


def _fwd(L, flags=0):
  b, P = L.b, L.P
  g = P['gru_h']
  L.ops.observe_scan_fwd(
      L.B, L.T, L.D, L.U, L.G, L.C, L.A, 1 | flags, L.unimix, b['first'], b['carry'], b['init_deter'],
      b['init_stoch'], b['u_post'], [w[1] for w in L.scan_w],
      [P['img_in'].gamma, P['img_in'].beta, g.gamma, g.beta, P['obs_out_h'].gamma,
       P['obs_out_h'].beta, P['obs_stats'].bias],
      [b['xin'], L.a_img_in.z, L.a_img_in.stats, b['gin'], b['z3'], b['gstats'], b['post'],
       L.a_obs_out.z, L.a_obs_out.out, L.a_obs_out.stats, L.a_obs_stats.z, b['post_logit']],
      P['img_in'].W, L.scan_idx, L.scan_sync)


def _bwd(L, flags=0):
  b, P = L.b, L.P
  Aq, Ao, Ai = L.a_obs_stats, L.a_obs_out, L.a_img_in
  g = P['gru_h']
  L.ops.observe_scan_bwd(
      L.B, L.T, L.D, L.U, L.G, L.C, flags, L.unimix, b['first'],
      [Aq.z, Ao.z, Ao.out, Ao.stats, b['z3'], b['gstats'], b['gin'], Ai.z, Ai.stats],
      b['dpost_logit'], [w[1] for w in L.scan_wb],
      [P['obs_out_h'].gamma, g.gamma, g.beta, P['img_in'].gamma],
      [b['dfeat'], Aq.dout, Ao.dout, Ao.dz, b['dz3'], b['dy3'], b['dgin'], Ai.dz, b['dxin_s']],
      L.scan_sync)


def test_scan_handoff_is_exact_under_uneven_load(hip):
  L = _learner(hip)
  b = L.b
  acts = {'a_img_in.z': L.a_img_in.z, 'a_obs_out.z': L.a_obs_out.z, 'a_obs_stats.z': L.a_obs_stats.z,
          'a_obs_out.dz': L.a_obs_out.dz, 'a_img_in.dz': L.a_img_in.dz}
  gen = torch.Generator(device='cuda').manual_seed(3)
  seed = {k: torch.randn(b[k].shape, generator=gen, device='cuda') * sc
          for k, sc in (('dfeat', 1e-2), ('dpost_logit', 1e-3))}
  carry0 = b['carry'].clone()
  zo0 = None

  def run(flags=0):
    nonlocal zo0
    # the forward scan accumulates into zo (the hoisted embed part) and the reverse scan into dfeat:
    # restore their inputs so that every run computes the same thing
    b['carry'].copy_(carry0)
    if zo0 is None:
      L.encoder_fwd()
      zo0 = L.a_obs_out.z.clone()
    L.a_obs_out.z.copy_(zo0)
    _fwd(L, flags)
    for k, v in seed.items():
      b[k].copy_(v)
    L.a_obs_stats.dout.zero_()
    _bwd(L, flags)
    torch.cuda.synchronize()
    out = {k: b[k].clone() for k in FWD_OUT + BWD_OUT}
    out.update({k: v.clone() for k, v in acts.items()})
    return out

  ref = run()
  assert int(L.scan_sync[1]) == 0
  assert all(torch.isfinite(v).all() for v in ref.values())
  old = run(128 | 256)            # grid-wide counter + release fences: the rounds-2/3 protocol
  for k in ref:
    assert torch.equal(ref[k], old[k]), f'{k}: write-through protocol != release-fence protocol'

  # uneven background load on another stream: contractions of varying size, a few in flight
  side = torch.cuda.Stream('cuda:0')
  mats = [torch.randn(n, n, device='cuda') for n in (512, 1024, 2048, 3072)]
  outs = [torch.empty_like(m) for m in mats]
  bad = []
  for it in range(150):
    with torch.cuda.stream(side):
      for j in range(1 + it % 4):
        m = mats[(it + j) % 4]
        torch.mm(m, m, out=outs[(it + j) % 4])
    got = run()
    for k in ref:
      if not torch.equal(ref[k], got[k]):
        bad.append((it, k, int((ref[k] != got[k]).sum())))
    if bad:
      break
  side.synchronize()
  assert int(L.scan_sync[1]) == 0, 'a barrier spin timed out'
  assert not bad, f'stale hand-off under load: {bad[:5]}'
