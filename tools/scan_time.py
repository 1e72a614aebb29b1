"""Time the fused observe scan (dd_observe_scan_fwd) at BASELINE configs[1]: the full kernel, its
grid barriers alone (dry mode: 4 barriers per step, no work), the weight-plane preparation,
and the per-layer launch sequence it replaces.  python tools/scan_time.py"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import helpers
from daydreamer_amd import learner as LM, hipops, synthetic, config as config_mod, spec as spec_mod

NAME = sys.argv[1] if len(sys.argv) > 1 else 'a1_vision'
cfg = helpers.make_config((NAME,))
plain = config_mod.to_plain(cfg)
obs, act = synthetic.config_spaces(NAME)
shapes = {k: v.shape for k, v in obs.items()}
adim = act['action'].shape[-1] if hasattr(act['action'], 'shape') else 16
sp = spec_mod.build_spec(plain, shapes, adim, getattr(act['action'], 'discrete', False))
B, T = plain['batch_size'], plain['replay_chunk']
data = synthetic.make_batch(obs, act, B, T, seed=0)
ops = hipops.HipOps('cuda:0')
L = LM.Learner(sp, ops, 'cuda:0', B, T, params=spec_mod.init_params(sp, 0))
L.upload(data)
L.train_step_device(False)
torch.cuda.synchronize()


def timed(label, fn, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  print(f'{label:44s} {e0.elapsed_time(e1) / reps:8.3f} ms', flush=True)


def scan(flag):
  b, P = L.b, L.P
  g = P['gru_h']
  ops.observe_scan_fwd(
      L.B, L.T, L.D, L.U, L.G, L.C, L.A, flag, L.unimix, b['first'], b['carry'], b['init_deter'],
      b['init_stoch'], b['u_post'], [w[1] for w in L.scan_w],
      [P['img_in'].gamma, P['img_in'].beta, g.gamma, g.beta, P['obs_out_h'].gamma,
       P['obs_out_h'].beta, P['obs_stats'].bias],
      [b['xin'], L.a_img_in.z, L.a_img_in.stats, b['gin'], b['z3'], b['gstats'], b['post'],
       L.a_obs_out.z, L.a_obs_out.out, L.a_obs_out.stats, L.a_obs_stats.z, b['post_logit']],
      P['img_in'].W, L.scan_idx, L.scan_sync)


timed('weight planes (3 x dd_scan_wprep)', lambda: [ops.scan_wprep(W, p, k) for W, p, k in L.scan_w[1:]])
timed('fused scan, full (T = %d steps)' % T, lambda: scan(1))
timed('fused scan, full, grid-wide barrier counter (flag 128)', lambda: scan(1 | 128))
timed('fused scan, full, release fence at every arrival (flag 256)', lambda: scan(1 | 256))
timed('fused scan, full (write-through stores, no release fence) again', lambda: scan(1))
timed('fused scan, full, release fence again', lambda: scan(1 | 256))
timed('fused scan, barriers only (4 per step)', lambda: scan(3))
for nm, bit in (('P1 img_in (gather)', 4), ('P2 gru gemm', 8), ('P3 gru gates + obs_out', 16), ('P4 obs_stats + draw', 32)):
  timed('fused scan without ' + nm, lambda bit=bit: scan(1 | bit))
print('error word', int(L.scan_sync[1]))
scan(1 | 64)
torch.cuda.synchronize()
ts = L.scan_sync[2:2 + 38].view(torch.int64).cpu().numpy()
names = ['step start', 'P1 body', 'barrier 1', 'P2 stats', 'P2 operand', 'P2 gemm', 'P2 stores', 'barrier 2',
         'P3 stats', 'P3 operand', 'P3 gemm', 'P3 stores', 'barrier 3', 'P4 stats', 'P4 operand', 'P4 gemm',
         'P4 bias/LDS', 'P4 draw', 'barrier 4']
print('step 10, workgroup 0 (us since step start / delta):')
for i in range(1, 19):
  print(f'  {names[i]:12s} {(ts[i] - ts[0]) / 100:7.2f} {(ts[i] - ts[i - 1]) / 100:6.2f}')
L.fused_scan = False
from daydreamer_amd import graphs
plan = graphs.GraphPlan('cuda:0')
keep, L.plan = L.plan, plan
plan.capture(lambda: L.observe_fwd(True))
L.plan = keep
timed('launch sequence (graph replay, incl. bulk prior)', plan.replay)

import numpy as np
arr = L.scan_sync[64:64 + 2 * 4 * 64].view(torch.int64).cpu().numpy().reshape(64, 4)
print('arrival of the workgroups at the 4 barriers of step 10 (us after the first arrival): median / last; slowest workgroup')
for i in range(4):
  d = (arr[:, i] - arr[:, i].min()) / 100
  print(f'  barrier {i + 1}: {np.median(d):5.2f} {d.max():5.2f}  wg {int(d.argmax())} (wg {int(d.argmax())});'
        f' released {(ts[[2, 7, 12, 18][i]] - arr[:, i].max()) / 100:5.2f} us after the last arrival')


# ---- reverse scan
L.fused_scan = True
gen = torch.Generator(device='cuda').manual_seed(3)
seed = {k: torch.randn(L.b[k].shape, generator=gen, device='cuda') * sc
        for k, sc in (('dfeat', 1e-2), ('dpost_logit', 1e-3), ('dprior_logit', 1e-3))}
def reseed():
  for k, v in seed.items():
    L.b[k].copy_(v)
def scan_bwd(flags=0):
  b, P = L.b, L.P
  Aq, Ao, Ai = L.a_obs_stats, L.a_obs_out, L.a_img_in
  g = P['gru_h']
  ops.observe_scan_bwd(
      L.B, L.T, L.D, L.U, L.G, L.C, flags, L.unimix, b['first'],
      [Aq.z, Ao.z, Ao.out, Ao.stats, b['z3'], b['gstats'], b['gin'], Ai.z, Ai.stats],
      b['dpost_logit'], [w[1] for w in L.scan_wb],
      [P['obs_out_h'].gamma, g.gamma, g.beta, P['img_in'].gamma],
      [b['dfeat'], Aq.dout, Ao.dout, Ao.dz, b['dz3'], b['dy3'], b['dgin'], Ai.dz, b['dxin_s']],
      L.scan_sync)
reseed()
L.fused_scan_bwd = True
L.observe_scan_bwd_fused()
timed('reverse scan kernel alone (T = %d steps)' % T, lambda: (reseed(), scan_bwd()))
timed('reverse scan kernel, grid-wide barrier counter (flag 128)', lambda: (reseed(), scan_bwd(128)))
timed('reverse scan kernel, release fence at every arrival (flag 256)', lambda: (reseed(), scan_bwd(256)))
timed('reverse scan kernel alone (write-through stores, no release fence) again', lambda: (reseed(), scan_bwd()))
timed('reverse scan kernel, release fence again', lambda: (reseed(), scan_bwd(256)))
timed('  (the three seed copies in that figure)', reseed)
for fused in (False, True):
  L.fused_scan_bwd = fused
  p2 = graphs.GraphPlan('cuda:0')
  keep, L.plan = L.plan, p2
  reseed()
  p2.capture(lambda: L.observe_bwd())
  L.plan = keep
  timed('observe_bwd incl. bulk weight gradients, ' + ('fused reverse scan' if fused else 'launch sequence'),
        lambda: (reseed(), p2.replay()))
reseed()
scan_bwd(64)
torch.cuda.synchronize()
ts = L.scan_sync[2:2 + 24].view(torch.int64).cpu().numpy()
names = ['start', 'Q1 dxo', 'barrier 1', 'Q2 ln + dgrad', 'barrier 2', 'Q3 gru + dgrad', 'barrier 3', 'Q4 ln + dgrad + stats', 'barrier 4']
print('reverse scan, step 10, workgroup 0 (us since step start / delta):')
for i in range(1, 9):
  print(f'  {names[i]:24s} {(ts[i] - ts[0]) / 100:7.2f} {(ts[i] - ts[i - 1]) / 100:6.2f}')
print(f'  inside Q3: loads + gate derivatives {(ts[9] - ts[4]) / 100:.2f}, row sums + operand split {(ts[10] - ts[9]) / 100:.2f}, '
      f'contraction {(ts[11] - ts[10]) / 100:.2f}, epilogue {(ts[5] - ts[11]) / 100:.2f} us')
print('error word', int(L.scan_sync[1]))
