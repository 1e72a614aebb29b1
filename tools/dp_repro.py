"""Reproducibility of data-parallel training under host-side perturbations (docs/LABLOG.md, end of
round 6): REPS agents in a row train the same 8 steps; from the second on, the host perturbs the
gaps between the calls; prints the number of distinct final (model loss, actor loss, gradient norm)
tuples - 1 when training is reproducible.
  torchrun --nproc-per-node 2 tools/dp_repro.py
Environment: CFG (a1_vision | xarm ...), BG global batch, REPS, KNOBS="hip.fused_scan=False ...",
PERTURB = each (random device syncs / sleeps, a different pattern per rank; default) | same | rank1 |
none | allsync (device sync after every step) | allsleep | sync | sleep | streamsync (null stream) |
plansync | kernel; SHOW=1 prints the outcomes; DETAIL=1 / PLANCHK=1 checksum the learner's
buffers after every step (null stream / plan stream) and print where a run first leaves run 0 -
PLANCHK also names the (batch row, time step) rows of the scan's outputs that differ;
NO_SIDE=1|2 drops the side launch contexts.  This is the tool that found the uncleared barrier
counters of the fused observe scan (hipMemsetAsync nodes in a captured graph; DD_SCAN_MEMSET=1
brings that form back: PERTURB=allsync then gives ~15 distinct outcomes in 16 runs)."""
import os, sys, pathlib, time, random
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
os.environ['LOCAL_RANK'] = '0'     # the ranks share the one GPU
import numpy as np, torch, torch.distributed as dist
from daydreamer_amd import agent as agent_mod, synthetic
import helpers
torch.cuda.set_device(0)
dist.init_process_group('gloo')
rank = dist.get_rank()
BG = int(os.environ.get('BG', 6))
name = os.environ.get('CFG', 'a1_vision')
cfg = helpers.make_config((name,), batch_size=BG, replay_chunk=8, imag_horizon=4)
for kv in os.environ.get('KNOBS', '').split():
  k, v = kv.split('=')
  cfg = cfg.update({k: {'True': True, 'False': False}.get(v, v)})
obs, act = synthetic.config_spaces(name)
batches = [synthetic.make_batch(obs, act, BG, 8, seed=s, smooth_images=True, terminals=0.1) for s in range(3)]
outs, hist, kept = [], [], []
KEEP = ('post', 'gin', 'xin', 'z3', 'post_logit', 'first')
CHECK2 = ('image', 'action', 'first', 'u_post', 'xin', 'gin', 'z3', 'post', 'post_logit', 'prior_logit', 'kl', 'dfeat', 'dpost_logit',
          'dz3', 'dgin', 'carry', 'loss_reward', 'loss_total', 'traj', 'dtraj', 'stat_sums', 'wmkl_scale')
junk = torch.zeros(16, device='cuda')
CHECK = ('image', 'action', 'u_post', 'u_prior', 'u_img', 'eps', 'post', 'gin', 'xin', 'z3', 'post_logit', 'prior_logit',
         'dfeat', 'kl', 'carry', 'loss_reward', 'loss_cont', 'traj', 'dtraj', 'dpost_logit', 'dprior_logit', 'dz3', 'dgin', 'loss_total')
for rep in range(int(os.environ.get('REPS', 12))):
  pert = os.environ.get('PERTURB', 'each')   # each: per-rank pattern; same: both ranks alike; rank1: only rank 1
  rng = random.Random((1000 * rank if pert != 'same' else 0) + rep)
  ag = agent_mod.Agent(obs, act, None, cfg.update({'hip.pipeline': False}))
  if os.environ.get('NO_SIDE'):
    ag.ops2 = None
    if os.environ['NO_SIDE'] == '2': ag.ops_b2 = None
  state = None
  steps = []
  chk = torch.zeros(8, 64, device='cuda') if os.environ.get('PLANCHK') else None
  keep = {} if os.environ.get('PLANCHK') else None
  step_i = [0]
  def hook_metrics(ag=ag, chk=chk, step_i=step_i, keep=keep):
    # checksums on the plan stream right after the replay, read once at the very end
    L = ag.learner
    if not hasattr(L, '_orig_read'):
      L._orig_read = L.read_metrics
      from daydreamer_amd import graphs
      ps = graphs.stream(torch.device('cuda', 0), 'plan')
      def rm(*a, **k):
        with torch.cuda.stream(ps):
          for key in KEEP:
            if key not in keep: keep[key] = torch.zeros((8,) + tuple(L.b[key].shape), device='cuda', dtype=L.b[key].dtype)
            keep[key][step_i[0]].copy_(L.b[key])
          for j, key in enumerate(CHECK2):
            t = L.b[key] if key in L.b else getattr(L, key)
            torch.sum(t.view(-1).float().abs() if t.dtype != torch.float32 else t.view(-1).abs(), dim=0, out=chk[step_i[0], j])
          for j, g in enumerate(L.groups.values()):
            torch.sum(g.flat.abs(), dim=0, out=chk[step_i[0], 40 + j])
            if hasattr(g, 'gflat'): torch.sum(g.gflat.abs(), dim=0, out=chk[step_i[0], 50 + j])
        return L._orig_read(*a, **k)
      L.read_metrics = rm
  for i in range(8):
    if chk is not None and ag.learner is not None: hook_metrics()
    step_i[0] = i
    _, state, m = ag.train(batches[i % 3], state)
    mm = {k: float(v) for k, v in m.items() if np.ndim(v) == 0} if os.environ.get('DETAIL') == '2' else {}
    if os.environ.get('DETAIL'):   # device-side checksums, no host synchronisation; read at the end
      L = ag.learner
      for gname, g in L.groups.items():
        mm['param_' + gname] = g.flat.double().abs().sum()
        if hasattr(g, 'gflat'): mm['grad_' + gname] = g.gflat.double().abs().sum()
      for key in CHECK:
        if key in L.b and isinstance(L.b[key], torch.Tensor):
          mm['buf_' + key] = L.b[key].double().abs().sum()
      mm['stat_sums'] = L.stat_sums.abs().sum()
    steps.append(mm)
    if rep and not (pert == 'rank1' and rank == 0) and pert != 'none':
      r = rng.random()
      if pert == 'allsync': torch.cuda.synchronize()
      elif pert == 'allsleep': time.sleep(0.02)
      elif pert == 'streamsync': torch.cuda.current_stream().synchronize()
      elif pert == 'plansync':
        from daydreamer_amd import graphs
        graphs.stream(torch.device('cuda', 0), 'plan').synchronize()
      elif pert == 'kernel': junk.add_(1)
      elif pert == 'sync':
        if r < 0.5: torch.cuda.synchronize()
      elif pert == 'sleep':
        if r < 0.5: time.sleep(rng.random() * 0.03)
      elif r < 0.4: torch.cuda.synchronize()
      elif r < 0.7: time.sleep(rng.random() * 0.03)
  outs.append((float(m['model_loss']), float(m['actor_loss']), float(m['model_grad_norm'])))
  if chk is not None:
    torch.cuda.synchronize()
    kept.append({k: v.cpu().numpy() for k, v in keep.items()})
    c = chk.cpu().numpy()
    names = list(CHECK2) + [''] * 64
    for j, gname in enumerate(ag.learner.groups): names[40 + j], names[50 + j] = 'param_' + gname, 'grad_' + gname
    steps = [{names[j]: float(c[i, j]) for j in range(64) if names[j]} for i in range(8)]
  hist.append([{k: float(v) for k, v in st.items()} for st in steps])
  del ag
if rank == 0:
  from collections import Counter
  c = Counter(outs)
  print(f"{name} KNOBS='{os.environ.get('KNOBS', '')}' {os.environ.get('NOTE', '')}: {len(c)} distinct outcome(s) in {len(outs)} runs: {sorted(c.values(), reverse=True)}", flush=True)
  if os.environ.get('SHOW'):
    for o, n in c.most_common():
      print('   ', n, ' '.join(f'{x:.9g}' for x in o), flush=True)
if os.environ.get('DETAIL') or os.environ.get('PLANCHK'):
  for rep, steps in enumerate(hist):
    for i, (a, b_) in enumerate(zip(hist[0], steps)):
      bad = [k for k in a if a[k] != b_[k] and not (a[k] != a[k] and b_[k] != b_[k])]
      if bad:
        if kept and bad[0] not in ('traj',):
          for key in KEEP:
            x0, x1 = kept[0][key][i].reshape(BG // dist.get_world_size(), 8, -1), kept[rep][key][i].reshape(BG // dist.get_world_size(), 8, -1)
            d = (x0 != x1)
            if d.any():
              bt = np.argwhere(d.any(-1))
              cols = np.flatnonzero(d.any((0, 1)))
              print(f'rank {rank} rep {rep} step {i} {key}: rows (b,t) differing {bt[:12].tolist()} ({len(bt)}), columns {cols[:6].tolist()}..{cols[-3:].tolist()} ({len(cols)} of {d.shape[-1]}), max abs diff {np.abs(x0 - x1).max():.3g}', flush=True)
        print(f'rank {rank} rep {rep}: first difference at step {i}: ' + ', '.join(f'{k} {a[k]:.9g}->{b_[k]:.9g}' for k in bad[:8]) + f' ({len(bad)} keys)', flush=True)
        break
dist.barrier(); dist.destroy_process_group()
