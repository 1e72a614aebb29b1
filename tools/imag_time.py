"""Section times inside the fused imagination rollout (csrc/imag.hip): the kernel stamps the
100 MHz wall clock at its section boundaries (step 1, block 0); plus the launch's total time.
  python tools/imag_time.py [batch] [length] [horizon]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import helpers
from daydreamer_amd import learner as LM, hipops, synthetic, config as config_mod, spec as spec_mod

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
H = int(sys.argv[3]) if len(sys.argv) > 3 else 15
cfg = helpers.make_config(('a1_vision',), batch_size=B, replay_chunk=T, imag_horizon=H)
import os
if os.environ.get('DD_IMAG_ROWS'):   # rows per workgroup of the fused kernels (16 / 32)
  cfg = cfg.update({'hip.imag_rows': int(os.environ['DD_IMAG_ROWS'])})
plain = config_mod.to_plain(cfg)
obs, act = synthetic.config_spaces('a1_vision')
sp = spec_mod.build_spec(plain, {k: v.shape for k, v in obs.items()}, 16, False)
data = synthetic.make_batch(obs, act, B, T, seed=0)
ops = hipops.HipOps('cuda:0')
L = LM.Learner(sp, ops, 'cuda:0', B, T, params=spec_mod.init_params(sp, 0))
L.upload(data)
L.train_step_device(use_carry=False)
torch.cuda.synchronize()
L.imag_stamps = torch.zeros(32, dtype=torch.int64, device='cuda:0')
for W, planes, col0 in L.imag_planes.values():
  ops.imag_wprep(W, planes, col0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
  e0.record()
  L.imagine_rollout_fused()
  e1.record()
  torch.cuda.synchronize()
  print(f'launch (incl. weight prep) {e0.elapsed_time(e1):.3f} ms')
ts = L.imag_stamps.cpu().numpy()
names = ['start', 'A0 gather+operand', 'A0 gemm', 'A0 norm', 'A1 gemm', 'A1 norm', 'A2 gemm', 'A2 norm',
         'A3 gemm', 'A3 norm', 'head', 'action', 'img_in gather', 'img_in norm', 'gru gemm', 'gru gates',
         'out0 gemm', 'out0 norm', 'out1 gemm', 'out1 norm', 'out2 gemm', 'out2 norm',
         'stats0 gemm', 'stats0 draw', 'stats1 gemm', 'stats1 draw']
prev = ts[0]
for i, n in enumerate(names[1:], 1):
  if ts[i]:
    print(f'{n:22s} {(ts[i] - prev) / 100.0:8.2f} us')
    prev = ts[i]
print(f'step total {(ts[25] - ts[0]) / 100.0:.2f} us')

# ---- reverse pass
L.imag_stamps_b = torch.zeros(8, dtype=torch.int64, device='cuda:0')
for rep in range(3):
  e0.record()
  L.imagine_reverse_fused()
  e1.record()
  torch.cuda.synchronize()
  print(f'reverse launch (incl. weight prep) {e0.elapsed_time(e1):.3f} ms')
ts = L.imag_stamps_b.cpu().numpy()
for i, n in enumerate(['draw bwd + stats^T', 'img_out 2..0', 'gru', 'dh + img_in ln', 'img_in^T'], 1):
  print(f'{n:22s} {(ts[i] - ts[i - 1]) / 100.0:8.2f} us')
print(f'reverse step total {(ts[5] - ts[0]) / 100.0:.2f} us')
