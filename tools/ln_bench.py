"""Bandwidth of the LayerNorm(+ELU) kernels at the step's big shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0', ws_bytes=1024 << 20)

def timeit(fn, n=10):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3

for rows, C in ((2500 * 31 * 31, 64), (2500 * 30 * 30, 64), (2500 * 14 * 14, 128), (2500 * 13 * 13, 128),
                (2500 * 6 * 6, 256), (40000, 512), (2500, 512), (2500, 256)):
  z = torch.randn(rows, C, device='cuda'); out = torch.empty_like(z); dout = torch.randn_like(z); dz = torch.empty_like(z)
  stats = torch.empty(rows, 2, device='cuda')
  g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda')
  dg = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda'); dbp = torch.zeros(C, device='cuda')
  f = timeit(lambda: ops.ln_act_fwd(z, g, b, out, stats, True))
  bw = timeit(lambda: ops.ln_act_bwd(dout, z, out, stats, g, dz, dg, db, False, True, dbp))
  bn = timeit(lambda: ops.ln_act_bwd(dout, z, out, stats, g, dz, None, None, False, True))
  # the activation recomputed from z instead of read from `out` (three tensors instead of four)
  rw = timeit(lambda: ops.ln_act_bwd(dout, z, out, stats, g, dz, dg, db, False, True, dbp, beta=b))
  rn = timeit(lambda: ops.ln_act_bwd(dout, z, out, stats, g, dz, None, None, False, True, beta=b))
  gb = rows * C * 4 / 1e9
  print(f'rows {rows:8d} C {C:4d} | fwd {f:8.1f} us {2*gb/f*1e3:6.2f} TB/s | bwd+params {bw:8.1f} us {4*gb/bw*1e3:6.2f} TB/s | bwd {bn:8.1f} us {4*gb/bn*1e3:6.2f} TB/s'
        f' | recompute: bwd+params {rw:8.1f} us {3*gb/rw*1e3:6.2f} TB/s | bwd {rn:8.1f} us {3*gb/rn*1e3:6.2f} TB/s', flush=True)
