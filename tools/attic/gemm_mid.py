"""Mid-size (2500-row) GEMM shapes of the imagination scans under the dispatcher's experiment
hooks (DD_FORCE_TILE, DD_SPLIT_MIN_TILES, DD_SPLIT_TARGET): python tools/gemm_mid.py <label>"""
import sys
sys.path.insert(0, '.')
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
out = []
for (M, N, K, tb) in [(2500, 512, 512, 0), (2500, 256, 1040, 0), (2500, 768, 512, 0), (2500, 256, 256, 0),
                      (2500, 1024, 256, 0), (2500, 512, 1280, 0), (2500, 256, 768, 1), (2500, 1040, 256, 1)]:
  A = torch.randn(M, K, device='cuda'); B = torch.randn((N, K) if tb else (K, N), device='cuda')
  C = torch.zeros(M, N, device='cuda')
  run = lambda: ops.gemm(A, B, C, False, bool(tb), defer=True)
  for _ in range(5): run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50): run()
  e1.record(); torch.cuda.synchronize()
  s = run()
  out.append(f'{N}x{K}{"T" if tb else ""}:{e0.elapsed_time(e1) / 50 * 1e3:5.1f}us/S{s.n if s else 1}')
print(f'{sys.argv[1]:22s} ' + ' '.join(out))
