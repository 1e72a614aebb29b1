"""Time the step's mid-size GEMM shapes under the current DD_FORCE_TILE (env) setting."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0', ws_bytes=512 << 20)
SHAPES = [(2500, 512, 512, 0, 0), (2500, 512, 512, 0, 1), (2500, 256, 256, 0, 0), (2500, 256, 256, 0, 1),
          (2500, 256, 768, 0, 1), (2500, 768, 256, 0, 0), (2500, 512, 1280, 0, 0), (2500, 1280, 512, 0, 1),
          (2500, 1040, 256, 0, 1), (2500, 256, 1040, 0, 0), (2500, 256, 1024, 0, 1), (2500, 1024, 256, 0, 0),
          (2500, 16, 512, 0, 0), (2500, 512, 16, 0, 1), (512, 512, 2500, 1, 0), (256, 256, 2500, 1, 0),
          (1280, 512, 2500, 1, 0)]
def t(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
out = []
for M, N, K, ta, tb in SHAPES:
  A = torch.randn((K, M) if ta else (M, K), device='cuda'); B = torch.randn((N, K) if tb else (K, N), device='cuda')
  C = torch.empty(M, N, device='cuda')
  out.append(t(lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb))))
print(os.environ.get('DD_FORCE_TILE', 'default'), ' '.join(f'{x:6.1f}' for x in out))
