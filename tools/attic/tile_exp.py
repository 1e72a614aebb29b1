import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0', ws_bytes=256 << 20)
dev = 'cuda:0'
for (M, N, K, tb) in [(2500,512,512,0),(2500,512,1280,0),(2500,256,256,0),(2500,768,256,0),(2500,256,768,1),(2500,1024,256,0),(2500,256,1040,0),(2500,1040,256,1),(2500,16,512,0)]:
  A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.empty(M, N, device=dev)
  for _ in range(3): ops.gemm(A, B, C, False, bool(tb))
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50): ops.gemm(A, B, C, False, bool(tb))
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 50
  print(f'{os.environ.get("DD_FORCE_TILE","auto"):8s} {M}x{N}x{K} tb{tb}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF')
