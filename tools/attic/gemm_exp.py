import sys, os, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pathlib
from daydreamer_amd import hipops
libs = [str(hipops._LIB_PATH)] + sorted(glob.glob(str(hipops._LIB_PATH.parent / 'libdd_exp_*.so')))
M = N = K = 4096
A = torch.randn(M, K, device='cuda'); B = torch.randn(K, N, device='cuda'); C = torch.empty(M, N, device='cuda')
for lib in libs:
  hipops._lib = None; hipops._LIB_PATH = pathlib.Path(lib)
  ops = hipops.HipOps('cuda:0', ws_bytes=64 << 20)
  for _ in range(3): ops.gemm(A, B, C)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): ops.gemm(A, B, C)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 10
  print(f'{os.path.basename(lib):60s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF')
