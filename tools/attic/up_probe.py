"""Where the banded transposed conv stands against plain contractions of the same extent."""
import sys
sys.path.insert(0, '.')
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
dev = 'cuda:0'
def bench(fn, flops, label, reps=5):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  print(f'{label:56s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TF', flush=True)
def gemm(M, N, K, ta, tb):
  A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
  C = torch.zeros(M, N, device=dev)
  bench(lambda: ops.gemm(A, B, C, bool(ta), bool(tb), 1.0, 0.0), 2.0*M*N*K, f'gemm {M}x{N}x{K} ta{ta} tb{tb}')
def up(n, hs, Cs, hb, Cb, k):
  big = torch.zeros(n, hb, hb, Cb, device=dev); small = torch.randn(n, hs, hs, Cs, device=dev); w = torch.randn(k, k, Cb, Cs, device=dev)
  fl = 2.0*n*hs*hs*k*k*Cb*Cs
  bench(lambda: ops.conv_up(small, w, None, big, k), fl, f'conv_up n{n} {hs}x{Cs}->{hb}x{Cb} k{k}')
# the interior band of the 13x128->30x64 k6 layer as a plain contraction
gemm(2500 * 121, 256, 1152, 0, 1)
gemm(2500 * 121, 256, 1152, 0, 0)
gemm(2500 * 225, 256, 864, 0, 1)
up(2500, 13, 128, 30, 64, 6)
up(150, 61, 128, 126, 64, 6)     # edges negligible: 59^2 / 63^2 interior
up(2500, 14, 128, 31, 64, 4)
up(150, 62, 128, 127, 64, 4)
gemm(2500 * 196, 256, 512, 0, 1)
