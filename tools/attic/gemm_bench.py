"""Micro-benchmark of the contraction kernels at the step's dominant shapes."""
import sys
sys.path.insert(0, '.')
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
dev = 'cuda:0'
def bench(fn, flops, label, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  print(f'{label:48s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TF')
def gemm(M, N, K, ta, tb, beta=0.0):
  A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
  C = torch.zeros(M, N, device=dev)
  bench(lambda: ops.gemm(A, B, C, bool(ta), bool(tb), 1.0, beta), 2.0*M*N*K, f'gemm {M}x{N}x{K} ta{ta} tb{tb} beta{beta}')
gemm(40000, 1280, 512, 0, 1, 1.0)
gemm(2500, 1280, 6400, 0, 0, 1.0)
for shp in [(40000,512,512,0,0),(40000,512,512,0,1),(40000,512,1280,0,0),(40000,1280,512,0,1),(512,512,40000,1,0),(1280,512,40000,1,0),(2500,512,512,0,0),(2500,256,256,0,0),(2500,512,1280,0,0),(2500,768,512,0,0),(2500,256,1040,0,0),(2500,1024,256,0,0),(2500,256,768,0,1),(2500,1040,256,0,1),(2500,1280,6400,0,0),(4096,4096,4096,0,0),(50,256,256,0,0),(50,256,1040,0,0)]:
  gemm(*shp)
def conv(n, hb, Cb, hs, Cs, k):
  big = torch.randn(n, hb, hb, Cb, device=dev); small = torch.randn(n, hs, hs, Cs, device=dev); w = torch.randn(k, k, Cb, Cs, device=dev)
  fl = 2.0*n*hs*hs*k*k*Cb*Cs
  bench(lambda: ops.conv_down(big, w, None, small, k), fl, f'conv_down n{n} {hb}x{Cb}->{hs}x{Cs} k{k}', 5)
  bench(lambda: ops.conv_up(small, w, None, big, k), fl, f'conv_up   n{n} {hs}x{Cs}->{hb}x{Cb} k{k}', 5)
  bench(lambda: ops.conv_wgrad(big, small, w, k), fl, f'conv_wgrad n{n} k{k}', 5)
for shp in [(2500,30,64,13,128,6),(2500,13,128,5,256,5),(2500,31,64,14,128,4),(2500,14,128,6,256,4),(2500,6,256,2,512,4),(2500,64,3,30,64,6)]:
  conv(*shp)
