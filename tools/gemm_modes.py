"""Accuracy (vs float64) and speed of the contraction kernels per arithmetic mode
(dd_gemm_set_mode: 0 native fp32 MFMA, 6 split-bf16 x6, 3 split-bf16 x3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops

ops = hipops.HipOps('cuda:0', ws_bytes=1024 << 20)
g = torch.Generator(device='cuda').manual_seed(0)
MODES = tuple(int(x) for x in os.environ.get('GEMM_MODES', '0,6').split(','))


def timeit(fn, n=10):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


def relerr(x, ref):
  return float((x.double() - ref).abs().max() / ref.abs().max())


def gemm_case(M, N, K, ta, tb, check=True):
  A = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
  B = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
  C = torch.empty(M, N, device='cuda')
  ref = None
  if check:
    a = A.double().T if ta else A.double(); b = B.double().T if tb else B.double()
    ref = a @ b
  out = []
  for mode in MODES:
    ops.lib.dd_gemm_set_mode(mode)
    C.zero_(); ops.gemm(A, B, C, ta=ta, tb=tb)
    err = relerr(C, ref) if check else float('nan')
    ms = timeit(lambda: ops.gemm(A, B, C, ta=ta, tb=tb))
    out.append(f'm{mode}: {2.0*M*N*K/ms/1e9:6.1f} TF err {err:.1e}')
  print(f'gemm {M}x{N}x{K} ta{int(ta)} tb{int(tb)} | ' + ' | '.join(out), flush=True)


def conv_case(n, hb, cb, cs, k):
  hs = (hb - k) // 2 + 1
  big = torch.randn(n, hb, hb, cb, device='cuda', generator=g)
  small = torch.randn(n, hs, hs, cs, device='cuda', generator=g)
  w = torch.randn(k, k, cb, cs, device='cuda', generator=g) * 0.05
  bias = torch.randn(cs, device='cuda', generator=g)
  nchk = min(n, 8)
  wt = w.double().permute(3, 2, 0, 1)
  ref_down = torch.nn.functional.conv2d(big[:nchk].double().permute(0, 3, 1, 2), wt, bias.double(), stride=2).permute(0, 2, 3, 1)
  ref_up = torch.nn.functional.conv_transpose2d(small[:nchk].double().permute(0, 3, 1, 2), wt, None, stride=2).permute(0, 2, 3, 1)
  o_small = torch.empty_like(small); o_big = torch.empty_like(big); dw = torch.empty_like(w)
  fl = 2.0 * n * hs * hs * k * k * cb * cs
  for name, fn, chk in (
      ('down', lambda: ops.conv_down(big, w, bias, o_small, k), lambda: relerr(o_small[:nchk], ref_down)),
      ('up', lambda: ops.conv_up(small, w, None, o_big, k), lambda: relerr(o_big[:nchk], ref_up)),
      ('wgrad', lambda: ops.conv_wgrad(big, small, dw, k), None)):
    out = []
    ref_w = None
    for mode in MODES:
      ops.lib.dd_gemm_set_mode(mode)
      fn()
      if chk: err = chk()
      else:
        if ref_w is None: ref_w = dw.double().clone(); err = 0.0  # first mode as the reference
        else: err = relerr(dw, ref_w)
      ms = timeit(fn, 5)
      out.append(f'm{mode}: {fl/ms/1e9:6.1f} TF err {err:.1e}')
    print(f'conv_{name} n{n} {hb}x{cb}<->{hs}x{cs} k{k} | ' + ' | '.join(out), flush=True)


if __name__ == '__main__':
  quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
  for ta in (False, True):
    for tb in (False, True):
      gemm_case(1000, 520, 777, ta, tb)  # ragged: slow loaders, K tail
      if not quick: gemm_case(300, 260, 1999, ta, tb)
  for ta in (False, True):
    for tb in (False, True):
      gemm_case(4096, 4096, 4096, ta, tb)
  gemm_case(40000, 512, 1280, False, False)
  gemm_case(40000, 512, 512, False, False)
  gemm_case(40000, 1280, 512, False, True)
  gemm_case(1280, 512, 40000, True, False)
  gemm_case(2500, 512, 1280, False, False)
  gemm_case(2500, 256, 1040, False, False)
  conv_case(2500, 30, 64, 128, 4)
  conv_case(2500, 14, 128, 256, 4)
  conv_case(2500, 13, 128, 256, 5)
  conv_case(2500, 30, 64, 128, 6)
