"""Time the plain GEMM entry point of the 256x128-tile experiment library (tools/exp_big.sh) on
the large shapes of the step, under the DD_FORCE_TILE of the environment.
usage: DD_FORCE_TILE=256x128|128x128 python tools/exp_big.py tools/exp_big_libs/lib_big.so"""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
P, L, I, F, Z = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
lib.dd_gemm_f32.argtypes = [P, P, P, I, I, I, L, L, L, I, I, F, F, P, P, Z, P, P]
ws = torch.empty(1024 << 20, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
out = []
for (M, N, K, ta, tb) in [(4096, 4096, 4096, 0, 0), (4096, 4096, 4096, 0, 1), (40000, 512, 512, 0, 0),
                          (40000, 512, 512, 0, 1), (40000, 512, 1280, 0, 0), (40000, 1280, 512, 0, 1),
                          (512, 512, 40000, 1, 0), (1280, 512, 40000, 1, 0), (2500, 1024, 1280, 0, 0),
                          (2500, 16384, 1280, 0, 0), (40000, 128, 512, 0, 0), (160000, 256, 1024, 0, 0)]:
  A = torch.randn((K, M) if ta else (M, K), device='cuda')
  B = torch.randn((N, K) if tb else (K, N), device='cuda')
  C = torch.zeros(M, N, device='cuda')
  def run():
    lib.dd_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, (M if ta else K), (K if tb else N), N,
                    ta, tb, 1.0, 0.0, None, ws.data_ptr(), ws.numel(), None, st)
  for _ in range(3): run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): run()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 10 * 1e3
  a, b = (A.T if ta else A)[:64].double(), (B.T if tb else B).double()
  ref = a @ b
  err = float((C[:64].double() - ref).abs().max() / ref.abs().max())
  out.append(f'{M}x{N}x{K}{"T" if ta else "N"}{"T" if tb else "N"}: {us:7.1f}us {2e-6 * M * N * K / us:6.1f}TF (err {err:.1e})')
print(os.environ.get('DD_FORCE_TILE', 'default'))
print('\n'.join(out))
