"""Time the plain GEMM entry point of one experiment library (tools/exp64.sh) on the
128-row-tile shapes of the step (ablation builds: no split arithmetic / no MFMA).  usage: python tools/exp64.py <lib.so> <label>"""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
P, L, I, F, Z = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
lib.dd_gemm_f32.argtypes = [P, P, P, I, I, I, L, L, L, I, I, F, F, P, P, Z, P, P]
ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
out = []
for (M, N, K, ta, tb) in [(40000, 512, 512, 0, 0), (40000, 512, 512, 0, 1), (40000, 1280, 512, 0, 1), (512, 512, 40000, 1, 0), (1280, 512, 40000, 1, 0), (512, 512, 40000, 1, 1), (4096, 4096, 4096, 0, 0)]:
  A = torch.randn((K, M) if ta else (M, K), device='cuda'); B = torch.randn((N, K) if tb else (K, N), device='cuda')
  C = torch.zeros(M, N, device='cuda')
  flag = ctypes.c_int(0)
  def run():
    lib.dd_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, (M if ta else K), (K if tb else N), N, ta, tb, 1.0, 0.0,
                    None, ws.data_ptr(), ws.numel(), ctypes.byref(flag), st)
  for _ in range(5): run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): run()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 10 * 1e3
  ref = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double())
  err = float((C.double() - ref).abs().max() / ref.abs().max()) if flag.value == 0 else -1.0
  out.append(f'{M}x{N}x{K}{"T" if ta else "N"}{"T" if tb else "N"}: {us:6.1f}us (S{flag.value} err {err:.1e})')
print(f'{sys.argv[2]:10s} ' + ' | '.join(out))
