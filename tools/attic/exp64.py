"""Time the plain GEMM entry point of one experiment library (tools/exp64.sh) on the
64x64-tile shapes of the step.  usage: python tools/exp64.py <lib.so> <label>"""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
P, L, I, F, Z = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
lib.dd_gemm_f32.argtypes = [P, P, P, I, I, I, L, L, L, I, I, F, F, P, P, Z, P, P]
ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
out = []
for (M, N, K, tb) in [(2500, 512, 512, 0), (2500, 256, 1040, 0), (2500, 768, 512, 0), (2500, 256, 256, 0),
                      (2500, 1024, 256, 0), (2500, 256, 768, 1), (50, 256, 1040, 0), (50, 768, 512, 0)]:
  A = torch.randn(M, K, device='cuda'); B = torch.randn((N, K) if tb else (K, N), device='cuda')
  C = torch.zeros(M, N, device='cuda')
  flag = ctypes.c_int(0)
  def run():
    lib.dd_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, (K if tb else N), N, 0, tb, 1.0, 0.0,
                    None, ws.data_ptr(), ws.numel(), ctypes.byref(flag), st)
  for _ in range(5): run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50): run()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 50 * 1e3
  ref = A.double() @ (B.double().T if tb else B.double())
  err = float((C.double() - ref).abs().max() / ref.abs().max()) if flag.value == 0 else -1.0
  out.append(f'{M}x{N}x{K}{"T" if tb else ""}: {us:6.1f}us (S{flag.value} err {err:.1e})')
print(f'{sys.argv[2]:10s} ' + ' | '.join(out))
