"""Wave-specialised contraction loop (tools/exp_ws/gemm_ws.hip) against the product loop in the same
process: time, TFLOP/s, and whether the results are bit-identical (same LDS image, same product
order).  Variant by environment: DD_WS_BK=16|32, DD_WS_OCC=1|2, DD_WS_TARGET (split-K slots).
usage: python tools/exp_ws.py tools/exp_ws/libs/lib_ws.so"""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
P, L, I, Z = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_size_t
for f in (lib.dd_gemm_ws, lib.dd_gemm_ref):
  f.argtypes = [P, P, P, I, I, I, L, L, L, I, I, P, Z, P]
lib.dd_last_error.restype = ctypes.c_char_p
ws = torch.empty(1024 << 20, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
print('variant', {k: v for k, v in os.environ.items() if k.startswith('DD_WS')})
for (M, N, K, ta, tb) in [(4096, 4096, 4096, 0, 0), (4096, 4096, 4096, 0, 1), (40000, 512, 512, 0, 0),
                          (40000, 512, 512, 0, 1), (40000, 1280, 512, 0, 1), (512, 512, 40000, 1, 0),
                          (2500, 1024, 1280, 0, 0), (160000, 256, 1024, 0, 0), (2500, 512, 512, 0, 0)]:
  A = torch.randn((K, M) if ta else (M, K), device='cuda')
  B = torch.randn((N, K) if tb else (K, N), device='cuda')
  out = {}
  for name, fn in (('ref', lib.dd_gemm_ref), ('ws', lib.dd_gemm_ws)):
    C = torch.zeros(M, N, device='cuda')
    def run():
      rc = fn(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, (M if ta else K), (K if tb else N), N, ta, tb,
              ws.data_ptr(), ws.numel(), st)
      assert rc == 0, lib.dd_last_error()
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    out[name] = (e0.elapsed_time(e1) / 10 * 1e3, C)
  a, b = (A.T if ta else A)[:64].double(), (B.T if tb else B).double()
  ref = a @ b
  err = float((out['ws'][1][:64].double() - ref).abs().max() / ref.abs().max())
  same = bool(torch.equal(out['ws'][1], out['ref'][1]))
  fl = 2e-6 * M * N * K
  print(f'{M}x{N}x{K}{"T" if ta else "N"}{"T" if tb else "N"}: product {out["ref"][0]:7.1f} us {fl / out["ref"][0]:6.1f} TF | '
        f'specialised {out["ws"][0]:7.1f} us {fl / out["ws"][0]:6.1f} TF | x{out["ref"][0] / out["ws"][0]:.2f} '
        f'err {err:.1e} bit-identical {same}')
