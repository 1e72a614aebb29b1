"""Ablations of the wave-specialised loop (DD_WS_ABL, timing only: results are wrong by design):
1 = staging waves idle in the loop, 2 = fragments read once, 3 = both (MFMAs + barriers only),
4 = no MFMAs (staging + fragment reads), 5 = neither staging nor MFMAs (fragment reads + barriers), 8 = B staged without the split arithmetic (what
pre-split weight planes would cost), 24 = A and B staged without it.
usage: python tools/exp_ws_abl.py tools/exp_ws/libs/lib_ws.so   (runs all variants in subprocesses)"""
import ctypes, os, subprocess, sys
if len(sys.argv) > 2:
  import torch
  lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
  P, L, I, Z = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_size_t
  lib.dd_gemm_ws.argtypes = [P, P, P, I, I, I, L, L, L, I, I, P, Z, P]
  ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
  st = torch.cuda.current_stream().cuda_stream
  out = []
  for (M, N, K) in [(4096, 4096, 4096), (40000, 512, 512)]:
    A = torch.randn(M, K, device='cuda'); B = torch.randn(K, N, device='cuda'); C = torch.zeros(M, N, device='cuda')
    run = lambda: lib.dd_gemm_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, N, N, 0, 0, ws.data_ptr(), ws.numel(), st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tiles_per_cu = -(-((M + 127) // 128) * ((N + 127) // 128) // 256)
    out.append(f'{M}x{N}x{K}: {us:7.1f} us = {us / (tiles_per_cu * K / 16) * 1e3:6.0f} ns per k-step')
  print(f'ABL={os.environ.get("DD_WS_ABL", "-"):2s} ' + ' | '.join(out))
else:
  for abl in (None, '1', '2', '3', '4', '5', '8', '24'):
    env = dict(os.environ, DD_WS_BK='32')
    if abl: env['DD_WS_ABL'] = abl
    subprocess.run([sys.executable, __file__, sys.argv[1], 'child'], env=env)
