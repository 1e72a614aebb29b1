"""Steady-state A/B of the product loop against the role-separated loop: blocks of back-to-back
launches (the chip boosts for the first few ms of a busy period and then settles at its
power-limited clock, so short timing loops measure the boost), per-launch HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 120
def block(fn, n):
  evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
  evs[0].record()
  for i in range(n):
    fn(); evs[i + 1].record()
  torch.cuda.synchronize()
  return [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]
for (M, N, K, zero) in [(4096, 4096, 4096, False), (4096, 4096, 4096, True), (422500, 128, 2304, False),
                        (40000, 512, 1280, False), (40000, 512, 512, False)]:
  A = torch.zeros(M, K, device='cuda') if zero else torch.randn(M, K, device='cuda')
  B = torch.zeros(K, N, device='cuda') if zero else torch.randn(K, N, device='cuda')
  C = torch.empty(M, N, device='cuda')
  fl = 2e-6 * M * N * K
  line = []
  for rep in range(2):
    for name, on in (('product', 0), ('roles', 1)):
      ops.lib.dd_gemm_set_ws(on, 256, 256)
      d = block(lambda: ops.gemm(A, B, C), NB)
      head, tail = sum(d[:3]) / 3, sum(d[-NB // 2:]) / (NB // 2)
      line.append(f'{name} first3 {head:7.1f} us ({fl / head:5.1f} TF) steady {tail:7.1f} us ({fl / tail:5.1f} TF)')
  print(f'{M}x{N}x{K}{" zeros" if zero else ""}:\n  ' + '\n  '.join(line), flush=True)
ops.lib.dd_gemm_set_ws(1, 1024, 1024)
