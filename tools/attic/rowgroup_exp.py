"""Does splitting the 2500 imagined trajectories into independent row groups on
separate HIP streams speed up a latency-bound layer chain?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops
dev = 'cuda:0'
N, F, U = 2500, 1280, 512
x = torch.randn(N, F, device=dev)
Ws = [torch.randn(F, U, device=dev) * 0.03] + [torch.randn(U, U, device=dev) * 0.05 for _ in range(3)]
gam, bet = torch.ones(U, device=dev), torch.zeros(U, device=dev)
zs = [torch.empty(N, U, device=dev) for _ in range(4)]
outs = [torch.empty(N, U, device=dev) for _ in range(4)]
st = torch.empty(N, 2, device=dev)
def chain(ops, r0, r1):
  h = x[r0:r1]
  for i in range(4):
    ops.gemm(h, Ws[i], zs[i][r0:r1])
    ops.ln_act_fwd(zs[i][r0:r1], gam, bet, outs[i][r0:r1], st[r0:r1], True)
    h = outs[i][r0:r1]
for G in (1, 2, 4):
  opss = [hipops.HipOps(dev, ws_bytes=64 << 20) for _ in range(G)]
  streams = [torch.cuda.Stream() for _ in range(G)]
  per = N // G
  def run():
    cur = torch.cuda.current_stream()
    for g in range(G):
      streams[g].wait_stream(cur)
      with torch.cuda.stream(streams[g]):
        chain(opss[g], g * per, (g + 1) * per if g < G - 1 else N)
    for g in range(G):
      cur.wait_stream(streams[g])
  # capture into a graph to remove host launch effects
  run(); torch.cuda.synchronize()
  gr = torch.cuda.CUDAGraph()
  cap = torch.cuda.Stream()
  cap.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(cap):
    gr.capture_begin(); 
    for _ in range(15): run()
    gr.capture_end()
  torch.cuda.current_stream().wait_stream(cap)
  for _ in range(3): gr.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): gr.replay()
  e1.record(); torch.cuda.synchronize()
  print(f'groups {G}: {e0.elapsed_time(e1) / 10 / 15 * 1e3:8.1f} us per 4-layer chain')
