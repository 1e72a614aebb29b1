"""Keeps the GPU busy with foreign kernels for SECS seconds (contention for tools/dp_repro.py)."""
import os, time, torch
x = torch.randn(int(os.environ.get('HOG_N', 2048)), int(os.environ.get('HOG_N', 2048)), device='cuda')
t0 = time.time()
while time.time() - t0 < float(os.environ.get('SECS', 60)):
  for _ in range(50):
    y = x @ x
  torch.cuda.synchronize()
