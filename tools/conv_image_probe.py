"""Time of the decoder image layer (dd_conv2d_s2_up, few output channels) at configs[1] size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
n, hs, Cs, hb, Cb, k = 2500, 30, 64, 64, 3, 6
small = torch.randn(n, hs, hs, Cs, device='cuda')
w = torch.randn(k, k, Cb, Cs, device='cuda') * 0.1
bias = torch.randn(Cb, device='cuda')
big = torch.empty(n, hb, hb, Cb, device='cuda')
for _ in range(3):
  ops.conv_up(small, w, bias, big, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
  ops.conv_up(small, w, bias, big, k)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
gb = (small.numel() + big.numel()) * 4 / 1e9
print(f'DD_IMG_DBG={os.environ.get("DD_IMG_DBG", "0")} DD_UP_IMAGE={os.environ.get("DD_UP_IMAGE", "1")}: {ms * 1e3:.1f} us  ({gb / ms:.2f} TB/s algorithmic)')
