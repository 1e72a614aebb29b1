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

# ---- image-side filter gradients (decoder last layer k6, encoder first layer k4 on the uint8 image)
def timeit(fn, reps=10):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps
dw6 = torch.zeros(6, 6, 3, 64, device='cuda')
dz = torch.randn(n, 64, 64, 3, device='cuda')
ms = timeit(lambda: ops.conv_wgrad(dz, small, dw6, 6))
print(f'conv_wgrad 64x3,30x64 k6: {ms * 1e3:.1f} us')
img = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, device='cuda')
s31 = torch.randn(n, 31, 31, 64, device='cuda')
dw4 = torch.zeros(4, 4, 3, 64, device='cuda')
ms = timeit(lambda: ops.conv_wgrad(img, s31, dw4, 4, 1.0 / 255.0))
print(f'conv_wgrad u8 64x3,31x64 k4: {ms * 1e3:.1f} us')
# ---- image-side conv_down: decoder image layer's data gradient (k6) and encoder first layer (k4, uint8)
w6 = torch.randn(6, 6, 3, 64, device='cuda') * 0.1
d30 = torch.empty(n, 30, 30, 64, device='cuda')
ms = timeit(lambda: ops.conv_down(dz, w6, None, d30, 6))
print(f'DD_DOWN_IMAGE={os.environ.get("DD_DOWN_IMAGE", "1")} conv_down 64x3->30x64 k6: {ms * 1e3:.1f} us')
w4 = torch.randn(4, 4, 3, 64, device='cuda') * 0.1
b4 = torch.randn(64, device='cuda')
z31 = torch.empty(n, 31, 31, 64, device='cuda')
ms = timeit(lambda: ops.conv_down(img, w4, b4, z31, 4, 1.0 / 255.0))
print(f'DD_DOWN_IMAGE={os.environ.get("DD_DOWN_IMAGE", "1")} conv_down u8 64x3->31x64 k4: {ms * 1e3:.1f} us')
