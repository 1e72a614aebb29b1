"""Image-side stride-2 convolution (k_conv_image_down): time of the two call sites of configs[1]
(decoder image layer's data gradient: float 64x64x3 -> 30x30x64, k 6; encoder first layer: uint8
64x64x3 -> 31x31x64, k 4) and their ablations (DD_IMG_DBG bits: 2 no MFMAs, 4 no stores, 8 no
staging), in TB/s of the algorithmic bytes."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'all':
  for dbg in (0, 2, 4, 8, 6, 12, 14):
    subprocess.run([sys.executable, __file__], env=dict(os.environ, DD_IMG_DBG=str(dbg)))
  sys.exit(0)
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
n = 2500
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=20):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps
dz = torch.randn(n, 64, 64, 3, device='cuda')
w6 = torch.randn(6, 6, 3, 64, device='cuda') * 0.1
d30 = torch.empty(n, 30, 30, 64, device='cuda')
ms6 = timeit(lambda: ops.conv_down(dz, w6, None, d30, 6))
img = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, device='cuda')
w4 = torch.randn(4, 4, 3, 64, device='cuda') * 0.1
b4 = torch.randn(64, device='cuda')
z31 = torch.empty(n, 31, 31, 64, device='cuda')
ms4 = timeit(lambda: ops.conv_down(img, w4, b4, z31, 4, 1.0 / 255.0))
b6, b4_ = (dz.numel() + d30.numel()) * 4, img.numel() + z31.numel() * 4
print(f'DD_IMG_DBG={os.environ.get("DD_IMG_DBG", "0"):>2}: float k6 {ms6 * 1e3:6.1f} us ({b6 / ms6 / 1e9:.2f} TB/s)   '
      f'uint8 k4 {ms4 * 1e3:6.1f} us ({b4_ / ms4 / 1e9:.2f} TB/s)')
