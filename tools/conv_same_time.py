"""Time of the stride-1 SAME convolution family (csrc/conv_same.hip) at the residual encoder's
full-size layers (2500 images = batch 50 x seq 50): forward, data gradient, filter gradient,
2x2 pooling / repetition.  python tools/conv_same_time.py"""
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from daydreamer_amd import hipops  # noqa: E402


def timed(fn, reps=5):
  fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps


def main():
  ops = hipops.HipOps('cuda:0')
  n = 2500
  if True:   # (launches go to torch's current stream, where the events are recorded)
    for h, cin, cout in ((32, 64, 64), (16, 128, 128), (8, 256, 256), (4, 512, 512), (64, 3, 64)):
      x = torch.randn(n, h, h, cin, device='cuda')
      y = torch.randn(n, h, h, cout, device='cuda')
      w = torch.randn(3, 3, cin, cout, device='cuda') * 0.1
      dx, dw = torch.empty_like(x), torch.empty_like(w)
      fl = 2.0 * n * h * h * 9 * cin * cout
      t1 = timed(lambda: ops.conv_same(x, w, None, y, 3))
      t2 = timed(lambda: ops.conv_same_bwd(y, w, dx, 3))
      t3 = timed(lambda: ops.conv_same_wgrad(x, y, dw, 3))
      print(f'conv_same n{n} {h}x{h} {cin}->{cout} k3: fwd {t1:.3f} ms ({fl / t1 * 1e-9:.1f} TFLOP/s) '
            f'bwd-data {t2:.3f} ms ({fl / t2 * 1e-9:.1f}) wgrad {t3:.3f} ms ({fl / t3 * 1e-9:.1f})', flush=True)
    x = torch.randn(n, 64, 64, 64, device='cuda')
    y = torch.empty(n, 32, 32, 64, device='cuda')
    t = timed(lambda: ops.pool2(x, y, 0.25))
    print(f'pool2 n{n} 64x64x64: {t:.3f} ms ({(x.numel() + y.numel()) * 4 / t * 1e-9:.2f} TB/s)')
    t = timed(lambda: ops.repeat2(y, x, 1.0))
    print(f'repeat2 n{n} 32x32x64: {t:.3f} ms ({(x.numel() + y.numel()) * 4 / t * 1e-9:.2f} TB/s)')


if __name__ == '__main__':
  main()
