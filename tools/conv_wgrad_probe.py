import sys, os
sys.path.insert(0, '.')
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
n = 2500
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps
small = torch.randn(n, 30, 30, 64, device='cuda')
dw6 = torch.zeros(6, 6, 3, 64, device='cuda')
dz = torch.randn(n, 64, 64, 3, device='cuda')
a = timeit(lambda: ops.conv_wgrad(dz, small, dw6, 6))
img = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, device='cuda')
s31 = torch.randn(n, 31, 31, 64, device='cuda')
dw4 = torch.zeros(4, 4, 3, 64, device='cuda')
b = timeit(lambda: ops.conv_wgrad(img, s31, dw4, 4, 1.0 / 255.0))
print(f'DBG={os.environ.get("DD_IMG_DBG","0")}: f32 k6 {a*1e3:.1f} us   u8 k4 {b*1e3:.1f} us')
# ---- encoder first layer: LayerNorm backward + filter gradient as two launches vs the fused pass
z = torch.randn(n, 31, 31, 64, device='cuda')
zz = z.view(-1, 64)
stats = torch.stack([zz.mean(1), (zz.var(1, unbiased=False) + 1e-3).rsqrt()], 1).contiguous()
gamma, beta = torch.ones(64, device='cuda'), torch.zeros(64, device='cuda')
dzb = torch.empty_like(z)
dg, db, dbias = (torch.zeros(64, device='cuda') for _ in range(3))
def two():
  ops.ln_act_bwd(s31.view(-1, 64), zz, None, stats, gamma, dzb.view(-1, 64), dg, db, False, True, dbias, beta=beta)
  ops.conv_wgrad(img, dzb, dw4, 4, 1.0 / 255.0)
c = timeit(two)
d = timeit(lambda: ops.conv_wgrad_ln(img, s31, z, stats, gamma, beta, dzb, dw4, dg, db, dbias, 4, 1.0 / 255.0))
print(f'encoder layer 1 backward: ln_act_bwd + conv_wgrad {c*1e3:.1f} us   fused {d*1e3:.1f} us')
# ---- encoder first layer forward: conv_down + ln_act_fwd vs the fused pass
w4 = torch.randn(4, 4, 3, 64, device='cuda') * 0.1
b4 = torch.zeros(64, device='cuda')
zf, of = torch.empty_like(z), torch.empty_like(z)
def two_f():
  ops.conv_down(img, w4, b4, zf, 4, 1.0 / 255.0)
  ops.ln_act_fwd(zf.view(-1, 64), gamma, beta, of.view(-1, 64), stats, True)
e = timeit(two_f)
f = timeit(lambda: ops.conv_down_ln(img, w4, b4, gamma, beta, zf, of, stats, 4, 1.0 / 255.0))
print(f'encoder layer 1 forward: conv_down + ln_act_fwd {e*1e3:.1f} us   fused {f*1e3:.1f} us')
# ---- decoder: data gradient of the image layer + LayerNorm backward of the layer in front of it
z30 = torch.randn(n, 30, 30, 64, device='cuda')
zz30 = z30.view(-1, 64)
st30 = torch.stack([zz30.mean(1), (zz30.var(1, unbiased=False) + 1e-3).rsqrt()], 1).contiguous()
do30, dz30 = torch.empty_like(z30), torch.empty_like(z30)
w6 = torch.randn(6, 6, 3, 64, device='cuda') * 0.1
def two_d():
  ops.conv_down(dz, w6, None, do30, 6)
  ops.ln_act_bwd(do30.view(-1, 64), zz30, None, st30, gamma, dz30.view(-1, 64), dg, db, False, True, dbias, beta=beta)
g_ = timeit(two_d)
h_ = timeit(lambda: ops.conv_down_lnbwd(dz, w6, z30, st30, gamma, beta, do30, dz30, dg, db, dbias, 6))
print(f'decoder image layer data gradient + LayerNorm backward: two launches {g_*1e3:.1f} us   fused {h_*1e3:.1f} us')
