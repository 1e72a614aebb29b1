"""Tile / split-K sweep of the small-row contractions of the step (2500 rows at configs[1]:
world-model heads on the posterior, 36 launches per step; 1250 rows x 512 at the xarm shard):
times each shape under the selector's environment hooks, one subprocess per variant.
  python tools/gemm_small_rows.py            (sweep)
  python tools/gemm_small_rows.py one        (this process's environment)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [
    ('default', {}),
    ('64x64', dict(DD_FORCE_TILE='64x64')),
    ('128x64', dict(DD_FORCE_TILE='128x64', DD_SPLIT_MIN_TILES='1')),
    ('128x64 split', dict(DD_FORCE_TILE='128x64', DD_SPLIT_MIN_TILES='192')),
    ('128x64 split t640', dict(DD_FORCE_TILE='128x64', DD_SPLIT_MIN_TILES='192', DD_SPLIT_TARGET='640', DD_SPLIT_FLOOR='0')),
    ('64x64 split2', dict(DD_FORCE_TILE='64x64', DD_SPLIT_MIN_TILES='400', DD_SPLIT_TARGET='640', DD_SPLIT_FLOOR='0')),
    ('128x128 split', dict(DD_FORCE_TILE='128x128', DD_SPLIT_MIN_TILES='192')),
]
if len(sys.argv) == 1:
  for name, env in VARIANTS:
    print(f'== {name} {env}', flush=True)
    subprocess.run([sys.executable, __file__, 'one'], env=dict(os.environ, **env))
  sys.exit(0)
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=30):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / reps
SHAPES = [(2500, 512, 512, 0, 0), (2500, 512, 512, 0, 1), (512, 512, 2500, 1, 0), (2500, 512, 1280, 0, 0),
          (2500, 1280, 512, 0, 1), (1280, 512, 2500, 1, 0), (1250, 512, 512, 0, 0), (1250, 512, 512, 0, 1),
          (512, 512, 1250, 1, 0), (2500, 256, 256, 0, 0)]
out = []
for M, N, K, ta, tb in SHAPES:
  A = torch.randn(*((K, M) if ta else (M, K)), device='cuda')
  B = torch.randn(*((N, K) if tb else (K, N)), device='cuda')
  C = torch.empty(M, N, device='cuda')
  gamma, beta = torch.ones(N, device='cuda'), torch.zeros(N, device='cuda')
  o, st = torch.empty(M, N, device='cuda'), torch.empty(M, 2, device='cuda')
  us = timeit(lambda: ops.gemm(A, B, C, bool(ta), bool(tb)))
  # the forward form of a normed layer: deferred split-K sum taken by the LayerNorm kernel
  def fwd():
    pre = ops.gemm(A, B, C, bool(ta), bool(tb), defer=True)
    ops.ln_act_fwd(C, gamma, beta, o, st, True, pre=pre)
  us2 = timeit(fwd) if not ta else float('nan')
  out.append(f'{M}x{N}x{K} ta{ta} tb{tb}: {us:6.1f} us {2e-6 * M * N * K / us:6.1f} TF | +LN deferred {us2:6.1f} us')
print('\n'.join(out))
