import sys, os, time
sys.path.insert(0, '.')
import torch
import bench
thr = int(sys.argv[1]); fb = int(sys.argv[2])
cfg = bench.make_config('a1_vision')
t = time.time()
print(thr, fb, bench.cpu_baseline(cfg, frac_batch=fb, threads=thr), 'total', time.time() - t, flush=True)
