"""Round quantisation of the contraction launches of a step: for every traced call site
(profiles/r03_contraction_call_sites.txt, from tools/trace_shapes.py) the tile shape and split-K
factor `run_mat` / `pick_split` (csrc/gemm_core.h) choose, the number of workgroups, and how far
the launch is from its ideal on 512 slots (256 CUs x 2 resident workgroups; a remainder of at most
256 workgroups costs half a round, each alone on its CU).  CPU only.
usage: python tools/round_quantisation.py [call_sites.txt]"""
import re
import sys


def ceil(a, b):
  return -(-a // b)


def pick(M, N, K):
  TMS = 128 if M > 64 else 64
  TNS = 128 if (M > 64 and N > 64) else 64
  shallow = K <= 1536
  if shallow and TMS == 128 and TNS == 128 and ceil(M, 128) * ceil(N, 128) < 256:
    TNS = 64
  if shallow and TMS == 128 and TNS == 64 and ceil(M, 128) * ceil(N, 64) < (128 if K > 768 else 512):
    TMS = 64
  tiles = ceil(M, TMS) * ceil(N, TNS)
  S = 1
  if not (tiles >= 192 or K < 128 or (tiles >= 100 and K <= 384)):
    s = ceil(512, tiles)
    if s > 1 and s * tiles > 512:
      s = 512 // tiles
    S = max(1, min(s, K // 64))
    kps = ceil(ceil(K, S), 16) * 16
    S = ceil(K, kps)
  return TMS, TNS, tiles, S


def main():
  path = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r03_contraction_call_sites.txt'
  rows = []
  for line in open(path):
    m = re.search(r'([\d.]+) ms n=\s*(\d+)\s+([\d.]+) TF.*?us\s+(\w+) (.*)', line)
    if not m:
      continue
    ms, tf, kind, rest = float(m[1]), float(m[3]), m[4], m[5]
    if kind == 'gemm':
      M, N, K = map(int, re.match(r'(\d+)x(\d+)x(\d+)', rest).groups())
    elif kind == 'conv_down':
      n, hb, cb, hs, cs, k = map(int, re.match(r'n(\d+) (\d+)x(\d+)->(\d+)x(\d+) k(\d+)', rest).groups())
      M, N, K = n * hs * hs, cs, k * k * cb
    elif kind == 'conv_wgrad':
      n, hb, cb, hs, cs, k = map(int, re.match(r'n(\d+) (\d+)x(\d+),(\d+)x(\d+) k(\d+)', rest).groups())
      M, N, K = k * k * cb, cs, n * hs * hs
    else:
      continue   # (banded transposed conv, fused kernels: own launch shapes)
    TMS, TNS, tiles, S = pick(M, N, K)
    wgs = tiles * S
    full, rem = divmod(wgs, 512)
    rounds = full + (0 if rem == 0 else (0.5 if rem <= 256 else 1.0))
    rows.append((ms, f'{kind} {rest[:32]}', f'{TMS}x{TNS}', tiles, S, wgs, round(wgs / 512 / rounds, 2), tf))
  print('ms_per_step  call site                                  tile  tiles  S  workgroups  ideal/rounds  TFLOP/s')
  for r in sorted(rows, reverse=True):
    print(f'{r[0]:8.3f}  {r[1]:42s} {r[2]:8s} {r[3]:6d} {r[4]:3d} {r[5]:7d} {r[6]:8.2f} {r[7]:8.1f}')


if __name__ == '__main__':
  main()
