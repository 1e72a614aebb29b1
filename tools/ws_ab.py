"""A/B of the role-separated contraction loop per call site of the C2 train step: runs
tools/trace_shapes.py (one traced eager step, HIP events around every contraction launch) in
separate processes under different selector settings and prints the per-label times side by side."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [
    ('off', dict(DD_WS='0')),
    ('k1024', dict(DD_WS='1', DD_WS_KMIN='1024', DD_WS_KMIN_TC='1024')),
    ('k512', dict(DD_WS='1', DD_WS_KMIN='512', DD_WS_KMIN_TC='512')),
    ('k2048', dict(DD_WS='1', DD_WS_KMIN='2048', DD_WS_KMIN_TC='2048')),
]
if len(sys.argv) > 1:
  VARIANTS = [v for v in VARIANTS if v[0] in sys.argv[1:]]
res, tot = {}, {}
for name, env in VARIANTS:
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_shapes.py'), '400'],
                       env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT).stdout
  for line in out.splitlines():
    m = re.match(r'\s*([\d.]+) ms n=\s*(\d+)\s+([\d.]+) TF\s+avg\s+([\d.]+) us\s+(.*)', line)
    if m:
      res.setdefault(m.group(5).strip(), {})[name] = (float(m.group(1)), int(m.group(2)), float(m.group(3)))
    m = re.match(r'total traced ms ([\d.]+)', line)
    if m:
      tot[name] = float(m.group(1))
names = [v[0] for v in VARIANTS]
print('total traced ms: ' + '  '.join(f'{n} {tot.get(n, float("nan")):.3f}' for n in names))
print(f'{"call site":58s} n  ' + '  '.join(f'{n:>9s}' for n in names) + '   (ms per step; TF of the first)')
for lab, d in sorted(res.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
  first = d.get(names[0], (float('nan'), 0, float('nan')))
  print(f'{lab[:58]:58s} {first[1]:3d} ' + '  '.join(f'{d.get(n, (float("nan"),))[0]:9.3f}' for n in names) + f'   {first[2]:.0f} TF')
