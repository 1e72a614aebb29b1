"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count, total,
average duration.  Usage: python tools/rocprof_summary.py results.db [n_steps] > summary.csv
With n_steps (the number of train steps the traced command ran) the last two columns are the
kernel's time and launches PER STEP; without it they are omitted."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = db.execute(
    'select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
    'from kernels group by name order by 3 desc').fetchall()
tot = sum(r[2] for r in rows)
per = f',ms_per_step(n={steps:g}),calls_per_step' if steps else ''
print('kernel,calls,total_ms,pct,avg_us,min_us,max_us' + per)
for name, n, s, a, mn, mx in rows:
  name = re.sub(r'\(anonymous namespace\)::', '', name).replace(',', ';')
  tail = f',{s/1e6/steps:.3f},{n/steps:.1f}' if steps else ''
  print(f'"{name[:160]}",{n},{s/1e6:.3f},{100*s/tot:.2f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f}' + tail)
ncalls = sum(r[1] for r in rows)
print(f'"TOTAL",{ncalls},{tot/1e6:.3f},100,,,' + (f',{tot/1e6/steps:.3f},{ncalls/steps:.1f}' if steps else ''))
