"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count, total,
average duration.  Usage: python tools/rocprof_summary.py results.db [n_steps] > summary.csv"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute(
    'select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
    'from kernels group by name order by 3 desc').fetchall()
tot = sum(r[2] for r in rows)
print('kernel,calls,total_ms,pct,avg_us,min_us,max_us,ms_per_step')
for name, n, s, a, mn, mx in rows:
  name = re.sub(r'\(anonymous namespace\)::', '', name).replace(',', ';')
  print(f'"{name[:160]}",{n},{s/1e6:.3f},{100*s/tot:.2f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{s/1e6/steps:.3f}')
print(f'"TOTAL",{sum(r[1] for r in rows)},{tot/1e6:.3f},100,,,,{tot/1e6/steps:.3f}')
