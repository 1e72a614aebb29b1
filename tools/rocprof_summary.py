"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count, total,
average duration.  Usage: python tools/rocprof_summary.py results.db [n_steps [skip]] > summary.csv
With n_steps (the number of train steps the traced command ran) the last two columns are the
kernel's time and launches PER STEP.  With `skip` they are counted over the steady-state window
only - the last n_steps - skip train steps, delimited by the optimizer kernel (k_adam: three
launches per train step, the actor's is the step's last big kernel) - so that what the process
does once (allocating and zero-filling the learner's buffers with torch, loading parameters, the
eager first step, graph capture) is not spread over the steps: rounds 1-4 divided everything by
n_steps and reported ~30 torch fill kernels and ~30 buffer copies "per step" that are start-up
work (the window column shows what a replayed step really launches)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
skip = int(sys.argv[3]) if len(sys.argv) > 3 else None
where, wsteps = '', None
if steps and skip is not None:
  adam = db.execute("select end from kernels where name like '%k_adam%' order by end").fetchall()
  per_step = round(len(adam) / steps)
  if per_step >= 1 and len(adam) == per_step * int(steps) and skip < steps:
    t0, t1 = adam[per_step * skip - 1][0], adam[-1][0]
    where, wsteps = f' where start >= {t0} and end <= {t1}', steps - skip
rows = db.execute(
    'select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
    'from kernels group by name order by 3 desc').fetchall()
win = {}
if where:
  win = {r[0]: (r[1], r[2]) for r in db.execute(
      f'select name, count(*), sum(end-start) from kernels{where} group by name').fetchall()}
tot = sum(r[2] for r in rows)
per = f',ms_per_step(n={steps:g}),calls_per_step' if steps else ''
if wsteps:
  per += f',steady_ms_per_step(last {wsteps:g} steps),steady_calls_per_step'
print('kernel,calls,total_ms,pct,avg_us,min_us,max_us' + per)
for name, n, s, a, mn, mx in rows:
  raw = name
  name = re.sub(r'\(anonymous namespace\)::', '', name).replace(',', ';')
  tail = f',{s/1e6/steps:.3f},{n/steps:.1f}' if steps else ''
  if wsteps:
    wn, ws = win.get(raw, (0, 0))
    tail += f',{ws/1e6/wsteps:.3f},{wn/wsteps:.1f}'
  print(f'"{name[:160]}",{n},{s/1e6:.3f},{100*s/tot:.2f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f}' + tail)
ncalls = sum(r[1] for r in rows)
tail = f',{tot/1e6/steps:.3f},{ncalls/steps:.1f}' if steps else ''
if wsteps:
  tail += f',{sum(v[1] for v in win.values())/1e6/wsteps:.3f},{sum(v[0] for v in win.values())/wsteps:.1f}'
print(f'"TOTAL",{ncalls},{tot/1e6:.3f},100,,,' + tail)
