"""Chip occupancy over one steady-state train step from a rocprofv3 (rocpd sqlite) kernel trace:
how long something runs at all, how long only partial-chip kernels run (the persistent scans /
rollouts, launches with fewer workgroups than CUs), per-queue busy time, and what runs during the
under-filled stretches.  Usage: python tools/timeline.py results.db n_steps skip"""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
steps, skip = int(sys.argv[2]), int(sys.argv[3])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
print('columns:', cols)
def col(*names):
  for n in names:
    if n in cols: return n
  return None
sel = 'name, start, end, queue_id, (grid_x / workgroup_x) * (grid_y / workgroup_y) * (grid_z / workgroup_z), 1'
adam = db.execute("select end from kernels where name like '%k_adam%' order by end").fetchall()
per = round(len(adam) / steps)
t0, t1 = adam[per * skip - 1][0], adam[-1][0]
rows = db.execute(f'select {sel} from kernels where start >= {t0} and end <= {t1} order by start').fetchall()
n = steps - skip
print(f'window {(t1 - t0) / 1e6:.2f} ms = {n} steps of {(t1 - t0) / 1e6 / n:.3f} ms; {len(rows) / n:.0f} launches per step')
def wgs(r):
  return r[4]
ev = []
for r in rows:
  full = wgs(r) >= 256
  ev.append((r[1], 1, full)); ev.append((r[2], -1, full))
ev.sort()
nf = npart = 0
last = t0
busy_full = busy_part = idle = 0
for t, d, full in ev:
  dt = t - last
  if nf > 0: busy_full += dt
  elif npart > 0: busy_part += dt
  else: idle += dt
  last = t
  if full: nf += d
  else: npart += d
print(f'per step: a chip-filling launch running {busy_full / 1e6 / n:.2f} ms, only partial-chip launches {busy_part / 1e6 / n:.2f} ms, nothing {idle / 1e6 / n:.2f} ms')
q = collections.defaultdict(float)
for r in rows: q[r[3]] += r[2] - r[1]
print('busy per queue (ms/step):', {k: round(v / 1e6 / n, 2) for k, v in sorted(q.items())})
# which partial-chip kernels run alone
alone = collections.defaultdict(float)
ev2 = []
for i, r in enumerate(rows):
  ev2.append((r[1], 1, i)); ev2.append((r[2], -1, i))
ev2.sort()
act = set(); last = t0
for t, d, i in ev2:
  dt = t - last
  if act and not any(wgs(rows[j]) >= 256 for j in act):
    for j in act: alone[re.sub(r'\(anonymous namespace\)::', '', rows[j][0])[:50]] += dt / len(act)
  last = t
  if d > 0: act.add(i)
  else: act.discard(i)
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
  print(f'  under-filled time attributed to {k:52s} {v / 1e6 / n:.3f} ms/step')

# what runs next to the persistent kernels
for pat in ('k_imagine_rollout', 'k_imagine_reverse', 'k_observe_scan_fwd', 'k_observe_scan_bwd'):
  tot = collections.defaultdict(float); dur = 0.0
  for r in rows:
    if pat in r[0]:
      dur += r[2] - r[1]
      for o in rows:
        if o is not r and o[2] > r[1] and o[1] < r[2]:
          tot[re.sub(r'\(anonymous namespace\)::', '', o[0])[:60] + f' wgs{o[4]}'] += min(o[2], r[2]) - max(o[1], r[1])
  print(f'{pat}: {dur / 1e6 / n:.2f} ms/step; concurrent launches (ms/step of overlap):')
  for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print(f'    {v / 1e6 / n:.3f}  {k}')

# one steady-state period as a list: launches longer than 0.12 ms, by start time (ms from the window's
# second-to-last optimizer step), queue id in front
ad = [a[0] for a in adam]
p0, p1 = ad[-per * 2 - 1], ad[-1]
print(f'--- last two steps ({(p1 - p0) / 1e6:.2f} ms): queue start end name')
for r in rows:
  if r[1] >= p0 and r[2] - r[1] > 120000:
    print(f'  q{r[3]} {(r[1] - p0) / 1e6:7.2f} {(r[2] - p0) / 1e6:7.2f}  {re.sub(r"[(]anonymous namespace[)]::", "", r[0])[:70]} wgs{r[4]}')

# everything that runs between the end of the world-model backward and the first kernel of the next
# world-model phase (around the behaviour phase's rollout), short launches merged per name
import os
if os.environ.get('TL_ZOOM'):
  ro = [r for r in rows if 'k_imagine_rollout' in r[0]][-2]
  z0, z1 = ro[1] - 1500000, ro[2] + 5500000
  print(f'--- zoom around the rollout ({(ro[1] - p0) / 1e6:.2f} .. {(ro[2] - p0) / 1e6:.2f} ms)')
  lastn = None
  for r in rows:
    if r[2] >= z0 and r[1] <= z1:
      nm = re.sub(r"[(]anonymous namespace[)]::", "", r[0])[:48]
      print(f'  q{r[3]} {(r[1] - p0) / 1e6:8.3f} {(r[2] - p0) / 1e6:8.3f}  {nm} wgs{r[4]}')
