"""How full is the GPU over a training step?  From a rocprofv3 --kernel-trace database: every kernel
is given a fill estimate min(1, workgroups / 512) (256 CUs x 2 resident workgroups), the running
kernels' fills are summed over time, and the steady-state window is split into
  full     (sum >= 0.5),  partial (0.1 <= sum < 0.5),  starved (0 < sum < 0.1),  idle (nothing running).
Also lists, per kernel name, the wall time during which that kernel was running while the GPU was
starved or partial - the launches whose latency is exposed.   usage: prof_timeline.py DB"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
wg = 'workgroup_x' if 'workgroup_x' in cols else None
q = "select start, end, name, grid_x, grid_y, grid_z%s from kernels order by start" % (
  ', workgroup_x, workgroup_y, workgroup_z' if wg else '')
rows = c.execute(q).fetchall()
n = len(rows)
rows = rows[n // 3: n - n // 10]
short = lambda s: re.sub(r'\(.*', '', s).replace('void ', '').replace('sg2im::', '')[:60]
ev = []
for i, r in enumerate(rows):
  s, e, name, gx, gy, gz = r[:6]
  wx, wy, wz = (r[6:9] if wg else (256, 1, 1))
  wgs = max(1, (gx // max(1, wx)) * (gy // max(1, wy)) * (gz // max(1, wz)))
  fill = min(1.0, wgs / 512.0)
  ev.append((s, 1, i, fill)); ev.append((e, -1, i, fill))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
running = {}
cls_time = collections.Counter()
blame = collections.Counter()
prev = t0
for t, kind, i, fill in ev:
  dt = t - prev
  if dt > 0:
    tot = sum(running.values())
    k = 'idle' if not running else 'full' if tot >= 0.5 else 'partial' if tot >= 0.1 else 'starved'
    cls_time[k] += dt
    if k in ('partial', 'starved'):
      for j in running:
        blame[short(rows[j][2])] += dt / len(running)
  prev = t
  if kind == 1: running[i] = fill
  else: running.pop(i, None)
wall = t1 - t0
print('window %.2f ms' % (wall / 1e6))
for k in ('full', 'partial', 'starved', 'idle'):
  print('%-8s %8.2f ms  %5.1f %%' % (k, cls_time[k] / 1e6, 100.0 * cls_time[k] / wall))
print('\nexposed (GPU partial/starved) time by kernel:')
for name, t in blame.most_common(25):
  print('%-60s %8.2f ms  %5.1f %%' % (name, t / 1e6, 100.0 * t / wall))
