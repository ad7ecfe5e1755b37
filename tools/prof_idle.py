"""GPU busy/idle analysis of a rocprofv3 --kernel-trace database: union of kernel intervals over the
steady-state part of the run, and the distribution of gaps between consecutive kernels.
usage: prof_idle.py DB"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end, name from kernels order by start").fetchall()
n = len(rows)
rows = rows[n // 3: n - n // 10]                # steady state
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
conc = 0
for s, e, _ in rows[1:]:
  if s > cur_e:
    busy += cur_e - cur_s
    gaps.append(s - cur_e)
    cur_s, cur_e = s, e
  else:
    conc += min(e, cur_e) - s
    cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print('window %.2f ms, busy (union) %.2f ms = %.1f %%, idle %.2f ms over %d gaps (mean %.2f us)' % (
  wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps), (wall - busy) / max(1, len(gaps)) / 1e3))
print('kernel-time sum %.2f ms, overlapped (concurrent) time %.2f ms' % (sum(e - s for s, e, _ in rows) / 1e6, conc / 1e6))
gaps.sort()
for q in (0.5, 0.9, 0.99):
  print('gap p%d = %.2f us' % (int(q * 100), gaps[int(q * (len(gaps) - 1))] / 1e3))
print('gaps > 20 us: %d, total %.2f ms' % (sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
