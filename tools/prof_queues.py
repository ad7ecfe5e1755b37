"""Per-replay queue usage out of a rocprofv3 --kernel-trace database: kernels are grouped into bursts (idle gap
> 100 us between them); for every distinct burst shape the hardware queues used, with kernel count, first start
and last end (us from the burst's start).   usage: prof_queues.py DB"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end, queue_id from kernels order by start").fetchall()
bursts, cur, last_end = [], [], None
for s, e, q in rows:
  if last_end is not None and s - last_end > 100000:
    bursts.append(cur)
    cur = []
  cur.append((s, e, q))
  last_end = e if last_end is None else max(last_end, e)
bursts.append(cur)
seen = {}
for b in bursts:
  t0 = b[0][0]
  qs = {}
  for s, e, q in b:
    d = qs.setdefault(q, [0, s, e])
    d[0] += 1
    d[2] = max(d[2], e)
  key = (len(b), tuple(sorted((q, d[0]) for q, d in qs.items())))
  seen.setdefault(key, []).append((b[-1][1] - t0, qs, t0))
for key, lst in seen.items():
  dur, qs, t0 = lst[-1]
  print('burst of %d kernels x%d: %.1f us' % (key[0], len(lst), max(d[2] for d in qs.values()) / 1e3 - t0 / 1e3))
  for q, d in sorted(qs.items()):
    print('   queue %s: %4d kernels, %8.1f .. %8.1f us' % (q, d[0], (d[1] - t0) / 1e3, (d[2] - t0) / 1e3))
