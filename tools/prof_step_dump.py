"""One steady-state training step out of a rocprofv3 --kernel-trace database, kernel by kernel:
start offset (us), duration (us), queue, workgroups, name.  A step starts with its stage_batch_kernel launch (one per
replayed iteration: the batch copied into the bucket's static buffers) and runs to the next one (round 4 delimited by
"the largest adam_guarded_kernel grid", which the early Adam slice made ambiguous: its dump was a 322 us window).
usage: prof_step_dump.py DB [step_index_from_end]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = c.execute("select start, end, name, queue_id, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z "
                 "from kernels order by start").fetchall()
short = lambda s: re.sub(r'\(.*', '', s).replace('void ', '').replace('sg2im::', '')[:58]
marks = [i for i, r in enumerate(rows) if 'stage_batch_kernel' in r[2]]
a, b = marks[-back - 1], marks[-back]
t0 = rows[a][0]
qs = {}
print('# step of %.1f us, %d kernels (stage_batch to the next stage_batch)' % ((rows[b][0] - t0) / 1e3, b - a))
for r in rows[a:b]:
  q = qs.setdefault(r[3], len(qs))
  wgs = (r[4] // max(1, r[7])) * (r[5] // max(1, r[8])) * (r[6] // max(1, r[9]))
  print('%9.1f %8.1f  q%d %6d  %s' % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, q, wgs, short(r[2])))
