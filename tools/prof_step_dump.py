"""One steady-state training step out of a rocprofv3 --kernel-trace database, kernel by kernel:
start offset (us), duration (us), queue, workgroups, name.  A step is delimited by the generator's
Adam launch (the largest adam_guarded_kernel grid).   usage: prof_step_dump.py DB [step_index_from_end]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = c.execute("select start, end, name, queue_id, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z "
                 "from kernels order by start").fetchall()
short = lambda s: re.sub(r'\(.*', '', s).replace('void ', '').replace('sg2im::', '')[:58]
adam = [i for i, r in enumerate(rows) if 'adam_guarded' in r[2]]
gmax = max(rows[i][4] for i in adam)
marks = [i for i in adam if rows[i][4] == gmax]
a, b = marks[-back - 1], marks[-back]
t0 = rows[a][1]
qs = {}
print('# step of %.1f us' % ((rows[b][1] - t0) / 1e3))
for r in rows[a + 1:b + 1]:
  q = qs.setdefault(r[3], len(qs))
  wgs = (r[4] // max(1, r[7])) * (r[5] // max(1, r[8])) * (r[6] // max(1, r[9]))
  print('%9.1f %8.1f  q%d %6d  %s' % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, q, wgs, short(r[2])))
