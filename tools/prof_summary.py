"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) per kernel and,
for the implicit-GEMM kernels, per launch shape.  usage: prof_summary.py DB STEPS"""
import re
import sqlite3
import sys

db, steps = sys.argv[1], float(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '').replace('sg2im::', '')[:64]
print('total kernel time %.3f ms over %g steps = %.3f ms/step' % (tot / 1e6, steps, tot / 1e6 / steps))
print('%-64s %8s %10s %9s %6s' % ('kernel', 'n/step', 'ms/step', 'avg us', '%'))
for r in rows[:40]:
  print('%-64s %8.1f %10.3f %9.1f %6.1f' % (short(r[0]), r[1] / steps, r[2] / 1e6 / steps, r[3] / 1e3, 100.0 * r[2] / tot))
print('\nimplicit-GEMM launches by shape')
rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start) from kernels "
                 "where name like '%conv_%' group by name, grid_x, grid_y, grid_z order by 6 desc").fetchall()
ctot = sum(r[5] for r in rows)
print('conv kernels: %.3f ms/step' % (ctot / 1e6 / steps))
for r in rows[:45]:
  print('%-40s grid=(%5d,%5d,%3d) n/step=%5.1f ms/step=%7.3f avg=%8.1fus' % (
    short(r[0]), r[1] // 256, r[2], r[3], r[4] / steps, r[5] / 1e6 / steps, r[6] / 1e3))
