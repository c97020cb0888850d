#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 rocpd database: python tools/kdur.py results.db [name filter]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = ("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(end-start), min(end-start), max(end-start) "
     "from kernels group by name, grid_x, grid_y, grid_z order by 7 desc")
for name, gx, gy, gz, wx, n, avg, mn, mx in c.execute(q):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} grid {gx // wx:5d}x{gy}x{gz:<3d} n={n:4d} avg {avg / 1e3:7.1f} min {mn / 1e3:7.1f} max {mx / 1e3:7.1f} us")
