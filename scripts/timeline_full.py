#!/usr/bin/env python
"""Every kernel of ONE query (rocprofv3 kernel trace): start, duration, gap to the previous kernel's end.
usage: timeline_full.py results.db [query_index=5]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
qi = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(c.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
names = [r[0].split('(')[0].replace('void spiral::', '').replace('spiral::', '') for r in rows]
enc = [i for i, n in enumerate(names) if n.startswith('k_encode')]
lo, hi = enc[qi - 1] + 1, enc[qi] + 1
t0 = rows[lo][1]
prev_end = t0
for r, n in zip(rows[lo:hi], names[lo:hi]):
    print(f"{(r[1]-t0)/1000:9.1f}  dur {(r[2]-r[1])/1000:8.1f}  gap {(r[1]-prev_end)/1000:7.1f}  {n} grid=({r[3]},{r[4]})")
    prev_end = max(prev_end, r[2])
print("query span us:", (rows[hi - 1][2] - t0) / 1000, " kernels:", hi - lo)
