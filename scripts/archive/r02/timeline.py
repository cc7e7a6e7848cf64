#!/usr/bin/env python
"""Compact per-query kernel timeline from a rocprofv3 results db: consecutive same-name kernels are merged."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start"))
names = [r[0].split('(')[0].replace('void spiral::', '').replace('spiral::', '') for r in rows]
sw = [i for i, n in enumerate(names) if n.startswith('k_sweep')]
q = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = sw[q - 1] + 1, sw[q] + 1
t0 = rows[lo][1]
agg = {}
prev = None
for r, n in zip(rows[lo:hi], names[lo:hi]):
    d = (r[2] - r[1]) / 1000
    agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += d
    if d > 60:
        print(f"{(r[1]-t0)/1000:9.1f} us  {d:8.1f} us  {n} grid=({r[3]},{r[4]},{r[5]})")
print("span us:", (rows[hi-1][2] - t0) / 1000)
for n, (cnt, tot) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"  {n:32s} {cnt:4d} calls {tot:9.1f} us")
