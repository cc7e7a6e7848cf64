#!/usr/bin/env python
"""Kernels of one query's expansion phase (between the previous query's last kernel and this query's sweep)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
names = [r[0].split('(')[0].replace('void spiral::', '').replace('spiral::', '') for r in rows]
sw = [i for i, n in enumerate(names) if n.startswith('k_sweep')]
q = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hi = sw[q]
lo = hi
while lo > 0 and not names[lo - 1].startswith('k_encode') and not names[lo - 1].startswith('k_sweep'):
    lo -= 1
t0 = rows[lo][1]
line = []
for r, n in zip(rows[lo:hi], names[lo:hi]):
    line.append(f"{n.replace('k_','')}:{(r[2]-r[1])/1000:.0f}")
print(" ".join(line))
print("expansion span us:", (rows[hi][1] - t0) / 1000, "kernels:", hi - lo, "busy us:", sum((r[2]-r[1]) for r in rows[lo:hi]) / 1000)
