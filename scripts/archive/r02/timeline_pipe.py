#!/usr/bin/env python
"""Kernel timeline of ONE pipelined query from a rocprofv3 results db (kernel trace): every kernel longer than
`min_us` from the query's first expansion kernel to its encode, with start/end relative to the query start, so the
overlap of the per-plane sweeps (main stream) with from_ntt + fold (second stream) is visible.
usage: timeline_pipe.py results.db [query_index=5] [min_us=40] [sweeps_per_query=4]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
qi = int(sys.argv[2]) if len(sys.argv) > 2 else 5
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
spq = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rows = list(c.execute("select name, start, end, grid_x from kernels order by start"))
names = [r[0].split('(')[0].replace('void spiral::', '').replace('spiral::', '') for r in rows]
enc = [i for i, n in enumerate(names) if n.startswith('k_encode')]
lo = enc[qi - 1] + 1 if qi > 0 else 0
hi = enc[qi] + 1
t0 = rows[lo][1]
for r, n in zip(rows[lo:hi], names[lo:hi]):
    d = (r[2] - r[1]) / 1000
    if d >= min_us or n.startswith('k_encode'):
        print(f"{(r[1]-t0)/1000:9.1f} -> {(r[2]-t0)/1000:9.1f}  {d:8.1f} us  {n}  grid={r[3]}")
print("query span us:", (rows[hi - 1][2] - t0) / 1000, " kernels:", hi - lo)
