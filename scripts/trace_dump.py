#!/usr/bin/env python
"""Dump a rocprofv3 kernel trace (rocpd sqlite) as a compact TSV for offline timeline work:
  # columns: short kernel name, start_us (from the first kernel), dur_us, queue/stream ids as found, grid_x, grid_y
usage: trace_dump.py results.db out.tsv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
want = [k for k in ("name", "start", "end", "queue_id", "stream_id", "grid_x", "grid_y", "workgroup_x", "lds_size", "scratch_size", "vgpr_count") if k in cols]
rows = list(c.execute("select %s from kernels order by start" % ", ".join(want)))
t0 = rows[0][want.index("start")] if rows else 0
with open(sys.argv[2], "w") as f:
    f.write("# all columns of `kernels`: %s\n" % " ".join(cols))
    f.write("\t".join(["name", "start_us", "dur_us"] + [k for k in want if k not in ("name", "start", "end")]) + "\n")
    for r in rows:
        d = dict(zip(want, r))
        n = d["name"].split('(')[0].replace('void spiral::', '').replace('spiral::', '')
        f.write("\t".join([n, "%.1f" % ((d["start"] - t0) / 1000), "%.1f" % ((d["end"] - d["start"]) / 1000)] +
                          [str(d[k]) for k in want if k not in ("name", "start", "end")]) + "\n")
print(len(rows), "kernels ->", sys.argv[2])
