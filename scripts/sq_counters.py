#!/usr/bin/env python
"""Per-kernel sums of the counters of one or more rocprofv3 --pmc passes (rocpd sqlite files) as a markdown table.
usage: sq_counters.py pass1.db [pass2.db ...] -- kernel-name-fragment [...]"""
import sqlite3
import sys

args = sys.argv[1:]
dbs, frags = args[:args.index("--")], args[args.index("--") + 1:]
rows = {}
for db in dbs:
    c = sqlite3.connect(db)
    for k, cn, n, sm in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        ks = k.split('(')[0].replace('void spiral::', '').replace('spiral::', '')
        if any(f in ks for f in frags):
            rows[(ks, cn)] = (n, sm)
print("| kernel | counter | dispatches | sum |\n|---|---|---|---|")
for (ks, cn), (n, sm) in sorted(rows.items()):
    print("| %s | %s | %d | %.6g |" % (ks, cn, n, sm))
