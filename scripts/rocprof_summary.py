#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel stats table kept under profiles/."""
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    pmc = []
    try:
        pmc = list(c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                             "group by kernel_name, counter_name order by kernel_name"))
    except Exception as e:  # no counters in a kernel-trace run
        pmc = []
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        f.write("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|\n")
        for n, calls, tot, avg, pct in rows:
            f.write("| %s | %d | %.1f | %.3f | %.2f |\n" % (n, calls, tot, avg, pct))
        if pmc:
            f.write("\n# PMC counters (per kernel: dispatches, sum, mean per dispatch)\n")
            f.write("| kernel | counter | dispatches | sum | mean |\n|---|---|---|---|---|\n")
            for k, cn, cnt, sm, av in pmc:
                f.write("| %s | %s | %d | %.6g | %.6g |\n" % (k, cn, cnt, sm, av))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
