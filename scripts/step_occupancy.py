#!/usr/bin/env python
"""Per-millisecond occupancy of one batched step from a kernel trace dump (scripts/trace_dump.py): how many kernels of each
class are in flight on average (pass / from_ntt / fold / expansion / other) and how many hardware queues are in use.
usage: step_occupancy.py trace.tsv [index of the batched pass that starts the window, default 2]"""
import collections
import sys

rows = [l.rstrip('\n').split('\t') for l in open(sys.argv[1]) if not l.startswith('#')][1:]
R = [(r[0], float(r[1]), float(r[2]), r[3], r[4]) for r in rows]
PASS_KERNELS = ('k_sweep_mfma_batch', 'k_sweep_planar', 'k_sweep_packed_batch')   # whichever pass the group took (r05: planar)
passes = [r for r in R if r[0].startswith(PASS_KERNELS)]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if len(passes) < 2:
    sys.exit("step_occupancy: %d batched passes in %s (kernels seen: %s) -- need two to delimit a step"
             % (len(passes), sys.argv[1], ", ".join(sorted({r[0].split('<')[0] for r in R}))[:400]))
k = min(k, len(passes) - 2)
t0, t1 = passes[k][1] - 200, passes[k + 1][1] - 200


def cls(n):
    if n.startswith(PASS_KERNELS):
        return 'pass'
    if n.startswith('k_fold'):
        return 'fold'
    if n.startswith('k_from'):
        return 'from_ntt'
    if n.startswith(('k_ntt_fwd', 'k_ntt_inv', 'k_mac', 'k_expand')):
        return 'expand'
    return 'other'


B = 1000.0
nb = int((t1 - t0) / B) + 1
buckets = [collections.Counter() for _ in range(nb)]
queues = [set() for _ in range(nb)]
for n, s, d, q, st in R:
    e = s + d
    if e < t0 or s > t1:
        continue
    for b in range(max(0, int((s - t0) / B)), min(nb - 1, int((e - t0) / B)) + 1):
        lo = t0 + b * B
        ov = max(0, min(e, lo + B) - max(s, lo))
        buckets[b][cls(n)] += ov
        queues[b].add(q)
print("# window: from pass %d to pass %d = %.2f ms (one step of the batched bench)" % (k, k + 1, (t1 - t0) / 1e3))
print("| ms | kernels in flight by class (time-average) | hardware queues |\n|---|---|---|")
for b, c in enumerate(buckets):
    print("| %d | %s | %d |" % (b, "  ".join("%s %.2f" % (kk, v / B) for kk, v in sorted(c.items())), len(queues[b])))
