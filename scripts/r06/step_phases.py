#!/usr/bin/env python
"""Phases of one batched step from a kernel trace dump (scripts/trace_dump.py): for the window between two consecutive batched
passes, the wall-clock union and the kernel-time sum of every kernel class, and the chain of the expansion phase kernel by kernel.
usage: step_phases.py trace.tsv [pass index, default 2] [--list]"""
import collections
import sys

rows = [l.rstrip('\n').split('\t') for l in open(sys.argv[1]) if not l.startswith('#')][1:]
R = [(r[0], float(r[1]), float(r[2]), r[3] if len(r) > 3 else '', r[5] if len(r) > 5 else '', r[6] if len(r) > 6 else '') for r in rows]
PASS = ('k_sweep_mfma_batch', 'k_sweep_planar', 'k_sweep_packed_batch')
passes = [r for r in R if r[0].startswith(PASS)]
args = [a for a in sys.argv[2:] if not a.startswith('--')]
k = int(args[0]) if args else 2
if len(passes) < 2:
    sys.exit("step_phases: %d batched passes in the trace" % len(passes))
k = min(k, len(passes) - 2)
t0, t1 = passes[k][1], passes[k + 1][1]


def cls(n):
    if n.startswith(PASS):
        return 'pass'
    if n.startswith('k_fold'):
        return 'fold'
    if n.startswith('k_from'):
        return 'from_ntt'
    if n.startswith(("k_expand", "k_ntt_inv_group", "k_ntt_fwd3", "k_mac2")):
        return 'expand_rounds'
    if n.startswith(('k_ntt_', 'k_mac', 'k_reorient', 'k_copy_polys', 'k_folding_neg', 'k_mats_to_wave', 'k_add_poly', 'k_query_')):
        return 'expand_tail+pack'
    return 'other'


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0.0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)


W = [r for r in R if r[1] >= t0 and r[1] < t1]
by = collections.defaultdict(list)
for n, s, d, *_ in W:
    by[cls(n)].append((s, s + d))
print("# step = pass %d .. pass %d: %.2f ms, %d kernels" % (k, k + 1, (t1 - t0) / 1e3, len(W)))
print("| class | launches | kernel-time sum ms | wall-clock union ms | first start ms | last end ms |\n|---|---|---|---|---|---|")
for c, iv in sorted(by.items(), key=lambda kv: min(x[0] for x in kv[1])):
    print("| %s | %d | %.2f | %.2f | %.2f | %.2f |" % (c, len(iv), sum(e - s for s, e in iv) / 1e3, union(iv) / 1e3,
                                                   (min(s for s, e in iv) - t0) / 1e3, (max(e for s, e in iv) - t0) / 1e3))
if '--list' in sys.argv:
    ex = [r for r in W if cls(r[0]) in ('expand_rounds',)]
    print("\n# the shared expansion rounds, launch by launch (start from the step's pass, duration, grid)")
    for n, s, d, q, gx, gy in ex:
        print("%-28s %9.1f %8.1f  grid %s x %s" % (n, s - t0, d, gx, gy))
