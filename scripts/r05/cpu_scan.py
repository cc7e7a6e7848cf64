"""CPU baseline diagnostics on the GPU box's host: one stage of the oracle's all-core mode at one team size,
under whatever OpenMP / allocator environment the caller set.  Prints one line.
usage: cpu_scan.py sweep|fold|expand THREADS [planes_fraction]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
what, threads = sys.argv[1], int(sys.argv[2])
os.environ["OMP_NUM_THREADS"] = str(threads)
import numpy as np

import bench
import oracle

cfg = bench.CONFIGS["c2"]
o = oracle.Params(cfg)
N = 2048
tag = "bind=%s places=%s alloc=%s aff=%d" % (os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"),
                                             os.environ.get("ORACLE_TUNE_ALLOCATOR", "1"), len(os.sched_getaffinity(0)))
oracle.set_threads(threads)
if what == "sweep":
    nz = int(sys.argv[3]) if len(sys.argv) > 3 else 1024       # z-rows of one plane (2048 = 16 GiB)
    dim0, num_per = o.dim0, o.num_per
    rng = np.random.default_rng(1)
    v_reg = rng.integers(0, 1 << 28, nz * dim0 * 2, dtype=np.uint64)
    dbs = oracle.words_first_touch(nz, num_per * dim0)
    oracle.sweep_rows_avx2(dbs[:num_per * dim0 * 8], v_reg[:8 * dim0 * 2], 8, dim0, num_per)
    best = 1e9
    for rep in range(3):
        t0 = time.time()
        oracle.sweep_rows_avx2(dbs, v_reg, nz, dim0, num_per)
        best = min(best, time.time() - t0)
    gb = nz * num_per * dim0 * 8 / 1e9
    print("sweep  threads %3d  %s: %.3f s for %.1f GB = %.0f GB/s  (whole C2 query: %.3f s)" % (threads, tag, best, gb, gb / best, best * N / nz * 4), flush=True)
else:
    cl = oracle.Client(o)
    pp = cl.generate_keys(11)
    q = cl.generate_query(12345 % o.num_items, 12)
    t0 = time.time()
    v_reg, v_fold = o.expand_query(pp, q)
    v_neg = o.get_v_folding_neg(v_fold)
    t_first = time.time() - t0
    if what == "expand":
        best = 1e9
        for rep in range(3):
            t0 = time.time()
            o.expand_query(pp, q)
            o.get_v_folding_neg(v_fold)
            best = min(best, time.time() - t0)
        print("expand threads %3d  %s: %.3f s (first call %.3f)" % (threads, tag, best, t_first), flush=True)
    else:
        ka = 11
        w = 2 * 2 * o.t_gsw * 2 * N
        rng = np.random.default_rng(5)
        cts = rng.integers(0, 249561089, (1 << ka) * 2 * 2 * N, dtype=np.uint64)
        best = 1e9
        for rep in range(2):
            t0 = time.time()
            o.from_ntt_fold_parallel(cts, v_fold[:ka * w], v_neg[:ka * w], nu=ka, classes=threads)
            best = min(best, time.time() - t0)
        print("fold   threads %3d  %s: %.3f s per plane (whole C2 query: %.3f s)" % (threads, tag, best, best * 4), flush=True)
