"""Soak of the batched entry point on the full-size C2 database (round 6: the group's expansions are shared launches, the planar copy
is built at load time): lists of 2 .. 33 DISTINCT queries, fresh queries every iteration, every response compared with the
one-at-a-time path (which tests/test_gpu_fullsize.py ties to the oracle).  A race between the leader's stream and a query's own
would show here as a mismatch on some iteration.  Usage: python scripts/r06/soak_batch.py [iterations per list size, default 12]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import sdk_amd as sp  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    cfg = bench.CONFIGS["c2"]
    p = sp.Params(cfg)
    pps = [sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1 + k)) for k in range(2)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    print("planar copy built at load time:", db.prepare_batch(), flush=True)
    seed = 1000
    bad = 0
    t00 = time.time()
    for B in (16, 9, 2, 3, 5, 8, 12, 19, 33, 16):
        t0 = time.time()
        for it in range(iters):
            qs = [bench.synthetic_wire_bytes(p.query_bytes(), seed + i) for i in range(B)]
            seed += B
            pl = [pps[(i + it) % 2] for i in range(B)]          # two clients' public parameters interleaved
            outs = sp.process_query_batch(p, pl, qs, db)
            for i in (range(B) if it % 4 == 0 else (0, B - 1, (it * 7) % B)):
                if outs[i] != sp.process_query(p, pl[i], qs[i], db):
                    bad += 1
                    print("MISMATCH: list of %d, iteration %d, query %d (sha %s)" % (B, it, i, hashlib.sha256(outs[i]).hexdigest()[:12]), flush=True)
        print("lists of %2d: %d iterations, %.1f s, mismatches so far %d" % (B, iters, time.time() - t0, bad), flush=True)
    print("soak done in %.0f s: %s" % (time.time() - t00, "ALL EQUAL" if bad == 0 else "%d MISMATCHES" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
