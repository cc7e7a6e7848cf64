"""Consecutive single queries on ONE C2 database, serial (finish k before begin k+1) against two in flight (begin + sweep of
query k+1 are queued before finish(k) is waited for): queries/s, responses hashed against the serial ones.
Usage: python scripts/r03_stream_overlap.py  (CFG=c2 STEPS=24)"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import sdk_amd as sp


def main():
    cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
    steps = int(os.environ.get("STEPS", "24"))
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    torch.cuda.synchronize()

    def serial(n):
        outs = []
        for i in range(n):
            run = sp.QueryRun(p, pp, qs[i % 4], db=db)
            run.sweep(db)
            outs.append(run.finish())
            run.free()
        return outs

    def two_in_flight(n, sweep_before_finish=True):
        outs, prev = [], None
        for i in range(n):
            run = sp.QueryRun(p, pp, qs[i % 4], db=db)
            if sweep_before_finish:
                run.sweep(db)
            if prev is not None:
                outs.append(prev.finish())
                prev.free()
            if not sweep_before_finish:
                run.sweep(db)
            prev = run
        outs.append(prev.finish())
        prev.free()
        return outs

    ref = [hashlib.sha256(o).hexdigest()[:12] for o in serial(4)]
    variants = (("serial", serial), ("two in flight (begin+sweep k+1, then finish k)", two_in_flight),
                ("expansion k+1 only under query k", lambda n: two_in_flight(n, False)))
    for name, fn in variants * int(os.environ.get("ROUNDS", "4")):
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = fn(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = all(hashlib.sha256(o).hexdigest()[:12] == ref[i % 4] for i, o in enumerate(outs))
        print("%-50s %.2f q/s (%.3f ms per query)  %s" % (name, steps / dt, 1e3 * dt / steps, "ok" if ok else "RESPONSES DIFFER"), flush=True)


if __name__ == "__main__":
    main()
