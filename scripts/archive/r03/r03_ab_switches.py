"""In-process A/B of run-time switches on ONE C2 database allocation: stand-alone ms per sweep launch and whole-query
queries/s (fold overlapped) per setting, responses hashed (must not change).  Usage: python scripts/r03_ab_switches.py
name=v[,name=v...] ... (each argument one variant; the baseline is run first and last)."""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np
import torch

import bench
import sdk_amd as sp


def main():
    cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
    steps = int(os.environ.get("STEPS", "12"))
    variants = [dict()] + [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[1:]] + [dict()]
    names = sorted({k for v in variants for k in v})
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    torch.cuda.synchronize()
    ref = None
    for v in variants:
        for k in names:
            sp.lib().sp_debug_set(k.encode(), C.c_long(v.get(k, DEFAULTS.get(k, 0))))
        run = sp.QueryRun(p, pp, qs[0], db=db)
        alone = run.bench_sweep(db, 3)
        run.free()
        stage = np.zeros(4)
        for i in range(2 + steps):
            if i == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                stage[:] = 0
            run = sp.QueryRun(p, pp, qs[i % 4], db=db)
            run.sweep(db)
            out = run.finish()
            stage += np.array(run.timings())
            run.free()
            if i == 0:
                sha = hashlib.sha256(out).hexdigest()[:12]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ref = ref or sha
        print("%-44s sweep alone %.3f ms/launch | %.2f q/s  expand %.3f sweep-span %.3f exposed-fold %.3f pack %.3f  %s" %
              (v or "baseline", alone, steps / dt, *(stage / steps), "ok" if sha == ref else "RESPONSE CHANGED"), flush=True)


DEFAULTS = {"pipe_wgs": 4, "pipe_unroll": 4, "sweep_prio": 1, "fold_variant": 5, "fused_min_pairs": 256, "expand_split": -1,
            "pipe_ring": 8, "pipe_ring_wgs": 1, "pipe_tail_defer": 256, "batch_in_flight": 3, "pipeline": 1, "sweep_nt_store": 1}

if __name__ == "__main__":
    main()
