"""In-process A/B of the per-plane sweep kernel variants on ONE C2 database allocation (HBM placement alone moves the
sweep by +-5 % between allocations): stand-alone ms per launch and the whole query (fold overlapped) for each."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
import sdk_amd as sp


def setv(**kw):
    for k, v in kw.items():
        sp.lib().sp_debug_set(k.encode(), C.c_long(v))


def main():
    cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
    steps = int(os.environ.get("STEPS", "10"))
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    torch.cuda.synchronize()
    variants = [dict(sweep_ring=0, pipe_wgs=4, sweep_dynamic=0)]
    for wgs in (5, 6):
        variants.append(dict(sweep_ring=0, pipe_wgs=wgs, sweep_dynamic=0))
    for ring, wgss in ((2, (4, 6)), (4, (2, 3, 4)), (8, (2,))):
        for wgs in wgss:
            for dyn in (0, 1):
                variants.append(dict(sweep_ring=ring, pipe_wgs=wgs, sweep_dynamic=dyn))
    variants.append(dict(sweep_ring=0, pipe_wgs=4, sweep_dynamic=0))
    ref = None
    for v in variants:
        setv(**v)
        run = sp.QueryRun(p, pp, qs[0], db=db)
        alone = run.bench_sweep(db, 3, per_plane=1)
        run.free()
        outs, stage = [], np.zeros(4)
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
        print(json.dumps({**v, "alone_ms_per_launch": round(alone, 4), "qps": round(steps / dt, 2),
                          "ms": round(dt * 1e3 / steps, 3), "stage_ms": [round(x / steps, 3) for x in stage],
                          "bytes_ok": sha == ref, "ring_path": "sweep_ring_prefetch" in sp.paths_taken()}), flush=True)


if __name__ == "__main__":
    main()
