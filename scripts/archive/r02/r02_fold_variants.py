"""Fused fold kernel variants on ONE C2 database allocation: un-overlapped fold time of a whole query (pipeline off: one
sweep launch, then all four planes folded together), the pipelined query, and batches of 8."""
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


def timed(p, pp, qs, db, steps, batch=1):
    stage = np.zeros(4)
    sha = None
    for i in range(2 + steps):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stage[:] = 0
        if batch > 1:
            out = sp.process_query_batch(p, pp, [qs[(i + k) % len(qs)] for k in range(batch)], db)[0]
        else:
            run = sp.QueryRun(p, pp, qs[i % len(qs)], db=db)
            run.sweep(db)
            out = run.finish()
            stage += np.array(run.timings())
            run.free()
        if i == 0:
            sha = hashlib.sha256(out).hexdigest()[:12]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return steps * batch / dt, [round(x / steps, 3) for x in stage], sha


def main():
    cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(8)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    torch.cuda.synchronize()
    ref = None
    for variant in [int(x) for x in os.environ.get("VARIANTS", "3,4,3,4").split(",")]:
        setv(fold_variant=variant)
        setv(pipeline=0)
        q0, st0, sha = timed(p, pp, qs, db, 6)
        setv(pipeline=1)
        q1, st1, sha1 = timed(p, pp, qs, db, 10)
        setv(batch_pipeline=0)
        qb, _, shab = timed(p, pp, qs, db, 3, batch=8)
        setv(batch_pipeline=1)
        qbp, _, shabp = timed(p, pp, qs, db, 3, batch=8)
        ref = ref or sha
        print(json.dumps({"fold_variant": variant, "unpipelined_qps": round(q0, 2), "unpipelined_stage_ms": st0,
                          "pipelined_qps": round(q1, 2), "pipelined_stage_ms": st1, "batch8_qps": round(qb, 1),
                          "batch8_per_plane_qps": round(qbp, 1), "bytes_ok": sha == ref and sha1 == ref and shab == ref and shabp == ref,
                          "paths": sorted(sp.paths_taken())}), flush=True)


if __name__ == "__main__":
    main()
