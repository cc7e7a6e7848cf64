"""C1 / P2 (narrow sweep, whole query ~1 ms): expansion subtrees on two streams or not, and in which enqueue order."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import torch
import bench
import sdk_amd as sp

def setv(**kw):
    for k, v in kw.items():
        sp.lib().sp_debug_set(k.encode(), C.c_long(v))

for name in ("c1", "p2"):
    cfg = bench.CONFIGS[name]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    ref = None
    for split, order in ((0, 0), (1, 0), (1, 1), (1, 2), (0, 0), (1, 1), (1, 2)):
        setv(expand_split=split, expand_order=order)
        outs = [sp.process_query(p, pp, q, db) for q in qs]
        if ref is None:
            ref = outs
        assert outs == ref, "schedule changes the response"
        n = 200
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            sp.process_query(p, pp, qs[i % 4], db)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        stage = np.zeros(4)
        for i in range(8):
            run = sp.QueryRun(p, pp, qs[i % 4], db=db)
            run.sweep(db); run.finish()
            stage += np.array(run.timings()); run.free()
        print(json.dumps({"cfg": name, "expand_split": split, "expand_order": order, "ms": round(dt * 1e3, 4), "qps": round(1 / dt, 1),
                          "stage_ms": [round(x / 8, 3) for x in stage]}), flush=True)
