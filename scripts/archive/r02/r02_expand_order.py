"""Stage timings (expand / sweep span / exposed fold / finish, ms) of whole C2 queries for the enqueue order of the two
expansion subtrees: expand_order 0 (odd side queued first, r02 first version), 1 (even side queued first, odd side
starts after round 0), 2 (odd side starts when the even side is done, i.e. beside the first sweep)."""
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

for name in os.environ.get("CFGS", "c2").split(","):
    cfg = bench.CONFIGS[name]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    ref = None
    for order in [int(x) for x in os.environ.get("ORDERS", "0,1,2,0,1,2").split(",")]:
        setv(expand_order=order)
        if name != "c2":
            setv(expand_split=1)
        stage = np.zeros(4)
        n = 20
        for i in range(4 + n):
            if i == 4:
                torch.cuda.synchronize(); t0 = time.perf_counter(); stage[:] = 0
            out = sp.process_query(p, pp, qs[i % 4], db) if i >= 4 else None
            if i < 4:
                run = sp.QueryRun(p, pp, qs[i % 4], db=db)
                run.sweep(db)
                out = run.finish()
                run.free()
                if ref is None:
                    ref = [None] * 4
                if ref[i % 4] is None:
                    ref[i % 4] = out
                assert out == ref[i % 4], "expand_order %d changes the response" % order
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        # stage split from the stepwise entry points
        for i in range(8):
            run = sp.QueryRun(p, pp, qs[i % 4], db=db)
            run.sweep(db)
            out = run.finish()
            assert out == ref[i % 4]
            stage += np.array(run.timings())
            run.free()
        print(json.dumps({"cfg": name, "expand_order": order, "ms": round(dt * 1e3, 3), "qps": round(1 / dt, 1),
                          "stage_ms": [round(x / 8, 3) for x in stage]}), flush=True)
