"""Stage timings (expand / sweep span / exposed fold / finish, ms) of whole queries with SPIRAL_EXPAND_SPLIT = 0 / 1."""
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

for name in os.environ.get("CFGS", "c1,p2,c2").split(","):
    cfg = bench.CONFIGS[name]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    for split in [int(x) for x in os.environ.get("SPLITS", "0,1,0,1").split(",")]:
        setv(expand_split=1 if split else 0, expand_fused=1 if split >= 2 else 0)   # 2: + k_expand_round_teams
        stage = np.zeros(4)
        n = 12
        for i in range(3 + n):
            if i == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter(); stage[:] = 0
            run = sp.QueryRun(p, pp, qs[i % 4], db=db)
            run.sweep(db)
            out = run.finish()
            stage += np.array(run.timings())
            run.free()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(json.dumps({"cfg": name, "split": split, "ms": round(dt * 1e3, 3), "qps": round(1 / dt, 1), "stage_ms": [round(x / n, 3) for x in stage]}), flush=True)
