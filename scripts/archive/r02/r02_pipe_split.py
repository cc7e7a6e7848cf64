"""Pipelined C2 query with the last plane (pipe_split = 1) or every plane (2) swept and folded as two chunk-parity classes."""
import ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
import sdk_amd as sp

def setv(**kw):
    for k, v in kw.items():
        sp.lib().sp_debug_set(k.encode(), C.c_long(v))

cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
p = sp.Params(cfg)
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
db = sp.Database(p).fill_synthetic(bench.SEED)
ref = None
for mode in [int(x) for x in os.environ.get("MODES", "0,1,2,0,1,2").split(",")]:
    setv(pipe_split=mode)
    stage = np.zeros(4)
    n = 12
    for i in range(3 + n):
        if i == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter(); stage[:] = 0
        r = sp.QueryRun(p, pp, qs[i % 4], db=db)
        r.sweep(db)
        out = r.finish()
        stage += np.array(r.timings())
        r.free()
        if i == 0:
            h = hashlib.sha256(out).hexdigest()[:12]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ref = ref or h
    print(json.dumps({"pipe_split": mode, "query_ms": round(dt * 1e3, 3), "qps": round(1 / dt, 1), "stage_ms": [round(x / n, 3) for x in stage],
                      "bytes_ok": h == ref, "launches": sp.QueryRun(p, pp, qs[0], db=db).sweep_launches(db)}), flush=True)
