"""sweep_spread A/B on ONE C2 database allocation: stand-alone ms per plane launch, whole-query ms, responses equal."""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import bench
import sdk_amd as sp

def setv(**kw):
    for k, v in kw.items():
        sp.lib().sp_debug_set(k.encode(), C.c_long(v))

p = sp.Params(bench.CONFIGS["c2"])
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
db = sp.Database(p).fill_synthetic(bench.SEED)
q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
ref = None
for spread, zmul in ((0, 1), (0, 5), (0, 129), (0, 683), (0, 1), (0, 129), (0, 1025), (1, 129)):
    setv(sweep_spread=spread, sweep_zmul=zmul)
    run = sp.QueryRun(p, pp, q, db=db)
    ms = [run.bench_sweep(db, 8) for _ in range(2)] + [run.bench_sweep(db, 2, 1)]
    run.sweep(db); out = run.finish(); run.free()
    if ref is None:
        ref = out
    assert out == ref, "sweep_spread changes the response"
    t0 = time.perf_counter()
    for i in range(16):
        sp.process_query(p, pp, q, db)
    dt = (time.perf_counter() - t0) / 16
    print(json.dumps({"sweep_spread": spread, "sweep_zmul": zmul, "sweep_ms_per_launch": [round(x, 4) for x in ms], "query_ms": round(dt * 1e3, 3)}), flush=True)
