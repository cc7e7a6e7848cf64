"""Stand-alone sweep time per plane launch at C2 as a function of WHEN the database is allocated: before or after the
public parameters (ORDER=db_first|pp_first), plain hipMalloc in both cases (db_contiguous stays off)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import sdk_amd as sp

order = os.environ.get("ORDER", "pp_first")
p = sp.Params(bench.CONFIGS["c2"])
if order == "db_first":
    db = sp.Database(p).fill_synthetic(bench.SEED)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
else:
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    db = sp.Database(p).fill_synthetic(bench.SEED)
q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
run = sp.QueryRun(p, pp, q, db=db)
ms = [run.bench_sweep(db, 8) for _ in range(3)]
run.sweep(db); run.finish(); run.free()
import time
t0 = time.perf_counter()
for i in range(20):
    sp.process_query(p, pp, q, db)
dt = (time.perf_counter() - t0) / 20
print(json.dumps({"order": order, "contiguous": os.environ.get("SPIRAL_DB_CONTIGUOUS", "0"), "sweep_ms_per_launch": [round(x, 4) for x in ms], "query_ms": round(dt * 1e3, 3)}), flush=True)
