"""Host-side enqueue time against device time of the expansion (sp_query_begin returns after the last launch was queued)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import sdk_amd as sp

for name in os.environ.get("CFGS", "c1,c2").split(","):
    cfg = bench.CONFIGS[name]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run = sp.QueryRun(p, pp, q)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        run.free()
        if i >= 3:
            print(json.dumps({"cfg": name, "begin_enqueue_us": round((t1 - t0) * 1e6, 1), "until_idle_us": round((t2 - t0) * 1e6, 1)}))
