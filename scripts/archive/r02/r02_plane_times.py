"""Stand-alone duration of the sweep launch of every plane (C2), optionally with a dummy allocation made BEFORE the database
(PAD_GB) -- run under rocprofv3 --kernel-trace; scripts/r02_plane_times.sh prints the per-plane table."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import sdk_amd as sp

pad = int(os.environ.get("PAD_GB", "0"))
hold = torch.empty(pad << 30, dtype=torch.uint8, device="cuda") if pad else None
if os.environ.get("PAD_FREE") == "1":   # allocate, then give it back before the database is allocated
    del hold
    hold = None
    torch.cuda.empty_cache()
cfg = bench.CONFIGS["c2"]
p = sp.Params(cfg)
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
db = sp.Database(p).fill_synthetic(bench.SEED)
run = sp.QueryRun(p, pp, q, db=db)
ms = C.c_float(0)
rc = sp.lib().sp_bench_sweep(C.c_void_p(run.h), C.c_void_p(db.h), C.c_int(int(os.environ.get("ITERS", "6"))), C.byref(ms))
print("rc", rc, "avg ms per launch", ms.value, "pad", pad)
run.free()
