"""C2 queries/s as a function of the CU partition of the fold / sweep overlap (SPIRAL_CU_SPLIT = CUs given to the
overlapped folds; 0 = shared CUs, the round-1 scheme), plus the placement probe of the CU masks."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
import sdk_amd as sp


def probe(lo, hi, blocks=2048):
    out = np.zeros(blocks * 2, dtype=np.uint32)
    rc = sp.lib().sp_debug_cu_probe(C.c_int(lo), C.c_int(hi), C.c_int(blocks), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != 0:
        return "probe failed: " + sp.lib().sp_last_error().decode()
    xcc = out[0::2] & 0xF
    hw = out[1::2] & 0x7FFFFFFF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 0x1
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = {x: len([1 for i in ids if i[0] == x]) for x in sorted(set(xcc.tolist()))}
    return "mask bits [%d,%d): %d distinct (xcc,se,sh,cu), per XCC %s" % (lo, hi, len(ids), per_xcc)


def main():
    for lo, hi in ((0, 0), (0, 64), (0, 8), (64, 256), (0, 32), (128, 256)):
        print(probe(lo, hi), flush=True)
    cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
    steps = int(os.environ.get("STEPS", "12"))
    for n in [int(x) for x in os.environ.get("SPLITS", "0,32,48,64,80,96,128,0").split(",")]:
        os.environ["SPIRAL_CU_SPLIT"] = str(n)
        p = sp.Params(cfg)
        pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
        qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
        db = sp.Database(p).fill_synthetic(bench.SEED)
        torch.cuda.synchronize()
        outs, stage = [], np.zeros(4)
        for i in range(3 + steps):
            if i == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                stage[:] = 0
            run = sp.QueryRun(p, pp, qs[i % 4], db=db)
            run.sweep(db)
            out = run.finish()
            stage += np.array(run.timings())
            run.free()
            if i < 4:
                outs.append(out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        import hashlib
        print(json.dumps({"split": n, "qps": steps / dt, "ms": dt * 1e3 / steps,
                          "stage_ms": [round(x / steps, 3) for x in stage],
                          "paths": sorted(sp.paths_taken()), "sha": hashlib.sha256(outs[0]).hexdigest()[:12]}), flush=True)
        del db, pp, p
        import gc
        gc.collect()


if __name__ == "__main__":
    main()
