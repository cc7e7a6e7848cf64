"""Batched sweep kernel variants on ONE C2 database allocation: queries/s of sp_process_query_batch for B = 4, 6, 8 with
the LDS-staged / scalar-load query forms and 2 or 4 row pairs per ping-pong buffer."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import sdk_amd as sp


def setv(**kw):
    for k, v in kw.items():
        sp.lib().sp_debug_set(k.encode(), C.c_long(v))


def main():
    cfg = bench.CONFIGS["c2"]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(8)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    torch.cuda.synchronize()
    single = [sp.process_query(p, pp, q, db) for q in qs]
    for B in [int(x) for x in os.environ.get("BS", "8,7,5,4,3,2").split(",")]:
        for qlds_min, lds_unroll in ((1, 2), (9, 4)):
            setv(batch_qlds_min=qlds_min, batch_lds_unroll=lds_unroll)
            outs = sp.process_query_batch(p, pp, qs[:B], db)
            ok = outs == single[:B]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 3
            for _ in range(steps):
                sp.process_query_batch(p, pp, qs[:B], db)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            print(json.dumps({"B": B, "query_rows": "lds" if B >= qlds_min else "scalar", "row_pairs_per_buffer": lds_unroll if B >= qlds_min else 4,
                              "ms_per_batch": round(dt * 1e3, 2), "qps": round(B / dt, 1), "bytes_ok": ok}), flush=True)


if __name__ == "__main__":
    main()
