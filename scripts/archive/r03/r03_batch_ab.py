"""In-process A/B of the batched step (8 queries per sp_process_query_batch) on ONE C2 database allocation.
Usage: python scripts/r03_batch_ab.py name=v[,name=v] ..."""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import torch

import bench
import sdk_amd as sp

DEFAULTS = {"batch_mfma": 1, "batch_mfma_nb": 2, "batch_mfma_cpw": 16, "batch_pipeline": 0, "batch_mfma_lds_pad": 0, "batch_group": 8,
            "fold_variant": 5, "fused_min_pairs": 256}


def main():
    cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
    steps = int(os.environ.get("STEPS", "4"))
    B = int(os.environ.get("BATCH", "8"))
    variants = [dict()] + [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[1:]] + [dict()]
    names = sorted({k for v in variants for k in v})
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(B)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    torch.cuda.synchronize()
    ref = None
    for v in variants:
        for k in names:
            sp.lib().sp_debug_set(k.encode(), C.c_long(v.get(k, DEFAULTS.get(k, 0))))
        outs = sp.process_query_batch(p, pp, qs, db)
        sha = hashlib.sha256(b"".join(outs)).hexdigest()[:12]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            sp.process_query_batch(p, pp, qs, db)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ref = ref or sha
        print("%-64s %.2f ms per batch of %d = %.1f q/s  %s" % (v or "baseline", dt * 1e3, B, B / dt, "ok" if sha == ref else "RESPONSES CHANGED"), flush=True)


if __name__ == "__main__":
    main()
