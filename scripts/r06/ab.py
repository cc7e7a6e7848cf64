"""Round-6 copy of scripts/r05/ab.py (new switches in DEFAULTS; BATCH may be a comma list in ONLY_BATCH mode): in-process A/B on ONE C2 database allocation (placement alone moves a process by +-3-10 %): per variant of the
run-time switches, (a) the pipelined single query: queries/s and stage times; (b) the same query with pipeline = 0, where the
`fold` stage is from_ntt + the whole fold tree of all planes with nothing else on the GPU; (c) 16 queries per step.  The
baseline runs first and last.  Responses are hashed: they must not change.
Usage: python scripts/r05/ab.py name=v[,name=v...] ...      (env: STEPS, BATCH, CFG)"""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import bench
import sdk_amd as sp

DEFAULTS = {"batch_planar": 1, "pipeline": 1, "fused_min_pairs": 256, "pipe_tail_defer": 256, "fold_skip_dead_digits": 1, "fold_variant": 5,
            "batch_in_flight": 3, "expand_split": -1, "sweep_prio": 1, "pipe_ring": 8, "pipe_ring_wgs": 1, "from_sweep_xcd": 1, "sweep_nt_store": 1, "expand_round_min": 2048, "expand_round_odd": 0, "expand_group": 1, "expand_group_round_min": 4096, "expand_group_tail": 1, "batch_tables_merged": 1, "expand_wave_min_digits": 16, "fold_neg_materialise": 0, "fused_min_pairs_cap": 1 << 62}


def single(p, pp, qs, db, steps):
    stage = np.zeros(4)
    sha = None
    for i in range(2 + steps):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stage[:] = 0
        run = sp.QueryRun(p, pp, qs[i % 4], db=db)
        run.sweep(db)
        out = run.finish()
        stage += np.array(run.timings())
        run.free()
        if i == 0:
            sha = hashlib.sha256(out).hexdigest()[:12]
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0), stage / steps, sha


def main():
    cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
    steps = int(os.environ.get("STEPS", "12"))
    Bs = [int(x) for x in os.environ.get("BATCH", "16").split(",")]
    B = max(Bs)
    ref_b = {}
    variants = [dict()] + [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[1:]] + [dict()]
    names = sorted({k for v in variants for k in v} | {"pipeline"})
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(B)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    torch.cuda.synchronize()
    ref = None

    def set_all(v, **over):
        for k in names:
            sp.lib().sp_debug_set(k.encode(), C.c_long(over.get(k, v.get(k, DEFAULTS.get(k, 0)))))

    only_batch = os.environ.get("ONLY_BATCH") == "1"
    for v in variants:
        set_all(v)
        if only_batch:
            line = "%-44s" % (v or "baseline")
            for Bk in Bs:
                outs = sp.process_query_batch(p, pp, qs[:Bk], db)
                sp.process_query_batch(p, pp, qs[:Bk], db)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(6):
                    sp.process_query_batch(p, pp, qs[:Bk], db)
                torch.cuda.synchronize()
                bt = (time.perf_counter() - t0) / 6
                shab = hashlib.sha256(b"".join(outs)).hexdigest()[:12]
                ref_b.setdefault(Bk, shab)
                line += " | batch%d %.2f ms = %.1f q/s %s" % (Bk, bt * 1e3, Bk / bt, "ok" if shab == ref_b[Bk] else "RESPONSE CHANGED")
            print(line, flush=True)
            continue
        qps, st, sha = single(p, pp, qs, db, steps)
        if os.environ.get("ONLY_SINGLE") == "1":
            ref = ref or sha
            print("%-36s | %6.2f q/s expand %.3f sweep %.3f fold %.3f | %s" % (v or "baseline", qps, st[0], st[1], st[2], "ok" if sha == ref else "RESPONSE CHANGED"), flush=True)
            continue
        set_all(v, pipeline=0)
        qps0, st0, sha0 = single(p, pp, qs, db, 6)
        set_all(v)
        outs = sp.process_query_batch(p, pp, qs, db)
        sp.process_query_batch(p, pp, qs, db)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            sp.process_query_batch(p, pp, qs, db)
        torch.cuda.synchronize()
        bt = (time.perf_counter() - t0) / 4
        shab = hashlib.sha256(outs[0]).hexdigest()[:12]
        ref = ref or sha
        ok = "ok" if (sha == ref and sha0 == ref and shab == ref) else "RESPONSE CHANGED"
        print("%-36s | %6.2f q/s expand %.3f sweep %.3f fold %.3f | un-pipelined: sweep %.3f from_ntt+fold %.3f | batch%d %.2f ms = %.1f q/s | %s" %
              (v or "baseline", qps, st[0], st[1], st[2], st0[1], st0[2], B, bt * 1e3, B / bt, ok), flush=True)


if __name__ == "__main__":
    main()
