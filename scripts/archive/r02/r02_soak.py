"""Repetition soak: the same queries over and over through every entry point (single, QueryRun split, batch 1..8, two host
threads), every response compared with the first one -- looks for ordering bugs between the streams of a workspace."""
import hashlib, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import sdk_amd as sp

def sha(b): return hashlib.sha256(b).hexdigest()[:16]

for name, reps in (("c1", 400), ("p2", 200), ("c2", 40)):
    cfg = bench.CONFIGS[name]
    p = sp.Params(cfg)
    pps = [sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1 + k)) for k in range(2)]
    qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(8)]
    db = sp.Database(p).fill_synthetic(bench.SEED)
    ref = {(k, i): sha(sp.process_query(p, pps[k], qs[i], db)) for k in range(2) for i in range(8)}
    bad = 0
    t0 = time.time()
    def worker(tid, n):
        global bad
        for r in range(n):
            k, i = (r + tid) % 2, (r * 3 + tid) % 8
            mode = r % 4
            if mode == 0:
                out = sp.process_query(p, pps[k], qs[i], db)
            elif mode == 1:
                run = sp.QueryRun(p, pps[k], qs[i], db=db)
                out = run.sweep(db).finish()
                run.free()
            elif mode == 2:
                run = sp.QueryRun(p, pps[k], qs[i])      # begun without the database (no split expansion)
                out = run.sweep(db).finish()
                run.free()
            else:
                B = 1 + r % 8
                outs = sp.process_query_batch(p, pps[k], [qs[(i + j) % 8] for j in range(B)], db)
                for j, o in enumerate(outs):
                    if sha(o) != ref[(k, (i + j) % 8)]:
                        bad += 1
                out = outs[0]
            if sha(out) != ref[(k, i)]:
                bad += 1
    ths = [threading.Thread(target=worker, args=(t, reps)) for t in range(2)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    print(name, "iterations", 2 * reps, "mismatches", bad, "seconds %.1f" % (time.time() - t0), flush=True)
    del db
