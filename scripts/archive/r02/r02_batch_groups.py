"""Batch throughput at C2 by batch size and group size (SPIRAL_BATCH_GROUP; 0 = the library's choice)."""
import ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import sdk_amd as sp

def setv(**kw):
    for k, v in kw.items():
        sp.lib().sp_debug_set(k.encode(), C.c_long(v))

cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
p = sp.Params(cfg)
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
qs = [bench.synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(8)]
db = sp.Database(p).fill_synthetic(bench.SEED)
ref = hashlib.sha256(sp.process_query(p, pp, qs[0], db)).hexdigest()[:12]
for batch, group in [(8, 8), (8, 4), (8, 0), (16, 8), (16, 4), (32, 8), (6, 0), (8, 8), (8, 4)]:
    setv(batch_group=group)
    outs = None
    for i in range(1 + 3):
        if i == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = sp.process_query_batch(p, pp, [qs[k % 8] for k in range(batch)], db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ok = all(hashlib.sha256(outs[k]).hexdigest()[:12] == hashlib.sha256(sp.process_query(p, pp, qs[k % 8], db)).hexdigest()[:12] for k in (0, batch - 1))
    print(json.dumps({"batch": batch, "group": group, "ms_per_batch": round(dt * 1e3, 2), "qps": round(batch / dt, 1), "bytes_ok": ok and hashlib.sha256(outs[0]).hexdigest()[:12] == ref}), flush=True)
