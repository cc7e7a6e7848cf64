"""db-sweep kernel time on a row shard of C2 (rank 0 of G), one GPU: ms per query sweep and GB/s of the shard's
algorithmic bytes, single launch vs one launch per plane."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import sdk_amd as sp

cfg = bench.CONFIGS["c2"]
p = sp.Params(cfg)
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
for G in (8, 4, 2, 1):
    db = sp.Database(p, 0, G).fill_synthetic(0x123456789)
    torch.cuda.synchronize()
    run = sp.QueryRun(p, pp, q)
    alg = bench.sweep_algorithmic_bytes(cfg, G)
    one = run.bench_sweep(db, 5, per_plane=0)
    four = run.bench_sweep(db, 5, per_plane=1) * 4
    print("G=%d  one launch %.3f ms (%.0f GB/s)   per-plane launches %.3f ms (%.0f GB/s)   ideal at G=1 rate: see G=1 / G" %
          (G, one, alg / one / 1e6, four, alg / four / 1e6), flush=True)
    run.free()
    del db
    torch.cuda.synchronize()
