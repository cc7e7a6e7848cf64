#!/bin/bash
cd "$(dirname "$0")/.."
for v in "$@"; do
  env $v python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench, sdk_amd as sp
cfg = bench.CONFIGS["c2"]
p = sp.Params(cfg)
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
db = sp.Database(p).fill_synthetic(1)
run = sp.QueryRun(p, pp, bench.synthetic_wire_bytes(p.query_bytes(), 2))
ms = [run.bench_sweep(db, 10) for _ in range(2)]
b = bench.sweep_algorithmic_bytes(cfg, 1)
print({k: v for k, v in os.environ.items() if k.startswith("SPIRAL_")}, ["%.3f ms %.0f GB/s" % (m, b / m / 1e6) for m in ms])
PY
done
