#!/usr/bin/env python
"""Time the db-sweep kernel variants (SPIRAL_SWEEP_VARIANT) on the C2 synthetic DB: one process per variant."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import bench, sdk_amd as sp
cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
p = sp.Params(cfg)
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
db = sp.Database(p).fill_synthetic(1)
run = sp.QueryRun(p, pp, bench.synthetic_wire_bytes(p.query_bytes(), 2))
ms = [run.bench_sweep(db, 10) for _ in range(3)]
b = bench.sweep_algorithmic_bytes(cfg, 1)
print(os.environ.get("SPIRAL_SWEEP_VARIANT"), ["%%.3f ms %%.0f GB/s" %% (m, b / m / 1e6) for m in ms])
''' % ROOT
for v in sys.argv[1:] or ["0", "1", "2", "3", "4"]:
    env = dict(os.environ, SPIRAL_SWEEP_VARIANT=v)
    subprocess.run([sys.executable, "-c", code], env=env)
