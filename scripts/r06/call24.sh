#!/bin/bash
# fold levels: fused-kernel threshold lowered at run time (fused_min_pairs_cap), batched steps and the single query
mkdir -p gpurun_out/r06_call24
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call24/build.log 2>&1
ONLY_BATCH=1 BATCH=8,16 timeout 1200 python scripts/r06/ab.py fused_min_pairs_cap=128 fused_min_pairs_cap=64 fused_min_pairs_cap=32 fused_min_pairs_cap=16 fused_min_pairs_cap=4 fused_min_pairs_cap=1 fused_min_pairs_cap=64 > gpurun_out/r06_call24/ab_batch.txt 2>&1
cat gpurun_out/r06_call24/ab_batch.txt
STEPS=20 BATCH=16 timeout 900 python scripts/r06/ab.py fused_min_pairs_cap=128 fused_min_pairs_cap=64 fused_min_pairs_cap=16 > gpurun_out/r06_call24/ab_c2.txt 2>&1
cat gpurun_out/r06_call24/ab_c2.txt
