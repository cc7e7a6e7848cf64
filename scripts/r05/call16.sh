#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for b in 16 12 9; do
  echo "== BATCH=$b"
  ONLY_BATCH=1 BATCH=$b timeout 300 python scripts/r05/ab.py batch_planar=0 batch_planar=1 batch_planar=0 batch_planar=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/r05c16_ab_batch_planar.txt
done
H="--headline-only --no-cpu-baseline"
for pl in 1 0 1 0; do
SPIRAL_BATCH_PLANAR=$pl timeout 200 python bench.py $H --batch 16 --steps 4 --warmup 1 2>/dev/null | tee $O/r05c16_bench_batch16_planar$pl.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bp=d['roofline'].get('batched_pass') or {}
print('bench batch 16 planar=$pl: %.1f q/s  %.2f ms/step  selfcheck %s  pass %s %.2f ms frac %.3f' % (d['value'], d['ms_per_step'], d.get('batch_selfcheck'), bp.get('kernel'), bp.get('ms_per_pass', 0), bp.get('frac', 0)))"
done
