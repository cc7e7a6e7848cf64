#!/bin/bash
# Round 5, GPU call 15: the 9 .. 16-query pass over the digit-planar copy of the database in the library -- parity of the batched
# paths, in-process A/B (batch_planar = 0 / 1) at 16 and 12 queries per step, the bench's own batched-pass timing.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_request_layer.py -x -q -m gpu -k "batch or request or private_read" ) > $O/r05c15_pytest.log 2>&1
tail -2 $O/r05c15_pytest.log
grep -q " passed" $O/r05c15_pytest.log && ! grep -q " failed\| error" $O/r05c15_pytest.log || { echo "parity FAILED"; tail -40 $O/r05c15_pytest.log; exit 1; }
for b in 16 12 32; do
  echo "== BATCH=$b"
  ONLY_BATCH=1 BATCH=$b timeout 300 python scripts/r05/ab.py batch_planar=0 batch_planar=1 batch_planar=0 batch_planar=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/r05c15_ab_batch_planar.txt
done
H="--headline-only --no-cpu-baseline"
for pl in 0 1; do
SPIRAL_BATCH_PLANAR=$pl timeout 200 python bench.py $H --batch 16 --steps 4 --warmup 1 2>/dev/null | tee $O/r05c15_bench_batch16_planar$pl.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bp=d['roofline'].get('batched_pass') or {}
print('bench batch 16 planar=$pl: %.1f q/s  %.2f ms/step  selfcheck %s  pass %s %.2f ms frac %.3f' % (d['value'], d['ms_per_step'], d.get('batch_selfcheck'), bp.get('kernel'), bp.get('ms_per_pass', 0), bp.get('frac', 0)))"
done
