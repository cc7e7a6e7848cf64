#!/bin/bash
# Round 5, GPU call 21: k_fold_cluster (one launch per small fold level, a workgroup per digit transform, the last workgroup of a
# step finishes it) -- parity first, then in-process A/B on one C2 allocation against the three-launch tail (fold_cluster=0) and
# with the cluster form also taking the levels of 256 / 512 steps (fold_cluster_below=512 / 1024); C1 and P2 single queries.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold or process_query_bytes" ) > $O/r05c21_pytest.log 2>&1
tail -2 $O/r05c21_pytest.log
grep -q " passed" $O/r05c21_pytest.log && ! grep -q " failed\| error" $O/r05c21_pytest.log || { echo "parity FAILED"; tail -40 $O/r05c21_pytest.log; exit 1; }
STEPS=16 timeout 400 python scripts/r05/ab.py fold_cluster=0 fold_cluster_below=512 fold_cluster_below=1024 fold_cluster=0 2>&1 | grep -v "^$" | tee $O/r05c21_ab_raw.txt
H="--headline-only --no-cpu-baseline"
for rep in 1 2; do for fc in 1 0; do for cfg in c1 p2; do
  SPIRAL_FOLD_CLUSTER=$fc timeout 150 python bench.py $H --config $cfg --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold_cluster=$fc rep $rep $cfg: %.1f q/s %.3f ms/step %s' % (d['value'], d['ms_per_step'], d['config'].get('stage_ms')))"
done; done; done 2>&1 | tee -a $O/r05c21_ab_raw.txt
