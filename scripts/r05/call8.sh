#!/bin/bash
# Round 5, GPU call 8: a group's expansions as ONE chain of table launches (launch recording, batch_expand) -- parity of the
# batched paths on the GPU, then in-process A/B at 16, 8 and 32 queries per step; kernel trace of the 16-query step.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_request_layer.py -x -q -m gpu -k "batch or request or private_read" ) > $O/r05c8_pytest.log 2>&1
tail -2 $O/r05c8_pytest.log
grep -q " passed" $O/r05c8_pytest.log && ! grep -q " failed\| error" $O/r05c8_pytest.log || { echo "parity FAILED"; tail -40 $O/r05c8_pytest.log; exit 1; }
for b in 16 8 32 4; do
  echo "== BATCH=$b"
  ONLY_BATCH=1 BATCH=$b timeout 300 python scripts/r05/ab.py batch_expand=0 batch_expand=1 batch_expand=0 batch_expand=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/r05c8_ab_batch_expand.txt
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/t1
H="--headline-only --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/t1 -o t1 -- python $R/bench.py $H --batch 16 --steps 3 --warmup 1 > $O/r05c8_batch16_profiled.json 2> /tmp/t1.err
python $R/scripts/rocprof_summary.py "$(find /tmp/t1 -name '*.db' | head -1)" $O/r05c8_batch16_kernel_stats.md > /dev/null 2>&1
python $R/scripts/trace_dump.py "$(find /tmp/t1 -name '*.db' | head -1)" $O/r05c8_batch16_trace.tsv
python $R/scripts/step_occupancy.py $O/r05c8_batch16_trace.tsv 2 > $O/r05c8_batch16_occupancy.md; cat $O/r05c8_batch16_occupancy.md
cd $R
for b in 8 16; do timeout 200 python bench.py $H --batch $b --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench batch $b: %.1f q/s  %.2f ms/step  selfcheck %s' % (d['value'], d['ms_per_step'], d.get('batch_selfcheck')))"; done
