#!/bin/bash
# Round 5, GPU call 7: a group's expansions enqueued by several host threads (batch_begin_threads) -- parity of the batched
# paths, then in-process A/B at 16 and 8 queries per step, with the default and with 16 hardware queues; a kernel trace of the
# 16-query step with the new default.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_request_layer.py -x -q -m gpu -k "batch or request or private_read" ) > $O/r05c7_pytest.log 2>&1
tail -2 $O/r05c7_pytest.log
for nq in 4 16; do
  for b in 16 8 32; do
    echo "== GPU_MAX_HW_QUEUES=$nq BATCH=$b"
    GPU_MAX_HW_QUEUES=$nq ONLY_BATCH=1 BATCH=$b timeout 300 python scripts/r05/ab.py batch_begin_threads=1 batch_begin_threads=2 batch_begin_threads=4 batch_begin_threads=8 batch_begin_threads=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/r05c7_ab_begin_threads.txt
  done
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/t1
H="--headline-only --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/t1 -o t1 -- python $R/bench.py $H --batch 16 --steps 3 --warmup 1 > $O/r05c7_batch16_profiled.json 2> /tmp/t1.err
python $R/scripts/trace_dump.py "$(find /tmp/t1 -name '*.db' | head -1)" $O/r05c7_batch16_trace.tsv
python $R/scripts/step_occupancy.py $O/r05c7_batch16_trace.tsv 2 > $O/r05c7_batch16_occupancy.md; cat $O/r05c7_batch16_occupancy.md
