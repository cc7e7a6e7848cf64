#!/bin/bash
# Round 5, GPU call 3: hardware queues.  Call 2's traces show HIP streams s3 and s4 (the second workspace's main and fold stream)
# sharing hardware queue 4, and the 16 expansions of a batched step running 1.5 kernels at a time: ROCclr maps all streams onto
# GPU_MAX_HW_QUEUES (default 4) hardware queues.  Same binary, same box, the variable alone.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
H="--headline-only --no-cpu-baseline"
for rep in 1 2; do
for nq in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$nq
  timeout 200 python bench.py $H --two-in-flight --steps 20 --warmup 4 2>/dev/null | tee $O/r05c3_q${nq}_single_rep$rep.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('two_in_flight') or {}
print('queues $nq rep $rep: single %.2f q/s (sweep in situ %.3f alone %.3f), two in flight %.2f q/s %s' % (d['value'], d['roofline']['ms_per_launch'], d['roofline']['standalone']['ms_per_launch'], t.get('value', 0), t.get('responses')))"
  for b in 8 16; do
  timeout 200 python bench.py $H --batch $b --steps 4 --warmup 1 2>/dev/null | tee $O/r05c3_q${nq}_batch${b}_rep$rep.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $nq rep $rep: batch $b %.1f q/s  %.2f ms/step  selfcheck %s' % (d['value'], d['ms_per_step'], d.get('batch_selfcheck')))"
  done
done
done
export GPU_MAX_HW_QUEUES=8
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/t1 /tmp/t2
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/t1 -o t1 -- python $R/bench.py $H --batch 16 --steps 3 --warmup 1 > /dev/null 2> /tmp/t1.err
python $R/scripts/trace_dump.py "$(find /tmp/t1 -name '*.db' | head -1)" $O/r05c3_q8_batch16_trace.tsv
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/t2 -o t2 -- python $R/bench.py $H --two-in-flight --steps 8 --warmup 2 > /dev/null 2> /tmp/t2.err
python $R/scripts/trace_dump.py "$(find /tmp/t2 -name '*.db' | head -1)" $O/r05c3_q8_two_in_flight_trace.tsv
