#!/bin/bash
# Round 5, GPU call 2: the full -m gpu suite on the lazy-inverse library; kernel traces (with queue ids) of 16-query steps, of two
# single queries in flight and of the plain pipelined query -- raw material for the overlap work.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 700 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/r05c2_pytest.log 2>&1
tail -4 $O/r05c2_pytest.log
cd /tmp; export TMPDIR=/tmp
H="--headline-only --no-cpu-baseline"
rm -rf /tmp/t1 /tmp/t2 /tmp/t3
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/t1 -o t1 -- python $R/bench.py $H --batch 16 --steps 3 --warmup 1 > $O/r05c2_batch16_profiled.json 2> /tmp/t1.err
python $R/scripts/rocprof_summary.py "$(find /tmp/t1 -name '*.db' | head -1)" $O/r05c2_batch16_kernel_stats.md > /dev/null 2>&1
python $R/scripts/trace_dump.py "$(find /tmp/t1 -name '*.db' | head -1)" $O/r05c2_batch16_trace.tsv
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/t2 -o t2 -- python $R/bench.py $H --two-in-flight --steps 8 --warmup 2 > $O/r05c2_two_in_flight_profiled.json 2> /tmp/t2.err
python $R/scripts/trace_dump.py "$(find /tmp/t2 -name '*.db' | head -1)" $O/r05c2_two_in_flight_trace.tsv
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/t3 -o t3 -- python $R/bench.py $H --batch 8 --steps 3 --warmup 1 > $O/r05c2_batch8_profiled.json 2> /tmp/t3.err
python $R/scripts/trace_dump.py "$(find /tmp/t3 -name '*.db' | head -1)" $O/r05c2_batch8_trace.tsv
cd $R
for b in 16 32 64; do
  timeout 200 python bench.py $H --batch $b --steps 3 --warmup 1 2>/dev/null | tee $O/r05c2_batch$b.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b: %.1f q/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
timeout 200 python bench.py $H --two-in-flight --steps 20 --warmup 4 2>/dev/null | tee $O/r05c2_two_in_flight.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single %.2f q/s, two in flight %s' % (d['value'], d.get('two_in_flight')))"
ls -la $O/r05c2_*trace.tsv
