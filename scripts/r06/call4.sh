#!/bin/bash
# r06 call 4: trace of the 16-query step with the group expansion: phases of a step, the shared rounds launch by launch
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r06_call4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p7
H="--headline-only --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o p7 -- python $R/bench.py $H --batch 16 --steps 4 --warmup 1 > $O/bench_c2_batch16_profiled.json 2> /tmp/p7.err
DB="$(find /tmp/p7 -name '*.db' | head -1)"
python $R/scripts/rocprof_summary.py "$DB" $O/batch16_kernel_stats.md > /dev/null 2>&1
python $R/scripts/trace_dump.py "$DB" /tmp/p7.tsv; gzip -c /tmp/p7.tsv > $O/batch16_trace.tsv.gz
python $R/scripts/r06/step_phases.py /tmp/p7.tsv 2 --list > $O/batch16_step_phases.md 2>&1
python $R/scripts/step_occupancy.py /tmp/p7.tsv 2 > $O/batch16_occupancy.md 2>&1
head -60 $O/batch16_step_phases.md
head -24 $O/batch16_kernel_stats.md
