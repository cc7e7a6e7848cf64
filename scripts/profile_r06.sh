#!/bin/bash
# Round-6 evidence set, ONE build on ONE box: the full -m gpu suite, the default bench line (C2 + secondary objects + CPU
# baseline), rocprofv3 kernel stats + per-query timeline of the headline command, kernel stats and per-kernel HBM traffic of the
# UN-pipelined query (from_ntt and the fold launches with the GPU to themselves), PMC traffic of the sweep (the record bench.py
# replays, with the kernel's signature), SQ counters of the fold kernels, kernel stats of the 8- and 16-query steps, short lines
# for C1 / P2 / lists / two in flight / forced distributed mode, one rank's critical path of the row-sharded query.
# Output under gpurun_out/<tag>_*; copy what is cited into profiles/.  Every step has its own tight timeout.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
O=$R/gpurun_out
mkdir -p $O
T=${1:-r06final}
bash scripts/box_fingerprint.sh > $O/${T}_box.txt 2>&1
( cat /sys/fs/cgroup/cpu.max; git -C $R rev-parse HEAD 2>/dev/null; python sdk_amd/kernel_signature.py ) >> $O/${T}_box.txt 2>&1
( time timeout 1200 python -m pytest tests -x -q -m gpu --durations=10 ) > $O/${T}_pytest.log 2>&1
tail -4 $O/${T}_pytest.log
grep -q " passed" $O/${T}_pytest.log || { echo "suite did not pass: stopping"; tail -30 $O/${T}_pytest.log; exit 1; }
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' ) > $O/${T}_smoke.txt 2>&1; tail -1 $O/${T}_smoke.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/${T}_bench_c2.json 2> $O/${T}_bench_c2.err || { echo "default bench failed"; tail -5 $O/${T}_bench_c2.err; exit 1; }
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5 /tmp/p6 /tmp/p7 /tmp/q2 /tmp/q3
H="--headline-only --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p1 -- python $R/bench.py $H > $O/${T}_bench_c2_profiled.json 2> /tmp/p1.err
python $R/scripts/rocprof_summary.py "$(find /tmp/p1 -name '*.db' | head -1)" $O/${T}_c2_kernel_stats.md > /dev/null 2>&1
python $R/scripts/timeline_full.py "$(find /tmp/p1 -name '*.db' | head -1)" 6 > $O/${T}_c2_query_timeline.txt 2>&1
SPIRAL_PIPELINE=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o p5 -- python $R/bench.py $H --steps 5 --warmup 2 > $O/${T}_bench_c2_unpipelined.json 2> /tmp/p5.err
python $R/scripts/rocprof_summary.py "$(find /tmp/p5 -name '*.db' | head -1)" $O/${T}_c2_unpipelined_kernel_stats.md > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o p2 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 2 > /tmp/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p3 -o p3 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 2 > /tmp/p3.log 2>&1
python $R/scripts/pmc_traffic.py "$(find /tmp/p2 -name '*.db' | head -1)" "$(find /tmp/p3 -name '*.db' | head -1)" $O/${T}_pmc_sweep_c2.json > /dev/null 2>&1
SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q2 -o q2 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 1 > /tmp/q2.log 2>&1
SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q3 -o q3 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 1 > /tmp/q3.log 2>&1
python $R/scripts/pmc_per_kernel.py "$(find /tmp/q2 -name '*.db' | head -1)" "$(find /tmp/q3 -name '*.db' | head -1)" > $O/${T}_pmc_per_kernel_unpipelined.md 2>&1
SPIRAL_PIPELINE=0 timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d /tmp/p6 -o p6 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 1 > /tmp/p6.log 2>&1
python $R/scripts/sq_counters.py "$(find /tmp/p6 -name '*.db' | head -1)" -- k_fold k_from_sweep > $O/${T}_fold_sq_counters.md 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o p4 -- python $R/bench.py $H --batch 8 --steps 4 --warmup 1 > $O/${T}_bench_c2_batch8_profiled.json 2> /tmp/p4.err
python $R/scripts/rocprof_summary.py "$(find /tmp/p4 -name '*.db' | head -1)" $O/${T}_batch8_kernel_stats.md > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o p7 -- python $R/bench.py $H --batch 16 --steps 4 --warmup 1 > $O/${T}_bench_c2_batch16_profiled.json 2> /tmp/p7.err
python $R/scripts/rocprof_summary.py "$(find /tmp/p7 -name '*.db' | head -1)" $O/${T}_batch16_kernel_stats.md > /dev/null 2>&1
python $R/scripts/trace_dump.py "$(find /tmp/p7 -name '*.db' | head -1)" /tmp/p7.tsv > /dev/null 2>&1 && python $R/scripts/step_occupancy.py /tmp/p7.tsv 2 > $O/${T}_batch16_occupancy.md 2>&1
python $R/scripts/r06/step_phases.py /tmp/p7.tsv 2 --list > $O/${T}_batch16_step_phases.md 2>&1
# a summary that is a Python traceback is not evidence: stop loudly
for f in $O/${T}_*.md; do
  if grep -q "Traceback (most recent call last)" "$f"; then echo "EVIDENCE FILE IS A TRACEBACK: $f"; cat "$f"; exit 1; fi
done
cd $R
timeout 150 python bench.py $H --config c1 > $O/${T}_bench_c1.json 2>/dev/null
timeout 150 python bench.py $H --config p2 > $O/${T}_bench_p2.json 2>/dev/null
timeout 150 python bench.py $H --config c1 --batch 8 --steps 8 --warmup 2 > $O/${T}_bench_c1_list8.json 2>/dev/null
timeout 150 python bench.py $H --config p2 --batch 8 --steps 8 --warmup 2 > $O/${T}_bench_p2_list8.json 2>/dev/null
timeout 150 python bench.py $H --two-in-flight > $O/${T}_bench_c2_two_in_flight.json 2>/dev/null
timeout 200 python bench.py $H --mode replicas --batch 8 --steps 4 --warmup 1 > $O/${T}_bench_c2_replicas8.json 2>/dev/null
SPIRAL_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 200 python bench.py $H > $O/${T}_bench_c2_dist1.json 2>/dev/null
timeout 400 python scripts/r05/rank_critical_path.py c2 8 4 2 > $O/${T}_rank_critical_path.jsonl 2>/dev/null
for f in $O/${T}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-44s %8.2f q/s %8.3f ms/step  sweep %.3f ms (frac %.3f, alone %.3f)" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["ms_per_launch"], r["frac"], r["standalone"]["ms_per_launch"]))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
head -12 $O/${T}_c2_kernel_stats.md
head -10 $O/${T}_c2_unpipelined_kernel_stats.md
head -8 $O/${T}_batch16_kernel_stats.md
head -8 $O/${T}_pmc_per_kernel_unpipelined.md
cat $O/${T}_pmc_sweep_c2.json | head -8
