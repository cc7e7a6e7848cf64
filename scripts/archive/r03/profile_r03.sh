#!/bin/bash
# Round-3 evidence set, all from ONE build on ONE box: the full -m gpu suite, the default bench line (C2), rocprofv3 kernel
# stats + per-query timeline + PMC traffic of the same command, bench lines for batch 8 / C4 / C1 / P2 / replicas / forced
# distributed mode, kernel stats of the batched step.  Output under gpurun_out/r03final_*; copy into profiles/.
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
T=${1:-r03final}
bash scripts/box_fingerprint.sh > $O/${T}_box.txt 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu > $O/${T}_pytest.log 2>&1
tail -3 $O/${T}_pytest.log
timeout 900 python bench.py > $O/${T}_bench_c2.json 2> $O/${T}_bench_c2.err
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p1 -- python $R/bench.py --no-cpu-baseline > $O/${T}_bench_c2_profiled.json 2> /tmp/p1.err
python $R/scripts/rocprof_summary.py "$(find /tmp/p1 -name '*.db' | head -1)" $O/${T}_c2_kernel_stats.md > /dev/null 2>&1
python $R/scripts/timeline_full.py "$(find /tmp/p1 -name '*.db' | head -1)" 6 > $O/${T}_c2_query_timeline.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o p2 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 2 --no-cpu-baseline > /tmp/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p3 -o p3 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 2 --no-cpu-baseline > /tmp/p3.log 2>&1
python $R/scripts/pmc_traffic.py "$(find /tmp/p2 -name '*.db' | head -1)" "$(find /tmp/p3 -name '*.db' | head -1)" $O/${T}_pmc_sweep_c2.json > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o p4 -- python $R/bench.py --batch 8 --steps 4 --warmup 1 --no-cpu-baseline > $O/${T}_bench_c2_batch8.json 2> /tmp/p4.err
python $R/scripts/rocprof_summary.py "$(find /tmp/p4 -name '*.db' | head -1)" $O/${T}_batch8_kernel_stats.md > /dev/null 2>&1
cd $R
timeout 600 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/${T}_bench_c2_batch8_unprofiled.json 2>/dev/null
timeout 600 python bench.py --mode replicas --steps 5 --warmup 2 --no-cpu-baseline > $O/${T}_bench_c2_replicas1.json 2>/dev/null
timeout 600 python bench.py --config c1 --no-cpu-baseline > $O/${T}_bench_c1.json 2>/dev/null
timeout 600 python bench.py --config p2 --no-cpu-baseline > $O/${T}_bench_p2.json 2>/dev/null
timeout 600 python bench.py --config c1 --batch 8 --steps 8 --warmup 2 --no-cpu-baseline > $O/${T}_bench_c1_list8.json 2>/dev/null
timeout 600 python bench.py --config p2 --batch 8 --steps 8 --warmup 2 --no-cpu-baseline > $O/${T}_bench_p2_list8.json 2>/dev/null
SPIRAL_FORCE_DIST=1 MASTER_PORT=29655 timeout 600 python bench.py --no-cpu-baseline > $O/${T}_bench_c2_dist1.json 2>/dev/null
timeout 900 python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline > $O/${T}_bench_c4.json 2>/dev/null
timeout 900 python bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline > $O/${T}_bench_c3.json 2>/dev/null
for f in $O/${T}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%-52s %8.2f q/s %8.3f ms/step  sweep %.3f ms (frac %.3f, alone %.3f)  %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["ms_per_launch"], r["frac"], r["standalone"]["ms_per_launch"], ("pass %.2f ms frac %.3f" % (r["batched_pass"]["ms_per_pass"], r["batched_pass"]["frac"])) if r.get("batched_pass") else ""))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
head -14 $O/${T}_c2_kernel_stats.md
head -8 $O/${T}_batch8_kernel_stats.md
