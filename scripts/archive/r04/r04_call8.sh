#!/bin/bash
# round 4, last GPU call: the suite once more after the test-side changes (shared oracle cases), per-kernel HBM traffic of one
# un-pipelined query (FETCH_SIZE / WRITE_SIZE passes), and the bench lines the evidence set did not have (C3, two queries in
# flight, replicas mode at one GPU with 16 queries per step)
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 600 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/r04last_pytest.log 2>&1
tail -3 $O/r04last_pytest.log
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/q2 /tmp/q3
H="--headline-only --no-cpu-baseline"
SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q2 -o q2 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 1 > /tmp/q2.log 2>&1
SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q3 -o q3 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 1 > /tmp/q3.log 2>&1
python - "$(find /tmp/q2 -name '*.db' | head -1)" "$(find /tmp/q3 -name '*.db' | head -1)" > $O/r04last_pmc_per_kernel.md 2>&1 <<'PY'
import sqlite3, sys
def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for k, n, sm in c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        out[k.split('(')[0].replace('void spiral::', '').replace('spiral::', '')] = (n, sm)
    return out
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), SPIRAL_PIPELINE=0, ONE C2 query (bench.py --steps 1 --warmup 0 --sweep-iters 1)")
print("# KB summed over the kernel's dispatches of the process (the sweep kernel: 1 query launch + 2 bench launches, 4 planes each);")
print("# gfx950: FETCH_SIZE counts half of a wide streaming read (MI355X_MICROARCH.md), small strided reads are counted in full")
print("| kernel | dispatches | FETCH_SIZE KB | WRITE_SIZE KB |\n|---|---|---|---|")
for k in sorted(set(f) | set(w), key=lambda x: -(f.get(x, (0, 0))[1] + w.get(x, (0, 0))[1])):
    print("| %s | %d | %.0f | %.0f |" % (k, f.get(k, (0, 0))[0], f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]))
PY
cd $R
timeout 200 python bench.py $H --config c3 --steps 5 --warmup 1 > $O/r04last_bench_c3.json 2>/dev/null
timeout 150 python bench.py $H --two-in-flight > $O/r04last_bench_c2_two_in_flight.json 2>/dev/null
timeout 200 python bench.py $H --mode replicas --batch 16 --steps 4 --warmup 1 > $O/r04last_bench_c2_replicas16.json 2>/dev/null
for f in $O/r04last_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-40s %8.2f q/s %8.3f ms/step %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], (d.get("two_in_flight") or {}).get("value")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
cat $O/r04last_pmc_per_kernel.md | head -14
