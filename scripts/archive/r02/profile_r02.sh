# round-2 profile set: bench line, rocprofv3 kernel stats + per-query timeline of the same command, PMC traffic of the sweep
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
python $R/bench.py --steps 20 --warmup 5 > $O/r02_bench_c2.json 2> $O/r02_bench.err
tail -c 300 $O/r02_bench_c2.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p1.log 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/r02_c2_kernel_stats.md > /dev/null
python $R/scripts/timeline_full.py $(find /tmp/p1 -name "*.db" | head -1) 12 > $O/r02_c2_query_timeline.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o p2 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 2 --no-cpu-baseline > /tmp/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p3 -o p3 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 2 --no-cpu-baseline > /tmp/p3.log 2>&1
python $R/scripts/pmc_traffic.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) $O/r02_pmc_sweep_c2.json | tail -6
head -10 $O/r02_c2_kernel_stats.md
timeout 300 python $R/bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline > $O/r02_bench_c2_batch8.json 2>> $O/r02_bench.err
timeout 300 python $R/bench.py --steps 5 --warmup 2 --mode replicas --no-cpu-baseline > $O/r02_bench_c2_replicas1.json 2>> $O/r02_bench.err
for c in c1 p2; do timeout 300 python $R/bench.py --config $c --steps 50 --warmup 10 --no-cpu-baseline > $O/r02_bench_$c.json 2>> $O/r02_bench.err; done
timeout 600 python $R/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_c4.json 2>> $O/r02_bench.err
timeout 600 python $R/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_c3.json 2>> $O/r02_bench.err
SPIRAL_FORCE_DIST=1 timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_c2_dist1.json 2>> $O/r02_bench.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r02_bench_*.json")):
    try:
        j=json.load(open(f)); r=j["roofline"]
        print(f.split("/")[-1], "%.1f q/s"%j["value"], "%.3f ms/step"%j["ms_per_step"], j["mode"], "frac %.3f"%r["frac"], "standalone %.3f"%r["standalone"]["frac"], r["kernel"])
    except Exception as e: print(f, "ERR", e)
P
