cd /tmp; export TMPDIR=/tmp
R=/root/repo
python $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err
tail -c 600 $R/gpurun_out/bench_final.json
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p1 -- python $R/bench.py --no-cpu-baseline > /tmp/p1.log 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/p1 -name "*.db" | head -1) $R/gpurun_out/kernel_stats.md > /dev/null
python $R/scripts/timeline_pipe.py $(find /tmp/p1 -name "*.db" | head -1) 6 > $R/gpurun_out/timeline.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o p2 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 2 --no-cpu-baseline > /tmp/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p3 -o p3 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 2 --no-cpu-baseline > /tmp/p3.log 2>&1
python $R/scripts/pmc_traffic.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) $R/gpurun_out/pmc_traffic.json | tail -8
head -12 $R/gpurun_out/kernel_stats.md
