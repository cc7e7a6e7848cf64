# round-2 measurement batch B: CU-partition sweep, batch-8 kernel stats, full single-query timeline
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
timeout 600 python $R/scripts/r02_cu_split.py > $O/r2b_cu_split.txt 2> $O/r2b_cu_split.err
tail -12 $O/r2b_cu_split.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o pb -- python $R/bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pb.log 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/pb -name "*.db" | head -1) $O/r2b_batch8_kernel_stats.md > /dev/null
head -14 $O/r2b_batch8_kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pc -o pc -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /tmp/pc.log 2>&1
python $R/scripts/timeline_full.py $(find /tmp/pc -name "*.db" | head -1) 6 > $O/r2b_timeline_full.txt 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/pc -name "*.db" | head -1) $O/r2b_c2_kernel_stats.md > /dev/null
tail -3 $O/r2b_timeline_full.txt
