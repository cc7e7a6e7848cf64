# minimal final profile set of one box: bench line, rocprofv3 kernel stats + per-query timeline of the same command
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
bash $R/scripts/box_fingerprint.sh 2>&1 | grep "Unique ID" > $O/r02f_box.txt
timeout 60 python $R/bench.py --steps 20 --warmup 5 > $O/r02f_bench_c2.json 2> $O/r02f_bench.err
timeout 70 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
python $R/scripts/rocprof_summary.py $DB $O/r02f_c2_kernel_stats.md > /dev/null 2>&1
python $R/scripts/timeline_full.py $DB 12 > $O/r02f_c2_query_timeline.txt 2>&1
head -8 $O/r02f_c2_kernel_stats.md; tail -c 300 $O/r02f_bench_c2.json
