# kernel timeline of one query of config $1 (default c1): rocprofv3 kernel trace -> scripts/timeline_full.py
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; C=${1:-c1}
rm -rf /tmp/pt
timeout 300 rocprofv3 --kernel-trace -d /tmp/pt -o pt -- python $R/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline > /tmp/pt.log 2>&1
python $R/scripts/timeline_full.py $(find /tmp/pt -name "*.db" | head -1) 5 > $O/timeline_$C.txt 2>&1
tail -2 $O/timeline_$C.txt
