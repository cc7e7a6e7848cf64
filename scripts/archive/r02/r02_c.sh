# split-expansion check: parity subset, then C2 / C1 / P2 bench lines with and without SPIRAL_EXPAND_SPLIT, and a timeline
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
cd /tmp
for split in 0 1; do
  for c in c2 c1 p2; do
    SPIRAL_EXPAND_SPLIT=$split timeout 300 python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$split', '$c', round(j['value'],1), 'q/s', round(j['ms_per_step'],3), 'ms', j.get('stage_ms'))"
  done
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pc -o pc -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /tmp/pc.log 2>&1
python $R/scripts/timeline_full.py $(find /tmp/pc -name "*.db" | head -1) 6 > $O/r2c_timeline_full.txt 2>&1
tail -3 $O/r2c_timeline_full.txt
