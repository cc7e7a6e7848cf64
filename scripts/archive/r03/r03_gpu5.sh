#!/bin/bash
# Round 3, GPU call 5: the full -m gpu suite, the default bench line (with the new cpu_baseline), batch-8 kernel stats,
# ref_check kit with the HIP path in the loop, library-level contiguous-allocation diagnosis.
set -u
O=$PWD/gpurun_out
R=$PWD
mkdir -p $O
T=$(date +%s)
bash scripts/box_fingerprint.sh 2>/dev/null | grep -i "unique" > $O/r03e_box_$T.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r03e_pytest_$T.log 2>&1
tail -4 $O/r03e_pytest_$T.log
timeout 900 python bench.py > $O/r03e_bench_c2_$T.json 2> $O/r03e_bench_c2_$T.err
python - <<PY
import json
d=json.load(open('$O/r03e_bench_c2_$T.json')); print('c2', d['value'], d['ms_per_step'], d['config']['stage_ms'], d['roofline']['frac'], d['roofline']['standalone']['ms_per_launch'], d['cpu_baseline'] and (d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sample'][:200]))
PY
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pb8
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb8 -o b8 -- python $R/bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03e_bench_batch8_prof_$T.json 2> /tmp/pb8.err
python $R/scripts/rocprof_summary.py "$(find /tmp/pb8 -name '*.db' | head -1)" $O/r03e_batch8_kernel_stats_$T.md > /dev/null 2>&1
head -12 $O/r03e_batch8_kernel_stats_$T.md
cd $R
timeout 600 python scripts/ref_check/dump_cases.py --gpu --c1 --out /tmp/ref_check > $O/r03e_ref_check_$T.txt 2>&1
tail -8 $O/r03e_ref_check_$T.txt
cp /tmp/ref_check/manifest.json $O/r03e_ref_check_manifest_$T.json
( SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2.py; SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2.py; SPIRAL_DB_CONTIGUOUS=0 timeout 300 python scripts/diag_c2.py ) > $O/r03e_diag_c2_contig_$T.txt 2>&1
cat $O/r03e_diag_c2_contig_$T.txt
cat $O/r03e_box_$T.txt
