#!/bin/bash
# Round 3, GPU call 4: matrix-core batch tests, bench --batch 8 (self-check + batched-pass roofline), contiguous-allocation
# reproducer incl. the fragmentation mode.
set -u
O=$PWD/gpurun_out
mkdir -p $O
T=$(date +%s)
bash scripts/box_fingerprint.sh 2>/dev/null | grep -i "unique" > $O/r03d_box_$T.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" > $O/r03d_pytest_batch_$T.log 2>&1
tail -5 $O/r03d_pytest_batch_$T.log
timeout 600 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03d_bench_batch8_$T.json 2> $O/r03d_bench_batch8_$T.err
python - <<PY
import json
d=json.load(open('$O/r03d_bench_batch8_$T.json')); print('batch8', d['value'], d['ms_per_step'], d['batch_selfcheck'], d['roofline'].get('batched_pass'))
PY
tail -3 $O/r03d_bench_batch8_$T.err
cd scripts/ubench
( timeout 250 ./contig_repro 56 96 2 1; echo "exit $?"; timeout 400 ./contig_repro 56 48 2 2; echo "exit $?" ) > $O/r03d_contig_repro_$T.txt 2>&1
grep -E "RESULT|exit|fragmented|allocation|changed" $O/r03d_contig_repro_$T.txt
cat $O/r03d_box_$T.txt
