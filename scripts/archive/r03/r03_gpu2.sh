#!/bin/bash
# Round 3, GPU call 2: where the MFMA batched sweep's time goes (issue-rate probes, DIAG variants, SQ counters), batch
# parity tests through the library, bench --batch 8 with the matrix-core pass on / off.
set -u
O=$PWD/gpurun_out
mkdir -p $O
T=$(date +%s)
R=$PWD
cd scripts/ubench
( MFMA_UBENCH_SHORT=1 timeout 300 ./mfma_sweep 2048 3 ) > $O/r03b_mfma_diag_$T.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "mfma|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INSTS_VALU |SQ_ACTIVE_INST_VALU|SQ_INST_CYCLES" | head -40 > $O/r03b_counter_names_$T.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pm$i
  ( cd $R/scripts/ubench && MFMA_UBENCH_SHORT=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm$i -o f -- ./mfma_sweep 2048 1 ) > /tmp/pm$i.log 2>&1
  python - "$(find /tmp/pm$i -name '*.db' | head -1)" <<'PY' >> $O/r03b_mfma_pmc_$T.txt 2>&1
import sqlite3, sys
try:
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name"))
except Exception as e:
    print("no counters", e); rows = []
for k, cn, n, sm in rows:
    if 'k_sweep_mfma' in k or 'k_mfma_rate' in k:
        print("| %s | %d | %s | %.5g |" % (k.split('(')[0].replace('void spiral::', ''), n, cn, sm))
PY
done
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" > $O/r03b_pytest_batch_$T.log 2>&1
tail -5 $O/r03b_pytest_batch_$T.log
timeout 600 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03b_bench_batch8_mfma_$T.json 2> $O/r03b_bench_batch8_mfma_$T.err
SPIRAL_BATCH_MFMA=0 timeout 600 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03b_bench_batch8_valu_$T.json 2> $O/r03b_bench_batch8_valu_$T.err
SPIRAL_BATCH_PIPELINE=1 timeout 600 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03b_bench_batch8_mfma_pipe_$T.json 2> $O/r03b_bench_batch8_mfma_pipe_$T.err
cat $O/r03b_mfma_diag_$T.txt
cat $O/r03b_mfma_pmc_$T.txt
for f in $O/r03b_bench_batch8_*_$T.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
tail -3 $O/r03b_bench_batch8_mfma_$T.err
