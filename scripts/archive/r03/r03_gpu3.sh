#!/bin/bash
# Round 3, GPU call 3: MFMA batched sweep -- corrected issue-rate probe, NB = 2 / 4 with DIAG variants, SQ + memory counters.
set -u
O=$PWD/gpurun_out
mkdir -p $O
T=$(date +%s)
R=$PWD
cd scripts/ubench
( MFMA_UBENCH_SHORT=1 timeout 300 ./mfma_sweep 2048 3 ) > $O/r03c_mfma_diag_$T.txt 2>&1
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" \
           "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_EA_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pm$i
  ( cd $R/scripts/ubench && MFMA_UBENCH_SHORT=1 MFMA_UBENCH_NORATE=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm$i -o f -- ./mfma_sweep 2048 1 ) > /tmp/pm$i.log 2>&1
  python - "$(find /tmp/pm$i -name '*.db' | head -1)" <<'PY' >> $O/r03c_mfma_pmc_$T.txt 2>&1
import sqlite3, sys
try:
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name"))
except Exception as e:
    print("no counters", e); rows = []
for k, cn, n, sm in rows:
    if 'k_sweep_mfma' in k or 'k_read' in k:
        print("| %s | %d | %s | %.5g |" % (k.split('(')[0].replace('void spiral::', ''), n, cn, sm))
PY
  tail -2 /tmp/pm$i.log >> $O/r03c_pmc_logs_$T.txt
done
cat $O/r03c_mfma_diag_$T.txt
cat $O/r03c_mfma_pmc_$T.txt
