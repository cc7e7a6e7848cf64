# SQ counters of the fused fold kernels, one un-pipelined C2 query per variant (SPIRAL_FOLD_VARIANT = 3 and 5)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
export SPIRAL_PIPELINE=0
for v in ${FOLD_VARIANTS:-3 5}; do
  export SPIRAL_FOLD_VARIANT=$v
  i=0
  for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
             "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
             "SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM"; do
    i=$((i+1))
    rm -rf /tmp/f$v$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/f$v$i -o f -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 1 --no-cpu-baseline > /tmp/f$v$i.log 2>&1
    python - "$(find /tmp/f$v$i -name '*.db' | head -1)" $v <<'PY'
import sqlite3, sys
try:
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name"))
except Exception as e:
    print("no counters", e); rows = []
for k, cn, n, sm in rows:
    k = k.split('(')[0].replace('void spiral::', '').replace('spiral::', '')
    if k.startswith(('k_fold_fused2', 'k_fold_wave')):
        print("| %s | %s | %d | %s | %.4g |" % (sys.argv[2], k, n, cn, sm))
PY
  done
done
