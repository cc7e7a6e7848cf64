cd /tmp; export TMPDIR=/tmp
R=/root/repo
export SPIRAL_PIPELINE=0
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d /tmp/s1 -o s1 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 1 --no-cpu-baseline > /tmp/s1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d /tmp/s2 -o s2 -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 1 --no-cpu-baseline > /tmp/s2.log 2>&1
for d in s1 s2; do python - "$(find /tmp/$d -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    rows = list(c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name"))
except Exception as e:
    print("no counters", e); rows = []
for k, cn, n, sm in rows:
    k = k.split('(')[0].replace('void spiral::', '').replace('spiral::', '')
    if k.startswith(('k_fold_fused2', 'k_from_sweep4', 'k_sweep_packed_persist')):
        print("| %s | %d | %s | %.4g |" % (k, n, cn, sm))
PY
done
tail -3 /tmp/s1.log | cut -c1-200
