# VALU busy / clock of the fused fold kernels, one un-pipelined C2 query per variant
cd /tmp; export TMPDIR=/tmp
R=/root/repo
export SPIRAL_PIPELINE=0
for v in ${FOLD_VARIANTS:-3 5}; do
  export SPIRAL_FOLD_VARIANT=$v
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" "SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LEVEL_WAVES" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    i=$((i+1))
    rm -rf /tmp/g$v$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/g$v$i -o g -- python $R/bench.py --steps 1 --warmup 0 --sweep-iters 1 --no-cpu-baseline > /tmp/g$v$i.log 2>&1
    python - "$(find /tmp/g$v$i -name '*.db' | head -1)" $v <<'PY'
import sqlite3, sys
try:
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name"))
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    dur = dict(c.execute(f"select s.kernel_name, sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name"))
except Exception as e:
    print("no counters", e); rows = []; dur = {}
for k, cn, n, sm in rows:
    ks_ = k.split('(')[0].replace('void spiral::', '').replace('spiral::', '')
    if ks_.startswith(('k_fold_fused2', 'k_fold_wave')):
        d = [v for kk, v in dur.items() if kk.split('(')[0].endswith(ks_.split('<')[0]) or ks_.split('<')[0] in kk]
        print("| %s | %s | %d | %s | %.4g | kernel ns %.4g |" % (sys.argv[2], ks_, n, cn, sm, d[0] if d else 0))
PY
  done
done
