# per-kernel times of one un-pipelined C2 query per fused-fold variant (rocprofv3 kernel trace)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
export SPIRAL_PIPELINE=0
for v in ${FOLD_VARIANTS:-3 5}; do
  export SPIRAL_FOLD_VARIANT=$v
  rm -rf /tmp/k$v
  timeout 300 rocprofv3 --kernel-trace -d /tmp/k$v -o k -- python $R/bench.py --steps 4 --warmup 1 --sweep-iters 1 --no-cpu-baseline > /tmp/k$v.log 2>&1
  python - "$(find /tmp/k$v -name '*.db' | head -1)" $v <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
for k, n, tot, mn, mx in rows[:14]:
    k = k.split('(')[0].replace('void spiral::', '').replace('spiral::', '')
    print("| %s | %s | %d | %.3f ms | min %.1f us | max %.1f us |" % (sys.argv[2], k[:40], n, tot / 1e6, mn / 1e3, mx / 1e3))
PY
done
