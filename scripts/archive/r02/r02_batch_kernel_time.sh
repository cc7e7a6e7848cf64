# duration of the batched sweep kernel (B = 8) in its one-wave-per-unit and split forms (rocprofv3 kernel trace)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
for split in ${SPLITS:-0 1}; do
  export SPIRAL_BATCH_SPLIT=$split
  rm -rf /tmp/bk$split
  timeout 300 rocprofv3 --kernel-trace -d /tmp/bk$split -o bk -- python $R/bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/bk$split.log 2>&1
  python - "$(find /tmp/bk$split -name '*.db' | head -1)" $split <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
for k, n, tot, mn, mx in c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc limit 4"):
    print("| split=%s | %s | %d | avg %.3f ms | min %.3f | max %.3f |" % (sys.argv[2], k[:60], n, tot / n / 1e6, mn / 1e6, mx / 1e6))
PY
done
