# effective shader clock of every sweep launch of a few C2 queries: GRBM_GUI_ACTIVE / 8 XCDs / duration
cd /tmp; export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/sc
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/sc -o sc -- python $R/bench.py --steps 3 --warmup 1 --sweep-iters 2 --no-cpu-baseline > /tmp/sc.log 2>&1
python - "$(find /tmp/sc -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
rows = list(c.execute(f"select d.id, s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
try:
    vals = dict(c.execute("select dispatch_id, sum(value) from counters_collection group by dispatch_id"))
except Exception as e:
    print("counters_collection:", e, [r[1] for r in c.execute("pragma table_info(counters_collection)")]); vals = {}
for i, k, st, en in rows:
    if "sweep_packed_persist" in k or "fold_wave" in k and (en - st) > 300000:
        v = vals.get(i)
        print("%-28s dur %.3f ms  clock %.2f GHz" % (k.split('(')[0][-28:], (en - st) / 1e6, (v / 8 / (en - st)) if v else -1))
PY
