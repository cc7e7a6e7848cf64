cd /tmp; export TMPDIR=/tmp
R=/root/repo
for pad in ${PADS:-0 20 60}; do
  rm -rf /tmp/pl$pad
  PAD_GB=$pad timeout 300 rocprofv3 --kernel-trace -d /tmp/pl$pad -o pl -- python $R/scripts/r02_plane_times.py > /tmp/pl$pad.log 2>&1
  tail -1 /tmp/pl$pad.log
  python - "$(find /tmp/pl$pad -name '*.db' | head -1)" $pad <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = [ (en - st) / 1e6 for k, st, en in c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start") if "sweep_packed_persist" in k]
per = [[], [], [], []]
for i, d in enumerate(rows): per[i % 4].append(d)
print("pad %s GB: " % sys.argv[2] + "  ".join("plane %d: %.3f ms (min %.3f)" % (i, sum(v) / len(v), min(v)) for i, v in enumerate(per)))
PY
done
