# box fingerprint, the C2-after-small-configurations probe, the whole gpu suite the way the driver runs it, bench
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=$(date +%s)
L=$O/r02p_box_$T.log
bash scripts/box_fingerprint.sh > $L 2>&1
for e in "X=1" "SPIRAL_DB_CONTIGUOUS=1"; do
  echo "=== $e" >> $L
  ( env $e timeout 200 python scripts/diag_c2.py 2>&1 | grep -v amdgpu.ids ) >> $L 2>&1
done
grep -E "Unique ID|===|ok|rror" $L
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02p_pytest_$T.log 2>&1
tail -14 $O/r02p_pytest_$T.log
( timeout 300 python bench.py ) > $O/r02p_bench_c2_$T.json 2> $O/r02p_bench.err
python - <<P
import json
j=json.load(open("$O/r02p_bench_c2_$T.json")); r=j["roofline"]
print("%.1f q/s %.3f ms/step frac %.3f standalone %.3f" % (j["value"], j["ms_per_step"], r["frac"], r["standalone"]["frac"]), j["config"].get("stage_ms"))
P
