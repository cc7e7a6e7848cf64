# box fingerprint, fresh-handle A/B (all mitigations off / default), the whole gpu suite the way the driver runs it, bench + rocprof
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=$(date +%s)
L=$O/r02k_box_$T.log
bash scripts/box_fingerprint.sh > $L 2>&1
for t in "h2d_cache_sync=0,alloc_zero=0,ws_prealloc=0" "h2d_cache_sync=1" "h2d_cache_sync=0,alloc_zero=0,ws_prealloc=0" "h2d_cache_sync=0,alloc_zero=0,ws_prealloc=0,query_cache_sync=1" "h2d_cache_sync=1"; do
  echo "=== $t" >> $L
  ( DIAG_TUNABLES=$t timeout 120 python scripts/diag_free.py ) >> $L 2>&1
done
grep -E "Unique ID|===|F'|failing|resident:|export|again" $L | cut -c1-300 | tail -40
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02k_pytest_$T.log 2>&1
tail -14 $O/r02k_pytest_$T.log
( timeout 300 python bench.py ) > $O/r02k_bench_c2_$T.json 2> $O/r02k_bench.err
python - <<P
import json
j=json.load(open("$O/r02k_bench_c2_$T.json")); r=j["roofline"]
print("%.1f q/s %.3f ms/step frac %.3f standalone %.3f" % (j["value"], j["ms_per_step"], r["frac"], r["standalone"]["frac"]), j["config"].get("stage_ms"))
P
