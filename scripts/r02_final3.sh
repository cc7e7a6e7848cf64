# the whole gpu suite the way the driver runs it, then the fresh-handle scenario loops, then bench
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02j_pytest.log 2>&1
tail -15 $O/r02j_pytest.log
( timeout 120 python scripts/diag_free.py ) > $O/r02j_diag_free.log 2>&1
grep -v amdgpu.ids $O/r02j_diag_free.log | tail -8
( timeout 300 python bench.py ) > $O/r02j_bench_c2.json 2> $O/r02j_bench.err
python - <<'P'
import json
j=json.load(open("/root/repo/gpurun_out/r02j_bench_c2.json")); r=j["roofline"]
print("%.1f q/s %.3f ms/step frac %.3f standalone %.3f" % (j["value"], j["ms_per_step"], r["frac"], r["standalone"]["frac"]), j["config"].get("stage_ms"))
P
