#!/bin/bash
# round 4, sixth GPU call: parity of the two-tile (16 queries per pass) matrix-core sweep in the library, staged
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out/r04_call6
( time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_query_tiles or matrix_core or batch_lds_staged or test_process_query_batch" ) > ${O}_stage1.txt 2>&1 || { tail -40 ${O}_stage1.txt; echo STOP1; exit 1; }
tail -2 ${O}_stage1.txt
( time timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "batch16 or batch8" ) > ${O}_stage2.txt 2>&1 || { tail -40 ${O}_stage2.txt; echo STOP2; exit 1; }
tail -2 ${O}_stage2.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustained 0 > ${O}_bench.json 2> ${O}_bench.err || { tail -5 ${O}_bench.err; echo STOP3; exit 1; }
python - ${O}_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline %.2f q/s" % d["value"], d["config"]["stage_ms"])
for k, v in d["secondary"].items():
    if isinstance(v, dict):
        print(k, v.get("value"), v.get("ms_per_step"), v.get("batch_selfcheck"), (v.get("batched_pass") or {}).get("ms_per_pass"))
PY
