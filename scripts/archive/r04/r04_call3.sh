#!/bin/bash
# round 4, third GPU call: staged, fail-fast (call 2 lost 40 minutes to a GPU wedged by a memory fault: every later process
# then sat in its timeout).  Each new piece is switched on in its own short pytest process; the first failing stage ends the
# call.  Then the full suite with the defaults, then A/B bench lines.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out/r04_call3
OFF="SPIRAL_EXPAND_PERSIST=0 SPIRAL_FINISH_PERSIST=0 SPIRAL_FROM_SWEEP_PIPE=0"
stage() {  # name, env assignments, -k expression
  local name=$1 envs=$2 expr=$3 tmo=${4:-240}
  ( time env $envs timeout $tmo python -m pytest tests/test_gpu_parity.py tests/test_sparse_bucket.py tests/test_request_layer.py -m gpu -x -q -k "$expr" ) > ${O}_stage_${name}.txt 2>&1
  local rc=$?
  echo "stage $name rc=$rc: $(grep -E 'passed|failed|error' ${O}_stage_${name}.txt | tail -1)"
  if [ $rc -ne 0 ]; then tail -25 ${O}_stage_${name}.txt; echo "STOP at stage $name"; exit 1; fi
}
stage s0_off "$OFF" "test_process_query_bytes_and_decode or wave_fold or fused_fold_kernel or overlapped_fold_many_planes"
stage s1_pipe "SPIRAL_EXPAND_PERSIST=0 SPIRAL_FINISH_PERSIST=0" "test_process_query_bytes_and_decode or overlapped_fold_many_planes or ring_sweep"
stage s2_expand "SPIRAL_FINISH_PERSIST=0" "expansion_variants or test_process_query_bytes_and_decode"
stage s3_finish "SPIRAL_EXPAND_PERSIST=0" "test_process_query_bytes_and_decode or config_sweep or sparse or private_read or pack or v1" 420
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=12 ) > ${O}_pytest.txt 2>&1
rc=$?
tail -4 ${O}_pytest.txt
if [ $rc -ne 0 ]; then tail -40 ${O}_pytest.txt; echo "STOP: full suite rc=$rc"; exit 1; fi
B="python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline"
run() { # name, envs, args
  env $2 timeout 150 $B $3 > ${O}_$1.json 2>> ${O}_bench.err || { echo "bench $1 FAILED rc=$?"; tail -5 ${O}_bench.err; exit 1; }
  python - "$1" ${O}_$1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-16s %9.2f q/s  %s" % (sys.argv[1], d["value"], {k: round(v, 3) for k, v in d["config"]["stage_ms"].items()}))
PY
}
run c2_default "X=1" "--config c2"
run c2_off "$OFF" "--config c2"
run c2_expand_only "SPIRAL_FINISH_PERSIST=0 SPIRAL_FROM_SWEEP_PIPE=0" "--config c2"
run c2_pipe_only "SPIRAL_FINISH_PERSIST=0 SPIRAL_EXPAND_PERSIST=0" "--config c2"
run c1_default "X=1" "--config c1"
run c1_off "$OFF" "--config c1"
run c1_wgs4 "SPIRAL_PROGRAM_WGS=4" "--config c1"
run c1_wgs1 "SPIRAL_PROGRAM_WGS=1" "--config c1"
run p2_default "X=1" "--config p2"
run p2_off "$OFF" "--config p2"
run c2_b8_default "X=1" "--config c2 --batch 8 --steps 5 --warmup 2"
run c2_b8_off "$OFF" "--config c2 --batch 8 --steps 5 --warmup 2"
run c1_b8_default "X=1" "--config c1 --batch 8"
( time timeout 300 python bench.py --steps 20 --warmup 5 ) > ${O}_bench_full.json 2>> ${O}_bench.err
python - ${O}_bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("full: %.2f q/s" % d["value"], d["config"]["stage_ms"], {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d["secondary"].items()}, d["cpu_baseline"]["value"])
PY
