#!/bin/bash
# round 4, second GPU call: the pruned library + the persistent expansion program + the prefetching from_ntt kernel.
# Targeted tests first (a trap in the new persistent kernel must not cost the whole call), then the suite, then A/B lines.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out/r04_call2
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "expansion_variants or test_process_query or smoke" ) > ${O}_pytest_expand.txt 2>&1
tail -3 ${O}_pytest_expand.txt
( time timeout 1000 python -m pytest tests -m gpu -x -q --durations=12 ) > ${O}_pytest.txt 2>&1
tail -3 ${O}_pytest.txt
B="python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline"
for cfg in c2 c1 p2; do
  timeout 300 $B --config $cfg > ${O}_${cfg}_default.json 2>> ${O}_bench.err
  SPIRAL_EXPAND_PERSIST=0 timeout 300 $B --config $cfg > ${O}_${cfg}_launches.json 2>> ${O}_bench.err
  SPIRAL_PROGRAM_WGS=4 timeout 300 $B --config $cfg > ${O}_${cfg}_wgs4.json 2>> ${O}_bench.err
done
SPIRAL_FROM_SWEEP_PIPE=0 timeout 300 $B --config c2 > ${O}_c2_nopipe.json 2>> ${O}_bench.err
SPIRAL_FROM_SWEEP_PIPE=0 timeout 300 $B --config c2 --batch 8 > ${O}_c2_b8_nopipe.json 2>> ${O}_bench.err
timeout 300 $B --config c2 --batch 8 > ${O}_c2_b8_default.json 2>> ${O}_bench.err
timeout 300 $B --config c1 --batch 8 > ${O}_c1_b8_default.json 2>> ${O}_bench.err
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > ${O}_bench_full.json 2>> ${O}_bench.err
for f in ${O}_*.json; do echo "$f $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],2), d['config']['stage_ms'])" 2>&1 | tail -1)"; done
