#!/bin/bash
# Round 5, GPU call 17: the whole -m gpu suite on the tree with the planar pass; one rank's critical path of the row-sharded query
# at HEAD; the N > 1 code path of bench.py at world size 1 with its per-rank diagnostics.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 900 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/r05c17_pytest.log 2>&1
tail -4 $O/r05c17_pytest.log
timeout 300 python scripts/r05/rank_critical_path.py c2 8 4 2 2>/dev/null | tee $O/r05c17_rank_critical_path.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('G=%d: serial %.2f ms, pipelined list %.2f ms, sweep span %.2f (%.3f per plane), tail %.2f; implied %.0f / %.0f q/s' % (d['G'], d['serial_ms_per_query'], d['pipelined_ms_per_query'], d['sweep_span_ms'], d['sweep_ms_per_plane'], d['tail_fold_gather_ms'], d['implied_qps_serial'], d['implied_qps_pipelined']))"
SPIRAL_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/r05c17_bench_c2_dist1.json 2> $O/r05c17_bench_c2_dist1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05c17_bench_c2_dist1.json").read().strip().splitlines()[-1])
print("dist1: %.2f q/s mode %s selfcheck %s" % (d["value"], d["mode"], d["overlap_selfcheck"]))
print(json.dumps(d["per_rank"], indent=1)[:1800])
PY
