#!/bin/bash
# r06 call 12: one rank's critical path of the row-sharded query with and without the split expansion; full-size loopback parity
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call12; mkdir -p $O
for v in 0 1 0 1; do
  echo "expand_split_shards=$v"
  SPIRAL_EXPAND_SPLIT_SHARDS=$v timeout 400 python scripts/r05/rank_critical_path.py c2 8 4 2>/dev/null | tee -a $O/rank_critical_path_split$v.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  G=%d serial %.3f ms pipelined %.3f ms sweep span %.3f tail %.3f' % (d['G'], d['serial_ms_per_query'], d['pipelined_ms_per_query'], d['sweep_span_ms'], d['tail_fold_gather_ms']))"
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "row_sharded or sharded or distributed_fold or c3_full" 2>&1 | tail -3 | tee $O/pytest.log
