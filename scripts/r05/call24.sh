#!/bin/bash
# Round 5, GPU call 24: two single queries in flight with their sweeps ordered by the library (sp_db::sweeps_done): the headline's
# one-at-a-time value against `two_in_flight` in the same process, ordering on and off.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for ser in 1 0 1; do
  SPIRAL_SWEEP_SERIALIZE=$ser timeout 200 python bench.py --headline-only --no-cpu-baseline --two-in-flight --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['two_in_flight']; print('sweep_serialize=$ser: one at a time %.2f q/s (%.3f ms) | two in flight %.2f q/s (%.3f ms) %s' % (d['value'], d['ms_per_step'], t['value'], t['ms_per_query'], t['responses']))"
done 2>&1 | tee $O/r05c24_raw.txt
