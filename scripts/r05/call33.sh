#!/bin/bash
# Round 5, GPU call 33: does the new fold kernel cost the sweep beside it more?  Pipelined single query, alternating builds:
# queries/s, sweep launches inside the query against the same launches alone (same process), exposed fold.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for rep in 1 2 3 4 5; do for tag in ${TAGS:-base final}; do
  lib=$R/sdk_amd/variants/libspiral_hip_$tag.so; [ $tag = final ] && lib=$R/sdk_amd/libspiral_hip.so
  SPIRAL_HIP_LIB=$lib timeout 150 python bench.py --headline-only --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['config']['stage_ms']
print('$tag rep $rep: %.2f q/s  sweep in situ %.3f alone %.3f (x%.3f)  expand %.3f fold %.3f' % (d['value'], r['ms_per_launch'], r['standalone']['ms_per_launch'], r['ms_per_launch']/r['standalone']['ms_per_launch'], s['expand'], s['fold']))"
done; done 2>&1 | tee $O/r05c33_raw.txt
