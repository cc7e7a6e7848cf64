#!/bin/bash
# Round 5, GPU call 32: alternating builds: k_from_sweep4 / k_fold_wave durations and the un-pipelined fold stage.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3 4; do for tag in ${TAGS:-new_a new_e}; do
  lib=$R/sdk_amd/variants/libspiral_hip_$tag.so
  rm -rf /tmp/fw_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/fw_$tag -o fw -- python $R/bench.py --headline-only --no-cpu-baseline --steps 8 --warmup 2 > /tmp/fw_$tag.json 2> /tmp/fw_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/fw_$tag -name '*.db' | head -1)" /tmp/fw_$tag.md > /dev/null 2>&1
  echo "$tag rep $rep: fold_wave $(grep -E 'k_fold_wave' /tmp/fw_$tag.md | awk -F'|' '{print $5}') us  from_sweep4 $(grep -E 'k_from_sweep4' /tmp/fw_$tag.md | awk -F'|' '{print $5}') us  $(python -c "
import json;d=json.loads(open('/tmp/fw_$tag.json').read().strip().splitlines()[-1]);print('%.2f q/s fold stage %.3f ms' % (d['value'], d['config']['stage_ms']['fold']))")"
done; done 2>&1 | tee $O/r05c32_${RAW:-raw}.txt
