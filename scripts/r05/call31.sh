#!/bin/bash
# Round 5, GPU call 31: k_from_sweep4 with one thing removed at a time (timing-only builds), un-pipelined C2 query under rocprofv3.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for tag in ${TAGS:-base fs_no_store fs_no_load fs_no_inv base}; do
  lib=$R/sdk_amd/variants/libspiral_hip_tv_$tag.so; [ $tag = base ] && lib=$R/sdk_amd/libspiral_hip.so
  rm -rf /tmp/tv_run_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/tv_run_$tag -o tv -- python $R/bench.py --headline-only --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> /tmp/tv_run_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/tv_run_$tag -name '*.db' | head -1)" /tmp/tv_$tag.md > /dev/null 2>&1
  echo "$tag: from_sweep4 $(grep -E 'k_from_sweep4' /tmp/tv_$tag.md | awk -F'|' '{print $5}') us   fold_wave $(grep -E 'k_fold_wave' /tmp/tv_$tag.md | awk -F'|' '{print $5}') us"
done 2>&1 | tee $O/r05c31_${RAW:-raw}.txt
