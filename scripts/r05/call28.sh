#!/bin/bash
# Round 5, GPU call 28: what a fold wave waits for -- k_fold_wave durations of TIMING-ONLY builds that each remove one thing
# (scripts/r05/build_timing_variants.py), un-pipelined C2 query under rocprofv3.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for tag in ${TAGS:-base no_operands no_prologue no_transpose all}; do
  lib=$R/sdk_amd/variants/libspiral_hip_tv_$tag.so; [ $tag = base ] && lib=$R/sdk_amd/libspiral_hip.so
  rm -rf /tmp/tv_run_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/tv_run_$tag -o tv -- python $R/bench.py --headline-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> /tmp/tv_run_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/tv_run_$tag -name '*.db' | head -1)" $O/r05c28_${tag}_kernel_stats.md > /dev/null 2>&1
  echo "== $tag: $(grep -E 'k_fold_wave' $O/r05c28_${tag}_kernel_stats.md | cut -c1-30,60-200 | head -2)"
done 2>&1 | tee $O/r05c28_${RAW:-raw}.txt
