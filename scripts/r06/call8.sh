#!/bin/bash
# r06 call 8: k_from_sweep4 with loads that bypass the L1 (from_sweep_nt), in-process A/B + kernel durations; the batched steps
# with the AVX2 keystream
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call8; mkdir -p $O
BATCH=16 STEPS=10 timeout 600 python scripts/r06/ab.py from_sweep_nt=1 from_sweep_nt=0 from_sweep_nt=1 2>&1 | grep -v amdgpu.ids | tee $O/from_sweep_nt_ab_raw.txt
cd /tmp
for v in 0 1; do
  rm -rf /tmp/p5
  SPIRAL_FROM_SWEEP_NT=$v SPIRAL_PIPELINE=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o p5 -- python $R/bench.py --headline-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> /tmp/p5.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/p5 -name '*.db' | head -1)" $O/unpipelined_kernel_stats_nt$v.md > /dev/null 2>&1
  echo "from_sweep_nt=$v"; grep -E "k_from_sweep4|k_fold_wave" $O/unpipelined_kernel_stats_nt$v.md
done
