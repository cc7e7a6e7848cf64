#!/bin/bash
# Round 5, GPU call 27: kernel durations of the forward-transform launches of a C2 query, wave-per-transform against cooperative.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for tag in wave coop; do
  min=2048; [ $tag = coop ] && min=1073741824
  rm -rf /tmp/fw_$tag
  SPIRAL_FWD_WAVE_MIN=$min SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/fw_$tag -o fw -- python $R/bench.py --headline-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> /tmp/fw_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/fw_$tag -name '*.db' | head -1)" $O/r05c27_${tag}_kernel_stats.md > /dev/null 2>&1
  python $R/scripts/trace_dump.py "$(find /tmp/fw_$tag -name "*.db" | head -1)" $O/r05c27_${tag}_trace.tsv > /dev/null 2>&1
  echo "== $tag"; grep -E "k_ntt_fwd|k_mac|k_ntt_inv" $O/r05c27_${tag}_kernel_stats.md | cut -c1-60,85-200 | head -8
done
