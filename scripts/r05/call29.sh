#!/bin/bash
# Round 5, GPU call 29: k_fold_wave with the table image staged by plain copies and the compose phase on four waves (A: ct_i words
# requested before the inverse transform, B: after it) against the previous build: kernel durations, un-pipelined C2 query.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold or process_query_bytes" ) > $O/r05c29_pytest.log 2>&1
tail -1 $O/r05c29_pytest.log
grep -q " passed" $O/r05c29_pytest.log && ! grep -q " failed\| error" $O/r05c29_pytest.log || { echo "parity FAILED"; tail -40 $O/r05c29_pytest.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
for tag in base new_a new_b base new_a; do
  lib=$R/sdk_amd/variants/libspiral_hip_$tag.so
  rm -rf /tmp/fw_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/fw_$tag -o fw -- python $R/bench.py --headline-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> /tmp/fw_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/fw_$tag -name '*.db' | head -1)" $O/r05c29_${tag}_kernel_stats.md > /dev/null 2>&1
  echo "== $tag: $(grep -E 'k_fold_wave' $O/r05c29_${tag}_kernel_stats.md | cut -c1-30,60-200 | head -1)"
done 2>&1 | tee $O/r05c29_raw.txt
