#!/bin/bash
# Round 5, GPU call 5: k_fold_wave8 (two fold steps per eight-wave workgroup, split by modulus and row) -- parity on the GPU, then
# in-process A/B against k_fold_wave on one database allocation, then kernel durations of the un-pipelined query for both.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave_fold or fused_fold or fold_pack or process_query_bytes or sparse or ring_sweep or batch" ) > $O/r05c5_pytest.log 2>&1
tail -2 $O/r05c5_pytest.log
grep -q " passed" $O/r05c5_pytest.log && ! grep -q " failed\| error" $O/r05c5_pytest.log || { echo "parity FAILED"; tail -30 $O/r05c5_pytest.log; exit 1; }
timeout 400 python scripts/r05/ab.py fold_variant=5 fold_variant=6 fold_variant=5 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r05c5_ab_fold_wave8.txt
cd /tmp; export TMPDIR=/tmp
H="--headline-only --no-cpu-baseline"
for v in 5 6; do
  rm -rf /tmp/f$v
  SPIRAL_FOLD_VARIANT=$v SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/f$v -o f -- python $R/bench.py $H --steps 5 --warmup 2 > $O/r05c5_variant${v}_unpipelined.json 2> /tmp/f$v.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/f$v -name '*.db' | head -1)" $O/r05c5_variant${v}_unpipelined_kernel_stats.md > /dev/null 2>&1
  echo "== fold_variant $v"; grep -E "k_from_sweep4|k_fold_wave" $O/r05c5_variant${v}_unpipelined_kernel_stats.md | cut -c1-50,95-200
done
