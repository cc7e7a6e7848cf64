#!/bin/bash
# r06 call 9: k_from_sweep8 against k_from_sweep4 (in-process A/B, un-pipelined kernel durations), the inverse kernel with 16-byte
# accesses (trace of a 16-query step), parity of the affected tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call9; mkdir -p $O
BATCH=16 STEPS=10 timeout 600 python scripts/r06/ab.py from_sweep8=0 from_sweep8=1 from_sweep8=0 2>&1 | grep -v amdgpu.ids | tee $O/from_sweep8_ab_raw.txt
cd /tmp
for v in 0 1; do
  rm -rf /tmp/p5
  SPIRAL_FROM_SWEEP8=$v SPIRAL_PIPELINE=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o p5 -- python $R/bench.py --headline-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> /tmp/p5.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/p5 -name '*.db' | head -1)" $O/unpipelined_kernel_stats_fs8_$v.md > /dev/null 2>&1
  echo "from_sweep8=$v"; grep -E "k_from_sweep|k_fold_wave|k_ntt_inv" $O/unpipelined_kernel_stats_fs8_$v.md
done
rm -rf /tmp/p7
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o p7 -- python $R/bench.py --headline-only --no-cpu-baseline --batch 16 --steps 4 --warmup 1 > $O/bench_c2_batch16_profiled.json 2> /tmp/p7.err
DB="$(find /tmp/p7 -name '*.db' | head -1)"
python $R/scripts/trace_dump.py "$DB" /tmp/p7.tsv; gzip -c /tmp/p7.tsv > $O/batch16_trace.tsv.gz
python $R/scripts/r06/step_phases.py /tmp/p7.tsv 2 --list > $O/batch16_step_phases.md 2>&1
cat $O/batch16_step_phases.md
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not c3 and not c4" 2>&1 | tail -4 | tee $O/pytest.log
