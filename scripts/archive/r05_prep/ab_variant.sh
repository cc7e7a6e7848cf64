#!/bin/bash
# GPU box, one call: parity of a variant build, then its kernels' durations against the product build's.
#   scripts/r05_prep/ab_variant.sh lazy_inverse
# Kernel durations of from_ntt / fold do not depend on where the database landed, so the UN-pipelined rocprofv3 kernel stats of
# two processes compare cleanly; the pipelined queries/s of two processes do not (placement, +-3-10 %) and are printed for
# orientation only, alternating A B A B.  Output: gpurun_out/r05_ab_<name>_*.
set -u
name=$1
R=$PWD
O=$R/gpurun_out
V=$R/sdk_amd/variants/libspiral_hip_$name.so
mkdir -p $O
[ -f "$V" ] || { echo "missing $V (scripts/r05_prep/build_variant.sh $name)"; exit 1; }
( SPIRAL_HIP_LIB=$V timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or from_ntt or fold or process_query_bytes or wave_fold or config_sweep" ) > $O/r05_ab_${name}_pytest.log 2>&1
tail -3 $O/r05_ab_${name}_pytest.log
grep -q " passed" $O/r05_ab_${name}_pytest.log && ! grep -q " failed" $O/r05_ab_${name}_pytest.log || { echo "variant parity FAILED: stopping"; exit 1; }
( SPIRAL_HIP_LIB=$V timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c2 and not sharded" ) > $O/r05_ab_${name}_fullsize.log 2>&1
tail -2 $O/r05_ab_${name}_fullsize.log
cd /tmp; export TMPDIR=/tmp
H="--headline-only --no-cpu-baseline"
for tag in base $name; do
  lib=$R/sdk_amd/libspiral_hip.so; [ $tag = base ] || lib=$V
  rm -rf /tmp/ab_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ab_$tag -o ab -- python $R/bench.py $H --steps 5 --warmup 2 > $O/r05_ab_${name}_${tag}_unpipelined.json 2> /tmp/ab_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/ab_$tag -name '*.db' | head -1)" $O/r05_ab_${name}_${tag}_unpipelined_kernel_stats.md > /dev/null 2>&1
  echo "== $tag (un-pipelined kernel stats)"; grep -E "k_from_sweep4|k_fold_wave|k_ntt_inv|k_fold_fused" $O/r05_ab_${name}_${tag}_unpipelined_kernel_stats.md | head -6
done
cd $R
for rep in 1 2; do
  for tag in base $name; do
    lib=$R/sdk_amd/libspiral_hip.so; [ $tag = base ] || lib=$V
    SPIRAL_HIP_LIB=$lib timeout 200 python bench.py $H --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag rep $rep: %.2f q/s, sweep in situ %.3f ms' % (d['value'], d['roofline']['ms_per_launch']))"
    SPIRAL_HIP_LIB=$lib timeout 200 python bench.py $H --batch 16 --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag rep $rep: batch16 %.1f q/s' % d['value'])"
  done
done
