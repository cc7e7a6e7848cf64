#!/bin/bash
# Round 5, GPU call 1: time what round 4 left un-timed, on ONE box.
#   base                     = HEAD (fold spill fix in)
#   fold_rr_old_scope        = HEAD with the fold kernel as it was before the spill fix
#   lazy_inverse             = HEAD + inverse transforms without the per-butterfly halving (all forms)
#   lazy_inverse_cooperative = the same, cooperative transforms only
# Kernel durations of the UN-pipelined query do not depend on where the database landed: they are the A/B.  The pipelined and
# 16-query lines of separate processes do (placement lottery) and are printed alternating, with the stand-alone sweep beside
# the in-situ one.  Per-kernel FETCH_SIZE / WRITE_SIZE of HEAD (the fold's write amplification after the fix).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
VARS="fold_rr_old_scope lazy_inverse lazy_inverse_cooperative"
bash scripts/box_fingerprint.sh > $O/r05c1_box.txt 2>&1
for v in lazy_inverse lazy_inverse_cooperative; do
  V=$R/sdk_amd/variants/libspiral_hip_$v.so
  ( SPIRAL_HIP_LIB=$V timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or from_ntt or fold or process_query_bytes or wave_fold" ) > $O/r05c1_${v}_pytest.log 2>&1
  echo "parity $v: $(tail -1 $O/r05c1_${v}_pytest.log)"
done
cd /tmp; export TMPDIR=/tmp
H="--headline-only --no-cpu-baseline"
for tag in base $VARS; do
  lib=$R/sdk_amd/libspiral_hip.so; [ $tag = base ] || lib=$R/sdk_amd/variants/libspiral_hip_$tag.so
  rm -rf /tmp/ab_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/ab_$tag -o ab -- python $R/bench.py $H --steps 5 --warmup 2 > $O/r05c1_${tag}_unpipelined.json 2> /tmp/ab_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/ab_$tag -name '*.db' | head -1)" $O/r05c1_${tag}_unpipelined_kernel_stats.md > /dev/null 2>&1
  echo "== $tag (un-pipelined kernel stats)"; grep -E "k_from_sweep4|k_fold_wave|k_ntt_inv|k_fold_fused" $O/r05c1_${tag}_unpipelined_kernel_stats.md | cut -c1-60,100-200 | head -5
done
rm -rf /tmp/q2 /tmp/q3
SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q2 -o q2 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 1 > /tmp/q2.log 2>&1
SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q3 -o q3 -- python $R/bench.py $H --steps 1 --warmup 0 --sweep-iters 1 > /tmp/q3.log 2>&1
python $R/scripts/pmc_per_kernel.py "$(find /tmp/q2 -name '*.db' | head -1)" "$(find /tmp/q3 -name '*.db' | head -1)" > $O/r05c1_base_pmc_per_kernel_unpipelined.md 2>&1
head -12 $O/r05c1_base_pmc_per_kernel_unpipelined.md
cd $R
for rep in 1 2; do
  for tag in base $VARS; do
    lib=$R/sdk_amd/libspiral_hip.so; [ $tag = base ] || lib=$R/sdk_amd/variants/libspiral_hip_$tag.so
    SPIRAL_HIP_LIB=$lib timeout 150 python bench.py $H --steps 20 --warmup 5 2>/dev/null | tee $O/r05c1_${tag}_pipelined_rep$rep.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('stage_ms') or {}
print('$tag rep $rep: %.2f q/s  sweep in situ %.3f alone %.3f (x%.3f)  stages %s' % (d['value'], r['ms_per_launch'], r['standalone']['ms_per_launch'], r['ms_per_launch']/r['standalone']['ms_per_launch'], {k: round(v,3) for k,v in s.items()} if isinstance(s, dict) else s))"
    [ $rep = 1 ] && SPIRAL_HIP_LIB=$lib timeout 150 python bench.py $H --batch 16 --steps 4 --warmup 1 2>/dev/null | tee $O/r05c1_${tag}_batch16.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag: batch16 %.1f q/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
  done
done
