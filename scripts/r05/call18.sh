#!/bin/bash
# Round 5, GPU call 18: the cooperative forward transform with the lazy five-instruction butterfly (k_ntt_fwd / k_ntt_fwd3 of the
# expansion, the fold tails and pack; database encoders) -- parity, then kernel durations against the previous build
# (sdk_amd/variants/libspiral_hip_base.so) in the un-pipelined query, query lists on the narrow databases, 16-query steps.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sparse_bucket.py tests/test_golden_vectors.py -x -q -m gpu ) > $O/r05c18_pytest.log 2>&1
tail -2 $O/r05c18_pytest.log
grep -q " passed" $O/r05c18_pytest.log && ! grep -q " failed\| error" $O/r05c18_pytest.log || { echo "parity FAILED"; tail -40 $O/r05c18_pytest.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
H="--headline-only --no-cpu-baseline"
for tag in base lazyfwd; do
  lib=$R/sdk_amd/libspiral_hip.so; [ $tag = base ] && lib=$R/sdk_amd/variants/libspiral_hip_base.so
  rm -rf /tmp/ab_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/ab_$tag -o ab -- python $R/bench.py $H --steps 5 --warmup 2 > $O/r05c18_${tag}_unpipelined.json 2> /tmp/ab_$tag.err
  python $R/scripts/rocprof_summary.py "$(find /tmp/ab_$tag -name '*.db' | head -1)" $O/r05c18_${tag}_unpipelined_kernel_stats.md > /dev/null 2>&1
  echo "== $tag (un-pipelined kernel stats)"; grep -E "k_ntt_fwd|k_ntt_inv|k_mac" $O/r05c18_${tag}_unpipelined_kernel_stats.md | cut -c1-40,85-200 | head -6
done
cd $R
for rep in 1 2; do for tag in base lazyfwd; do
  lib=$R/sdk_amd/libspiral_hip.so; [ $tag = base ] && lib=$R/sdk_amd/variants/libspiral_hip_base.so
  for cfgb in "c1 8" "p2 8" "c1 1"; do set -- $cfgb
    SPIRAL_HIP_LIB=$lib timeout 150 python bench.py $H --config $1 --batch $2 --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag rep $rep $1 batch $2: %.1f q/s %.3f ms/step %s' % (d['value'], d['ms_per_step'], d['config'].get('stage_ms')))"
  done
  SPIRAL_HIP_LIB=$lib timeout 150 python bench.py $H --batch 16 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag rep $rep c2 batch 16: %.1f q/s %.2f ms/step' % (d['value'], d['ms_per_step']))"
  SPIRAL_HIP_LIB=$lib timeout 150 python bench.py $H --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag rep $rep c2 single: %.2f q/s %.3f ms/step %s' % (d['value'], d['ms_per_step'], d['config'].get('stage_ms')))"
done; done
