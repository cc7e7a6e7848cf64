#!/bin/bash
# r06 call 10: the batched per-plane pipeline on disjoint CU sets: pass CUs 0 (off) / 32 / 48 / 64 / 80 / 96 / 128, parity on gfx950
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "disjoint_cus or two_query_tiles" 2>&1 | tail -3 | tee $O/pytest.log
ONLY_BATCH=1 BATCH=16,12 timeout 900 python scripts/r06/ab.py batch_pass_cus=0 batch_pass_cus=32 batch_pass_cus=48 batch_pass_cus=64 batch_pass_cus=80 batch_pass_cus=96 batch_pass_cus=128 batch_pass_cus=0 2>&1 | grep -v amdgpu.ids | tee $O/cu_split_ab_raw.txt
