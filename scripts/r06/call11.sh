#!/bin/bash
# r06 call 11: only the pass confined to a CU set (folds on the queries' own un-masked streams)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call11; mkdir -p $O
ONLY_BATCH=1 BATCH=16,12 timeout 900 python scripts/r06/ab.py batch_pass_cus=0 batch_pass_cus=32 batch_pass_cus=64 batch_pass_cus=96 batch_pass_cus=128 batch_pass_cus=160 batch_pass_cus=192 batch_pass_cus=0 2>&1 | grep -v amdgpu.ids | tee $O/pass_masked_ab_raw.txt
