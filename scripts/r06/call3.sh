#!/bin/bash
# r06 call 3: the group expansion (shared launches per round) against sixteen separate chains, in one process on one database;
# thresholds of the one-launch rounds; then parity of the batch tests on gfx950
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_call3; mkdir -p $O
ONLY_BATCH=1 BATCH=16,8,4 timeout 600 python scripts/r06/ab.py expand_group=0 expand_group_round_min=1024 expand_group_round_min=2048 expand_group_round_min=8192 expand_group_round_min=16384 expand_group_round_min=1099511627776 expand_group=0 2>&1 | grep -v amdgpu.ids | tee $O/group_expand_ab_raw.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or request or concurrent" 2>&1 | tail -4 | tee $O/pytest_batch.log
