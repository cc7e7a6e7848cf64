#!/bin/bash
# Round 5, GPU call 4: the dead gadget digit (8-bit digits of a 56-bit Q: the top one of t_gsw = 8 is identically zero) skipped by
# the fused fold kernels -- parity on the GPU, then in-process A/B on one database allocation.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave_fold or fused_fold or fold_pack or process_query_bytes or sparse" ) > $O/r05c4_pytest.log 2>&1
tail -2 $O/r05c4_pytest.log
timeout 400 python scripts/r05/ab.py fold_skip_dead_digits=0 fold_skip_dead_digits=1 fold_skip_dead_digits=0 2>&1 | grep -v Warning | tee $O/r05c4_ab_dead_digits.txt
