#!/bin/bash
# Round 5, GPU call 37: 16- and 8-query steps with k_expand_round at several thresholds, repeated (in-process, one allocation).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ONLY_BATCH=1 BATCH=16 timeout 600 python scripts/r05/ab.py expand_round_min=1073741824 expand_round_min=1024 expand_round_min=4096 expand_round_min=1073741824 expand_round_min=1024 expand_round_min=4096 2>&1 | grep -v "^$" | tee $O/r05c37_ab_raw.txt
ONLY_BATCH=1 BATCH=8 timeout 600 python scripts/r05/ab.py expand_round_min=1073741824 expand_round_min=1073741824 2>&1 | grep -v "^$" | tee -a $O/r05c37_ab_raw.txt
