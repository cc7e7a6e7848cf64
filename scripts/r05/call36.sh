#!/bin/bash
# Round 5, GPU call 36: k_expand_round thresholds with the odd subtree of a split expansion kept on the three-launch rounds.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
STEPS=16 timeout 600 python scripts/r05/ab.py expand_round_min=1073741824 expand_round_min=512 expand_round_min=128 expand_round_odd=1 expand_round_min=1073741824 2>&1 | grep -v "^$" | tee $O/r05c36_ab_raw.txt
