#!/bin/bash
# Round 5, GPU call 38: with the large rounds in one launch, is the split expansion (odd subtree beside the first sweep) still the
# better schedule for the pipelined single query?  In-process A/B on one C2 allocation.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ONLY_SINGLE=1 STEPS=24 timeout 500 python scripts/r05/ab.py expand_split=0 expand_split=1 expand_split=0 expand_split=0,expand_round_min=512 expand_split=0,expand_round_min=4096 2>&1 | grep -v "^$" | tee $O/r05c38_ab_raw.txt
