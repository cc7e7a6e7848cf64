#!/bin/bash
# Round 5, GPU call 25: does the ordering event itself cost a one-at-a-time query anything?  In-process A/B.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ONLY_SINGLE=1 STEPS=20 timeout 500 python scripts/r05/ab.py sweep_serialize=0 sweep_serialize=1 sweep_serialize=0 2>&1 | grep -v "^$" | tee $O/r05c25_ab_raw.txt
