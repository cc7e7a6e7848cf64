#!/bin/bash
# Round 5, GPU call 22: the pipelining switches re-measured with the round's fold kernel (the spill fix changed what the fold costs
# the sweep beside it; pipe_tail_defer was last measured in round 4): in-process A/B on one C2 allocation.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ONLY_SINGLE=1 STEPS=20 timeout 500 python scripts/r05/ab.py pipe_tail_defer=128 pipe_tail_defer=512 pipe_tail_defer=64 pipe_tail_defer=1024 sweep_prio=0 pipe_tail_defer=128 2>&1 | grep -v "^$" | tee $O/r05c22_ab_raw.txt
