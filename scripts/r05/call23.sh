#!/bin/bash
# Round 5, GPU call 23: the remaining pipelining / sweep switches re-measured with the round's kernels, one C2 allocation.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ONLY_SINGLE=1 STEPS=20 timeout 500 python scripts/r05/ab.py pipe_ring_wgs=2 pipe_ring=4 fold_variant=3 expand_split=0 expand_split=1 sweep_nt_store=0 from_sweep_xcd=0 2>&1 | grep -v "^$" | tee $O/r05c23_ab_raw.txt
