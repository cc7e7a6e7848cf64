#!/bin/bash
# round 4, fifth GPU call: the batched pass with TWO query tiles (16 queries per database pass) in the microbenchmark
cd "${GRAFT_REPO_ROOT:-.}/scripts/ubench"
mkdir -p ../../gpurun_out
MFMA_UBENCH_B16=1 MFMA_UBENCH_NORATE=1 timeout 200 ./mfma_sweep 2048 3 > ../../gpurun_out/r04_call5_mfma16.txt 2>&1
grep -E "mfma|verify|MISMATCH" ../../gpurun_out/r04_call5_mfma16.txt | head -40
