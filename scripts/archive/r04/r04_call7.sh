#!/bin/bash
# round 4, seventh GPU call: is the exact-sum fold reduction (fold_sum64) behind the larger in-situ loss of the sweep seen on
# the last boxes?  In-process A/B, three repetitions each way, whatever box comes up.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
STEPS=20 timeout 400 python scripts/r04_ab.py fold_sum64=0 fold_sum64=1 fold_sum64=0 fold_sum64=1 fold_sum64=0 > gpurun_out/r04_call7_ab.txt 2> gpurun_out/r04_call7_ab.err || tail -5 gpurun_out/r04_call7_ab.err
cat gpurun_out/r04_call7_ab.txt
