#!/bin/bash
# round 4: A/B of the fold kernel's parking layout (fold_park) on one allocation, three repetitions each way, after a quick
# parity stage of the fold tests with the new default
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wave_fold or fused_fold_kernel or test_process_query_bytes_and_decode" ) > gpurun_out/r04_call9_stage.txt 2>&1 || { tail -30 gpurun_out/r04_call9_stage.txt; echo STOP; exit 1; }
tail -n 1 gpurun_out/r04_call9_stage.txt
STEPS=20 timeout 300 python scripts/r04_ab.py fold_park=0 fold_park=1 fold_park=0 fold_park=1 fold_park=0 > gpurun_out/r04_call9_ab.txt 2> gpurun_out/r04_call9_ab.err || tail -5 gpurun_out/r04_call9_ab.err
cat gpurun_out/r04_call9_ab.txt
