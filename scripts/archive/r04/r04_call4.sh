#!/bin/bash
# round 4, fourth GPU call: in-process A/B of this round's two kernel changes on one C2 allocation (+ C1), after a quick
# parity stage of the rebuilt library (the phase program was removed since call 3)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out/r04_call4
( time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_process_query_bytes_and_decode or wave_fold or fused_fold_kernel or overlapped_fold_many_planes or expansion_variants" ) > ${O}_stage.txt 2>&1 || { tail -30 ${O}_stage.txt; echo STOP; exit 1; }
tail -2 ${O}_stage.txt
timeout 400 python scripts/r04_ab.py from_sweep_pipe=0 fold_sum64=0 from_sweep_pipe=0,fold_sum64=0 fused_min_pairs=128 fused_min_pairs=512 pipe_tail_defer=128 pipe_tail_defer=512 > ${O}_ab_c2.txt 2> ${O}_ab_c2.err || { tail -5 ${O}_ab_c2.err; }
cat ${O}_ab_c2.txt
