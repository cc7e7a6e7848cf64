#!/bin/bash
# r06 call 19: every batch / request test with the wave expansion kernel forced onto every side of every round
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call19; mkdir -p $O
SPIRAL_EXPAND_WAVE_MIN_DIGITS=1 SPIRAL_EXPAND_GROUP_ROUND_MIN=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_request_layer.py tests/test_gpu_fullsize.py -m gpu -x -q -k "(batch or request) and not matrix_core_sweep" 2>&1 | tail -3 | tee $O/pytest_forced_wave.log
