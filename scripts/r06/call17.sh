#!/bin/bash
# r06 call 17: k_expand_wave on / off, alternating, six times each in one process
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call17; mkdir -p $O
ONLY_BATCH=1 BATCH=16,8 timeout 900 python scripts/r06/ab.py expand_wave_min_digits=0 expand_wave_min_digits=16 expand_wave_min_digits=0 expand_wave_min_digits=16 expand_wave_min_digits=0 expand_wave_min_digits=16 expand_wave_min_digits=0 expand_wave_min_digits=16 expand_wave_min_digits=0 expand_wave_min_digits=16 2>&1 | grep -v amdgpu.ids | tee $O/expand_wave_ab_raw.txt
