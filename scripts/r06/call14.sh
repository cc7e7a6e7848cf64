#!/bin/bash
# r06 call 14: soak of the batched entry point at full size
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call14; mkdir -p $O
timeout 1500 python scripts/r06/soak_batch.py ${SOAK_ITERS:-12} 2>&1 | grep -v amdgpu.ids | tee $O/soak_batch.txt | tail -14
