#!/bin/bash
set -u
O=$PWD/gpurun_out; mkdir -p $O; T=$(date +%s)
( echo "== contiguous=1"; SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2b.py
  echo "== contiguous=1, AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3"; AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2b.py
  echo "== contiguous=1, HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2b.py
  echo "== contiguous=1, debug_sync=1 (device sync after every launch)"; SPIRAL_DEBUG_SYNC=1 SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2b.py
) 2>&1 | grep -v amdgpu.ids > $O/r03h_diag_c2b_$T.txt
cat $O/r03h_diag_c2b_$T.txt
