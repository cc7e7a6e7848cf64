#!/bin/bash
# Round 3, GPU call 6: contiguous-allocation corruption -- stand-alone "history" mode, library-level variants with
# allocation addresses; re-run of the test that failed in call 5.
set -u
O=$PWD/gpurun_out
mkdir -p $O
T=$(date +%s)
bash scripts/box_fingerprint.sh 2>/dev/null | grep -i "unique" > $O/r03f_box_$T.txt
cd scripts/ubench
( timeout 250 ./contig_repro 56 96 2 3; echo "exit $?" ) > $O/r03f_contig_history_$T.txt 2>&1
grep -E "RESULT|exit|five|allocation|changed" $O/r03f_contig_history_$T.txt
cd ../..
( echo "== contiguous=1, small configurations first"; SPIRAL_ALLOC_DEBUG=1 SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2.py
  echo "== contiguous=1, C2 alone"; SMALL_FIRST=0 SPIRAL_ALLOC_DEBUG=1 SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2.py
  echo "== contiguous=1, C2 alone, fresh allocations not zero-filled"; SMALL_FIRST=0 SPIRAL_ALLOC_ZERO=0 SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2.py
) > $O/r03f_diag_c2_$T.txt 2>&1
grep -v "hipMalloc [0-9]* bytes" $O/r03f_diag_c2_$T.txt | grep -v amdgpu.ids
echo "--- allocations around the database (first run)"
grep -n "contiguous allocation" -B6 -A3 $O/r03f_diag_c2_$T.txt | head -60
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_modes or spill or add_and or reorient or boundary" 2>&1 | tail -4
cat $O/r03f_box_$T.txt
