#!/bin/bash
# Round 3, GPU call 7: capture the library's allocation / free trace on the failing flow, replay it stand-alone.
set -u
O=$PWD/gpurun_out
mkdir -p $O
T=$(date +%s)
bash scripts/box_fingerprint.sh 2>/dev/null | grep -i "unique" > $O/r03g_box_$T.txt
SPIRAL_ALLOC_DEBUG=1 SPIRAL_DB_CONTIGUOUS=1 timeout 300 python scripts/diag_c2.py > $O/r03g_trace_$T.txt 2>&1
grep -v "^\[spiral\]" $O/r03g_trace_$T.txt | grep -v amdgpu.ids
grep "^\[spiral\]" $O/r03g_trace_$T.txt > $O/r03g_alloc_trace_$T.txt
wc -l $O/r03g_alloc_trace_$T.txt
cd scripts/ubench
( echo "== replay WITH contiguous allocations"; timeout 300 ./alloc_replay $O/r03g_alloc_trace_$T.txt 1 2; echo "exit $?"
  echo "== replay, same trace, plain hipMalloc only"; timeout 300 ./alloc_replay $O/r03g_alloc_trace_$T.txt 0 2; echo "exit $?"
  echo "== replay WITH contiguous allocations, full-size database"; timeout 300 ./alloc_replay $O/r03g_alloc_trace_$T.txt 1 64; echo "exit $?" ) > $O/r03g_replay_$T.txt 2>&1
cat $O/r03g_replay_$T.txt
cat $O/r03g_box_$T.txt
