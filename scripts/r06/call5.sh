#!/bin/bash
# r06 call 5: host time stamps of a 16-query call
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call5; mkdir -p $O
SPIRAL_BATCH_TRACE=1 timeout 300 python bench.py --headline-only --no-cpu-baseline --batch 16 --steps 3 --warmup 1 > $O/bench.json 2> $O/trace.txt
grep "spiral\] batch" $O/trace.txt | tail -24
