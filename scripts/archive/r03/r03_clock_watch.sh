#!/bin/bash
# Sample clocks / power / temperature once a second while consecutive C2 queries run (is the decay over a long run throttling?)
out=gpurun_out/r03_clock_watch.txt
mkdir -p gpurun_out
rocm-smi --showmaxpower --showperflevel --showclkfrq 2>&1 | grep -v "^$" | head -60 > $out
( STEPS=${STEPS:-120} ROUNDS=${ROUNDS:-3} python scripts/r03_stream_overlap.py > gpurun_out/r03_clock_watch_run.txt 2>&1 ) &
pid=$!
while kill -0 $pid 2>/dev/null; do
  echo "t=$(date +%s.%N | cut -c1-14)" >> $out
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|Temperature" >> $out
  sleep 1
done
cat gpurun_out/r03_clock_watch_run.txt
