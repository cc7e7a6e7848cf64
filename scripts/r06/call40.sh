#!/bin/bash
mkdir -p gpurun_out/r06_call40
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call40/build.log 2>&1 || { tail -3 gpurun_out/r06_call40/build.log; exit 1; }
O=gpurun_out/r06_call40
for k in 1 2 3 4 5 6 7 8; do
  timeout 800 python scripts/r06/soak_batch_multiproc.py 1500 $k > $O/batch_$k.txt 2>&1 &
done
wait
tail -q -n 1 $O/batch_*.txt | cut -c1-200; grep -h "^rep" $O/batch_*.txt | head
