#!/bin/bash
mkdir -p gpurun_out/r06_call39
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call39/build.log 2>&1 || { tail -3 gpurun_out/r06_call39/build.log; exit 1; }
O=gpurun_out/r06_call39
for k in 1 2 3 4 5 6 7 8; do
  timeout 800 python scripts/r06/soak_updates_multiproc.py 3000 $k > $O/upd_$k.txt 2>&1 &
done
wait
tail -q -n 1 $O/upd_*.txt; grep -h "^rep" $O/upd_*.txt | head
