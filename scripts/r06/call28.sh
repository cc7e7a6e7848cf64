#!/bin/bash
mkdir -p gpurun_out/r06_call28
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call28/build.log 2>&1
O=gpurun_out/r06_call28
for k in 1 2 3 4 5 6 7 8; do
  timeout 1200 python scripts/r06/repro_fuzz_204.py 250 0 > $O/par_$k.txt 2>&1 &
done
wait
tail -q -n 1 $O/par_*.txt
grep -h -A6 "DIFFERS" $O/*.txt | head -80
