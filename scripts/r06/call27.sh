#!/bin/bash
mkdir -p gpurun_out/r06_call27
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call27/build.log 2>&1
O=gpurun_out/r06_call27
timeout 600 python scripts/r06/repro_fuzz_204.py 150 0 > $O/alone_m0.txt 2>&1; tail -2 $O/alone_m0.txt
timeout 600 python scripts/r06/repro_fuzz_204.py 150 1 > $O/alone_m1.txt 2>&1; tail -2 $O/alone_m1.txt
for k in 1 2 3 4; do
  timeout 900 python scripts/r06/repro_fuzz_204.py 150 0 > $O/par_m0_$k.txt 2>&1 &
  timeout 900 python scripts/r06/repro_fuzz_204.py 150 1 > $O/par_m1_$k.txt 2>&1 &
done
wait
tail -q -n 1 $O/par_*.txt
grep -h "DIFFERS\|same handles" $O/*.txt | head -40
