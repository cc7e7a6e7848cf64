#!/bin/bash
# differential fuzz of the REAL library on gfx950 against the oracle: 8 processes, different seeds, one GPU
mkdir -p gpurun_out/r06_fuzz_gpu4
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_fuzz_gpu4/build.log 2>&1
nproc > gpurun_out/r06_fuzz_gpu4/nproc.txt
for s in 401 402 403 404 405 406 407 408; do
  timeout 1000 python scripts/emu_fuzz.py --device --seed $s --minutes 14 > gpurun_out/r06_fuzz_gpu4/seed_$s.log 2>&1 &
done
wait
grep -h "fuzz:\|FAIL\|MISMATCH\|Traceback" gpurun_out/r06_fuzz_gpu4/seed_*.log
