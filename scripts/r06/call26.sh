#!/bin/bash
# differential fuzz of the REAL library on gfx950 against the oracle: 8 processes, different seeds, one GPU
mkdir -p gpurun_out/r06_fuzz_gpu2
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_fuzz_gpu2/build.log 2>&1
nproc > gpurun_out/r06_fuzz_gpu2/nproc.txt
for s in 201 202 203 204 205 206 207 208; do
  timeout 1000 python scripts/emu_fuzz.py --device --seed $s --minutes 14 > gpurun_out/r06_fuzz_gpu2/seed_$s.log 2>&1 &
done
wait
grep -h "fuzz:\|FAIL\|MISMATCH\|Traceback" gpurun_out/r06_fuzz_gpu2/seed_*.log
# + the batched path's soak on the final binary (shorter than the first: 60 lists per size)
timeout 1200 python scripts/r06/soak_batch.py 60 > gpurun_out/r06_fuzz_gpu2/soak.txt 2>&1
tail -3 gpurun_out/r06_fuzz_gpu2/soak.txt
