#!/bin/bash
mkdir -p gpurun_out/r06_call29
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call29/build.log 2>&1
O=gpurun_out/r06_call29
run() {  # tag, env...
  tag=$1; shift
  for k in 1 2 3 4 5 6 7 8; do
    env "$@" timeout 400 python scripts/r06/repro_fuzz_204.py 1500 0 dbonly > $O/${tag}_$k.txt 2>&1 &
  done
  wait
  echo "== $tag"; tail -q -n 1 $O/${tag}_*.txt | sed 's/materialise=0 same_handles=False //'
}
run base X=1
run h2dsync SPIRAL_H2D_CACHE_SYNC=1
run nozero SPIRAL_ALLOC_ZERO=0
run allsync SPIRAL_H2D_CACHE_SYNC=1 SPIRAL_ALLOC_CACHE_SYNC=1
run base2 X=1
