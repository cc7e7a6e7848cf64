#!/bin/bash
mkdir -p gpurun_out/r06_call32
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call32/build.log 2>&1
O=gpurun_out/r06_call32
run() {  # tag, mode, env...
  tag=$1; mode=$2; shift; shift
  for k in 1 2 3 4 5 6 7 8; do
    env "$@" timeout 400 python scripts/r06/repro_fuzz_204.py 1500 0 $mode > $O/${tag}_$k.txt 2>&1 &
  done
  wait
  echo "== $tag"; tail -q -n 1 $O/${tag}_*.txt | sed 's/materialise=0 //'
}
run reload_keep reload X=1
run reload_churn reload SPIRAL_DB_STAGE_KEEP=0
run dbonly_keep dbonly X=1
run reload_keep2 reload X=1
