#!/bin/bash
mkdir -p gpurun_out/r06_call34
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call34/build.log 2>&1 || { tail -3 gpurun_out/r06_call34/build.log; exit 1; }
O=gpurun_out/r06_call34
run() {  # tag, mode, env...
  tag=$1; mode=$2; shift; shift
  for k in 1 2 3 4 5 6 7 8; do
    env "$@" timeout 400 python scripts/r06/repro_fuzz_204.py 1500 0 $mode > $O/${tag}_$k.txt 2>&1 &
  done
  wait
  echo "== $tag"; tail -q -n 1 $O/${tag}_*.txt | sed 's/materialise=0 //' | cut -c1-110
}
run dbonly_keep dbonly X=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "db_ or planar_copy or update or preprocessing or c1" > $O/pytest_db.log 2>&1; tail -2 $O/pytest_db.log
