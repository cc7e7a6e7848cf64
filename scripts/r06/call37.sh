#!/bin/bash
mkdir -p gpurun_out/r06_call37
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call37/build.log 2>&1 || { tail -3 gpurun_out/r06_call37/build.log; exit 1; }
O=gpurun_out/r06_call37
run() {  # tag, mode, reps
  tag=$1; mode=$2; reps=$3
  for k in 1 2 3 4 5 6 7 8; do
    timeout 500 python scripts/r06/repro_fuzz_204.py $reps 0 $mode > $O/${tag}_$k.txt 2>&1 &
  done
  wait
  echo "== $tag"; tail -q -n 1 $O/${tag}_*.txt | sed 's/materialise=0 //' | cut -c1-110
  grep -h -A5 "DIFFERS" $O/${tag}_*.txt | head -24
}
run pponly pponly 1500
run fresh_everything x 400
