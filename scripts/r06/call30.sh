#!/bin/bash
mkdir -p gpurun_out/r06_call30
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call30/build.log 2>&1
O=gpurun_out/r06_call30
run() {  # tag, env...
  tag=$1; shift
  for k in 1 2 3 4 5 6 7 8; do
    env "$@" timeout 400 python scripts/r06/repro_fuzz_204.py 1500 0 dbonly > $O/${tag}_$k.txt 2>&1 &
  done
  wait
  echo "== $tag"; tail -q -n 1 $O/${tag}_*.txt | sed 's/materialise=0 same_handles=False //'
}
run uncached X=1
run cached SPIRAL_STAGE_UNCACHED=0
run uncached2 X=1
# the query's own H2D path: same handles, a different query every repetition
for k in 1 2 3 4 5 6 7 8; do
  timeout 400 python scripts/r06/repro_fuzz_204.py 6000 0 queries > $O/queries_$k.txt 2>&1 &
done
wait
echo "== queries"; tail -q -n 1 $O/queries_*.txt; grep -h "^rep" $O/queries_*.txt | head
