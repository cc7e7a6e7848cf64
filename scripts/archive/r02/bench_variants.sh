#!/bin/bash
# usage: scripts/bench_variants.sh "ENV1=a ENV2=b" "ENV1=c" ...  -- C2 bench under env-selected kernel variants
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "== $v"
  env $v timeout 200 python bench.py --config ${CONFIG:-c2} --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(round(d['value'],2), 'q/s', {k: round(v,3) for k,v in d['config']['stage_ms'].items()}, 'sweep GB/s', round(d['roofline']['achieved'],1))
"
done
