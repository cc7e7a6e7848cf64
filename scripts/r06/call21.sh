#!/bin/bash
# r06 call 21: the whole -m gpu suite + smoke + default bench on the final tree (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call21; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
( time timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call21/bench_c2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(round(v.get('value'),1) if isinstance(v,dict) and v.get('value') else v) for k,v in d['secondary'].items()}, d['cpu_baseline']['value'])
PY
