#!/bin/bash
# r06 call 13: wall clock of the default bench command (the driver's run) and of smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call13; mkdir -p $O
( time python -c 'import __graft_entry__ as g; g.smoke()' ) 2>&1 | tail -5
( time python bench.py > $O/bench_c2.json 2> $O/bench_c2.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call13/bench_c2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['secondary'].items()})
PY
