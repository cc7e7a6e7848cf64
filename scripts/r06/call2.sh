#!/bin/bash
# r06 call 2: the whole -m gpu suite after the ADVICE r05 changes (planar lifecycle, OOM ladder, below-Q flag, two-tile PACKED pass
# without its scratch) + the default bench line with the repaired secondary records
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_call2; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/pytest.log 2>&1
tail -14 $O/pytest.log
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call2/bench_c2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('secondary',{}).items():
    print(k, v.get('value'), json.dumps(v.get('batched_pass', v.get('batch8', '')))[:700])
PY
