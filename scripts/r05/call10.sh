#!/bin/bash
# Round 5, call 10: the default bench line (headline + secondary + cpu_baseline) with the quota-aware CPU baseline.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/r05c10_bench_c2.json 2> $O/r05c10_bench_c2.err
tail -3 $O/r05c10_bench_c2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05c10_bench_c2.json").read().strip().splitlines()[-1])
print("value %.2f q/s  ms/step %.3f  roofline frac %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic")))
print(d["roofline"].get("traffic_source"))
c = d["cpu_baseline"]
print(json.dumps({k: c[k] for k in c if k not in ("sample", "unsampled")}, indent=1))
print(c["sample"][:900])
for u in c.get("unsampled", []):
    print(u["config"], u["all_core_qps"], u.get("all_core_avx2_ntt_qps"), u["faithful_qps"], u["team_per_stage"])
s = d.get("secondary", {})
for k in s:
    v = s[k]
    print(k, v.get("value") if isinstance(v, dict) else v)
PY
