#!/bin/bash
# r06 call 16: k_expand_wave against the cooperative expansion kernel (in-process A/B), parity on gfx950, trace of a 16-query step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or request" 2>&1 | tail -3 | tee $O/pytest.log
ONLY_BATCH=1 BATCH=16,8 timeout 600 python scripts/r06/ab.py expand_wave_min_digits=0 expand_wave_min_digits=1 expand_wave_min_digits=0 2>&1 | grep -v amdgpu.ids | tee $O/ab_raw.txt
cd /tmp; rm -rf /tmp/p7
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o p7 -- python $R/bench.py --headline-only --no-cpu-baseline --batch 16 --steps 4 --warmup 1 > $O/bench_c2_batch16_profiled.json 2> /tmp/p7.err
DB="$(find /tmp/p7 -name '*.db' | head -1)"
python $R/scripts/trace_dump.py "$DB" /tmp/p7.tsv; gzip -c /tmp/p7.tsv > $O/batch16_trace.tsv.gz
python - <<'PY'
rows=[l.rstrip('\n').split('\t') for l in open('/tmp/p7.tsv') if not l.startswith('#')][1:]
R=[(r[0],float(r[1]),float(r[2])) for r in rows]
ps=[r for r in R if r[0].startswith('k_sweep_planar')]
t0,t1=ps[2][1],ps[3][1]
print("step %.2f ms"%((t1-t0)/1e3))
tot=0
for n,s,d in R:
    if t0<=s<t1 and (n.endswith('_group') or n.startswith(('k_expand','k_query_'))):
        print("%-28s %9.1f %8.1f"%(n,s-t0,d)); tot+=d
print("sum of the group's expansion launches %.2f ms"%(tot/1e3))
PY
