#!/bin/bash
# r06 call 18: k_expand_unit (one wave per ciphertext and modulus) for the left-hand side (1) / both sides (2), alternating with off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call18; mkdir -p $O
ONLY_BATCH=1 BATCH=16,8 timeout 900 python scripts/r06/ab.py expand_unit=1 expand_unit=0 expand_unit=2 expand_unit=0 expand_unit=1 expand_unit=0 expand_unit=2 expand_unit=0 expand_unit=1 2>&1 | grep -v amdgpu.ids | tee $O/expand_unit_ab_raw.txt
cd /tmp
for v in 1 2; do
rm -rf /tmp/p7
SPIRAL_EXPAND_UNIT=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o p7 -- python $R/bench.py --headline-only --no-cpu-baseline --batch 16 --steps 4 --warmup 1 > /dev/null 2> /tmp/p7.err
DB="$(find /tmp/p7 -name '*.db' | head -1)"
python $R/scripts/trace_dump.py "$DB" /tmp/p7.tsv > /dev/null
echo "expand_unit=$v"
python - <<'PY'
rows=[l.rstrip('\n').split('\t') for l in open('/tmp/p7.tsv') if not l.startswith('#')][1:]
R=[(r[0],float(r[1]),float(r[2])) for r in rows]
ps=[r for r in R if r[0].startswith('k_sweep_planar')]
t0,t1=ps[2][1],ps[3][1]
tot=0
for n,s,d in R:
    if t0<=s<t1 and n.startswith('k_expand'):
        print("%-28s %9.1f %8.1f"%(n,s-t0,d)); tot+=d
print("sum of k_expand* %.2f ms, step %.2f ms"%(tot/1e3,(t1-t0)/1e3))
PY
done
