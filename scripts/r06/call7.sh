#!/bin/bash
# r06 call 7: the batched step after the prologue merge: A/B against separate expansions, host stamps, a trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call7; mkdir -p $O
ONLY_BATCH=1 BATCH=16,8 timeout 600 python scripts/r06/ab.py expand_group=0 expand_group=0 2>&1 | grep -v amdgpu.ids | tee $O/ab_raw.txt
SPIRAL_BATCH_TRACE=1 timeout 300 python bench.py --headline-only --no-cpu-baseline --batch 16 --steps 3 --warmup 1 2>&1 >/dev/null | grep "spiral\] batch" | tail -8
cd /tmp; rm -rf /tmp/p7
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o p7 -- python $R/bench.py --headline-only --no-cpu-baseline --batch 16 --steps 4 --warmup 1 > $O/bench_c2_batch16_profiled.json 2> /tmp/p7.err
DB="$(find /tmp/p7 -name '*.db' | head -1)"
python $R/scripts/trace_dump.py "$DB" /tmp/p7.tsv; gzip -c /tmp/p7.tsv > $O/batch16_trace.tsv.gz
python $R/scripts/r06/step_phases.py /tmp/p7.tsv 2 > $O/batch16_step_phases.md 2>&1
cat $O/batch16_step_phases.md
