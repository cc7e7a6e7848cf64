#!/bin/bash
# r06 call 6: A/B of the batched step's overhead changes in one process; parity of the batch tests; a trace of the 16-query step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call6; mkdir -p $O
ONLY_BATCH=1 BATCH=16,8 timeout 600 python scripts/r06/ab.py expand_group_tail=0 batch_tables_merged=0 expand_group=0 2>&1 | grep -v amdgpu.ids | tee $O/overheads_ab_raw.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_request_layer.py -m gpu -x -q -k "batch or request or concurrent or planar or pp_deserialize or process_query_bytes" 2>&1 | tail -4 | tee $O/pytest_batch.log
cd /tmp; rm -rf /tmp/p7
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o p7 -- python $R/bench.py --headline-only --no-cpu-baseline --batch 16 --steps 4 --warmup 1 > $O/bench_c2_batch16_profiled.json 2> /tmp/p7.err
DB="$(find /tmp/p7 -name '*.db' | head -1)"
python $R/scripts/trace_dump.py "$DB" /tmp/p7.tsv; gzip -c /tmp/p7.tsv > $O/batch16_trace.tsv.gz
python $R/scripts/r06/step_phases.py /tmp/p7.tsv 2 > $O/batch16_step_phases.md 2>&1
cat $O/batch16_step_phases.md
