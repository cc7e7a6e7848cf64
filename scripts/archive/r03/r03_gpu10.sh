set -u
O=$PWD/gpurun_out; T=$(date +%s)
( cd scripts/ubench && MFMA_UBENCH_SHORT=1 MFMA_UBENCH_NORATE=1 ./mfma_sweep 2048 3 2>&1 | grep "mfma a\]\|verify a\]\|mfma b\]\|verify b" )
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" 2>&1 | tail -2
timeout 900 python bench.py > $O/r03j_bench_c2_$T.json 2> $O/r03j_bench_c2_$T.err
python - <<PY
import json
d=json.load(open('$O/r03j_bench_c2_$T.json')); c=d['cpu_baseline']
print('c2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['standalone']['ms_per_launch'])
print('cpu', c['value'], c['cores'], c['thread_scan_seconds'], c['modes'], c['seconds_per_query'])
print([(u['config'], round(u['all_core_qps'],2), round(u['faithful_qps'],2)) for u in c['unsampled']])
PY
timeout 600 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch8', d['value'], d['ms_per_step'], d['batch_selfcheck'], d['roofline']['batched_pass']['ms_per_pass'])"
