#!/bin/bash
# the query path without G - C: parity tests touching the fold operands + in-process A/B (single C2, batch 8/16, C1)
mkdir -p gpurun_out/r06_call23
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call23/build.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shipped_configs.py tests/test_request_layer.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06_call23/pytest.log 2>&1
tail -3 gpurun_out/r06_call23/pytest.log
STEPS=20 BATCH=16 timeout 900 python scripts/r06/ab.py fold_neg_materialise=1 fold_neg_materialise=0 fold_neg_materialise=1 > gpurun_out/r06_call23/ab_c2.txt 2>&1
cat gpurun_out/r06_call23/ab_c2.txt
ONLY_BATCH=1 BATCH=8,16 timeout 900 python scripts/r06/ab.py fold_neg_materialise=1 fold_neg_materialise=0 fold_neg_materialise=1 > gpurun_out/r06_call23/ab_batch.txt 2>&1
cat gpurun_out/r06_call23/ab_batch.txt
CFG=c1 STEPS=400 BATCH=8 timeout 600 python scripts/r06/ab.py fold_neg_materialise=1 fold_neg_materialise=0 fold_neg_materialise=1 > gpurun_out/r06_call23/ab_c1.txt 2>&1
cat gpurun_out/r06_call23/ab_c1.txt
