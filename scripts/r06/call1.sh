#!/bin/bash
# r06 call 1: the new shipped-config tests + changed tests on gfx950, then this round's baseline bench line (HEAD = r05 kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_call1
bash scripts/box_fingerprint.sh > gpurun_out/r06_call1/box.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_shipped_configs.py "tests/test_gpu_fullsize.py::test_c2_batch16_two_query_tiles_at_full_size" -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r06_call1/pytest_new.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multiply_reg_by_database_shapes" 2>&1 | tail -5 >> gpurun_out/r06_call1/pytest_new.log
timeout 900 python bench.py > gpurun_out/r06_call1/bench_c2.json 2> gpurun_out/r06_call1/bench_c2.err
tail -3 gpurun_out/r06_call1/pytest_new.log
head -c 1500 gpurun_out/r06_call1/bench_c2.json
