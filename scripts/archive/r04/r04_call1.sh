#!/bin/bash
# round 4, first GPU call: the full -m gpu suite (with the new full-size matrix-core batch test and the bench re-exec
# test) and the default bench line with its new `secondary` objects
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r04_call1_pytest.txt 2>&1
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04_call1_bench.json 2> gpurun_out/r04_call1_bench.err
tail -3 gpurun_out/r04_call1_pytest.txt
tail -c 600 gpurun_out/r04_call1_bench.err
