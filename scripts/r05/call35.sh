#!/bin/bash
# Round 5, GPU call 35: k_expand_round (a large expansion round's digit transforms and products in one launch): parity, then
# in-process A/B on one C2 allocation against the three-launch rounds (expand_round_min = 2^30) and for other thresholds.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "expansion_variants or process_query_bytes or query_list or process_query_batch" ) > $O/r05c35_pytest.log 2>&1
tail -1 $O/r05c35_pytest.log
grep -q " passed" $O/r05c35_pytest.log && ! grep -q " failed\| error" $O/r05c35_pytest.log || { echo "parity FAILED"; tail -40 $O/r05c35_pytest.log; exit 1; }
STEPS=16 timeout 500 python scripts/r05/ab.py expand_round_min=1073741824 expand_round_min=512 expand_round_min=8192 expand_round_min=1073741824 2>&1 | grep -v "^$" | tee $O/r05c35_ab_raw.txt
