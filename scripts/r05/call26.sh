#!/bin/bash
# Round 5, GPU call 26: k_ntt_fwd3_wave (forward transforms of a large launch on the wave-per-transform NTT): parity, then
# in-process A/B on one C2 allocation against the cooperative kernels (fwd_wave_min = 2^30) and for other thresholds.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "expansion_variants or process_query_bytes or test_ntt or regev or pack" ) > $O/r05c26_pytest.log 2>&1
tail -2 $O/r05c26_pytest.log
grep -q " passed" $O/r05c26_pytest.log && ! grep -q " failed\| error" $O/r05c26_pytest.log || { echo "parity FAILED"; tail -40 $O/r05c26_pytest.log; exit 1; }
STEPS=16 timeout 500 python scripts/r05/ab.py fwd_wave_min=1073741824 fwd_wave_min=512 fwd_wave_min=8192 fwd_wave_min=1073741824 2>&1 | grep -v "^$" | tee $O/r05c26_ab_raw.txt
