#!/bin/bash
set -u
O=$PWD/gpurun_out; mkdir -p $O; T=$(date +%s)
bash scripts/box_fingerprint.sh 2>/dev/null | grep -i "unique" > $O/r03i_box_$T.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or loopback" 2>&1 | tail -4
( timeout 600 python scripts/r03_rank_critical_path.py c2 8 4 2; timeout 600 python scripts/r03_rank_critical_path.py c3 8 ) 2>&1 | grep -v amdgpu.ids > $O/r03i_rank_critical_path_$T.jsonl
cat $O/r03i_rank_critical_path_$T.jsonl
cd scripts/ubench
( ./contig_repro 0 0 4 4; echo "exit $?"; ./contig_repro 0 0 4 5; echo "exit $?"; ./contig_repro 0 0 4 6; echo "exit $?" ) > $O/r03i_contig_reuse_$T.txt 2>&1
cat $O/r03i_contig_reuse_$T.txt
cat $O/r03i_box_$T.txt
