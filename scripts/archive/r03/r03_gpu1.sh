#!/bin/bash
# Round 3, GPU call 1: MFMA batched-sweep microbenchmark, contiguous-allocation reproducer, VMM placement probe.
set -u
O=gpurun_out
mkdir -p $O
T=$(date +%s)
bash scripts/box_fingerprint.sh > $O/r03a_box_$T.txt 2>&1
cd scripts/ubench
( timeout 300 ./mfma_sweep 2048 3 ) > ../../$O/r03a_mfma_ubench_$T.txt 2>&1
( timeout 200 ./contig_repro 56 96 3 1; echo "exit $?"; timeout 200 ./contig_repro 56 96 2 0; echo "exit $?" ) > ../../$O/r03a_contig_repro_$T.txt 2>&1
(
for i in 1 2 3; do timeout 60 ./hbm_map b 0; done
for i in 1 2 3; do timeout 60 ./hbm_map v 1; done
for i in 1 2; do timeout 60 ./hbm_map v 2; done
for i in 1 2; do timeout 60 ./hbm_map v 0.25; done
timeout 60 ./hbm_map v 1 8
timeout 60 ./hbm_map b 8
timeout 60 ./hbm_map c 0
) > ../../$O/r03a_vmm_probe_$T.txt 2>&1
cd ../..
tail -n 40 $O/r03a_mfma_ubench_$T.txt
tail -n 12 $O/r03a_contig_repro_$T.txt
cat $O/r03a_vmm_probe_$T.txt
