#!/bin/bash
# Round 5, GPU call 6: why does the fold kernel not get faster with fewer instructions?  SQ counters of both wave fold kernels
# (un-pipelined query: the folds have the GPU to themselves), two passes of eight counters each.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
H="--headline-only --no-cpu-baseline --steps 1 --warmup 0 --sweep-iters 1"
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU"
B="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM"
Cc="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT"
for v in 5 6; do
  for pass in A B Cc; do
    rm -rf /tmp/s$v$pass
    SPIRAL_FOLD_VARIANT=$v SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --pmc ${!pass} -d /tmp/s$v$pass -o s -- python $R/bench.py $H > /tmp/s$v$pass.log 2>&1 || { echo "pass $pass variant $v failed"; tail -3 /tmp/s$v$pass.log; }
  done
  python $R/scripts/sq_counters.py $(find /tmp/s${v}A /tmp/s${v}B /tmp/s${v}Cc -name '*.db') -- k_fold_wave k_from_sweep4 > $O/r05c6_variant${v}_sq_counters.md 2>&1
  cat $O/r05c6_variant${v}_sq_counters.md
done
