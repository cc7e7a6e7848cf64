#!/bin/bash
# Round 5, GPU call 34: the expansion's forward-transform launches with one thing removed at a time (timing-only builds).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for tag in ${TAGS:-base fw_no_load fw_no_store fw_no_transform base}; do
  lib=$R/sdk_amd/variants/libspiral_hip_tv_$tag.so; [ $tag = base ] && lib=$R/sdk_amd/libspiral_hip.so
  rm -rf /tmp/tv_run_$tag
  SPIRAL_HIP_LIB=$lib SPIRAL_PIPELINE=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/tv_run_$tag -o tv -- python $R/bench.py --headline-only --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> /tmp/tv_run_$tag.err
  python $R/scripts/trace_dump.py "$(find /tmp/tv_run_$tag -name '*.db' | head -1)" /tmp/tv_$tag.tsv > /dev/null 2>&1
  echo "$tag: $(awk -F'\t' '$1 ~ /k_ntt_fwd3/ && $6 == 2162688 {n++; s+=$3} END {printf "fwd3 8448x2: %.1f us (%d)", s/n, n}' /tmp/tv_$tag.tsv)  $(awk -F'\t' '$1 ~ /k_ntt_fwd3/ && $6 == 1081344 {n++; s+=$3} END {printf "4224x2: %.1f us", s/n}' /tmp/tv_$tag.tsv)  $(awk -F'\t' '$1 ~ /k_ntt_fwd3/ {s+=$3} END {printf "all fwd3 per query: %.1f us", s/10}' /tmp/tv_$tag.tsv)"
done 2>&1 | tee $O/r05c34_raw.txt
