cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
: > $O/diag_libs.log
( timeout 100 scripts/ubench/alloc_churn 1500 ) >> $O/diag_libs.log 2>&1
for l in head new head new; do
  echo "=== lib $l" >> $O/diag_libs.log
  if [ $l = head ]; then export SPIRAL_HIP_LIB=/root/repo/sdk_amd/libspiral_hip_head.so; else unset SPIRAL_HIP_LIB; fi
  ( timeout 120 python scripts/diag_free.py ) >> $O/diag_libs.log 2>&1
done
grep -v amdgpu.ids $O/diag_libs.log | tail -60
