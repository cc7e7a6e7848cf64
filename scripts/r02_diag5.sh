cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
: > $O/diag_state.log
for e in "X=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_SDMA=0" "SPIRAL_DB_CONTIGUOUS=0" "AMD_SERIALIZE_KERNEL=3" "SPIRAL_NARROW1=1"; do
  echo "=== env $e" >> $O/diag_state.log
  ( env $e timeout 120 python scripts/diag_state.py ) >> $O/diag_state.log 2>&1
done
grep -v amdgpu.ids $O/diag_state.log | tail -80
