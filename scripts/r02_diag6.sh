cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
: > $O/diag_free.log
for t in "none=0" "ws_pinned=0" "debug_sync=2" "db_contiguous=0" "none=0"; do
  echo "=== tunables $t" >> $O/diag_free.log
  ( DIAG_TUNABLES=$t timeout 120 python scripts/diag_free.py ) >> $O/diag_free.log 2>&1
done
grep -v amdgpu.ids $O/diag_free.log | tail -80
