cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
: > $O/diag_cachesync.log
for t in "h2d_cache_sync=0" "h2d_cache_sync=1" "h2d_cache_sync=0" "h2d_cache_sync=1"; do
  echo "=== $t" >> $O/diag_cachesync.log
  ( DIAG_TUNABLES=$t timeout 120 python scripts/diag_free.py ) >> $O/diag_cachesync.log 2>&1
done
grep -v amdgpu.ids $O/diag_cachesync.log | cut -c1-400 | tail -60
