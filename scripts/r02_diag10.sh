cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
L=$O/diag_ab_$(date +%s).log
: > $L
for t in "h2d_cache_sync=0,alloc_zero=0,ws_prealloc=0" "h2d_cache_sync=0" "h2d_cache_sync=1" "h2d_cache_sync=0" "h2d_cache_sync=1" "h2d_cache_sync=0,alloc_zero=0,ws_prealloc=0" "h2d_cache_sync=1"; do
  echo "=== $t" >> $L
  ( DIAG_TUNABLES=$t timeout 120 python scripts/diag_free.py ) >> $L 2>&1
done
grep -v amdgpu.ids $L | cut -c1-330 | grep -v "TTT', 'TTT', 'TTT', 'TTT', 'TTT', 'TTT', 'TTT', 'TTT', 'TTT', 'TTT'\] | last handle resident: rc=0 tw:ok neg1:ok gadget:ok lists:ok pp.all:ok pp.pack_cat:ok" | tail -60
