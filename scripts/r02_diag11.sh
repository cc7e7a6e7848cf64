cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
L=$O/diag_c2test.log
: > $L
for e in "X=1" "SPIRAL_WS_PREALLOC=0" "SPIRAL_ALLOC_ZERO=0" "SPIRAL_WS_PREALLOC=0 SPIRAL_ALLOC_ZERO=0 SPIRAL_H2D_CACHE_SYNC=0"; do
  echo "=== $e" >> $L
  ( env $e timeout 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "c2_full_size_response_bytes" 2>&1 | tail -4 ) >> $L 2>&1
done
cat $L
