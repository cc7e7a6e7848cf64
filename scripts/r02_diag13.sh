cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=$(date +%s)
L=$O/r02l_box_$T.log
bash scripts/box_fingerprint.sh > $L 2>&1
for e in "SPIRAL_ALLOC_CACHE_SYNC=0" "X=1" "SPIRAL_ALLOC_CACHE_SYNC=0 SPIRAL_QUERY_CACHE_SYNC=1" "SPIRAL_ALLOC_CACHE_SYNC=0" "X=1"; do
  echo "=== $e" >> $L
  ( env $e timeout 200 python -m pytest tests/test_golden_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x -k "golden or c2_full_size_response or process_query_bytes_and_decode" 2>&1 | tail -4 ) >> $L 2>&1
done
grep -E "Unique ID|===|passed|failed|FAILED" $L
