cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=$(date +%s)
L=$O/r02m_box_$T.log
bash scripts/box_fingerprint.sh > $L 2>&1
for e in "SPIRAL_HIP_LIB=/root/repo/sdk_amd/libspiral_hip_head.so" "X=1" "SPIRAL_WS_PREALLOC=0" "SPIRAL_ALLOC_ZERO=0" "SPIRAL_WS_PREALLOC=0 SPIRAL_ALLOC_ZERO=0 SPIRAL_ALLOC_CACHE_SYNC=0 SPIRAL_H2D_CACHE_SYNC=0" "SPIRAL_HIP_LIB=/root/repo/sdk_amd/libspiral_hip_head.so"; do
  echo "=== $e" >> $L
  ( env $e timeout 200 python -m pytest tests/test_golden_vectors.py tests/test_gpu_fullsize.py -m gpu -q -x -k "golden or c2_full_size_response" 2>&1 | tail -4 ) >> $L 2>&1
done
grep -E "Unique ID|===|passed|failed|FAILED" $L
