cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=$(date +%s)
L=$O/r02n_box_$T.log
bash scripts/box_fingerprint.sh > $L 2>&1
for e in "X=1" "SPIRAL_DB_CONTIGUOUS=0" "SPIRAL_ALLOC_ZERO=0" "SMALL_FIRST=0"; do
  echo "=== $e" >> $L
  ( env $e timeout 200 python scripts/diag_c2.py 2>&1 | grep -v amdgpu.ids ) >> $L 2>&1
done
grep -E "Unique ID|===|ok|rror" $L
