cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
L=$O/r02_confirm_$(date +%s).log
bash scripts/box_fingerprint.sh 2>&1 | grep "Unique ID" > $L
for e in "X=1" "SPIRAL_DB_CONTIGUOUS=1" "X=1"; do
  echo "=== $e" >> $L
  ( env $e timeout 100 python scripts/diag_free.py 2>&1 | grep -v amdgpu.ids | cut -c1-120 ) >> $L 2>&1
done
cat $L | grep -E "Unique|===|F'|failing" | cut -c1-200
