cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
: > $O/diag_resident.log
for i in 1 2; do ( timeout 120 python scripts/diag_free.py ) >> $O/diag_resident.log 2>&1; done
grep -v amdgpu.ids $O/diag_resident.log | tail -60
