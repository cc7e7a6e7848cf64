cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
( time timeout 400 python scripts/diag_guard.py ) > $O/diag_guard.log 2>&1
tail -150 $O/diag_guard.log
