cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
( time timeout 400 python scripts/diag_race.py ) > $O/diag_race.log 2>&1
tail -60 $O/diag_race.log
