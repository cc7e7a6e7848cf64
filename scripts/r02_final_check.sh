# HEAD check: default bench line, then the whole gpu suite with durations
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
( time python $R/bench.py ) > $O/r02h_bench_c2.json 2> $O/r02h_bench.err
tail -c 400 $O/r02h_bench_c2.json
cd $R
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/r02h_pytest.log 2>&1
tail -40 $O/r02h_pytest.log
