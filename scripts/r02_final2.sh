# expand_order A/B at C2 (and C1/P2 with the split forced), then the whole gpu suite the way the driver runs it, then bench
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
( time timeout 300 python scripts/r02_expand_order.py ) > $O/r02_expand_order.log 2>&1
grep -v amdgpu.ids $O/r02_expand_order.log | tail -12
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02i_pytest.log 2>&1
tail -22 $O/r02i_pytest.log
( timeout 300 python bench.py ) > $O/r02i_bench_c2.json 2> $O/r02i_bench.err
tail -c 600 $O/r02i_bench_c2.json
