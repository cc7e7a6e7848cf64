cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out
( timeout 300 python scripts/diag_nu2_1.py ) > $O/diag_nu2_1.log 2>&1
tail -50 $O/diag_nu2_1.log
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_request_layer.py tests/test_sparse_bucket.py -m gpu -q ) > $O/r02h_pytest_rest.log 2>&1
tail -30 $O/r02h_pytest_rest.log
