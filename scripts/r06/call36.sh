#!/bin/bash
mkdir -p gpurun_out/r06_call36
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_call36/build.log 2>&1 || { tail -3 gpurun_out/r06_call36/build.log; exit 1; }
O=gpurun_out/r06_call36
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k test_database_reload_beside_other_processes ) > $O/pytest_reload.log 2>&1; tail -5 $O/pytest_reload.log
( SPIRAL_DB_STAGE_KEEP=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k test_database_reload_beside_other_processes ) > $O/pytest_reload_old_loader.log 2>&1; tail -3 $O/pytest_reload_old_loader.log; grep -h "reload-worker" $O/pytest_reload_old_loader.log | head
bash scripts/profile_r06.sh r06final3
