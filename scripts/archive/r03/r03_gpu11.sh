set -u
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "process_query or sharded or fold or golden or config_sweep" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c2" 2>&1 | tail -2
python scripts/r03_ab_switches.py from_sweep_wave=0 from_sweep_wave=0 2>&1 | grep -v amdgpu
python scripts/r03_batch_ab.py from_sweep_wave=0 from_sweep_wave=0 2>&1 | grep -v amdgpu
CFG=c1 STEPS=200 python scripts/r03_ab_switches.py from_sweep_wave=0 2>&1 | grep -v amdgpu
