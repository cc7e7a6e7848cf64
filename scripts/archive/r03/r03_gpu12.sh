# stand-alone duration of the from_ntt-of-the-sweep-output kernels: un-pipelined C2 query (one sweep launch, then from_ntt of all 4 planes, then the folds)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
for v in "SPIRAL_FROM_SWEEP_WAVE=1" "SPIRAL_FROM_SWEEP_WAVE=0" "SPIRAL_FROM_SWEEP_WAVE=1 SPIRAL_FROM_SWEEP_FAKE_TILED=1" "SPIRAL_FROM_SWEEP_WAVE=1 SPIRAL_FROM_SWEEP_NT=1"; do
  rm -rf /tmp/fs
  env SPIRAL_PIPELINE=0 $v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fs -o fs -- python $R/bench.py --steps 4 --warmup 1 --sweep-iters 1 --no-cpu-baseline > /tmp/fs.log 2>&1
  echo "== $v"
  python $R/scripts/rocprof_summary.py "$(find /tmp/fs -name '*.db' | head -1)" /tmp/fs.md 2>/dev/null | grep -E "from_sweep|fold_wave|k_sweep_packed" | cut -c1-160
done
