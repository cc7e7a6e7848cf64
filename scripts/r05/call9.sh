#!/bin/bash
# Round 5, call 9 (host CPUs of the GPU box, no GPU work): why is the CPU baseline fastest on 32 of 256 logical CPUs?
# Topology, then one stage at a time under thread-binding and allocator variants (each its own process: libgomp reads its
# environment when it is loaded).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
echo "== lscpu"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|^CPU\(s\)|MHz" 
echo "== nodes"; for n in /sys/devices/system/node/node*; do echo "$n: cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4, $5}')"; done
echo "== affinity"; python -c "import os; print(len(os.sched_getaffinity(0)), 'cpus allowed')"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
echo "== THP"; cat /sys/kernel/mm/transparent_hugepage/enabled; cat /proc/sys/kernel/numa_balancing 2>/dev/null
echo "== sweep (1024 z-rows of a C2 plane = 8.6 GB)"
for env in "OMP_PROC_BIND=false OMP_PLACES=" "OMP_PROC_BIND=spread OMP_PLACES=cores" "OMP_PROC_BIND=close OMP_PLACES=threads"; do
  for t in 32 64 128 256; do
    env $env timeout 120 python scripts/r05/cpu_scan.py sweep $t 1024 2>&1 | tail -1
  done
done
echo "== expand"
for alloc in 0 1; do for env in "OMP_PROC_BIND=false OMP_PLACES=" "OMP_PROC_BIND=spread OMP_PLACES=cores"; do for t in 32 64 128; do
  env $env ORACLE_TUNE_ALLOCATOR=$alloc timeout 120 python scripts/r05/cpu_scan.py expand $t 2>&1 | tail -1
done; done; done
echo "== fold"
for alloc in 0 1; do for env in "OMP_PROC_BIND=false OMP_PLACES=" "OMP_PROC_BIND=spread OMP_PLACES=cores"; do for t in 32 64 128; do
  env $env ORACLE_TUNE_ALLOCATOR=$alloc timeout 200 python scripts/r05/cpu_scan.py fold $t 2>&1 | tail -1
done; done; done
} 2>&1 | tee $O/r05c9_cpu_scan.txt
