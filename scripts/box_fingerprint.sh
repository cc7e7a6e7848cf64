# what kind of box is this? (compared between boxes where recycled device memory is read stale and boxes where it is not)
echo "host $(hostname) kernel $(uname -r)"
cat /sys/module/amdgpu/version 2>/dev/null | sed 's/^/amdgpu module version /'
for f in /sys/module/amdgpu/parameters/{mtype_local,noretry,vm_fragment_size,vm_update_mode,sched_policy,mes,cwsr_enable,sdma_phase_quantum,no_system_mem_limit}; do [ -r $f ] && echo "$(basename $f)=$(cat $f)"; done 2>/dev/null | tr '\n' ' '; echo
/opt/rocm/bin/rocminfo 2>/dev/null | grep -iE "Marketing Name|xnack|Compute Unit|Uuid|Node:|Chip ID|ASIC Revision|Cacheline|Max Clock|Internal Node|Coherent Host Access|Memory Properties" | sort | uniq -c | head -30
/opt/rocm/bin/rocm-smi --showuniqueid --showfwinfo --showcomputepartition --showmemorypartition 2>/dev/null | grep -vE "^=|^$" | head -40
env | grep -E "^(HSA|HIP|ROC|GPU|AMD|NCCL|RCCL)_" | sort | tr '\n' ' '; echo
ls /dev/dri 2>/dev/null | tr '\n' ' '; echo
