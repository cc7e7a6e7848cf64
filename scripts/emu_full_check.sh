#!/bin/bash
# The long form of the emulated-device checks (the CPU suite runs a subset): every parity test the emulation can run, under
# AddressSanitizer with shuffled work-item schedules, then the multi-stream paths with every stream starved in turn.
# CPU only, ~1.5-2 h on 8 cores.  Output: /tmp/emu_full_check_*.log; exit status 0 when everything passed.
#   scripts/emu_full_check.sh [quick]      quick: skips the large batched shapes and starves 6 streams instead of 18
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
quick=${1:-}
python tests/emu/build_emulated_library.py > /dev/null || exit 1
python tests/emu/build_emulated_library.py --asan > /dev/null || exit 1
L=$R/tests/emu/_build/libspiral_emu.so
LA=$R/tests/emu/_build/libspiral_emu_asan.so
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
# needs a GPU whatever the library: torch views of device pointers, bench.py, a real RCCL
SKIP="not c2_full_size and not bench_modes and not distributed_fold_single_gpu and not column_sharded_single and not rccl_world1 and not row_sharded_partials and not sharded_c_abi_loopback and not sharded_pipelined_list and not process_query_c1"
[ -n "$quick" ] && SKIP="$SKIP and not 256x256 and not 512x128 and not 128x128 and not 32x256"
rc=0
rm -f /tmp/emu_full_check_asan.*
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:log_path=/tmp/emu_full_check_asan \
  SPIRAL_EMU_SCHEDULE=random:1 SPIRAL_HIP_LIB=$LA python -m pytest tests/test_gpu_parity.py tests/test_sparse_bucket.py tests/test_request_layer.py \
  -m gpu -q --timeout=1800 -p no:cacheprovider -k "$SKIP" > /tmp/emu_full_check_parity.log 2>&1 || rc=1
tail -2 /tmp/emu_full_check_parity.log
ls /tmp/emu_full_check_asan.* 2>/dev/null && rc=1
STREAMS="(test_process_query_bytes_and_decode and (fast-0 or fast56 or nu2_0)) or (test_ring_sweep_and_batched_tails_parity and (5-10-4-8-1-256 or 5-10-4-8-1-0)) or (test_expansion_variants_response_parity and (0-split or 1-launches)) or test_overlapped_fold or (test_process_query_batch and (narrow-3 or narrow-4 or packed)) or test_query_list_in_flight or (test_process_query_batch_matrix_core_sweep and 64x128)"
n=18; [ -n "$quick" ] && n=6
for k in $(seq 1 $n) r1 r2 r3; do
  pol=starve:$k; case $k in r*) pol=random:${k#r};; esac
  SPIRAL_EMU_STREAMS=$pol SPIRAL_HIP_LIB=$L python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=1800 -p no:cacheprovider -k "$STREAMS" > /tmp/emu_full_check_streams_$k.log 2>&1 || rc=1
  echo "$pol: $(tail -1 /tmp/emu_full_check_streams_$k.log)"
done
exit $rc
