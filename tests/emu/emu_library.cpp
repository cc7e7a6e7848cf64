// TEST INFRASTRUCTURE: what the emulated build of the library (tests/emu/build_emulated_library.py) needs beside the product
// sources -- the storage behind the kernels' `extern __shared__` arrays and a marker symbol by which bench.py and
// __graft_entry__.smoke() refuse to run on it.
#include <hip/hip_runtime.h>

namespace spiral {
// dynamic LDS: one 160 KiB array per name and host thread (a workgroup in flight); emu_launch checks the requested size
alignas(16) thread_local uint32_t smem_fw[160 * 1024 / 4];
alignas(16) thread_local unsigned char smem_q[160 * 1024];
alignas(16) thread_local unsigned char smem[160 * 1024];
alignas(16) thread_local unsigned char smem_rq[160 * 1024];
}  // namespace spiral

extern "C" int sp_emulated_device_marker() { return 1; }
