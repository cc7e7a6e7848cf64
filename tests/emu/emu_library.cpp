// TEST INFRASTRUCTURE: what the emulated build of the library (tests/emu/build_emulated_library.py) needs beside the product
// sources -- the storage behind the kernels' `extern __shared__` arrays and a marker symbol by which bench.py and
// __graft_entry__.smoke() refuse to run on it.
#include <hip/hip_runtime.h>

namespace spiral {
// dynamic LDS: one array per name and host thread (a workgroup in flight): 160 KiB + room for the canary the runtime puts right
// behind the size a launch asked for (a workgroup that writes past that size fails the launch)
constexpr size_t LDS_ARRAY = 160 * 1024 + 256;
alignas(16) thread_local uint32_t smem_fw[LDS_ARRAY / 4];
alignas(16) thread_local unsigned char smem_q[LDS_ARRAY];
alignas(16) thread_local unsigned char smem[LDS_ARRAY];
alignas(16) thread_local unsigned char smem_rq[LDS_ARRAY];
alignas(16) thread_local unsigned char smem_pl[LDS_ARRAY];
}  // namespace spiral

typedef unsigned char* (*emu_lds_getter)();
void emu_register_dynamic_lds(emu_lds_getter f);
namespace {
struct RegisterLds {
  RegisterLds() {
    emu_register_dynamic_lds([]() -> unsigned char* { return reinterpret_cast<unsigned char*>(spiral::smem_fw); });
    emu_register_dynamic_lds([]() -> unsigned char* { return spiral::smem_q; });
    emu_register_dynamic_lds([]() -> unsigned char* { return spiral::smem; });
    emu_register_dynamic_lds([]() -> unsigned char* { return spiral::smem_rq; });
    emu_register_dynamic_lds([]() -> unsigned char* { return spiral::smem_pl; });
  }
} g_register_lds;
}  // namespace

extern "C" int sp_emulated_device_marker() { return 1; }
