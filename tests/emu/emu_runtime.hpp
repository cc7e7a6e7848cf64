// TEST INFRASTRUCTURE: see hip/hip_runtime.h in this directory.
#pragma once
#include <hip/hip_runtime.h>

void emu_launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()>& work_item);  // emu_runtime.cpp: runs the grid, returns

namespace emu {
// body() runs once per work-item of ONE workgroup of `nthreads`
inline void run_block(unsigned nthreads, const std::function<void()>& body) { emu_launch(dim3(1), dim3(nthreads), 0, body); }
}  // namespace emu
