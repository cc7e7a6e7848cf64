// TEST INFRASTRUCTURE: runs a callable as one workgroup of host threads (see hip/hip_runtime.h in this directory).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>

namespace emu {
// body() runs once per work-item with threadIdx.x = 0 .. nthreads-1 and blockIdx = (bx, by); one workgroup at a time
void run_block(unsigned nthreads, unsigned bx, unsigned by, const std::function<void()>& body);
}  // namespace emu
