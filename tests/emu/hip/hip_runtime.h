// TEST INFRASTRUCTURE, not a runtime: the few names of <hip/hip_runtime.h> that sdk_amd/csrc/device_common.hpp needs, so that
// g++ can compile the DEVICE helper functions of the product headers unchanged and tests/emu/device_bodies_emu.cpp can run
// them as one 256-thread workgroup on host threads (a thread per work-item, a pthread barrier for __syncthreads).  It lets the
// CPU suite check the kernels' arithmetic and LDS index patterns against the oracle; nothing in sdk_amd/ includes it and
// nothing here is a fallback for the GPU path (the product library is built by hipcc for gfx950 only and fails without a GPU).
#pragma once
#include <cstddef>
#include <cstdint>

#define __device__
#define __host__
#define __global__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))

struct dim3 {
  unsigned x = 1, y = 1, z = 1;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 {
  unsigned x = 0, y = 0, z = 0;
};
extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
typedef int hipError_t;

void emu_syncthreads();
int emu_syncthreads_or(int v);
inline void __syncthreads() { emu_syncthreads(); }
inline int __syncthreads_or(int v) { return emu_syncthreads_or(v); }

inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

struct uint2 {
  uint32_t x, y;
};
struct uint4 {
  uint32_t x, y, z, w;
};
struct ulonglong2 {
  unsigned long long x, y;
};
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
template <typename T>
inline T min(T a, T b) { return a < b ? a : b; }
template <typename T>
inline T max(T a, T b) { return a > b ? a : b; }
// atomics on global / LDS words: the emulated work-items are real host threads
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// ---- wave-level operations (a wave = 64 consecutive work-items of the emulated workgroup) ----
typedef uint32_t emu_u32x2 __attribute__((ext_vector_type(2)));
void emu_wave_barrier();
emu_u32x2 emu_permlane32_swap(uint32_t vdst, uint32_t vsrc);
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
// v_permlane32_swap_b32 vdst, vsrc: the upper 32 lanes of vdst trade places with the lower 32 lanes of vsrc
#define __builtin_amdgcn_permlane32_swap(vdst, vsrc, fi, bc) emu_permlane32_swap((vdst), (vsrc))

// gfx950 inline assembly cannot run here: the statement is replaced by a call that throws, so a test that reaches one fails
// loudly instead of computing something else (wave_ntt.hpp's forward butterfly batch is the one user; its inverse has none).
[[noreturn]] void emu_unsupported_asm();
#define asm(...) emu_unsupported_asm()
