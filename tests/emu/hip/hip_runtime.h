// TEST INFRASTRUCTURE, not a runtime and not a fallback.
//
// A stand-in for <hip/hip_runtime.h> with which the HOST compiler can build the sources of sdk_amd/csrc unchanged, so that the
// CPU test suite can run the kernels' code against the oracle without a GPU (tests/test_device_bodies_emulated.py,
// tests/test_emulated_library.py):
//   * a workgroup is a set of fibers (one per work-item) on one host thread, scheduled round-robin; __syncthreads, the
//     lockstep points of a wave (LDS hand-over inside a wave, v_permlane32_swap, __shfl_xor, readfirstlane) are barriers among
//     them; the workgroups of a launch are dealt to a few host threads (emu_runtime.cpp);
//   * "device memory" is host memory, every launch and copy completes before the call returns, streams and events are names;
//   * gfx950 builtins are restated in C (the matrix-core instruction included: see emu_mfma_i32_16x16x64_i8).
// Nothing in sdk_amd/ includes this file; the product library is built by hipcc for gfx950 only and fails without a GPU.  The
// emulation says nothing about speed, occupancy, memory ordering between workgroups, or anything else the hardware decides.
#pragma once
// every standard header the product sources use, BEFORE the keyword macros at the end of this file
#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <random>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <sys/random.h>
#include <thread>
#include <unordered_map>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __constant__
// LDS: one copy per host thread = per workgroup in flight (all fibers of a workgroup run on one host thread).  At block scope
// thread_local implies static; `extern __shared__ T name[];` becomes a declaration of spiral::name, defined in emu_library.cpp.
#define __shared__ thread_local
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
  unsigned x = 1, y = 1, z = 1;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 {
  unsigned x = 0, y = 0, z = 0;
};
struct emu_workitem {  // what a fiber knows about itself
  emu_uint3 tid, bid;
  dim3 bdim, gdim;
};
emu_workitem* emu_self();
#define threadIdx (emu_self()->tid)
#define blockIdx (emu_self()->bid)
#define blockDim (emu_self()->bdim)
#define gridDim (emu_self()->gdim)

// ---- vector types ----------------------------------------------------------------------------------------------------
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

// ---- work-item functions ---------------------------------------------------------------------------------------------
void emu_syncthreads();
int emu_syncthreads_or(int v);
inline void __syncthreads() { emu_syncthreads(); }
inline int __syncthreads_or(int v) { return emu_syncthreads_or(v); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffs(int v) { return __builtin_ffs(v); }

// atomics on global / LDS words: workgroups of one launch run on several host threads
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return o;
}
inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
// scoped atomic loads (k_fold_cluster reads the other workgroups' sums with device scope): host memory is coherent
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(PTR, ORDER, SCOPE) __atomic_load_n((PTR), (ORDER))
#define __hip_atomic_store(PTR, VAL, ORDER, SCOPE) __atomic_store_n((PTR), (VAL), (ORDER))

// ---- wave-level operations (a wave = 64 consecutive work-items of the workgroup) -----------------------------------------
typedef uint32_t emu_u32x2 __attribute__((ext_vector_type(2)));
typedef int emu_i32x4 __attribute__((ext_vector_type(4)));
void emu_wave_barrier();
emu_u32x2 emu_permlane32_swap(uint32_t vdst, uint32_t vsrc);
uint32_t emu_wave_exchange(uint32_t mine, unsigned from_lane);  // every lane contributes `mine`, gets lane `from_lane`'s
template <typename T>
inline T emu_shfl(T v, unsigned from_lane) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte shuffles");
  if (sizeof(T) == 4) {
    uint32_t b;
    std::memcpy(&b, &v, 4);
    b = emu_wave_exchange(b, from_lane);
    std::memcpy(&v, &b, 4);
    return v;
  }
  uint32_t b[2];
  std::memcpy(b, &v, 8);
  b[0] = emu_wave_exchange(b[0], from_lane);
  b[1] = emu_wave_exchange(b[1], from_lane);
  std::memcpy(&v, b, 8);
  return v;
}
unsigned emu_lane();
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return emu_shfl(v, emu_lane() ^ (unsigned)mask); }
template <typename T>
inline T __shfl(T v, int lane, int width = 64) { (void)width; return emu_shfl(v, (unsigned)lane & 63u); }
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; const unsigned l = emu_lane(); return emu_shfl(v, l + d < 64 ? l + d : l); }

#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
// v_permlane32_swap_b32 vdst, vsrc: the upper 32 lanes of vdst trade places with the lower 32 lanes of vsrc
#define __builtin_amdgcn_permlane32_swap(vdst, vsrc, fi, bc) emu_permlane32_swap((vdst), (vsrc))
#define __builtin_amdgcn_readfirstlane(v) ((decltype(v))emu_wave_exchange((uint32_t)(v), 0))
// v_alignbit_b32: low 32 bits of ({hi, lo} >> (shift & 31))
inline uint32_t emu_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
#define __builtin_amdgcn_alignbit(hi, lo, sh) emu_alignbit((hi), (lo), (sh))
// v_bfe_i32 / v_bfe_u32: `width` bits of src from bit `off`, sign- / zero-extended (width 0 gives 0)
inline int emu_sbfe(int src, unsigned off, unsigned width) {
  off &= 31;
  width &= 31;
  if (!width) return 0;
  const uint32_t f = ((uint32_t)src >> off) & ((width == 32 ? 0u : (1u << width)) - 1u);
  const uint32_t sign = 1u << (width - 1);
  return (int)((f ^ sign) - sign);
}
inline uint32_t emu_ubfe(uint32_t src, unsigned off, unsigned width) {
  off &= 31;
  width &= 31;
  if (!width) return 0;
  return (src >> off) & ((1u << width) - 1u);
}
#define __builtin_amdgcn_sbfe(src, off, width) emu_sbfe((src), (off), (width))
#define __builtin_amdgcn_ubfe(src, off, width) emu_ubfe((src), (off), (width))
// v_mfma_i32_16x16x64_i8 (one wave): D[16][16] = A[16][64] * B[64][16] + C, int8 operands, int32 sums.  What the emulation relies
// on: lane l holds 16 bytes of row l % 16 of A and 16 bytes of column l % 16 of B, both for the SAME 16 values of k (the group
// l / 16 selects which); byte t of a lane of A meets byte t of the lane of B in the same group.  Which k a byte stands for inside
// its group does not change a sum over k.  Of C / D lane l holds column l % 16, rows 4 (l / 16) .. + 3.  (The kernels' own operand
// packing was validated on the GPU against the oracle; the batched tests pass here with this statement of it.)
emu_i32x4 emu_mfma_i32_16x16x64_i8(emu_i32x4 a, emu_i32x4 b, emu_i32x4 c);
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, cbsz, abid, blgp) emu_mfma_i32_16x16x64_i8((a), (b), (c))

// ---- host API ----------------------------------------------------------------------------------------------------------
// "Device memory" is host memory.  Streams are queues of operations (emu_streams.cpp).  SPIRAL_EMU_STREAMS selects when they run:
//   eager (default)  every operation runs when it is enqueued -- the strongest ordering there is;
//   lazy             nothing runs until the host waits for something; the executor then prefers the stream whose next
//                    operation was enqueued LAST among those that may run (events honoured) -- the order a correct program
//                    must survive and a program with a missing event dependency does not;
//   random:SEED      as lazy, choosing at random;
//   starve:K         as lazy, but the K-th stream created (0 = the NULL stream) only runs when no other stream can: everything
//                    the other streams do without waiting for it happens before it.  A missing dependency of stream B on
//                    stream A shows under starve:A, every time.
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
struct emu_stream;
typedef emu_stream* hipStream_t;
struct emu_event;
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum : unsigned { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2, hipHostMallocDefault = 0,
                  hipDeviceMallocDefault = 0, hipDeviceMallocContiguous = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
  int multiProcessorCount = 256;
  char gcnArchName[32] = "host-emulation";
  size_t totalGlobalMem = (size_t)1 << 36;
};

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "emulated HIP error"; }
inline const char* hipGetErrorName(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipErrorEmulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* fr, size_t* tot);   // emu_streams.cpp: 32 GiB free unless a device-memory budget is set (emu_set_device_budget)
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeSharedMemPerBlockOptin = 97 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 160 * 1024; return hipSuccess; }  // gfx950: 160 KiB of LDS per workgroup
hipError_t emu_malloc(void** p, size_t bytes, bool pinned_host);
template <typename T>
inline hipError_t hipMalloc(T** p, size_t bytes) { return emu_malloc((void**)p, bytes, false); }
template <typename T>
inline hipError_t hipExtMallocWithFlags(T** p, size_t bytes, unsigned) { return emu_malloc((void**)p, bytes, false); }
template <typename T>
inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned = 0) { return emu_malloc((void**)p, bytes, true); }
hipError_t hipFree(void* p);      // waits for the device first, as the real one does
hipError_t hipHostFree(void* p);
hipError_t hipDeviceSynchronize();
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);  // ordered after the NULL stream only
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
inline hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamDefault); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);

// an operation of the caller's own on a stream (the RCCL stand-in queues its collectives this way)
void emu_enqueue(hipStream_t s, std::function<void()> op);
// a launch is one operation of its stream; grid / block / LDS size are checked when it is enqueued (as hipLaunchKernelGGL
// reports a bad configuration at once), every workgroup has run to completion when the operation ends
void emu_enqueue_launch(hipStream_t s, dim3 grid, dim3 block, size_t dynamic_lds_bytes, std::function<void()> work_item);
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  emu_enqueue_launch((stream), dim3(grid), dim3(block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); })

// ---- keyword macros (last: nothing from the standard library is parsed after this point by way of this header) -----------
// Inline assembly.  `asm volatile("" : "+v"(x))` (an optimisation barrier; the gfx950 constraints do not exist here) becomes an
// empty statement.  `asm("...")` with text: the product headers contain exactly ONE such statement outside the placement probe
// (wave_ntt.hpp, ct_bfly_batch: four v_mad_u64_u32, nl[b + i] += qt[b + i] * q as 64-bit sums); it is restated here BY OPERAND
// NAME -- any other one fails to compile (unknown names) instead of computing something else.  That one line of arithmetic is
// therefore the emulation's, not the product's.
static const int EMU_ASM_ = 0;
#define asm (void)EMU_ASM_
#define EMU_ASM_(...)                                                                   \
  [&] {                                                                                 \
    for (int i_ = 0; i_ < 4; i_++) nl[b + i_] += (uint64_t)qt[b + i_] * (uint64_t)q;    \
  }()
#define volatile(...)
