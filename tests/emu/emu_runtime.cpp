// TEST INFRASTRUCTURE: a workgroup as host threads.  One std::thread per work-item, pthread barriers for s_barrier and for the
// lockstep points of a wave (LDS hand-over inside a wave, v_permlane32_swap).  Slow and simple on purpose.
#include "emu_runtime.hpp"

#include <pthread.h>

#include <stdexcept>
#include <thread>
#include <vector>

thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace {
struct Group {
  unsigned n = 0;
  pthread_barrier_t all;
  pthread_barrier_t wave[16];
  int or_acc = 0;
  uint32_t swap_a[16][64], swap_b[16][64];
};
Group* g_group = nullptr;
}  // namespace

void emu_syncthreads() { pthread_barrier_wait(&g_group->all); }
int emu_syncthreads_or(int v) {
  if (v) __atomic_store_n(&g_group->or_acc, 1, __ATOMIC_RELAXED);
  pthread_barrier_wait(&g_group->all);
  const int r = __atomic_load_n(&g_group->or_acc, __ATOMIC_RELAXED);
  pthread_barrier_wait(&g_group->all);
  if (threadIdx.x == 0) g_group->or_acc = 0;
  pthread_barrier_wait(&g_group->all);
  return r;
}
void emu_wave_barrier() { pthread_barrier_wait(&g_group->wave[threadIdx.x >> 6]); }
emu_u32x2 emu_permlane32_swap(uint32_t vdst, uint32_t vsrc) {
  const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  g_group->swap_a[w][lane] = vdst;
  g_group->swap_b[w][lane] = vsrc;
  emu_wave_barrier();
  emu_u32x2 r;
  r[0] = lane < 32 ? vdst : g_group->swap_b[w][lane - 32];
  r[1] = lane < 32 ? g_group->swap_a[w][lane + 32] : vsrc;
  emu_wave_barrier();
  return r;
}

namespace emu {
void run_block(unsigned nthreads, unsigned bx, unsigned by, const std::function<void()>& body) {
  if (nthreads == 0 || nthreads > 1024 || (nthreads & 63)) throw std::invalid_argument("workgroup size must be a multiple of 64, at most 1024");
  Group g;
  g.n = nthreads;
  pthread_barrier_init(&g.all, nullptr, nthreads);
  for (unsigned w = 0; w < nthreads / 64; w++) pthread_barrier_init(&g.wave[w], nullptr, 64);
  g_group = &g;
  std::vector<std::thread> ts;
  ts.reserve(nthreads);
  for (unsigned t = 0; t < nthreads; t++)
    ts.emplace_back([&, t] {
      threadIdx.x = t;
      threadIdx.y = threadIdx.z = 0;
      blockIdx.x = bx;
      blockIdx.y = by;
      blockIdx.z = 0;
      blockDim = dim3(nthreads);
      body();
    });
  for (auto& t : ts) t.join();
  g_group = nullptr;
  pthread_barrier_destroy(&g.all);
  for (unsigned w = 0; w < nthreads / 64; w++) pthread_barrier_destroy(&g.wave[w]);
}
}  // namespace emu

void emu_unsupported_asm() { throw std::logic_error("gfx950 inline assembly reached in the host emulation"); }
