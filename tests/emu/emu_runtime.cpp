// TEST INFRASTRUCTURE: the execution model behind tests/emu/hip/hip_runtime.h.
//
// A workgroup = one fiber per work-item on ONE host thread, run round-robin by a central scheduler: a fiber runs until it
// waits (workgroup barrier, wave barrier, lane exchange) or returns; a wait yields to the scheduler until the barrier's
// generation has moved.  The workgroups of a launch are dealt to a small pool of host threads (LDS = thread_local storage, so
// one workgroup per host thread at a time).  Deliberately simple: no divergence model, no memory-ordering model, no timing.
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

// x86-64 context switch: callee-saved registers and the stack pointer (written with the __asm__ spelling: the header below
// turns the `asm` keyword into a macro)
extern "C" void emu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size emu_switch,.-emu_switch\n");

#include <hip/hip_runtime.h>

// AddressSanitizer build (build_emulated_library.py --asan): tell it about the stack switches
#if defined(__has_feature)
#if __has_feature(address_sanitizer) && defined(EMU_ASAN_STACKS)  // (only needed when stack variables are instrumented too)
#define EMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif
#endif

// the arrays behind `extern __shared__` (emu_library.cpp registers one getter per name: the address is per host thread)
typedef unsigned char* (*emu_lds_getter)();
static emu_lds_getter g_lds_arrays[8];
static int g_n_lds_arrays = 0;
void emu_register_dynamic_lds(emu_lds_getter f) {
  if (g_n_lds_arrays < 8) g_lds_arrays[g_n_lds_arrays++] = f;
}

namespace {
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr size_t LDS_CANARY = 64;
constexpr unsigned MAX_THREADS = 1024;

struct Fiber {
  emu_workitem wi;
  void* sp = nullptr;
  bool done = false;
  const void* stack_lo = nullptr;  // (ASan annotations)
  void* fake = nullptr;
};
struct Barrier {
  unsigned count = 0, gen = 0;
};
struct Block {
  std::vector<Fiber> fibers;
  unsigned n = 0;
  unsigned cur = 0;
  void* sched_sp = nullptr;
  const void* sched_lo = nullptr;  // (ASan annotations: the scheduler's own stack)
  size_t sched_size = 0;
  void* sched_fake = nullptr;
  const std::function<void()>* body = nullptr;
  Barrier all;
  Barrier wave[MAX_THREADS / 64];
  unsigned long events = 0;  // barrier arrivals: what "progress" means for a workgroup whose fibers are all waiting
  int or_acc = 0;
  uint32_t xch_a[MAX_THREADS / 64][64], xch_b[MAX_THREADS / 64][64];
  char* stacks = nullptr;
  size_t stacks_for = 0;
  std::string error;
  ~Block() {
    if (stacks) munmap(stacks, stacks_for * STACK_BYTES);
  }
};
int g_schedule = 0;  // 0 forward, 1 reverse, 2 random
unsigned long long g_schedule_seed = 1;
struct ScheduleInit {
  ScheduleInit() {
    if (const char* e = std::getenv("SPIRAL_EMU_SCHEDULE")) {
      if (!std::strncmp(e, "reverse", 7)) g_schedule = 1;
      if (!std::strncmp(e, "random", 6)) {
        g_schedule = 2;
        if (e[6] == ':') g_schedule_seed = std::strtoull(e + 7, nullptr, 10);
      }
    }
  }
} g_schedule_init;
thread_local Block* tl_block = nullptr;
thread_local Fiber* tl_fiber = nullptr;
thread_local emu_workitem tl_host_item;  // threadIdx etc. read outside a launch

void yield() {
  Block* b = tl_block;
#ifdef EMU_ASAN
  Fiber& f = b->fibers[b->cur];
  __sanitizer_start_switch_fiber(f.done ? nullptr : &f.fake, b->sched_lo, b->sched_size);
  emu_switch(&f.sp, b->sched_sp);
  __sanitizer_finish_switch_fiber(f.fake, nullptr, nullptr);
#else
  emu_switch(&b->fibers[b->cur].sp, b->sched_sp);
#endif
}
void wait_on(Barrier& bar, unsigned n) {
  const unsigned g = bar.gen;
  tl_block->events++;
  if (++bar.count == n) {
    bar.count = 0;
    bar.gen++;
    return;
  }
  while (bar.gen == g) yield();
}
extern "C" void emu_fiber_entry() {
  Block* b = tl_block;
  Fiber* f = tl_fiber;
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &b->sched_lo, &b->sched_size);
#endif
  try {
    (*b->body)();
  } catch (const std::exception& e) {
    if (b->error.empty()) b->error = e.what();
  }
  f->done = true;
  for (;;) yield();  // never scheduled again
}
void run_block(Block& b, dim3 grid, dim3 block, size_t lds_bytes, unsigned bx, unsigned by, unsigned bz, const std::function<void()>& body) {
  const unsigned n = block.x * block.y * block.z;
  // a kernel may use `lds_bytes` of its dynamic LDS array: the bytes right behind them must come back untouched
  for (int i = 0; i < g_n_lds_arrays; i++) std::memset(g_lds_arrays[i]() + lds_bytes, 0xC7, LDS_CANARY);
  if (n == 0 || n > MAX_THREADS) throw std::invalid_argument("emulated workgroup size out of range");
  if (b.stacks_for < n) {
    if (b.stacks) munmap(b.stacks, b.stacks_for * STACK_BYTES);
    b.stacks = (char*)mmap(nullptr, (size_t)n * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (b.stacks == MAP_FAILED) {
      b.stacks = nullptr;
      b.stacks_for = 0;
      throw std::bad_alloc();
    }
    b.stacks_for = n;
  }
  b.fibers.assign(n, Fiber());
  b.n = n;
  b.body = &body;
  b.all = Barrier();
  for (auto& w : b.wave) w = Barrier();
  b.or_acc = 0;
  b.error.clear();
  for (unsigned t = 0; t < n; t++) {
    Fiber& f = b.fibers[t];
    f.wi.tid.x = t % block.x;
    f.wi.tid.y = (t / block.x) % block.y;
    f.wi.tid.z = t / (block.x * block.y);
    f.wi.bid.x = bx;
    f.wi.bid.y = by;
    f.wi.bid.z = bz;
    f.wi.bdim = block;
    f.wi.gdim = grid;
    // initial frame: six callee-saved registers, the entry point, a null return address (entry sees rsp = 8 mod 16)
    uintptr_t top = (uintptr_t)(b.stacks + (size_t)(t + 1) * STACK_BYTES);
    top &= ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;
    *--sp = (void*)&emu_fiber_entry;
    for (int i = 0; i < 6; i++) *--sp = nullptr;
    f.sp = sp;
    f.stack_lo = b.stacks + (size_t)t * STACK_BYTES;
  }
  tl_block = &b;
  // the order in which the work-items of a round run: 0 .. n-1, reversed, or shuffled every round (SPIRAL_EMU_SCHEDULE).  A
  // kernel whose result depends on it has a data race (a missing barrier between an LDS write and another work-item's read)
  std::vector<unsigned> order(n);
  for (unsigned t = 0; t < n; t++) order[t] = g_schedule == 1 ? n - 1 - t : t;
  unsigned long long rng = g_schedule_seed ^ ((unsigned long long)bx * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)by << 40) ^ bz;
  unsigned alive = n;
  unsigned idle_rounds = 0;
  b.events = 0;
  while (alive) {
    unsigned progressed = 0;
    const unsigned long ev0 = b.events;
    if (g_schedule == 2) {  // a fresh order every round
      for (unsigned i = n - 1; i > 0; i--) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        std::swap(order[i], order[(unsigned)((rng >> 33) % (i + 1))]);
      }
    }
    for (unsigned oi = 0; oi < n; oi++) {
      const unsigned t = order[oi];
      Fiber& f = b.fibers[t];
      if (f.done) continue;
      b.cur = t;
      tl_fiber = &f;
#ifdef EMU_ASAN
      __sanitizer_start_switch_fiber(&b.sched_fake, f.stack_lo, STACK_BYTES);
      emu_switch(&b.sched_sp, f.sp);
      __sanitizer_finish_switch_fiber(b.sched_fake, nullptr, nullptr);
#else
      emu_switch(&b.sched_sp, f.sp);
#endif
      if (f.done) {
        alive--;
        progressed++;
      }
    }
    // a round in which no fiber finished and none ARRIVED at a barrier: every live fiber waits for something that can no
    // longer happen (work-items that left before a barrier the others reach) -- fail instead of hanging the test
    if (progressed || b.events != ev0) {
      idle_rounds = 0;
    } else if (++idle_rounds > 2) {
      b.error = "emulated workgroup made no progress (a barrier some work-items never reach?)";
      break;
    }
  }
  tl_fiber = nullptr;
  tl_block = nullptr;
  if (b.error.empty())
    for (int i = 0; i < g_n_lds_arrays; i++) {
      const unsigned char* c = g_lds_arrays[i]() + lds_bytes;
      for (size_t k = 0; k < LDS_CANARY; k++)
        if (c[k] != 0xC7) {
          b.error = "a workgroup wrote past the dynamic LDS size of its launch (" + std::to_string(lds_bytes) + " bytes requested)";
          break;
        }
    }
  if (!b.error.empty()) throw std::runtime_error(b.error);
}

// ---- host threads that run the workgroups of a launch ---------------------------------------------------------------------
struct Pool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  bool stop = false;
  // the launch in flight
  unsigned long epoch = 0;
  dim3 grid, block;
  size_t lds_bytes = 0;
  const std::function<void()>* body = nullptr;
  std::atomic<unsigned long> next{0};
  unsigned long total = 0;
  unsigned busy = 0;
  std::string error;
  Pool() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("SPIRAL_EMU_THREADS")) n = (unsigned)std::atoi(e);
    if (n < 1) n = 1;
    if (n > 32) n = 32;
    for (unsigned i = 0; i < n; i++) threads.emplace_back([this] { worker(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(m);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : threads) t.join();
  }
  void worker() {
    Block blk;
    unsigned long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m);
      cv_work.wait(lk, [&] { return stop || epoch != seen; });
      if (stop) return;
      seen = epoch;
      const dim3 g = grid, b = block;
      const size_t lds = lds_bytes;
      const std::function<void()>* fn = body;
      lk.unlock();
      std::string err;
      for (;;) {
        const unsigned long i = next.fetch_add(1);
        if (i >= total) break;
        const unsigned bx = (unsigned)(i % g.x), by = (unsigned)((i / g.x) % g.y), bz = (unsigned)(i / ((unsigned long)g.x * g.y));
        try {
          run_block(blk, g, b, lds, bx, by, bz, *fn);
        } catch (const std::exception& e) {
          if (err.empty()) err = e.what();
          next.store(total);  // abandon the rest of the launch
        }
      }
      lk.lock();
      if (!err.empty() && error.empty()) error = err;
      if (--busy == 0) cv_done.notify_all();
    }
  }
  void launch(dim3 g, dim3 b, size_t lds, const std::function<void()>& fn) {
    std::unique_lock<std::mutex> lk(m);
    grid = g;
    block = b;
    lds_bytes = lds;
    body = &fn;
    total = (unsigned long)g.x * g.y * g.z;
    next.store(0);
    busy = (unsigned)threads.size();
    error.clear();
    epoch++;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return busy == 0; });
    body = nullptr;
    if (!error.empty()) throw std::runtime_error("emulated launch failed: " + error);
  }
};
Pool& pool() {
  static Pool p;
  return p;
}
std::mutex g_launch_mutex;  // one launch at a time (host threads of the library may launch concurrently)

// dynamic LDS of the launch in flight: the arrays behind `extern __shared__` are thread_local objects of fixed size
size_t g_dynamic_lds_limit = 160 * 1024;
}  // namespace

emu_workitem* emu_self() { return tl_fiber ? &tl_fiber->wi : &tl_host_item; }
unsigned emu_lane() { return tl_fiber->wi.tid.x & 63u; }

void emu_syncthreads() { wait_on(tl_block->all, tl_block->n); }
int emu_syncthreads_or(int v) {
  Block* b = tl_block;
  if (v) b->or_acc = 1;
  wait_on(b->all, b->n);
  const int r = b->or_acc;
  wait_on(b->all, b->n);
  b->or_acc = 0;  // everyone has read it; all write the same value
  wait_on(b->all, b->n);
  return r;
}
static unsigned wave_size(Block* b, unsigned w) { return std::min(64u, b->n - 64u * w); }
static unsigned flat_tid() {
  const emu_workitem& wi = tl_fiber->wi;
  return wi.tid.x + wi.bdim.x * (wi.tid.y + wi.bdim.y * wi.tid.z);
}
void emu_wave_barrier() {
  Block* b = tl_block;
  const unsigned w = flat_tid() >> 6;
  wait_on(b->wave[w], wave_size(b, w));
}
emu_u32x2 emu_permlane32_swap(uint32_t vdst, uint32_t vsrc) {
  Block* b = tl_block;
  const unsigned t = flat_tid(), w = t >> 6, lane = t & 63;
  b->xch_a[w][lane] = vdst;
  b->xch_b[w][lane] = vsrc;
  wait_on(b->wave[w], wave_size(b, w));
  emu_u32x2 r;
  r[0] = lane < 32 ? vdst : b->xch_b[w][lane - 32];
  r[1] = lane < 32 ? b->xch_a[w][lane + 32] : vsrc;
  wait_on(b->wave[w], wave_size(b, w));
  return r;
}
uint32_t emu_wave_exchange(uint32_t mine, unsigned from_lane) {
  Block* b = tl_block;
  const unsigned t = flat_tid(), w = t >> 6, lane = t & 63;
  b->xch_a[w][lane] = mine;
  wait_on(b->wave[w], wave_size(b, w));
  const uint32_t r = b->xch_a[w][from_lane & 63];
  wait_on(b->wave[w], wave_size(b, w));
  return r;
}
emu_i32x4 emu_mfma_i32_16x16x64_i8(emu_i32x4 a, emu_i32x4 bb, emu_i32x4 c) {
  Block* b = tl_block;
  const unsigned t = flat_tid(), w = t >> 6, lane = t & 63;
  // operands of the 64 lanes through a per-wave staging area (two exchanges of four dwords each)
  static thread_local int8_t A[MAX_THREADS / 64][64][16], B[MAX_THREADS / 64][64][16];
  std::memcpy(A[w][lane], &a, 16);
  std::memcpy(B[w][lane], &bb, 16);
  wait_on(b->wave[w], wave_size(b, w));
  const unsigned j = lane & 15, rb = 4 * (lane >> 4);
  emu_i32x4 d = c;
  for (unsigned r = 0; r < 4; r++) {
    const unsigned i = rb + r;
    int s = 0;
    for (unsigned g = 0; g < 4; g++)
      for (unsigned k = 0; k < 16; k++) s += (int)A[w][i + 16 * g][k] * (int)B[w][j + 16 * g][k];
    d[r] += s;
  }
  wait_on(b->wave[w], wave_size(b, w));
  return d;
}

void emu_launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()>& work_item) {
  std::lock_guard<std::mutex> g(g_launch_mutex);
  pool().launch(grid, block, dynamic_lds_bytes, work_item);
}
void emu_check_launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes) {
  if (dynamic_lds_bytes > g_dynamic_lds_limit) throw std::runtime_error("emulated launch: more dynamic LDS than a CU has");
  if ((unsigned long)grid.x * grid.y * grid.z == 0) throw std::runtime_error("emulated launch: empty grid");
  const unsigned long n = (unsigned long)block.x * block.y * block.z;
  if (n == 0 || n > MAX_THREADS) throw std::runtime_error("emulated launch: workgroup size out of range");
}
