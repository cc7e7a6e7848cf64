// TEST INFRASTRUCTURE: streams, events and copies of the emulated device (see hip/hip_runtime.h).
//
// A stream is a queue of operations; an event is a ticket counter.  In the default mode every operation runs when it is
// enqueued.  With SPIRAL_EMU_STREAMS=lazy or random:SEED nothing runs until the host WAITS (stream / event / device
// synchronisation, a blocking copy, hipFree), and the executor then runs the queued operations in an order chosen to be as
// unkind as the program's own dependencies allow: among the streams whose next operation may run (its event waits satisfied) it
// prefers the one enqueued LAST (or a random one).  A host pipeline with a missing hipStreamWaitEvent / synchronisation passes on
// a GPU most of the time and fails here every time.
#include <hip/hip_runtime.h>

#include <map>
#include <set>

void emu_launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()>& work_item);  // emu_runtime.cpp
void emu_check_launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes);

struct emu_event {
  unsigned long recorded = 0;   // tickets handed out by hipEventRecord
  unsigned long completed = 0;  // tickets whose record operation has run
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
};
namespace {
struct Op {
  unsigned long seq = 0;                // global enqueue order
  std::function<void()> run;            // empty for pure waits
  emu_event* wait = nullptr;            // runnable once wait->completed >= wait_ticket
  unsigned long wait_ticket = 0;
};
}  // namespace
struct emu_stream {
  std::deque<Op> q;
  int id = 0;             // order of creation (0 = the NULL stream): what "starve:K" names
  bool blocking = false;  // created without hipStreamNonBlocking: ordered with the NULL stream (legacy semantics)
};

namespace {
std::recursive_mutex g_m;
int g_mode = -1;  // 0 eager, 1 lazy (latest first), 2 random, 3 starve one stream
int g_starve = 0;
int g_next_id = 1;
unsigned long long g_rng = 88172645463325252ull;
unsigned long g_seq = 0;
emu_stream g_null;  // the NULL stream
std::set<emu_stream*> g_streams;
std::set<const void*> g_pinned;  // hipHostMalloc blocks: async copies to / from them really are asynchronous

int mode() {
  if (g_mode < 0) {
    g_mode = 0;
    if (const char* e = std::getenv("SPIRAL_EMU_STREAMS")) {
      if (!std::strncmp(e, "lazy", 4)) g_mode = 1;
      if (!std::strncmp(e, "random", 6)) {
        g_mode = 2;
        if (e[6] == ':') g_rng ^= std::strtoull(e + 7, nullptr, 10) * 0x9E3779B97F4A7C15ull;
      }
      if (!std::strncmp(e, "starve:", 7)) {
        g_mode = 3;
        g_starve = std::atoi(e + 7);
      }
    }
  }
  return g_mode;
}
emu_stream* S(hipStream_t s) { return s ? s : &g_null; }
bool g_consumed_wait = false;  // head_runnable dropped a satisfied wait: that is progress too
bool head_runnable(emu_stream* s) {
  while (!s->q.empty()) {
    Op& o = s->q.front();
    if (o.wait) {
      if (o.wait->completed < o.wait_ticket) return false;
      if (!o.run) {  // a satisfied pure wait
        s->q.pop_front();
        g_consumed_wait = true;
        continue;
      }
    }
    return true;
  }
  return false;
}
// run one operation of some stream; false if nothing can run
bool step() {
  std::vector<emu_stream*> ready;
  g_consumed_wait = false;
  if (head_runnable(&g_null)) ready.push_back(&g_null);
  for (emu_stream* s : g_streams)
    if (head_runnable(s)) ready.push_back(s);
  if (ready.empty()) return g_consumed_wait;
  emu_stream* pick = ready[0];
  if (g_mode == 2) {
    g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
    pick = ready[(size_t)((g_rng >> 33) % ready.size())];
  } else {
    // latest first; in starve mode the K-th stream created only runs when nothing else can -- whatever the other streams do
    // without waiting for it, they do before it
    pick = nullptr;
    for (emu_stream* s : ready) {
      if (g_mode == 3 && s->id == g_starve) continue;
      if (!pick || s->q.front().seq > pick->q.front().seq) pick = s;
    }
    if (!pick) pick = ready[0];
  }
  Op o = std::move(pick->q.front());
  pick->q.pop_front();
  static const bool trace = std::getenv("SPIRAL_EMU_TRACE") != nullptr;
  if (trace) std::fprintf(stderr, "emu: op %lu of stream %p (%zu streams ready)\n", o.seq, (void*)pick, ready.size());
  if (o.run) o.run();
  return true;
}
template <typename Pred>
void run_until(Pred&& done, const char* what = "") {
  while (!done()) {
    if (!step()) {
      std::string msg = std::string("emulated device: ") + what + ": the host waits for work that can never run (an event waited for before it is recorded?)";
      auto describe = [&](emu_stream* s) {
        if (s->q.empty()) return;
        const Op& o = s->q.front();
        char b[160];
        std::snprintf(b, sizeof(b), "; stream %d: %zu queued, head op %lu%s", s->id, s->q.size(), o.seq, o.wait ? " waits" : "");
        msg += b;
        if (o.wait) {
          std::snprintf(b, sizeof(b), " for ticket %lu of an event with %lu recorded / %lu completed", o.wait_ticket, o.wait->recorded, o.wait->completed);
          msg += b;
        }
      };
      describe(&g_null);
      for (emu_stream* s : g_streams) describe(s);
      throw std::runtime_error(msg);
    }
  }
}
void drain(emu_stream* s) { run_until([&] { return s->q.empty(); }, "stream drain"); }
void drain_all() {
  run_until([&] {
    if (!g_null.q.empty()) return false;
    for (emu_stream* s : g_streams)
      if (!s->q.empty()) return false;
    return true;
  }, "device drain");
}
// legacy NULL-stream semantics: an operation on the NULL stream is ordered after everything enqueued so far on the blocking
// streams, and the other way round.  Modelled by draining (stronger than needed, and only for streams created blocking).
void order_with_null(emu_stream* s) {
  if (s == &g_null) {
    for (emu_stream* b : g_streams)
      if (b->blocking) drain(b);
  } else if (s->blocking) {
    drain(&g_null);
  }
}
void enqueue(hipStream_t st, std::function<void()> f) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  emu_stream* s = S(st);
  if (mode() == 0) {
    f();
    return;
  }
  order_with_null(s);
  Op o;
  o.seq = ++g_seq;
  o.run = std::move(f);
  s->q.push_back(std::move(o));
}
// blocks are looked up by START address: the product copies to / from the start of its pinned staging buffers; an interior
// pointer of a pinned block counts as pageable, which only makes the copy MORE synchronous than the GPU's
bool pinned(const void* p) { return g_pinned.count(p) != 0; }
}  // namespace

// A budget of "device" memory (0 = none): with one set, hipMalloc fails with hipErrorOutOfMemory once the live device bytes would
// exceed it and hipMemGetInfo reports what is left -- how the tests drive the library's out-of-memory paths (the batched call's
// ladder), which no GPU run reaches on purpose.  Pinned host memory is not counted.
namespace {
size_t g_dev_budget = 0, g_dev_live = 0;
std::map<void*, size_t> g_dev_sizes;
}  // namespace
extern "C" void emu_set_device_budget(size_t bytes) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  g_dev_budget = bytes;
}
extern "C" size_t emu_device_live_bytes() {
  std::lock_guard<std::recursive_mutex> g(g_m);
  return g_dev_live;
}
hipError_t hipMemGetInfo(size_t* fr, size_t* tot) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (g_dev_budget) {
    *tot = g_dev_budget;
    *fr = g_dev_budget > g_dev_live ? g_dev_budget - g_dev_live : 0;
  } else {
    *fr = (size_t)1 << 35;
    *tot = (size_t)1 << 36;
  }
  return hipSuccess;
}

hipError_t emu_malloc(void** p, size_t bytes, bool pinned_host) {
  void* q = nullptr;
  if (!pinned_host) {
    std::lock_guard<std::recursive_mutex> g(g_m);
    if (g_dev_budget && g_dev_live + bytes > g_dev_budget) return hipErrorOutOfMemory;
  }
  // + 16 bytes: the host compiler loads a three-dword vector (global_load_dwordx3 on the GPU: 12 bytes) as 16 bytes, so the
  // last lane of the last PACKED unit of a buffer touches one dword past its end
  if (posix_memalign(&q, 256, bytes + 16) != 0) return hipErrorOutOfMemory;
  // fresh device memory holds whatever it held: here a byte pattern, so that a kernel that relies on it being zero is wrong every
  // time instead of on the day the allocator returns a used page (SPIRAL_EMU_POISON=-1 leaves it as malloc returns it)
  static const int poison = std::getenv("SPIRAL_EMU_POISON") ? std::atoi(std::getenv("SPIRAL_EMU_POISON")) : 0xA5;
  if (poison >= 0) std::memset(q, poison & 0xff, bytes + 16);
  {
    std::lock_guard<std::recursive_mutex> g(g_m);
    if (pinned_host) {
      g_pinned.insert(q);
    } else {
      g_dev_sizes[q] = bytes;
      g_dev_live += bytes;
    }
  }
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) drain_all();
  auto it = g_dev_sizes.find(p);
  if (it != g_dev_sizes.end()) {
    g_dev_live -= it->second;
    g_dev_sizes.erase(it);
  }
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostFree(void* p) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) drain_all();
  g_pinned.erase(p);
  std::free(p);
  return hipSuccess;
}
hipError_t hipDeviceSynchronize() {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) drain_all();
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) {  // a blocking copy is ordered after the NULL stream (and the blocking streams), NOT after non-blocking ones
    order_with_null(&g_null);
    drain(&g_null);
  }
  if (n) std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) {
    order_with_null(&g_null);
    drain(&g_null);
  }
  if (n) std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() == 0 || n == 0) {
    if (n) std::memmove(d, s, n);
    return hipSuccess;
  }
  const bool h2d = k == hipMemcpyHostToDevice, d2h = k == hipMemcpyDeviceToHost;
  if (h2d && !pinned(s)) {
    // pageable source: the runtime stages it before it returns -- the caller may reuse the buffer at once
    auto staged = std::make_shared<std::vector<char>>((const char*)s, (const char*)s + n);
    enqueue(st, [d, staged] { std::memcpy(d, staged->data(), staged->size()); });
    return hipSuccess;
  }
  if (d2h && !pinned(d)) {
    // pageable destination: the call returns when the data is there
    order_with_null(S(st));
    drain(S(st));
    std::memmove(d, s, n);
    return hipSuccess;
  }
  enqueue(st, [d, s, n] { std::memmove(d, s, n); });
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t st) {
  enqueue(st, [=] {
    for (size_t r = 0; r < height; r++) std::memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  });
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  enqueue(st, [=] {
    if (n) std::memset(d, v, n);
  });
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  auto* st = new emu_stream;
  st->blocking = !(flags & hipStreamNonBlocking);
  st->id = g_next_id++;
  g_streams.insert(st);
  *s = st;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (!s) return hipErrorInvalidValue;
  if (mode() != 0) drain(s);
  g_streams.erase(s);
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) {
    order_with_null(S(s));
    drain(S(s));
  }
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() == 0) return hipSuccess;
  step();  // a polling host makes progress
  return S(s)->q.empty() ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  *e = new emu_event;
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  // operations that still refer to it: let them run first (the real runtime keeps the event alive for them)
  if (mode() != 0) run_until([&] { return e->completed >= e->recorded; }, "hipEventDestroy");
  if (mode() != 0) drain_all();
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  const unsigned long ticket = ++e->recorded;
  enqueue(s, [e, ticket] {
    e->t = std::chrono::steady_clock::now();
    if (e->completed < ticket) e->completed = ticket;
  });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() == 0 || e->recorded == 0) return hipSuccess;  // never recorded: no-op, as on the GPU
  emu_stream* st = S(s);
  order_with_null(st);
  Op o;
  o.seq = ++g_seq;
  o.wait = e;
  o.wait_ticket = e->recorded;  // the most recent record at the time of THIS call
  st->q.push_back(std::move(o));
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) {
    const unsigned long want = e->recorded;
    run_until([&] { return e->completed >= want; }, "hipEventSynchronize");
  }
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() == 0) return hipSuccess;
  step();
  return e->completed >= e->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  std::lock_guard<std::recursive_mutex> g(g_m);
  if (mode() != 0) {
    if (a->completed < a->recorded || b->completed < b->recorded) return hipErrorNotReady;
  }
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
void emu_enqueue(hipStream_t s, std::function<void()> op) { enqueue(s, std::move(op)); }
void emu_enqueue_launch(hipStream_t s, dim3 grid, dim3 block, size_t dynamic_lds_bytes, std::function<void()> work_item) {
  emu_check_launch(grid, block, dynamic_lds_bytes);
  enqueue(s, [grid, block, dynamic_lds_bytes, work_item] { emu_launch(grid, block, dynamic_lds_bytes, work_item); });
}
