// TEST INFRASTRUCTURE: collectives among PROCESSES of this host over POSIX shared memory (see rccl/rccl.h).
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <rccl/rccl.h>

namespace {
constexpr size_t SLOT_BYTES = (size_t)256 << 20;  // staging per rank (sparse: only touched pages exist)
constexpr int MAX_RANKS = 16;
struct Header {
  pthread_barrier_t barrier;
  volatile int ready;    // set by rank 0 once the barrier exists
  volatile int failed;   // a rank that cannot go on says so before the barrier, so that the others do not hang
  int nranks;
};
constexpr size_t HEADER_BYTES = 4096;
}  // namespace

struct emu_nccl_comm {
  int nranks = 1, rank = 0;
  char name[64] = {0};
  char* base = nullptr;
  size_t bytes = 0;
  Header* hdr() const { return reinterpret_cast<Header*>(base); }
  char* slot(int r) const { return base + HEADER_BYTES + (size_t)r * SLOT_BYTES; }
};

static size_t elem(ncclDataType_t t) { return t == ncclUint64 ? 8 : 4; }

ncclResult_t ncclGetVersion(int* version) {
  *version = 0;   // the stand-in: not an RCCL
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "emulated RCCL: shared-memory rendezvous failed";
    case ncclInvalidArgument: return "emulated RCCL: invalid argument (message larger than the staging slot?)";
    default: return "emulated RCCL: error on another rank";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id->internal, 0, sizeof(id->internal));
  unsigned long long rnd = 0;
  if (getrandom(&rnd, sizeof(rnd), 0) != (ssize_t)sizeof(rnd)) rnd = (unsigned long long)time(nullptr) ^ ((unsigned long long)getpid() << 32);
  std::snprintf(id->internal, sizeof(id->internal), "/spiral_emu_rccl_%d_%016llx", (int)getpid(), rnd);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
  auto c = new emu_nccl_comm;
  c->nranks = nranks;
  c->rank = rank;
  std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->bytes = HEADER_BYTES + (size_t)nranks * SLOT_BYTES;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
      if (fd >= 0) close(fd);
      delete c;
      return ncclSystemError;
    }
  } else {
    for (int tries = 0; tries < 60000 && fd < 0; tries++) {  // up to a minute for rank 0 to appear
      fd = shm_open(c->name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->bytes)) {
        close(fd);
        fd = -1;
      }
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) {
      delete c;
      return ncclSystemError;
    }
  }
  c->base = (char*)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  Header* h = c->hdr();
  if (rank == 0) {
    pthread_barrierattr_t a;
    pthread_barrierattr_init(&a);
    pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&h->barrier, &a, (unsigned)nranks);
    pthread_barrierattr_destroy(&a);
    h->nranks = nranks;
    h->failed = 0;
    __atomic_store_n(&h->ready, 1, __ATOMIC_RELEASE);
  } else {
    for (int tries = 0; tries < 60000 && !__atomic_load_n(&h->ready, __ATOMIC_ACQUIRE); tries++) usleep(1000);
    if (!h->ready || h->nranks != nranks) {
      munmap(c->base, c->bytes);
      delete c;
      return ncclSystemError;
    }
  }
  pthread_barrier_wait(&h->barrier);  // everyone is attached: the name can go
  if (rank == 0) shm_unlink(c->name);
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->base) munmap(c->base, c->bytes);
  delete c;
  return ncclSuccess;
}

// copy into the own slot, meet, let `compute` read every slot, meet again (the slots are free for the next collective)
template <typename F>
static ncclResult_t collective(ncclComm_t c, const void* send, size_t send_bytes, F&& compute) {
  Header* h = c->hdr();
  if (send_bytes > SLOT_BYTES) h->failed = 1;
  else std::memcpy(c->slot(c->rank), send, send_bytes);
  pthread_barrier_wait(&h->barrier);
  const bool bad = h->failed != 0;
  if (!bad) compute();
  pthread_barrier_wait(&h->barrier);
  return bad ? ncclInvalidArgument : ncclSuccess;
}

// (a collective is an operation of its stream, like a kernel: it runs after what was enqueued before it there.  A failure
// inside a deferred collective cannot be returned to the caller any more: it is thrown where the host waits.)
static void deferred(hipStream_t s, std::function<ncclResult_t()> f) {
  emu_enqueue(s, [f] {
    if (f() != ncclSuccess) throw std::runtime_error("emulated RCCL: collective failed (message larger than the staging slot?)");
  });
}

ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t, ncclComm_t c, hipStream_t st) {
  const size_t es = elem(t);
  deferred(st, [=]() -> ncclResult_t { return collective(c, send, recvcount * es * (size_t)c->nranks, [&] {
    if (t == ncclUint64) {
      auto* out = (uint64_t*)recv;
      for (size_t i = 0; i < recvcount; i++) {
        uint64_t s = 0;
        for (int k = 0; k < c->nranks; k++) s += ((const uint64_t*)c->slot(k))[(size_t)c->rank * recvcount + i];
        out[i] = s;
      }
    } else {
      auto* out = (uint32_t*)recv;
      for (size_t i = 0; i < recvcount; i++) {
        uint32_t s = 0;
        for (int k = 0; k < c->nranks; k++) s += ((const uint32_t*)c->slot(k))[(size_t)c->rank * recvcount + i];
        out[i] = s;
      }
    }
  }); });
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
  const size_t nb = sendcount * elem(t);
  deferred(st, [=]() -> ncclResult_t { return collective(c, send, nb, [&] {
    for (int k = 0; k < c->nranks; k++) std::memcpy((char*)recv + (size_t)k * nb, c->slot(k), nb);
  }); });
  return ncclSuccess;
}
