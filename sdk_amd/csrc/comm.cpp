// Multi-GPU answer path behind the C ABI: the row-sharded process_query with its exchange step issued by the library
// itself (RCCL over xGMI, linked directly) instead of by the caller.  One process per GPU; every rank calls
// sp_process_query_sharded with the same query, rank 0 receives the response.
//
// Reference shape: the reference is single-node CPU code with no collectives; the host that
// would call this is lib/server's request loop (lib/server/src/bin/server.rs:98-141), which today calls one
// process_query per request (server.rs:650-655).  The flow:
//
//   every rank : Query::deserialize + expand_query pruned to the shard's rows          (sp_query_begin_for_db)
//   per plane p: sweep the shard's rows -> partial residues, columns interleaved by destination rank
//                reduce-scatter (ncclSum, u32) of plane p on the communicator's stream WHILE plane p+1 is swept
//   every rank : % q, from_ntt, the top nu_2 - log2 G fold levels on its num_per / G columns (sp_query_fold_local)
//   all-gather : one ciphertext per plane per rank (G * planes * 32 KiB)
//   rank 0     : the last log2 G fold levels, pack, encode                              (sp_query_finish_gathered)
//
// Everything is ordered with HIP events between the query's stream and the communicator's stream; the host only
// waits once, for the response (rank 0) or for the rank's last enqueued work (other ranks).
// This unit uses nothing but the public C ABI of the library + HIP + RCCL: it is exactly what a host program
// driving sp_query_* itself would do.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spiral_hip.h"

extern "C" void sp_set_last_error_(const char* msg);  // capi.cpp
extern "C" void sp_note_path_(uint64_t bits);         // capi.cpp
extern "C" void sp_shard_split_hint_(int on);         // capi.cpp: may the next sp_query_begin_for_db of this thread split the expansion?

namespace {

struct Fail {
  int rc;
  std::string msg;
};
void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw Fail{SP_E_HIP, std::string(what) + ": " + hipGetErrorString(e)};
  }
}
void nccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw Fail{SP_E_HIP, std::string(what) + ": " + ncclGetErrorString(r)};
}
void sp_ok(int rc, const char* what) {
  if (rc != SP_OK) throw Fail{rc, std::string(what) + ": " + sp_last_error()};
}

}  // namespace

struct sp_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t nccl = nullptr;      // built-in transport
  sp_comm_ops_t ops{};            // custom transport (ops.reduce_scatter_u32 != nullptr)
  bool custom = false;
  hipStream_t stream = nullptr;   // the exchange stream: collectives are enqueued here, in the same order on every rank
  std::vector<hipEvent_t> ev_plane;
  hipEvent_t ev_x = nullptr, ev_f = nullptr, ev_g = nullptr, ev_t[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_rs = nullptr;     // (timing) the last plane's reduce-scatter is done, on the exchange stream
  size_t rs_recv_bytes = 0, ag_send_bytes = 0, planes_last = 0;   // sizes of the last sharded query's collectives (sp_comm_describe)
  void* mine = nullptr;           // this rank's summed chunk of every plane: [plane][r][crt][z][ii / G] u32
  size_t mine_bytes = 0;
  void* gathered = nullptr;       // [g][plane][2][N] u64
  size_t gathered_bytes = 0;
  float ms[3] = {0, 0, 0};
  std::mutex mu;                  // one sharded query at a time per communicator (collective order must match on all ranks)

  int reduce_scatter(const void* send, void* recv, size_t recv_count) {
    if (custom) return ops.reduce_scatter_u32(ops.user, send, recv, recv_count, (void*)stream);
    nccl_ok(ncclReduceScatter(send, recv, recv_count, ncclUint32, ncclSum, nccl, stream), "ncclReduceScatter");
    return 0;
  }
  int all_gather(const void* send, void* recv, size_t send_count) {
    if (custom) return ops.all_gather_u64(ops.user, send, recv, send_count, (void*)stream);
    nccl_ok(ncclAllGather(send, recv, send_count, ncclUint64, nccl, stream), "ncclAllGather");
    return 0;
  }
  void ensure(void*& p, size_t& have, size_t need) {
    if (have >= need) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    have = 0;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      throw Fail{SP_E_OOM, "hipMalloc of the exchange buffer failed"};
    }
    have = need;
  }
  ~sp_comm() {
    if (stream) (void)hipStreamSynchronize(stream);
    if (nccl) (void)ncclCommDestroy(nccl);
    for (auto e : ev_plane)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ev_x, ev_f, ev_g, ev_t[0], ev_t[1], ev_t[2], ev_rs})
      if (e) (void)hipEventDestroy(e);
    if (mine) (void)hipFree(mine);
    if (gathered) (void)hipFree(gathered);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

template <typename F>
static int guarded_comm(F&& f) {
  try {
    f();
    return SP_OK;
  } catch (const Fail& e) {
    sp_set_last_error_(e.msg.c_str());
    return e.rc;
  } catch (const std::exception& e) {
    sp_set_last_error_(e.what());
    return SP_E_ARG;
  }
}

static sp_comm_t* comm_new(int rank, int world) {
  if (world < 1 || world > SP_MAX_ROW_SHARDS || (world & (world - 1)) != 0 || rank < 0 || rank >= world)
    throw Fail{SP_E_ARG, "world must be a power of two <= SP_MAX_ROW_SHARDS and 0 <= rank < world"};
  auto c = new sp_comm();
  try {
    c->rank = rank;
    c->world = world;
    hip_ok(hipGetDevice(&c->device), "hipGetDevice");
    hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
    for (hipEvent_t* e : {&c->ev_x, &c->ev_f, &c->ev_g}) hip_ok(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate");
    for (auto& e : c->ev_t) hip_ok(hipEventCreate(&e), "hipEventCreate");
    hip_ok(hipEventCreate(&c->ev_rs), "hipEventCreate");
  } catch (...) {
    delete c;
    throw;
  }
  return c;
}

extern "C" {

int sp_comm_unique_id(uint8_t* id128) {
  return guarded_comm([&] {
    if (!id128) throw Fail{SP_E_ARG, "null argument"};
    static_assert(SP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, id.internal, SP_COMM_ID_BYTES);
  });
}

sp_comm_t* sp_comm_create(int rank, int world, const uint8_t* id128) {
  sp_comm_t* out = nullptr;
  int rc = guarded_comm([&] {
    if (!id128) throw Fail{SP_E_ARG, "null argument"};
    sp_comm_t* c = comm_new(rank, world);
    ncclUniqueId id;
    memcpy(id.internal, id128, SP_COMM_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
      c->nccl = nullptr;
      delete c;
      nccl_ok(r, "ncclCommInitRank");
    }
    out = c;
  });
  return rc == SP_OK ? out : nullptr;
}

sp_comm_t* sp_comm_create_custom(int rank, int world, const sp_comm_ops_t* ops) {
  sp_comm_t* out = nullptr;
  int rc = guarded_comm([&] {
    if (!ops || !ops->reduce_scatter_u32 || !ops->all_gather_u64) throw Fail{SP_E_ARG, "ops must provide both collectives"};
    sp_comm_t* c = comm_new(rank, world);
    c->ops = *ops;
    c->custom = true;
    out = c;
  });
  return rc == SP_OK ? out : nullptr;
}

void sp_comm_free(sp_comm_t* c) { delete c; }
int sp_comm_rank(const sp_comm_t* c) { return c ? c->rank : -1; }
int sp_comm_world(const sp_comm_t* c) { return c ? c->world : 0; }
void* sp_comm_stream(sp_comm_t* c) { return c ? (void*)c->stream : nullptr; }

// One line of JSON about the communicator and the last sharded query's collectives, for the first run on a multi-GPU node to be
// diagnosable from its log: transport, RCCL version, what one rank sends and receives per plane and per query.
int sp_comm_describe(const sp_comm_t* c, char* buf, size_t cap) {
  if (!c || !buf || cap == 0) return SP_E_ARG;
  int ver = 0;
  if (!c->custom && ncclGetVersion(&ver) != ncclSuccess) ver = -1;
  const int G = c->world;
  const int n = snprintf(buf, cap,
                         "{\"transport\": \"%s\", \"rccl_version\": %d, \"rank\": %d, \"world\": %d, \"device\": %d, "
                         "\"planes\": %zu, \"reduce_scatter_u32\": {\"per_plane_send_bytes\": %zu, \"per_plane_recv_bytes\": %zu, "
                         "\"per_query_link_bytes_ring\": %zu}, \"all_gather_u64\": {\"send_bytes\": %zu, \"recv_bytes\": %zu}, "
                         "\"last_query_ms\": {\"sweeps_with_overlapped_exchange\": %.4f, \"tail_fold_gather\": %.4f, "
                         "\"exposed_exchange_after_last_sweep\": %.4f}}",
                         c->custom ? "custom" : "rccl", ver, c->rank, G, c->device, c->planes_last, c->rs_recv_bytes * (size_t)G,
                         c->rs_recv_bytes, c->planes_last * c->rs_recv_bytes * (size_t)(G - 1), c->ag_send_bytes,
                         c->ag_send_bytes * (size_t)G, (double)c->ms[0], (double)c->ms[1], (double)c->ms[2]);
  return n > 0 && (size_t)n < cap ? SP_OK : SP_E_ARG;
}

int sp_comm_timings(const sp_comm_t* c, float* ms3) {
  if (!c || !ms3) return SP_E_ARG;
  memcpy(ms3, c->ms, sizeof(c->ms));
  return SP_OK;
}

// sizes of the exchange buffers for `h`, allocated now (sp_comm_reserve) or, failing that, by the first sharded query
static void comm_reserve(sp_comm_t* c, const sp_params_t* h) {
  const int G = c->world;
  const size_t planes = (size_t)sp_params_get(h, "instances") * sp_params_get(h, "n") * sp_params_get(h, "n");
  const size_t num_per = (size_t)1 << sp_params_get(h, "db_dim_2");
  if ((size_t)G > num_per) throw Fail{SP_E_ARG, "more ranks than second-dimension columns (num_per): use fewer shards"};
  if (num_per % (size_t)G != 0) throw Fail{SP_E_ARG, "num_per is not a multiple of the number of ranks: the column chunks of the exchange would be ragged"};
  const size_t chunk = 4 * (size_t)sp_params_get(h, "poly_len") * num_per / (size_t)G;   // u32 [r][crt][z][ii / G] of one plane
  const size_t local_words = planes * 2 * (size_t)sp_params_get(h, "poly_len");         // one raw ciphertext per plane
  c->ensure(c->mine, c->mine_bytes, planes * chunk * sizeof(uint32_t));
  c->ensure(c->gathered, c->gathered_bytes, (size_t)G * local_words * sizeof(uint64_t));
  if (c->ev_plane.size() < planes) {
    const size_t old = c->ev_plane.size();
    c->ev_plane.resize(planes, nullptr);
    for (size_t i = old; i < planes; i++) hip_ok(hipEventCreateWithFlags(&c->ev_plane[i], hipEventDisableTiming), "hipEventCreate");
  }
}

int sp_comm_reserve(sp_comm_t* c, const sp_params_t* h) {
  if (!c || !h) {
    sp_set_last_error_("null argument");
    return SP_E_ARG;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  return guarded_comm([&] { comm_reserve(c, h); });
}

namespace {
// One sharded query as three enqueue stages, so that a list of queries can be software-pipelined: while query k's planes are
// swept, query k + 1 is already expanding on its own streams.
struct ShardedRun {
  sp_query_t* q = nullptr;
  hipStream_t main = nullptr;
  size_t planes = 0, local_words = 0;
};
// on_critical_path: nothing of this rank's runs beside the expansion (a single query, the first of a list) -- then its odd subtree
// and GSW side may move beside the sweeps that follow (capi.cpp, expand_split_shards); a list's later queries expand under their
// predecessor's sweeps and keep the one-stream form
void sharded_begin(sp_comm_t* c, const sp_params_t* h, const sp_pp_t* pp, const uint8_t* query, size_t query_len,
                   const sp_db_t* shard, ShardedRun& r, bool on_critical_path) {
  r.planes = (size_t)sp_params_get(h, "instances") * sp_params_get(h, "n") * sp_params_get(h, "n");
  sp_shard_split_hint_(on_critical_path ? 1 : 0);
  r.q = sp_query_begin_for_db(h, pp, query, query_len, shard);
  sp_shard_split_hint_(1);
  if (!r.q) throw Fail{SP_E_ARG, std::string("sp_query_begin_for_db: ") + sp_last_error()};
  r.main = (hipStream_t)sp_query_stream(r.q);
  r.local_words = sp_query_local_cts_words(r.q);
  (void)c;
}
void sharded_sweeps(sp_comm_t* c, const sp_db_t* shard, ShardedRun& r, bool timed) {
  const int G = c->world;
  uint32_t* part = (uint32_t*)sp_query_partial_ptr(r.q);
  if (!part) throw Fail{SP_E_OOM, std::string("partial buffer: ") + sp_last_error()};
  const size_t words = sp_query_partial_words(r.q), pw = words / r.planes, chunk = pw / (size_t)G;
  // comm_reserve sized `mine` / `gathered` from the parameters alone; this is the partial layout the query really produced.
  // A disagreement (a future output layout, a column shard) must fail here, not write past `mine`.
  if (words % r.planes != 0 || pw % (size_t)G != 0 || r.planes * chunk * sizeof(uint32_t) > c->mine_bytes ||
      (size_t)G * r.local_words * sizeof(uint64_t) > c->gathered_bytes)
    throw Fail{SP_E_ARG, "internal: the query's partial buffer does not match the reserved exchange buffers"};
  if (timed) hip_ok(hipEventRecord(c->ev_t[0], r.main), "hipEventRecord");
  for (size_t pl = 0; pl < r.planes; pl++) {
    sp_ok(sp_query_sweep_scatter_plane(r.q, shard, G, (int)pl), "sp_query_sweep_scatter_plane");
    hip_ok(hipEventRecord(c->ev_plane[pl], r.main), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(c->stream, c->ev_plane[pl], 0), "hipStreamWaitEvent");
    // plane pl's region is [g][r][crt][z][ii / G]: rank g receives the sum of everybody's chunk g
    if (c->reduce_scatter(part + pl * pw, (uint32_t*)c->mine + pl * chunk, chunk) != 0)
      throw Fail{SP_E_HIP, "custom reduce_scatter_u32 failed"};
  }
  if (timed) hip_ok(hipEventRecord(c->ev_t[1], r.main), "hipEventRecord");
  if (timed) hip_ok(hipEventRecord(c->ev_rs, c->stream), "hipEventRecord");
  c->rs_recv_bytes = chunk * sizeof(uint32_t);
  c->ag_send_bytes = r.local_words * sizeof(uint64_t);
  c->planes_last = r.planes;
  hip_ok(hipEventRecord(c->ev_x, c->stream), "hipEventRecord");
  hip_ok(hipStreamWaitEvent(r.main, c->ev_x, 0), "hipStreamWaitEvent");
}
void sharded_finish(sp_comm_t* c, ShardedRun& r, uint8_t* out, size_t out_cap, size_t* out_len, bool timed) {
  const int G = c->world;
  sp_ok(sp_query_fold_local(r.q, c->mine, G), "sp_query_fold_local");
  hip_ok(hipEventRecord(c->ev_f, r.main), "hipEventRecord");
  hip_ok(hipStreamWaitEvent(c->stream, c->ev_f, 0), "hipStreamWaitEvent");
  if (c->all_gather(sp_query_local_cts_ptr(r.q), c->gathered, r.local_words) != 0)
    throw Fail{SP_E_HIP, "custom all_gather_u64 failed"};
  hip_ok(hipEventRecord(c->ev_g, c->stream), "hipEventRecord");
  hip_ok(hipStreamWaitEvent(r.main, c->ev_g, 0), "hipStreamWaitEvent");
  if (timed) hip_ok(hipEventRecord(c->ev_t[2], r.main), "hipEventRecord");
  if (c->rank == 0) {
    sp_ok(sp_query_finish_gathered(r.q, c->gathered, G, out, out_cap, out_len), "sp_query_finish_gathered");
  } else {
    *out_len = 0;
    sp_ok(sp_query_sync(r.q), "sp_query_sync");
  }
  // the exchange buffers (mine / gathered) are reused by the next query: its reduce-scatters are enqueued on c->stream
  // after this query's all-gather, and this query's local fold (reader of `mine`) is complete once the host returned
  // from finish / sync above
  hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
}
void note_transport(const sp_comm_t* c) {
  // PATH_RCCL (bit 19): the collectives were RCCL's, issued by the library; PATH_CUSTOM_TRANSPORT (bit 25): the host's
  sp_note_path_(c->custom ? (1ull << 25) : (1ull << 19));
}
}  // namespace

int sp_process_query_sharded(sp_comm_t* c, const sp_params_t* h, const sp_pp_t* pp, const uint8_t* query,
                             size_t query_len, const sp_db_t* shard, uint8_t* out, size_t out_cap, size_t* out_len) {
  if (!c || !h || !pp || !query || !shard || !out_len || (c->rank == 0 && !out)) {
    sp_set_last_error_("null argument");
    return SP_E_ARG;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  ShardedRun r;
  int rc = guarded_comm([&] {
    int dev = 0;
    hip_ok(hipGetDevice(&dev), "hipGetDevice");
    if (dev != c->device) throw Fail{SP_E_ARG, "the communicator was created on another HIP device"};
    comm_reserve(c, h);   // no-op after sp_comm_reserve / the first query with these params
    sharded_begin(c, h, pp, query, query_len, shard, r, true);
    sharded_sweeps(c, shard, r, true);
    sharded_finish(c, r, out, out_cap, out_len, true);
    // [0] sweep launches incl. the exchanges overlapped with them, [1] exchange tail + local fold + all-gather
    (void)hipEventElapsedTime(&c->ms[0], c->ev_t[0], c->ev_t[1]);
    (void)hipEventElapsedTime(&c->ms[1], c->ev_t[1], c->ev_t[2]);
    // [2] what of the exchange is NOT hidden behind the sweeps: last sweep launch done -> last plane's reduce-scatter done
    if (hipEventElapsedTime(&c->ms[2], c->ev_t[1], c->ev_rs) != hipSuccess || c->ms[2] < 0) c->ms[2] = 0;
    note_transport(c);
  });
  if (r.q) {
    if (rc != SP_OK) {
      (void)hipStreamSynchronize(c->stream);
      (void)sp_query_sync(r.q);
    }
    sp_query_free(r.q);
  }
  return rc;
}

int sp_process_queries_sharded(sp_comm_t* c, const sp_params_t* h, const sp_pp_t* const* pps, const uint8_t* const* queries,
                               const size_t* query_lens, int n, const sp_db_t* shard, uint8_t* out, size_t out_stride,
                               size_t* out_len) {
  if (!c || !h || !pps || !queries || !query_lens || n < 0 || !shard || !out_len || (c->rank == 0 && n > 0 && !out)) {
    sp_set_last_error_("null argument");
    return SP_E_ARG;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  ShardedRun cur, nxt;
  int rc = guarded_comm([&] {
    int dev = 0;
    hip_ok(hipGetDevice(&dev), "hipGetDevice");
    if (dev != c->device) throw Fail{SP_E_ARG, "the communicator was created on another HIP device"};
    *out_len = 0;
    if (n == 0) return;
    comm_reserve(c, h);
    sharded_begin(c, h, pps[0], queries[0], query_lens[0], shard, cur, true);
    for (int k = 0; k < n; k++) {
      sharded_sweeps(c, shard, cur, k == n - 1);
      // query k + 1 expands (on its own workspace's streams) while query k's planes are swept and exchanged
      if (k + 1 < n) sharded_begin(c, h, pps[k + 1], queries[k + 1], query_lens[k + 1], shard, nxt, false);
      size_t len = 0;
      sharded_finish(c, cur, c->rank == 0 ? out + (size_t)k * out_stride : nullptr, out_stride, &len, k == n - 1);
      if (c->rank == 0) *out_len = len;
      sp_query_free(cur.q);
      cur = nxt;
      nxt = ShardedRun{};
    }
    cur = ShardedRun{};
    (void)hipEventElapsedTime(&c->ms[0], c->ev_t[0], c->ev_t[1]);
    (void)hipEventElapsedTime(&c->ms[1], c->ev_t[1], c->ev_t[2]);
    if (hipEventElapsedTime(&c->ms[2], c->ev_t[1], c->ev_rs) != hipSuccess || c->ms[2] < 0) c->ms[2] = 0;
    note_transport(c);
  });
  for (ShardedRun* r : {&cur, &nxt})
    if (r->q) {
      (void)hipStreamSynchronize(c->stream);
      (void)sp_query_sync(r->q);
      sp_query_free(r->q);
    }
  return rc;
}

int sp_comm_barrier(sp_comm_t* c) {
  if (!c) return SP_E_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  return guarded_comm([&] {
    c->ensure(c->gathered, c->gathered_bytes, (size_t)c->world * sizeof(uint64_t));
    c->ensure(c->mine, c->mine_bytes, sizeof(uint64_t));
    hip_ok(hipMemsetAsync(c->mine, 0, sizeof(uint64_t), c->stream), "hipMemsetAsync");
    if (c->all_gather(c->mine, c->gathered, 1) != 0) throw Fail{SP_E_HIP, "custom all_gather_u64 failed"};
    hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  });
}

}  // extern "C"
