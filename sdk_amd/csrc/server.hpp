// Host-side orchestration of the Spiral answer path on one MI355X.
// Mirrors the spiral-rs server surface (lib/spiral-rs/src/server.rs): PublicParameters / Query
// deserialisation (client.rs:212-259, 303-329), expand_query, multiply_reg_by_database,
// fold_ciphertexts, pack, encode, process_query -- with all polynomial data device-resident.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.hpp"
#include "params.hpp"

namespace spiral {

struct HipError : std::runtime_error {
  explicit HipError(const std::string& s) : std::runtime_error(s) {}
};
struct ArgError : std::runtime_error {
  explicit ArgError(const std::string& s) : std::runtime_error(s) {}
};
struct OomError : std::runtime_error {
  explicit OomError(const std::string& s) : std::runtime_error(s) {}
};

void hip_check(hipError_t e, const char* what, const char* file, int line);
#define HIP_CHECK(x) ::spiral::hip_check((x), #x, __FILE__, __LINE__)
u64 paths_taken(bool reset);
void set_tunable(const char* name, long v);
void tunables_new_call();  // a public entry point begins: cached switch values are re-resolved (server.cpp)  // thread-local PathBit mask accumulated by launched() / note_path()
// Debug hook called on every fresh DevBuf allocation (server.cpp): SPIRAL_POISON_WS / sp_debug_set("poison_ws", b)
// fills the buffer with byte b (1..255) so that a read of never-written device memory shows up deterministically
// instead of depending on what the pages held before; "poison_skip" = k leaves the k-th allocation since the
// switch was set zero-filled instead (bisection of the offending buffer).  Off (0) by default: no cost.
void devbuf_fresh(void* p, size_t bytes);
// Allocation behind DevBuf (server.cpp).  guard_ws = g > 0 (debug): every buffer gets g bytes of guard region on either
// side, filled with the poison byte (zero for allocation number poison_skip): an out-of-bounds READ then returns the
// same garbage in every run instead of whatever the neighbouring allocation holds, and an out-of-bounds WRITE is
// reported when the buffer is released.  Throws OomError.
void* devbuf_alloc(size_t bytes, size_t* guard, long* serial);
void devbuf_free(void* p, size_t bytes, size_t guard, long serial);
bool devbuf_guards_on();
void devbuf_cache_sync();

// RAII device buffer
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t guard = 0;  // bytes of guard region on either side of the buffer (debug: guard_ws)
  long serial = -1;  // allocation number (debug)
  DevBuf() {}
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), guard(o.guard), serial(o.serial) { o.p = nullptr; o.n = 0; o.guard = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; guard = o.guard; serial = o.serial; o.p = nullptr; o.n = 0; o.guard = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    if (count == 0) return;
    void* q = devbuf_alloc(count * sizeof(T), &guard, &serial);
    p = (T*)q;
    n = count;
  }
  // Large streaming buffers (the database).  want_contiguous (switch db_contiguous, OFF by default) asks for PHYSICALLY
  // CONTIGUOUS device memory first: how a plain hipMalloc of tens of GiB is backed is a lottery (the same read stream runs
  // at 6.7-7.07 TB/s depending on the process, a contiguous allocation gave 7.05-7.08 TB/s in every run:
  // scripts/ubench/hbm_map.hip, profiles/r02_sweep_experiments.md) -- but on this stack memory of a FREED contiguous
  // allocation is zeroed after its next owner has already written to it (profiles/r03_contiguous_alloc.md).
  void alloc_streaming(size_t count, bool want_contiguous) {
    release();
    if (count == 0) return;
    if (want_contiguous && !devbuf_guards_on()) {
      void* q = nullptr;
      const hipError_t e = hipExtMallocWithFlags(&q, count * sizeof(T), hipDeviceMallocContiguous);  // see db_create_impl: unsafe
      if (getenv("SPIRAL_ALLOC_DEBUG")) fprintf(stderr, "[spiral] contiguous allocation of %zu bytes: %s -> [%p, %p)\n", count * sizeof(T), hipGetErrorName(e), q, (char*)q + count * sizeof(T));
      if (e == hipSuccess && q) {
        p = (T*)q;
        n = count;
        devbuf_cache_sync();
        (void)hipDeviceSynchronize();
        return;
      }
      (void)hipGetLastError();
    }
    alloc(count);
  }
  void ensure(size_t count) { if (count > n) alloc(count); }
  void release() { if (p) { devbuf_free(p, n * sizeof(T), guard, serial); p = nullptr; n = 0; guard = 0; serial = -1; } }
  size_t bytes() const { return n * sizeof(T); }
};

// Synchronous host -> device upload of resident data (tables, index lists), device drained afterwards.
void upload_words(void* dst, const void* host, size_t bytes);
// Synchronous host -> device copy; h2d_cache_sync = 1 (diagnostic, default 0) adds k_cache_sync on the null stream:
// every XCD writes back and invalidates its L2 (profiles/r02_stale_reads.md).
void h2d_sync(void* dst, const void* host, size_t bytes);

// ChaCha20 keystream as rand_chacha 0.3.1's ChaCha20Rng::from_seed produces it (key = seed, 64-bit
// block counter from 0, stream 0); u64 = two consecutive u32 words, low first (client.rs:47-49).
void chacha20_keystream_u64(const uint8_t seed[32], u64* out, size_t count);

// Per-(Params, device) constants: twiddles, -x^(N-2^r) polys (params.rs:98-107), NTT of the GSW
// gadget (server.rs:509), expansion schedules and the index lists the batched kernels consume.
struct RoundPlan {
  int num_in = 0, t_auto = 0;
  int n_all = 0, n_left = 0, n_right = 0;
  // offsets (in ints) into DeviceState::lists
  size_t all_ct = 0;      // global ct index of each active ct
  size_t all_row1 = 0;    // poly index (ct*2+1) of each active ct
  size_t left_pos = 0;    // position within the active list of each left-group ct
  size_t left_out = 0;    // poly index (ct*2) of each left-group ct
  size_t right_pos = 0, right_out = 0;
  size_t left_ct = 0, right_ct = 0;  // ciphertext index of each left / right group member (k_expand_round)
  size_t skip2 = 0;       // second-half cts that are pruned (only scalar-multiplied), server.rs:40-47
  int n_skip2 = 0;
};

struct DeviceState {
  int device = -1;
  DevTables T;
  DevBuf<u32> tw;
  DevBuf<u32> neg1;        // [g][1 poly]
  DevBuf<u32> gadget_gsw;  // [2][2 t_gsw] polys
  DevBuf<int> lists;
  std::vector<RoundPlan> rounds;
  // the same schedule split along the expansion tree: after round 0 the even-indexed ciphertexts (first-dimension
  // query, server.rs:566-568) and the odd-indexed ones (GSW bits, server.rs:569-571) never meet again, so the two
  // subtrees can run on different streams.  rounds_even[0] / rounds_odd[0] are empty / round 0 itself.
  std::vector<RoundPlan> rounds_even, rounds_odd;
  size_t max_all = 0, max_left = 0, max_right = 0;
  // Expansion schedules pruned to the first-dimension rows [j0, j0 + nj) a row shard needs (the even "left" subtree
  // only computes the ancestors of those leaves; the GSW side is always complete).  Built on first use.
  struct PrunedPlan {
    int j0 = 0, nj = 0;
    std::vector<RoundPlan> rounds;  // offsets into `lists` below
    std::vector<RoundPlan> rounds_even, rounds_odd;  // as in DeviceState
    DevBuf<int> lists;
  };
  std::vector<std::unique_ptr<PrunedPlan>> pruned;
  std::mutex pruned_mu;
  const PrunedPlan& pruned_plan(const Params& P, int j0, int nj);
  // regev_to_gsw lists (batch b = d*t_gsw + j)
  size_t gsw_src_ct = 0;     // ct index 2b+1 (or b when nu_2 == 0: unused)
  size_t gsw_src_poly = 0;   // poly index (2b+1)*2
  size_t gsw_out_even = 0;   // fold_mats poly index of column 2j (row 0)
  size_t gsw_out_odd = 0;    // column 2j+1
  // pack lists (batch b = (inst*n + c)*n + r)
  size_t pack_src_ct = 0;    // plane index inst*n*n + r*n + c
  size_t pack_out = 0;       // result poly index inst*(n+1)*n + c           (per (inst,c))
  size_t pack_row = 0;       // result poly index inst*(n+1)*n + (1+r)*n + c (per b)
  // packing version 1 (lib/server/src/compute/pack.rs:46-99), n == 2: b = (inst*2 + c)*2 + r; e = inst*2 + c
  size_t v1_row1 = 0;        // [b]  poly index b*3 + 1 in the prod buffer
  size_t v1_shift_src = 0;   // [e]  poly index of prod[b(r=1)][0]
  size_t v1_rot_dst = 0, v1_rot_a = 0, v1_rot_b = 0;  // [2e] P2[e][1] += P[b1][2], P2[e][2] += P[b1][1]
  size_t v1_sum_dst = 0, v1_sum_a = 0, v1_sum_b = 0;  // [3e] result[inst][rr][c] = P[b0][rr] + P2[e][rr]
};

struct Workspace;

}  // namespace spiral

// ---- the opaque C-ABI handle types --------------------------------------------------------------
struct sp_params {
  spiral::Params p;
  std::mutex mu;
  std::vector<std::unique_ptr<spiral::DeviceState>> dev;        // one per device used
  std::vector<std::unique_ptr<spiral::Workspace>> ws_pool;      // idle workspaces
  spiral::DeviceState& device_state();                          // for the current device
  std::unique_ptr<spiral::Workspace> acquire_ws();
  void release_ws(std::unique_ptr<spiral::Workspace> ws);
  ~sp_params();
};

struct sp_pp {
  const sp_params* params = nullptr;
  int device = -1;
  // NTT-form matrices, device resident
  spiral::DevBuf<spiral::u32> all;  // wire order: v_packing[n], v_expansion_left[g], [v_expansion_right], v_conversion
  size_t n_polys = 0;
  size_t off_packing = 0, off_left = 0, off_right = 0, off_conv = 0;  // poly offsets
  bool has_right = false;
  spiral::DevBuf<spiral::u32> pack_cat;  // [(n+1)][n*t_conv] = [W_0 | W_1 | ...]
  // r06: `all` once more in WAVE layout (k_mats_to_wave, same polynomial order) for k_expand_wave, followed by one more slot that
  // holds the constant polynomials 0 and 1 (N words each); empty when the parameters have no query expansion
  spiral::DevBuf<spiral::u32> all_w;
};

struct sp_db {
  const sp_params* params = nullptr;
  int device = -1;
  int shard = 0, num_shards = 1;
  int j0 = 0, nj = 0;
  int packed = 0;                     // 7-byte PACKED device format (kernels.hpp) vs 8-byte words
  int col_g = 0, col_G = 1;           // column shard: holds columns ii = col_g (mod col_G)
  int np_local = 0;                   // columns held = num_per / col_G
  spiral::ColMap colmap() const { return spiral::ColMap{col_g, col_G, np_local * col_G}; }
  spiral::DevBuf<spiral::u64> words;  // [plane][z][j_local][ii] (or the PACKED unit stream)
  // digit-planar copy of a PACKED database for the 9 .. 16-query pass (sweep_planar.hpp): built from `words` by
  // sp_db_prepare_batch or by the first such group when the device has the room; bulk writers of `words` invalidate it AFTER their
  // write, under `mu` (a build between an early invalidation and the end of the write would stay valid and stale),
  // sp_db_update_item patches its 8 entries per (plane, z) in place.  planar_state: 0 = not built, 1 = valid, -1 = the shape has
  // no planar form (never tried again), -2 = no room when last tried (tried again by the next group / prepare call).  The memory is
  // released when the switch `batch_planar` is off, when the copy is invalidated, and when a batched call runs out of memory.
  spiral::DevBuf<spiral::u64> planar;
  int planar_state = 0;
  const unsigned char* ensure_planar(hipStream_t s);   // capi.cpp; nullptr: use the PACKED kernels
  void drop_planar() {                                  // caller holds mu
    if (planar_state != -1) planar_state = 0;
    planar.release();
  }
  std::mutex mu;
  // ---- sparse bucket (sp_db_create_sparse; lib/server/src/db/sparse_db.rs:5-48): only present items are stored
  bool sparse = false;
  std::unordered_map<size_t, size_t> slot_of;  // item index -> slot (db_idx_to_vec_idx)
  spiral::DevBuf<spiral::u64> polys;           // [slot][plane][N] packed NTT words
  size_t slots_cap = 0;
  bool index_dirty = true;                     // a NEW key was added: the index below is rebuilt before the next query
  bool rows_dirty = true;                      // ... and the set of occupied rows changed: the expansion plan too
  // Immutable snapshot of the index: present items by column (CSR: row j and slot) + the expansion schedule pruned to the
  // rows that hold items.  A query takes a shared_ptr when it begins and keeps it until it is freed, so an update that
  // publishes a new snapshot never frees device lists a query in flight still reads.  (Overwriting an existing item
  // changes neither; the polynomial store itself follows the reference's rule -- writers are exclusive,
  // lib/server/src/bin/server.rs:24 RwLock.)
  struct SparseIndex {
    spiral::DevBuf<int> col_ptr, col_rows, col_slots;
    std::shared_ptr<spiral::DeviceState::PrunedPlan> plan;
  };
  std::shared_ptr<const SparseIndex> sparse_index;
  spiral::DevBuf<uint8_t> staging;             // one item's bytes (sp_db_update_item), reused
  spiral::DevBuf<spiral::u64> load_stage;              // sp_db_load_plane's upload window, kept for the handle's life (switch db_stage_keep; capi.cpp)
  std::shared_ptr<const SparseIndex> ensure_sparse_index();  // capi.cpp
};
