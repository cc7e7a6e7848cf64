// Host-side orchestration of the Spiral answer path on one MI355X (see server.hpp).
// Reference call tree reproduced: lib/spiral-rs/src/server.rs:650-741 (process_query) ->
// :525-591 expand_query -> :19-121 coefficient_expansion, :123-151 regev_to_gsw; :505-523
// get_v_folding_neg; :155-221 multiply_reg_by_database; :388-427 fold_ciphertexts; :429-468 pack;
// :470-503 encode.  Deserialisers: client.rs:212-259, 303-329.
#include "server.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "pipeline.hpp"

namespace spiral {

void hip_check(hipError_t e, const char* what, const char* file, int line) {
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw HipError(std::string(what) + " failed at " + file + ":" + std::to_string(line) + ": " + hipGetErrorString(e));
  }
}

// ---- tunables: sp_debug_set(name, v) overrides, else SPIRAL_<NAME> from the environment, else the default
// Looked up hundreds of times per query (every launch asks for debug_sync, every stage for its switches), so the answer
// is cached per calling thread and per call site and only re-resolved -- override list under the mutex, then getenv -- when
// the EPOCH has moved: sp_debug_set and every public entry point that starts a piece of work (guarded() in capi.cpp)
// bump it.  A test that changes the environment between two calls is therefore still seen; inside a call the launch
// path touches neither the mutex nor getenv.
static std::mutex g_tun_mu;
static std::vector<std::pair<std::string, long>> g_tun;
static std::atomic<unsigned> g_tun_epoch{1};
static std::atomic<long> g_fresh_allocs{0};  // allocations seen by devbuf_fresh since poison_skip was last set
void tunables_new_call() { g_tun_epoch.fetch_add(1, std::memory_order_relaxed); }
void set_tunable(const char* name, long v) {
  if (!strcmp(name, "poison_skip")) g_fresh_allocs = 0;
  {
    std::lock_guard<std::mutex> lk(g_tun_mu);
    bool found = false;
    for (auto& kv : g_tun)
      if (kv.first == name) {
        kv.second = v;
        found = true;
        break;
      }
    if (!found) g_tun.emplace_back(name, v);
  }
  tunables_new_call();
}
static bool tunable_resolve(const char* name, long* out) {
  {
    std::lock_guard<std::mutex> lk(g_tun_mu);
    for (auto& kv : g_tun)
      if (kv.first == name) {
        *out = kv.second;
        return true;
      }
  }
  std::string env = "SPIRAL_";
  for (const char* c = name; *c; c++) env += (char)toupper((unsigned char)*c);
  const char* e = getenv(env.c_str());
  if (!e) return false;
  *out = atol(e);
  return true;
}
long tunable(const char* name, long dflt) {
  struct Ent {
    std::string name;  // compared by CONTENT: call sites pass literals today, but an address is no identity for a heap or
                       // stack string whose storage is reused
    unsigned epoch;
    bool has;
    long val;
  };
  thread_local std::vector<Ent> cache;
  const unsigned ep = g_tun_epoch.load(std::memory_order_relaxed);
  for (auto& e : cache)
    if (!strcmp(e.name.c_str(), name)) {
      if (e.epoch != ep) {
        e.has = tunable_resolve(name, &e.val);
        e.epoch = ep;
      }
      return e.has ? e.val : dflt;
    }
  Ent e{name, ep, false, 0};
  e.has = tunable_resolve(name, &e.val);
  cache.push_back(e);
  return e.has ? e.val : dflt;
}

void devbuf_fresh(void* p, size_t bytes) {
  const long b = tunable("poison_ws", 0);
  if (b <= 0 || !p || tunable("guard_ws", 0) > 0) return;
  const long k = g_fresh_allocs.fetch_add(1);
  const bool skip = tunable("poison_skip", -1) == k;
  if (getenv("SPIRAL_ALLOC_DEBUG")) fprintf(stderr, "[spiral] alloc #%ld: %zu bytes%s\n", k, bytes, skip ? " (zeroed)" : "");
  HIP_CHECK(hipMemset(p, skip ? 0 : (int)(b & 255), bytes));
  HIP_CHECK(hipDeviceSynchronize());
}
void devbuf_cache_sync();
bool devbuf_guards_on() { return tunable("guard_ws", 0) > 0; }
void* devbuf_alloc(size_t bytes, size_t* guard, long* serial) {
  const long g = tunable("guard_ws", 0);
  const size_t gb = g > 0 ? ((size_t)g + 255) / 256 * 256 : 0;
  void* q = nullptr;
  const hipError_t e = hipMalloc(&q, bytes + 2 * gb);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw OomError(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
  }
  *guard = gb;
  *serial = -1;
  if (getenv("SPIRAL_ALLOC_DEBUG")) fprintf(stderr, "[spiral] hipMalloc %zu bytes -> [%p, %p)\n", bytes + 2 * gb, q, (char*)q + bytes + 2 * gb);
  void* const block = q;  // what hipMalloc returned (q moves past the front guard below)
  try {
  if (gb) {
    const long k = g_fresh_allocs.fetch_add(1);
    *serial = k;
    const bool skip = tunable("poison_skip", -1) == k;
    const int fill = skip ? 0 : (int)(tunable("poison_ws", 0xA5) & 255);
    if (getenv("SPIRAL_ALLOC_DEBUG")) fprintf(stderr, "[spiral] alloc #%ld: %zu bytes%s\n", k, bytes, skip ? " (guards zeroed)" : "");
    HIP_CHECK(hipMemset(q, fill, gb));
    HIP_CHECK(hipMemset((char*)q + gb + bytes, fill, gb));
    HIP_CHECK(hipDeviceSynchronize());
    q = (char*)q + gb;
  }
  devbuf_fresh(q, bytes);
  // Fresh device memory is zero-filled and the device drained before the buffer is handed out (alloc_zero, default 1):
  // no kernel of the library ever reads a word it did not write, but recycled pages hold whatever the previous owner
  // (this process or an earlier tenant of the GPU) left there, and a defect of that kind would otherwise only show on
  // some machines.  Allocation is off the query path (workspaces are allocated whole when they are created).
  if (!gb && tunable("poison_ws", 0) <= 0 && tunable("alloc_zero", 1) != 0) {
    HIP_CHECK(hipMemset(q, 0, bytes));
    devbuf_cache_sync();
    HIP_CHECK(hipDeviceSynchronize());
  }
  } catch (...) {  // a failed fill must not leak the block
    (void)hipFree(block);
    throw;
  }
  return q;
}
// alloc_cache_sync (diagnostic, default 0): a fresh allocation ends with k_cache_sync -- every XCD writes back and
// invalidates its L2 (used while hunting the corruption that turned out to come from contiguous allocations,
// profiles/r02_stale_reads.md).
void devbuf_cache_sync() {
  if (tunable("alloc_cache_sync", 0) != 0) launch_cache_sync(nullptr, 0);
}
void devbuf_free(void* p, size_t bytes, size_t guard, long serial) {
  if (!p) return;
  if (getenv("SPIRAL_ALLOC_DEBUG")) fprintf(stderr, "[spiral] hipFree %p\n", (void*)((char*)p - guard));
  if (guard) {
    // every guard byte still equals the byte farthest from the buffer (the fill value) unless something wrote there
    (void)hipDeviceSynchronize();
    std::vector<unsigned char> h(2 * guard);
    if (hipMemcpy(h.data(), (char*)p - guard, guard, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(h.data() + guard, (char*)p + bytes, guard, hipMemcpyDeviceToHost) == hipSuccess) {
      long first_before = -1, last_after = -1;  // distance from the buffer, in bytes
      for (size_t i = 0; i < guard; i++)
        if (h[i] != h[0]) { first_before = (long)(guard - i); break; }
      for (size_t i = 0; i < guard; i++)
        if (h[guard + i] != h[2 * guard - 1]) last_after = (long)i;
      if (first_before >= 0 || last_after >= 0)
        fprintf(stderr, "[spiral] OUT-OF-BOUNDS WRITE around alloc #%ld (%zu bytes): reaches %ld bytes before / %ld bytes after\n",
                serial, bytes, first_before, last_after);
    }
    (void)hipGetLastError();
  }
  (void)hipFree((char*)p - guard);
}

extern thread_local int g_debug_stage;
void h2d_sync(void* dst, const void* host, size_t bytes) {
  if (bytes == 0) return;
  HIP_CHECK(hipMemcpy(dst, host, bytes, hipMemcpyHostToDevice));
  if (tunable("h2d_cache_sync", 0) != 0) launch_cache_sync(nullptr, 0);
}
void upload_words(void* dst, const void* host, size_t bytes) {
  h2d_sync(dst, host, bytes);
  HIP_CHECK(hipDeviceSynchronize());
}

static thread_local u64 g_paths = 0;
void note_path(u64 bits) { g_paths |= bits; }
u64 paths_taken(bool reset) {
  const u64 v = g_paths;
  if (reset) g_paths = 0;
  return v;
}
void launched(u64 bits, const char* kernel) {
  g_paths |= bits;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw HipError(std::string("launch of ") + kernel + " failed: " + hipGetErrorString(e));
  // debug_sync: 1 = device-wide synchronisation after every launch (tells a race from a logic error); 11 / 12 / 13 =
  // only after the launches of the expansion / sweep / fold-pack-encode stage; 2 = only at the stage boundaries (capi.cpp)
  const long ds = tunable("debug_sync", 0);
  if (ds == 1 || (ds > 10 && ds == 10 + g_debug_stage)) HIP_CHECK(hipDeviceSynchronize());
}
thread_local int g_debug_stage = 0;
void debug_stage(int stage) {
  g_debug_stage = stage;
  if (tunable("debug_sync", 0) == 2) HIP_CHECK(hipDeviceSynchronize());
}

// ---------------------------------------------------------------------------------- ChaCha20
// One block at a time (any x86-64), and eight blocks side by side with AVX2 where the CPU has it (one __m256i per state word,
// counters ctr0 .. ctr0 + 7): a query's 16 KiB of keystream takes 28 us of host time the first way, 8 the second -- and a
// 16-query call pays it sixteen times before its first kernel can start (r06).  Same keystream either way.
static inline u32 rotl(u32 v, int c) { return (v << c) | (v >> (32 - c)); }
static inline void quarter(u32* s, int a, int b, int c, int d) {
  s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
  s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
  s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
  s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
}
static void chacha_block(const u32 init[16], u64 ctr, u32 out[16]) {
  u32 in0[16], s[16];
  memcpy(in0, init, sizeof(in0));
  in0[12] = (u32)ctr;
  in0[13] = (u32)(ctr >> 32);
  in0[14] = in0[15] = 0;
  memcpy(s, in0, sizeof(s));
  for (int rnd = 0; rnd < 10; rnd++) {
    quarter(s, 0, 4, 8, 12); quarter(s, 1, 5, 9, 13); quarter(s, 2, 6, 10, 14); quarter(s, 3, 7, 11, 15);
    quarter(s, 0, 5, 10, 15); quarter(s, 1, 6, 11, 12); quarter(s, 2, 7, 8, 13); quarter(s, 3, 4, 9, 14);
  }
  for (int i = 0; i < 16; i++) out[i] = s[i] + in0[i];
}
#if defined(__x86_64__)
}  // namespace spiral
#include <immintrin.h>
namespace spiral {
__attribute__((target("avx2"))) static void chacha_blocks8_avx2(const u32 init[16], u64 ctr0, u32 (*out)[16]) {
  __m256i in0[16], s[16];
  for (int i = 0; i < 16; i++) in0[i] = _mm256_set1_epi32((int)init[i]);
  alignas(32) u32 lo[8], hi[8];
  for (int l = 0; l < 8; l++) {
    const u64 c = ctr0 + (u64)l;
    lo[l] = (u32)c;
    hi[l] = (u32)(c >> 32);
  }
  in0[12] = _mm256_load_si256(reinterpret_cast<const __m256i*>(lo));
  in0[13] = _mm256_load_si256(reinterpret_cast<const __m256i*>(hi));
  in0[14] = in0[15] = _mm256_setzero_si256();
  for (int i = 0; i < 16; i++) s[i] = in0[i];
#define SP_ROT(x, n) _mm256_or_si256(_mm256_slli_epi32(x, n), _mm256_srli_epi32(x, 32 - n))
#define SP_QR(a, b, c, d)                                                                              \
  s[a] = _mm256_add_epi32(s[a], s[b]); s[d] = _mm256_xor_si256(s[d], s[a]); s[d] = SP_ROT(s[d], 16);   \
  s[c] = _mm256_add_epi32(s[c], s[d]); s[b] = _mm256_xor_si256(s[b], s[c]); s[b] = SP_ROT(s[b], 12);   \
  s[a] = _mm256_add_epi32(s[a], s[b]); s[d] = _mm256_xor_si256(s[d], s[a]); s[d] = SP_ROT(s[d], 8);    \
  s[c] = _mm256_add_epi32(s[c], s[d]); s[b] = _mm256_xor_si256(s[b], s[c]); s[b] = SP_ROT(s[b], 7);
  for (int rnd = 0; rnd < 10; rnd++) {
    SP_QR(0, 4, 8, 12) SP_QR(1, 5, 9, 13) SP_QR(2, 6, 10, 14) SP_QR(3, 7, 11, 15)
    SP_QR(0, 5, 10, 15) SP_QR(1, 6, 11, 12) SP_QR(2, 7, 8, 13) SP_QR(3, 4, 9, 14)
  }
#undef SP_QR
#undef SP_ROT
  alignas(32) u32 tmp[16][8];
  for (int i = 0; i < 16; i++) _mm256_store_si256(reinterpret_cast<__m256i*>(tmp[i]), _mm256_add_epi32(s[i], in0[i]));
  for (int l = 0; l < 8; l++)
    for (int i = 0; i < 16; i++) out[l][i] = tmp[i][l];
}
static bool cpu_has_avx2() { return __builtin_cpu_supports("avx2"); }
#else
static void chacha_blocks8_avx2(const u32*, u64, u32 (*)[16]) {}
static bool cpu_has_avx2() { return false; }
#endif

void chacha20_keystream_u64(const uint8_t seed[32], u64* out, size_t count) {
  u32 init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 8; i++) {
    u32 w;
    memcpy(&w, seed + 4 * i, 4);
    init[4 + i] = w;
  }
  static const bool wide = cpu_has_avx2() && tunable("chacha_scalar", 0) == 0;
  u64 ctr = 0;
  size_t done = 0;
  u32 blk[8][16];
  while (done < count) {
    int have = 1;
    if (wide && count - done >= 16) {   // (a short tail -- or a short request -- goes block by block)
      chacha_blocks8_avx2(init, ctr, blk);
      have = 8;
    } else {
      chacha_block(init, ctr, blk[0]);
    }
    ctr += (u64)have;
    for (int l = 0; l < have && done < count; l++)
      for (int i = 0; i < 16 && done < count; i += 2, done++) out[done] = (u64)blk[l][i] | ((u64)blk[l][i + 1] << 32);
  }
}

// x mod Q for the 56-bit Q of the parameter sets (Q < 2^63): the quotient estimate mulhi(x, floor(2^64 / Q)) is at most 2 short
struct FastMod {
  u64 Q, m;
  explicit FastMod(u64 q) : Q(q), m(q > 1 ? (u64)(((unsigned __int128)1 << 64) / q) : 0) {}
  u64 operator()(u64 x) const {
    u64 r = x - (u64)(((unsigned __int128)x * m) >> 64) * Q;
    while (r >= Q) r -= Q;
    return r;
  }
};

// ---------------------------------------------------------------------------------- device state
// Expansion schedule (server.rs:19-121) restricted to the first-dimension rows [j0, j0 + nj): output ct c of round
// r (index < 2^(r+1)) is an ancestor of leaf L iff L = c (mod 2^(r+1)); leaf 2j is row j (server.rs:566-568), the odd
// leaves are the GSW selector bits and are always kept.  j0 = 0, nj = dim0 gives the reference's own schedule.
// side: 0 = the whole schedule, 1 = only the even subtree (rounds >= 1, even i), 2 = round 0 and the odd subtree
static std::vector<RoundPlan> build_round_plans(const Params& P, int j0, int nj, std::vector<int>& L,
                                                const std::vector<char>* row_set = nullptr, int side = 0) {
  auto put = [&](const std::vector<int>& v) {
    size_t off = L.size();
    L.insert(L.end(), v.begin(), v.end());
    return off;
  };
  std::vector<RoundPlan> rounds;
  const size_t g = P.g();
  const size_t nu2 = P.db_dim_2;
  const size_t stop_round = nu2 > 0 ? P.stop_round() : 0;
  const size_t max_bits_right = nu2 > 0 ? P.t_gsw * nu2 : 0;
  // row_set: an arbitrary set of rows (sparse buckets; lib/server/src/compute/query_expansion.rs:213-248
  // to_per_round_set keeps exactly the ancestors of the even leaves 2j, j in the set, and of every odd leaf)
  const bool prune = nu2 > 0 && (row_set != nullptr || j0 != 0 || nj != (int)P.dim0());
  for (size_t r = 0; r < g; r++) {
    RoundPlan rp;
    rp.num_in = 1 << r;
    rp.t_auto = (int)(POLY_LEN >> r) + 1;
    // even cts needed by the row range: c = 2j (mod 2^(r+1)) for some j in [j0, j0 + nj)
    std::vector<char> need_even((size_t)2 << r, 1);
    if (prune) {
      std::fill(need_even.begin(), need_even.end(), 0);
      const size_t mod = (size_t)2 << r;
      if (row_set) {
        for (size_t j = 0; j < row_set->size(); j++)
          if ((*row_set)[j]) need_even[((size_t)2 * j) % mod] = 1;
      } else if ((size_t)nj * 2 >= mod) {
        std::fill(need_even.begin(), need_even.end(), 1);
      } else {
        for (int j = j0; j < j0 + nj; j++) need_even[((size_t)2 * j) % mod] = 1;
      }
    }
    std::vector<int> all_ct, all_row1, lpos, lout, rpos, rout, skip2, lct, rct;
    for (int half = 0; half < 2; half++)
      for (int i = 0; i < rp.num_in; i++) {  // both halves enumerate from 0 (server.rs:112-119)
        bool skip = (stop_round > 0 && r > stop_round && (i % 2) == 1) ||
                    (stop_round > 0 && r == stop_round && (i % 2) == 1 && (size_t)(i / 2) >= max_bits_right);
        const int ct = half * rp.num_in + i;
        const bool even_tree = r != 0 && (i % 2) == 0;
        if ((side == 1 && !even_tree) || (side == 2 && even_tree)) continue;
        if (skip) {
          if (half == 1 && !prune) skip2.push_back(rp.num_in + i);
          continue;
        }
        if (r != 0 && (i % 2) == 0 && !need_even[ct]) continue;
        int pos = (int)all_ct.size();
        all_ct.push_back(ct);
        all_row1.push_back(ct * 2 + 1);
        if (r != 0 && (i % 2) == 0) {
          lpos.push_back(pos);
          lout.push_back(ct * 2);
          lct.push_back(ct);
        } else {
          rpos.push_back(pos);
          rout.push_back(ct * 2);
          rct.push_back(ct);
        }
      }
    rp.n_all = (int)all_ct.size();
    rp.n_left = (int)lpos.size();
    rp.n_right = (int)rpos.size();
    rp.all_ct = put(all_ct);
    rp.all_row1 = put(all_row1);
    rp.left_pos = put(lpos);
    rp.left_out = put(lout);
    rp.right_pos = put(rpos);
    rp.right_out = put(rout);
    rp.left_ct = put(lct);
    rp.right_ct = put(rct);
    rp.skip2 = put(skip2);
    rp.n_skip2 = (int)skip2.size();
    rounds.push_back(rp);
  }
  return rounds;
}

const DeviceState::PrunedPlan& DeviceState::pruned_plan(const Params& P, int j0, int nj) {
  std::lock_guard<std::mutex> lk(pruned_mu);
  for (auto& pp : pruned)
    if (pp->j0 == j0 && pp->nj == nj) return *pp;
  auto pl = std::make_unique<PrunedPlan>();
  pl->j0 = j0;
  pl->nj = nj;
  std::vector<int> L;
  pl->rounds = build_round_plans(P, j0, nj, L);
  pl->rounds_even = build_round_plans(P, j0, nj, L, nullptr, 1);
  pl->rounds_odd = build_round_plans(P, j0, nj, L, nullptr, 2);
  pl->lists.alloc(std::max<size_t>(L.size(), 1));
  if (!L.empty()) upload_words(pl->lists.p, L.data(), L.size() * sizeof(int));
  pruned.push_back(std::move(pl));
  return *pruned.back();
}

std::unique_ptr<DeviceState::PrunedPlan> build_pruned_plan_rows(const Params& P, const std::vector<char>& rows) {
  auto pl = std::make_unique<DeviceState::PrunedPlan>();
  pl->j0 = -1;
  pl->nj = -1;
  std::vector<int> L;
  pl->rounds = build_round_plans(P, 0, 0, L, &rows);
  pl->rounds_even = build_round_plans(P, 0, 0, L, &rows, 1);
  pl->rounds_odd = build_round_plans(P, 0, 0, L, &rows, 2);
  pl->lists.alloc(std::max<size_t>(L.size(), 1));
  if (!L.empty()) upload_words(pl->lists.p, L.data(), L.size() * sizeof(int));
  return pl;
}

static std::unique_ptr<DeviceState> build_device_state(const Params& P, int device) {
  auto D = std::make_unique<DeviceState>();
  D->device = device;
  D->tw.alloc(P.ntt_tables.size());
  upload_words(D->tw.p, P.ntt_tables.data(), P.ntt_tables.size() * sizeof(u32));
  D->T.tw = D->tw.p;
  D->T.c = P.dc;
  const size_t g = P.expand_queries ? P.g() : 0;
  const size_t nu2 = P.db_dim_2;

  // -x^(N - 2^r) in NTT form (params.rs:98-107): coefficient Q-1 at idx; the other coefficients are
  // Q in the reference (invert_poly of 0), i.e. 0 after reduction.
  {
    std::vector<u64> raw(std::max<size_t>(g, 1) * POLY_LEN, 0);
    for (size_t r = 0; r < g && r < POLY_LEN_LOG2; r++) raw[r * POLY_LEN + (POLY_LEN - ((size_t)1 << r))] = P.modulus - 1;
    DevBuf<u64> d_raw(raw.size());
    h2d_sync(d_raw.p, raw.data(), raw.size() * 8);
    D->neg1.alloc(std::max<size_t>(g, 1) * 2 * POLY_LEN);
    FwdDesc f{d_raw.p, nullptr, D->neg1.p, (int)g, 1, 1, 1, 64, 1, 0, 1};
    launch_ntt_fwd(D->T, f, 0);
    HIP_CHECK(hipDeviceSynchronize());
  }
  // build_gadget(params, 2, 2*t_gsw).ntt()  (gadget.rs:11-32, server.rs:509)
  {
    const size_t cols = 2 * P.t_gsw, bits = P.bits_per(P.t_gsw);
    std::vector<u64> raw(2 * cols * POLY_LEN, 0);
    for (size_t i = 0; i < 2; i++)
      for (size_t j = 0; j < P.t_gsw; j++) {
        if (bits * j >= 64) continue;
        raw[(i * cols + (i + j * 2)) * POLY_LEN] = 1ULL << (bits * j);
      }
    DevBuf<u64> d_raw(raw.size());
    h2d_sync(d_raw.p, raw.data(), raw.size() * 8);
    D->gadget_gsw.alloc(2 * cols * 2 * POLY_LEN);
    FwdDesc f{d_raw.p, nullptr, D->gadget_gsw.p, (int)(2 * cols), 1, 1, 1, 64, 1, 0, 1};
    launch_ntt_fwd(D->T, f, 0);
    HIP_CHECK(hipDeviceSynchronize());
  }

  // index lists
  std::vector<int> L;
  auto put = [&](const std::vector<int>& v) {
    size_t off = L.size();
    L.insert(L.end(), v.begin(), v.end());
    return off;
  };
  if (P.expand_queries) {
    D->rounds = build_round_plans(P, 0, (int)P.dim0(), L);
    D->rounds_even = build_round_plans(P, 0, (int)P.dim0(), L, nullptr, 1);
    D->rounds_odd = build_round_plans(P, 0, (int)P.dim0(), L, nullptr, 2);
    for (const RoundPlan& rp : D->rounds) {
      D->max_all = std::max(D->max_all, (size_t)rp.n_all);
      D->max_left = std::max(D->max_left, (size_t)rp.n_left);
      D->max_right = std::max(D->max_right, (size_t)rp.n_right);
    }
  }
  {
    std::vector<int> src_ct, src_poly, out_even, out_odd;
    for (size_t d = 0; d < nu2; d++)
      for (size_t j = 0; j < P.t_gsw; j++) {
        int b = (int)(d * P.t_gsw + j);
        src_ct.push_back(2 * b + 1);
        src_poly.push_back((2 * b + 1) * 2);
        int row0 = (int)(d * 2 * 4 * P.t_gsw + 2 * P.t_gsw);
        out_even.push_back(row0 + (int)(2 * j));
        out_odd.push_back(row0 + (int)(2 * j + 1));
      }
    D->gsw_src_ct = put(src_ct);
    D->gsw_src_poly = put(src_poly);
    D->gsw_out_even = put(out_even);
    D->gsw_out_odd = put(out_odd);
  }
  {
    std::vector<int> src_ct, out, row;
    const int n = (int)P.n;
    for (int inst = 0; inst < (int)P.instances; inst++)
      for (int c = 0; c < n; c++) {
        out.push_back(inst * (n + 1) * n + c);
        for (int r = 0; r < n; r++) {
          src_ct.push_back(inst * n * n + r * n + c);
          row.push_back(inst * (n + 1) * n + (1 + r) * n + c);
        }
      }
    D->pack_src_ct = put(src_ct);
    D->pack_out = put(out);
    D->pack_row = put(row);
  }
  if (P.version == 1 && P.n == 2) {
    std::vector<int> row1, ssrc, rd, ra, rb, sd, sa, sb;
    const int ne = (int)P.instances * 2;
    for (int b = 0; b < ne * 2; b++) row1.push_back(b * 3 + 1);
    for (int e = 0; e < ne; e++) {
      const int inst = e / 2, c = e % 2, b0 = e * 2, b1 = e * 2 + 1;
      ssrc.push_back(b1 * 3);
      rd.push_back(e * 3 + 1); ra.push_back(e * 3 + 1); rb.push_back(b1 * 3 + 2);
      rd.push_back(e * 3 + 2); ra.push_back(e * 3 + 2); rb.push_back(b1 * 3 + 1);
      for (int rr = 0; rr < 3; rr++) {
        sd.push_back(inst * 6 + rr * 2 + c);
        sa.push_back(b0 * 3 + rr);
        sb.push_back(e * 3 + rr);
      }
    }
    D->v1_row1 = put(row1);
    D->v1_shift_src = put(ssrc);
    D->v1_rot_dst = put(rd); D->v1_rot_a = put(ra); D->v1_rot_b = put(rb);
    D->v1_sum_dst = put(sd); D->v1_sum_a = put(sa); D->v1_sum_b = put(sb);
  }
  D->lists.alloc(std::max<size_t>(L.size(), 1));
  if (!L.empty()) upload_words(D->lists.p, L.data(), L.size() * sizeof(int));
  return D;
}

}  // namespace spiral

using namespace spiral;

DeviceState& sp_params::device_state() {
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  for (auto& d : this->dev)
    if (d->device == dev) return *d;
  this->dev.push_back(build_device_state(p, dev));
  return *this->dev.back();
}

std::unique_ptr<Workspace> sp_params::acquire_ws() {
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < ws_pool.size(); i++)
      if (ws_pool[i]->device == dev) {
        auto w = std::move(ws_pool[i]);
        ws_pool.erase(ws_pool.begin() + i);
        return w;
      }
  }
  // A new workspace gets every buffer of the expanded-query flow NOW (sizes depend on the parameters only) and the
  // device is drained afterwards: no hipMalloc / hipFree happens while kernels of a query are in flight, and the
  // first query of a workspace costs what the later ones cost.  (Direct-upload and sparse flows size theirs later.)
  auto w = std::make_unique<Workspace>(p, device_state());
  if (tunable("ws_prealloc", 1) != 0) {
    if (p.expand_queries) w->ensure_expand();
    w->ensure_sweep();
    w->ensure_finish();
    HIP_CHECK(hipDeviceSynchronize());
  }
  return w;
}

void sp_params::release_ws(std::unique_ptr<Workspace> ws) {
  std::lock_guard<std::mutex> lk(mu);
  ws_pool.push_back(std::move(ws));
}

sp_params::~sp_params() {}

namespace spiral {

// ---------------------------------------------------------------------------------- workspace
static int tail_defer_levels(const Params& p, long stop);
Workspace::Workspace(const Params& P, DeviceState& D) : P(&P), D(&D) {
  device = D.device;
  HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&ev_fold, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_round0, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_right, hipEventDisableTiming));
  ev_plane.resize(P.planes());
  for (auto& e : ev_plane) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
  for (auto& e : ev_sw) HIP_CHECK(hipEventCreate(&e));
  h_packed_words = P.instances * (P.n + 1) * P.n * POLY_LEN;
  host_pinned = tunable("ws_pinned", 1) != 0;  // 0 (debug): pageable host staging, copies become synchronous
  if (host_pinned) {
    HIP_CHECK(hipHostMalloc((void**)&h_query, 2 * POLY_LEN * sizeof(u64), hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void**)&h_packed, h_packed_words * sizeof(u64), hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void**)&h_response, P.response_bytes() + 16, hipHostMallocDefault));
  } else {
    h_query = (u64*)malloc(2 * POLY_LEN * sizeof(u64));
    h_packed = (u64*)malloc(h_packed_words * sizeof(u64));
    h_response = (uint8_t*)malloc(P.response_bytes() + 16);
    if (!h_query || !h_packed || !h_response) throw OomError("host staging allocation failed");
  }
  enc_out.alloc(P.response_bytes() / 8 + 2);
  q_raw.alloc(2 * POLY_LEN);
  fused_min_pairs = tunable("fused_min_pairs", 256);
}

Workspace::~Workspace() {
  if (host_pinned) {
    if (h_query) (void)hipHostFree(h_query);
    if (h_group_query) (void)hipHostFree(h_group_query);
    if (h_packed) (void)hipHostFree(h_packed);
    if (h_response) (void)hipHostFree(h_response);
  } else {
    free(h_query);
    free(h_group_query);
    free(h_packed);
    free(h_response);
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_plane)
    if (e) (void)hipEventDestroy(e);
  if (ev_fold) (void)hipEventDestroy(ev_fold);
  for (hipEvent_t e : {ev_sw[0], ev_sw[1], ev_round0, ev_right})
    if (e) (void)hipEventDestroy(e);
  if (stream2) (void)hipStreamDestroy(stream2);
  if (stream) (void)hipStreamDestroy(stream);
}

void Workspace::ensure_expand() {
  const Params& p = *P;
  const size_t g = p.g();
  v.ensure(((size_t)1 << g) * 2 * 2 * POLY_LEN);
  exp_raw.ensure(std::max<size_t>(D->max_all, 1) * 2 * POLY_LEN);
  size_t dig = D->max_left * p.t_exp_left + D->max_right * p.t_exp_right;  // both groups of a round coexist
  exp_dig.ensure(std::max<size_t>(dig, 1) * 2 * POLY_LEN);
  exp_ct1.ensure(std::max<size_t>(D->max_all, 1) * 2 * POLY_LEN);
  if (p.db_dim_2 > 0) {  // scratch of the odd subtree (its rounds run on stream2 beside the even subtree's)
    exp_raw_r.ensure(std::max<size_t>(D->max_all, 1) * 2 * POLY_LEN);
    exp_dig_r.ensure(std::max<size_t>(D->max_right * p.t_exp_right, 1) * 2 * POLY_LEN);
    exp_ct1_r.ensure(std::max<size_t>(D->max_all, 1) * 2 * POLY_LEN);
  }
  qv.ensure(POLY_LEN * p.dim0() * 2);
  const size_t nb = p.db_dim_2 * p.t_gsw;
  fold_mats.ensure(std::max<size_t>(p.db_dim_2, 1) * 2 * 4 * p.t_gsw * 2 * POLY_LEN);
  if (p.db_dim_2 > 0 && fused_fold_supported(p)) fold_mats_w.ensure(p.db_dim_2 * 2 * 4 * p.t_gsw * 2 * POLY_LEN);  // run_mats_to_wave
  gsw_raw.ensure(std::max<size_t>(nb, 1) * 2 * POLY_LEN);
  gsw_dig.ensure(std::max<size_t>(nb, 1) * 2 * p.t_conv * 2 * POLY_LEN);
}

size_t Workspace::plane_group() const {
  const Params& p = *P;
  const bool fused_ok = fused_fold_supported(p) && fused_min_pairs < (1L << 40);
  // raw ct staging (X, Y) per plane; plus the digit staging when the unfused path handles whole levels
  size_t per_plane = p.num_per() * 2 * POLY_LEN * sizeof(u64) * 2;
  if (!fused_ok) per_plane += p.num_per() * 2 * p.t_gsw * 2 * POLY_LEN * sizeof(u32);
  size_t budget = (size_t)2 << 30;
  size_t pg = std::max<size_t>(1, budget / std::max<size_t>(per_plane, 1));
  return std::min(pg, p.planes());
}

void Workspace::ensure_sweep() {
  const Params& p = *P;
  sweep_out.ensure(p.planes() * 4 * POLY_LEN * p.num_per());
}

void Workspace::ensure_finish() {
  const Params& p = *P;
  const size_t pg = plane_group();
  foldX.ensure(pg * p.num_per() * 2 * POLY_LEN);
  foldY.ensure(std::max<size_t>(pg * p.num_per() / 2, 1) * 2 * POLY_LEN);
  // the digit-NTT staging is only used by the unfused tree tail (levels with < fused_min_pairs units)
  const bool fused_ok = fused_fold_supported(p);
  size_t dig_cts = pg * p.num_per();
  if (fused_ok && fused_min_pairs < (1L << 40)) dig_cts = std::min<size_t>(dig_cts, std::max<size_t>(2 * (size_t)fused_min_pairs, 2 * pg));
  fold_dig.ensure(dig_cts * 2 * p.t_gsw * 2 * POLY_LEN);
  fold_ntt.ensure(std::max<size_t>(dig_cts / 2, 1) * 2 * 2 * POLY_LEN);
  final_cts.ensure(p.planes() * 2 * POLY_LEN);
  const size_t nb = p.planes();
  pack_dig.ensure(nb * p.t_conv * 2 * POLY_LEN);
  pack_ct2.ensure(nb * 2 * POLY_LEN);
  pack_res.ensure(p.instances * (p.n + 1) * p.n * 2 * POLY_LEN);
  pack_raw.ensure(p.instances * (p.n + 1) * p.n * POLY_LEN);
  // parking buffer of the pipelined query's batched fold tails, sized from the same switch run_sweep_pipelined reads
  // (pipe_tail_defer, default 256 ciphertexts per plane): nothing is allocated inside a pipelined query
  const int defer = tail_defer_levels(p, tunable("pipe_tail_defer", 256));
  if (defer > 0 && p.num_per() >= 1024) fold_tail.ensure(2 * p.planes() * (p.num_per() >> defer) * 2 * POLY_LEN);
}

// ---------------------------------------------------------------------------------- pipeline stages
// server.rs:19-121; v[0] holds the NTT'd query ct.
static void on_stream(Workspace& W, hipStream_t s, const std::function<void()>& f) {
  hipStream_t saved = W.stream;
  W.stream = s;
  try {
    f();
  } catch (...) {
    W.stream = saved;
    throw;
  }
  W.stream = saved;
}

void join_right(Workspace& W);

// the launches of ONE expansion round of one query (server.rs:80-110), as descriptors
struct RoundLaunches {
  InvDesc inv;            // (1) v[num_in + i] = neg1[r] * v[i] fused into ct = from_ntt(v_i); ct_auto = automorph(ct, t)
  FwdDesc fd[3];          // (2) gadget_invert_rdim(ct_auto, rdim = 1) -> to_ntt_no_reduce for both sides, and to_ntt(ct_auto row 1)
  MacDesc md[2];          // (3) v_i += W * ginv + [0; to_ntt(ct_auto row 1)]   (server.rs:89-102)
  ExpandSideDesc es[2];   // (2) + (3) as one launch (k_expand_round)
  long round_transforms;  // digit transforms per modulus
  bool digits_fit;        // k_expand_round takes the round's digit widths
};
static RoundLaunches round_launches(Workspace& W, const sp_pp& pp, const RoundPlan& rp, size_t r, const int* L, int tree) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  // the odd subtree has its own scratch: it runs concurrently with the even subtree (run_begin)
  u64* const exp_raw = tree == 2 ? W.exp_raw_r.p : W.exp_raw.p;
  u32* const exp_dig = tree == 2 ? W.exp_dig_r.p : W.exp_dig.p;
  u32* const exp_ct1 = tree == 2 ? W.exp_ct1_r.p : W.exp_ct1.p;
  const int tl = (int)p.t_exp_left, tr = (int)p.t_exp_right;
  RoundLaunches R{};
  InvDesc& inv = R.inv;
  inv.src = W.v.p;
  inv.idx = L + rp.all_ct;
  inv.polys_per_idx = 2;
  inv.idx_stride = 4 * POLY_LEN;
  inv.poly_stride = 2 * POLY_LEN;
  inv.crt_stride = POLY_LEN;
  inv.z_stride = 1;
  inv.dst = exp_raw;
  inv.n_polys = rp.n_all * 2;
  inv.automorph_t = rp.t_auto;
  inv.scal = D.neg1.p + r * 2 * POLY_LEN;
  inv.scal_dst = W.v.p;
  inv.scal_thresh = rp.num_in;
  inv.scal_only_idx = L + rp.skip2;
  inv.n_scalar_only = rp.n_skip2;
  for (int side = 0; side < 2; side++) {
    const int cnt = side == 0 ? rp.n_left : rp.n_right;
    const int t = side == 0 ? tl : tr;
    FwdDesc f{};
    f.src = exp_raw;
    f.src_idx = L + (side == 0 ? rp.left_pos : rp.right_pos);
    f.dst = exp_dig + (side == 0 ? 0 : (size_t)rp.n_left * tl * 2 * POLY_LEN);
    f.n_out = cnt * t;
    f.rdim = 1;
    f.cols = 1;
    f.t = t;
    f.bits = (int)p.bits_per(t);
    f.src_batch_stride = 2;
    f.src_row0 = 0;
    f.src_cols = 1;
    R.fd[side] = f;
  }
  {
    FwdDesc f1{};
    f1.src = exp_raw;
    f1.dst = exp_ct1;
    f1.n_out = rp.n_all;
    f1.rdim = 1;
    f1.cols = 1;
    f1.t = 1;
    f1.bits = 64;
    f1.src_batch_stride = 2;
    f1.src_row0 = 1;
    f1.src_cols = 1;
    R.fd[2] = f1;
  }
  R.round_transforms = (long)rp.n_left * tl + (long)rp.n_right * tr;
  R.digits_fit = R.fd[0].bits <= 28 && R.fd[1].bits <= 28;
  for (int side = 0; side < 2; side++) {
    const int cnt = side == 0 ? rp.n_left : rp.n_right;
    const int t = side == 0 ? tl : tr;
    // nu_2 == 0: expand_query passes v_w_left for both sides (server.rs:573)
    const bool use_right = side == 1 && pp.has_right && p.db_dim_2 > 0;
    const size_t woff = use_right ? pp.off_right + r * 2 * tr : pp.off_left + r * 2 * tl;
    MacDesc m{};
    m.A = pp.all.p + woff * 2 * POLY_LEN;
    m.B = R.fd[side].dst;
    m.out = W.v.p;
    m.addend = W.v.p;
    m.out_idx = L + (side == 0 ? rp.left_out : rp.right_out);
    m.R = 2;
    m.K = t;
    m.batch_inner = cnt;
    m.batch_outer = 1;
    m.B_inner_stride = t;
    m.B_outer_stride = 0;
    m.split_k = t;
    m.split_off = 0;
    m.out_batch_stride = 0;
    m.out_row_stride = 1;
    m.extra = exp_ct1;
    m.extra_idx = L + (side == 0 ? rp.left_pos : rp.right_pos);
    m.extra_row = 1;
    R.md[side] = m;
    R.es[side] = ExpandSideDesc{R.fd[side].src, R.fd[side].src_idx, m.out_idx, m.A, m.out, m.batch_inner, R.fd[side].t, R.fd[side].bits};
  }
  return R;
}

void run_coefficient_expansion(Workspace& W, const sp_pp& pp, size_t g_rounds, const DeviceState::PrunedPlan* plan,
                               int tree, size_t r_begin) {
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  const int* L = plan ? plan->lists.p : D.lists.p;
  const std::vector<RoundPlan>& rounds = tree == 1   ? (plan ? plan->rounds_even : D.rounds_even)
                                         : tree == 2 ? (plan ? plan->rounds_odd : D.rounds_odd)
                                                     : (plan ? plan->rounds : D.rounds);
  for (size_t r = r_begin; r < g_rounds; r++) {
    const RoundPlan& rp = rounds[r];
    if (rp.n_all == 0 && (tree != 0 || rp.n_skip2 == 0)) continue;  // (a subtree's skipped cts are never read again)
    const RoundLaunches R = round_launches(W, pp, rp, r, L, tree);
    // three launches per round -- or two: (2) + (3) in one launch where the round is large enough to fill the chip with one
    // workgroup per (ciphertext, modulus): nothing but the results is written (kernels.hpp, launch_expand_round; switch
    // expand_round_min = digit transforms per modulus).
    // NOT for the odd subtree of a split expansion (tree 2: it runs on the second stream beside the first sweep launches, and
    // its 57-pass workgroups cost the sweep more than the rounds gain: profiles/r05_expand_round.md; switch expand_round_odd)
    const bool one_launch = R.round_transforms >= tunable("expand_round_min", EXPAND_ROUND_MIN_DEFAULT) && R.digits_fit &&
                            (tree != 2 || tunable("expand_round_odd", 0) != 0);
    launch_ntt_inv(D.T, R.inv, s);
    if (one_launch) {
      launch_expand_round(D.T, R.es[0], R.es[1], s);
    } else {
      launch_ntt_fwd3(D.T, R.fd[0], R.fd[1], R.fd[2], s);
      launch_mac2(D.T, R.md[0], R.md[1], s);
    }
  }
}

// The same rounds for a GROUP of B queries (same Params; each its own workspace and public parameters), every launch shared: one
// more grid dimension = the query, query qi's buffers addressed as query 0's plus a byte offset per class (kernels.hpp, GroupOff).
// Whole tree, no pruning, on Ws[0]'s stream -- the caller has ordered that stream after every query's v[0] and orders the queries'
// own streams after it.  A round is one launch (k_expand_round) from `expand_group_round_min` digit transforms of the whole group.
void run_group_expansion(Workspace* const* Ws, const sp_pp* const* pps, int B, size_t g_rounds) {
  Workspace& W0 = *Ws[0];
  DeviceState& D = *W0.D;
  hipStream_t s = W0.stream;
  const int* L = D.lists.p;
  if (B < 1 || B > GROUP_MAX) throw ArgError("group size out of range");
  GroupOff g{};
  // (integer arithmetic on addresses: the buffers are separate allocations, a pointer difference between them is not defined)
  auto off = [](const void* q, const void* q0) { return (long long)((uintptr_t)q - (uintptr_t)q0); };
  for (int i = 0; i < B; i++) {
    Workspace& W = *Ws[i];
    if (W.P != W0.P || W.D != W0.D) throw ArgError("a group's queries must share params and device");
    g.v[i] = off(W.v.p, W0.v.p);
    g.raw[i] = off(W.exp_raw.p, W0.exp_raw.p);
    g.dig[i] = off(W.exp_dig.p, W0.exp_dig.p);
    g.ct1[i] = off(W.exp_ct1.p, W0.exp_ct1.p);
    g.pp[i] = off(pps[i]->all.p, pps[0]->all.p);
    if (pps[i]->has_right != pps[0]->has_right || pps[i]->off_left != pps[0]->off_left || pps[i]->off_right != pps[0]->off_right)
      throw ArgError("a group's public parameters must have one layout");
  }
  const long group_min = tunable("expand_group_round_min", EXPAND_GROUP_ROUND_MIN_DEFAULT);
  const long wave_min_digits = tunable("expand_wave_min_digits", EXPAND_WAVE_MIN_DIGITS_DEFAULT);   // 0: never
  bool wave_ok = wave_min_digits > 0;
  GroupOff gw = g;   // k_expand_wave reads the public parameters' WAVE-layout copies
  for (int i = 0; i < B; i++) {
    wave_ok = wave_ok && pps[i]->all_w.p != nullptr && pps[i]->all_w.n == pps[0]->all_w.n;
    if (wave_ok) gw.pp[i] = off(pps[i]->all_w.p, pps[0]->all_w.p);
  }
  for (size_t r = 0; r < g_rounds; r++) {
    const RoundPlan& rp = D.rounds[r];
    if (rp.n_all == 0 && rp.n_skip2 == 0) continue;
    const RoundLaunches R = round_launches(W0, *pps[0], rp, r, L, 0);
    launch_ntt_inv_group(D.T, R.inv, g, B, s);
    if (R.round_transforms * B >= group_min && R.digits_fit) {
      // a side with many digits per ciphertext (the right-hand side's 56 one-bit digits) on the wave-per-transform engine, the
      // other side on the cooperative kernel (a left-hand ciphertext's 9 transforms would be overhead-bound in the wave shape)
      ExpandSideDesc coop[2] = {R.es[0], R.es[1]};
      for (int side = 1; side >= 0; side--)
        if (wave_ok && coop[side].cnt > 0 && coop[side].t >= wave_min_digits && coop[side].t <= 56) {
          ExpandWaveDesc w{};
          w.raw = coop[side].raw;
          w.pos = coop[side].pos;
          w.out_idx = coop[side].out_idx;
          w.A_w = pps[0]->all_w.p + (coop[side].A - pps[0]->all.p);   // the same polynomials, wave layout
          w.const_w = pps[0]->all_w.p + pps[0]->all.n;                  // (0 | 1 live behind the copy, at the same offset in every pp)
          w.v = coop[side].v;
          w.cnt = coop[side].cnt;
          w.t = coop[side].t;
          w.bits = coop[side].bits;
          launch_expand_wave(D.T, w, gw, B, s);
          coop[side].cnt = 0;
        }
      launch_expand_round_group(D.T, coop[0], coop[1], g, B, s);
    } else {
      launch_ntt_fwd3_group(D.T, R.fd[0], R.fd[1], R.fd[2], g, B, s);
      launch_mac2_group(D.T, R.md[0], R.md[1], g, B, s);
    }
  }
}

// server.rs:123-151 with idx_factor 1, idx_offset 0, reading the inputs v[2b+1] in place and writing
// the GSW matrices into the right half of fold_mats.
void run_regev_to_gsw(Workspace& W, const sp_pp& pp, const u32* v_src, const int* src_ct, const int* src_poly) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  const int* L = D.lists.p;
  const int nb = (int)(p.db_dim_2 * p.t_gsw);
  if (nb == 0) return;
  const int four_t = (int)(4 * p.t_gsw);
  launch_copy_polys(W.fold_mats.p, L + D.gsw_out_odd, four_t, v_src, src_poly, 1, 2, nb, s);
  InvDesc inv{};
  inv.src = v_src;
  inv.idx = src_ct;
  inv.polys_per_idx = 2;
  inv.idx_stride = 4 * POLY_LEN;
  inv.poly_stride = 2 * POLY_LEN;
  inv.crt_stride = POLY_LEN;
  inv.z_stride = 1;
  inv.dst = W.gsw_raw.p;
  inv.n_polys = nb * 2;
  launch_ntt_inv(D.T, inv, s);
  FwdDesc f{};
  f.src = W.gsw_raw.p;
  f.dst = W.gsw_dig.p;
  f.n_out = nb * 2 * (int)p.t_conv;
  f.rdim = 2;
  f.cols = 1;
  f.t = (int)p.t_conv;
  f.bits = (int)p.bits_per(p.t_conv);
  f.src_batch_stride = 2;
  f.src_row0 = 0;
  f.src_cols = 1;
  launch_ntt_fwd(D.T, f, s);
  MacDesc m{};
  m.A = pp.all.p + pp.off_conv * 2 * POLY_LEN;
  m.B = W.gsw_dig.p;
  m.out = W.fold_mats.p;
  m.out_idx = L + D.gsw_out_even;
  m.R = 2;
  m.K = 2 * (int)p.t_conv;
  m.batch_inner = nb;
  m.batch_outer = 1;
  m.B_inner_stride = 2 * p.t_conv;
  m.split_k = m.K;
  m.out_row_stride = four_t;
  launch_mac(D.T, m, s);
}

void run_folding_neg(Workspace& W) {
  const Params& p = *W.P;
  launch_folding_neg(W.D->T, W.fold_mats.p, W.D->gadget_gsw.p, (int)p.db_dim_2, (int)(2 * p.t_gsw), W.stream);
  run_mats_to_wave(W, p.db_dim_2);
}

// What the QUERY path does in get_v_folding_neg's place (server.rs:505-523, 660): nothing but the wave-layout copy of the C halves.
// Every fold step of this library is the delta form  ct_i + C (*) (G^-1(ct_{i+half}) - G^-1(ct_i))  -- exactly
// (G - C) (*) ct_i + C (*) ct_{i+half}  mod q0, q1, because G G^-1(x) = x for the unsigned digit slices -- in the fused kernels
// (fold.hip: kk = two_t + ...) and in the tree tail (run_fold, delta_tail): G - C is never read.  r01-r05 computed it all the same
// (k_folding_neg, 9-16 us per query, 73 us per 16-query group) and converted both halves to the wave layout.  The stage export
// sp_get_v_folding_neg still computes it (run_folding_neg), and sp_fold_ciphertexts takes the caller's.
// Switch fold_neg_materialise = 1: the old behaviour.
void run_fold_operands(Workspace& W) {
  if (tunable("fold_neg_materialise", 0) != 0) {
    run_folding_neg(W);
    return;
  }
  run_mats_to_wave(W, W.P->db_dim_2, true);
}

// the fold operands once more in wave layout when the wave-per-transform fold kernel is selected (fold_variant 5)
void run_mats_to_wave(Workspace& W, size_t levels, bool c_only) {
  const Params& p = *W.P;
  W.mats_w_ready = false;
  if (tunable("fold_variant", FOLD_VARIANT_DEFAULT) != 5 || !fused_fold_supported(p) || levels == 0) return;
  const size_t words = levels * 2 * 4 * p.t_gsw * 2 * POLY_LEN;
  W.fold_mats_w.ensure(words);
  launch_mats_to_wave(W.fold_mats_w.p, W.fold_mats.p, words, W.stream, c_only ? (int)(2 * p.t_gsw) : 0);
  W.mats_w_ready = true;
}

// Non-expanded ("direct_upload") queries: Query::deserialize (client.rs:315-327) regenerates the public
// halves from the seed -- row 0 of each of the dim0 Regev cts (interleave_rng_data, client.rs:105-128), then
// row 0 of each GSW matrix -- and process_query uses v_buf / v_ct.ntt() as they are (server.rs:666-679).
void run_begin_direct(Workspace& W, const uint8_t* query) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  note_path(PATH_DIRECT_UPLOAD);
  const size_t dim0 = p.dim0(), nu2 = p.db_dim_2, two_t = 2 * p.t_gsw;
  const size_t n_reg = dim0 * POLY_LEN;
  const size_t gsw_polys = nu2 * 2 * two_t;
  const size_t n_rng = n_reg + nu2 * two_t * POLY_LEN;
  std::vector<u64> ks(n_rng);
  chacha20_keystream_u64(query, ks.data(), n_rng);
  for (auto& x : ks) x = p.modulus - (x % p.modulus);  // get_inv_from_rng, client.rs:47-49
  const uint8_t* wire = query + SEED_LENGTH;
  // raw image: [dim0 reg row-0 polys][nu2 GSW matrices 2 x 2t]
  std::vector<u64> raw((dim0 + gsw_polys) * POLY_LEN);
  memcpy(raw.data(), ks.data(), n_reg * 8);
  const uint8_t* wire_gsw = wire + n_reg * 8;
  for (size_t d = 0; d < nu2; d++) {
    u64* m = raw.data() + (dim0 + d * 2 * two_t) * POLY_LEN;
    memcpy(m, ks.data() + n_reg + d * two_t * POLY_LEN, two_t * POLY_LEN * 8);                 // row 0
    memcpy(m + two_t * POLY_LEN, wire_gsw + d * two_t * POLY_LEN * 8, two_t * POLY_LEN * 8);     // row 1
  }
  W.du_raw.ensure(raw.size());
  W.du_wire.ensure(n_reg);
  W.du_ntt.ensure((dim0 + gsw_polys) * 2 * POLY_LEN);
  W.qv.ensure(POLY_LEN * dim0 * 2);
  W.fold_mats.ensure(std::max<size_t>(nu2, 1) * 2 * 2 * two_t * 2 * POLY_LEN);
  HIP_CHECK(hipMemcpyAsync(W.du_raw.p, raw.data(), raw.size() * 8, hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemcpyAsync(W.du_wire.p, wire, n_reg * 8, hipMemcpyHostToDevice, s));
  FwdDesc f{W.du_raw.p, nullptr, W.du_ntt.p, (int)(dim0 + gsw_polys), 1, 1, 1, 64, 1, 0, 1};
  launch_ntt_fwd(D.T, f, s);
  launch_interleave_query(W.qv.p, W.du_ntt.p, W.du_wire.p, (int)dim0, s);
  const u32* gsw = W.du_ntt.p + dim0 * 2 * POLY_LEN;
  for (size_t d = 0; d < nu2; d++)
    for (size_t r = 0; r < 2; r++)
      HIP_CHECK(hipMemcpyAsync(W.fold_mats.p + ((d * 2 + r) * 2 * two_t + two_t) * 2 * POLY_LEN,
                               gsw + ((d * 2 + r) * two_t) * 2 * POLY_LEN, two_t * 2 * POLY_LEN * sizeof(u32),
                               hipMemcpyDeviceToDevice, s));
  run_fold_operands(W);
  HIP_CHECK(hipStreamSynchronize(s));  // the host staging vectors go out of scope
}

// Query::deserialize (client.rs:303-314): the query ciphertext's two raw polynomials into `host` -- row 0 = Q - (rng.gen::<u64>() % Q)
// from the seed (client.rs:47-49), row 1 from the wire
static void query_ct_host(Workspace& W, const uint8_t* query, size_t query_len, u64* host) {
  const Params& p = *W.P;
  if (query_len != p.query_bytes()) throw ArgError("query length " + std::to_string(query_len) + " != query_bytes " + std::to_string(p.query_bytes()));
  if (p.db_dim_2 == 0 && p.t_exp_left != p.t_exp_right) throw ArgError("nu_2 == 0 requires t_exp_left == t_exp_right (server.rs:573)");
  chacha20_keystream_u64(query, host, POLY_LEN);
  const FastMod modq(p.modulus);
  for (size_t i = 0; i < POLY_LEN; i++) host[i] = p.modulus - modq(host[i]);
  memcpy(host + POLY_LEN, query + SEED_LENGTH, POLY_LEN * sizeof(u64));
}

// ... up to v[0] = query.ct.ntt() (server.rs:545), enqueued on W.stream
static void run_begin_query_ct(Workspace& W, const uint8_t* query, size_t query_len) {
  query_ct_host(W, query, query_len, W.h_query);
  W.ensure_expand();
  hipStream_t s = W.stream;
  join_right(W);  // a previous query of this workspace whose odd subtree nobody waited for
  W.right_pending = false;
  HIP_CHECK(hipMemcpyAsync(W.q_raw.p, W.h_query, 2 * POLY_LEN * sizeof(u64), hipMemcpyHostToDevice, s));
  FwdDesc f{W.q_raw.p, nullptr, W.v.p, 2, 1, 1, 1, 64, 1, 0, 1};
  launch_ntt_fwd(W.D->T, f, s);
}

// what follows the rounds of an un-pruned, un-split expansion: v_reg, the GSW side, G - C (server.rs:566-591, 505-523)
static void run_begin_after_rounds(Workspace& W, const sp_pp& pp) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  const int* L = D.lists.p;
  if (p.db_dim_2 > 0) {
    launch_reorient(W.qv.p, W.v.p, 0, 2, (int)p.dim0(), W.stream);  // v_reg_inp[i] = v[2i]   (server.rs:566-568)
    run_regev_to_gsw(W, pp, W.v.p, L + D.gsw_src_ct, L + D.gsw_src_poly);
    run_fold_operands(W);
  } else {
    launch_reorient(W.qv.p, W.v.p, 0, 1, (int)p.dim0(), W.stream);  // server.rs:574-576
  }
}

// run_begin_after_rounds for the group, as shared launches on Ws[0]'s stream (nu_2 > 0, wave-layout fold operands): reorient,
// regev_to_gsw (copy, inverse transform, digit transforms, products: server.rs:123-151), G - C, wave layout
static void run_group_after_rounds(Workspace* const* Ws, const sp_pp* const* pps, int B) {
  Workspace& W0 = *Ws[0];
  const Params& p = *W0.P;
  DeviceState& D = *W0.D;
  hipStream_t s = W0.stream;
  const int* L = D.lists.p;
  // (integer arithmetic on addresses: the buffers are separate allocations, a pointer difference between them is not defined)
  auto off = [](const void* q, const void* q0) { return (long long)((uintptr_t)q - (uintptr_t)q0); };
  const int nb = (int)(p.db_dim_2 * p.t_gsw);
  const int four_t = (int)(4 * p.t_gsw);
  const size_t mats_words = p.db_dim_2 * 2 * 4 * p.t_gsw * 2 * POLY_LEN;
  for (int i = 0; i < B; i++) Ws[i]->fold_mats_w.ensure(mats_words);
  // offset classes per launch: see the launchers' comments in kernels.hpp
  GroupOff g_reo{}, g_copy{}, g_inv{}, g_fwd{}, g_mac{}, g_neg{}, g_wave{};
  for (int i = 0; i < B; i++) {
    Workspace& W = *Ws[i];
    const long long dv = off(W.v.p, W0.v.p), dm = off(W.fold_mats.p, W0.fold_mats.p);
    g_reo.v[i] = dv;   g_reo.raw[i] = off(W.qv.p, W0.qv.p);
    g_copy.v[i] = dv;  g_copy.raw[i] = dm;
    g_inv.v[i] = dv;   g_inv.raw[i] = off(W.gsw_raw.p, W0.gsw_raw.p);
    g_fwd.raw[i] = g_inv.raw[i];  g_fwd.dig[i] = off(W.gsw_dig.p, W0.gsw_dig.p);
    g_mac.pp[i] = off(pps[i]->all.p, pps[0]->all.p);  g_mac.dig[i] = g_fwd.dig[i];  g_mac.v[i] = dm;
    g_neg.v[i] = dm;
    g_wave.v[i] = dm;  g_wave.raw[i] = off(W.fold_mats_w.p, W0.fold_mats_w.p);
    if (pps[i]->off_conv != pps[0]->off_conv) throw ArgError("a group's public parameters must have one layout");
  }
  launch_reorient_group(W0.qv.p, W0.v.p, 0, 2, (int)p.dim0(), g_reo, B, s);  // v_reg_inp[i] = v[2i]   (server.rs:566-568)
  if (nb > 0) {
    launch_copy_polys_group(W0.fold_mats.p, L + D.gsw_out_odd, four_t, W0.v.p, L + D.gsw_src_poly, 1, 2, nb, g_copy, B, s);
    InvDesc inv{};
    inv.src = W0.v.p;
    inv.idx = L + D.gsw_src_ct;
    inv.polys_per_idx = 2;
    inv.idx_stride = 4 * POLY_LEN;
    inv.poly_stride = 2 * POLY_LEN;
    inv.crt_stride = POLY_LEN;
    inv.z_stride = 1;
    inv.dst = W0.gsw_raw.p;
    inv.n_polys = nb * 2;
    launch_ntt_inv_group(D.T, inv, g_inv, B, s);
    FwdDesc f{};
    f.src = W0.gsw_raw.p;
    f.dst = W0.gsw_dig.p;
    f.n_out = nb * 2 * (int)p.t_conv;
    f.rdim = 2;
    f.cols = 1;
    f.t = (int)p.t_conv;
    f.bits = (int)p.bits_per(p.t_conv);
    f.src_batch_stride = 2;
    f.src_row0 = 0;
    f.src_cols = 1;
    FwdDesc none{};
    launch_ntt_fwd3_group(D.T, f, none, none, g_fwd, B, s);
    MacDesc m{};
    m.A = pps[0]->all.p + pps[0]->off_conv * 2 * POLY_LEN;
    m.B = W0.gsw_dig.p;
    m.out = W0.fold_mats.p;
    m.out_idx = L + D.gsw_out_even;
    m.R = 2;
    m.K = 2 * (int)p.t_conv;
    m.batch_inner = nb;
    m.batch_outer = 1;
    m.B_inner_stride = 2 * p.t_conv;
    m.split_k = m.K;
    m.out_row_stride = four_t;
    MacDesc none_m{};
    launch_mac2_group(D.T, m, none_m, g_mac, B, s);
  }
  // (G - C is not needed by the query path: run_fold_operands)
  const bool neg = tunable("fold_neg_materialise", 0) != 0;
  if (neg) launch_folding_neg_group(D.T, W0.fold_mats.p, D.gadget_gsw.p, (int)p.db_dim_2, (int)(2 * p.t_gsw), g_neg, B, s);
  launch_mats_to_wave_group(W0.fold_mats_w.p, W0.fold_mats.p, mats_words, neg ? 0 : (int)(2 * p.t_gsw), g_wave, B, s);
  for (int i = 0; i < B; i++) Ws[i]->mats_w_ready = true;
}

// run_begin for a GROUP of queries that share one database pass (sp_process_query_batch): every query's ciphertext on its own
// stream, the rounds of all expansions as shared launches on the first query's stream (run_group_expansion), then every query's
// v_reg / GSW side / G - C on its own stream again.  Same values as B run_begin calls: the kernels are the same bodies.
void run_begin_group(Workspace* const* Ws, const sp_pp* const* pps, const uint8_t* const* queries, const size_t* query_lens, int B) {
  Workspace& W0 = *Ws[0];
  const Params& p = *W0.P;
  if (!p.expand_queries) throw ArgError("run_begin_group: direct-upload queries have no expansion to share");
  // the B query ciphertexts: one pinned staging buffer, ONE upload, one launch of their transforms (sixteen uploads + launches on
  // sixteen streams were 0.66 ms before the first shared round could start)
  if (!W0.h_group_query) {
    if (W0.host_pinned)
      HIP_CHECK(hipHostMalloc((void**)&W0.h_group_query, (size_t)GROUP_MAX * 2 * POLY_LEN * sizeof(u64), hipHostMallocDefault));
    else if (!(W0.h_group_query = (u64*)malloc((size_t)GROUP_MAX * 2 * POLY_LEN * sizeof(u64))))
      throw OomError("host staging allocation failed");
  }
  W0.group_q_raw.ensure((size_t)GROUP_MAX * 2 * POLY_LEN);
  GroupOff gq{};
  for (int i = 0; i < B; i++) {
    Workspace& W = *Ws[i];
    query_ct_host(W, queries[i], query_lens[i], W0.h_group_query + (size_t)i * 2 * POLY_LEN);
    W.ensure_expand();
    if (W.right_pending) {   // a previous query of this workspace whose odd subtree nobody waited for
      HIP_CHECK(hipStreamWaitEvent(W0.stream, W.ev_right, 0));
      W.right_pending = false;
    }
    gq.raw[i] = (long long)((size_t)i * 2 * POLY_LEN * sizeof(u64));
    gq.dig[i] = (long long)((uintptr_t)W.v.p - (uintptr_t)W0.v.p);
  }
  HIP_CHECK(hipMemcpyAsync(W0.group_q_raw.p, W0.h_group_query, (size_t)B * 2 * POLY_LEN * sizeof(u64), hipMemcpyHostToDevice, W0.stream));
  {
    FwdDesc f{W0.group_q_raw.p, nullptr, W0.v.p, 2, 1, 1, 1, 64, 1, 0, 1};  // v[0] = query.ct.ntt()  (server.rs:545)
    FwdDesc none{};
    launch_ntt_fwd3_group(W0.D->T, f, none, none, gq, B, W0.stream);
  }
  note_path(PATH_EXPAND_GROUP);
  run_group_expansion(Ws, pps, B, p.g());
  if (tunable("expand_group_tail", 1) != 0 && fused_fold_supported(p) && tunable("fold_variant", FOLD_VARIANT_DEFAULT) == 5 && p.db_dim_2 > 0)
    run_group_after_rounds(Ws, pps, B);   // seven shared launches
  else
    for (int i = 0; i < B; i++) {          // (the single-query form of the tail on W0's stream; rare shapes)
      Workspace& W = *Ws[i];
      hipStream_t own = W.stream;
      W.stream = W0.stream;
      try {
        run_begin_after_rounds(W, *pps[i]);
      } catch (...) {
        W.stream = own;
        throw;
      }
      W.stream = own;
    }
  HIP_CHECK(hipEventRecord(W0.ev_round0, W0.stream));
  for (int i = 1; i < B; i++) HIP_CHECK(hipStreamWaitEvent(Ws[i]->stream, W0.ev_round0, 0));
}

// Query::deserialize (client.rs:303-314) + expand_query (server.rs:525-591) + get_v_folding_neg
void run_begin(Workspace& W, const sp_pp& pp, const uint8_t* query, size_t query_len, int j0, int nj,
               const DeviceState::PrunedPlan* plan) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  if (query_len != p.query_bytes()) throw ArgError("query length " + std::to_string(query_len) + " != query_bytes " + std::to_string(p.query_bytes()));
  if (!p.expand_queries) {
    run_begin_direct(W, query);
    return;
  }
  run_begin_query_ct(W, query, query_len);
  hipStream_t s = W.stream;
  const size_t g = p.g();
  // a row shard only needs the first-dimension ciphertexts of its rows: prune the even subtree of the expansion
  const bool prune = p.db_dim_2 > 0 && nj > 0 && (j0 != 0 || nj != (int)p.dim0());
  if (prune || (plan && p.db_dim_2 > 0)) note_path(PATH_EXPAND_PRUNED);
  const DeviceState::PrunedPlan* pl = plan && p.db_dim_2 > 0 ? plan : (prune ? &D.pruned_plan(p, j0, nj) : nullptr);
  const int* L = D.lists.p;
  const long split_mode = tunable("expand_split", -1);  // -1: only when a long sweep follows (it hides the odd subtree)
  if (p.db_dim_2 > 0 && g >= 2 && (split_mode > 0 || (split_mode < 0 && W.long_sweep_follows))) {
    // expand_split (default -1: only before a per-plane pipelined sweep, i.e. wide packed databases; 1: always; 0: never):
    // after round 0 the tree falls into the even subtree (-> v_reg, what the sweep waits for) and the odd subtree (-> the
    // GSW bits, regev_to_gsw and G - C, what the FOLD waits for; 56 digits per ciphertext against 8: more than half of the
    // transforms).  The odd side then runs on stream2, beside the even side and the first plane's sweep; consumers of
    // fold_mats order themselves after ev_right (join_right).  Measured (profiles/r02_expand_experiments.md): -6 % at C2,
    // nothing at C1/P2, where the odd subtree then competes with the short sweep.  The enqueue order of the two sides and a
    // one-launch-per-round kernel (one workgroup per ciphertext) were measured in round 2 and bought nothing
    // (profiles/r02_expand_order.txt, r02_expand_experiments.md); both are gone.  Round 4 ran the WHOLE chain as one
    // persistent launch with device-wide barriers between its ~40 phases: byte-identical and 2.5-4x
    // SLOWER (C2 expansion 0.71 -> 2.7 ms) -- on this eight-XCD part a device-wide barrier is an L2 write-back +
    // invalidate per XCD plus hundreds of serialised memory-side atomics, i.e. what a kernel boundary costs, paid by every
    // workgroup; source and numbers: scripts/archive/r04_phase_program/, profiles/r04_phase_program.md.
    note_path(PATH_EXPAND_SPLIT);
    run_coefficient_expansion(W, pp, 1, pl, 0, 0);
    HIP_CHECK(hipEventRecord(W.ev_round0, s));
    HIP_CHECK(hipStreamWaitEvent(W.stream2, W.ev_round0, 0));
    on_stream(W, W.stream2, [&] {
      run_coefficient_expansion(W, pp, g, pl, 2, 1);
      run_regev_to_gsw(W, pp, W.v.p, L + D.gsw_src_ct, L + D.gsw_src_poly);
      run_fold_operands(W);
    });
    HIP_CHECK(hipEventRecord(W.ev_right, W.stream2));
    W.right_pending = true;
    run_coefficient_expansion(W, pp, g, pl, 1, 1);
    launch_reorient(W.qv.p, W.v.p, 0, 2, (int)p.dim0(), s);  // v_reg_inp[i] = v[2i]   (server.rs:566-568)
    return;
  }
  run_coefficient_expansion(W, pp, g, pl);
  run_begin_after_rounds(W, pp);
}

// orders the current stream after the odd expansion subtree of this workspace's query (no-op when it was not split off)
void join_right(Workspace& W) {
  if (W.right_pending) HIP_CHECK(hipStreamWaitEvent(W.stream, W.ev_right, 0));
}

void run_sweep(Workspace& W, const sp_db& db) {
  const Params& p = *W.P;
  W.ensure_sweep();
  SweepDesc d{db.words.p, W.qv.p, W.sweep_out.p, (int)p.planes(), db.np_local, (int)p.dim0(), db.j0, db.nj, db.packed, W.out_G};
  d.nt_store = (int)tunable("sweep_nt_store", 1);
  launch_sweep(W.D->T, d, W.stream);
}

// multiply_reg_by_sparse_database (lib/server/src/compute/dot_product.rs:13-220): only the present items, read from the
// expanded ciphertexts directly (row j = ct 2j, or ct j when nu_2 = 0)
void run_sweep_sparse(Workspace& W, const sp_db& db, const int* col_ptr, const int* col_rows, const int* col_slots) {
  const Params& p = *W.P;
  W.ensure_sweep();
  launch_sweep_sparse(W.D->T, col_ptr, col_rows, col_slots, db.polys.p, (int)p.planes(), W.v.p, 0,
                      p.db_dim_2 > 0 ? 2 : 1, W.sweep_out.p, (int)p.num_per(), W.stream);
}

// from_ntt + fold of `np` planes starting at plane pg0 (sweep_out -> final_cts), on W.stream
// run_fold on ciphertexts this library has just produced itself (from_ntt / fold outputs: canonical, below Q) -- the only callers
// that may let the fused kernels skip the dead top gadget digit (FoldDesc::t_live).  Everything else (stage-level entry points,
// ciphertexts gathered from peer ranks or handed in by the caller of the split API) folds with every digit.
static u64* run_fold_canonical(Workspace& W, u64* X, u64* Y, int np, int num_cts, int top, int d_begin = 0, int d_end = -1) {
  struct Mark {
    Workspace& W;
    explicit Mark(Workspace& w) : W(w) { W.fold_inputs_below_q = true; }
    ~Mark() { W.fold_inputs_below_q = false; }
  } mark(W);
  return run_fold(W, X, Y, np, num_cts, top, d_begin, d_end);
}

static void fold_planes(Workspace& W, size_t pg0, int np, bool premod) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  if (p.num_per() % 4 == 0 && !tunable("from_sweep1", 0)) {
    launch_from_sweep4(D.T, W.sweep_out.p + pg0 * 4 * POLY_LEN * p.num_per(), (int)p.num_per(), np, premod ? 1 : 0,
                       W.foldX.p, s);
  } else {
    InvDesc inv{};
    inv.src = W.sweep_out.p + pg0 * 4 * POLY_LEN * p.num_per();
    inv.sweep_np = (int)p.num_per();
    inv.dst = W.foldX.p;
    inv.n_polys = np * (int)p.num_per() * 2;
    inv.premod = premod ? 1 : 0;
    launch_ntt_inv(D.T, inv, s);
  }
  u64* res = run_fold_canonical(W, W.foldX.p, W.foldY.p, np, (int)p.num_per(), -1);
  HIP_CHECK(hipMemcpyAsync(W.final_cts.p + pg0 * 2 * POLY_LEN, res, (size_t)np * 2 * POLY_LEN * sizeof(u64), hipMemcpyDeviceToDevice, s));
}

// Pipelined queries (switch pipe_tail_defer = cts per plane, 0 = off): under the sweeps a plane is folded only down to
// `stop` ciphertexts (the levels with enough independent steps for one fused launch each) and parked; the remaining
// levels of ALL planes run as one batch once the last plane is there.  The tail levels are latency-bound -- 24 launches of
// 10 us alone, ~45 us each beside a sweep -- so a plane's overlapped fold took longer than the sweep it hides behind and
// the second stream fell further behind with every plane; batched, the four tails cost what one costs.
static int tail_defer_levels(const Params& p, long stop) {  // levels folded per plane before parking (0 = do not defer)
  if (stop <= 0 || !fused_fold_supported(p) || p.planes() < 2) return 0;
  int lv = 0;
  size_t cur = p.num_per();
  while (cur > (size_t)stop && cur / 2 >= 1 && (cur & 1) == 0) { cur >>= 1; lv++; }
  return cur == (size_t)stop && lv > 0 && stop >= 2 ? lv : 0;
}
static void fold_plane_head(Workspace& W, size_t pl, int levels) {  // from_ntt + the first `levels` levels, parked
  const Params& p = *W.P;
  launch_from_sweep4(W.D->T, W.sweep_out.p + pl * 4 * POLY_LEN * p.num_per(), (int)p.num_per(), 1, 0, W.foldX.p, W.stream);
  u64* res = run_fold_canonical(W, W.foldX.p, W.foldY.p, 1, (int)p.num_per(), -1, 0, levels);
  const size_t cts = p.num_per() >> levels;
  HIP_CHECK(hipMemcpyAsync(W.fold_tail.p + pl * cts * 2 * POLY_LEN, res, cts * 2 * POLY_LEN * sizeof(u64), hipMemcpyDeviceToDevice, W.stream));
}
static void fold_tails(Workspace& W, int levels) {  // the parked planes together, down to one ciphertext each
  const Params& p = *W.P;
  const size_t cts = p.num_per() >> levels;
  u64* other = W.fold_tail.p + p.planes() * cts * 2 * POLY_LEN;
  u64* res = run_fold_canonical(W, W.fold_tail.p, other, (int)p.planes(), (int)p.num_per(), -1, levels, -1);
  HIP_CHECK(hipMemcpyAsync(W.final_cts.p, res, p.planes() * 2 * POLY_LEN * sizeof(u64), hipMemcpyDeviceToDevice, W.stream));
}

// Single-GPU query on a wide PACKED database: the database is swept one (instance, trial) plane per launch, and
// the from_ntt + fold of plane p (integer-ALU-bound, ~20 % of the sweep's duration) runs on the second stream while
// plane p+1 is being swept (HBM-bound, ~20 % VALU use, 40 VGPRs per wave: the fold's workgroups fit beside it).
// Only the last plane's fold is exposed.  Worth it when one plane's sweep outlasts one plane's fold, i.e. for
// num_per >= 1024; narrow databases keep the single launch + all-planes fold (their fold is latency-bound).
bool sweep_is_pipelined(const Params& p, const sp_db& db) {
  const bool enabled = tunable("pipeline", 1) != 0;
  return enabled && db.packed && db.num_shards == 1 && db.col_G == 1 && p.planes() > 1 && p.num_per() >= 1024;
}

void launch_plane_sweep(Workspace& W, const sp_db& db, size_t pl) {
  const Params& p = *W.P;
  const size_t np_ = (size_t)db.np_local;
  const size_t plane_db_words = db_bytes(1, db.np_local, db.nj, db.packed) / 8;  // N*nj*np*{7,8} is a multiple of 8
  SweepDesc d{db.words.p + pl * plane_db_words, W.qv.p, W.sweep_out.p + pl * 4 * POLY_LEN * np_, 1, db.np_local,
              (int)p.dim0(), db.j0, db.nj, db.packed, W.out_G};
  d.nt_store = (int)tunable("sweep_nt_store", 1);
  const int wgs = (int)tunable("pipe_wgs", 4);
  if (db.packed && wgs > 0)
    launch_sweep_persist(W.D->T, d, wgs, W.stream);
  else
    launch_sweep(W.D->T, d, W.stream);
}

void run_sweep_pipelined(Workspace& W, const sp_db& db) {
  const Params& p = *W.P;
  W.ensure_sweep();
  W.ensure_finish();
  const size_t planes = p.planes();
  HIP_CHECK(hipEventRecord(W.ev_sw[0], W.stream));
  const int defer_levels = tail_defer_levels(p, tunable("pipe_tail_defer", 256));
  if (defer_levels > 0) {
    W.fold_tail.ensure(2 * planes * (p.num_per() >> defer_levels) * 2 * POLY_LEN);  // (no-op: ensure_finish sized it)
    note_path(PATH_FOLD_TAIL_BATCHED);
  }
  for (size_t pl = 0; pl < planes; pl++) {
    launch_plane_sweep(W, db, pl);
    HIP_CHECK(hipEventRecord(W.ev_plane[pl], W.stream));
    HIP_CHECK(hipStreamWaitEvent(W.stream2, W.ev_plane[pl], 0));
    if (defer_levels > 0) {
      on_stream(W, W.stream2, [&] {
        fold_plane_head(W, pl, defer_levels);
        if (pl + 1 == planes) fold_tails(W, defer_levels);
      });
      continue;
    }
    on_stream(W, W.stream2, [&] { fold_planes(W, pl, 1, false); });
  }
  HIP_CHECK(hipEventRecord(W.ev_sw[1], W.stream));
  HIP_CHECK(hipEventRecord(W.ev_fold, W.stream2));
  W.have_sweep_span = true;
  W.pipelined = true;
  note_path(PATH_PIPELINED);
}

// k_fold_fused* keep gadget digits in u32 and need digit < 2q, i.e. at most 28 bits per digit (t_gsw >= 2)
bool fused_fold_supported(const Params& p) { return 4 * p.t_gsw <= 128 && p.bits_per(p.t_gsw) <= 28; }

// fold_ciphertexts (server.rs:388-427) on `np` planes of `num_cts` raw cts each, dense in X;
// result ct of plane i ends up at the returned buffer + i*2N.
u64* run_fold(Workspace& W, u64* X, u64* Y, int np, int num_cts, int top, int d_begin, int d_end) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  const int two_t = (int)(2 * p.t_gsw);
  join_right(W);
  int further = 0;
  while (((int)1 << further) < num_cts) further++;
  // level d uses GSW ciphertext v_folding[top - d]; a whole tree has top = further - 1 (server.rs:413,420)
  const int top_idx = top < 0 ? further - 1 : top;
  // levels [d_begin, d_end) of the tree (default: all)
  if (d_end < 0) d_end = further;
  int cur = num_cts >> d_begin;
  // the workspace's threshold (read when it was created: the digit staging of the unfused levels is sized from it), lowered -- never
  // raised -- by the run-time switch fused_min_pairs_cap (in-process A/B)
  const long fused_min = std::min(W.fused_min_pairs, tunable("fused_min_pairs_cap", 1L << 62));
  for (int d = d_begin; d < d_end; d++) {
    const int half = cur / 2;
    // enough independent pairs to fill the chip: one fused workgroup per pair; otherwise (tree tail)
    // the three-kernel form, which parallelises over digits
    if (W.zero_shortcuts && !fused_fold_supported(p))
      throw ArgError("sparse buckets need gadget parameters the fused fold supports (3 <= t_gsw <= 32)");
    if (W.zero_shortcuts || ((long)np * half >= fused_min && fused_fold_supported(p))) {
      FoldDesc fd{};
      fd.zero_shortcuts = W.zero_shortcuts ? 1 : 0;
      fd.mats_w = W.mats_w_ready ? W.fold_mats_w.p + (size_t)(top_idx - d) * 2 * 2 * two_t * 2 * POLY_LEN : nullptr;
      fd.X = X;
      fd.Y = Y;
      fd.mats = W.fold_mats.p + (size_t)(top_idx - d) * 2 * 2 * two_t * 2 * POLY_LEN;
      fd.cur = cur;
      fd.half = half;
      fd.planes = np;
      fd.t = (int)p.t_gsw;
      fd.bits = (int)p.bits_per(p.t_gsw);
      // digits kd with kd * bits >= modulus_log2 are zero for coefficients below Q (t_gsw = 8: 7 live 8-bit digits of 56 bits)
      fd.t_live = W.fold_inputs_below_q && tunable("fold_skip_dead_digits", 1)
                      ? (int)std::min<size_t>(p.t_gsw, (p.modulus_log2 + fd.bits - 1) / fd.bits) : fd.t;
      launch_fold_fused(D.T, fd, s);
      std::swap(X, Y);
      cur = half;
      continue;
    }
    if (W.delta_tail) {
      // tree tail in the same delta form as k_fold_fused (half the digit transforms, C only), digit-parallel
      FwdDesc f{};
      f.src = X;
      f.dst = W.fold_dig.p;
      f.n_out = np * half * two_t;
      f.rdim = 2;
      f.cols = 1;
      f.t = (int)p.t_gsw;
      f.bits = (int)p.bits_per(p.t_gsw);
      f.src_batch_stride = 2;
      f.src_row0 = 0;
      f.src_cols = 1;
      f.delta_off = (long)half * 2;
      f.delta_inner = half;
      f.delta_outer_stride = cur;
      launch_ntt_fwd(D.T, f, s);
      MacDesc m{};
      m.A = W.fold_mats.p + ((size_t)(top_idx - d) * 2 * 2 * two_t + two_t) * 2 * POLY_LEN;  // C block of row 0
      m.A_row_stride = 2 * two_t;
      m.B = W.fold_dig.p;
      m.out = W.fold_ntt.p;
      m.R = 2;
      m.K = two_t;
      m.batch_inner = np * half;
      m.batch_outer = 1;
      m.B_inner_stride = two_t;
      m.split_k = two_t;
      m.out_batch_stride = 2;
      m.out_row_stride = 1;
      launch_mac(D.T, m, s);
      InvDesc inv{};
      inv.src = W.fold_ntt.p;
      inv.poly_stride = 2 * POLY_LEN;
      inv.crt_stride = POLY_LEN;
      inv.z_stride = 1;
      inv.dst = Y;
      inv.n_polys = np * half * 2;
      inv.addend = X;
      inv.add_inner2 = half * 2;
      inv.add_outer_stride = cur * 2;
      launch_ntt_inv(D.T, inv, s);
      note_path(PATH_FOLD_TAIL);
      std::swap(X, Y);
      cur = half;
      continue;
    }
    FwdDesc f{};
    f.src = X;
    f.dst = W.fold_dig.p;
    f.n_out = np * cur * two_t;
    f.rdim = 2;
    f.cols = 1;
    f.t = (int)p.t_gsw;
    f.bits = (int)p.bits_per(p.t_gsw);
    f.src_batch_stride = 2;
    f.src_row0 = 0;
    f.src_cols = 1;
    launch_ntt_fwd(D.T, f, s);
    MacDesc m{};
    m.A = W.fold_mats.p + (size_t)(top_idx - d) * 2 * 2 * two_t * 2 * POLY_LEN;
    m.B = W.fold_dig.p;
    m.out = W.fold_ntt.p;
    m.R = 2;
    m.K = 2 * two_t;
    m.batch_inner = half;
    m.batch_outer = np;
    m.B_inner_stride = two_t;
    m.B_outer_stride = (long)cur * two_t;
    m.split_k = two_t;
    m.split_off = (long)half * two_t;
    m.out_batch_stride = 2;
    m.out_row_stride = 1;
    launch_mac(D.T, m, s);
    InvDesc inv{};
    inv.src = W.fold_ntt.p;
    inv.poly_stride = 2 * POLY_LEN;
    inv.crt_stride = POLY_LEN;
    inv.z_stride = 1;
    inv.dst = Y;
    inv.n_polys = np * half * 2;
    launch_ntt_inv(D.T, inv, s);
    note_path(PATH_FOLD_TAIL_LITERAL);
    std::swap(X, Y);
    cur = half;
  }
  return X;
}

// packing version 1 (lib/server/src/compute/pack.rs:46-99), n == 2: prod = w_key*G^-1(ct_1) + e_1*ct_2;
// the r = 1 ciphertexts are moved down one row with w_shift; column c = prod(r=0) + shifted prod(r=1).
static void run_pack_v1(Workspace& W, const sp_pp& pp) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  const int* L = D.lists.p;
  const int tc = (int)p.t_conv;
  note_path(PATH_PACK_V1);
  const int nb = (int)p.planes();         // (inst, c, r)
  const int ne = (int)p.instances * 2;    // (inst, c)
  const size_t PW = 2 * POLY_LEN;
  W.pack_v1_P.ensure((size_t)nb * 3 * PW);
  W.pack_v1_P2.ensure((size_t)ne * 3 * PW);
  W.pack_v1_raw.ensure((size_t)ne * POLY_LEN);
  const u32* w_key = pp.all.p + (pp.off_packing + 0 * 3 * tc) * PW;
  const u32* w_shift = pp.all.p + (pp.off_packing + 1 * 3 * tc) * PW;
  FwdDesc f{};
  f.src = W.final_cts.p;
  f.src_idx = L + D.pack_src_ct;
  f.dst = W.pack_dig.p;
  f.n_out = nb * tc;
  f.rdim = 1; f.cols = 1; f.t = tc; f.bits = (int)p.bits_per(tc);
  f.src_batch_stride = 2; f.src_row0 = 0; f.src_cols = 1;
  launch_ntt_fwd(D.T, f, s);
  MacDesc m{};
  m.A = w_key; m.B = W.pack_dig.p; m.out = W.pack_v1_P.p;
  m.R = 3; m.K = tc; m.batch_inner = nb; m.batch_outer = 1; m.B_inner_stride = tc; m.split_k = tc;
  m.out_batch_stride = 3; m.out_row_stride = 1;
  launch_mac(D.T, m, s);
  FwdDesc f2 = f;
  f2.dst = W.pack_ct2.p; f2.n_out = nb; f2.t = 1; f2.bits = 64; f2.src_row0 = 1;
  launch_ntt_fwd(D.T, f2, s);
  launch_add_poly_into(D.T, W.pack_v1_P.p, L + D.v1_row1, W.pack_ct2.p, nb, s);  // add_into_at(prod, ct_2_ntt, 1, 0)
  // one shift for the r = 1 elements
  InvDesc inv{};
  inv.src = W.pack_v1_P.p; inv.idx = L + D.v1_shift_src; inv.polys_per_idx = 1;
  inv.idx_stride = (long)PW; inv.poly_stride = (long)PW; inv.crt_stride = POLY_LEN; inv.z_stride = 1;
  inv.dst = W.pack_v1_raw.p; inv.n_polys = ne;
  launch_ntt_inv(D.T, inv, s);
  FwdDesc f3{};
  f3.src = W.pack_v1_raw.p; f3.dst = W.pack_dig.p; f3.n_out = ne * tc;
  f3.rdim = 1; f3.cols = 1; f3.t = tc; f3.bits = (int)p.bits_per(tc);
  f3.src_batch_stride = 1; f3.src_row0 = 0; f3.src_cols = 1;
  launch_ntt_fwd(D.T, f3, s);
  MacDesc m2 = m;
  m2.A = w_shift; m2.out = W.pack_v1_P2.p; m2.batch_inner = ne;
  launch_mac(D.T, m2, s);
  launch_add_polys_idx(D.T, W.pack_v1_P2.p, L + D.v1_rot_dst, W.pack_v1_P2.p, L + D.v1_rot_a, W.pack_v1_P.p, L + D.v1_rot_b, 2 * ne, s);
  launch_add_polys_idx(D.T, W.pack_res.p, L + D.v1_sum_dst, W.pack_v1_P.p, L + D.v1_sum_a, W.pack_v1_P2.p, L + D.v1_sum_b, 3 * ne, s);
  InvDesc inv2{};
  inv2.src = W.pack_res.p; inv2.poly_stride = (long)PW; inv2.crt_stride = POLY_LEN; inv2.z_stride = 1;
  inv2.dst = W.pack_raw.p; inv2.n_polys = (int)(p.instances * 3 * 2);
  launch_ntt_inv(D.T, inv2, s);
}

// pack (server.rs:429-468) for all instances, then .raw() (server.rs:736) into pack_raw
void run_pack(Workspace& W, const sp_pp& pp) {
  const Params& p = *W.P;
  if (p.version == 1) {
    run_pack_v1(W, pp);
    return;
  }
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  const int* L = D.lists.p;
  const int n = (int)p.n, tc = (int)p.t_conv;
  const int nb = (int)p.planes();
  FwdDesc f{};
  f.src = W.final_cts.p;
  f.src_idx = L + D.pack_src_ct;
  f.dst = W.pack_dig.p;
  f.n_out = nb * tc;
  f.rdim = 1;
  f.cols = 1;
  f.t = tc;
  f.bits = (int)p.bits_per(tc);
  f.src_batch_stride = 2;
  f.src_row0 = 0;
  f.src_cols = 1;
  launch_ntt_fwd(D.T, f, s);
  MacDesc m{};
  m.A = pp.pack_cat.p;
  m.B = W.pack_dig.p;
  m.out = W.pack_res.p;
  m.out_idx = L + D.pack_out;
  m.R = n + 1;
  m.K = n * tc;
  m.batch_inner = (int)p.instances * n;
  m.batch_outer = 1;
  m.B_inner_stride = (long)n * tc;
  m.split_k = m.K;
  m.out_row_stride = n;
  launch_mac(D.T, m, s);
  FwdDesc f2 = f;
  f2.dst = W.pack_ct2.p;
  f2.n_out = nb;
  f2.t = 1;
  f2.bits = 64;
  f2.src_row0 = 1;
  launch_ntt_fwd(D.T, f2, s);
  launch_add_poly_into(D.T, W.pack_res.p, L + D.pack_row, W.pack_ct2.p, nb, s);
  InvDesc inv{};
  inv.src = W.pack_res.p;
  inv.poly_stride = 2 * POLY_LEN;
  inv.crt_stride = POLY_LEN;
  inv.z_stride = 1;
  inv.dst = W.pack_raw.p;
  inv.n_polys = (int)(p.instances * (p.n + 1) * p.n);
  launch_ntt_inv(D.T, inv, s);
}

// from_ntt of the first-dimension outputs + fold, for every plane (server.rs:707-711, 731)
void run_fold_all(Workspace& W, bool premod) {
  const Params& p = *W.P;
  const size_t pg = W.plane_group();
  for (size_t pg0 = 0; pg0 < p.planes(); pg0 += pg) fold_planes(W, pg0, (int)std::min(pg, p.planes() - pg0), premod);
}

// ---- encode (server.rs:470-503) on the host: rescale (arith.rs:429-444) + LSB-first bit packing
static inline u64 rescale_coeff(u64 a, u64 inp_mod, u64 out_mod) {
  typedef __int128 i128;
  const int64_t im = (int64_t)inp_mod;
  int64_t v = (int64_t)(a % inp_mod);
  if (v >= im / 2) v -= im;
  const int64_t sign = v >= 0 ? 1 : -1;
  i128 num = (i128)v * (i128)out_mod + (i128)(sign * (im / 2));
  i128 res = num / (i128)inp_mod;  // truncating, as Rust's i128 `/`
  const i128 om = (i128)out_mod;
  res = (res + (i128)((inp_mod / out_mod) * out_mod) + 2 * om) % om;
  return (u64)((res + om) % om);
}

size_t encode_response(const Params& p, const u64* packed, uint8_t* out) {
  const u64 q1 = 4 * p.pt_modulus;
  size_t q1_bits = 0;
  while (((u64)1 << q1_bits) < q1) q1_bits++;
  const u64 q2 = p.q2();
  const size_t q2_bits = p.q2_bits;
  const size_t total = p.response_bytes();
  memset(out, 0, total);
  u64* w = reinterpret_cast<u64*>(out);  // caller guarantees 8-byte alignment of `out`? no: use memcpy below
  (void)w;
  std::vector<u64> words(total / 8, 0);
  size_t bit = 0;
  auto put = [&](u64 val, size_t nbits) {
    val &= (nbits >= 64) ? ~0ULL : (((u64)1 << nbits) - 1);
    size_t wi = bit >> 6, off = bit & 63;
    words[wi] |= val << off;
    if (off + nbits > 64) words[wi + 1] |= val >> (64 - off);
    bit += nbits;
  };
  const size_t mat = (p.n + 1) * p.n * POLY_LEN;
  for (size_t inst = 0; inst < p.instances; inst++) {
    const u64* m = packed + inst * mat;
    for (size_t i = 0; i < p.n * POLY_LEN; i++) put(rescale_coeff(m[i], p.modulus, q2), q2_bits);
    const u64* rest = m + p.n * POLY_LEN;
    for (size_t i = 0; i < p.n * p.n * POLY_LEN; i++) put(rescale_coeff(rest[i], p.modulus, q1), q1_bits);
  }
  memcpy(out, words.data(), total);
  return total;
}

// ---- distributed fold (multi-GPU reduce-scatter path) ------------------------------------------------
// This rank's reduced chunk holds columns ii = g + G*i (i < num_per/G) of every plane:
// [plane][r][crt][z][i].  Folding them uses the top nu_2 - log2(G) selector bits.
// % q, from_ntt and the local fold levels of planes [pg0, pg0 + np) of a reduced chunk ([plane][r][crt][z][npl],
// `chunk` points at plane pg0) on W.stream
static void fold_local_planes(Workspace& W, const u32* chunk, int G, size_t pg0, int np) {
  const Params& p = *W.P;
  DeviceState& D = *W.D;
  hipStream_t s = W.stream;
  const int npl = (int)p.num_per() / G;
  if (npl % 4 == 0) {
    launch_from_sweep4(D.T, chunk, npl, np, 1, W.foldX.p, s);
  } else {
    InvDesc inv{};
    inv.src = chunk;
    inv.sweep_np = npl;
    inv.dst = W.foldX.p;
    inv.n_polys = np * npl * 2;
    inv.premod = 1;
    launch_ntt_inv(D.T, inv, s);
  }
  u64* res = run_fold_canonical(W, W.foldX.p, W.foldY.p, np, npl, (int)p.db_dim_2 - 1);
  HIP_CHECK(hipMemcpyAsync(W.final_cts.p + pg0 * 2 * POLY_LEN, res, (size_t)np * 2 * POLY_LEN * sizeof(u64), hipMemcpyDeviceToDevice, s));
}

void run_fold_local(Workspace& W, const u32* reduced_chunk, int G) {
  const Params& p = *W.P;
  W.ensure_finish();
  const int npl = (int)p.num_per() / G;
  const size_t pg = W.plane_group();
  for (size_t pg0 = 0; pg0 < p.planes(); pg0 += pg)
    fold_local_planes(W, reduced_chunk + pg0 * 4 * POLY_LEN * npl, G, pg0, (int)std::min(pg, p.planes() - pg0));
}

// One plane of the local fold on the SECOND stream (the caller has ordered stream2 after the plane's exchange):
// runs beside the sweeps / exchanges of the later planes.  run_fold_local_join orders the main stream after them.
void run_fold_local_plane(Workspace& W, const u32* reduced_plane_chunk, int G, int plane) {
  W.ensure_finish();
  std::swap(W.stream, W.stream2);
  try {
    fold_local_planes(W, reduced_plane_chunk, G, (size_t)plane, 1);
  } catch (...) {
    std::swap(W.stream, W.stream2);
    throw;
  }
  std::swap(W.stream, W.stream2);
}

void run_fold_local_join(Workspace& W) {
  HIP_CHECK(hipEventRecord(W.ev_fold, W.stream2));
  HIP_CHECK(hipStreamWaitEvent(W.stream, W.ev_fold, 0));
}

// gathered: [G][planes][2][N] raw cts (rank g's local results).  Leaf g of the remaining tree is rank g.
void run_finish_gathered(Workspace& W, const sp_pp& pp, const u64* gathered, int G) {
  const Params& p = *W.P;
  hipStream_t s = W.stream;
  W.ensure_finish();
  const size_t ctw = 2 * POLY_LEN;
  const int planes = (int)p.planes();
  W.foldX.ensure((size_t)planes * G * ctw);
  W.foldY.ensure((size_t)planes * std::max(G / 2, 1) * ctw);
  // [g][plane] -> [plane][g]
  HIP_CHECK(hipMemcpy2DAsync(W.foldX.p, (size_t)G * ctw * 8, gathered, ctw * 8, ctw * 8, (size_t)planes, hipMemcpyDeviceToDevice, s));
  for (int g = 1; g < G; g++)
    HIP_CHECK(hipMemcpy2DAsync(W.foldX.p + (size_t)g * ctw, (size_t)G * ctw * 8, gathered + (size_t)g * planes * ctw, ctw * 8,
                               ctw * 8, (size_t)planes, hipMemcpyDeviceToDevice, s));
  int lg = 0;
  while ((1 << lg) < G) lg++;
  u64* res = run_fold(W, W.foldX.p, W.foldY.p, planes, G, lg - 1);
  HIP_CHECK(hipMemcpyAsync(W.final_cts.p, res, (size_t)planes * ctw * sizeof(u64), hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipEventRecord(W.ev[3], s));
  run_pack(W, pp);
  run_encode_device(W);
  HIP_CHECK(hipEventRecord(W.ev[4], s));
}

void run_encode_device(Workspace& W) {
  const Params& p = *W.P;
  hipStream_t s = W.stream;
  HIP_CHECK(hipMemsetAsync(W.enc_out.p, 0, W.enc_out.bytes(), s));
  EncodeDesc e{};
  e.packed = W.pack_raw.p;
  e.out = reinterpret_cast<unsigned long long*>(W.enc_out.p);
  e.instances = (int)p.instances;
  e.n = (int)p.n;
  e.Q = p.modulus;
  e.q1 = 4 * p.pt_modulus;
  e.q2 = p.q2();
  int q1_bits = 0;
  while (((u64)1 << q1_bits) < e.q1) q1_bits++;
  e.q1_bits = q1_bits;
  e.q2_bits = (int)p.q2_bits;
  launch_encode(e, s);
  HIP_CHECK(hipMemcpyAsync(W.h_response, W.enc_out.p, p.response_bytes(), hipMemcpyDeviceToHost, s));
}

void run_finish(Workspace& W, const sp_pp& pp, bool premod) {
  const Params& p = *W.P;
  W.ensure_finish();
  if (W.pipelined) {  // folds were issued by run_sweep_pipelined on stream2
    HIP_CHECK(hipStreamWaitEvent(W.stream, W.ev_fold, 0));
    W.pipelined = false;
  } else {
    run_fold_all(W, premod);
  }
  HIP_CHECK(hipEventRecord(W.ev[3], W.stream));
  run_pack(W, pp);
  run_encode_device(W);
  HIP_CHECK(hipEventRecord(W.ev[4], W.stream));
  (void)p;
}

}  // namespace spiral
