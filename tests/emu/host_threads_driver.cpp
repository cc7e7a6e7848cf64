// TEST INFRASTRUCTURE: several HOST threads through the C ABI at once -- what an actix worker pool does to the library
// (lib/server/src/bin/server.rs: the database behind a read lock, one query per request), linked against the emulated library
// (plain or AddressSanitizer build).  No Python in the process.
//   host_threads_driver params.json pp.bin query.bin db.bin expected.bin [threads] [queries per thread]
// Thread 0 additionally answers a LIST of queries (sp_process_query_batch); every response must equal expected.bin.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "spiral_hip.h"

static std::vector<unsigned char> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path);
    exit(2);
  }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> b((size_t)n + 1);
  if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
  fclose(f);
  b[(size_t)n] = 0;
  b.resize((size_t)n);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  std::vector<unsigned char> json = slurp(argv[1]), ppb = slurp(argv[2]), qb = slurp(argv[3]), dbw = slurp(argv[4]), want = slurp(argv[5]);
  json.push_back(0);
  const int n_threads = argc > 6 ? atoi(argv[6]) : 4, per_thread = argc > 7 ? atoi(argv[7]) : 2;
  sp_params_t* p = sp_params_from_json((const char*)json.data());
  if (!p) return 3;
  const size_t rb = (size_t)sp_params_get(p, "response_bytes");
  sp_db_t* db = sp_db_create(p, 0, 1);
  if (!db || sp_db_load(db, (const uint64_t*)dbw.data(), dbw.size() / 8) != SP_OK) return 3;
  sp_pp_t* pp = sp_pp_deserialize(p, ppb.data(), ppb.size());
  if (!pp) return 3;
  std::atomic<int> bad{0};
  std::vector<std::thread> ts;
  for (int t = 0; t < n_threads; t++)
    ts.emplace_back([&, t] {
      std::vector<unsigned char> out(rb);
      for (int k = 0; k < per_thread; k++) {
        size_t len = 0;
        if (sp_process_query(p, pp, qb.data(), qb.size(), db, out.data(), rb, &len) != SP_OK || len != want.size() ||
            memcmp(out.data(), want.data(), len) != 0) {
          fprintf(stderr, "thread %d query %d: %s\n", t, k, sp_last_error());
          bad++;
        }
      }
      if (t == 0) {  // a list beside the single queries of the other threads
        const int B = 3;
        const sp_pp_t* pps[B] = {pp, pp, pp};
        const uint8_t* qs[B] = {qb.data(), qb.data(), qb.data()};
        size_t lens[B] = {qb.size(), qb.size(), qb.size()};
        std::vector<unsigned char> outs((size_t)B * rb);
        size_t len = 0;
        if (sp_process_query_batch(p, pps, qs, lens, B, db, outs.data(), rb, &len) != SP_OK) {
          fprintf(stderr, "list: %s\n", sp_last_error());
          bad++;
        } else {
          for (int i = 0; i < B; i++)
            if (memcmp(outs.data() + (size_t)i * rb, want.data(), want.size()) != 0) bad++;
        }
      }
    });
  for (auto& t : ts) t.join();
  sp_pp_free(pp);
  sp_db_free(db);
  sp_params_free(p);
  printf("%d threads x %d queries + a list of 3: %s\n", n_threads, per_thread, bad ? "MISMATCH" : "all equal to the oracle's");
  return bad ? 1 : 0;
}
