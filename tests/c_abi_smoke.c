/* tests/c_abi_smoke.c -- one query through the C ABI of libspiral_hip.so with no Python in the loop.
 *
 * Built with plain gcc against include/spiral_hip.h and driven by tests/test_c_abi_smoke.py, which writes the
 * fixture files (params JSON, serialized public parameters, serialized query, reference-layout database words and
 * the oracle's response) and checks the exit code.  This is the call sequence a C / Rust host (lib/server) makes:
 *   sp_params_from_json -> sp_db_create + sp_db_load -> sp_pp_deserialize -> sp_process_query
 * and, with argv[1] = "sharded", the same query through sp_comm_create (RCCL, world size 1) +
 * sp_process_query_sharded.  "host" mode only exercises the entry points that need no device.
 *
 *   c_abi_smoke host    params.json
 *   c_abi_smoke query   params.json pp.bin query.bin db.bin expected.bin
 *   c_abi_smoke sharded params.json pp.bin query.bin db.bin expected.bin
 * exit 0 = response byte-identical to expected.bin.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spiral_hip.h"

static unsigned char* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path);
    exit(2);
  }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char* buf = (unsigned char*)malloc((size_t)n + 1);
  if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) {
    fprintf(stderr, "cannot read %s\n", path);
    exit(2);
  }
  buf[n] = 0;
  fclose(f);
  *len = (size_t)n;
  return buf;
}

#define CHECK(call)                                                         \
  do {                                                                      \
    int rc_ = (call);                                                       \
    if (rc_ != SP_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sp_last_error());       \
      return 3;                                                             \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s host|query|sharded params.json [pp.bin query.bin db.bin expected.bin]\n", argv[0]);
    return 2;
  }
  size_t n_json = 0;
  char* json = (char*)slurp(argv[2], &n_json);
  sp_params_t* p = sp_params_from_json(json);
  if (!p) {
    fprintf(stderr, "sp_params_from_json: %s\n", sp_last_error());
    return 3;
  }
  const size_t resp_bytes = (size_t)sp_params_get(p, "response_bytes");
  printf("setup_bytes=%llu query_bytes=%llu response_bytes=%zu db_words=%llu devices=%d\n",
         (unsigned long long)sp_params_get(p, "setup_bytes"), (unsigned long long)sp_params_get(p, "query_bytes"), resp_bytes,
         (unsigned long long)sp_params_get(p, "db_words"), sp_device_count());
  if (strcmp(argv[1], "host") == 0) {
    /* no device needed: sizes, tables, the synthetic-word hash; a compute call must fail loudly without a GPU */
    uint64_t tab[2048];
    CHECK(sp_params_ntt_table(p, 0, 0, tab));
    if (sp_params_get(p, "no_such_field") != UINT64_MAX || sp_synth_word(1, 2) == sp_synth_word(1, 3)) return 4;
    if (sp_path_name(0) == NULL || sp_path_name(1000) != NULL) return 4;
    sp_params_free(p);
    puts("host-ok");
    return 0;
  }
  if (argc < 7) return 2;
  size_t n_pp, n_q, n_db, n_exp;
  unsigned char* pp_bytes = slurp(argv[3], &n_pp);
  unsigned char* q_bytes = slurp(argv[4], &n_q);
  unsigned char* db_words = slurp(argv[5], &n_db);
  unsigned char* expected = slurp(argv[6], &n_exp);
  if (sp_device_count() < 1) {
    fprintf(stderr, "no HIP device\n");
    return 5;
  }
  CHECK(sp_set_device(0));
  sp_db_t* db = sp_db_create(p, 0, 1);
  if (!db) {
    fprintf(stderr, "sp_db_create: %s\n", sp_last_error());
    return 3;
  }
  CHECK(sp_db_load(db, (const uint64_t*)db_words, n_db / 8));
  sp_pp_t* pp = sp_pp_deserialize(p, pp_bytes, n_pp);
  if (!pp) {
    fprintf(stderr, "sp_pp_deserialize: %s\n", sp_last_error());
    return 3;
  }
  unsigned char* out = (unsigned char*)calloc(resp_bytes, 1);
  size_t out_len = 0;
  if (strcmp(argv[1], "sharded") == 0) {
    uint8_t id[SP_COMM_ID_BYTES];
    CHECK(sp_comm_unique_id(id));
    sp_comm_t* comm = sp_comm_create(0, 1, id);
    if (!comm) {
      fprintf(stderr, "sp_comm_create: %s\n", sp_last_error());
      return 3;
    }
    CHECK(sp_process_query_sharded(comm, p, pp, q_bytes, n_q, db, out, resp_bytes, &out_len));
    CHECK(sp_comm_barrier(comm));
    sp_comm_free(comm);
  } else {
    CHECK(sp_process_query(p, pp, q_bytes, n_q, db, out, resp_bytes, &out_len));
    /* a wrong query length is an argument error, not a crash (client.rs:304 asserts in the reference) */
    size_t dummy = 0;
    if (sp_process_query(p, pp, q_bytes, n_q - 8, db, out + 0, resp_bytes, &dummy) != SP_E_ARG) return 6;
    CHECK(sp_process_query(p, pp, q_bytes, n_q, db, out, resp_bytes, &out_len));
  }
  uint64_t paths = sp_paths_taken(1);
  printf("paths:");
  for (int b = 0; sp_path_name(b); b++)
    if (paths >> b & 1) printf(" %s", sp_path_name(b));
  printf("\n");
  const int same = out_len == n_exp && memcmp(out, expected, n_exp) == 0;
  printf("%s: %zu response bytes %s the oracle's\n", argv[1], out_len, same ? "==" : "!=");
  sp_pp_free(pp);
  sp_db_free(db);
  sp_params_free(p);
  return same ? 0 : 1;
}
