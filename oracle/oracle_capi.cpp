// ============================================================================
// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY (see spiral_oracle.h).
// Flat extern "C" surface over the CPU restatement so pytest / bench.py's
// cpu_baseline leg can drive it through ctypes.  All polynomial arrays use the
// reference's own layouts (PolyMatrixRaw: rows*cols*N u64; PolyMatrixNTT:
// rows*cols*crt*N u64, poly.rs:31-35, 263-265).
// ============================================================================
#include <chrono>
#include <cstring>
#include <string>

#include "spiral_oracle.h"

using namespace oracle;

static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH(ret)              \
  }                                 \
  catch (const std::exception& e) { \
    g_err = e.what();               \
    return ret;                     \
  }

static PolyMatrixNTT ntt_from_flat(const Params* p, size_t rows, size_t cols, const u64* src) {
  PolyMatrixNTT m(p, rows, cols);
  memcpy(m.data.data(), src, m.data.size() * 8);
  return m;
}
static PolyMatrixRaw raw_from_flat(const Params* p, size_t rows, size_t cols, const u64* src) {
  PolyMatrixRaw m(p, rows, cols);
  memcpy(m.data.data(), src, m.data.size() * 8);
  return m;
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ------------------------------------------------------------------ params
void* orc_params_new(uint64_t n, uint64_t nu_1, uint64_t nu_2, uint64_t p, uint64_t q2_bits, uint64_t t_gsw,
                     uint64_t t_conv, uint64_t t_exp_left, uint64_t t_exp_right, uint64_t instances,
                     uint64_t db_item_size, uint64_t version, int direct_upload) {
  ORC_TRY
  return new Params(params_from_fields(n, nu_1, nu_2, p, q2_bits, t_gsw, t_conv, t_exp_left, t_exp_right, instances,
                                       db_item_size, version, direct_upload != 0));
  ORC_CATCH(nullptr)
}
// Params::init with explicit moduli (used for the q2 single-modulus params and the KATs)
void* orc_params_init(uint64_t poly_len, const uint64_t* moduli, uint64_t crt_count, uint64_t n, uint64_t p,
                      uint64_t q2_bits, uint64_t t_conv, uint64_t t_exp_left, uint64_t t_exp_right, uint64_t t_gsw,
                      int expand_queries, uint64_t nu_1, uint64_t nu_2, uint64_t instances, uint64_t db_item_size,
                      uint64_t version) {
  ORC_TRY
  std::vector<u64> m(moduli, moduli + crt_count);
  return new Params(Params::init(poly_len, m, 6.4, n, p, q2_bits, t_conv, t_exp_left, t_exp_right, t_gsw,
                                 expand_queries != 0, nu_1, nu_2, instances, db_item_size, version));
  ORC_CATCH(nullptr)
}
void orc_params_free(void* h) { delete (Params*)h; }

uint64_t orc_params_get(void* h, const char* name) {
  const Params& p = *(Params*)h;
  std::string s(name);
  if (s == "poly_len") return p.poly_len;
  if (s == "poly_len_log2") return p.poly_len_log2;
  if (s == "crt_count") return p.crt_count;
  if (s == "modulus") return p.modulus;
  if (s == "modulus_log2") return p.modulus_log2;
  if (s == "moduli0") return p.moduli[0];
  if (s == "moduli1") return p.moduli[1];
  if (s == "barrett_cr_0_0") return p.barrett_cr_0[0];
  if (s == "barrett_cr_0_1") return p.barrett_cr_0[1];
  if (s == "barrett_cr_1_0") return p.barrett_cr_1[0];
  if (s == "barrett_cr_1_1") return p.barrett_cr_1[1];
  if (s == "barrett_cr_0_modulus") return p.barrett_cr_0_modulus;
  if (s == "barrett_cr_1_modulus") return p.barrett_cr_1_modulus;
  if (s == "mod0_inv_mod1") return p.mod0_inv_mod1;
  if (s == "mod1_inv_mod0") return p.mod1_inv_mod0;
  if (s == "n") return p.n;
  if (s == "pt_modulus") return p.pt_modulus;
  if (s == "q2_bits") return p.q2_bits;
  if (s == "t_conv") return p.t_conv;
  if (s == "t_exp_left") return p.t_exp_left;
  if (s == "t_exp_right") return p.t_exp_right;
  if (s == "t_gsw") return p.t_gsw;
  if (s == "expand_queries") return p.expand_queries;
  if (s == "db_dim_1") return p.db_dim_1;
  if (s == "db_dim_2") return p.db_dim_2;
  if (s == "instances") return p.instances;
  if (s == "db_item_size") return p.db_item_size;
  if (s == "version") return p.version;
  if (s == "g") return p.g();
  if (s == "stop_round") return p.stop_round();
  if (s == "setup_bytes") return p.setup_bytes();
  if (s == "query_bytes") return p.query_bytes();
  if (s == "num_items") return p.num_items();
  if (s == "bytes_per_chunk") return p.bytes_per_chunk();
  if (s == "modp_words_per_chunk") return p.modp_words_per_chunk();
  return ~0ULL;
}
void orc_ntt_table(void* h, uint64_t crt, uint64_t which, uint64_t* out) {
  const Params& p = *(Params*)h;
  memcpy(out, p.ntt_tables[crt][which].data(), p.poly_len * 8);
}

// ------------------------------------------------------------------- arith
void orc_get_barrett_crs(uint64_t modulus, uint64_t* out2) { get_barrett_crs(modulus, &out2[0], &out2[1]); }
void orc_divide_uint192(const uint64_t* num3, uint64_t den, uint64_t* rem3, uint64_t* quot3) {
  divide_uint192_inplace(num3, den, rem3, quot3);
}
uint64_t orc_div2_uint_mod(uint64_t a, uint64_t m) { return div2_uint_mod(a, m); }
uint64_t orc_barrett_raw_u64(uint64_t x, uint64_t cr1, uint64_t m) { return barrett_raw_u64(x, cr1, m); }
uint64_t orc_barrett_reduction_u128_raw(uint64_t m, uint64_t cr0, uint64_t cr1, uint64_t lo, uint64_t hi) {
  return barrett_reduction_u128_raw(m, cr0, cr1, ((u128)hi << 64) | lo);
}
uint64_t orc_rescale(uint64_t a, uint64_t inp, uint64_t outm) { return rescale(a, inp, outm); }
uint64_t orc_recenter_mod(uint64_t v, uint64_t s, uint64_t l) { return recenter_mod(v, s, l); }
uint64_t orc_recenter(uint64_t v, uint64_t f, uint64_t t) { return recenter(v, f, t); }
uint64_t orc_reverse_bits(uint64_t x, uint64_t bits) { return reverse_bits(x, bits); }
uint64_t orc_exponentiate_uint_mod(uint64_t a, uint64_t e, uint64_t m) { return exponentiate_uint_mod(a, e, m); }
uint64_t orc_invert_uint_mod(uint64_t a, uint64_t m) { return invert_uint_mod(a, m); }
uint64_t orc_get_minimal_primitive_root(uint64_t degree, uint64_t m) { return get_minimal_primitive_root(degree, m); }
uint64_t orc_crt_compose_2(void* h, uint64_t x, uint64_t y) { return ((Params*)h)->crt_compose_2(x, y); }
uint64_t orc_calc_index(const uint64_t* ind, const uint64_t* len, uint64_t n) {
  return calc_index((const size_t*)ind, (const size_t*)len, n);
}
uint64_t orc_read_arbitrary_bits(const uint8_t* d, uint64_t off, uint64_t nb) { return read_arbitrary_bits(d, off, nb); }
void orc_write_arbitrary_bits(uint8_t* d, uint64_t v, uint64_t off, uint64_t nb) { write_arbitrary_bits(d, v, off, nb); }

// ---------------------------------------------------------------- ntt/poly
void orc_ntt_forward(void* h, uint64_t* data, uint64_t count) {
  const Params& p = *(Params*)h;
  for (uint64_t i = 0; i < count; i++) ntt_forward(p, data + i * p.crt_count * p.poly_len);
}
void orc_ntt_inverse(void* h, uint64_t* data, uint64_t count) {
  const Params& p = *(Params*)h;
  for (uint64_t i = 0; i < count; i++) ntt_inverse(p, data + i * p.crt_count * p.poly_len);
}
// raw[count][N] -> ntt[count][crt*N]
void orc_to_ntt(void* h, const uint64_t* raw, uint64_t* out, uint64_t count, int no_reduce) {
  const Params* p = (Params*)h;
  PolyMatrixRaw r = raw_from_flat(p, count, 1, raw);
  PolyMatrixNTT o(p, count, 1);
  if (no_reduce)
    to_ntt_no_reduce(o, r);
  else
    to_ntt(o, r);
  memcpy(out, o.data.data(), o.data.size() * 8);
}
void orc_from_ntt(void* h, const uint64_t* ntt, uint64_t* out, uint64_t count) {
  const Params* p = (Params*)h;
  PolyMatrixNTT m = ntt_from_flat(p, count, 1, ntt);
  PolyMatrixRaw o(p, count, 1);
  from_ntt(o, m);
  memcpy(out, o.data.data(), o.data.size() * 8);
}
// res[ar x bc] = a[ar x ac] * b[ac x bc]
void orc_multiply(void* h, const uint64_t* a, uint64_t ar, uint64_t ac, const uint64_t* b, uint64_t bc, uint64_t* res) {
  const Params* p = (Params*)h;
  PolyMatrixNTT A = ntt_from_flat(p, ar, ac, a), B = ntt_from_flat(p, ac, bc, b), R(p, ar, bc);
  multiply(R, A, B);
  memcpy(res, R.data.data(), R.data.size() * 8);
}
void orc_automorph(void* h, const uint64_t* a, uint64_t count, uint64_t t, uint64_t* res) {
  const Params* p = (Params*)h;
  PolyMatrixRaw A = raw_from_flat(p, count, 1, a), R(p, count, 1);
  automorph(R, A, t);
  memcpy(res, R.data.data(), R.data.size() * 8);
}
// inp[rows_in x cols] -> out[rows_out x cols]
void orc_gadget_invert_rdim(void* h, const uint64_t* inp, uint64_t rows_in, uint64_t cols, uint64_t* out,
                            uint64_t rows_out, uint64_t rdim) {
  const Params* p = (Params*)h;
  PolyMatrixRaw I = raw_from_flat(p, rows_in, cols, inp), O(p, rows_out, cols);
  gadget_invert_rdim(O, I, rdim);
  memcpy(out, O.data.data(), O.data.size() * 8);
}
void orc_build_gadget(void* h, uint64_t rows, uint64_t cols, uint64_t* out) {
  PolyMatrixRaw g = build_gadget(*(Params*)h, rows, cols);
  memcpy(out, g.data.data(), g.data.size() * 8);
}
uint64_t orc_get_bits_per(void* h, uint64_t dim) { return get_bits_per(*(Params*)h, dim); }

// ----------------------------------------------------------------- chacha
void orc_chacha20_block(const uint32_t* in16, uint32_t* out16) { chacha20_block(in16, out16); }
void orc_chacha20_rng_u64(const uint8_t* seed32, uint64_t* out, uint64_t count) {
  ChaCha20Rng rng(seed32);
  for (uint64_t i = 0; i < count; i++) out[i] = rng.next_u64();
}

// ----------------------------------------------------------------- client
void* orc_client_new(void* h) { return new Client((Params*)h); }
void orc_client_free(void* c) { delete (Client*)c; }
int64_t orc_client_generate_keys(void* c, const uint8_t* seed32, uint8_t* out, uint64_t cap) {
  ORC_TRY
  std::vector<uint8_t> b = ((Client*)c)->generate_keys(seed32).serialize();
  if (b.size() > cap) return -(int64_t)b.size();
  memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
  ORC_CATCH(-1)
}
int64_t orc_client_generate_query(void* c, uint64_t idx, const uint8_t* seed32, uint8_t* out, uint64_t cap) {
  ORC_TRY
  std::vector<uint8_t> b = ((Client*)c)->generate_query(idx, seed32).serialize();
  if (b.size() > cap) return -(int64_t)b.size();
  memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
  ORC_CATCH(-1)
}
int64_t orc_client_decode_response(void* c, const uint8_t* data, uint64_t len, uint8_t* out, uint64_t cap) {
  ORC_TRY
  std::vector<uint8_t> b = ((Client*)c)->decode_response(data, len);
  if (b.size() > cap) return -(int64_t)b.size();
  memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
  ORC_CATCH(-1)
}
// decrypt `count` NTT-form 2x1 Regev cts -> raw polys (client.rs:474-476 then .raw())
void orc_client_decrypt_reg(void* c, const uint64_t* cts_ntt, uint64_t count, uint64_t* out_raw) {
  Client* cl = (Client*)c;
  const Params* p = cl->params;
  for (uint64_t i = 0; i < count; i++) {
    PolyMatrixNTT ct = ntt_from_flat(p, 2, 1, cts_ntt + i * 2 * p->crt_count * p->poly_len);
    PolyMatrixRaw dec = from_ntt_alloc(cl->decrypt_matrix_reg(ct));
    memcpy(out_raw + i * p->poly_len, dec.data.data(), p->poly_len * 8);
  }
}
// encrypt a plaintext raw poly (1x1) as a fresh 2x1 NTT-form Regev ct (tests of stage functions)
void orc_client_encrypt_reg(void* c, const uint64_t* pt_raw, const uint8_t* seed32, const uint8_t* seed_pub32,
                            uint64_t* out_ct_ntt) {
  Client* cl = (Client*)c;
  const Params* p = cl->params;
  ChaCha20Rng rng(seed32), rng_pub(seed_pub32);
  PolyMatrixNTT ct = cl->encrypt_matrix_reg(to_ntt_alloc(raw_from_flat(p, 1, 1, pt_raw)), rng, rng_pub);
  memcpy(out_ct_ntt, ct.data.data(), ct.data.size() * 8);
}

// ----------------------------------------------------------------- server
// db must hold instances*n*n*num_items*N u64; item_out holds instances*n*n*N u64 (PolyMatrixRaw (inst*n) x n)
int orc_generate_random_db_and_get_item(void* h, uint64_t item_idx, uint64_t seed, uint64_t* db, uint64_t* item_out) {
  ORC_TRY
  const Params& p = *(Params*)h;
  PolyMatrixRaw item;
  std::vector<u64> v;
  generate_random_db_and_get_item(p, item_idx, seed, item, v);
  memcpy(db, v.data(), v.size() * 8);
  memcpy(item_out, item.data.data(), item.data.size() * 8);
  return 0;
  ORC_CATCH(-1)
}
int orc_load_db_from_bytes(void* h, const uint8_t* file, uint64_t len, uint64_t* db) {
  ORC_TRY
  std::vector<u64> v;
  load_db_from_bytes(*(Params*)h, file, len, v);
  memcpy(db, v.data(), v.size() * 8);
  return 0;
  ORC_CATCH(-1)
}
// corr_item.to_vec(p_bits, modp_words_per_chunk) -- what decode_response is compared with (server.rs:1034-1042)
int64_t orc_item_to_vec(void* h, const uint64_t* item, uint8_t* out, uint64_t cap) {
  ORC_TRY
  const Params* p = (Params*)h;
  PolyMatrixRaw m = raw_from_flat(p, p->instances * p->n, p->n, item);
  std::vector<uint8_t> b = m.to_vec((size_t)log2_ceil(p->pt_modulus), p->modp_words_per_chunk());
  if (b.size() > cap) return -(int64_t)b.size();
  memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
  ORC_CATCH(-1)
}

int64_t orc_process_query(void* h, const uint8_t* pp_bytes, uint64_t pp_len, const uint8_t* q_bytes, uint64_t q_len,
                          const uint64_t* db, uint8_t* out, uint64_t cap) {
  ORC_TRY
  const Params& p = *(Params*)h;
  PublicParameters pp = PublicParameters::deserialize(p, pp_bytes, pp_len);
  Query q = Query::deserialize(p, q_bytes, q_len);
  std::vector<uint8_t> r = process_query(p, pp, q, db);
  if (r.size() > cap) return -(int64_t)r.size();
  memcpy(out, r.data(), r.size());
  return (int64_t)r.size();
  ORC_CATCH(-1)
}

// Deserialised public parameters in NTT form, flattened in wire order:
// v_packing[n] ((n+1) x t_conv), v_expansion_left[g] (2 x t_exp_left),
// v_expansion_right[stop_round+1] (2 x t_exp_right), v_conversion[1] (2 x 2t_conv).
int64_t orc_pp_deserialize_flat(void* h, const uint8_t* pp_bytes, uint64_t pp_len, uint64_t* out, uint64_t cap_words) {
  ORC_TRY
  const Params& p = *(Params*)h;
  PublicParameters pp = PublicParameters::deserialize(p, pp_bytes, pp_len);
  std::vector<u64> flat;
  auto app = [&](const std::vector<PolyMatrixNTT>& v) {
    for (auto& m : v) flat.insert(flat.end(), m.data.begin(), m.data.end());
  };
  app(pp.v_packing);
  app(pp.v_expansion_left);
  if (p.expand_queries && (p.version == 0 || p.t_exp_right != p.t_exp_left)) app(pp.v_expansion_right);
  app(pp.v_conversion);
  if (flat.size() > cap_words) return -(int64_t)flat.size();
  memcpy(out, flat.data(), flat.size() * 8);
  return (int64_t)flat.size();
  ORC_CATCH(-1)
}
// Query::deserialize -> raw 2x1 ct (2N words)
int orc_query_deserialize_ct(void* h, const uint8_t* q_bytes, uint64_t q_len, uint64_t* out_raw) {
  ORC_TRY
  const Params& p = *(Params*)h;
  Query q = Query::deserialize(p, q_bytes, q_len);
  memcpy(out_raw, q.ct.data.data(), q.ct.data.size() * 8);
  return 0;
  ORC_CATCH(-1)
}

// expand_query: v_reg_out[N*dim0*2], v_folding_out[nu_2][2 x 2t_gsw NTT]
int orc_expand_query(void* h, const uint8_t* pp_bytes, uint64_t pp_len, const uint8_t* q_bytes, uint64_t q_len,
                     uint64_t* v_reg_out, uint64_t* v_folding_out) {
  ORC_TRY
  const Params& p = *(Params*)h;
  PublicParameters pp = PublicParameters::deserialize(p, pp_bytes, pp_len);
  Query q = Query::deserialize(p, q_bytes, q_len);
  std::vector<u64> v_reg;
  std::vector<PolyMatrixNTT> v_folding;
  expand_query(p, pp, q, v_reg, v_folding);
  memcpy(v_reg_out, v_reg.data(), v_reg.size() * 8);
  size_t off = 0;
  for (auto& m : v_folding) {
    memcpy(v_folding_out + off, m.data.data(), m.data.size() * 8);
    off += m.data.size();
  }
  return 0;
  ORC_CATCH(-1)
}

// coefficient_expansion on v[2^g] 2x1 NTT cts (in place), pp from bytes
int orc_coefficient_expansion(void* h, const uint8_t* pp_bytes, uint64_t pp_len, uint64_t* v, uint64_t g,
                              uint64_t stop_round, uint64_t max_bits_to_gen_right) {
  ORC_TRY
  const Params& p = *(Params*)h;
  PublicParameters pp = PublicParameters::deserialize(p, pp_bytes, pp_len);
  size_t words = 2 * p.crt_count * p.poly_len;
  std::vector<PolyMatrixNTT> vv;
  for (size_t i = 0; i < ((size_t)1 << g); i++) vv.push_back(ntt_from_flat(&p, 2, 1, v + i * words));
  std::vector<PolyMatrixNTT> v_neg1 = get_v_neg1(p);
  const std::vector<PolyMatrixNTT>& right = pp.has_expansion_right ? pp.v_expansion_right : pp.v_expansion_left;
  coefficient_expansion(vv, g, stop_round, p, pp.v_expansion_left, stop_round == 0 ? pp.v_expansion_left : right, v_neg1,
                        max_bits_to_gen_right);
  for (size_t i = 0; i < vv.size(); i++) memcpy(v + i * words, vv[i].data.data(), words * 8);
  return 0;
  ORC_CATCH(-1)
}

// regev_to_gsw: v_inp[count_inp] 2x1 NTT; v_conv 2 x 2t_conv NTT; out v_gsw[num_gsw] 2 x 2t_gsw NTT
int orc_regev_to_gsw(void* h, const uint64_t* v_inp, uint64_t count_inp, const uint64_t* v_conv, uint64_t* v_gsw_out,
                     uint64_t num_gsw) {
  ORC_TRY
  const Params& p = *(Params*)h;
  size_t words = 2 * p.crt_count * p.poly_len;
  std::vector<PolyMatrixNTT> inp;
  for (size_t i = 0; i < count_inp; i++) inp.push_back(ntt_from_flat(&p, 2, 1, v_inp + i * words));
  PolyMatrixNTT V = ntt_from_flat(&p, 2, 2 * p.t_conv, v_conv);
  std::vector<PolyMatrixNTT> gsw;
  for (size_t i = 0; i < num_gsw; i++) gsw.emplace_back(&p, 2, 2 * p.t_gsw);
  regev_to_gsw(gsw, inp, V, p, 1, 0);
  size_t off = 0;
  for (auto& m : gsw) {
    memcpy(v_gsw_out + off, m.data.data(), m.data.size() * 8);
    off += m.data.size();
  }
  return 0;
  ORC_CATCH(-1)
}

int orc_get_v_folding_neg(void* h, const uint64_t* v_folding, uint64_t* out) {
  ORC_TRY
  const Params& p = *(Params*)h;
  size_t words = 2 * 2 * p.t_gsw * p.crt_count * p.poly_len;
  std::vector<PolyMatrixNTT> vf;
  for (size_t i = 0; i < p.db_dim_2; i++) vf.push_back(ntt_from_flat(&p, 2, 2 * p.t_gsw, v_folding + i * words));
  std::vector<PolyMatrixNTT> neg = get_v_folding_neg(p, vf);
  for (size_t i = 0; i < neg.size(); i++) memcpy(out + i * words, neg[i].data.data(), words * 8);
  return 0;
  ORC_CATCH(-1)
}

// out[num_per] 2x1 NTT cts
int orc_multiply_reg_by_database(void* h, const uint64_t* db, const uint64_t* v_firstdim, uint64_t dim0, uint64_t num_per,
                                 uint64_t* out) {
  ORC_TRY
  const Params& p = *(Params*)h;
  size_t words = 2 * p.crt_count * p.poly_len;
  std::vector<PolyMatrixNTT> o;
  for (size_t i = 0; i < num_per; i++) o.emplace_back(&p, 2, 1);
  multiply_reg_by_database(o, db, v_firstdim, p, dim0, num_per);
  for (size_t i = 0; i < num_per; i++) memcpy(out + i * words, o[i].data.data(), words * 8);
  return 0;
  ORC_CATCH(-1)
}
// Only rows z in [z0, z1) of the sweep, for the bounded cpu_baseline sample and sampled parity at
// full size: out[(z - z0) * num_per * 4 + ii*4 + {n0_0, n0_1, n1_0, n1_1}].  db_zslice points at
// the reference-layout words of rows z0.. (i.e. db + z0*num_per*dim0), v_firstdim likewise at z0.
void orc_sweep_rows(const uint64_t* db_zslice, const uint64_t* v_firstdim_zslice, uint64_t nz, uint64_t dim0,
                    uint64_t num_per, uint64_t q0, uint64_t q1, uint64_t* out) {
  for (uint64_t z = 0; z < nz; z++)
    for (uint64_t i = 0; i < num_per; i++) {
      u128 s00 = 0, s01 = 0, s10 = 0, s11 = 0;
      const u64* b = db_zslice + (z * num_per + i) * dim0;
      const u64* a = v_firstdim_zslice + z * dim0 * 2;
      for (uint64_t j = 0; j < dim0; j++) {
        u64 bw = b[j], a0 = a[2 * j], a1 = a[2 * j + 1];
        s00 += (u128)((u64)(u32)a0 * (u64)(u32)bw);
        s01 += (u128)((u64)(u32)a1 * (u64)(u32)bw);
        s10 += (u128)((a0 >> 32) * (bw >> 32));
        s11 += (u128)((a1 >> 32) * (bw >> 32));
      }
      u64* o = out + (z * num_per + i) * 4;
      o[0] = (u64)(s00 % q0);
      o[1] = (u64)(s01 % q0);
      o[2] = (u64)(s10 % q1);
      o[3] = (u64)(s11 % q1);
    }
}

// fold: cts[num_per] raw 2x1 (in/out; result in cts[0]); v_folding / v_folding_neg [nu] 2 x 2t_gsw NTT
int orc_fold_ciphertexts(void* h, uint64_t* cts, uint64_t num_per, const uint64_t* v_folding, const uint64_t* v_folding_neg,
                         uint64_t nu) {
  ORC_TRY
  const Params& p = *(Params*)h;
  size_t rw = 2 * p.poly_len;
  size_t words = 2 * 2 * p.t_gsw * p.crt_count * p.poly_len;
  std::vector<PolyMatrixRaw> v;
  for (size_t i = 0; i < num_per; i++) v.push_back(raw_from_flat(&p, 2, 1, cts + i * rw));
  std::vector<PolyMatrixNTT> vf, vfn;
  for (size_t i = 0; i < nu; i++) {
    vf.push_back(ntt_from_flat(&p, 2, 2 * p.t_gsw, v_folding + i * words));
    vfn.push_back(ntt_from_flat(&p, 2, 2 * p.t_gsw, v_folding_neg + i * words));
  }
  fold_ciphertexts(p, v, vf, vfn);
  for (size_t i = 0; i < num_per; i++) memcpy(cts + i * rw, v[i].data.data(), rw * 8);
  return 0;
  ORC_CATCH(-1)
}

// pack: v_ct[n*n] raw 2x1; v_w[n] (n+1) x t_conv NTT; out (n+1) x n NTT
int orc_pack(void* h, const uint64_t* v_ct, const uint64_t* v_w, uint64_t* out) {
  ORC_TRY
  const Params& p = *(Params*)h;
  std::vector<PolyMatrixRaw> cts;
  for (size_t i = 0; i < p.n * p.n; i++) cts.push_back(raw_from_flat(&p, 2, 1, v_ct + i * 2 * p.poly_len));
  size_t ww = (p.n + 1) * p.t_conv * p.crt_count * p.poly_len;
  std::vector<PolyMatrixNTT> w;
  for (size_t i = 0; i < p.n; i++) w.push_back(ntt_from_flat(&p, p.n + 1, p.t_conv, v_w + i * ww));
  PolyMatrixNTT r = pack(p, cts, w);
  memcpy(out, r.data.data(), r.data.size() * 8);
  return 0;
  ORC_CATCH(-1)
}

// encode: v_packed[instances] raw (n+1) x n
int64_t orc_encode(void* h, const uint64_t* v_packed, uint8_t* out, uint64_t cap) {
  ORC_TRY
  const Params& p = *(Params*)h;
  std::vector<PolyMatrixRaw> v;
  size_t w = (p.n + 1) * p.n * p.poly_len;
  for (size_t i = 0; i < p.instances; i++) v.push_back(raw_from_flat(&p, p.n + 1, p.n, v_packed + i * w));
  std::vector<uint8_t> r = encode(p, v);
  if (r.size() > cap) return -(int64_t)r.size();
  memcpy(out, r.data(), r.size());
  return (int64_t)r.size();
  ORC_CATCH(-1)
}

void orc_reorient_reg_ciphertexts(void* h, const uint64_t* v_reg, uint64_t* out) {
  const Params& p = *(Params*)h;
  size_t words = 2 * p.crt_count * p.poly_len;
  std::vector<PolyMatrixNTT> v;
  for (size_t i = 0; i < ((size_t)1 << p.db_dim_1); i++) v.push_back(ntt_from_flat(&p, 2, 1, v_reg + i * words));
  reorient_reg_ciphertexts(p, out, v);
}

// Stage timings of one process_query, for bench.py's cpu_baseline leg:
// t_out[0]=expand (+folding_neg), [1]=sweep, [2]=from_ntt+fold, [3]=pack+encode  (seconds)
int64_t orc_process_query_timed(void* h, const uint8_t* pp_bytes, uint64_t pp_len, const uint8_t* q_bytes, uint64_t q_len,
                                const uint64_t* db, uint8_t* out, uint64_t cap, double* t_out) {
  ORC_TRY
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const Params& p = *(Params*)h;
  PublicParameters pp = PublicParameters::deserialize(p, pp_bytes, pp_len);
  Query q = Query::deserialize(p, q_bytes, q_len);
  size_t dim0 = (size_t)1 << p.db_dim_1, num_per = (size_t)1 << p.db_dim_2;
  size_t db_slice_sz = dim0 * num_per * p.poly_len;
  for (int i = 0; i < 4; i++) t_out[i] = 0;
  auto t0 = clk::now();
  std::vector<u64> v_reg;
  std::vector<PolyMatrixNTT> v_folding;
  expand_query(p, pp, q, v_reg, v_folding);
  std::vector<PolyMatrixNTT> v_folding_neg = get_v_folding_neg(p, v_folding);
  t_out[0] = secs(t0, clk::now());
  std::vector<PolyMatrixRaw> v_packed(p.instances);
  for (size_t instance = 0; instance < p.instances; instance++) {
    std::vector<PolyMatrixNTT> inter;
    std::vector<PolyMatrixRaw> inter_raw;
    for (size_t i = 0; i < num_per; i++) {
      inter.emplace_back(&p, 2, 1);
      inter_raw.emplace_back(&p, 2, 1);
    }
    std::vector<PolyMatrixRaw> v_ct;
    for (size_t trial = 0; trial < p.n * p.n; trial++) {
      auto a = clk::now();
      multiply_reg_by_database(inter, db + (instance * p.n * p.n + trial) * db_slice_sz, v_reg.data(), p, dim0, num_per);
      auto b = clk::now();
      for (size_t i = 0; i < num_per; i++) from_ntt(inter_raw[i], inter[i]);
      fold_ciphertexts(p, inter_raw, v_folding, v_folding_neg);
      auto c = clk::now();
      t_out[1] += secs(a, b);
      t_out[2] += secs(b, c);
      v_ct.push_back(inter_raw[0]);
    }
    auto a = clk::now();
    v_packed[instance] = from_ntt_alloc(pack(p, v_ct, pp.v_packing));
    t_out[3] += secs(a, clk::now());
  }
  auto a = clk::now();
  std::vector<uint8_t> r = encode(p, v_packed);
  t_out[3] += secs(a, clk::now());
  if (r.size() > cap) return -(int64_t)r.size();
  memcpy(out, r.data(), r.size());
  return (int64_t)r.size();
  ORC_CATCH(-1)
}

}  // extern "C"
