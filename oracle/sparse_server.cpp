// ============================================================================
// oracle/sparse_server.cpp -- TEST INFRASTRUCTURE ONLY (see spiral_oracle.h).
// CPU restatement of lib/server's SPARSE answer path, the caller the GPU library would serve
// inside blyssprivacy/sdk.  Citations are relative to /root/reference/lib/server/src/.
//   SparseDb                        db/sparse_db.rs:5-48
//   convert_pt_to_poly, pack_ntt_poly, update_item_raw          db/loading.rs:34-41, 278-359
//   multiply_reg_by_sparse_database compute/dot_product.rs:136-220 (scalar form)
//   to_per_round_set, coefficient_expansion (pruned), reorient_reg_ciphertexts, expand_query
//                                   compute/query_expansion.rs:33-148, 180-248, 250-357
//   fold_ciphertexts (all-zero shortcuts)                        compute/fold.rs:6-80
//   process_query                   server.rs:17-99
// ONE deliberate difference, stated in SURVEY.md 8(f)-1: the reference accumulates the first-dimension products in
// wrapping u64 (its MAX_SUMMED guard does not fire per accumulator) and can overflow for >= 512 present items in a
// column; this restatement keeps the exact sums (u128, one % q), which is what the GPU path is held to.
// ============================================================================
#include <cmath>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>

#include "spiral_oracle.h"

#define ORACLE_CHECK(cond)                                                       \
  do {                                                                           \
    if (!(cond)) throw std::runtime_error("oracle check failed: " #cond);        \
  } while (0)

namespace oracle {

static size_t log2_ceil_usize(size_t a) { return a == 0 ? 0 : (size_t)std::ceil(std::log2((double)a)); }  // arith.rs:21-23
static inline u64 barrett_coeff_u64(const Params& p, u64 val, size_t n) {  // arith.rs:140-142
  return barrett_raw_u64(val, p.barrett_cr_1[n], p.moduli[n]);
}

// db/sparse_db.rs:5-48
struct SparseDb {
  std::vector<std::vector<u64>> data;                   // series of polynomials (N packed words each)
  std::unordered_map<size_t, size_t> db_idx_to_vec_idx; // db_idx to data vector index
  const size_t* get_idx(size_t idx) const {
    auto it = db_idx_to_vec_idx.find(idx);
    return it == db_idx_to_vec_idx.end() ? nullptr : &it->second;
  }
  void upsert(size_t idx, const std::vector<u64>& poly) {
    if (const size_t* v = get_idx(idx)) {
      data[*v] = poly;
    } else {
      data.push_back(poly);
      db_idx_to_vec_idx[idx] = data.size() - 1;
    }
  }
};

// db/loading.rs:278-304: one chunk of plaintext bytes (log2 p = 8) -> centred coefficients -> NTT
static PolyMatrixNTT convert_pt_to_poly(const Params& params, const uint8_t* data, size_t len) {
  ORACLE_CHECK(params.pt_modulus == 256);  // loading.rs:290 assert_eq!(logp, 8)
  ORACLE_CHECK(len <= params.poly_len);
  PolyMatrixRaw item(&params, 1, 1);
  for (size_t i = 0; i < len; i++) item.data[i] = recenter_mod((u64)data[i], params.pt_modulus, params.modulus);
  return to_ntt_alloc(item);
}
// db/loading.rs:34-41
static std::vector<u64> pack_ntt_poly(const Params& params, const PolyMatrixNTT& poly) {
  std::vector<u64> v(params.poly_len);
  for (size_t z = 0; z < params.poly_len; z++) v[z] = poly.data[z] | (poly.data[params.poly_len + z] << 32);
  return v;
}
// db/loading.rs:317-359
void sparse_update_item_raw(const Params& params, size_t db_idx, const uint8_t* data, size_t len, SparseDb& db) {
  const size_t instances = params.instances, trials = params.n * params.n;
  const size_t pt_data_len = params.bytes_per_chunk();
  std::vector<uint8_t> new_bucket(instances * trials * pt_data_len, 0);
  ORACLE_CHECK(len <= new_bucket.size());
  memcpy(new_bucket.data(), data, len);
  ORACLE_CHECK(db_idx < params.num_items());
  for (size_t inst_trial = 0; inst_trial < instances * trials; inst_trial++) {
    PolyMatrixNTT ntt = convert_pt_to_poly(params, new_bucket.data() + inst_trial * pt_data_len, pt_data_len);
    db.upsert(inst_trial * params.num_items() + db_idx, pack_ntt_poly(params, ntt));
  }
}

// compute/dot_product.rs:136-220; query: [dim0][2][N] (compute/query_expansion.rs:180-211), out[i]: 2 x 1 NTT.
// Exact accumulation (see the header).
static void multiply_reg_by_sparse_database(std::vector<PolyMatrixNTT>& out, const SparseDb& db, const u64* query,
                                            const Params& params, size_t dim0, size_t num_per, size_t db_idx) {
  const size_t N = params.poly_len;
  std::vector<u128> acc(num_per * 4 * N, 0);
  for (size_t j = 0; j < dim0; j++)
    for (size_t i = 0; i < num_per; i++) {
      const size_t full_idx = db_idx * (dim0 * num_per) + j * num_per + i;
      const size_t* real_idx = db.get_idx(full_idx);
      if (!real_idx) continue;
      const u64* b_poly = db.data[*real_idx].data();
      u128* a = acc.data() + i * 4 * N;
      for (size_t z = 0; z < N; z++) {
        const u64 a1 = query[(j * 2) * N + z], a2 = query[(j * 2 + 1) * N + z], b = b_poly[z];
        const u64 a1_lo = (u32)a1, a1_hi = a1 >> 32, a2_lo = (u32)a2, a2_hi = a2 >> 32, b_lo = (u32)b, b_hi = b >> 32;
        a[z] += (u128)(a1_lo * b_lo);            // out_0: row 0, crt 0
        a[N + z] += (u128)(a1_hi * b_hi);        // out_1: row 0, crt 1
        a[2 * N + z] += (u128)(a2_lo * b_lo);    // out_2: row 1, crt 0
        a[3 * N + z] += (u128)(a2_hi * b_hi);    // out_3: row 1, crt 1
      }
    }
  for (size_t i = 0; i < num_per; i++)
    for (size_t k = 0; k < 4; k++)
      for (size_t z = 0; z < N; z++)
        out[i].data[k * N + z] = (u64)(acc[(i * 4 + k) * N + z] % (u128)params.moduli[k & 1]);
}

// compute/query_expansion.rs:213-248
static std::set<std::pair<size_t, size_t>> to_per_round_set(const Params& params, const std::set<size_t>& indices) {
  std::set<std::pair<size_t, size_t>> to_do;
  const size_t g = params.g();
  for (size_t i = 0; i < ((size_t)1 << g); i++)
    if ((i % 2 == 0 && indices.count(i / 2)) || (i % 2 == 1)) to_do.insert({g - 1, i});
  for (size_t rr = g - 1; rr-- > 0;) {
    const size_t r = rr;
    for (size_t i = 0; i < ((size_t)1 << (r + 1)); i++) {
      const bool left = to_do.count({r + 1, i}) != 0;
      const bool right = to_do.count({r + 1, i + ((size_t)1 << (r + 1))}) != 0;
      if (left || right) to_do.insert({r, i});
    }
  }
  return to_do;
}

// compute/query_expansion.rs:33-148: coefficient_expansion with the (round, out_idx) work set
static void coefficient_expansion_pruned(std::vector<PolyMatrixNTT>& v, size_t g, size_t stop_round, const Params& params,
                                         const std::vector<PolyMatrixNTT>& v_w_left,
                                         const std::vector<PolyMatrixNTT>& v_w_right,
                                         const std::vector<PolyMatrixNTT>& v_neg1, size_t max_bits_to_gen_right,
                                         const std::set<std::pair<size_t, size_t>>* indices) {
  const size_t poly_len = params.poly_len;
  for (size_t r = 0; r < g; r++) {
    const size_t num_in = (size_t)1 << r, num_out = 2 * num_in;
    const size_t t = (poly_len / ((size_t)1 << r)) + 1;
    const PolyMatrixNTT& neg1 = v_neg1[r];
    auto action_expand = [&](size_t i, PolyMatrixNTT& v_i, size_t j) {
      if ((stop_round > 0 && r > stop_round && (i % 2) == 1) ||
          (stop_round > 0 && r == stop_round && (i % 2) == 1 && (i / 2) >= max_bits_to_gen_right))
        return;
      const size_t out_idx = j * num_in + i;
      if (indices && !indices->count({r, out_idx})) return;
      PolyMatrixRaw ct(&params, 2, 1), ct_auto(&params, 2, 1), ct_auto_1(&params, 1, 1);
      PolyMatrixNTT ct_auto_1_ntt(&params, 1, 1), w_times_ginv_ct(&params, 2, 1);
      const bool left = (r != 0) && (i % 2 == 0);
      const PolyMatrixNTT& w = left ? v_w_left[r] : v_w_right[r];
      const size_t gadget_dim = left ? params.t_exp_left : params.t_exp_right;
      PolyMatrixRaw gi_ct(&params, gadget_dim, 1);
      PolyMatrixNTT gi_ct_ntt(&params, gadget_dim, 1);
      from_ntt(ct, v_i);
      automorph(ct_auto, ct, t);
      gadget_invert_rdim(gi_ct, ct_auto, 1);
      to_ntt_no_reduce(gi_ct_ntt, gi_ct);
      memcpy(ct_auto_1.data.data(), ct_auto.get_poly(1, 0), poly_len * sizeof(u64));
      to_ntt(ct_auto_1_ntt, ct_auto_1);
      multiply(w_times_ginv_ct, w, gi_ct_ntt);
      size_t idx = 0;
      for (size_t jj = 0; jj < 2; jj++)
        for (size_t n = 0; n < params.crt_count; n++)
          for (size_t z = 0; z < poly_len; z++) {
            const u64 sum = v_i.data[idx] + w_times_ginv_ct.data[idx] + jj * ct_auto_1_ntt.data[n * poly_len + z];
            v_i.data[idx] = barrett_coeff_u64(params, sum, n);
            idx++;
          }
    };
    for (size_t i = 0; i < num_in; i++) scalar_multiply(v[num_in + i], neg1, v[i]);
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < num_in; i++) action_expand(i, v[i], 0);
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < num_out - num_in; i++) action_expand(i, v[num_in + i], 1);
  }
}

// compute/query_expansion.rs:180-211: out[j][r][z] = lo | hi << 32
static void reorient_reg_ciphertexts_server(const Params& params, u64* out, const std::vector<PolyMatrixNTT>& v_reg) {
  const size_t N = params.poly_len, dim0 = (size_t)1 << params.db_dim_1;
  for (size_t j = 0; j < dim0; j++)
    for (size_t r = 0; r < 2; r++)
      for (size_t z = 0; z < N; z++) {
        const size_t idx_a_in = r * (params.crt_count * N);
        const u64 val1 = v_reg[j].data[idx_a_in + z] % params.moduli[0];
        const u64 val2 = v_reg[j].data[idx_a_in + N + z] % params.moduli[1];
        out[j * (2 * N) + r * N + z] = val1 | (val2 << 32);
      }
}

// compute/query_expansion.rs:250-357
static void expand_query_sparse(const Params& params, const PublicParameters& pp, const Query& query,
                                const std::unordered_map<size_t, size_t>* indices, std::vector<u64>& v_reg_reoriented,
                                std::vector<PolyMatrixNTT>& v_folding) {
  const size_t dim0 = (size_t)1 << params.db_dim_1, further_dims = params.db_dim_2;
  const size_t num_bits_to_gen = params.t_gsw * further_dims + dim0;
  const size_t g = log2_ceil_usize(num_bits_to_gen);
  const size_t right_expanded = params.t_gsw * further_dims;
  const size_t stop_round = log2_ceil_usize(right_expanded);
  std::vector<PolyMatrixNTT> v;
  for (size_t i = 0; i < ((size_t)1 << g); i++) v.emplace_back(&params, 2, 1);
  v[0].copy_into(to_ntt_alloc(query.ct), 0, 0);
  const PolyMatrixNTT& v_conversion = pp.v_conversion[0];
  const std::vector<PolyMatrixNTT>& v_w_left = pp.v_expansion_left;
  const std::vector<PolyMatrixNTT>& v_w_right = pp.has_expansion_right ? pp.v_expansion_right : v_w_left;
  std::vector<PolyMatrixNTT> v_neg1 = get_v_neg1(params);
  std::set<std::pair<size_t, size_t>> to_do;
  const std::set<std::pair<size_t, size_t>>* indices_to_do = nullptr;
  if (indices) {
    std::set<size_t> set_dim0;
    for (const auto& kv : *indices)
      if (kv.first < params.num_items()) set_dim0.insert(kv.first / ((size_t)1 << params.db_dim_2));
    to_do = to_per_round_set(params, set_dim0);
    indices_to_do = &to_do;
  }
  std::vector<PolyMatrixNTT> v_reg_inp, v_gsw_inp;
  if (further_dims > 0) {
    coefficient_expansion_pruned(v, g, stop_round, params, v_w_left, v_w_right, v_neg1, params.t_gsw * params.db_dim_2,
                                 indices_to_do);
    ORACLE_CHECK(2 * std::max(dim0, right_expanded) <= v.size());
    for (size_t i = 0; i < dim0; i++) v_reg_inp.push_back(v[2 * i]);
    for (size_t i = 0; i < right_expanded; i++) v_gsw_inp.push_back(v[2 * i + 1]);
  } else {
    coefficient_expansion_pruned(v, g, 0, params, v_w_left, v_w_left, v_neg1, 0, indices_to_do);
    for (size_t i = 0; i < dim0; i++) v_reg_inp.push_back(v[i]);
  }
  v_reg_reoriented.assign(dim0 * 2 * params.poly_len, 0);
  reorient_reg_ciphertexts_server(params, v_reg_reoriented.data(), v_reg_inp);
  v_folding.clear();
  for (size_t i = 0; i < params.db_dim_2; i++) v_folding.emplace_back(&params, 2, 2 * params.t_gsw);
  regev_to_gsw(v_folding, v_gsw_inp, v_conversion, params, 1, 0);
}

static bool is_all_zeros(const std::vector<u64>& v) {
  for (u64 x : v)
    if (x != 0) return false;
  return true;
}

// compute/fold.rs:16-80
void fold_ciphertexts_sparse(const Params& params, std::vector<PolyMatrixRaw>& v_cts,
                             const std::vector<PolyMatrixNTT>& v_folding, const std::vector<PolyMatrixNTT>& v_folding_neg) {
  if (v_cts.size() == 1) return;
  const size_t further_dims = (size_t)log2_floor(v_cts.size());
  const size_t ell = v_folding[0].cols / 2;
  size_t num_per = v_cts.size();
  for (size_t cur_dim = 0; cur_dim < further_dims; cur_dim++) {
    num_per = num_per / 2;
    for (size_t i = 0; i < num_per; i++) {
      PolyMatrixRaw ginv_c(&params, 2 * ell, 1);
      PolyMatrixNTT ginv_c_ntt(&params, 2 * ell, 1);
      PolyMatrixNTT prod(&params, 2, 1), sum(&params, 2, 1);
      // crucial for correctness (fold.rs:38-44)
      if (is_all_zeros(v_cts[i].data)) {
        v_cts[i].copy_into(v_cts[num_per + i], 0, 0);
        continue;
      } else if (is_all_zeros(v_cts[i + num_per].data)) {
        continue;
      }
      gadget_invert(ginv_c, v_cts[i]);
      to_ntt(ginv_c_ntt, ginv_c);
      multiply(prod, v_folding_neg[further_dims - 1 - cur_dim], ginv_c_ntt);
      gadget_invert(ginv_c, v_cts[num_per + i]);
      to_ntt(ginv_c_ntt, ginv_c);
      multiply(sum, v_folding[further_dims - 1 - cur_dim], ginv_c_ntt);
      add_into(sum, prod);
      from_ntt(v_cts[i], sum);
    }
  }
}

// server.rs:17-99
std::vector<uint8_t> process_query_sparse(const Params& params, const PublicParameters& pp, const Query& query,
                                          const SparseDb& db) {
  const size_t dim0 = (size_t)1 << params.db_dim_1, num_per = (size_t)1 << params.db_dim_2;
  std::vector<u64> v_reg_reoriented;
  std::vector<PolyMatrixNTT> v_folding;
  if (params.expand_queries) {
    expand_query_sparse(params, pp, query, &db.db_idx_to_vec_idx, v_reg_reoriented, v_folding);
  } else {
    // the non-expanded query carries v_buf in spiral-rs's [z][j][r] order (client.rs:105-128); lib/server's multiply
    // indexes [j][r][z] -- the reference server is only ever run with expand_queries = true (its test params and
    // bin/server.rs default), so the restatement rejects the combination instead of guessing a layout
    throw std::runtime_error("process_query_sparse: direct_upload params are not served by lib/server's sparse path");
  }
  std::vector<PolyMatrixNTT> v_folding_neg = get_v_folding_neg(params, v_folding);
  const size_t trials = params.n * params.n;
  std::vector<PolyMatrixRaw> v_cts(params.instances * trials, PolyMatrixRaw(&params, 2, 1));
#pragma omp parallel for
  for (size_t instance_trial = 0; instance_trial < params.instances * trials; instance_trial++) {
    std::vector<PolyMatrixNTT> intermediate;
    std::vector<PolyMatrixRaw> intermediate_raw;
    for (size_t i = 0; i < num_per; i++) {
      intermediate.emplace_back(&params, 2, 1);
      intermediate_raw.emplace_back(&params, 2, 1);
    }
    multiply_reg_by_sparse_database(intermediate, db, v_reg_reoriented.data(), params, dim0, num_per, instance_trial);
    for (size_t i = 0; i < intermediate.size(); i++) from_ntt(intermediate_raw[i], intermediate[i]);
    fold_ciphertexts_sparse(params, intermediate_raw, v_folding, v_folding_neg);
    v_cts[instance_trial] = intermediate_raw[0];
  }
  std::vector<PolyMatrixRaw> v_packed_ct;
  for (size_t inst = 0; inst < params.instances; inst++) {
    std::vector<PolyMatrixRaw> chunk(v_cts.begin() + inst * trials, v_cts.begin() + (inst + 1) * trials);
    v_packed_ct.push_back(from_ntt_alloc(pack_dispatch(params, chunk, pp.v_packing)));
  }
  return encode(params, v_packed_ct);
}

}  // namespace oracle

// ------------------------------------------------------------------------------------------------ C surface
using namespace oracle;
static thread_local std::string g_sparse_err;

extern "C" {

const char* orc_sparse_last_error() { return g_sparse_err.c_str(); }
void* orc_sparse_db_new() { return new SparseDb(); }
void orc_sparse_db_free(void* d) { delete (SparseDb*)d; }
uint64_t orc_sparse_db_polys(void* d) { return ((SparseDb*)d)->data.size(); }

int orc_sparse_update_item_raw(void* h, void* d, uint64_t db_idx, const uint8_t* data, uint64_t len) {
  try {
    sparse_update_item_raw(*(Params*)h, db_idx, data, len, *(SparseDb*)d);
    return 0;
  } catch (const std::exception& e) {
    g_sparse_err = e.what();
    return -1;
  }
}

int64_t orc_process_query_sparse(void* h, const uint8_t* pp_bytes, uint64_t pp_len, const uint8_t* q_bytes, uint64_t q_len,
                                 void* d, uint8_t* out, uint64_t cap) {
  try {
    const Params& p = *(Params*)h;
    PublicParameters pp = PublicParameters::deserialize(p, pp_bytes, pp_len);
    Query q = Query::deserialize(p, q_bytes, q_len);
    std::vector<uint8_t> r = process_query_sparse(p, pp, q, *(SparseDb*)d);
    if (r.size() > cap) return -(int64_t)r.size();
    memcpy(out, r.data(), r.size());
    return (int64_t)r.size();
  } catch (const std::exception& e) {
    g_sparse_err = e.what();
    return -1;
  }
}

// the dense equivalent of a sparse bucket (absent items = zero polynomials), reference layout [plane][z][ii][j]
int orc_sparse_to_dense(void* h, void* d, uint64_t* out) {
  const Params& p = *(Params*)h;
  const SparseDb& db = *(SparseDb*)d;
  const size_t N = p.poly_len, dim0 = (size_t)1 << p.db_dim_1, num_per = (size_t)1 << p.db_dim_2, planes = p.instances * p.n * p.n;
  memset(out, 0, planes * N * num_per * dim0 * 8);
  for (const auto& kv : db.db_idx_to_vec_idx) {
    const size_t plane = kv.first / p.num_items(), idx = kv.first % p.num_items();
    const size_t j = idx / num_per, ii = idx % num_per;
    const u64* poly = db.data[kv.second].data();
    for (size_t z = 0; z < N; z++) out[((plane * N + z) * num_per + ii) * dim0 + j] = poly[z];
  }
  return 0;
}

}  // extern "C"
