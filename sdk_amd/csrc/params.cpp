// Host-side scheme parameters (see params.hpp).  Reference: lib/spiral-rs/src/params.rs,
// ntt.rs:6-65 (table construction), number_theory.rs:41-96, util.rs:219-263 (JSON keys).
#include "params.hpp"
#include "kernels.hpp"   // N, wtw_phys: the wave transform's table image is built here

#include <cctype>
#include <cmath>
#include <map>
#include <stdexcept>

namespace spiral {

typedef unsigned __int128 u128;

u64 mul_mod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }
u64 pow_mod(u64 a, u64 e, u64 m) {
  u64 r = 1 % m;
  a %= m;
  while (e) {
    if (e & 1) r = mul_mod(r, a, m);
    a = mul_mod(a, a, m);
    e >>= 1;
  }
  return r;
}
u64 inv_mod(u64 a, u64 m) {
  // m need not be prime (Q = q0*q1 is used by callers): extended Euclid on signed 128-bit
  __int128 t = 0, nt = 1, r = m, nr = a % m;
  while (nr != 0) {
    __int128 q = r / nr;
    __int128 tmp = t - q * nt;
    t = nt;
    nt = tmp;
    tmp = r - q * nr;
    r = nr;
    nr = tmp;
  }
  if (r != 1) throw std::runtime_error("inv_mod: not invertible");
  if (t < 0) t += m;
  return (u64)t;
}

static size_t ceil_log2(size_t a) {
  size_t l = 0;
  while (((size_t)1 << l) < a) l++;
  return l;
}

size_t Params::g() const { return ceil_log2(t_gsw * db_dim_2 + dim0()); }       // params.rs:129-132
size_t Params::stop_round() const { return ceil_log2(t_gsw * db_dim_2); }        // params.rs:134-136

size_t Params::setup_bytes() const {  // params.rs:146-167
  size_t polys = (version == 0 ? n : 2) * n * t_conv;
  if (expand_queries) {
    size_t right = (stop_round() + 1) * t_exp_right;
    if (version > 0 && t_exp_left == t_exp_right) right = 0;
    polys += g() * t_exp_left + right + 2 * t_conv;
  }
  return SEED_LENGTH + polys * poly_len * sizeof(u64);
}

size_t Params::query_bytes() const {  // params.rs:169-182
  size_t polys = expand_queries ? 1 : dim0() + db_dim_2 * 2 * t_gsw;
  return SEED_LENGTH + polys * poly_len * sizeof(u64);
}

size_t Params::response_bytes() const {  // server.rs:471-480
  size_t q1_bits = ceil_log2(4 * pt_modulus);
  size_t bits = instances * (q2_bits * n * poly_len + q1_bits * n * n * poly_len);
  return ((bits + 63) / 64) * 8;
}

size_t Params::bits_per(size_t dim) const {  // gadget.rs:3-9
  if (dim == modulus_log2) return 1;
  return (size_t)(modulus_log2 / dim) + 1;
}

u64 Params::q2() const {  // params.rs:8-46
  static const u64 tab[37] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12289, 12289, 61441, 65537, 65537,
                              520193, 786433, 786433, 3604481, 7340033, 16515073, 33292289, 67043329, 132120577,
                              268369921, 469762049, 1073479681, 2013265921, 4293918721ULL, 8588886017ULL,
                              17175674881ULL, 34359214081ULL, 68718428161ULL};
  if (q2_bits >= 37) throw std::runtime_error("q2_bits out of range");
  return tab[q2_bits];
}

// Smallest primitive `degree`-th root of unity mod prime q (number_theory.rs:41-55 picks the
// minimum over all of them, so any starting root gives the same answer).
static u64 minimal_primitive_root(u64 degree, u64 q) {
  if ((q - 1) % degree) throw std::runtime_error("modulus does not support NTT of this size");
  u64 cof = (q - 1) / degree, root = 0;
  for (u64 c = 2; c < 1000; c++) {
    u64 r = pow_mod(c, cof, q);
    if (pow_mod(r, degree / 2, q) == q - 1) {
      root = r;
      break;
    }
  }
  if (!root) throw std::runtime_error("no primitive root found");
  u64 sq = mul_mod(root, root, q), cur = root, best = root;
  for (u64 i = 0; i < degree / 2; i++) {  // odd powers = all primitive roots
    if (cur < best) best = cur;
    cur = mul_mod(cur, sq, q);
  }
  return best;
}

static u32 bitrev(u32 x, int bits) {
  u32 r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}

void Params::finish() {
  modulus = moduli[0] * moduli[1];
  modulus_log2 = 0;
  while (((u128)1 << modulus_log2) < (u128)modulus) modulus_log2++;  // log2_ceil, arith.rs:13-15
  // [crt][4][N] as the reference's, then [crt][2][N]: the kernels' inverse tables, then [crt][2][N]: the wave transform's LDS image
  ntt_tables.assign(CRT_COUNT * 8 * poly_len, 0);
  for (int c = 0; c < 2; c++) {
    u64 q = moduli[c];
    u64 psi = minimal_primitive_root(2 * poly_len, q);
    u64 psi_inv = inv_mod(psi, q);
    u32* fw = ntt_tables.data() + ((size_t)c * 4 + 0) * poly_len;
    u32* fwp = fw + poly_len;
    u32* iw = fwp + poly_len;
    u32* iwp = iw + poly_len;
    // The kernels' inverse transform (device_common.hpp, gs_bfly) uses the UNHALVED inverse twiddles psi^-i and multiplies by
    // N^-1 in its last stage instead of halving in every butterfly (ntt.rs:236-249): entries 0 and 1 -- the last stage's
    // (x + y) and (x - y) factors -- carry N^-1.  Same residues, so the same canonical results.
    u32* lw = ntt_tables.data() + (size_t)CRT_COUNT * 4 * poly_len + (size_t)c * 2 * poly_len;
    u32* lwp = lw + poly_len;
    const u64 n_inv = inv_mod((u64)poly_len % q, q);
    u64 pw = 1, ipw = 1;
    for (size_t i = 0; i < poly_len; i++) {
      // ntt.rs:6-17: table[bitrev(i)] = root^i; inverse table additionally halved (ntt.rs:50-53)
      size_t idx = bitrev((u32)i, (int)poly_len_log2);
      u64 half = (ipw & 1) ? (ipw + q) >> 1 : ipw >> 1;  // div2_uint_mod, arith.rs:78-89
      fw[idx] = (u32)pw;
      fwp[idx] = (u32)((pw << 32) / q);      // scale_powers_u32, ntt.rs:29-37
      iw[idx] = (u32)half;
      iwp[idx] = (u32)((half << 32) / q);
      const u64 lz = idx < 2 ? mul_mod(ipw, n_inv, q) : ipw;
      lw[idx] = (u32)lz;
      lwp[idx] = (u32)((lz << 32) / q);
      pw = mul_mod(pw, psi, q);
      ipw = mul_mod(ipw, psi_inv, q);
    }
    // the wave transform's LDS image of the forward tables (kernels.hpp, wave_fwd_image): [ -w | w' ] at wtw_phys(idx); only
    // defined for the transform length the kernels are built for
    if (poly_len == (size_t)N) {
      u32* img = ntt_tables.data() + (size_t)CRT_COUNT * 6 * poly_len + (size_t)c * 2 * poly_len;
      for (size_t i = 0; i < poly_len; i++) {
        const int ph = wtw_phys((int)i);
        img[ph] = 0u - fw[i];
        img[poly_len + ph] = fwp[i];
      }
    }
    dc.mod[c].q = (u32)q;
    dc.mod[c].two_q = (u32)(2 * q);
    dc.mod[c].m64 = (u64)((((u128)1) << 64) / q);
  }
  dc.Q = modulus;
  u64 inv = inv_mod(moduli[0] % moduli[1], moduli[1]);
  dc.q0_inv_q1 = (u32)inv;
  dc.q0_inv_q1_sh = (u32)((inv << 32) / moduli[1]);
}

// ---- minimal JSON object reader: flat {"key": number, ...}; single or double quotes.
static std::map<std::string, double> parse_flat_json(const std::string& s) {
  std::map<std::string, double> out;
  size_t i = 0, n = s.size();
  auto skip = [&]() {
    while (i < n && isspace((unsigned char)s[i])) i++;
  };
  skip();
  if (i >= n || s[i] != '{') throw std::runtime_error("params json: expected '{'");
  i++;
  for (;;) {
    skip();
    if (i < n && s[i] == '}') break;
    if (i >= n || (s[i] != '"' && s[i] != '\'')) throw std::runtime_error("params json: expected key");
    char qc = s[i++];
    size_t k0 = i;
    while (i < n && s[i] != qc) i++;
    if (i >= n) throw std::runtime_error("params json: unterminated key");
    std::string key = s.substr(k0, i - k0);
    i++;
    skip();
    if (i >= n || s[i] != ':') throw std::runtime_error("params json: expected ':'");
    i++;
    skip();
    size_t v0 = i;
    if (i < n && (s[i] == '"' || s[i] == '\'')) {  // string value: ignored (not used by any key we read)
      char q2c = s[i++];
      while (i < n && s[i] != q2c) i++;
      i++;
      out[key] = 0;
    } else {
      while (i < n && s[i] != ',' && s[i] != '}' && !isspace((unsigned char)s[i])) i++;
      std::string tok = s.substr(v0, i - v0);
      if (tok == "true")
        out[key] = 1;
      else if (tok == "false" || tok == "null")
        out[key] = 0;
      else {
        try {
          out[key] = std::stod(tok);
        } catch (...) {
          throw std::runtime_error("params json: bad value for " + key);
        }
      }
    }
    skip();
    if (i < n && s[i] == ',') {
      i++;
      continue;
    }
    skip();
    if (i < n && s[i] == '}') break;
    throw std::runtime_error("params json: expected ',' or '}'");
  }
  return out;
}

Params Params::from_json(const std::string& json) {  // util.rs:224-263
  auto kv = parse_flat_json(json);
  auto need = [&](const char* k) -> u64 {
    auto it = kv.find(k);
    if (it == kv.end()) throw std::runtime_error(std::string("params json: missing key ") + k);
    if (it->second < 0) throw std::runtime_error(std::string("params json: negative ") + k);
    return (u64)it->second;
  };
  auto opt = [&](const char* k, u64 dflt) -> u64 {
    auto it = kv.find(k);
    return it == kv.end() ? dflt : (u64)it->second;
  };
  Params p;
  p.n = need("n");
  p.db_dim_1 = need("nu_1");
  p.db_dim_2 = need("nu_2");
  p.instances = opt("instances", 1);
  if (p.instances == 0) p.instances = 1;
  p.pt_modulus = need("p");
  p.q2_bits = std::max<u64>(need("q2_bits"), 14);  // MIN_Q2_BITS, params.rs:7
  p.t_gsw = need("t_gsw");
  p.t_conv = need("t_conv");
  p.t_exp_left = need("t_exp_left");
  p.t_exp_right = need("t_exp_right");
  p.expand_queries = kv.find("direct_upload") == kv.end();
  p.db_item_size = opt("db_item_size", 0);
  if (p.db_item_size == 0) {
    size_t logp = ceil_log2(p.pt_modulus);
    p.db_item_size = p.instances * p.n * p.n * 2048 * logp / 8;
  }
  p.version = opt("version", 0);
  if (p.n == 0 || p.n > 8) throw std::runtime_error("params: n out of range");
  // Q2_VALUES ends at index 36 (params.rs:8-46): the reference indexes out of bounds beyond (smaller values were raised to
  // MIN_Q2_BITS above, as util.rs:230 does)
  if (p.q2_bits > 36) throw std::runtime_error("params: q2_bits above 36 (params.rs:8-46)");
  if (p.pt_modulus < 2 || (p.pt_modulus & (p.pt_modulus - 1))) throw std::runtime_error("params: p must be a power of two");
  if (p.t_gsw == 0 || p.t_conv == 0 || p.t_exp_left == 0 || p.t_exp_right == 0) throw std::runtime_error("params: zero gadget dimension");
  if (p.db_dim_1 > 20 || p.db_dim_2 > 20) throw std::runtime_error("params: db dimensions out of range");
  if (p.version > 1) throw std::runtime_error("params: unknown packing version (pack.rs:110-112)");
  if (p.version == 1 && p.n != 2)
    throw std::runtime_error("params: version 1 needs n == 2 (client.rs:221 reads n packing matrices, params.rs:149 sizes 2)");
  p.finish();
  if (p.expand_queries && ((size_t)1 << p.g()) > p.poly_len) throw std::runtime_error("params: query does not fit one polynomial");
  // expand_query reads v[2i] for i < dim0 and v[2i+1] for i < t_gsw*nu_2 out of 2^g expanded ciphertexts
  // (server.rs:566-571); the reference panics (index out of bounds) when they do not fit
  if (p.expand_queries && p.db_dim_2 > 0 && 2 * std::max(p.dim0(), p.t_gsw * p.db_dim_2) > ((size_t)1 << p.g()))
    throw std::runtime_error("params: 2*max(2^nu_1, t_gsw*nu_2) exceeds 2^g expanded ciphertexts (server.rs:566-571)");
  // coefficient_expansion prunes the odd subtree only when stop_round > 0 (server.rs:40-47).  With stop_round = 0, i.e.
  // t_gsw * nu_2 = 1, every round r >= 1 takes v_w_right[r] -- of which the public parameters hold stop_round + 1 = 1
  // (client.rs:146-152, params.rs:146-167): the reference indexes out of bounds and panics; so would the expansion plan here
  // (found by scripts/emu_fuzz.py: the multiply-accumulate of round 1 read past the public parameters)
  if (p.expand_queries && p.db_dim_2 > 0 && p.stop_round() == 0 && p.g() > 1)
    throw std::runtime_error("params: t_gsw * nu_2 = 1 leaves one right expansion matrix for g > 1 rounds (server.rs:40-47, 60-73)");
  return p;
}

}  // namespace spiral
