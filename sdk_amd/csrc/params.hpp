// Host-side scheme parameters for the MI355X Spiral answer path.
// Mirrors spiral_rs::params::Params (lib/spiral-rs/src/params.rs:49-82) -- same field names and
// derived sizes -- but carries the tables in the form the HIP kernels consume (u32 twiddles,
// Shoup quotients, Barrett/Garner constants).
#pragma once
// The wire formats are the reference's to_ne_bytes / from_ne_bytes images (client.rs:58,77,292; util.rs:294-319;
// server.rs:369), i.e. little-endian on the x86-64 hosts both run on.
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "spiral_hip assumes a little-endian host");
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace spiral {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr size_t POLY_LEN = 2048;      // util.rs:245 (hard-wired in params_from_json)
constexpr size_t POLY_LEN_LOG2 = 11;
constexpr size_t CRT_COUNT = 2;
constexpr u64 MODULUS_0 = 268369921ULL;  // util.rs:246
constexpr u64 MODULUS_1 = 249561089ULL;
constexpr size_t SEED_LENGTH = 32;     // client.rs:12

// Per-modulus constants handed to kernels by value.
struct ModConst {
  u32 q;        // modulus (< 2^28)
  u32 two_q;
  u64 m64;      // floor(2^64 / q): 64-bit Barrett multiplier
};

// Everything a kernel needs that is not a table.
struct DevConsts {
  ModConst mod[2];
  u64 Q;             // q0 * q1
  u32 q0_inv_q1;     // q0^{-1} mod q1 (Garner)
  u32 q0_inv_q1_sh;  // floor(q0_inv_q1 * 2^32 / q1) (Shoup quotient)
};

struct Params {
  // --- spiral-rs field names (params.rs:49-82)
  size_t poly_len = POLY_LEN, poly_len_log2 = POLY_LEN_LOG2, crt_count = CRT_COUNT;
  u64 moduli[2] = {MODULUS_0, MODULUS_1};
  u64 modulus = 0, modulus_log2 = 0;
  double noise_width = 6.4;
  size_t n = 0;
  u64 pt_modulus = 0, q2_bits = 0;
  size_t t_conv = 0, t_exp_left = 0, t_exp_right = 0, t_gsw = 0;
  bool expand_queries = true;
  size_t db_dim_1 = 0, db_dim_2 = 0, instances = 1, db_item_size = 0, version = 0;
  // ntt_tables[crt][which][i] as in params.rs:85-96 (0 fwd, 1 fwd', 2 inv, 3 inv')
  std::vector<u32> ntt_tables;  // [crt][4][N], then [crt][2][N]: the kernels' lazy inverse tables (params.cpp, finish)
  DevConsts dc;

  // --- derived sizes (params.rs:116-200, server.rs:476-480)
  size_t dim0() const { return (size_t)1 << db_dim_1; }
  size_t num_per() const { return (size_t)1 << db_dim_2; }
  size_t num_items() const { return dim0() * num_per(); }
  size_t trials() const { return n * n; }
  size_t planes() const { return instances * n * n; }
  size_t g() const;
  size_t stop_round() const;
  size_t setup_bytes() const;
  size_t query_bytes() const;
  size_t response_bytes() const;
  size_t db_words() const { return planes() * num_items() * poly_len; }
  bool has_expansion_right_on_wire() const { return version == 0 || t_exp_right != t_exp_left; }
  size_t bits_per(size_t dim) const;  // gadget.rs:3-9
  u64 q2() const;                     // Q2_VALUES[q2_bits], params.rs:8-46
  const u32* table(int crt, int which) const { return ntt_tables.data() + ((size_t)crt * 4 + which) * poly_len; }

  // util.rs:219-263.  Throws std::runtime_error on malformed input.
  static Params from_json(const std::string& json);
  void finish();  // compute derived constants + tables (params.rs:224-296, ntt.rs:39-65)
};

u64 mul_mod(u64 a, u64 b, u64 m);
u64 pow_mod(u64 a, u64 e, u64 m);
u64 inv_mod(u64 a, u64 m);

}  // namespace spiral
