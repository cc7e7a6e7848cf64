// ============================================================================
// oracle/spiral_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see spiral_oracle.h).
// Scalar CPU restatement of lib/spiral-rs; citations are file:line under
// /root/reference/lib/spiral-rs/src/.
// ============================================================================
#include "spiral_oracle.h"

#include <immintrin.h>  // the reference's AVX2 bodies, restated with the same intrinsics (set_avx2_bodies)

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace oracle {

// params.rs:8-46
const u64 Q2_VALUES[37] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12289, 12289, 61441, 65537,
                           65537, 520193, 786433, 786433, 3604481, 7340033, 16515073, 33292289,
                           67043329, 132120577, 268369921, 469762049, 1073479681, 2013265921,
                           4293918721ULL, 8588886017ULL, 17175674881ULL, 34359214081ULL,
                           68718428161ULL};

#define ORACLE_CHECK(cond)                                                       \
  do {                                                                           \
    if (!(cond)) throw std::runtime_error("oracle check failed: " #cond);        \
  } while (0)

// ============================================================== arith.rs
u64 multiply_uint_mod(u64 a, u64 b, u64 modulus) { return (u64)(((u128)a * (u128)b) % (u128)modulus); }

u64 log2_floor(u64 a) { return 63 - (u64)__builtin_clzll(a); }

// Rust's float -> integer `as` casts saturate: ceil(log2(0.0)) = -inf casts to 0 (relevant for
// stop_round() when nu_2 == 0); the same cast is undefined behaviour in C++, so spell it out.
u64 log2_ceil(u64 a) { return a == 0 ? 0 : (u64)std::ceil(std::log2((double)a)); }

static size_t log2_ceil_usize(size_t a) { return a == 0 ? 0 : (size_t)std::ceil(std::log2((double)a)); }

// arith.rs:41-67
u64 exponentiate_uint_mod(u64 operand, u64 exponent, u64 modulus) {
  if (exponent == 0) return 1;
  if (exponent == 1) return operand;
  u64 power = operand, intermediate = 1;
  for (;;) {
    if (exponent & 1) intermediate = multiply_uint_mod(power, intermediate, modulus);
    exponent >>= 1;
    if (exponent == 0) break;
    power = multiply_uint_mod(power, power, modulus);
  }
  return intermediate;
}

// arith.rs:69-76
u64 reverse_bits(u64 x, size_t bit_count) {
  if (bit_count == 0) return 0;
  u64 r = 0;
  for (int i = 0; i < 64; i++) r |= ((x >> i) & 1ULL) << (63 - i);
  return r >> (64 - bit_count);
}

// arith.rs:78-89
u64 div2_uint_mod(u64 operand, u64 modulus) {
  if (operand & 1) {
    u64 res = operand + modulus;
    bool overflow = res < operand;
    if (overflow) return (res >> 1) | (1ULL << 63);
    return res >> 1;
  }
  return operand >> 1;
}

// arith.rs:91-104
u64 recenter(u64 val, u64 from_modulus, u64 to_modulus) {
  ORACLE_CHECK(from_modulus >= to_modulus);
  int64_t from_i = (int64_t)from_modulus, to_i = (int64_t)to_modulus;
  int64_t a_val = (int64_t)val;
  if (val >= from_modulus / 2) a_val -= from_i;
  a_val = a_val + (from_i / to_i) * to_i + 2 * to_i;
  a_val %= to_i;
  return (u64)a_val;
}

// arith.rs:335-413.  The reference ports SEAL's shift-subtract 192-bit division; the quotient
// and remainder of an integer division are unique, so this is the same function computed by
// schoolbook limb division.  Pinned by the two vectors of arith.rs:462-474.
void divide_uint192_inplace(const u64 numerator[3], u64 denominator, u64 rem[3], u64 quot[3]) {
  u128 r = 0;
  for (int i = 2; i >= 0; i--) {
    u128 cur = (r << 64) | numerator[i];
    quot[i] = (u64)(cur / denominator);
    r = cur % denominator;
  }
  rem[0] = (u64)r;
  rem[1] = 0;
  rem[2] = 0;
}

// arith.rs:106-111
void get_barrett_crs(u64 modulus, u64* cr0, u64* cr1) {
  u64 numerator[3] = {0, 0, 1}, rem[3], quot[3];
  divide_uint192_inplace(numerator, modulus, rem, quot);
  *cr0 = quot[0];
  *cr1 = quot[1];
}

// arith.rs:122-134
u64 barrett_raw_u64(u64 input, u64 const_ratio_1, u64 modulus) {
  u64 tmp = (u64)(((u128)input * (u128)const_ratio_1) >> 64);
  u64 res = input - tmp * modulus;
  return res >= modulus ? res - modulus : res;
}

static inline u64 barrett_coeff_u64(const Params& p, u64 val, size_t n) {  // arith.rs:140-142
  return barrett_raw_u64(val, p.barrett_cr_1[n], p.moduli[n]);
}

// arith.rs:155-163: NOTE the quirk -- on overflow `*out` is left untouched (not the wrapped sum).
static inline u64 add_u64_quirk(u64 op1, u64 op2, u64* out) {
  u64 s = op1 + op2;
  if (s < op1) return 1;  // checked_add == None
  *out = s;
  return 0;
}

// arith.rs:165-180 (literal, including the add_u64 quirk)
u64 barrett_raw_u128(u128 val, u64 cr0, u64 cr1, u64 modulus) {
  u64 zx = (u64)val, zy = (u64)(val >> 64);
  u64 tmp1 = 0, tmp3, carry;
  u128 prod = (u128)zx * cr0;
  carry = (u64)(prod >> 64);
  u128 tmp2 = (u128)zx * cr1;
  u64 tmp2x = (u64)tmp2, tmp2y = (u64)(tmp2 >> 64);
  tmp3 = tmp2y + add_u64_quirk(tmp2x, carry, &tmp1);
  tmp2 = (u128)zy * cr0;
  tmp2x = (u64)tmp2;
  tmp2y = (u64)(tmp2 >> 64);
  carry = tmp2y + add_u64_quirk(tmp1, tmp2x, &tmp1);
  tmp1 = zy * cr1 + tmp3 + carry;
  tmp3 = zx - tmp1 * modulus;
  return tmp3;
}

// arith.rs:198-202
u64 barrett_reduction_u128_raw(u64 modulus, u64 cr0, u64 cr1, u128 val) {
  u64 reduced_val = barrett_raw_u128(val, cr0, cr1, modulus);
  reduced_val -= modulus * (u64)(reduced_val >= modulus);
  return reduced_val;
}

// arith.rs:415-427
u64 recenter_mod(u64 val, u64 small_modulus, u64 large_modulus) {
  ORACLE_CHECK(val < small_modulus);
  int64_t v = (int64_t)val, s = (int64_t)small_modulus, l = (int64_t)large_modulus;
  if (v > s / 2) v -= s;
  if (v < 0) v += l;
  return (u64)v;
}

// arith.rs:429-444 (Rust i128 `/` and `%` truncate toward zero, as C++ does)
u64 rescale(u64 a, u64 inp_mod, u64 out_mod) {
  int64_t inp_mod_i64 = (int64_t)inp_mod;
  i128 out_mod_i128 = (i128)out_mod;
  int64_t inp_val = (int64_t)(a % inp_mod);
  if (inp_val >= inp_mod_i64 / 2) inp_val -= inp_mod_i64;
  int64_t sign = inp_val >= 0 ? 1 : -1;
  i128 val = (i128)inp_val * (i128)out_mod;
  i128 result = (val + (i128)(sign * (inp_mod_i64 / 2))) / (i128)inp_mod;
  result = (result + (i128)((inp_mod / out_mod) * out_mod) + 2 * out_mod_i128) % out_mod_i128;
  ORACLE_CHECK(result >= 0);
  return (u64)((result + out_mod_i128) % out_mod_i128);
}

// ====================================================== number_theory.rs
static bool is_primitive_root(u64 root, u64 degree, u64 modulus) {  // number_theory.rs:6-12
  if (root == 0) return false;
  return exponentiate_uint_mod(root, degree >> 1, modulus) == modulus - 1;
}

// number_theory.rs:14-39 draws random candidates; :41-55 then minimises over all primitive roots
// of that degree, so the result does not depend on which one was found first.
u64 get_minimal_primitive_root(u64 degree, u64 modulus) {
  u64 size_entire_group = modulus - 1;
  u64 size_quotient_group = size_entire_group / degree;
  ORACLE_CHECK(size_entire_group - size_quotient_group * degree == 0);
  u64 root = 0;
  for (u64 cand = 2;; cand++) {
    root = exponentiate_uint_mod(cand, size_quotient_group, modulus);
    if (is_primitive_root(root, degree, modulus)) break;
    ORACLE_CHECK(cand < 100000);
  }
  u64 generator_sq = multiply_uint_mod(root, root, modulus);
  u64 current_generator = root;
  for (u64 i = 0; i < degree; i++) {
    if (current_generator < root) root = current_generator;
    current_generator = multiply_uint_mod(current_generator, generator_sq, modulus);
  }
  return root;
}

// number_theory.rs:57-96
u64 invert_uint_mod(u64 value, u64 modulus) {
  ORACLE_CHECK(value != 0);
  u64 x = value, y = modulus;
  int64_t prev_a = 1, a = 0;
  while (y != 0) {
    int64_t q = (int64_t)(x / y);
    u64 temp = x % y;
    x = y;
    y = temp;
    int64_t t = a;
    a = prev_a - q * a;
    prev_a = t;
  }
  ORACLE_CHECK(x == 1);
  if (prev_a < 0) return (u64)prev_a + modulus;
  return (u64)prev_a;
}

// ================================================================ ntt.rs
static std::vector<u64> powers_of_primitive_root(u64 root, u64 modulus, size_t poly_len_log2) {  // :6-17
  size_t poly_len = (size_t)1 << poly_len_log2;
  std::vector<u64> root_powers(poly_len, 0);
  u64 power = root;
  for (size_t i = 1; i < poly_len; i++) {
    size_t idx = (size_t)reverse_bits(i, poly_len_log2);
    root_powers[idx] = power;
    power = multiply_uint_mod(power, root, modulus);
  }
  root_powers[0] = 1;
  return root_powers;
}

static std::vector<u64> scale_powers_u32(u32 modulus, size_t poly_len, const std::vector<u64>& inp) {  // :29-37
  std::vector<u64> scaled(poly_len, 0);
  for (size_t i = 0; i < poly_len; i++) {
    u64 wide_val = inp[i] << 32;
    u64 quotient = wide_val / (u64)modulus;
    scaled[i] = (u64)(u32)quotient;
  }
  return scaled;
}

// ntt.rs:39-65
std::vector<std::vector<std::vector<u64>>> build_ntt_tables(size_t poly_len, const std::vector<u64>& moduli) {
  size_t poly_len_log2 = (size_t)log2_floor(poly_len);
  std::vector<std::vector<std::vector<u64>>> output(moduli.size());
  for (size_t coeff_mod = 0; coeff_mod < moduli.size(); coeff_mod++) {
    u64 modulus = moduli[coeff_mod];
    ORACLE_CHECK(modulus <= 0xFFFFFFFFULL);
    u32 modulus_as_u32 = (u32)modulus;
    u64 root = get_minimal_primitive_root(2 * (u64)poly_len, modulus);
    u64 inv_root = invert_uint_mod(root, modulus);
    std::vector<u64> root_powers = powers_of_primitive_root(root, modulus, poly_len_log2);
    std::vector<u64> scaled_root_powers = scale_powers_u32(modulus_as_u32, poly_len, root_powers);
    std::vector<u64> inv_root_powers = powers_of_primitive_root(inv_root, modulus, poly_len_log2);
    for (size_t i = 0; i < poly_len; i++) inv_root_powers[i] = div2_uint_mod(inv_root_powers[i], modulus);
    std::vector<u64> scaled_inv_root_powers = scale_powers_u32(modulus_as_u32, poly_len, inv_root_powers);
    output[coeff_mod] = {root_powers, scaled_root_powers, inv_root_powers, scaled_inv_root_powers};
  }
  return output;
}

// ---------------------------------------------------------------------------------------------------------------------
// Which of the reference's two sets of bodies runs.  spiral-rs selects at COMPILE time (`cfg(target_feature = "avx2")`,
// ntt.rs:66/115/212/260, poly.rs:407/437/460); here it is a run-time switch so that one test process can compare the two:
// 0 = the `cfg(not(target_feature = "avx2"))` bodies (default), 1 = the AVX2 bodies restated with the same `_mm256_*`
// intrinsics (ntt_forward_avx2 / ntt_inverse_avx2 / multiply_avx2 below).  The two differ in representatives, not in
// residues: the AVX2 forward transform corrects with strict compares (`_mm256_cmpgt_epi64`), so it can leave q where the
// scalar body leaves 0, and the AVX2 multiply accumulates unreduced and reduces once.
// ---------------------------------------------------------------------------------------------------------------------
static int g_avx2_bodies = 0;
void set_avx2_bodies(int on) { g_avx2_bodies = on ? 1 : 0; }
int get_avx2_bodies() { return g_avx2_bodies; }
static void ntt_forward_avx2(const Params& params, u64* operand_overall);
static void ntt_inverse_avx2(const Params& params, u64* operand_overall);

// ntt.rs:67-113 (scalar body)
void ntt_forward(const Params& params, u64* operand_overall) {
  if (g_avx2_bodies) return ntt_forward_avx2(params, operand_overall);
  size_t log_n = params.poly_len_log2;
  size_t n = (size_t)1 << log_n;
  for (size_t coeff_mod = 0; coeff_mod < params.crt_count; coeff_mod++) {
    u64* operand = operand_overall + coeff_mod * n;
    const u64* forward_table = params.get_ntt_forward_table(coeff_mod);
    const u64* forward_table_prime = params.get_ntt_forward_prime_table(coeff_mod);
    u32 modulus_small = (u32)params.moduli[coeff_mod];
    u32 two_times_modulus_small = 2 * modulus_small;
    for (size_t mm = 0; mm < log_n; mm++) {
      size_t m = (size_t)1 << mm;
      size_t t = n >> (mm + 1);
      for (size_t i = 0; i < m; i++) {
        u64 w = forward_table[m + i];
        u64 w_prime = forward_table_prime[m + i];
        u64* op = operand + i * 2 * t;
        for (size_t j = 0; j < t; j++) {
          u32 x = (u32)op[j];
          u32 y = (u32)op[t + j];
          u32 curr_x = x - (two_times_modulus_small * (u32)(x >= two_times_modulus_small));
          u64 q_tmp = ((u64)y * w_prime) >> 32;
          u64 q_new = w * (u64)y - q_tmp * (u64)modulus_small;
          op[j] = (u64)curr_x + q_new;
          op[t + j] = (u64)curr_x + ((u64)two_times_modulus_small - q_new);
        }
      }
    }
    for (size_t i = 0; i < n; i++) {
      operand[i] -= (u64)(operand[i] >= (u64)two_times_modulus_small) * (u64)two_times_modulus_small;
      operand[i] -= (u64)(operand[i] >= (u64)modulus_small) * (u64)modulus_small;
    }
  }
}

// ntt.rs:212-258 (scalar body)
void ntt_inverse(const Params& params, u64* operand_overall) {
  if (g_avx2_bodies) return ntt_inverse_avx2(params, operand_overall);
  for (size_t coeff_mod = 0; coeff_mod < params.crt_count; coeff_mod++) {
    size_t n = params.poly_len;
    u64* operand = operand_overall + coeff_mod * n;
    const u64* inverse_table = params.get_ntt_inverse_table(coeff_mod);
    const u64* inverse_table_prime = params.get_ntt_inverse_prime_table(coeff_mod);
    u64 modulus = params.moduli[coeff_mod];
    u64 two_times_modulus = 2 * modulus;
    for (size_t mm = params.poly_len_log2; mm-- > 0;) {
      size_t h = (size_t)1 << mm;
      size_t t = n >> (mm + 1);
      for (size_t i = 0; i < h; i++) {
        u64 w = inverse_table[h + i];
        u64 w_prime = inverse_table_prime[h + i];
        u64* op = operand + i * 2 * t;
        for (size_t j = 0; j < t; j++) {
          u64 x = op[j];
          u64 y = op[t + j];
          u64 t_tmp = two_times_modulus - y + x;
          u64 curr_x = x + y - (two_times_modulus * (u64)((x << 1) >= t_tmp));
          u64 h_tmp = (t_tmp * w_prime) >> 32;
          u64 res_x = (curr_x + (modulus * (u64)(t_tmp & 1))) >> 1;
          u64 res_y = w * t_tmp - h_tmp * modulus;
          op[j] = res_x;
          op[t + j] = res_y;
        }
      }
    }
    for (size_t i = 0; i < n; i++) {
      operand[i] -= (u64)(operand[i] >= two_times_modulus) * two_times_modulus;
      operand[i] -= (u64)(operand[i] >= modulus) * modulus;
    }
  }
}


// ---- the reference's AVX2 bodies (`cfg(target_feature = "avx2")`), intrinsic for intrinsic.  The reference's buffers are
// 64-byte aligned (AlignedMemory64) and it uses the aligned load / store forms; the restatement's buffers are std::vectors, so
// the unaligned forms stand in for them (same values).  `_mm256_cmpgt_epi64` is a SIGNED compare, as in the reference; every
// operand here is far below 2^63.
// ntt.rs:115-210
static void ntt_forward_avx2(const Params& params, u64* operand_overall) {
  size_t log_n = params.poly_len_log2;
  size_t n = (size_t)1 << log_n;
  for (size_t coeff_mod = 0; coeff_mod < params.crt_count; coeff_mod++) {
    u64* operand = operand_overall + coeff_mod * n;
    const u64* forward_table = params.get_ntt_forward_table(coeff_mod);
    const u64* forward_table_prime = params.get_ntt_forward_prime_table(coeff_mod);
    u32 modulus_small = (u32)params.moduli[coeff_mod];
    u32 two_times_modulus_small = 2 * modulus_small;
    for (size_t mm = 0; mm < log_n; mm++) {
      size_t m = (size_t)1 << mm;
      size_t t = n >> (mm + 1);
      for (size_t i = 0; i < m; i++) {
        u64 w = forward_table[m + i];
        u64 w_prime = forward_table_prime[m + i];
        u64* op = operand + i * 2 * t;
        if (t < 4) {  // ntt.rs:142-154: the scalar butterfly for the last two stages
          for (size_t j = 0; j < t; j++) {
            u32 x = (u32)op[j];
            u32 y = (u32)op[t + j];
            u32 curr_x = x - (two_times_modulus_small * (u32)(x >= two_times_modulus_small));
            u64 q_tmp = ((u64)y * w_prime) >> 32;
            u64 q_new = w * (u64)y - q_tmp * (u64)modulus_small;
            op[j] = (u64)curr_x + q_new;
            op[t + j] = (u64)curr_x + ((u64)two_times_modulus_small - q_new);
          }
        } else {      // ntt.rs:156-188
          for (size_t j = 0; j < t; j += 4) {
            __m256i* p_x = (__m256i*)&op[j];
            __m256i* p_y = (__m256i*)&op[j + t];
            __m256i x = _mm256_loadu_si256(p_x);
            __m256i y = _mm256_loadu_si256(p_y);
            __m256i cmp_val = _mm256_set1_epi64x((long long)two_times_modulus_small);
            __m256i gt_mask = _mm256_cmpgt_epi64(x, cmp_val);
            __m256i to_subtract = _mm256_and_si256(gt_mask, cmp_val);
            __m256i curr_x = _mm256_sub_epi64(x, to_subtract);
            __m256i w_prime_vec = _mm256_set1_epi64x((long long)w_prime);
            __m256i product = _mm256_mul_epu32(y, w_prime_vec);
            __m256i q_val = _mm256_srli_epi64(product, 32);
            __m256i w_vec = _mm256_set1_epi64x((long long)w);
            __m256i w_times_y = _mm256_mul_epu32(y, w_vec);
            __m256i modulus_small_vec = _mm256_set1_epi64x((long long)modulus_small);
            __m256i q_scaled = _mm256_mul_epu32(q_val, modulus_small_vec);
            __m256i q_final = _mm256_sub_epi64(w_times_y, q_scaled);
            __m256i new_x = _mm256_add_epi64(curr_x, q_final);
            __m256i q_final_inverted = _mm256_sub_epi64(cmp_val, q_final);
            __m256i new_y = _mm256_add_epi64(curr_x, q_final_inverted);
            _mm256_storeu_si256(p_x, new_x);
            _mm256_storeu_si256(p_y, new_y);
          }
        }
      }
    }
    for (size_t i = 0; i < n; i += 4) {  // ntt.rs:193-208: strict compares -- 2q and q stay where the scalar body yields 0
      __m256i* p_x = (__m256i*)&operand[i];
      __m256i cmp_val1 = _mm256_set1_epi64x((long long)two_times_modulus_small);
      __m256i x = _mm256_loadu_si256(p_x);
      __m256i gt_mask = _mm256_cmpgt_epi64(x, cmp_val1);
      __m256i to_subtract = _mm256_and_si256(gt_mask, cmp_val1);
      x = _mm256_sub_epi64(x, to_subtract);
      __m256i cmp_val2 = _mm256_set1_epi64x((long long)modulus_small);
      gt_mask = _mm256_cmpgt_epi64(x, cmp_val2);
      to_subtract = _mm256_and_si256(gt_mask, cmp_val2);
      x = _mm256_sub_epi64(x, to_subtract);
      _mm256_storeu_si256(p_x, x);
    }
  }
}

// ntt.rs:260-365
static void ntt_inverse_avx2(const Params& params, u64* operand_overall) {
  for (size_t coeff_mod = 0; coeff_mod < params.crt_count; coeff_mod++) {
    size_t n = params.poly_len;
    u64* operand = operand_overall + coeff_mod * n;
    const u64* inverse_table = params.get_ntt_inverse_table(coeff_mod);
    const u64* inverse_table_prime = params.get_ntt_inverse_prime_table(coeff_mod);
    u64 modulus = params.moduli[coeff_mod];
    u64 two_times_modulus = 2 * modulus;
    for (size_t mm = params.poly_len_log2; mm-- > 0;) {
      size_t h = (size_t)1 << mm;
      size_t t = n >> (mm + 1);
      for (size_t i = 0; i < h; i++) {
        u64 w = inverse_table[h + i];
        u64 w_prime = inverse_table_prime[h + i];
        u64* op = operand + i * 2 * t;
        if (t < 4) {  // ntt.rs:283-296
          for (size_t j = 0; j < t; j++) {
            u64 x = op[j];
            u64 y = op[t + j];
            u64 t_tmp = two_times_modulus - y + x;
            u64 curr_x = x + y - (two_times_modulus * (u64)((x << 1) >= t_tmp));
            u64 h_tmp = (t_tmp * w_prime) >> 32;
            u64 res_x = (curr_x + (modulus * (u64)(t_tmp & 1))) >> 1;
            u64 res_y = w * t_tmp - h_tmp * modulus;
            op[j] = res_x;
            op[t + j] = res_y;
          }
        } else {      // ntt.rs:298-338
          for (size_t j = 0; j < t; j += 4) {
            __m256i* p_x = (__m256i*)&op[j];
            __m256i* p_y = (__m256i*)&op[j + t];
            __m256i x = _mm256_loadu_si256(p_x);
            __m256i y = _mm256_loadu_si256(p_y);
            __m256i modulus_vec = _mm256_set1_epi64x((long long)modulus);
            __m256i two_times_modulus_vec = _mm256_set1_epi64x((long long)two_times_modulus);
            __m256i t_tmp = _mm256_set1_epi64x((long long)two_times_modulus);
            t_tmp = _mm256_sub_epi64(t_tmp, y);
            t_tmp = _mm256_add_epi64(t_tmp, x);
            __m256i gt_mask = _mm256_cmpgt_epi64(_mm256_slli_epi64(x, 1), t_tmp);
            __m256i to_subtract = _mm256_and_si256(gt_mask, two_times_modulus_vec);
            __m256i curr_x = _mm256_add_epi64(x, y);
            curr_x = _mm256_sub_epi64(curr_x, to_subtract);
            __m256i w_prime_vec = _mm256_set1_epi64x((long long)w_prime);
            __m256i h_tmp = _mm256_mul_epu32(t_tmp, w_prime_vec);
            h_tmp = _mm256_srli_epi64(h_tmp, 32);
            __m256i and_mask = _mm256_set_epi64x(1, 1, 1, 1);
            __m256i eq_mask = _mm256_cmpeq_epi64(_mm256_and_si256(t_tmp, and_mask), and_mask);
            __m256i to_add = _mm256_and_si256(eq_mask, modulus_vec);
            __m256i new_x = _mm256_srli_epi64(_mm256_add_epi64(curr_x, to_add), 1);
            __m256i w_vec = _mm256_set1_epi64x((long long)w);
            __m256i w_times_t_tmp = _mm256_mul_epu32(t_tmp, w_vec);
            __m256i h_tmp_times_modulus = _mm256_mul_epu32(h_tmp, modulus_vec);
            __m256i new_y = _mm256_sub_epi64(w_times_t_tmp, h_tmp_times_modulus);
            _mm256_storeu_si256(p_x, new_x);
            _mm256_storeu_si256(p_y, new_y);
          }
        }
      }
    }
    for (size_t i = 0; i < n; i++) {  // ntt.rs:343-346 (the vector form of this loop is commented out in the reference)
      operand[i] -= (u64)(operand[i] >= two_times_modulus) * two_times_modulus;
      operand[i] -= (u64)(operand[i] >= modulus) * modulus;
    }
  }
}

// ============================================================= params.rs
size_t Params::g() const {  // params.rs:129-132
  size_t num_bits_to_gen = t_gsw * db_dim_2 + num_expanded();
  return log2_ceil_usize(num_bits_to_gen);
}
size_t Params::stop_round() const { return log2_ceil_usize(t_gsw * db_dim_2); }  // params.rs:134-136

size_t Params::setup_bytes() const {  // params.rs:146-167
  size_t sz_polys = 0;
  size_t num_packing_mats = version == 0 ? n : 2;
  size_t packing_sz = ((n + 1) - 1) * t_conv;
  sz_polys += num_packing_mats * packing_sz;
  if (expand_queries) {
    size_t expansion_left_sz = g() * t_exp_left;
    size_t expansion_right_sz = (stop_round() + 1) * t_exp_right;
    size_t conversion_sz = 2 * t_conv;
    if (version > 0 && t_exp_left == t_exp_right) expansion_right_sz = 0;
    sz_polys += expansion_left_sz + expansion_right_sz + conversion_sz;
  }
  return SEED_LENGTH + sz_polys * poly_len * sizeof(u64);
}

size_t Params::query_bytes() const {  // params.rs:169-182
  size_t sz_polys;
  if (expand_queries) {
    sz_polys = 1;
  } else {
    sz_polys = num_expanded() + db_dim_2 * (2 * t_gsw);
  }
  return SEED_LENGTH + sz_polys * poly_len * sizeof(u64);
}

size_t Params::bytes_per_chunk() const {  // params.rs:188-193
  size_t chunks = instances * n * n;
  return (size_t)std::ceil((double)db_item_size / (double)chunks);
}

size_t Params::modp_words_per_chunk() const {  // params.rs:195-200
  size_t bpc = bytes_per_chunk();
  u64 logp = log2_floor(pt_modulus);
  return (size_t)std::ceil((double)(bpc * 8) / (double)logp);
}

u64 Params::crt_compose_2(u64 x, u64 y) const {  // params.rs:207-214
  ORACLE_CHECK(crt_count == 2);
  u128 val = (u128)x * (u128)mod1_inv_mod0;
  val += (u128)y * (u128)mod0_inv_mod1;
  return barrett_reduction_u128_raw(modulus, barrett_cr_0_modulus, barrett_cr_1_modulus, val);
}

u64 Params::crt_compose(const u64* a, size_t idx) const {  // params.rs:216-222
  if (crt_count == 1) return a[idx];
  return crt_compose_2(a[idx], a[idx + poly_len]);
}

// params.rs:224-296
Params Params::init(size_t poly_len, const std::vector<u64>& moduli, double noise_width, size_t n,
                    u64 pt_modulus, u64 q2_bits, size_t t_conv, size_t t_exp_left, size_t t_exp_right,
                    size_t t_gsw, bool expand_queries, size_t db_dim_1, size_t db_dim_2,
                    size_t instances, size_t db_item_size, size_t version) {
  ORACLE_CHECK(q2_bits >= 14);
  Params p;
  p.poly_len = poly_len;
  p.poly_len_log2 = (size_t)log2_floor(poly_len);
  p.crt_count = moduli.size();
  ORACLE_CHECK(p.crt_count <= MAX_MODULI);
  for (size_t i = 0; i < p.crt_count; i++) p.moduli[i] = moduli[i];
  p.ntt_tables = build_ntt_tables(poly_len, moduli);
  u64 modulus = 1;
  for (u64 m : moduli) modulus *= m;
  p.modulus = modulus;
  p.modulus_log2 = log2_ceil(modulus);
  for (size_t i = 0; i < p.crt_count; i++) get_barrett_crs(moduli[i], &p.barrett_cr_0[i], &p.barrett_cr_1[i]);
  get_barrett_crs(modulus, &p.barrett_cr_0_modulus, &p.barrett_cr_1_modulus);
  if (p.crt_count == 2) {
    p.mod0_inv_mod1 = moduli[0] * invert_uint_mod(moduli[0], moduli[1]);
    p.mod1_inv_mod0 = moduli[1] * invert_uint_mod(moduli[1], moduli[0]);
  }
  p.noise_width = noise_width;
  p.n = n;
  p.pt_modulus = pt_modulus;
  p.q2_bits = q2_bits;
  p.t_conv = t_conv;
  p.t_exp_left = t_exp_left;
  p.t_exp_right = t_exp_right;
  p.t_gsw = t_gsw;
  p.expand_queries = expand_queries;
  p.db_dim_1 = db_dim_1;
  p.db_dim_2 = db_dim_2;
  p.instances = instances;
  p.db_item_size = db_item_size;
  p.version = version;
  return p;
}

// util.rs:224-263
Params params_from_fields(size_t n, size_t nu_1, size_t nu_2, u64 p, u64 q2_bits, size_t t_gsw,
                          size_t t_conv, size_t t_exp_left, size_t t_exp_right, size_t instances,
                          size_t db_item_size, size_t version, bool direct_upload) {
  q2_bits = std::max<u64>(q2_bits, 14);
  if (instances == 0) instances = 1;
  if (db_item_size == 0) {
    db_item_size = instances * n * n;
    db_item_size = db_item_size * 2048 * (size_t)log2_ceil(p) / 8;
  }
  return Params::init(2048, {268369921ULL, 249561089ULL}, 6.4, n, p, q2_bits, t_conv, t_exp_left,
                      t_exp_right, t_gsw, !direct_upload, nu_1, nu_2, instances, db_item_size, version);
}

// =============================================================== poly.rs
void PolyMatrixRaw::copy_into(const PolyMatrixRaw& p, size_t target_row, size_t target_col) {  // :41-53
  ORACLE_CHECK(target_row < rows && target_col < cols);
  ORACLE_CHECK(target_row + p.rows <= rows && target_col + p.cols <= cols);
  for (size_t r = 0; r < p.rows; r++)
    for (size_t c = 0; c < p.cols; c++)
      memcpy(get_poly(target_row + r, target_col + c), p.get_poly(r, c), num_words() * sizeof(u64));
}
void PolyMatrixNTT::copy_into(const PolyMatrixNTT& p, size_t target_row, size_t target_col) {
  ORACLE_CHECK(target_row < rows && target_col < cols);
  ORACLE_CHECK(target_row + p.rows <= rows && target_col + p.cols <= cols);
  for (size_t r = 0; r < p.rows; r++)
    for (size_t c = 0; c < p.cols; c++)
      memcpy(get_poly(target_row + r, target_col + c), p.get_poly(r, c), num_words() * sizeof(u64));
}
PolyMatrixRaw PolyMatrixRaw::submatrix(size_t tr, size_t tc, size_t r_, size_t c_) const {  // :127-141
  PolyMatrixRaw m(params, r_, c_);
  ORACLE_CHECK(tr < rows && tc < cols && tr + r_ <= rows && tc + c_ <= cols);
  for (size_t r = 0; r < r_; r++)
    for (size_t c = 0; c < c_; c++) memcpy(m.get_poly(r, c), get_poly(tr + r, tc + c), num_words() * sizeof(u64));
  return m;
}
PolyMatrixNTT PolyMatrixNTT::submatrix(size_t tr, size_t tc, size_t r_, size_t c_) const {  // :302-316
  PolyMatrixNTT m(params, r_, c_);
  ORACLE_CHECK(tr < rows && tc < cols && tr + r_ <= rows && tc + c_ <= cols);
  for (size_t r = 0; r < r_; r++)
    for (size_t c = 0; c < c_; c++) memcpy(m.get_poly(r, c), get_poly(tr + r, tc + c), num_words() * sizeof(u64));
  return m;
}
PolyMatrixNTT PolyMatrixNTT::pad_top(size_t pad_rows) const {  // :296-300
  PolyMatrixNTT padded(params, rows + pad_rows, cols);
  padded.copy_into(*this, pad_rows, 0);
  return padded;
}

// poly.rs:213-235
std::vector<uint8_t> PolyMatrixRaw::to_vec(size_t modulus_bits, size_t num_coeffs) const {
  size_t sz_bits = rows * cols * num_coeffs * modulus_bits;
  size_t sz_bytes = (size_t)std::ceil((double)sz_bits / 8.0) + 32;
  size_t sz_bytes_roundup_16 = ((sz_bytes + 15) / 16) * 16;
  std::vector<uint8_t> data(sz_bytes_roundup_16, 0);
  size_t bit_offs = 0;
  for (size_t r = 0; r < rows; r++)
    for (size_t c = 0; c < cols; c++) {
      for (size_t z = 0; z < num_coeffs; z++) {
        write_arbitrary_bits(data.data(), get_poly(r, c)[z], bit_offs, modulus_bits);
        bit_offs += modulus_bits;
      }
      bit_offs = (bit_offs / 8) * 8;
    }
  return data;
}

// poly.rs:360-367: multiply_add_modular = barrett(a*b + x)  (arith.rs:25-27)
static void multiply_add_poly(const Params& params, u64* res, const u64* a, const u64* b) {
  for (size_t c = 0; c < params.crt_count; c++)
    for (size_t i = 0; i < params.poly_len; i++) {
      size_t idx = c * params.poly_len + i;
      res[idx] = barrett_coeff_u64(params, a[idx] * b[idx] + res[idx], c);
    }
}
static void multiply_poly(const Params& params, u64* res, const u64* a, const u64* b) {  // :351-358
  for (size_t c = 0; c < params.crt_count; c++)
    for (size_t i = 0; i < params.poly_len; i++) {
      size_t idx = c * params.poly_len + i;
      res[idx] = barrett_coeff_u64(params, a[idx] * b[idx], c);
    }
}
static void add_poly(const Params& params, u64* res, const u64* a, const u64* b) {  // :369-376
  for (size_t c = 0; c < params.crt_count; c++)
    for (size_t i = 0; i < params.poly_len; i++) {
      size_t idx = c * params.poly_len + i;
      res[idx] = barrett_coeff_u64(params, a[idx] + b[idx], c);
    }
}
static void invert_poly(const Params& params, u64* res, const u64* a) {  // :387-391
  for (size_t i = 0; i < params.poly_len; i++) res[i] = params.modulus - a[i];
}
static void automorph_poly(const Params& params, u64* res, const u64* a, size_t t) {  // :393-405
  size_t poly_len = params.poly_len;
  for (size_t i = 0; i < poly_len; i++) {
    size_t num = (i * t) / poly_len;
    size_t rem = (i * t) % poly_len;
    if (num % 2 == 0)
      res[rem] = a[i];
    else
      res[rem] = params.modulus - a[i];
  }
}

// poly.rs:437-458 (scalar)
// poly.rs:407-426: products of the low 32-bit halves accumulated WITHOUT reduction
static void multiply_add_poly_avx(const Params& params, u64* res, const u64* a, const u64* b) {
  for (size_t c = 0; c < params.crt_count; c++)
    for (size_t i = 0; i < params.poly_len; i += 4) {
      const __m256i* p_x = (const __m256i*)&a[c * params.poly_len + i];
      const __m256i* p_y = (const __m256i*)&b[c * params.poly_len + i];
      __m256i* p_z = (__m256i*)&res[c * params.poly_len + i];
      __m256i x = _mm256_loadu_si256(p_x);
      __m256i y = _mm256_loadu_si256(p_y);
      __m256i z = _mm256_loadu_si256(p_z);
      __m256i product = _mm256_mul_epu32(x, y);
      __m256i out = _mm256_add_epi64(z, product);
      _mm256_storeu_si256(p_z, out);
    }
}
static void modular_reduce(const Params& params, u64* res) {  // poly.rs:428-435
  for (size_t c = 0; c < params.crt_count; c++)
    for (size_t i = 0; i < params.poly_len; i++) {
      size_t idx = c * params.poly_len + i;
      res[idx] = barrett_coeff_u64(params, res[idx], c);
    }
}
static void multiply_avx2(PolyMatrixNTT& res, const PolyMatrixNTT& a, const PolyMatrixNTT& b) {  // poly.rs:460-481
  ORACLE_CHECK(res.rows == a.rows && res.cols == b.cols && a.cols == b.rows);
  const Params& params = *res.params;
  for (size_t i = 0; i < a.rows; i++)
    for (size_t j = 0; j < b.cols; j++) {
      u64* res_poly = res.get_poly(i, j);
      for (size_t z = 0; z < params.poly_len * params.crt_count; z++) res_poly[z] = 0;
      for (size_t k = 0; k < a.cols; k++) multiply_add_poly_avx(params, res_poly, a.get_poly(i, k), b.get_poly(k, j));
      modular_reduce(params, res_poly);
    }
}
void multiply(PolyMatrixNTT& res, const PolyMatrixNTT& a, const PolyMatrixNTT& b) {  // poly.rs:437-458 (scalar body)
  if (g_avx2_bodies) return multiply_avx2(res, a, b);
  ORACLE_CHECK(res.rows == a.rows && res.cols == b.cols && a.cols == b.rows);
  const Params& params = *res.params;
  for (size_t i = 0; i < a.rows; i++)
    for (size_t j = 0; j < b.cols; j++) {
      u64* res_poly = res.get_poly(i, j);
      for (size_t z = 0; z < params.poly_len * params.crt_count; z++) res_poly[z] = 0;
      for (size_t k = 0; k < a.cols; k++) multiply_add_poly(params, res_poly, a.get_poly(i, k), b.get_poly(k, j));
    }
}
void add(PolyMatrixNTT& res, const PolyMatrixNTT& a, const PolyMatrixNTT& b) {  // :483-498
  ORACLE_CHECK(res.rows == a.rows && res.cols == a.cols && a.rows == b.rows && a.cols == b.cols);
  for (size_t i = 0; i < a.rows; i++)
    for (size_t j = 0; j < a.cols; j++) add_poly(*res.params, res.get_poly(i, j), a.get_poly(i, j), b.get_poly(i, j));
}
void add_into(PolyMatrixNTT& res, const PolyMatrixNTT& a) {  // :500-512
  ORACLE_CHECK(res.rows == a.rows && res.cols == a.cols);
  for (size_t i = 0; i < res.rows; i++)
    for (size_t j = 0; j < res.cols; j++) add_poly(*res.params, res.get_poly(i, j), res.get_poly(i, j), a.get_poly(i, j));
}
void add_into_at(PolyMatrixNTT& res, const PolyMatrixNTT& a, size_t t_row, size_t t_col) {  // :514-523
  for (size_t i = 0; i < a.rows; i++)
    for (size_t j = 0; j < a.cols; j++) {
      u64* rp = res.get_poly(t_row + i, t_col + j);
      add_poly(*res.params, rp, rp, a.get_poly(i, j));
    }
}
void invert(PolyMatrixRaw& res, const PolyMatrixRaw& a) {  // :525-537
  ORACLE_CHECK(res.rows == a.rows && res.cols == a.cols);
  for (size_t i = 0; i < a.rows; i++)
    for (size_t j = 0; j < a.cols; j++) invert_poly(*res.params, res.get_poly(i, j), a.get_poly(i, j));
}
void automorph(PolyMatrixRaw& res, const PolyMatrixRaw& a, size_t t) {  // :539-551
  ORACLE_CHECK(res.rows == a.rows && res.cols == a.cols);
  for (size_t i = 0; i < a.rows; i++)
    for (size_t j = 0; j < a.cols; j++) automorph_poly(*res.params, res.get_poly(i, j), a.get_poly(i, j), t);
}
PolyMatrixRaw stack(const PolyMatrixRaw& a, const PolyMatrixRaw& b) {  // :559-565
  ORACLE_CHECK(a.cols == b.cols);
  PolyMatrixRaw c(a.params, a.rows + b.rows, a.cols);
  c.copy_into(a, 0, 0);
  c.copy_into(b, a.rows, 0);
  return c;
}
void scalar_multiply(PolyMatrixNTT& res, const PolyMatrixNTT& a, const PolyMatrixNTT& b) {  // :575-588
  ORACLE_CHECK(a.rows == 1 && a.cols == 1);
  const u64* pol2 = a.get_poly(0, 0);
  for (size_t i = 0; i < b.rows; i++)
    for (size_t j = 0; j < b.cols; j++) multiply_poly(*res.params, res.get_poly(i, j), b.get_poly(i, j), pol2);
}
PolyMatrixRaw single_poly(const Params& params, u64 val) {  // :599-603
  PolyMatrixRaw res(&params, 1, 1);
  res.data[0] = val;
  return res;
}
static void reduce_copy(const Params& params, u64* out, const u64* inp) {  // :605-611
  for (size_t n = 0; n < params.crt_count; n++)
    for (size_t z = 0; z < params.poly_len; z++) out[n * params.poly_len + z] = barrett_coeff_u64(params, inp[z], n);
}
void to_ntt(PolyMatrixNTT& a, const PolyMatrixRaw& b) {  // :613-623
  const Params& params = *a.params;
  for (size_t r = 0; r < a.rows; r++)
    for (size_t c = 0; c < a.cols; c++) {
      reduce_copy(params, a.get_poly(r, c), b.get_poly(r, c));
      ntt_forward(params, a.get_poly(r, c));
    }
}
void to_ntt_no_reduce(PolyMatrixNTT& a, const PolyMatrixRaw& b) {  // :625-638
  const Params& params = *a.params;
  for (size_t r = 0; r < a.rows; r++)
    for (size_t c = 0; c < a.cols; c++) {
      const u64* src = b.get_poly(r, c);
      u64* dst = a.get_poly(r, c);
      for (size_t n = 0; n < params.crt_count; n++) memcpy(dst + n * params.poly_len, src, params.poly_len * sizeof(u64));
      ntt_forward(params, dst);
    }
}
PolyMatrixNTT to_ntt_alloc(const PolyMatrixRaw& b) {  // :640-644
  PolyMatrixNTT a(b.params, b.rows, b.cols);
  to_ntt(a, b);
  return a;
}
void from_ntt(PolyMatrixRaw& a, const PolyMatrixNTT& b) {  // :646-663
  const Params& params = *a.params;
  std::vector<u64> scratch(params.crt_count * params.poly_len);
  for (size_t r = 0; r < a.rows; r++)
    for (size_t c = 0; c < a.cols; c++) {
      memcpy(scratch.data(), b.get_poly(r, c), scratch.size() * sizeof(u64));
      ntt_inverse(params, scratch.data());
      u64* dst = a.get_poly(r, c);
      for (size_t z = 0; z < params.poly_len; z++) dst[z] = params.crt_compose(scratch.data(), z);
    }
}
PolyMatrixRaw from_ntt_alloc(const PolyMatrixNTT& b) {  // :665-669
  PolyMatrixRaw a(b.params, b.rows, b.cols);
  from_ntt(a, b);
  return a;
}
PolyMatrixRaw neg(const PolyMatrixRaw& a) {  // :671-679
  PolyMatrixRaw out(a.params, a.rows, a.cols);
  invert(out, a);
  return out;
}
static PolyMatrixNTT mul_alloc(const PolyMatrixNTT& a, const PolyMatrixNTT& b) {  // :681-689
  PolyMatrixNTT out(a.params, a.rows, b.cols);
  multiply(out, a, b);
  return out;
}
static PolyMatrixNTT add_alloc(const PolyMatrixNTT& a, const PolyMatrixNTT& b) {  // :691-699
  PolyMatrixNTT out(a.params, a.rows, a.cols);
  add(out, a, b);
  return out;
}
static PolyMatrixRaw automorph_alloc(const PolyMatrixRaw& a, size_t t) {  // :553-557
  PolyMatrixRaw res(a.params, a.rows, a.cols);
  automorph(res, a, t);
  return res;
}
static PolyMatrixRaw identity_raw(const Params* params, size_t rows, size_t cols) {  // :160-174
  PolyMatrixRaw m(params, rows, cols);
  for (size_t r = 0; r < rows; r++) m.data[r * cols * params->poly_len + r * params->poly_len] = 1;
  return m;
}
static PolyMatrixRaw random_rng_raw(const Params* params, size_t rows, size_t cols, ChaCha20Rng& rng) {  // :105-117
  PolyMatrixRaw out(params, rows, cols);
  for (size_t r = 0; r < rows; r++)
    for (size_t c = 0; c < cols; c++)
      for (size_t i = 0; i < params->poly_len; i++) out.get_poly(r, c)[i] = rng.next_u64() % params->modulus;
  return out;
}

// ============================================================= gadget.rs
size_t get_bits_per(const Params& params, size_t dim) {  // :3-9
  u64 modulus_log2 = params.modulus_log2;
  if ((u64)dim == modulus_log2) return 1;
  return (size_t)std::floor((double)modulus_log2 / (double)dim) + 1;
}
PolyMatrixRaw build_gadget(const Params& params, size_t rows, size_t cols) {  // :11-32
  PolyMatrixRaw g(&params, rows, cols);
  size_t nx = g.rows, m = g.cols;
  ORACLE_CHECK(m % nx == 0);
  size_t num_elems = m / nx;
  size_t bits_per = get_bits_per(params, num_elems);
  for (size_t i = 0; i < nx; i++)
    for (size_t j = 0; j < num_elems; j++) {
      if (bits_per * j >= 64) continue;
      g.get_poly(i, i + j * nx)[0] = 1ULL << (bits_per * j);
    }
  return g;
}
void gadget_invert_rdim(PolyMatrixRaw& out, const PolyMatrixRaw& inp, size_t rdim) {  // :34-60
  ORACLE_CHECK(out.cols == inp.cols);
  const Params& params = *inp.params;
  size_t mx = out.rows;
  size_t num_elems = mx / rdim;
  size_t bits_per = get_bits_per(params, num_elems);
  u64 mask = (1ULL << bits_per) - 1;
  for (size_t i = 0; i < inp.cols; i++)
    for (size_t j = 0; j < rdim; j++)
      for (size_t z = 0; z < params.poly_len; z++) {
        u64 val = inp.get_poly(j, i)[z];
        for (size_t k = 0; k < num_elems; k++) {
          size_t bit_offs = std::min<size_t>(k * bits_per, 64);
          u64 piece = bit_offs >= 64 ? 0 : ((val >> bit_offs) & mask);  // checked_shr -> None => 0
          out.get_poly(j + k * rdim, i)[z] = piece;
        }
      }
}
void gadget_invert(PolyMatrixRaw& out, const PolyMatrixRaw& inp) { gadget_invert_rdim(out, inp, inp.rows); }  // :62-64

// =============================================================== util.rs
size_t calc_index(const size_t* indices, const size_t* lengths, size_t n) {  // :36-44
  size_t idx = 0, prod = 1;
  for (size_t i = n; i-- > 0;) {
    idx += indices[i] * prod;
    prod *= lengths[i];
  }
  return idx;
}
u64 read_arbitrary_bits(const uint8_t* data, size_t bit_offs, size_t num_bits) {  // :289-301
  size_t word_off = bit_offs / 64, bit_off_within_word = bit_offs % 64;
  if (bit_off_within_word + num_bits <= 64) {
    u64 val;
    memcpy(&val, data + word_off * 8, 8);
    return (val >> bit_off_within_word) & ((1ULL << num_bits) - 1);
  }
  u128 val;
  memcpy(&val, data + word_off * 8, 16);
  return (u64)((val >> bit_off_within_word) & (((u128)1 << num_bits) - 1));
}
void write_arbitrary_bits(uint8_t* data, u64 val, size_t bit_offs, size_t num_bits) {  // :303-321
  size_t word_off = bit_offs / 64, bit_off_within_word = bit_offs % 64;
  val = val & ((1ULL << num_bits) - 1);
  if (bit_off_within_word + num_bits <= 64) {
    u64 cur_val;
    memcpy(&cur_val, data + word_off * 8, 8);
    cur_val &= ~(((1ULL << num_bits) - 1) << bit_off_within_word);
    cur_val |= val << bit_off_within_word;
    memcpy(data + word_off * 8, &cur_val, 8);
  } else {
    u128 cur_val;
    memcpy(&cur_val, data + word_off * 8, 16);
    u128 mask = ~((((u128)1 << num_bits) - 1) << bit_off_within_word);
    cur_val &= mask;
    cur_val |= (u128)val << bit_off_within_word;
    memcpy(data + word_off * 8, &cur_val, 16);
  }
}
void reorient_reg_ciphertexts(const Params& params, u64* out, const std::vector<PolyMatrixNTT>& v_reg) {  // :323-355
  size_t poly_len = params.poly_len, crt_count = params.crt_count;
  ORACLE_CHECK(crt_count == 2);
  ORACLE_CHECK(log2_floor(params.moduli[0]) <= 32);
  size_t num_reg_expanded = (size_t)1 << params.db_dim_1;
  size_t ct_rows = v_reg[0].rows, ct_cols = v_reg[0].cols;
  ORACLE_CHECK(ct_rows == 2 && ct_cols == 1);
  for (size_t j = 0; j < num_reg_expanded; j++)
    for (size_t r = 0; r < ct_rows; r++)
      for (size_t m = 0; m < ct_cols; m++)
        for (size_t z = 0; z < poly_len; z++) {
          size_t idx_a_in = r * (ct_cols * crt_count * poly_len) + m * (crt_count * poly_len);
          size_t idx_a_out = z * (num_reg_expanded * ct_cols * ct_rows) + j * (ct_cols * ct_rows) + m * ct_rows + r;
          u64 val1 = v_reg[j].data[idx_a_in + z] % params.moduli[0];
          u64 val2 = v_reg[j].data[idx_a_in + poly_len + z] % params.moduli[1];
          out[idx_a_out] = val1 | (val2 << 32);
        }
}

// =================================================== rand_chacha stand-in
static inline u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define CHACHA_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);
void chacha20_block(const u32 in[16], u32 out[16]) {
  u32 x[16];
  for (int i = 0; i < 16; i++) x[i] = in[i];
  for (int i = 0; i < 10; i++) {
    CHACHA_QR(x[0], x[4], x[8], x[12]) CHACHA_QR(x[1], x[5], x[9], x[13])
    CHACHA_QR(x[2], x[6], x[10], x[14]) CHACHA_QR(x[3], x[7], x[11], x[15])
    CHACHA_QR(x[0], x[5], x[10], x[15]) CHACHA_QR(x[1], x[6], x[11], x[12])
    CHACHA_QR(x[2], x[7], x[8], x[13]) CHACHA_QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
ChaCha20Rng::ChaCha20Rng(const uint8_t seed[32]) {
  for (int i = 0; i < 8; i++)
    key[i] = (u32)seed[4 * i] | ((u32)seed[4 * i + 1] << 8) | ((u32)seed[4 * i + 2] << 16) | ((u32)seed[4 * i + 3] << 24);
}
u32 ChaCha20Rng::next_u32() {
  if (idx >= 16) {
    u32 st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) st[4 + i] = key[i];
    st[12] = (u32)counter;
    st[13] = (u32)(counter >> 32);
    st[14] = 0;
    st[15] = 0;
    chacha20_block(st, buf);
    counter++;
    idx = 0;
  }
  return buf[idx++];
}
u64 ChaCha20Rng::next_u64() {
  u64 lo = next_u32();
  u64 hi = next_u32();
  return lo | (hi << 32);
}

u64 splitmix64(u64& state) {
  u64 z = (state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// ===================================================== discrete_gaussian.rs
void DiscreteGaussian::init(double noise_width) {  // :80-103
  const size_t NUM_WIDTHS = 4;
  max_val = (int64_t)std::ceil(noise_width * (double)NUM_WIDTHS);
  std::vector<double> table;
  double total = 0.0;
  for (int64_t i = -max_val; i < max_val + 1; i++) {
    double p_val = std::exp(-M_PI * std::pow((double)i, 2) / std::pow(noise_width, 2));
    table.push_back(p_val);
    total += p_val;
  }
  cdf_table.clear();
  double cum_prob = 0.0;
  for (double p_val : table) {
    cum_prob += p_val / total;
    double scaled = std::round(cum_prob * 18446744073709551615.0);
    u64 v = scaled >= 18446744073709551615.0 ? ~0ULL : (u64)scaled;  // Rust `as u64` saturates
    cdf_table.push_back(v);
  }
}
u64 DiscreteGaussian::sample(u64 modulus, ChaCha20Rng& rng) const {  // :105-127
  u64 sampled_val = rng.next_u64();
  size_t len = (size_t)(2 * max_val + 1);
  u64 to_output = 0;
  for (size_t i = len; i-- > 0;) {
    int64_t out_val = (int64_t)i - max_val;
    if (out_val < 0) out_val += (int64_t)modulus;
    if (!(sampled_val > cdf_table[i])) to_output = (u64)out_val;
  }
  return to_output;
}
void DiscreteGaussian::sample_matrix(PolyMatrixRaw& p, ChaCha20Rng& rng) const {  // :129-140
  u64 modulus = p.params->modulus;
  for (size_t r = 0; r < p.rows; r++)
    for (size_t c = 0; c < p.cols; c++) {
      u64* poly = p.get_poly(r, c);
      for (size_t z = 0; z < p.params->poly_len; z++) poly[z] = sample(modulus, rng);
    }
}

// ============================================================== client.rs
static u64 get_inv_from_rng(const Params& params, ChaCha20Rng& rng) {  // :47-49
  return params.modulus - (rng.next_u64() % params.modulus);
}
static size_t mat_sz_bytes_excl_first_row(const PolyMatrixRaw& a) {  // :51-53
  return (a.rows - 1) * a.cols * a.params->poly_len * sizeof(u64);
}
static void serialize_polymatrix_for_rng(std::vector<uint8_t>& vec, const PolyMatrixRaw& a) {  // :55-60
  size_t offs = a.cols * a.params->poly_len;
  size_t cnt = (a.rows - 1) * a.cols * a.params->poly_len;
  size_t old = vec.size();
  vec.resize(old + cnt * 8);
  memcpy(vec.data() + old, a.data.data() + offs, cnt * 8);
}
static size_t deserialize_polymatrix_rng(PolyMatrixRaw& a, const uint8_t* data, ChaCha20Rng& rng) {  // :68-80
  size_t first = a.cols * a.params->poly_len;
  for (size_t i = 0; i < first; i++) a.data[i] = get_inv_from_rng(*a.params, rng);
  size_t bytes = mat_sz_bytes_excl_first_row(a);
  memcpy(a.data.data() + first, data, bytes);
  return bytes;
}
static size_t deserialize_vec_polymatrix_rng(std::vector<PolyMatrixRaw>& a, const uint8_t* data, ChaCha20Rng& rng) {  // :82-93
  size_t bytes_read = 0;
  for (size_t i = 0; i < a.size(); i++) bytes_read += deserialize_polymatrix_rng(a[i], data + bytes_read, rng);
  return bytes_read;
}
static std::vector<PolyMatrixRaw> new_vec_raw(const Params& params, size_t num, size_t rows, size_t cols) {  // :34-45
  std::vector<PolyMatrixRaw> v;
  for (size_t i = 0; i < num; i++) v.emplace_back(&params, rows, cols);
  return v;
}
static std::vector<PolyMatrixNTT> to_ntt_alloc_vec(const std::vector<PolyMatrixRaw>& v) {  // :185-187
  std::vector<PolyMatrixNTT> o;
  for (auto& m : v) o.push_back(to_ntt_alloc(m));
  return o;
}

// client.rs:95-128
static std::vector<u64> interleave_rng_data(const Params& params, const std::vector<u64>& v_buf, ChaCha20Rng& rng) {
  std::vector<PolyMatrixNTT> reg_cts;
  for (size_t i = 0; i < params.num_expanded(); i++) {
    PolyMatrixRaw sigma(&params, 2, 1);
    for (size_t z = 0; z < params.poly_len; z++) sigma.data[z] = get_inv_from_rng(params, rng);
    reg_cts.push_back(to_ntt_alloc(sigma));
  }
  size_t reg_cts_buf_words = params.num_expanded() * 2 * params.poly_len;
  std::vector<u64> reg_cts_buf(reg_cts_buf_words, 0);
  reorient_reg_ciphertexts(params, reg_cts_buf.data(), reg_cts);
  ORACLE_CHECK(reg_cts_buf_words == 2 * v_buf.size());
  std::vector<u64> out;
  for (size_t i = 0; i < v_buf.size(); i++) {
    out.push_back(reg_cts_buf[2 * i]);
    out.push_back(v_buf[i]);
  }
  return out;
}

std::vector<uint8_t> PublicParameters::serialize() const {  // :198-210
  std::vector<uint8_t> data(seed, seed + 32);
  for (auto& m : v_packing) serialize_polymatrix_for_rng(data, from_ntt_alloc(m));
  for (auto& m : v_expansion_left) serialize_polymatrix_for_rng(data, from_ntt_alloc(m));
  if (has_expansion_right)
    for (auto& m : v_expansion_right) serialize_polymatrix_for_rng(data, from_ntt_alloc(m));
  for (auto& m : v_conversion) serialize_polymatrix_for_rng(data, from_ntt_alloc(m));
  return data;
}

PublicParameters PublicParameters::deserialize(const Params& params, const uint8_t* data, size_t len) {  // :212-259
  ORACLE_CHECK(params.setup_bytes() == len);
  PublicParameters pp;
  size_t idx = 0;
  memcpy(pp.seed, data, SEED_LENGTH);
  ChaCha20Rng rng(pp.seed);
  idx += SEED_LENGTH;
  // NB client.rs:221 allocates params.n packing matrices regardless of version.
  std::vector<PolyMatrixRaw> v_packing = new_vec_raw(params, params.n, params.n + 1, params.t_conv);
  idx += deserialize_vec_polymatrix_rng(v_packing, data + idx, rng);
  pp.v_packing = to_ntt_alloc_vec(v_packing);
  if (params.expand_queries) {
    std::vector<PolyMatrixRaw> v_expansion_left = new_vec_raw(params, params.g(), 2, params.t_exp_left);
    idx += deserialize_vec_polymatrix_rng(v_expansion_left, data + idx, rng);
    std::vector<PolyMatrixRaw> v_expansion_right = v_expansion_left;
    if (params.version == 0 || params.t_exp_right != params.t_exp_left) {
      std::vector<PolyMatrixRaw> tmp = new_vec_raw(params, params.stop_round() + 1, 2, params.t_exp_right);
      idx += deserialize_vec_polymatrix_rng(tmp, data + idx, rng);
      v_expansion_right = tmp;
    }
    std::vector<PolyMatrixRaw> v_conversion = new_vec_raw(params, 1, 2, 2 * params.t_conv);
    deserialize_vec_polymatrix_rng(v_conversion, data + idx, rng);
    pp.v_expansion_left = to_ntt_alloc_vec(v_expansion_left);
    pp.v_expansion_right = to_ntt_alloc_vec(v_expansion_right);
    pp.has_expansion_right = true;
    pp.v_conversion = to_ntt_alloc_vec(v_conversion);
  }
  return pp;
}

std::vector<uint8_t> Query::serialize() const {  // :279-301
  std::vector<uint8_t> data(seed, seed + 32);
  if (has_ct) serialize_polymatrix_for_rng(data, ct);
  if (!v_buf.empty()) {
    for (size_t i = 0; i < v_buf.size(); i++)
      if (i % 2 == 1) {
        size_t old = data.size();
        data.resize(old + 8);
        memcpy(data.data() + old, &v_buf[i], 8);
      }
  }
  for (auto& x : v_ct) serialize_polymatrix_for_rng(data, x);
  return data;
}

Query Query::deserialize(const Params& params, const uint8_t* data, size_t len) {  // :303-329
  ORACLE_CHECK(params.query_bytes() == len);
  Query out;
  memcpy(out.seed, data, SEED_LENGTH);
  ChaCha20Rng rng(out.seed);
  data += SEED_LENGTH;
  if (params.expand_queries) {
    out.ct = PolyMatrixRaw(&params, 2, 1);
    deserialize_polymatrix_rng(out.ct, data, rng);
    out.has_ct = true;
  } else {
    size_t v_buf_bytes = params.query_v_buf_bytes();
    std::vector<u64> v_buf(v_buf_bytes / 8);
    memcpy(v_buf.data(), data, v_buf_bytes);
    out.v_buf = interleave_rng_data(params, v_buf, rng);
    out.v_ct = new_vec_raw(params, params.db_dim_2, 2, 2 * params.t_gsw);
    deserialize_vec_polymatrix_rng(out.v_ct, data + v_buf_bytes, rng);
  }
  return out;
}

static PolyMatrixRaw matrix_with_identity(const PolyMatrixRaw& p) {  // :332-338
  ORACLE_CHECK(p.cols == 1);
  PolyMatrixRaw r(p.params, p.rows, p.rows + 1);
  r.copy_into(p, 0, 0);
  r.copy_into(identity_raw(p.params, p.rows, p.rows), 0, 1);
  return r;
}

// client.rs:130-144; `pol.shuffle(rng)` is rand's Fisher-Yates over secret randomness (unpinned, not required)
static void gen_ternary_mat(PolyMatrixRaw& mat, size_t hamming, ChaCha20Rng& rng) {
  u64 modulus = mat.params->modulus;
  size_t N = mat.params->poly_len;
  for (size_t r = 0; r < mat.rows; r++)
    for (size_t c = 0; c < mat.cols; c++) {
      u64* pol = mat.get_poly(r, c);
      for (size_t i = 0; i < N; i++) pol[i] = 0;
      for (size_t i = 0; i < hamming; i++) pol[i] = 1;
      for (size_t i = hamming; i < 2 * hamming; i++) pol[i] = modulus - 1;
      for (size_t i = N; i-- > 1;) {
        size_t j = (size_t)(rng.next_u64() % (i + 1));
        std::swap(pol[i], pol[j]);
      }
    }
}

Client::Client(const Params* p) : params(p) {  // :371-389
  sk_gsw = PolyMatrixRaw(p, p->n, 1);
  sk_reg = PolyMatrixRaw(p, 1, 1);
  sk_gsw_full = matrix_with_identity(sk_gsw);
  sk_reg_full = matrix_with_identity(sk_reg);
  dg.init(p->noise_width);
}

PolyMatrixRaw Client::get_fresh_gsw_public_key(size_t m, ChaCha20Rng& rng, ChaCha20Rng& rng_pub) const {  // :401-417
  size_t n = params->n;
  PolyMatrixRaw a = random_rng_raw(params, 1, m, rng_pub);
  PolyMatrixRaw e(params, n, m);
  dg.sample_matrix(e, rng);
  PolyMatrixRaw a_inv = neg(a);
  PolyMatrixNTT b_p = mul_alloc(to_ntt_alloc(sk_gsw), to_ntt_alloc(a));
  PolyMatrixNTT b = add_alloc(to_ntt_alloc(e), b_p);
  return stack(a_inv, from_ntt_alloc(b));
}
PolyMatrixNTT Client::get_regev_sample(ChaCha20Rng& rng, ChaCha20Rng& rng_pub) const {  // :419-433
  PolyMatrixRaw a = random_rng_raw(params, 1, 1, rng_pub);
  PolyMatrixRaw e(params, 1, 1);
  dg.sample_matrix(e, rng);
  PolyMatrixNTT b_p = mul_alloc(to_ntt_alloc(sk_reg), to_ntt_alloc(a));
  PolyMatrixNTT b = add_alloc(to_ntt_alloc(e), b_p);
  PolyMatrixNTT p(params, 2, 1);
  p.copy_into(to_ntt_alloc(neg(a)), 0, 0);
  p.copy_into(b, 1, 0);
  return p;
}
PolyMatrixNTT Client::get_fresh_reg_public_key(size_t m, ChaCha20Rng& rng, ChaCha20Rng& rng_pub) const {  // :435-449
  PolyMatrixNTT p(params, 2, m);
  for (size_t i = 0; i < m; i++) p.copy_into(get_regev_sample(rng, rng_pub), 0, i);
  return p;
}
PolyMatrixNTT Client::encrypt_matrix_gsw(const PolyMatrixNTT& ag, ChaCha20Rng& rng, ChaCha20Rng& rng_pub) const {  // :451-461
  PolyMatrixRaw p = get_fresh_gsw_public_key(ag.cols, rng, rng_pub);
  return add_alloc(to_ntt_alloc(p), ag.pad_top(1));
}
PolyMatrixNTT Client::encrypt_matrix_reg(const PolyMatrixNTT& a, ChaCha20Rng& rng, ChaCha20Rng& rng_pub) const {  // :463-472
  PolyMatrixNTT p = get_fresh_reg_public_key(a.cols, rng, rng_pub);
  return add_alloc(p, a.pad_top(1));
}
PolyMatrixNTT Client::decrypt_matrix_reg(const PolyMatrixNTT& a) const {  // :474-476
  return mul_alloc(to_ntt_alloc(sk_reg_full), a);
}
std::vector<PolyMatrixNTT> Client::generate_expansion_params(size_t num_exp, size_t m_exp, ChaCha20Rng& rng,
                                                             ChaCha20Rng& rng_pub) const {  // :482-502
  PolyMatrixRaw g_exp = build_gadget(*params, 1, m_exp);
  PolyMatrixNTT g_exp_ntt = to_ntt_alloc(g_exp);
  std::vector<PolyMatrixNTT> res;
  for (size_t i = 0; i < num_exp; i++) {
    size_t t = (params->poly_len / ((size_t)1 << i)) + 1;
    PolyMatrixRaw tau_sk_reg = automorph_alloc(sk_reg, t);
    PolyMatrixNTT prod = mul_alloc(to_ntt_alloc(tau_sk_reg), g_exp_ntt);
    res.push_back(encrypt_matrix_reg(prod, rng, rng_pub));
  }
  return res;
}

static void derive_seed(ChaCha20Rng& rng, uint8_t out[32]) {
  for (int i = 0; i < 4; i++) {
    u64 v = rng.next_u64();
    memcpy(out + 8 * i, &v, 8);
  }
}

// client.rs:533-616 (version 0 and version > 0 keygen)
PublicParameters Client::generate_keys(const uint8_t secret_seed[32]) {
  ChaCha20Rng rng(secret_seed);
  gen_ternary_mat(sk_gsw, HAMMING_WEIGHT, rng);
  gen_ternary_mat(sk_reg, HAMMING_WEIGHT, rng);
  sk_gsw_full = matrix_with_identity(sk_gsw);
  sk_reg_full = matrix_with_identity(sk_reg);
  PolyMatrixNTT sk_reg_ntt = to_ntt_alloc(sk_reg);
  PolyMatrixNTT sk_gsw_ntt = to_ntt_alloc(sk_gsw);

  PublicParameters pp;
  derive_seed(rng, pp.seed);
  ChaCha20Rng rng_pub(pp.seed);

  PolyMatrixRaw gadget_conv = build_gadget(*params, 1, params->t_conv);
  PolyMatrixNTT gadget_conv_ntt = to_ntt_alloc(gadget_conv);
  size_t num_packing_mats = params->version == 0 ? params->n : 1;
  for (size_t i = 0; i < num_packing_mats; i++) {
    PolyMatrixNTT scaled(params, gadget_conv_ntt.rows, gadget_conv_ntt.cols);
    scalar_multiply(scaled, sk_reg_ntt, gadget_conv_ntt);
    PolyMatrixNTT ag(params, params->n, params->t_conv);
    ag.copy_into(scaled, i, 0);
    pp.v_packing.push_back(encrypt_matrix_gsw(ag, rng, rng_pub));
  }
  if (params->version > 0) {
    PolyMatrixNTT scaled = mul_alloc(sk_gsw_ntt, gadget_conv_ntt);
    // shift_rows_by_one, poly.rs:340-349
    PolyMatrixNTT rotated = scaled;
    if (scaled.rows > 1) {
      PolyMatrixNTT all_but_last = scaled.submatrix(0, 0, scaled.rows - 1, scaled.cols);
      PolyMatrixNTT last = scaled.submatrix(scaled.rows - 1, 0, 1, scaled.cols);
      rotated = PolyMatrixNTT(params, scaled.rows, scaled.cols);
      rotated.copy_into(last, 0, 0);
      rotated.copy_into(all_but_last, 1, 0);
    }
    pp.v_packing.push_back(encrypt_matrix_gsw(rotated, rng, rng_pub));
  }
  if (params->expand_queries) {
    pp.v_expansion_left = generate_expansion_params(params->g(), params->t_exp_left, rng, rng_pub);
    if (params->version == 0 || params->t_exp_right != params->t_exp_left) {
      pp.v_expansion_right = generate_expansion_params(params->stop_round() + 1, params->t_exp_right, rng, rng_pub);
      pp.has_expansion_right = true;
    }
    PolyMatrixRaw g_conv = build_gadget(*params, 2, 2 * params->t_conv);
    PolyMatrixNTT sk_reg_squared_ntt = mul_alloc(sk_reg_ntt, sk_reg_ntt);
    pp.v_conversion.emplace_back(params, 2, 2 * params->t_conv);
    for (size_t i = 0; i < 2 * params->t_conv; i++) {
      PolyMatrixNTT sigma;
      if (i % 2 == 0) {
        u64 val = g_conv.get_poly(0, i)[0];
        sigma = mul_alloc(sk_reg_squared_ntt, to_ntt_alloc(single_poly(*params, val)));
      } else {
        u64 val = g_conv.get_poly(1, i)[0];
        sigma = mul_alloc(sk_reg_ntt, to_ntt_alloc(single_poly(*params, val)));
      }
      PolyMatrixNTT ct = encrypt_matrix_reg(sigma, rng, rng_pub);
      pp.v_conversion[0].copy_into(ct, 0, i);
    }
  }
  return pp;
}

// client.rs:618-721
Query Client::generate_query(size_t idx_target, const uint8_t secret_seed[32]) {
  const Params& p = *params;
  size_t further_dims = p.db_dim_2;
  size_t idx_dim0 = idx_target / ((size_t)1 << further_dims);
  size_t idx_further = idx_target % ((size_t)1 << further_dims);
  u64 scale_k = p.modulus / p.pt_modulus;
  size_t bits_per = get_bits_per(p, p.t_gsw);

  ChaCha20Rng rng(secret_seed);
  Query query;
  derive_seed(rng, query.seed);
  ChaCha20Rng rng_pub(query.seed);
  if (p.expand_queries) {
    PolyMatrixRaw sigma(params, 1, 1);
    u64 inv_2_g_first = invert_uint_mod(1ULL << p.g(), p.modulus);
    u64 inv_2_g_rest = invert_uint_mod(1ULL << (p.stop_round() + 1), p.modulus);
    if (p.db_dim_2 == 0) {
      for (size_t i = 0; i < ((size_t)1 << p.db_dim_1); i++)
        if (i == idx_dim0) sigma.data[i] = scale_k;
      for (size_t i = 0; i < p.poly_len; i++) sigma.data[i] = multiply_uint_mod(sigma.data[i], inv_2_g_first, p.modulus);
    } else {
      for (size_t i = 0; i < ((size_t)1 << p.db_dim_1); i++)
        if (i == idx_dim0) sigma.data[2 * i] = scale_k;
      for (size_t i = 0; i < further_dims; i++) {
        u64 mask = 1ULL << i;
        bool bit = ((u64)idx_further & mask) == mask;
        for (size_t j = 0; j < p.t_gsw; j++) {
          u64 val = bit ? (1ULL << (bits_per * j)) : 0;
          size_t idx = i * p.t_gsw + j;
          sigma.data[2 * idx + 1] = val;
        }
      }
      for (size_t i = 0; i < p.poly_len / 2; i++) {
        sigma.data[2 * i] = multiply_uint_mod(sigma.data[2 * i], inv_2_g_first, p.modulus);
        sigma.data[2 * i + 1] = multiply_uint_mod(sigma.data[2 * i + 1], inv_2_g_rest, p.modulus);
      }
    }
    query.ct = from_ntt_alloc(encrypt_matrix_reg(to_ntt_alloc(sigma), rng, rng_pub));
    query.has_ct = true;
  } else {
    size_t num_expanded = (size_t)1 << p.db_dim_1;
    std::vector<PolyMatrixNTT> reg_cts;
    for (size_t i = 0; i < num_expanded; i++) {
      u64 value = (u64)(i == idx_dim0) * scale_k;
      PolyMatrixRaw sigma = single_poly(p, value);
      reg_cts.push_back(encrypt_matrix_reg(to_ntt_alloc(sigma), rng, rng_pub));
    }
    std::vector<u64> reg_cts_buf(num_expanded * 2 * p.poly_len, 0);
    reorient_reg_ciphertexts(p, reg_cts_buf.data(), reg_cts);
    for (size_t i = 0; i < further_dims; i++) {
      u64 bit = ((u64)idx_further & (1ULL << i)) >> i;
      PolyMatrixNTT ct_gsw(params, 2, 2 * p.t_gsw);
      for (size_t j = 0; j < p.t_gsw; j++) {
        u64 value = (1ULL << (bits_per * j)) * bit;
        PolyMatrixNTT sigma_ntt = to_ntt_alloc(single_poly(p, value));
        PolyMatrixNTT prod = mul_alloc(to_ntt_alloc(sk_reg), sigma_ntt);
        ct_gsw.copy_into(encrypt_matrix_reg(prod, rng, rng_pub), 0, 2 * j);
        ct_gsw.copy_into(encrypt_matrix_reg(sigma_ntt, rng, rng_pub), 0, 2 * j + 1);
      }
      query.v_ct.push_back(from_ntt_alloc(ct_gsw));
    }
    query.v_buf = reg_cts_buf;
  }
  return query;
}

// client.rs:732-810
std::vector<uint8_t> Client::decode_response(const uint8_t* data, size_t len) const {
  const Params& p = *params;
  u64 pm = p.pt_modulus;
  u64 p_bits = log2_ceil(pm);
  u64 q1 = 4 * pm;
  size_t q1_bits = (size_t)log2_ceil(q1);
  u64 q2 = Q2_VALUES[p.q2_bits];
  size_t q2_bits = (size_t)p.q2_bits;
  Params q2_params = Params::init(p.poly_len, {q2}, p.noise_width, p.n, p.pt_modulus, p.q2_bits, p.t_conv,
                                  p.t_exp_left, p.t_exp_right, p.t_gsw, p.expand_queries, p.db_dim_1, p.db_dim_2,
                                  p.instances, p.db_item_size, p.version);
  PolyMatrixRaw sk_gsw_q2(&q2_params, p.n, 1);
  for (size_t i = 0; i < p.poly_len * p.n; i++) sk_gsw_q2.data[i] = recenter(sk_gsw.data[i], p.modulus, q2);
  PolyMatrixNTT sk_gsw_q2_ntt(&q2_params, p.n, 1);
  to_ntt(sk_gsw_q2_ntt, sk_gsw_q2);

  PolyMatrixRaw result(params, p.instances * p.n, p.n);
  size_t bit_offs = 0;
  // the response is exactly as long as the bits read; pad so the u128 window never runs off the end
  std::vector<uint8_t> padded(data, data + len);
  padded.resize(len + 16, 0);
  for (size_t instance = 0; instance < p.instances; instance++) {
    PolyMatrixRaw first_row(&q2_params, 1, p.n);
    PolyMatrixRaw rest_rows(params, p.n, p.n);
    for (size_t i = 0; i < p.n * p.poly_len; i++) {
      first_row.data[i] = read_arbitrary_bits(padded.data(), bit_offs, q2_bits);
      bit_offs += q2_bits;
    }
    for (size_t i = 0; i < p.n * p.n * p.poly_len; i++) {
      rest_rows.data[i] = read_arbitrary_bits(padded.data(), bit_offs, q1_bits);
      bit_offs += q1_bits;
    }
    PolyMatrixNTT first_row_q2(&q2_params, 1, p.n);
    to_ntt(first_row_q2, first_row);
    PolyMatrixRaw sk_prod = from_ntt_alloc(mul_alloc(sk_gsw_q2_ntt, first_row_q2));
    int64_t q1_i64 = (int64_t)q1, q2_i64 = (int64_t)q2;
    i128 p_i128 = (i128)pm;
    for (size_t i = 0; i < p.n * p.n * p.poly_len; i++) {
      int64_t val_first = (int64_t)sk_prod.data[i];
      if (val_first >= q2_i64 / 2) val_first -= q2_i64;
      int64_t val_rest = (int64_t)rest_rows.data[i];
      if (val_rest >= q1_i64 / 2) val_rest -= q1_i64;
      int64_t denom = (int64_t)(q2 * (q1 / pm));
      int64_t r = val_first * q1_i64;
      r += val_rest * q2_i64;
      int64_t sign = r >= 0 ? 1 : -1;
      i128 res = ((i128)(r + sign * (denom / 2))) / (i128)denom;
      res = (res + ((i128)denom / p_i128) * p_i128 + 2 * p_i128) % p_i128;
      size_t idx = instance * p.n * p.n * p.poly_len + i;
      result.data[idx] = (u64)res;
    }
  }
  return result.to_vec((size_t)p_bits, p.modp_words_per_chunk());
}

// ============================================================== server.rs
std::vector<PolyMatrixNTT> get_v_neg1(const Params& params) {  // params.rs:98-107
  std::vector<PolyMatrixNTT> v_neg1;
  for (size_t i = 0; i < params.poly_len_log2; i++) {
    size_t idx = params.poly_len - ((size_t)1 << i);
    PolyMatrixRaw ng1(&params, 1, 1);
    ng1.data[idx] = 1;
    v_neg1.push_back(to_ntt_alloc(neg(ng1)));
  }
  return v_neg1;
}

// server.rs:19-121
void coefficient_expansion(std::vector<PolyMatrixNTT>& v, size_t g, size_t stop_round, const Params& params,
                           const std::vector<PolyMatrixNTT>& v_w_left, const std::vector<PolyMatrixNTT>& v_w_right,
                           const std::vector<PolyMatrixNTT>& v_neg1, size_t max_bits_to_gen_right) {
  size_t poly_len = params.poly_len;
  for (size_t r = 0; r < g; r++) {
    size_t num_in = (size_t)1 << r;
    size_t num_out = 2 * num_in;
    size_t t = (poly_len / ((size_t)1 << r)) + 1;
    const PolyMatrixNTT& neg1 = v_neg1[r];

    auto action_expand = [&](size_t i, PolyMatrixNTT& v_i) {
      if ((stop_round > 0 && r > stop_round && (i % 2) == 1) ||
          (stop_round > 0 && r == stop_round && (i % 2) == 1 && (i / 2) >= max_bits_to_gen_right))
        return;
      PolyMatrixRaw ct(&params, 2, 1), ct_auto(&params, 2, 1), ct_auto_1(&params, 1, 1);
      PolyMatrixNTT ct_auto_1_ntt(&params, 1, 1), w_times_ginv_ct(&params, 2, 1);
      bool left = (r != 0) && (i % 2 == 0);
      const PolyMatrixNTT& w = left ? v_w_left[r] : v_w_right[r];
      size_t gadget_dim = left ? params.t_exp_left : params.t_exp_right;
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
      for (size_t j = 0; j < 2; j++)
        for (size_t n = 0; n < params.crt_count; n++)
          for (size_t z = 0; z < poly_len; z++) {
            u64 sum = v_i.data[idx] + w_times_ginv_ct.data[idx] + j * ct_auto_1_ntt.data[n * poly_len + z];
            v_i.data[idx] = barrett_coeff_u64(params, sum, n);
            idx++;
          }
    };

    for (size_t i = 0; i < num_in; i++) scalar_multiply(v[num_in + i], neg1, v[i]);
    // NB: both halves are enumerated from 0 (server.rs:112-119)
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < num_in; i++) action_expand(i, v[i]);
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < num_out - num_in; i++) action_expand(i, v[num_in + i]);
  }
}

// server.rs:123-151
void regev_to_gsw(std::vector<PolyMatrixNTT>& v_gsw, const std::vector<PolyMatrixNTT>& v_inp, const PolyMatrixNTT& v,
                  const Params& params, size_t idx_factor, size_t idx_offset) {
  ORACLE_CHECK(v.rows == 2);
  ORACLE_CHECK(v.cols == 2 * params.t_conv);
#pragma omp parallel for
  for (size_t i = 0; i < v_gsw.size(); i++) {
    PolyMatrixNTT& ct = v_gsw[i];
    PolyMatrixRaw ginv_c_inp(&params, 2 * params.t_conv, 1);
    PolyMatrixNTT ginv_c_inp_ntt(&params, 2 * params.t_conv, 1);
    PolyMatrixRaw tmp_ct_raw(&params, 2, 1);
    PolyMatrixNTT tmp_ct(&params, 2, 1);
    for (size_t j = 0; j < params.t_gsw; j++) {
      size_t idx_ct = i * params.t_gsw + j;
      size_t idx_inp = idx_factor * idx_ct + idx_offset;
      ct.copy_into(v_inp[idx_inp], 0, 2 * j + 1);
      from_ntt(tmp_ct_raw, v_inp[idx_inp]);
      gadget_invert(ginv_c_inp, tmp_ct_raw);
      to_ntt(ginv_c_inp_ntt, ginv_c_inp);
      multiply(tmp_ct, v, ginv_c_inp_ntt);
      ct.copy_into(tmp_ct, 0, 2 * j);
    }
  }
}

// server.rs:155-221 (ct_rows=2, ct_cols=1, pt_rows=1, pt_cols=1)
void multiply_reg_by_database(std::vector<PolyMatrixNTT>& out, const u64* db, const u64* v_firstdim,
                              const Params& params, size_t dim0, size_t num_per) {
  const size_t ct_rows = 2, ct_cols = 1, pt_rows = 1, pt_cols = 1;
  for (size_t z = 0; z < params.poly_len; z++) {
    size_t idx_a_base = z * (ct_cols * dim0 * ct_rows);
    size_t idx_b_base = z * (num_per * pt_cols * dim0 * pt_rows);
    for (size_t i = 0; i < num_per; i++)
      for (size_t c = 0; c < pt_cols; c++) {
        u128 sums_out_n0_0 = 0, sums_out_n0_1 = 0, sums_out_n1_0 = 0, sums_out_n1_1 = 0;
        for (size_t jm = 0; jm < dim0 * pt_rows; jm++) {
          u64 b = db[idx_b_base];
          idx_b_base += 1;
          u64 v_a0 = v_firstdim[idx_a_base + jm * ct_rows];
          u64 v_a1 = v_firstdim[idx_a_base + jm * ct_rows + 1];
          u32 b_lo = (u32)b, b_hi = (u32)(b >> 32);
          u32 v_a0_lo = (u32)v_a0, v_a0_hi = (u32)(v_a0 >> 32);
          u32 v_a1_lo = (u32)v_a1, v_a1_hi = (u32)(v_a1 >> 32);
          sums_out_n0_0 += (u128)((u64)v_a0_lo * (u64)b_lo);
          sums_out_n0_1 += (u128)((u64)v_a1_lo * (u64)b_lo);
          sums_out_n1_0 += (u128)((u64)v_a0_hi * (u64)b_hi);
          sums_out_n1_1 += (u128)((u64)v_a1_hi * (u64)b_hi);
        }
        size_t crt_count = params.crt_count, poly_len = params.poly_len;
        size_t n = 0;
        size_t idx_c = c * (crt_count * poly_len) + n * poly_len + z;
        out[i].data[idx_c] = (u64)(sums_out_n0_0 % (u128)params.moduli[0]);
        idx_c += pt_cols * crt_count * poly_len;
        out[i].data[idx_c] = (u64)(sums_out_n0_1 % (u128)params.moduli[0]);
        n = 1;
        idx_c = c * (crt_count * poly_len) + n * poly_len + z;
        out[i].data[idx_c] = (u64)(sums_out_n1_0 % (u128)params.moduli[1]);
        idx_c += pt_cols * crt_count * poly_len;
        out[i].data[idx_c] = (u64)(sums_out_n1_1 % (u128)params.moduli[1]);
      }
  }
}

// server.rs:223-275
void generate_random_db_and_get_item(const Params& params, size_t item_idx, u64 seed, PolyMatrixRaw& item,
                                     std::vector<u64>& db) {
  size_t instances = params.instances, trials = params.n * params.n;
  size_t dim0 = (size_t)1 << params.db_dim_1, num_per = (size_t)1 << params.db_dim_2;
  size_t num_items = dim0 * num_per;
  size_t db_size_words = instances * trials * num_items * params.poly_len;
  db.assign(db_size_words, 0);
  item = PolyMatrixRaw(&params, params.instances * params.n, params.n);
  for (size_t instance = 0; instance < instances; instance++)
    for (size_t trial = 0; trial < trials; trial++) {
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < num_items; i++) {
        size_t ii = i % num_per, j = i / num_per;
        // per-item stream so that generation is order-independent
        u64 st = seed ^ (0xD1B54A32D192ED03ULL * (u64)(((instance * trials + trial) * num_items + i) + 1));
        PolyMatrixRaw db_item(&params, 1, 1);
        for (size_t z = 0; z < params.poly_len; z++) db_item.data[z] = (splitmix64(st) % params.modulus) % params.pt_modulus;
        if (i == item_idx) item.copy_into(db_item, instance * params.n + trial / params.n, trial % params.n);
        for (size_t z = 0; z < params.poly_len; z++)
          db_item.data[z] = recenter_mod(db_item.data[z], params.pt_modulus, params.modulus);
        PolyMatrixNTT db_item_ntt = to_ntt_alloc(db_item);
        for (size_t z = 0; z < params.poly_len; z++) {
          size_t ind[5] = {instance, trial, z, ii, j};
          size_t lens[5] = {instances, trials, params.poly_len, num_per, dim0};
          size_t idx_dst = calc_index(ind, lens, 5);
          db[idx_dst] = db_item_ntt.data[z] | (db_item_ntt.data[params.poly_len + z] << PACKED_OFFSET_2);
        }
      }
    }
}

// server.rs:277-318 over an in-memory "file"
PolyMatrixRaw load_item_from_bytes(const Params& params, const uint8_t* file, size_t file_len, size_t instance,
                                   size_t trial, size_t item_idx) {
  size_t db_item_size = params.db_item_size;
  size_t trials = params.n * params.n;
  size_t chunks = params.instances * trials;
  size_t bytes_per_chunk = (size_t)std::ceil((double)db_item_size / (double)chunks);
  size_t logp = (size_t)std::ceil(std::log2((double)params.pt_modulus));
  size_t modp_words_per_chunk = (size_t)std::ceil((double)(bytes_per_chunk * 8) / (double)logp);
  ORACLE_CHECK(modp_words_per_chunk <= params.poly_len);
  size_t idx_item_in_file = item_idx * db_item_size;
  size_t idx_chunk = instance * trials + trial;
  size_t idx_poly_in_file = idx_item_in_file + idx_chunk * bytes_per_chunk;
  PolyMatrixRaw out(&params, 1, 1);
  std::vector<uint8_t> data(2 * bytes_per_chunk + 16, 0);
  size_t bytes_read = 0;
  if (idx_poly_in_file < file_len) {
    bytes_read = std::min(bytes_per_chunk, file_len - idx_poly_in_file);
    memcpy(data.data(), file + idx_poly_in_file, bytes_read);
  }
  size_t modp_words_read = (size_t)std::ceil((double)(bytes_read * 8) / (double)logp);
  ORACLE_CHECK(modp_words_read <= params.poly_len);
  for (size_t i = 0; i < modp_words_read; i++) {
    out.data[i] = read_arbitrary_bits(data.data(), i * logp, logp);
    ORACLE_CHECK(out.data[i] <= params.pt_modulus);
  }
  return out;
}

// server.rs:320-357
void load_db_from_bytes(const Params& params, const uint8_t* file, size_t file_len, std::vector<u64>& db) {
  size_t instances = params.instances, trials = params.n * params.n;
  size_t dim0 = (size_t)1 << params.db_dim_1, num_per = (size_t)1 << params.db_dim_2;
  size_t num_items = dim0 * num_per;
  db.assign(instances * trials * num_items * params.poly_len, 0);
  for (size_t instance = 0; instance < instances; instance++)
    for (size_t trial = 0; trial < trials; trial++) {
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < num_items; i++) {
        size_t ii = i % num_per, j = i / num_per;
        PolyMatrixRaw db_item = load_item_from_bytes(params, file, file_len, instance, trial, i);
        for (size_t z = 0; z < params.poly_len; z++)
          db_item.data[z] = recenter_mod(db_item.data[z], params.pt_modulus, params.modulus);
        PolyMatrixNTT db_item_ntt = to_ntt_alloc(db_item);
        for (size_t z = 0; z < params.poly_len; z++) {
          size_t ind[5] = {instance, trial, z, ii, j};
          size_t lens[5] = {instances, trials, params.poly_len, num_per, dim0};
          db[calc_index(ind, lens, 5)] = db_item_ntt.data[z] | (db_item_ntt.data[params.poly_len + z] << PACKED_OFFSET_2);
        }
      }
    }
}

// server.rs:388-427
void fold_ciphertexts(const Params& params, std::vector<PolyMatrixRaw>& v_cts, const std::vector<PolyMatrixNTT>& v_folding,
                      const std::vector<PolyMatrixNTT>& v_folding_neg) {
  if (v_cts.size() == 1) return;
  size_t further_dims = (size_t)log2_floor(v_cts.size());
  size_t ell = v_folding[0].cols / 2;
  PolyMatrixRaw ginv_c(&params, 2 * ell, 1);
  PolyMatrixNTT ginv_c_ntt(&params, 2 * ell, 1);
  PolyMatrixNTT prod(&params, 2, 1), sum(&params, 2, 1);
  size_t num_per = v_cts.size();
  for (size_t cur_dim = 0; cur_dim < further_dims; cur_dim++) {
    num_per = num_per / 2;
    for (size_t i = 0; i < num_per; i++) {
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

// server.rs:429-468
PolyMatrixNTT pack(const Params& params, const std::vector<PolyMatrixRaw>& v_ct, const std::vector<PolyMatrixNTT>& v_w) {
  ORACLE_CHECK(v_ct.size() >= params.n * params.n);
  ORACLE_CHECK(v_w.size() == params.n);
  ORACLE_CHECK(v_ct[0].rows == 2 && v_ct[0].cols == 1);
  ORACLE_CHECK(v_w[0].rows == params.n + 1 && v_w[0].cols == params.t_conv);
  PolyMatrixNTT result(&params, params.n + 1, params.n);
  PolyMatrixRaw ginv(&params, params.t_conv, 1);
  PolyMatrixNTT ginv_nttd(&params, params.t_conv, 1);
  PolyMatrixNTT prod(&params, params.n + 1, 1);
  PolyMatrixRaw ct_1(&params, 1, 1), ct_2(&params, 1, 1);
  PolyMatrixNTT ct_2_ntt(&params, 1, 1);
  for (size_t c = 0; c < params.n; c++) {
    PolyMatrixNTT v_int(&params, params.n + 1, 1);
    for (size_t r = 0; r < params.n; r++) {
      const PolyMatrixNTT& w = v_w[r];
      const PolyMatrixRaw& ct = v_ct[r * params.n + c];
      memcpy(ct_1.get_poly(0, 0), ct.get_poly(0, 0), params.poly_len * 8);
      memcpy(ct_2.get_poly(0, 0), ct.get_poly(1, 0), params.poly_len * 8);
      to_ntt(ct_2_ntt, ct_2);
      gadget_invert(ginv, ct_1);
      to_ntt(ginv_nttd, ginv);
      multiply(prod, w, ginv_nttd);
      add_into_at(v_int, ct_2_ntt, 1 + r, 0);
      add_into(v_int, prod);
    }
    result.copy_into(v_int, 0, c);
  }
  return result;
}

// /root/reference/lib/server/src/compute/pack.rs:46-99
PolyMatrixNTT pack_v1(const Params& params, const std::vector<PolyMatrixRaw>& v_ct, const std::vector<PolyMatrixNTT>& v_w) {
  ORACLE_CHECK(v_ct.size() >= params.n * params.n);
  ORACLE_CHECK(v_w.size() == 2);
  ORACLE_CHECK(v_ct[0].rows == 2 && v_ct[0].cols == 1);
  ORACLE_CHECK(v_w[0].rows == params.n + 1 && v_w[0].cols == params.t_conv);
  const PolyMatrixNTT& w_key = v_w[0];
  const PolyMatrixNTT& w_shift = v_w[1];
  PolyMatrixNTT result(&params, params.n + 1, params.n);
  PolyMatrixRaw ginv(&params, params.t_conv, 1);
  PolyMatrixNTT ginv_nttd(&params, params.t_conv, 1);
  PolyMatrixRaw ct_1(&params, 1, 1), ct_2(&params, 1, 1);
  PolyMatrixNTT ct_2_ntt(&params, 1, 1);
  for (size_t c = 0; c < params.n; c++) {
    PolyMatrixNTT v_int(&params, params.n + 1, 1);
    for (size_t r = 0; r < params.n; r++) {
      const PolyMatrixRaw& ct = v_ct[r * params.n + c];
      memcpy(ct_1.get_poly(0, 0), ct.get_poly(0, 0), params.poly_len * 8);
      memcpy(ct_2.get_poly(0, 0), ct.get_poly(1, 0), params.poly_len * 8);
      to_ntt(ct_2_ntt, ct_2);
      gadget_invert(ginv, ct_1);
      to_ntt(ginv_nttd, ginv);
      PolyMatrixNTT prod(&params, params.n + 1, 1);
      multiply(prod, w_key, ginv_nttd);
      add_into_at(prod, ct_2_ntt, 1, 0);
      for (size_t sft = 0; sft < r; sft++) {  // shift until correct position
        PolyMatrixNTT prod_ct_1 = prod.submatrix(0, 0, 1, 1);
        PolyMatrixNTT prod_ct_rest = prod.submatrix(1, 0, prod.rows - 1, 1);
        PolyMatrixRaw g2(&params, params.t_conv, 1);
        gadget_invert(g2, from_ntt_alloc(prod_ct_1));
        PolyMatrixNTT shifted_part_1 = mul_alloc(w_shift, to_ntt_alloc(g2));
        // shift_rows_by_one (poly.rs:340-349): last row first
        PolyMatrixNTT rot = prod_ct_rest;
        if (prod_ct_rest.rows > 1) {
          rot = PolyMatrixNTT(&params, prod_ct_rest.rows, 1);
          rot.copy_into(prod_ct_rest.submatrix(prod_ct_rest.rows - 1, 0, 1, 1), 0, 0);
          rot.copy_into(prod_ct_rest.submatrix(0, 0, prod_ct_rest.rows - 1, 1), 1, 0);
        }
        prod = add_alloc(shifted_part_1, rot.pad_top(1));
      }
      add_into(v_int, prod);
    }
    result.copy_into(v_int, 0, c);
  }
  return result;
}

// pack.rs:101-113
PolyMatrixNTT pack_dispatch(const Params& params, const std::vector<PolyMatrixRaw>& v_ct, const std::vector<PolyMatrixNTT>& v_w) {
  if (params.version == 0) return pack(params, v_ct, v_w);
  ORACLE_CHECK(params.version == 1);
  std::vector<PolyMatrixNTT> two(v_w.begin(), v_w.begin() + 2);
  return pack_v1(params, v_ct, two);
}

// server.rs:470-503
std::vector<uint8_t> encode(const Params& params, const std::vector<PolyMatrixRaw>& v_packed_ct) {
  u64 q1 = 4 * params.pt_modulus;
  size_t q1_bits = (size_t)log2_ceil(q1);
  u64 q2 = Q2_VALUES[params.q2_bits];
  size_t q2_bits = (size_t)params.q2_bits;
  size_t num_bits = params.instances * ((q2_bits * params.n * params.poly_len) + (q1_bits * params.n * params.n * params.poly_len));
  size_t round_to = 64;
  size_t num_bytes_rounded_up = ((num_bits + round_to - 1) / round_to) * round_to / 8;
  std::vector<uint8_t> result(num_bytes_rounded_up, 0);
  size_t bit_offs = 0;
  for (size_t instance = 0; instance < params.instances; instance++) {
    const PolyMatrixRaw& packed_ct = v_packed_ct[instance];
    PolyMatrixRaw first_row = packed_ct.submatrix(0, 0, 1, packed_ct.cols);
    PolyMatrixRaw rest_rows = packed_ct.submatrix(1, 0, packed_ct.rows - 1, packed_ct.cols);
    for (auto& x : first_row.data) x = rescale(x, params.modulus, q2);
    for (auto& x : rest_rows.data) x = rescale(x, params.modulus, q1);
    for (size_t i = 0; i < params.n * params.poly_len; i++) {
      write_arbitrary_bits(result.data(), first_row.data[i], bit_offs, q2_bits);
      bit_offs += q2_bits;
    }
    for (size_t i = 0; i < params.n * params.n * params.poly_len; i++) {
      write_arbitrary_bits(result.data(), rest_rows.data[i], bit_offs, q1_bits);
      bit_offs += q1_bits;
    }
  }
  return result;
}

// server.rs:505-523
std::vector<PolyMatrixNTT> get_v_folding_neg(const Params& params, const std::vector<PolyMatrixNTT>& v_folding) {
  PolyMatrixNTT gadget_ntt = to_ntt_alloc(build_gadget(params, 2, 2 * params.t_gsw));
  std::vector<PolyMatrixNTT> v_folding_neg;
  for (size_t i = 0; i < params.db_dim_2; i++) {
    PolyMatrixRaw ct_gsw_inv(&params, 2, 2 * params.t_gsw);
    invert(ct_gsw_inv, from_ntt_alloc(v_folding[i]));
    PolyMatrixNTT ct_gsw_neg(&params, 2, 2 * params.t_gsw);
    add(ct_gsw_neg, gadget_ntt, to_ntt_alloc(ct_gsw_inv));
    v_folding_neg.push_back(ct_gsw_neg);
  }
  return v_folding_neg;
}

// server.rs:525-591
void expand_query(const Params& params, const PublicParameters& pp, const Query& query,
                  std::vector<u64>& v_reg_reoriented, std::vector<PolyMatrixNTT>& v_folding) {
  size_t dim0 = (size_t)1 << params.db_dim_1;
  size_t further_dims = params.db_dim_2;
  size_t num_bits_to_gen = params.t_gsw * further_dims + dim0;
  size_t g = log2_ceil_usize(num_bits_to_gen);
  size_t right_expanded = params.t_gsw * further_dims;
  size_t stop_round = log2_ceil_usize(right_expanded);

  std::vector<PolyMatrixNTT> v;
  for (size_t i = 0; i < ((size_t)1 << g); i++) v.emplace_back(&params, 2, 1);
  v[0].copy_into(to_ntt_alloc(query.ct), 0, 0);

  const PolyMatrixNTT& v_conversion = pp.v_conversion[0];
  const std::vector<PolyMatrixNTT>& v_w_left = pp.v_expansion_left;
  const std::vector<PolyMatrixNTT>& v_w_right = pp.has_expansion_right ? pp.v_expansion_right : v_w_left;
  std::vector<PolyMatrixNTT> v_neg1 = get_v_neg1(params);

  std::vector<PolyMatrixNTT> v_reg_inp, v_gsw_inp;
  if (further_dims > 0) {
    coefficient_expansion(v, g, stop_round, params, v_w_left, v_w_right, v_neg1, params.t_gsw * params.db_dim_2);
    ORACLE_CHECK(2 * std::max(dim0, right_expanded) <= v.size());  // Rust: index out of bounds panic
    for (size_t i = 0; i < dim0; i++) v_reg_inp.push_back(v[2 * i]);
    for (size_t i = 0; i < right_expanded; i++) v_gsw_inp.push_back(v[2 * i + 1]);
  } else {
    coefficient_expansion(v, g, 0, params, v_w_left, v_w_left, v_neg1, 0);
    for (size_t i = 0; i < dim0; i++) v_reg_inp.push_back(v[i]);
  }
  v_reg_reoriented.assign(dim0 * 2 * params.poly_len, 0);
  reorient_reg_ciphertexts(params, v_reg_reoriented.data(), v_reg_inp);

  v_folding.clear();
  for (size_t i = 0; i < params.db_dim_2; i++) v_folding.emplace_back(&params, 2, 2 * params.t_gsw);
  regev_to_gsw(v_folding, v_gsw_inp, v_conversion, params, 1, 0);
}

// server.rs:650-741
std::vector<uint8_t> process_query(const Params& params, const PublicParameters& pp, const Query& query, const u64* db) {
  size_t dim0 = (size_t)1 << params.db_dim_1;
  size_t num_per = (size_t)1 << params.db_dim_2;
  size_t db_slice_sz = dim0 * num_per * params.poly_len;
  const std::vector<PolyMatrixNTT>& v_packing = pp.v_packing;

  std::vector<u64> v_reg_reoriented;
  std::vector<PolyMatrixNTT> v_folding;
  if (params.expand_queries) {
    expand_query(params, pp, query, v_reg_reoriented, v_folding);
  } else {
    v_reg_reoriented = query.v_buf;
    for (auto& x : query.v_ct) v_folding.push_back(to_ntt_alloc(x));
  }
  std::vector<PolyMatrixNTT> v_folding_neg = get_v_folding_neg(params, v_folding);

  std::vector<PolyMatrixRaw> v_packed_ct(params.instances);
#pragma omp parallel for
  for (size_t instance = 0; instance < params.instances; instance++) {
    std::vector<PolyMatrixNTT> intermediate;
    std::vector<PolyMatrixRaw> intermediate_raw;
    for (size_t i = 0; i < num_per; i++) {
      intermediate.emplace_back(&params, 2, 1);
      intermediate_raw.emplace_back(&params, 2, 1);
    }
    std::vector<PolyMatrixRaw> v_ct;
    for (size_t trial = 0; trial < params.n * params.n; trial++) {
      size_t idx = (instance * (params.n * params.n) + trial) * db_slice_sz;
      const u64* cur_db = db + idx;
      multiply_reg_by_database(intermediate, cur_db, v_reg_reoriented.data(), params, dim0, num_per);
      // fold_ciphertexts leaves intermediate_raw at its full length (server.rs:405-426)
      for (size_t i = 0; i < intermediate.size(); i++) from_ntt(intermediate_raw[i], intermediate[i]);
      fold_ciphertexts(params, intermediate_raw, v_folding, v_folding_neg);
      v_ct.push_back(intermediate_raw[0]);
    }
    // spiral-rs packs with version 0 only (server.rs:734); version 1 follows lib/server (pack.rs:101-113)
    PolyMatrixNTT packed_ct = pack_dispatch(params, v_ct, v_packing);
    v_packed_ct[instance] = from_ntt_alloc(packed_ct);
  }
  return encode(params, v_packed_ct);
}

}  // namespace oracle
