// Device-side helpers shared by the kernel translation units: modular arithmetic, the 2048-point negacyclic NTT
// core (register passes + LDS exchanges) and the lane <-> limb mapping of the 7-byte PACKED database format.
#pragma once
#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>

namespace spiral {

// ------------------------------------------------------------------------------------------------
// modular helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 reduce64(u64 x, const ModConst m) {
  // x mod q for any u64 x:  floor(x * floor(2^64/q) / 2^64) is floor(x/q) or one less
  u64 qe = __umul64hi(x, m.m64);
  u64 r = x - qe * (u64)m.q;
  u32 r32 = (u32)r;  // r < 2q < 2^29
  return r32 >= m.q ? r32 - m.q : r32;
}

// A database / query word as the sweep kernels need it: both 32-bit limbs reduced mod their prime.  The reference
// multiplies the limbs as they come into u128 sums and takes one % q at the end (server.rs:196-217), so reducing
// them first yields the same residues for ANY input word, while the kernels' u64 accumulation of 256 products and
// the 28-bit PACKED format both rely on limbs < q < 2^28.
__device__ __forceinline__ u64 canon_word(u64 w) {
  return (u64)((u32)w % (u32)MODULUS_0) | ((u64)((u32)(w >> 32) % (u32)MODULUS_1) << 32);
}

__device__ __forceinline__ u32 add_mod(u32 a, u32 b, u32 q) {
  u32 s = a + b;
  return s >= q ? s - q : s;
}

// Cooley-Tukey butterfly with Shoup quotient (ntt.rs:92-103): x,y in [0,4q) -> [0,4q)
__device__ __forceinline__ void ct_bfly(u32& x, u32& y, u32 w, u32 wp, u32 q, u32 q2) {
  u32 cx = x - (x >= q2 ? q2 : 0u);
  u32 qt = __umulhi(y, wp);
  u32 qn = w * y - qt * q;
  x = cx + qn;
  y = cx + q2 - qn;
}
// Gentleman-Sande butterfly, lazy: x, y in [0, 2q) -> [0, 2q).  The reference halves in every butterfly (ntt.rs:236-249:
// (x + y) / 2 and (x - y) w / 2 with w / 2 in the table -- thirteen instructions); here w is the UNHALVED twiddle
// (inv_tables) and the factor N^-1 = 2^-11 is applied once, in the last stage (gs_bfly_last): nine instructions.  The residues
// mod q are the same in every position after the last stage, so the canonical results are too.
// Shoup product w t - floor(t w' / 2^32) q: in [0, q + t q / 2^32), i.e. < 2q for any t < 2^32.
__device__ __forceinline__ void gs_bfly(u32& x, u32& y, u32 w, u32 wp, u32 q, u32 q2) {
  const u32 u = x + y;        // < 4q
  const u32 tt = x - y + q2;  // in (0, 4q)
  const u32 ur = u - q2;      // wraps above u when u < 2q
  x = u < ur ? u : ur;
  const u32 ht = __umulhi(tt, wp);
  y = w * tt - ht * q;
}
// last stage (distance N/2): x' = (x + y) N^-1, y' = (x - y) w N^-1; (ninv, ninvp) and (w, wp) = entries 0 and 1 of inv_tables
__device__ __forceinline__ void gs_bfly_last(u32& x, u32& y, u32 ninv, u32 ninvp, u32 w, u32 wp, u32 q, u32 q2) {
  const u32 u = x + y;
  const u32 tt = x - y + q2;
  x = ninv * u - __umulhi(u, ninvp) * q;
  y = w * tt - __umulhi(tt, wp) * q;
}

// ------------------------------------------------------------------------------------------------------------------
// Software pipelining: with 128 accumulator registers next to a transform only two waves fit a SIMD, so nothing but
// the code itself hides latencies.  Every table read is therefore issued one stage (or one group of butterflies)
// before its use and pinned there with a scheduling barrier; the caller can hook its own loads and arithmetic into the
// last stages (the fold kernel fetches its multiply-accumulate operands there and consumes finished quarters).
// ------------------------------------------------------------------------------------------------------------------
#define SP_SB() __builtin_amdgcn_sched_barrier(0)

// B independent Cooley-Tukey butterflies issued phase by phase: left to itself the compiler emits one butterfly after
// the other, each a chain of dependent multiplies (v_mul_hi -> v_mul_lo -> v_sub -> v_add) that a wave can only issue at
// the dependent-operation latency; phase-wise every instruction's operands are B issue slots old.
// LAZY range reduction: the Shoup product w y - floor(w' y / 2^32) q lies in [0, 2q) for ANY y < 2^32, so only the
// operand that is added (x) has to be kept small enough for x + 2q not to wrap -- and with q < 2^28 a 32-bit word holds
// 16q.  Both outputs are < x + 2q: values grow by 2q per stage.  Instead of the reference's conditional subtraction of
// 2q in every butterfly (ntt.rs:92-103; two of nine instructions) x is reduced by 8q in two of the eleven stages only
// (CORR; see wntt_fwd for the bounds).  The residues mod q are the same, so every canonical result is too.
// Five instructions per butterfly instead of seven: with NEGATED twiddles nw = -w (mod 2^32; the LDS copy and the scalar
// entries are negated once, wtw_stage / wntt_scalar_tw) the Shoup product comes out negated,
//     nl = floor(w' y / 2^32) q - w y   (mod 2^32)   = -(w y mod q, lazily in [0, 2q)),
// as two chained v_mad_u64_u32 (nw * y, then + qt * q; only the low 32 bits of the sum are used -- the second one is inline
// assembly because the compiler would narrow it to v_mul_lo + v_add) in place of v_mul_lo, v_mul_lo, v_sub, and the
// two outputs are
//     y' = x + 2q + nl  (v_add3_u32)      x' = x - nl
// in place of add, sub, add.  Same residues as before in every register (all arithmetic is mod 2^32).
template <int B, bool CORR>
__device__ __forceinline__ void ct_bfly_batch(u32 (&x)[B], u32 (&y)[B], const u32 (&nw)[B], const u32 (&wp)[B], u32 q, u32 q2) {
  u32 qt[B], t[B];
  u64 nl[B];
#pragma unroll
  for (int b = 0; b < B; b++) qt[b] = __umulhi(y[b], wp[b]);
  SP_SB();
#pragma unroll
  for (int b = 0; b < B; b++) nl[b] = (u64)nw[b] * y[b];
  SP_SB();
  if (CORR) {
#pragma unroll
    for (int b = 0; b < B; b++) t[b] = x[b] - 4 * q2;
    SP_SB();
  }
  // (one assembly statement per four: between separate statements the compiler puts an s_nop each)
  static_assert(B % 4 == 0, "butterfly batches come in fours");
#pragma unroll
  for (int b = 0; b < B; b += 4)
    asm("v_mad_u64_u32 %0, vcc, %4, %8, %0\n\tv_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %6, %8, %2\n\tv_mad_u64_u32 %3, vcc, %7, %8, %3"
        : "+v"(nl[b]), "+v"(nl[b + 1]), "+v"(nl[b + 2]), "+v"(nl[b + 3])
        : "v"(qt[b]), "v"(qt[b + 1]), "v"(qt[b + 2]), "v"(qt[b + 3]), "s"(q)
        : "vcc");
  SP_SB();
  if (CORR) {
#pragma unroll
    for (int b = 0; b < B; b++) x[b] = x[b] < t[b] ? x[b] : t[b];  // x - (x >= 8q ? 8q : 0)
    SP_SB();
  }
#pragma unroll
  for (int b = 0; b < B; b++) y[b] = x[b] + q2 + (u32)nl[b];
  SP_SB();
#pragma unroll
  for (int b = 0; b < B; b++) x[b] = x[b] - (u32)nl[b];
  SP_SB();
}
// LDS index padding.  PAD_A keeps the {tau+256k}, {256b+o+32k} and {32b+o+4k} access patterns
// conflict-free for 4-byte accesses; PAD_B does the same for {32b+o+4k} and {8tau+k}.
#define PAD_A(a) ((a) + (((a) >> 5) << 2))
#define PAD_B(a) ((a) + ((a) >> 3))
constexpr int LDS_WORDS = N + N / 8;  // >= max(PAD_A, PAD_B)

// One register pass of the forward transform on the 8 elements {b*8S + o + S*k}: half-distances
// 4S, 2S, S (stage A is skipped for the final S = 1 pass, where only distances 2 and 1 remain).
template <int S, bool DO_A>
__device__ __forceinline__ void fwd_pass(u32 (&v)[8], int b, const u32* __restrict__ fw, const u32* __restrict__ fwp,
                                         u32 q, u32 q2) {
  if (DO_A) {
    int i = N / (8 * S) + b;
    u32 w = fw[i], wp = fwp[i];
#pragma unroll
    for (int k = 0; k < 4; k++) ct_bfly(v[k], v[k + 4], w, wp, q, q2);
  }
  {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int i = N / (4 * S) + 2 * b + h;
      u32 w = fw[i], wp = fwp[i];
      ct_bfly(v[4 * h + 0], v[4 * h + 2], w, wp, q, q2);
      ct_bfly(v[4 * h + 1], v[4 * h + 3], w, wp, q, q2);
    }
  }
  {
#pragma unroll
    for (int h = 0; h < 4; h++) {
      int i = N / (2 * S) + 4 * b + h;
      ct_bfly(v[2 * h], v[2 * h + 1], fw[i], fwp[i], q, q2);
    }
  }
}
// Inverse: half-distances S, 2S, 4S on the same element set.
template <int S, bool DO_A>
__device__ __forceinline__ void inv_pass(u32 (&v)[8], int b, const u32* __restrict__ iw, const u32* __restrict__ iwp,
                                         u32 q, u32 q2) {
  {
#pragma unroll
    for (int h = 0; h < 4; h++) {
      int i = N / (2 * S) + 4 * b + h;
      gs_bfly(v[2 * h], v[2 * h + 1], iw[i], iwp[i], q, q2);
    }
  }
  {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int i = N / (4 * S) + 2 * b + h;
      u32 w = iw[i], wp = iwp[i];
      gs_bfly(v[4 * h + 0], v[4 * h + 2], w, wp, q, q2);
      gs_bfly(v[4 * h + 1], v[4 * h + 3], w, wp, q, q2);
    }
  }
  if (DO_A) {
    int i = N / (8 * S) + b;
    u32 w = iw[i], wp = iwp[i];
    if (S == 256) {  // distance N/2: the last stage (i = 1)
      const u32 ninv = iw[0], ninvp = iwp[0];
#pragma unroll
      for (int k = 0; k < 4; k++) gs_bfly_last(v[k], v[k + 4], ninv, ninvp, w, wp, q, q2);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) gs_bfly(v[k], v[k + 4], w, wp, q, q2);
    }
  }
}

// ---- lazy forward passes (r05): the cooperative transform with the wave transform's five-instruction butterfly ----------------
// ct_bfly_batch keeps only the added operand small: values grow by 2q per stage and are cut back by 8q in stages 7 (distance
// 16) and 10 (distance 2) -- bounds as in wntt_fwd: < 2q in, < 14q before either cut, < 12q after stage 11 -- and every residue
// equals the reference's.  One pass = the three stages (two in the last pass) on the 8 elements a thread holds, for M
// polynomials at once: the butterflies of a stage that share nothing go out as one phase-ordered batch of 4 M.
// nw = 0 - w (negated twiddles, computed here: one instruction per twiddle, shared by M polynomials).
template <int S, bool DO_A, int M>
__device__ __forceinline__ void fwd_pass_lazy_m(u32 (&v)[M][8], int b, const u32* __restrict__ fw, const u32* __restrict__ fwp,
                                                u32 q, u32 q2) {
  constexpr bool CORR_A = (S == 4);     // stage 7: distance 4 S = 16
  constexpr bool CORR_MID = (S == 1);   // stage 10: distance 2 S = 2
  if (DO_A) {
    const int i = N / (8 * S) + b;
    const u32 nw1 = 0u - fw[i], wp1 = fwp[i];
    u32 x[4 * M], y[4 * M], nw[4 * M], wp[4 * M];
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        x[4 * m + k] = v[m][k]; y[4 * m + k] = v[m][k + 4]; nw[4 * m + k] = nw1; wp[4 * m + k] = wp1;
      }
    ct_bfly_batch<4 * M, CORR_A>(x, y, nw, wp, q, q2);
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        v[m][k] = x[4 * m + k]; v[m][k + 4] = y[4 * m + k];
      }
  }
  {
    u32 nwh[2], wph[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int i = N / (4 * S) + 2 * b + h;
      nwh[h] = 0u - fw[i];
      wph[h] = fwp[i];
    }
    u32 x[4 * M], y[4 * M], nw[4 * M], wp[4 * M];
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int k = 0; k < 4; k++) {   // butterfly k: h = k / 2, elements 4 h + (k % 2) and + 2
        const int h = k >> 1, e = 4 * h + (k & 1);
        x[4 * m + k] = v[m][e]; y[4 * m + k] = v[m][e + 2]; nw[4 * m + k] = nwh[h]; wp[4 * m + k] = wph[h];
      }
    ct_bfly_batch<4 * M, CORR_MID>(x, y, nw, wp, q, q2);
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int h = k >> 1, e = 4 * h + (k & 1);
        v[m][e] = x[4 * m + k]; v[m][e + 2] = y[4 * m + k];
      }
  }
  {
    u32 nwh[4], wph[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const int i = N / (2 * S) + 4 * b + h;
      nwh[h] = 0u - fw[i];
      wph[h] = fwp[i];
    }
    u32 x[4 * M], y[4 * M], nw[4 * M], wp[4 * M];
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int h = 0; h < 4; h++) {
        x[4 * m + h] = v[m][2 * h]; y[4 * m + h] = v[m][2 * h + 1]; nw[4 * m + h] = nwh[h]; wp[4 * m + h] = wph[h];
      }
    ct_bfly_batch<4 * M, false>(x, y, nw, wp, q, q2);
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int h = 0; h < 4; h++) {
        v[m][2 * h] = x[4 * m + h]; v[m][2 * h + 1] = y[4 * m + h];
      }
  }
}
// < 12q (after the last lazy stage) -> canonical (ntt.rs:107-111 from the reference's < 4q)
__device__ __forceinline__ u32 canon_from_12q(u32 x, u32 q, u32 q2) {
  x -= (x >= 4 * q2 ? 4 * q2 : 0u);
  x -= (x >= 2 * q2 ? 2 * q2 : 0u);
  x -= (x >= q2 ? q2 : 0u);
  x -= (x >= q ? q : 0u);
  return x;
}

// Forward 2048-point negacyclic NTT of the 8 values per thread held in pattern {tau + 256k}
// (natural order in), leaving the result in pattern {8 tau + k} (reference output order).
template <int M>
__device__ __forceinline__ void ntt_fwd_block_m(u32 (&v)[M][8], int tau, u32* la, u32* lb, const u32* __restrict__ fw,
                                                const u32* __restrict__ fwp, u32 q, u32 q2);
// inputs < 2q (residues, or digits of at most 28 bits); r05: the lazy passes above (fwd_pass / ct_bfly remain for reference)
__device__ __forceinline__ void ntt_fwd_block(u32 (&v)[8], int tau, u32* ldsA, u32* ldsB, const u32* __restrict__ fw,
                                              const u32* __restrict__ fwp, u32 q, u32 q2) {
  u32 w[1][8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[0][k] = v[k];
  ntt_fwd_block_m<1>(w, tau, ldsA, ldsB, fw, fwp, q, q2);
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = w[0][k];
}

// Inverse: values in pattern {8 tau + k} (< 2q) -> pattern {tau + 256k}, canonical.  iw / iwp: inv_tables (NOT the reference's
// halved tables).
__device__ __forceinline__ void ntt_inv_block(u32 (&v)[8], int tau, u32* ldsA, u32* ldsB, const u32* __restrict__ iw,
                                              const u32* __restrict__ iwp, u32 q, u32 q2) {
  inv_pass<1, false>(v, tau, iw, iwp, q, q2);
#pragma unroll
  for (int k = 0; k < 8; k++) ldsA[PAD_B(8 * tau + k)] = v[k];
  __syncthreads();
  {
    int b = tau >> 2, o = tau & 3;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = ldsA[PAD_B(32 * b + o + 4 * k)];
    inv_pass<4, true>(v, b, iw, iwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) ldsB[PAD_A(32 * b + o + 4 * k)] = v[k];
  }
  __syncthreads();
  {
    int b = tau >> 5, o = tau & 31;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = ldsB[PAD_A(256 * b + o + 32 * k)];
    inv_pass<32, true>(v, b, iw, iwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) ldsA[PAD_A(256 * b + o + 32 * k)] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = ldsA[PAD_A(tau + 256 * k)];
  inv_pass<256, true>(v, 0, iw, iwp, q, q2);
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] -= (v[k] >= q ? q : 0u);  // ntt.rs:253-256, from < 2q
}

// ---- M transforms at once per thread: the twiddles, Shoup quotients and LDS addresses are computed once
// and applied to M independent coefficient vectors (same modulus); each barrier serves M transforms.
// la / lb hold M consecutive LDS_WORDS-sized buffers.
template <int S, bool DO_A, int M>
__device__ __forceinline__ void fwd_pass_m(u32 (&v)[M][8], int b, const u32* __restrict__ fw,
                                           const u32* __restrict__ fwp, u32 q, u32 q2) {
  if (DO_A) {
    int i = N / (8 * S) + b;
    u32 w = fw[i], wp = fwp[i];
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int k = 0; k < 4; k++) ct_bfly(v[m][k], v[m][k + 4], w, wp, q, q2);
  }
#pragma unroll
  for (int h = 0; h < 2; h++) {
    int i = N / (4 * S) + 2 * b + h;
    u32 w = fw[i], wp = fwp[i];
#pragma unroll
    for (int m = 0; m < M; m++) {
      ct_bfly(v[m][4 * h + 0], v[m][4 * h + 2], w, wp, q, q2);
      ct_bfly(v[m][4 * h + 1], v[m][4 * h + 3], w, wp, q, q2);
    }
  }
#pragma unroll
  for (int h = 0; h < 4; h++) {
    int i = N / (2 * S) + 4 * b + h;
    u32 w = fw[i], wp = fwp[i];
#pragma unroll
    for (int m = 0; m < M; m++) ct_bfly(v[m][2 * h], v[m][2 * h + 1], w, wp, q, q2);
  }
}
template <int S, bool DO_A, int M>
__device__ __forceinline__ void inv_pass_m(u32 (&v)[M][8], int b, const u32* __restrict__ iw,
                                           const u32* __restrict__ iwp, u32 q, u32 q2) {
#pragma unroll
  for (int h = 0; h < 4; h++) {
    int i = N / (2 * S) + 4 * b + h;
    u32 w = iw[i], wp = iwp[i];
#pragma unroll
    for (int m = 0; m < M; m++) gs_bfly(v[m][2 * h], v[m][2 * h + 1], w, wp, q, q2);
  }
#pragma unroll
  for (int h = 0; h < 2; h++) {
    int i = N / (4 * S) + 2 * b + h;
    u32 w = iw[i], wp = iwp[i];
#pragma unroll
    for (int m = 0; m < M; m++) {
      gs_bfly(v[m][4 * h + 0], v[m][4 * h + 2], w, wp, q, q2);
      gs_bfly(v[m][4 * h + 1], v[m][4 * h + 3], w, wp, q, q2);
    }
  }
  if (DO_A) {
    int i = N / (8 * S) + b;
    u32 w = iw[i], wp = iwp[i];
    if (S == 256) {  // distance N/2: the last stage (i = 1)
      const u32 ninv = iw[0], ninvp = iwp[0];
#pragma unroll
      for (int m = 0; m < M; m++)
#pragma unroll
        for (int k = 0; k < 4; k++) gs_bfly_last(v[m][k], v[m][k + 4], ninv, ninvp, w, wp, q, q2);
    } else {
#pragma unroll
      for (int m = 0; m < M; m++)
#pragma unroll
        for (int k = 0; k < 4; k++) gs_bfly(v[m][k], v[m][k + 4], w, wp, q, q2);
    }
  }
}
template <int M>
__device__ __forceinline__ void ntt_fwd_block_m(u32 (&v)[M][8], int tau, u32* la, u32* lb,
                                                const u32* __restrict__ fw, const u32* __restrict__ fwp, u32 q, u32 q2) {
  fwd_pass_lazy_m<256, true, M>(v, 0, fw, fwp, q, q2);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int a = PAD_A(tau + 256 * k);
#pragma unroll
    for (int m = 0; m < M; m++) la[m * LDS_WORDS + a] = v[m][k];
  }
  __syncthreads();
  {
    int b = tau >> 5, o = tau & 31;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_A(256 * b + o + 32 * k);
#pragma unroll
      for (int m = 0; m < M; m++) v[m][k] = la[m * LDS_WORDS + a];
    }
    fwd_pass_lazy_m<32, true, M>(v, b, fw, fwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_A(256 * b + o + 32 * k);
#pragma unroll
      for (int m = 0; m < M; m++) lb[m * LDS_WORDS + a] = v[m][k];
    }
  }
  __syncthreads();
  {
    int b = tau >> 2, o = tau & 3;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_A(32 * b + o + 4 * k);
#pragma unroll
      for (int m = 0; m < M; m++) v[m][k] = lb[m * LDS_WORDS + a];
    }
    fwd_pass_lazy_m<4, true, M>(v, b, fw, fwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_B(32 * b + o + 4 * k);
#pragma unroll
      for (int m = 0; m < M; m++) la[m * LDS_WORDS + a] = v[m][k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int a = PAD_B(8 * tau + k);
#pragma unroll
    for (int m = 0; m < M; m++) v[m][k] = la[m * LDS_WORDS + a];
  }
  fwd_pass_lazy_m<1, false, M>(v, tau, fw, fwp, q, q2);
#pragma unroll
  for (int m = 0; m < M; m++)
#pragma unroll
    for (int k = 0; k < 8; k++) v[m][k] = canon_from_12q(v[m][k], q, q2);
}
template <int M>
__device__ __forceinline__ void ntt_inv_block_m(u32 (&v)[M][8], int tau, u32* la, u32* lb,
                                                const u32* __restrict__ iw, const u32* __restrict__ iwp, u32 q, u32 q2) {
  inv_pass_m<1, false, M>(v, tau, iw, iwp, q, q2);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int a = PAD_B(8 * tau + k);
#pragma unroll
    for (int m = 0; m < M; m++) la[m * LDS_WORDS + a] = v[m][k];
  }
  __syncthreads();
  {
    int b = tau >> 2, o = tau & 3;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_B(32 * b + o + 4 * k);
#pragma unroll
      for (int m = 0; m < M; m++) v[m][k] = la[m * LDS_WORDS + a];
    }
    inv_pass_m<4, true, M>(v, b, iw, iwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_A(32 * b + o + 4 * k);
#pragma unroll
      for (int m = 0; m < M; m++) lb[m * LDS_WORDS + a] = v[m][k];
    }
  }
  __syncthreads();
  {
    int b = tau >> 5, o = tau & 31;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_A(256 * b + o + 32 * k);
#pragma unroll
      for (int m = 0; m < M; m++) v[m][k] = lb[m * LDS_WORDS + a];
    }
    inv_pass_m<32, true, M>(v, b, iw, iwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = PAD_A(256 * b + o + 32 * k);
#pragma unroll
      for (int m = 0; m < M; m++) la[m * LDS_WORDS + a] = v[m][k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int a = PAD_A(tau + 256 * k);
#pragma unroll
    for (int m = 0; m < M; m++) v[m][k] = la[m * LDS_WORDS + a];
  }
  inv_pass_m<256, true, M>(v, 0, iw, iwp, q, q2);
#pragma unroll
  for (int m = 0; m < M; m++)
#pragma unroll
    for (int k = 0; k < 8; k++) v[m][k] -= (v[m][k] >= q ? q : 0u);  // ntt.rs:253-256, from < 2q
}

// Dword offset of unit (zp = plane*N + z, row pair jp, 128-column chunk) in the PACKED database.  The units one
// sweep wave reads (fixed zp and chunk, jp = 0 .. npairs-1) are contiguous: every wave walks ONE sequential stream of
// npairs * 1792 bytes instead of striding chunks * 1792 bytes between row pairs (measured 9.5 -> 9.0 ms per C2 sweep on
// the boxes where the strided order was slow, profiles/r01_overlap.md section 7).
__device__ __forceinline__ size_t packed_unit_offset(size_t zp, int jp, int chunk, int npairs, int chunks) {
  return ((zp * (size_t)chunks + (size_t)chunk) * (size_t)npairs + (size_t)jp) * 448;
}

// Packing helpers shared by the writers of the PACKED format.
__device__ __forceinline__ void pack_unit_lane(u32* unit, int lane, u64 w00, u64 w01, u64 w10, u64 w11) {
  // w{row}{iiofs}: limbs f0..f7 = lo/hi of w00, w01, w10, w11
  const u32 M = 0x0FFFFFFFu;
  const u32 f0 = (u32)w00 & M, f1 = (u32)(w00 >> 32) & M, f2 = (u32)w01 & M, f3 = (u32)(w01 >> 32) & M;
  const u32 f4 = (u32)w10 & M, f5 = (u32)(w10 >> 32) & M, f6 = (u32)w11 & M, f7 = (u32)(w11 >> 32) & M;
  u32* p4 = unit + lane * 4;
  u32* p3 = unit + 256 + lane * 3;
  p4[0] = f0 | (f1 << 28);
  p4[1] = (f1 >> 4) | (f2 << 24);
  p4[2] = (f2 >> 8) | (f3 << 20);
  p4[3] = (f3 >> 12) | (f4 << 16);
  p3[0] = (f4 >> 16) | (f5 << 12);
  p3[1] = (f5 >> 20) | (f6 << 8);
  p3[2] = (f6 >> 24) | (f7 << 4);
}
__device__ __forceinline__ u64 unpack_word(const u32* unit, int lane, int which) {  // which = row*2 + iiofs
  const u32* p4 = unit + lane * 4;
  const u32* p3 = unit + 256 + lane * 3;
  u32 dd[8] = {p4[0], p4[1], p4[2], p4[3], p3[0], p3[1], p3[2], 0u};
  u32 f[2];
  for (int h = 0; h < 2; h++) {
    const int bit = 28 * (which * 2 + h);
    const int w = bit >> 5, sh = bit & 31;
    u64 two = (u64)dd[w] | ((u64)dd[w + 1 < 8 ? w + 1 : 7] << 32);
    f[h] = (u32)(two >> sh) & 0x0FFFFFFFu;
  }
  return (u64)f[0] | ((u64)f[1] << 32);
}

// query qi's buffer from query 0's pointer (GroupOff, kernels.hpp): a null pointer stays null
template <typename P>
__device__ __forceinline__ P* group_rebase(P* p, long long byte_off) {
  return p ? reinterpret_cast<P*>(reinterpret_cast<unsigned long long>(p) + (unsigned long long)byte_off) : p;
}

}  // namespace spiral
