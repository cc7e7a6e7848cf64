// Fused fold step of fold_ciphertexts (server.rs:388-427) for gfx950: digits -> NTT -> multiply-accumulate -> iNTT -> CRT
// in registers / LDS.  See DESIGN.md section 3 for the algebra and the roofline.
#include <type_traits>

#include "device_common.hpp"
#include "wave_ntt.hpp"
#include "bodies.hpp"

namespace spiral {

// ------------------------------------------------------------------------------------------------
// fused fold step.  grid (half, planes).
//   reference (server.rs:407-424):  out = from_ntt( (G - C) * NTT(G^-1(ct_i)) + C * NTT(G^-1(ct_{i+half})) )
//   computed as:                    out = ct_i + from_ntt( C * NTT(G^-1(ct_{i+half}) - G^-1(ct_i)) )   (mod Q)
// which is the same element of Z_Q[x]/(x^N+1): G * G^-1(ct_i) = ct_i exactly (the digits recompose the
// coefficient, gadget.rs:11-60), NTT / from_ntt are linear, and both sides are the canonical representative
// in [0, Q).  That halves the transforms: per modulus 2t forward NTTs of digit differences, each consumed at
// once by the 2-row multiply-accumulate against C, then two inverse NTTs; Garner; add ct_i.
// Requires v_folding_neg == G - v_folding, which holds inside process_query by construction.
// The two LDS buffers alternate roles between consecutive transforms: three barriers per transform.
// ------------------------------------------------------------------------------------------------
// lib/server/src/compute/fold.rs:38-44 ("crucial for correctness" there): ct_i all zero -> the step yields ct_{i+half};
// ct_{i+half} all zero -> ct_i stays.  Returns true when the workgroup is done.
__device__ __forceinline__ bool fold_zero_shortcut(const u64* ct0, const u64* ct1, u64* out, int tau) {
  int nz0 = 0, nz1 = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    nz0 |= ct0[tau + 256 * k] != 0;
    nz1 |= ct1[tau + 256 * k] != 0;
  }
  const int any0 = __syncthreads_or(nz0), any1 = __syncthreads_or(nz1);
  if (any0 && any1) return false;
  const u64* src = any0 ? ct0 : ct1;  // ct_i unless it is the all-zero one
#pragma unroll
  for (int k = 0; k < 16; k++) out[tau + 256 * k] = src[tau + 256 * k];
  return true;
}

__global__ __launch_bounds__(256, 2) void k_fold_fused(DevTables T, FoldDesc d) {
  __shared__ u32 lds0[LDS_WORDS];
  __shared__ u32 lds1[LDS_WORDS];
  const int tau = threadIdx.x;
  const int i = (int)blockIdx.x, plane = blockIdx.y;
  const int two_t = 2 * d.t, four_t = 4 * d.t;
  const int t_live = d.t_live > 0 && d.t_live < d.t ? d.t_live : d.t;
  const u64* ct0 = d.X + ((size_t)plane * d.cur + i) * 2 * N;
  const u64* ct1 = ct0 + (size_t)d.half * 2 * N;
  const u64 mask = (1ULL << d.bits) - 1ULL;
  u32* la = lds0;
  u32* lb = lds1;
  u64* out = d.Y + ((size_t)plane * d.half + i) * 2 * N;
  if (d.zero_shortcuts && fold_zero_shortcut(ct0, ct1, out, tau)) return;
#pragma unroll 1
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    const u32* fw = T.tw + (size_t)c * 4 * N;
    u64 acc0[8], acc1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc0[k] = acc1[k] = 0;
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
      u64 x0[8], x1[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        x0[k] = ct0[(size_t)j * N + tau + 256 * k];
        x1[k] = ct1[(size_t)j * N + tau + 256 * k];
      }
#pragma unroll 1
      for (int kd = 0; kd < t_live; kd++) {   // (the digits above t_live are identically zero: FoldDesc::t_live)
        const int sh = kd * d.bits;
        u32 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          u32 d0 = sh >= 64 ? 0u : (u32)((x0[k] >> sh) & mask);
          u32 d1 = sh >= 64 ? 0u : (u32)((x1[k] >> sh) & mask);
          if (d.bits >= 28) {  // 28-bit digits can exceed q (q < 2^28 < 2q): canonical residues first
            d0 = d0 >= m.q ? d0 - m.q : d0;
            d1 = d1 >= m.q ? d1 - m.q : d1;
          }
          v[k] = d1 >= d0 ? d1 - d0 : d1 + m.q - d0;
        }
        const u32* fwk = fw;
        int tk = tau;
        // keep twiddle loads and LDS address arithmetic inside the loop (fewer live VGPRs)
        asm volatile("" : "+s"(fwk));
        asm volatile("" : "+v"(tk));
        ntt_fwd_block(v, tk, la, lb, fwk, fwk + N, m.q, m.two_q);
        {
          u32* tmp = la;
          la = lb;
          lb = tmp;
        }
        const int kk = two_t + j + 2 * kd;  // column of C inside the [G-C | C] row
        const uint4* a0 = reinterpret_cast<const uint4*>(d.mats + ((size_t)kk * 2 + c) * N + 8 * tk);
        const uint4* a1 = reinterpret_cast<const uint4*>(d.mats + ((size_t)(four_t + kk) * 2 + c) * N + 8 * tk);
        const uint4 p0 = a0[0], p1 = a0[1], r0 = a1[0], r1 = a1[1];
        acc0[0] += (u64)p0.x * v[0]; acc0[1] += (u64)p0.y * v[1]; acc0[2] += (u64)p0.z * v[2]; acc0[3] += (u64)p0.w * v[3];
        acc0[4] += (u64)p1.x * v[4]; acc0[5] += (u64)p1.y * v[5]; acc0[6] += (u64)p1.z * v[6]; acc0[7] += (u64)p1.w * v[7];
        acc1[0] += (u64)r0.x * v[0]; acc1[1] += (u64)r0.y * v[1]; acc1[2] += (u64)r0.z * v[2]; acc1[3] += (u64)r0.w * v[3];
        acc1[4] += (u64)r1.x * v[4]; acc1[5] += (u64)r1.y * v[5]; acc1[6] += (u64)r1.z * v[6]; acc1[7] += (u64)r1.w * v[7];
      }
    }
    const u32* iw = inv_tables(T.tw, c);
    u32 v0[8], v1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      v0[k] = reduce64(acc0[k], m);
      v1[k] = reduce64(acc1[k], m);
    }
    ntt_inv_block(v0, tau, la, lb, iw, iw + N, m.q, m.two_q);
    ntt_inv_block(v1, tau, lb, la, iw, iw + N, m.q, m.two_q);
    if (c == 0) {
      // park the modulus-0 residues in the output slot (the same thread re-reads them below)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        out[tau + 256 * k] = v0[k];
        out[(size_t)N + tau + 256 * k] = v1[k];
      }
    } else {
      const u32 q0 = T.c.mod[0].q, q1 = T.c.mod[1].q;
#pragma unroll
      for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const size_t zi = (size_t)r * N + tau + 256 * k;
          u32 x = (u32)out[zi], y = r == 0 ? v0[k] : v1[k];
          u32 xm = x >= q1 ? x - q1 : x;
          u32 dd = y >= xm ? y - xm : y + q1 - xm;
          u32 qt = __umulhi(dd, T.c.q0_inv_q1_sh);
          u32 e = dd * T.c.q0_inv_q1 - qt * q1;
          e = e >= q1 ? e - q1 : e;
          u64 val = (u64)x + (u64)q0 * (u64)e + ct0[zi];  // ct_i < Q (from_ntt output)
          out[zi] = val >= T.c.Q ? val - T.c.Q : val;
        }
      }
    }
  }
}
// k_fold_fused with two digit transforms in flight per thread (shared twiddles / addresses / barriers).
// Requires an even digit count t.
// The forward twiddles + Shoup quotients of the current modulus (16 KiB) are staged in LDS once per modulus, so the
// 4 x 14 twiddle reads of every digit transform are ds_reads (no vector-memory latency, and no queueing behind a
// concurrent sweep's load stream when the fold runs on the second stream).
__global__ __launch_bounds__(256, 2) void k_fold_fused2(DevTables T, FoldDesc d) {
  __shared__ u32 lds0[2 * LDS_WORDS];
  __shared__ u32 lds1[2 * LDS_WORDS];
  __shared__ u32 ltw[2 * N];
  const int tau = threadIdx.x;
  const int i = (int)blockIdx.x, plane = blockIdx.y;
  const int two_t = 2 * d.t, four_t = 4 * d.t;
  const u64* ct0 = d.X + ((size_t)plane * d.cur + i) * 2 * N;
  const u64* ct1 = ct0 + (size_t)d.half * 2 * N;
  const u64 mask = (1ULL << d.bits) - 1ULL;
  u32* la = lds0;
  u32* lb = lds1;
  u64* out = d.Y + ((size_t)plane * d.half + i) * 2 * N;
  if (d.zero_shortcuts && fold_zero_shortcut(ct0, ct1, out, tau)) return;
#pragma unroll 1
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    const u32* fw = T.tw + (size_t)c * 4 * N;
    if (c == 1) __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++)  // [fw | fwp] = 2N words
      reinterpret_cast<uint4*>(ltw)[tau + 256 * k] = reinterpret_cast<const uint4*>(fw)[tau + 256 * k];
    __syncthreads();
    u64 acc0[8], acc1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc0[k] = acc1[k] = 0;
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
      u64 x0[8], x1[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        x0[k] = ct0[(size_t)j * N + tau + 256 * k];
        x1[k] = ct1[(size_t)j * N + tau + 256 * k];
      }
#pragma unroll 1
      for (int kd = 0; kd < d.t; kd += 2) {
        u32 v[2][8];
#pragma unroll
        for (int mm = 0; mm < 2; mm++) {
          const int sh = (kd + mm) * d.bits;
#pragma unroll
          for (int k = 0; k < 8; k++) {
            u32 d0 = sh >= 64 ? 0u : (u32)((x0[k] >> sh) & mask);
            u32 d1 = sh >= 64 ? 0u : (u32)((x1[k] >> sh) & mask);
            if (d.bits >= 28) {  // 28-bit digits can exceed q (q < 2^28 < 2q): canonical residues first
              d0 = d0 >= m.q ? d0 - m.q : d0;
              d1 = d1 >= m.q ? d1 - m.q : d1;
            }
            v[mm][k] = d1 >= d0 ? d1 - d0 : d1 + m.q - d0;
          }
        }
        int tk = tau;
        asm volatile("" : "+v"(tk));
        ntt_fwd_block_m<2>(v, tk, la, lb, ltw, ltw + N, m.q, m.two_q);
        {
          u32* tmp = la;
          la = lb;
          lb = tmp;
        }
#pragma unroll
        for (int mm = 0; mm < 2; mm++) {
          const int kk = two_t + j + 2 * (kd + mm);
          const uint4* a0 = reinterpret_cast<const uint4*>(d.mats + ((size_t)kk * 2 + c) * N + 8 * tk);
          const uint4* a1 = reinterpret_cast<const uint4*>(d.mats + ((size_t)(four_t + kk) * 2 + c) * N + 8 * tk);
          const uint4 p0 = a0[0], p1 = a0[1], r0 = a1[0], r1 = a1[1];
          acc0[0] += (u64)p0.x * v[mm][0]; acc0[1] += (u64)p0.y * v[mm][1]; acc0[2] += (u64)p0.z * v[mm][2]; acc0[3] += (u64)p0.w * v[mm][3];
          acc0[4] += (u64)p1.x * v[mm][4]; acc0[5] += (u64)p1.y * v[mm][5]; acc0[6] += (u64)p1.z * v[mm][6]; acc0[7] += (u64)p1.w * v[mm][7];
          acc1[0] += (u64)r0.x * v[mm][0]; acc1[1] += (u64)r0.y * v[mm][1]; acc1[2] += (u64)r0.z * v[mm][2]; acc1[3] += (u64)r0.w * v[mm][3];
          acc1[4] += (u64)r1.x * v[mm][4]; acc1[5] += (u64)r1.y * v[mm][5]; acc1[6] += (u64)r1.z * v[mm][6]; acc1[7] += (u64)r1.w * v[mm][7];
        }
      }
    }
    const u32* iw = inv_tables(T.tw, c);
    u32 vv[2][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      vv[0][k] = reduce64(acc0[k], m);
      vv[1][k] = reduce64(acc1[k], m);
    }
    ntt_inv_block_m<2>(vv, tau, la, lb, iw, iw + N, m.q, m.two_q);
    if (c == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        out[tau + 256 * k] = vv[0][k];
        out[(size_t)N + tau + 256 * k] = vv[1][k];
      }
    } else {
      const u32 q0 = T.c.mod[0].q, q1 = T.c.mod[1].q;
#pragma unroll
      for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const size_t zi = (size_t)r * N + tau + 256 * k;
          u32 x = (u32)out[zi], y = vv[r][k];
          u32 xm = x >= q1 ? x - q1 : x;
          u32 dd = y >= xm ? y - xm : y + q1 - xm;
          u32 qt = __umulhi(dd, T.c.q0_inv_q1_sh);
          u32 e = dd * T.c.q0_inv_q1 - qt * q1;
          e = e >= q1 ? e - q1 : e;
          u64 val = (u64)x + (u64)q0 * (u64)e + ct0[zi];
          out[zi] = val >= T.c.Q ? val - T.c.Q : val;
        }
      }
    }
  }
}
// ------------------------------------------------------------------------------------------------
// Fused fold step on the wave-per-transform NTT (wave_ntt.hpp).  grid (half, planes), 4 waves per step.
// Same algebra as k_fold_fused*: out = ct_i + from_ntt( C * NTT(G^-1(ct_{i+half}) - G^-1(ct_i)) ).
//   prologue   the 2t digit differences d = G^-1(ct_{i+half})_dg - G^-1(ct_i)_dg are small SIGNED integers that do
//              not depend on the modulus: all 256 threads fetch the two ciphertexts once (32 loads in flight per
//              thread) and park the differences in LDS: the low 8 ES bits in a byte/short/word plane per digit, the
//              sign in a bit plane (ES = 1 for the 8-bit digits of t_gsw = 8: 9-bit differences in 36 KiB)
//   per modulus  the digit polynomials are dealt to the four waves (digit dg -> wave dg % 4); a wave reads a digit from
//              LDS (two b128 per lane at ES = 1), transforms it without any workgroup barrier and multiply-accumulates
//              it into private 64-bit sums for both output rows (32 coefficients per lane) -- the operands are fetched
//              during the transform's last stages and consumed quarter by quarter as it finishes (FoldMac);
//              the four partial sums are combined through LDS (two rounds), wave 0 ends up with row 0 and wave 1 with
//              row 1
//   at the end  four inverse transforms (2 rows x 2 moduli), one per wave; Garner and + ct_i as before.
// mats_w: the level's [G-C | C] operands in wave layout (wave_layout_word), same polynomial order as FoldDesc::mats.
// LDS: [4 transpose buffers 18 KiB | forward tables of the current modulus 16 KiB | digit planes 2t * 2048 * ES | sign planes 2t * 256];
// the first 32 KiB double as the cross-wave reduction scratch.
// ------------------------------------------------------------------------------------------------
constexpr int FOLD_WAVE_FIXED_LDS = (4 * WBUF_WORDS + 2 * N) * 4;

struct FoldMac {  // hooks into wntt_fwd: operand vectors of coefficient group g (4 coefficients per lane) for both rows
  u64 (&acc0)[32];
  u64 (&acc1)[32];
  const u32x4w_t* a0;
  const u32x4w_t* a1;
  u32x4w_t m0[8], m1[8];
  __device__ __forceinline__ void fetch(int g) {
    m0[g] = a0[64 * g];
    m1[g] = a1[64 * g];
  }
  __device__ __forceinline__ void mac(int g, const u32 (&v)[32]) {
    acc0[4 * g] += (u64)m0[g].x * v[4 * g]; acc0[4 * g + 1] += (u64)m0[g].y * v[4 * g + 1];
    acc0[4 * g + 2] += (u64)m0[g].z * v[4 * g + 2]; acc0[4 * g + 3] += (u64)m0[g].w * v[4 * g + 3];
    acc1[4 * g] += (u64)m1[g].x * v[4 * g]; acc1[4 * g + 1] += (u64)m1[g].y * v[4 * g + 1];
    acc1[4 * g + 2] += (u64)m1[g].z * v[4 * g + 2]; acc1[4 * g + 3] += (u64)m1[g].w * v[4 * g + 3];
  }
  __device__ __forceinline__ void before_t4() {
    fetch(0); fetch(1); fetch(2); fetch(3);
  }
  __device__ __forceinline__ void before_t1() {
    fetch(4); fetch(5);
  }
  __device__ __forceinline__ void after_quarter(int qq, u32 (&v)[32]) {  // quarter qq = coefficients 8qq .. 8qq+7, final
    if (qq == 0) {
      fetch(6); fetch(7);
      SP_SB();
    }
    mac(2 * qq, v);
    mac(2 * qq + 1, v);
  }
};

template <int ES>
__global__ __launch_bounds__(256, 2) void k_fold_wave(DevTables T, FoldDesc d, const u32* __restrict__ mats_w) {
  extern __shared__ __attribute__((aligned(16))) u32 smem_fw[];
  u32* wbuf = smem_fw;                      // transposes (one region per wave) ...
  u32* ltw = smem_fw + 4 * WBUF_WORDS;      // forward tables of the current modulus (swizzled, wtw_stage)
  // live digits only (FoldDesc::t_live): plane li = j * t_live + kd holds digit kd of row j
  const int t_live = d.t_live > 0 && d.t_live < d.t ? d.t_live : d.t;
  unsigned char* dig = reinterpret_cast<unsigned char*>(ltw + 2 * N);
  unsigned char* sgn = dig + (size_t)2 * t_live * (N * ES);  // sign bits: plane li, lane l -> one dword, bit k = coefficient 64k + l
  constexpr int E = 16 / ES;                // digit differences per 16-byte vector
  const int tau = threadIdx.x, lane = tau & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tau >> 6);  // wave-uniform: digit index and row pointers stay scalar
  const int i = (int)blockIdx.x, plane = blockIdx.y;
  const int two_t = 2 * d.t, four_t = 4 * d.t;
  const u64* ct0 = d.X + ((size_t)plane * d.cur + i) * 2 * N;
  const u64* ct1 = ct0 + (size_t)d.half * 2 * N;
  u64* out = d.Y + ((size_t)plane * d.half + i) * 2 * N;
  if (d.zero_shortcuts && fold_zero_shortcut(ct0, ct1, out, tau)) return;
  u32* mybuf = wbuf + wv * WBUF_WORDS;
  // which planes a wave transforms: wv0, wv0 + 4, ...  When 2 t_live is not a multiple of 4 (14 live digits of t_gsw = 8) some
  // waves get one transform less; the assignment is rotated by the workgroup's parity so that the two workgroups sharing a CU
  // do not put their short waves on the same pair of SIMDs
  const int wv0 = (wv + 2 * ((int)blockIdx.x & 1)) & 3;
  {  // digit differences of coefficients 64k + lane, k = 8 wv .. 8 wv + 7, both rows.  Element (digit, n = 64k + lane) lives
     // at byte (digit * 2048 * ES) + (k / E) * 1024 + lane * 16 + (k % E) * ES: a lane's 32 values are 2 ES vectors of
     // 16 bytes, vector h of all lanes one contiguous KiB (conflict-free b128 reads)
    const u64 mask = (1ULL << d.bits) - 1ULL;
    u64 x0[2][8], x1[2][8];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const size_t n = (size_t)j * N + 64 * (8 * wv + e) + lane;
        x0[j][e] = ct0[n];
        x1[j][e] = ct1[n];
      }
#pragma unroll 1
    for (int kd = 0; kd < t_live; kd++) {
      const int sh = (kd * d.bits) & 63;
      const u64 dmask = kd * d.bits >= 64 ? 0ULL : mask;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        u32 df[8];
#pragma unroll
        for (int e = 0; e < 8; e++) df[e] = (u32)((x1[j][e] >> sh) & dmask) - (u32)((x0[j][e] >> sh) & dmask);
        unsigned char* base = dig + (size_t)(j * t_live + kd) * (N * ES);
        u32 sb = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) sb |= (df[e] >> 31) << e;
        sgn[(j * t_live + kd) * 256 + lane * 4 + wv] = (unsigned char)sb;
        if (ES == 1) {
          u32x2w_t o;
          o.x = (df[0] & 0xff) | ((df[1] & 0xff) << 8) | ((df[2] & 0xff) << 16) | (df[3] << 24);
          o.y = (df[4] & 0xff) | ((df[5] & 0xff) << 8) | ((df[6] & 0xff) << 16) | (df[7] << 24);
          *reinterpret_cast<u32x2w_t*>(base + (wv >> 1) * 1024 + lane * 16 + 8 * (wv & 1)) = o;
        } else if (ES == 2) {
          u32x4w_t o;
          o.x = (df[0] & 0xffff) | (df[1] << 16); o.y = (df[2] & 0xffff) | (df[3] << 16);
          o.z = (df[4] & 0xffff) | (df[5] << 16); o.w = (df[6] & 0xffff) | (df[7] << 16);
          *reinterpret_cast<u32x4w_t*>(base + wv * 1024 + lane * 16) = o;
        } else {
          u32x4w_t o;
          o.x = df[0]; o.y = df[1]; o.z = df[2]; o.w = df[3];
          *reinterpret_cast<u32x4w_t*>(base + (2 * wv) * 1024 + lane * 16) = o;
          o.x = df[4]; o.y = df[5]; o.z = df[6]; o.w = df[7];
          *reinterpret_cast<u32x4w_t*>(base + (2 * wv + 1) * 1024 + lane * 16) = o;
        }
      }
    }
  }
#pragma unroll 1
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    const u32* fw = T.tw + (size_t)c * 4 * N;
    if (c == 1) __syncthreads();  // the reduction scratch of modulus 0 overlapped the table area
    wtw_stage(ltw, wave_fwd_image(T.tw, c), tau);
    __syncthreads();              // tables (and, the first time, the digit differences) are in place
    u64 acc0[32], acc1[32];
#pragma unroll
    for (int k = 0; k < 32; k++) acc0[k] = acc1[k] = 0;
#pragma unroll 1
    for (int dg = wv0; dg < 2 * t_live; dg += 4) {   // dg: plane index (live digits of row 0, then of row 1)
      const int j = dg / t_live, kd = dg - j * t_live;
      // everything derived from the lane id or the table pointer is loop invariant; left alone the compiler hoists some
      // sixty addresses and as many scalar twiddles out of this loop and spills the accumulators to make room for them
      int ln = lane;
      const u32* fwi = fw;
      asm volatile("" : "+v"(ln));
      asm volatile("" : "+s"(fwi));
      const unsigned char* src = dig + (size_t)dg * (N * ES) + ln * 16;
      u32x4w_t dv[2 * ES];
#pragma unroll
      for (int h = 0; h < 2 * ES; h++) dv[h] = *reinterpret_cast<const u32x4w_t*>(src + h * 1024);
      const int sg = *reinterpret_cast<const int*>(sgn + dg * 256 + ln * 4);
      WaveScalarTw stw;
      wntt_scalar_tw(stw, fwi);
      SP_SB();
      u32 v[32];
#pragma unroll
      for (int h = 0; h < 2 * ES; h++) {
        const u32 wd[4] = {dv[h].x, dv[h].y, dv[h].z, dv[h].w};
#pragma unroll
        for (int e = 0; e < E; e++) {
          // low 8 ES bits from the digit plane, everything above from the sign plane (two's complement)
          const int sm = __builtin_amdgcn_sbfe(sg, h * E + e, 1);
          const int sd = ES == 4 ? (int)wd[e] : (int)(__builtin_amdgcn_ubfe(wd[(e * ES) >> 2], ((e * ES) & 3) * 8, 8 * ES) | ((u32)sm << (8 * ES)));
          u32 x = (u32)(sd + (sm & (int)m.q));  // |difference| < q unless the digits have 28+ bits
          if (ES == 4) {
            x += (u32)(((int)x >> 31) & (int)m.q);
            x -= (x >= m.q ? m.q : 0u);
          }
          v[h * E + e] = x;
        }
      }
      const int kk = two_t + j + 2 * kd;  // column of C inside the [G-C | C] row
      FoldMac hk{acc0, acc1, reinterpret_cast<const u32x4w_t*>(mats_w + ((size_t)kk * 2 + c) * N) + ln,
                 reinterpret_cast<const u32x4w_t*>(mats_w + ((size_t)(four_t + kk) * 2 + c) * N) + ln};
      wntt_fwd<false>(v, ln, mybuf, fwi, stw, ltw, m.q, m.two_q, hk);
    }
    // partial sums of the four waves -> wave 0 (row 0) and wave 1 (row 1): every wave reduces its 64 sums, the u32 residues
    // travel through LDS as b128 vectors at [(g * 64 + lane)] (conflict-free) in two rounds; region s holds one row of one wave
    // (2048 words).  (r04 tried the alternative -- the four waves' raw 64-bit sums added exactly in LDS with ds_add_u64 and
    // reduced once: 1655 instead of 2125 vector instructions per transform, 3 instead of 5 barriers -- and measured it on one
    // allocation: equal alone and in batches, 1.5-3 % SLOWER for the pipelined query, whose sweeps lose more beside it:
    // profiles/r04_fold_from_ntt_ab.md.)
    int lt = lane;  // (as above: keeps the tail's addresses from being hoisted over the digit loop)
    asm volatile("" : "+v"(lt));
    // (parking layout: vector g of all lanes contiguous -- whole 1-KiB runs per store instruction; with a 128-byte lane stride
    // the 16-byte pieces made the kernel write 3.3x its algorithmic bytes, profiles/r04_final_pmc_per_kernel_unpipelined.md;
    // measured on one allocation: -1 % on the un-pipelined fold and on 8-query steps, profiles/r04_fold_park_ab_raw.txt)
    u32* park = reinterpret_cast<u32*>(out);
    const int irow = wv & 1, imod = wv < 2 ? 1 : 0;
    u32 r0[32], r1[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
      r0[k] = reduce64(acc0[k], m);
      r1[k] = reduce64(acc1[k], m);
    }
    u32x4w_t* sc = reinterpret_cast<u32x4w_t*>(smem_fw);
#define SP_PUT(R, REGION)                                                                              \
_Pragma("unroll") for (int g = 0; g < 8; g++) {                                                      \
  u32x4w_t t4;                                                                                       \
  t4.x = R[4 * g]; t4.y = R[4 * g + 1]; t4.z = R[4 * g + 2]; t4.w = R[4 * g + 3];                    \
  sc[(REGION) * 512 + g * 64 + lt] = t4;                                                             \
}
#define SP_ADD(R, REGION)                                                                              \
_Pragma("unroll") for (int g = 0; g < 8; g++) {                                                      \
  const u32x4w_t t4 = sc[(REGION) * 512 + g * 64 + lt];                                              \
  R[4 * g] = add_mod(R[4 * g], t4.x, m.q); R[4 * g + 1] = add_mod(R[4 * g + 1], t4.y, m.q);          \
  R[4 * g + 2] = add_mod(R[4 * g + 2], t4.z, m.q); R[4 * g + 3] = add_mod(R[4 * g + 3], t4.w, m.q);  \
}
    __syncthreads();  // every wave is done with its transpose buffer and the tables
    if (wv == 2) { SP_PUT(r0, 0) SP_PUT(r1, 1) }
    if (wv == 3) { SP_PUT(r0, 2) SP_PUT(r1, 3) }
    __syncthreads();
    if (wv == 0) { SP_ADD(r0, 0) SP_ADD(r0, 2) }
    if (wv == 1) { SP_ADD(r1, 1) SP_ADD(r1, 3) }
    __syncthreads();
    if (wv == 0) { SP_PUT(r1, 0) }
    if (wv == 1) { SP_PUT(r0, 1) }
    __syncthreads();
    if (wv == 0) { SP_ADD(r0, 1) }
    if (wv == 1) { SP_ADD(r1, 0) }
    __syncthreads();  // the scratch is free again: waves 0 and 1 use their own regions for the inverse transform
#undef SP_PUT
#undef SP_ADD
    if (c == 0) {
      if (wv < 2) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
          u32x4w_t t4;
          t4.x = wv == 0 ? r0[4 * g] : r1[4 * g]; t4.y = wv == 0 ? r0[4 * g + 1] : r1[4 * g + 1];
          t4.z = wv == 0 ? r0[4 * g + 2] : r1[4 * g + 2]; t4.w = wv == 0 ? r0[4 * g + 3] : r1[4 * g + 3];
          *reinterpret_cast<u32x4w_t*>(park + wv * N + g * 256 + 4 * lt) = t4;
        }
      }
    }
    if (c == 1) {
      const ModConst mi = T.c.mod[imod];
      // the row this wave transforms back.  Declared HERE, where every path defines it: declared at the top of the modulus loop
      // (as until round 4) it was undefined on the c == 0 path, the compiler carried 64 registers of zeros for it through the
      // whole kernel and spilled them -- 316 bytes of scratch per lane, ~0.9 GB of scratch stores per C2 query, most of the
      // kernel's measured 5x write amplification.  16 bytes now.
      u32 rr[32];
      if (wv < 2) {
#pragma unroll
        for (int k = 0; k < 32; k++) rr[k] = wv == 0 ? r0[k] : r1[k];
      } else {
#pragma unroll
        for (int g = 0; g < 8; g++) {
          const u32x4w_t t4 = *reinterpret_cast<const u32x4w_t*>(park + irow * N + g * 256 + 4 * lt);
          rr[4 * g] = t4.x; rr[4 * g + 1] = t4.y; rr[4 * g + 2] = t4.z; rr[4 * g + 3] = t4.w;
        }
      }
      // compose + add ct_i + store: each of the four waves does HALF a row (waves 0 / 2 share row 0, waves 1 / 3 row 1; the
      // modulus-1 waves 0, 1 take coefficients 64 k + lane with k < 16, the modulus-0 waves 2, 3 those with k >= 16) -- with
      // waves 2 and 3 doing whole rows this phase was 11.7 % of the kernel (profiles/r05_fold_dissection.md).  The ct_i words
      // are requested before the inverse transform.
      const int khalf = wv >> 1;
      const u64* crow = ct0 + (size_t)irow * N + (size_t)khalf * 1024;
      u64 cpre[16];
#pragma unroll
      for (int kk = 0; kk < 16; kk++) cpre[kk] = crow[64 * kk + lt];
      wntt_inv(rr, lt, mybuf, inv_tables(T.tw, imod), mi.q, mi.two_q);  // -> coefficient 64 k + lane
      u32* exch = smem_fw + 4 * WBUF_WORDS;  // the table area: region (row, modulus) holds the 16 x 64 residues its partner needs
      u32* mine_out = exch + (irow * 2 + imod) * (N / 2);
      if (wv < 2) {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) mine_out[64 * kk + lt] = rr[16 + kk];
      } else {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) mine_out[64 * kk + lt] = rr[kk];
      }
      __syncthreads();  // (also: waves 2 and 3 have read their parked rows before anybody overwrites the output slot)
      {
        const u32* theirs = exch + (irow * 2 + (1 - imod)) * (N / 2);
        u64* orow = out + (size_t)irow * N + (size_t)khalf * 1024;
        const u32 q0 = T.c.mod[0].q, q1 = T.c.mod[1].q;
        // (two copies of the loop with compile-time register indices: a run-time choice between rr[kk] and rr[16 + kk] turns rr
        // into an indexed array in scratch memory)
        auto compose = [&](auto mod1_wave) {
          constexpr bool M1 = decltype(mod1_wave)::value;
#pragma unroll
          for (int kk = 0; kk < 16; kk++) {
            const u32 other = theirs[64 * kk + lt];
            const u32 own = rr[M1 ? kk : 16 + kk];
            const u32 x = M1 ? other : own, y = M1 ? own : other;   // residues mod q0, q1
            const u32 xm = x >= q1 ? x - q1 : x;
            const u32 dd = y >= xm ? y - xm : y + q1 - xm;
            const u32 qt = __umulhi(dd, T.c.q0_inv_q1_sh);
            u32 e = dd * T.c.q0_inv_q1 - qt * q1;
            e = e >= q1 ? e - q1 : e;
            const u64 val = (u64)x + (u64)q0 * (u64)e + cpre[kk];
            const u64 res = val >= T.c.Q ? val - T.c.Q : val;
            orow[64 * kk + lt] = res;
          }
        };
        if (wv < 2) compose(std::true_type{});
        else compose(std::false_type{});
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// One expansion round's MANY-DIGIT side on the wave-per-transform NTT (r06): k_expand_round's work for the ciphertexts of ONE side,
// in the shape of k_fold_wave -- a workgroup per (ciphertext, modulus, query), the t digit polynomials of the ciphertext's first row
// dealt to four waves (digit dg -> wave dg % 4), each transformed without a workgroup barrier and multiply-accumulated into the
// wave's private 64-bit sums of both output rows (FoldMac: the side's expansion key in WAVE layout, ExpandWaveDesc::A_w), the four
// partial sums combined through LDS, + the addend, stored.  The transform of the ciphertext's second row rides along as one more
// "digit" whose operands are the constant polynomials 0 (row 0) and 1 (row 1) (ExpandWaveDesc::const_w).
// Why: k_expand_round runs at its cooperative transform core's rate (3.75 ns per transform at two workgroups per CU,
// profiles/r01_ntt_core.md); the wave transform runs at 2.0 -- but only fused with its consumer (profiles/r05_fwd_wave.md).  A
// right-hand ciphertext has 56 one-bit digits: 14 per wave, the table staging and the reduction paid once per 57 transforms.
// Sums: at most 15 terms of < 12 q x q < 2^59.6 per wave -- below 2^64.
// LDS: [4 transpose buffers 18 KiB | forward tables 16 KiB | the ciphertext's first row, raw, 16 KiB].  grid (cnt, 2 moduli, queries).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_expand_wave(DevTables T, ExpandWaveDesc d, GroupOff g) {
  extern __shared__ __attribute__((aligned(16))) u32 smem_fw[];
  u32* wbuf = smem_fw;
  u32* ltw = smem_fw + 4 * WBUF_WORDS;
  u64* raw0 = reinterpret_cast<u64*>(ltw + 2 * N);
  const int tau = threadIdx.x, lane = tau & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tau >> 6);
  const int b = (int)blockIdx.x, c = (int)blockIdx.y, qi = (int)blockIdx.z;
  const ModConst m = T.c.mod[c];
  const u32* fw = T.tw + (size_t)c * 4 * N;
  const u64* src = group_rebase(d.raw, g.raw[qi]) + (size_t)d.pos[b] * 2 * N;
  const u32* A_w = group_rebase(d.A_w, g.pp[qi]);
  u32* vout = group_rebase(d.v, g.v[qi]);
  u32* mybuf = wbuf + wv * WBUF_WORDS;
  wtw_stage(ltw, wave_fwd_image(T.tw, c), tau);
#pragma unroll
  for (int k = 0; k < 8; k++) raw0[tau + 256 * k] = src[tau + 256 * k];
  __syncthreads();
  const u64 mask = (1ULL << d.bits) - 1ULL;
  u64 acc0[32], acc1[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc0[k] = acc1[k] = 0;
  // "digit" d.t is the ciphertext's second row (poly.rs:613-623: reduced mod q first), multiplied by (0, 1)
#pragma unroll 1
  for (int dg = wv; dg <= d.t; dg += 4) {
    int ln = lane;
    const u32* fwi = fw;
    asm volatile("" : "+v"(ln));     // (as in k_fold_wave: keeps the loop's addresses and scalar twiddles from being hoisted and spilled)
    asm volatile("" : "+s"(fwi));
    WaveScalarTw stw;
    wntt_scalar_tw(stw, fwi);
    u32 v[32];
    if (dg < d.t) {
      const int sh = dg * d.bits;
#pragma unroll
      for (int k = 0; k < 32; k++) v[k] = sh < 64 ? (u32)((raw0[64 * k + ln] >> (sh & 63)) & mask) : 0u;   // gadget.rs:48-53
    } else {
#pragma unroll
      for (int k = 0; k < 32; k++) v[k] = reduce64(src[(size_t)N + 64 * k + ln], m);
    }
    SP_SB();
    const u32* a0 = dg < d.t ? A_w + ((size_t)dg * 2 + c) * N : d.const_w;
    const u32* a1 = dg < d.t ? A_w + ((size_t)(d.t + dg) * 2 + c) * N : d.const_w + N;
    FoldMac hk{acc0, acc1, reinterpret_cast<const u32x4w_t*>(a0) + ln, reinterpret_cast<const u32x4w_t*>(a1) + ln};
    wntt_fwd<false>(v, ln, mybuf, fwi, stw, ltw, m.q, m.two_q, hk);
  }
  // the four waves' partial sums -> wave 0 (row 0) and wave 1 (row 1), as in k_fold_wave
  int lt = lane;
  asm volatile("" : "+v"(lt));
  u32 r0[32], r1[32];
#pragma unroll
  for (int k = 0; k < 32; k++) {
    r0[k] = reduce64(acc0[k], m);
    r1[k] = reduce64(acc1[k], m);
  }
  u32x4w_t* sc = reinterpret_cast<u32x4w_t*>(smem_fw);
#define SP_PUT(R, REGION)                                                                              \
_Pragma("unroll") for (int g4 = 0; g4 < 8; g4++) {                                                   \
  u32x4w_t t4;                                                                                       \
  t4.x = R[4 * g4]; t4.y = R[4 * g4 + 1]; t4.z = R[4 * g4 + 2]; t4.w = R[4 * g4 + 3];                \
  sc[(REGION) * 512 + g4 * 64 + lt] = t4;                                                            \
}
#define SP_ADD(R, REGION)                                                                              \
_Pragma("unroll") for (int g4 = 0; g4 < 8; g4++) {                                                   \
  const u32x4w_t t4 = sc[(REGION) * 512 + g4 * 64 + lt];                                             \
  R[4 * g4] = add_mod(R[4 * g4], t4.x, m.q); R[4 * g4 + 1] = add_mod(R[4 * g4 + 1], t4.y, m.q);      \
  R[4 * g4 + 2] = add_mod(R[4 * g4 + 2], t4.z, m.q); R[4 * g4 + 3] = add_mod(R[4 * g4 + 3], t4.w, m.q); \
}
  __syncthreads();  // every wave is done with its transpose buffer, the tables and the raw row
  if (wv == 2) { SP_PUT(r0, 0) SP_PUT(r1, 1) }
  if (wv == 3) { SP_PUT(r0, 2) SP_PUT(r1, 3) }
  __syncthreads();
  if (wv == 0) { SP_ADD(r0, 0) SP_ADD(r0, 2) }
  if (wv == 1) { SP_ADD(r1, 1) SP_ADD(r1, 3) }
  __syncthreads();
  if (wv == 0) { SP_PUT(r1, 0) }
  if (wv == 1) { SP_PUT(r0, 1) }
  __syncthreads();
  if (wv == 0) { SP_ADD(r0, 1) }
  if (wv == 1) { SP_ADD(r1, 0) }
#undef SP_PUT
#undef SP_ADD
  // register k of lane L is NTT-domain coefficient 32 L + k (wave_layout_word): the natural layout, 128 contiguous bytes per lane
  if (wv < 2) {
    u32* o = vout + ((size_t)d.out_idx[b] * 2 + c) * N + (size_t)wv * 2 * N + 32 * lt;   // row wv of the output ciphertext
#pragma unroll
    for (int g4 = 0; g4 < 8; g4++) {
      const uint4 a = reinterpret_cast<const uint4*>(o)[g4];
      uint4 w;
      w.x = add_mod(wv == 0 ? r0[4 * g4] : r1[4 * g4], a.x, m.q);
      w.y = add_mod(wv == 0 ? r0[4 * g4 + 1] : r1[4 * g4 + 1], a.y, m.q);
      w.z = add_mod(wv == 0 ? r0[4 * g4 + 2] : r1[4 * g4 + 2], a.z, m.q);
      w.w = add_mod(wv == 0 ? r0[4 * g4 + 3] : r1[4 * g4 + 3], a.w, m.q);
      reinterpret_cast<uint4*>(o)[g4] = w;
    }
  }
}
// (r06, measured and removed: the same work with ONE WAVE per (ciphertext, modulus) unit, four units per workgroup, digits cut from
// the raw words as they are read from global memory, no cross-wave reduction -- meant for the 8-digit left-hand side.  6.8 ns per
// transform against the cooperative kernel's 4.4 on the two left-hand-only rounds, 965 against 718 us on round 7's right-hand side:
// every digit waits for 32 dependent-free but un-hidden loads with two waves per SIMD.  profiles/r06_group_expansion.md,
// scripts/archive/r06_expand_unit/.)
void launch_expand_wave(const DevTables& T, const ExpandWaveDesc& d, const GroupOff& g, int B, hipStream_t s) {
  if (d.cnt <= 0 || B <= 0) return;
  const size_t lds = (size_t)(4 * WBUF_WORDS + 2 * N) * 4 + (size_t)N * 8;
  hipLaunchKernelGGL(k_expand_wave, dim3(d.cnt, 2, B), dim3(256), lds, s, T, d, g);
  launched(PATH_EXPAND_FUSED | PATH_EXPAND_WAVE, "k_expand_wave");
}

// fold_mats (NTT polynomials [crt][z]) -> wave layout, same polynomial order: one thread per word
__global__ __launch_bounds__(256) void k_mats_to_wave(MatsToWaveDesc d) { mats_to_wave_body(d, blockIdx.x); }
void launch_mats_to_wave(u32* dst, const u32* src, size_t n_words, hipStream_t s, int half_polys) {
  if (n_words == 0) return;
  const MatsToWaveDesc d{dst, src, n_words, half_polys};
  hipLaunchKernelGGL(k_mats_to_wave, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, d);
  launched(0, "k_mats_to_wave");
}

// grouped form (kernels.hpp, GroupOff): grid.y = query
__global__ __launch_bounds__(256) void k_mats_to_wave_group(MatsToWaveDesc d, GroupOff g) {
  const int qi = blockIdx.y;
  d.dst = group_rebase(d.dst, g.raw[qi]);
  d.src = group_rebase(d.src, g.v[qi]);
  mats_to_wave_body(d, blockIdx.x);
}
void launch_mats_to_wave_group(u32* dst, const u32* src, size_t n_words, int half_polys, const GroupOff& g, int B, hipStream_t s) {
  if (n_words == 0 || B <= 0) return;
  const MatsToWaveDesc d{dst, src, n_words, half_polys};
  hipLaunchKernelGGL(k_mats_to_wave_group, dim3((unsigned)((n_words + 255) / 256), B), dim3(256), 0, s, d, g);
  launched(PATH_EXPAND_GROUP, "k_mats_to_wave_group");
}

void launch_fold_fused(const DevTables& T, const FoldDesc& d, hipStream_t s) {
  if (d.half <= 0 || d.planes <= 0) return;
  // fold_variant: 5 (default) = k_fold_wave where two workgroups fit a CU's LDS, else the cooperative kernels;
  // 3 = k_fold_fused2 (two cooperative transforms in flight, even t_gsw); 0 = k_fold_fused (one).
  // (r05 built and removed a variant 6, k_fold_wave8: TWO steps per eight-wave workgroup, the work of a step split by (modulus,
  // ciphertext row) -- 7 forward transforms per wave instead of 8, one reduction and five barriers per step instead of two and
  // twelve, nothing parked in global memory.  6 % fewer vector instructions, 43 % fewer LDS instructions, byte-identical -- and
  // exactly as fast alone (241.1 vs 242.5 us per launch), slower beside the sweep, whose resident wave per SIMD leaves no room
  // for an eight-wave workgroup: profiles/r05_fold_wave8.md, scripts/archive/r05_fold_wave8/.)
  const int variant = (int)tunable("fold_variant", FOLD_VARIANT_DEFAULT);
  const dim3 grid(d.half, d.planes), block(256);
  if (variant == 5 && d.mats_w && d.bits <= 28) {  // (wider digits never reach a fused kernel: fused_fold_supported)
    const int es = d.bits <= 8 ? 1 : d.bits <= 16 ? 2 : 4;
    const int t_live = d.t_live > 0 && d.t_live < d.t ? d.t_live : d.t;
    const size_t lds = FOLD_WAVE_FIXED_LDS + (size_t)2 * t_live * (N * es + 256);
    if (lds <= 80 * 1024) {  // two workgroups per CU; with one the barrier-synchronised kernels are the faster ones
      // (> 64 KiB of dynamic LDS needs no opt-in on gfx950, scripts/ubench/dyn_lds.hip)
      if (es == 1)
        hipLaunchKernelGGL(k_fold_wave<1>, grid, block, lds, s, T, d, d.mats_w);
      else if (es == 2)
        hipLaunchKernelGGL(k_fold_wave<2>, grid, block, lds, s, T, d, d.mats_w);
      else
        hipLaunchKernelGGL(k_fold_wave<4>, grid, block, lds, s, T, d, d.mats_w);
      launched(PATH_FOLD_FUSED | PATH_FOLD_WAVE, "k_fold_wave");
      return;
    }
  }
  if (variant != 0 && (d.t % 2) == 0)
    hipLaunchKernelGGL(k_fold_fused2, grid, block, 0, s, T, d);
  else
    hipLaunchKernelGGL(k_fold_fused, grid, block, 0, s, T, d);
  launched(PATH_FOLD_FUSED, "k_fold_fused");
}

}  // namespace spiral
