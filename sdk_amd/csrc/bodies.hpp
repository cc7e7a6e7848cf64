// Workgroup-level bodies of the small kernels of the answer path, written against an explicit unit index instead of
// blockIdx: the stand-alone kernels (ntt.hip, elementwise.hip, fold.hip) call them with their block index.  (Round 4 also ran
// them in a loop inside one persistent "phase program" kernel with device-wide barriers between the phases -- 2.5-4x
// slower than the launches on this eight-XCD part; source and numbers under scripts/archive/r04_phase_program/ and
// profiles/r04_phase_program.md.)
#pragma once
#include "device_common.hpp"

namespace spiral {

// ---- forward NTT of one (output polynomial o, modulus c) of a FwdDesc batch ------------------------------------------
__device__ __forceinline__ void ntt_fwd_body(const DevTables& T, const FwdDesc& d, int o, int c, u32* ldsA, u32* ldsB) {
  const int tau = threadIdx.x;
  const int rows = d.rdim * d.t;
  const int per_b = rows * d.cols;
  const int b = o / per_b;
  const int rem = o - b * per_b;
  const int row = rem / d.cols, col = rem - row * d.cols;
  const int kdig = row / d.rdim, j = row - kdig * d.rdim;
  long sb = d.src_idx ? (long)d.src_idx[b] : (long)b;
  if (d.delta_off) sb = (long)(b / d.delta_inner) * d.delta_outer_stride + (b % d.delta_inner);
  const u64* src = d.src + (sb * d.src_batch_stride + (long)(d.src_row0 + j) * d.src_cols + col) * N;
  const ModConst m = T.c.mod[c];
  const int sh = kdig * d.bits;
  const bool plain = (d.bits >= 64);
  const u64 mask = plain ? ~0ULL : ((1ULL << d.bits) - 1ULL);
  u32 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    u64 x = src[tau + 256 * k];
    u64 piece = (sh >= 64) ? 0ULL : ((x >> sh) & mask);  // gadget.rs:48-53
    u32 val = (d.bits <= 28) ? (u32)piece : reduce64(piece, m);
    if (d.delta_off) {
      u64 x2 = src[(size_t)d.delta_off * N + tau + 256 * k];
      u64 piece2 = (sh >= 64) ? 0ULL : ((x2 >> sh) & mask);
      u32 val2 = (d.bits <= 28) ? (u32)piece2 : reduce64(piece2, m);
      u32 a = val >= m.q ? val - m.q : val, b2 = val2 >= m.q ? val2 - m.q : val2;  // digits may equal 2^28-1 > q
      val = b2 >= a ? b2 - a : b2 + m.q - a;
    }
    v[k] = val;
  }
  const u32* fw = T.tw + (size_t)c * 4 * N;
  ntt_fwd_block(v, tau, ldsA, ldsB, fw, fw + N, m.q, m.two_q);
  uint4* dst = reinterpret_cast<uint4*>(d.dst + ((size_t)o * 2 + c) * N + 8 * tau);
  dst[0] = make_uint4(v[0], v[1], v[2], v[3]);
  dst[1] = make_uint4(v[4], v[5], v[6], v[7]);
}

// ---- inverse NTT (both moduli) + Garner CRT -> raw u64 of one polynomial of an InvDesc batch --------------------------
// The composed value is the unique v in [0, Q) with v = x mod q0, v = y mod q1, i.e. exactly
// (x*q1*(q1^-1 mod q0) + y*q0*(q0^-1 mod q1)) mod Q of params.rs:207-214.
// blk = index of the workgroup's unit: polynomial (or scalar-only entry) in launch order
__device__ __forceinline__ void ntt_inv_body(const DevTables& T, const InvDesc& d, int blk, u32* ldsA, u32* ldsB) {
  const int tau = threadIdx.x;
  int p = blk;
  long base;
  long crt_stride = d.crt_stride, z_stride = d.z_stride;
  if (d.sweep_np > 0) {
    const long np = d.sweep_np;
    // The 16 columns ii that share a 64-byte line of the [z][ii] source should be read through ONE XCD's
    // L2 (blocks are dealt to XCDs round-robin): block b -> XCD b % 8 handles group (b/8/16)*8 + b%8.
    if ((np % 16) == 0 && ((long)d.n_polys % 128) == 0) {
      const int b = blk;
      const int xcd = b & 7, slot = b >> 3;
      const int grp = (slot >> 4) * 8 + xcd, within = slot & 15;  // group = (plane, r, ii/16)
      const int groups_per_plane = (int)(np / 16) * 2;
      const int plane_g = grp / groups_per_plane, rem_g = grp % groups_per_plane;
      const int r_g = rem_g / (int)(np / 16), iig = rem_g % (int)(np / 16);
      p = (int)(((long)plane_g * np + iig * 16 + within) * 2 + r_g);
    }
    const long ct = p >> 1, r = p & 1;
    const long plane = ct / np, ii = ct - plane * np;
    base = plane * 4 * N * np + r * 2 * N * np + ii;
    crt_stride = N * np;
    z_stride = np;
  } else if (d.idx) {
    int e = p / d.polys_per_idx, r = p - e * d.polys_per_idx;
    base = (long)d.idx[e] * d.idx_stride + (long)r * d.poly_stride;
  } else {
    base = (long)p * d.poly_stride;
  }
  // fused scalar multiply of coefficient_expansion (contiguous sources only)
  long scal_store = -1;
  if (d.scal) {
    if (p >= d.n_polys) {  // scalar-only entries: form and store, no transform
      const int e = p - d.n_polys;
      const int ct = d.scal_only_idx[e >> 1], r = e & 1;
      const long sp = ((long)(ct - d.scal_thresh) * 2 + r) * 2 * N, dp = ((long)ct * 2 + r) * 2 * N;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const uint4* a = reinterpret_cast<const uint4*>(d.src + sp + c * N + 8 * tau);     // (polynomial offsets are multiples of N)
        const uint4* b = reinterpret_cast<const uint4*>(d.scal + c * N + 8 * tau);
        const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
        const ModConst m = T.c.mod[c];
        uint4* o = reinterpret_cast<uint4*>(d.scal_dst + dp + c * N + 8 * tau);
        o[0] = make_uint4(reduce64((u64)a0.x * b0.x, m), reduce64((u64)a0.y * b0.y, m), reduce64((u64)a0.z * b0.z, m), reduce64((u64)a0.w * b0.w, m));
        o[1] = make_uint4(reduce64((u64)a1.x * b1.x, m), reduce64((u64)a1.y * b1.y, m), reduce64((u64)a1.z * b1.z, m), reduce64((u64)a1.w * b1.w, m));
      }
      return;
    }
    const int e = p / d.polys_per_idx, r = p - e * d.polys_per_idx;
    const int ct = d.idx[e];
    if (ct >= d.scal_thresh) {
      scal_store = base;                                                // destination = this ct's own slot
      base = (long)(ct - d.scal_thresh) * d.idx_stride + (long)r * d.poly_stride;  // source = v[ct - num_in]
    }
  }
  u32 res[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    const u32* src = d.src + base + (long)c * crt_stride;
    u32 v[8];
    if (z_stride == 1 && ((base | crt_stride | (scal_store < 0 ? 0 : scal_store)) & 3) == 0) {
      // contiguous polynomials (every caller but the sweep-source mode): a thread's 8 coefficients are 32 contiguous bytes -- two
      // 16-byte accesses instead of eight 4-byte ones at a 32-byte lane stride, for the source, the scalar and the stored product
      // (r06: the expansion's inverse launches ran at 15 ns per transform, three times their transform's cost)
      const uint4 a0 = reinterpret_cast<const uint4*>(src + 8 * tau)[0], a1 = reinterpret_cast<const uint4*>(src + 8 * tau)[1];
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      if (d.premod) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] %= m.q;
      }
      if (scal_store >= 0) {
        const uint4 s0 = reinterpret_cast<const uint4*>(d.scal + c * N + 8 * tau)[0], s1 = reinterpret_cast<const uint4*>(d.scal + c * N + 8 * tau)[1];
        const u32 sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = reduce64((u64)v[k] * (u64)sc[k], m);
        uint4* o = reinterpret_cast<uint4*>(d.scal_dst + scal_store + (long)c * crt_stride + 8 * tau);
        o[0] = make_uint4(v[0], v[1], v[2], v[3]);
        o[1] = make_uint4(v[4], v[5], v[6], v[7]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        u32 x = src[(long)(8 * tau + k) * z_stride];
        if (d.premod) x = x % m.q;
        if (scal_store >= 0) {
          x = reduce64((u64)x * (u64)d.scal[c * N + 8 * tau + k], m);
          d.scal_dst[scal_store + (long)c * crt_stride + 8 * tau + k] = x;
        }
        v[k] = x;
      }
    }
    const u32* iw = inv_tables(T.tw, c);
    if (c == 1) __syncthreads();
    ntt_inv_block(v, tau, ldsA, ldsB, iw, iw + N, m.q, m.two_q);
#pragma unroll
    for (int k = 0; k < 8; k++) res[c][k] = v[k];
  }
  const u32 q0 = T.c.mod[0].q, q1 = T.c.mod[1].q;
  u64* dst = d.dst + (size_t)p * N;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    u32 x = res[0][k], y = res[1][k];
    u32 xm = x >= q1 ? x - q1 : x;  // q0 < 2*q1
    u32 dd = y >= xm ? y - xm : y + q1 - xm;
    u32 qt = __umulhi(dd, T.c.q0_inv_q1_sh);
    u32 e = dd * T.c.q0_inv_q1 - qt * q1;
    e = e >= q1 ? e - q1 : e;
    u64 val = (u64)x + (u64)q0 * (u64)e;
    int z = tau + 256 * k;
    if (d.automorph_t) {  // poly.rs:393-405
      unsigned zt = (unsigned)z * (unsigned)d.automorph_t;
      unsigned num = zt >> POLY_LEN_LOG2, rem = zt & (N - 1);
      dst[rem] = (num & 1u) ? T.c.Q - val : val;
    } else if (d.addend) {
      const long ap = (long)(p / d.add_inner2) * d.add_outer_stride + (p % d.add_inner2);
      u64 sres = val + d.addend[(size_t)ap * N + z];
      dst[z] = sres >= T.c.Q ? sres - T.c.Q : sres;
    } else {
      dst[z] = val;
    }
  }
}

// ---- NTT-domain multiply-accumulate: one (batch element, 256-entry chunk of [crt][z]) ---------------------------------
// ychunk: which 256 of the 2 N [crt][z] entries (0 .. 15)
// CH operands of each matrix are loaded before they are multiplied (2 CH loads in flight per thread)
template <int CH = 28>
__device__ __forceinline__ void mac_body(const DevTables& T, const MacDesc& d, int inner, int outer, int ychunk) {
  const int e = ychunk * 256 + threadIdx.x;  // index into [crt][z]
  const int c = e >> POLY_LEN_LOG2;
  const int b = outer * d.batch_inner + inner;
  const ModConst m = T.c.mod[c];
  const u32* B = d.B + ((size_t)outer * d.B_outer_stride + (size_t)inner * d.B_inner_stride) * 2 * N + e;
  const long ob = d.out_idx ? (long)d.out_idx[b] : (long)b * d.out_batch_stride;
  const size_t PW = 2 * N;
  for (int r = 0; r < d.R; r++) {
    const u32* A = d.A + (size_t)r * (d.A_row_stride ? d.A_row_stride : d.K) * PW + e;
    const long op = (ob + (long)r * d.out_row_stride) * 2 * N + e;
    u64 acc = d.addend ? (u64)d.addend[op] : 0ULL;
    if (d.extra && r == d.extra_row) acc += (u64)d.extra[(size_t)(d.extra_idx ? d.extra_idx[b] : b) * PW + e];
    // two segments of B (k < split_k, k >= split_k), each walked 8 operands at a time with all 16
    // loads issued before the multiplies: small batches are latency-bound, not bandwidth-bound.
    // products < 2^56: <= 64 terms between Barrett folds stay < 2^63.
    for (int seg = 0; seg < 2; seg++) {
      const int k_lo = seg == 0 ? 0 : d.split_k, k_hi = seg == 0 ? min(d.split_k, d.K) : d.K;
      const u32* Bs = seg == 0 ? B : B + (size_t)(d.split_off - d.split_k) * PW;
      int k = k_lo, since = 0;
      for (; k + CH <= k_hi; k += CH) {  // 2 CH loads in flight: the small expansion rounds are pure latency
        u32 a[CH], bb[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
          a[u] = A[(size_t)(k + u) * PW];
          bb[u] = Bs[(size_t)(k + u) * PW];
        }
#pragma unroll
        for (int u = 0; u < CH; u++) acc += (u64)a[u] * (u64)bb[u];
        since += CH;
        if (since >= 56) {
          acc = reduce64(acc, m);
          since = 0;
        }
      }
      for (; k + 8 <= k_hi; k += 8) {
        u32 a[8], bb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          a[u] = A[(size_t)(k + u) * PW];
          bb[u] = Bs[(size_t)(k + u) * PW];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += (u64)a[u] * (u64)bb[u];
        since += 8;
        if (since >= 64) {
          acc = reduce64(acc, m);
          since = 0;
        }
      }
      for (; k < k_hi; k++) acc += (u64)A[(size_t)k * PW] * (u64)Bs[(size_t)k * PW];
      acc = reduce64(acc, m);
    }
    d.out[op] = (u32)acc;
  }
}

__device__ __forceinline__ void add_poly_into_body(const DevTables& T, u32* dst, const int* idx, const u32* src, int b, int ychunk) {
  const int e = ychunk * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const long dp = (long)idx[b] * 2 * N + e;
  dst[dp] = add_mod(dst[dp], src[(size_t)b * 2 * N + e], T.c.mod[c].q);
}

__device__ __forceinline__ void copy_polys_body(const CopyPolysDesc& d, int bx, int ychunk) {
  const int e = ychunk * 256 + threadIdx.x;
  const int b = bx / d.R, r = bx % d.R;
  d.dst[((long)d.dst_idx[b] + (long)r * d.dst_row_stride) * 2 * N + e] =
      d.src[((long)d.src_idx[b] + (long)r * d.src_row_stride) * 2 * N + e];
}

__device__ __forceinline__ void folding_neg_body(const DevTables& T, const FoldingNegDesc& d, int ychunk, int by, int dd) {
  const int e = ychunk * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const int col = by % d.two_t, r = by / d.two_t;
  const u32 q = T.c.mod[c].q;
  u32* row = d.mats + ((size_t)(dd * 2 + r) * 2 * d.two_t) * 2 * N;
  const u32 cv = row[(size_t)(d.two_t + col) * 2 * N + e];
  const u32 g = d.gadget_ntt[((size_t)r * d.two_t + col) * 2 * N + e];
  row[(size_t)col * 2 * N + e] = add_mod(g, cv ? q - cv : 0u, q);
}

// tile over (z, j), BOTH rows r: bx -> j tile, by -> z tile; tile = 2 x 32 x 33 u64 of LDS.  (r06: with one row per workgroup
// every 16-byte (j, r = 0 | 1) pair of the output was written half by one workgroup and half by another -- 8 of every 16 bytes per
// store instruction; with both rows a wave stores 512 contiguous bytes per z.)
__device__ __forceinline__ void reorient_body(const ReorientDesc& d, int bx, int by, u64* tile) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int j0 = bx * 32, z0 = by * 32;
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int j = j0 + ty + 8 * i;
      if (j < d.dim0) {
        const u32* p = d.v + ((size_t)(d.first + d.step * j) * 2 + r) * 2 * N;
        tile[r * (32 * 33) + (ty + 8 * i) * 33 + tx] = (u64)p[z0 + tx] | ((u64)p[N + z0 + tx] << 32);
      }
    }
  __syncthreads();
  const int w = threadIdx.x & 63, jl = w >> 1, r = w & 1, zr = threadIdx.x >> 6;  // 64 (j, r) pairs x 4 z per pass
  if (j0 + jl < d.dim0) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int zl = zr + 4 * i;
      d.out[((size_t)(z0 + zl) * d.dim0 + j0 + jl) * 2 + r] = tile[r * (32 * 33) + jl * 33 + zl];
    }
  }
}

// fold_mats (NTT polynomials [crt][z]) -> wave layout (wave_layout_word), same polynomial order: one thread per word
__host__ __device__ inline int wave_layout_word(int n) { return ((n & 31) >> 2) * 256 + (n >> 5) * 4 + (n & 3); }
__device__ __forceinline__ void mats_to_wave_body(const MatsToWaveDesc& d, size_t blk) {
  const size_t idx = blk * 256 + threadIdx.x;
  if (idx >= d.n_words) return;
  const size_t poly = idx >> POLY_LEN_LOG2;  // (polynomial, crt) pairs are N words each
  if (d.half_polys > 0 && (((poly >> 1) / (size_t)d.half_polys) & 1) == 0) return;   // a G - C polynomial: not needed
  const int n = (int)(idx & (N - 1));
  d.dst[poly * N + wave_layout_word(n)] = d.src[idx];
}

// rescale(a, Q, out_mod) of arith.rs:429-444 without 128-bit division: the truncated quotient
// floor((|v| * out_mod + Q/2) / Q) is < 2^38, so a double estimate is within +-1 and is corrected exactly.
__device__ __forceinline__ u64 rescale_dev(u64 a, u64 Q, u64 out_mod) {
  u64 v = a % Q;
  const bool neg = v >= Q / 2;  // inp_val -= inp_mod
  const u64 mag = neg ? Q - v : v;
  // num = mag * out_mod + Q/2  (up to ~2^93): 128-bit as (hi, lo)
  u64 lo = mag * out_mod, hi = __umul64hi(mag, out_mod);
  const u64 half = Q / 2;
  lo += half;
  hi += lo < half ? 1 : 0;
  u64 qd = (u64)(((double)hi * 18446744073709551616.0 + (double)lo) / (double)Q);
  // correct: want qd*Q <= num < (qd+1)*Q
  for (int it = 0; it < 4; it++) {
    const u64 plo = qd * Q, phi = __umul64hi(qd, Q);
    const bool gt = phi > hi || (phi == hi && plo > lo);  // qd*Q > num
    if (gt) {
      qd--;
      continue;
    }
    // rem = num - qd*Q  (fits 64 bits when qd is within 1 of the truth and Q < 2^57)
    const u64 rlo = lo - plo, rhi = hi - phi - (lo < plo ? 1 : 0);
    if (rhi != 0 || rlo >= Q) {
      qd++;
      continue;
    }
    break;
  }
  // result = (sign*qd + (Q/out)*out + 2*out) % out, then (+out) % out; all terms fit i64 magnitudes
  const u64 base = (Q / out_mod) * out_mod + 2 * out_mod;
  const u64 r = neg ? (base - qd) % out_mod : (base + qd) % out_mod;
  return (r + out_mod) % out_mod;
}

__device__ __forceinline__ void encode_body(const EncodeDesc& d, int blk) {
  const int per_inst_first = d.n * N, per_inst_rest = d.n * d.n * N;
  const int per_inst = per_inst_first + per_inst_rest;
  const long i = (long)blk * 256 + threadIdx.x;
  if (i >= (long)d.instances * per_inst) return;
  const int inst = (int)(i / per_inst), k = (int)(i % per_inst);
  const u64* m = d.packed + (size_t)inst * (d.n + 1) * d.n * N;
  const size_t inst_bits = (size_t)per_inst_first * d.q2_bits + (size_t)per_inst_rest * d.q1_bits;
  u64 val;
  size_t bit;
  int nb;
  if (k < per_inst_first) {
    val = rescale_dev(m[k], d.Q, d.q2);
    nb = d.q2_bits;
    bit = (size_t)inst * inst_bits + (size_t)k * d.q2_bits;
  } else {
    const int kk = k - per_inst_first;
    val = rescale_dev(m[per_inst_first + kk], d.Q, d.q1);
    nb = d.q1_bits;
    bit = (size_t)inst * inst_bits + (size_t)per_inst_first * d.q2_bits + (size_t)kk * d.q1_bits;
  }
  val &= nb >= 64 ? ~0ULL : ((1ULL << nb) - 1ULL);
  const size_t w = bit >> 6;
  const int off = (int)(bit & 63);
  atomicOr(d.out + w, (unsigned long long)(val << off));
  if (off + nb > 64) atomicOr(d.out + w + 1, (unsigned long long)(val >> (64 - off)));
}

}  // namespace spiral
