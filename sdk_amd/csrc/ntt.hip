// gfx950 (MI355X, CDNA4) kernels for the Spiral PIR answer path.  Integer mod-q arithmetic on the
// vector ALU (u32 mul_lo/mul_hi, v_mad_u64_u32); no MFMA (this is not a floating-point contraction).
// wave = 64 lanes; 256-thread workgroups unless stated.
//
// Reference semantics reproduced here (lib/spiral-rs/src): ntt.rs:67-113, 212-258 (negacyclic NTT,
// Harvey lazy butterflies), poly.rs:351-663, gadget.rs:34-60, util.rs:323-355, server.rs:155-221.
// Residues leave every kernel canonical (< q), so all intermediates equal the scalar reference's.
//
// This unit: forward / inverse transform kernels, from_ntt of the sweep output, the transform-core micro-benchmark.
#include "device_common.hpp"
#include "bodies.hpp"

namespace spiral {

// ---- micro-benchmark of the transform core (no memory traffic besides twiddles): `reps` chained transforms of M
// coefficient vectors per workgroup; used by sp_bench_ntt to separate the NTT core cost from the fused kernels'.
template <int M>
__global__ __launch_bounds__(256) void k_ntt_core_bench(DevTables T, u32* out, int reps) {
  __shared__ u32 lds0[M * LDS_WORDS];
  __shared__ u32 lds1[M * LDS_WORDS];
  const int tau = threadIdx.x;
  const ModConst m = T.c.mod[blockIdx.x & 1];
  const u32* fw = T.tw + (size_t)(blockIdx.x & 1) * 4 * N;
  u32 v[M][8];
#pragma unroll
  for (int mm = 0; mm < M; mm++)
#pragma unroll
    for (int k = 0; k < 8; k++) v[mm][k] = (tau * 2654435761u + k * 40503u + mm + blockIdx.x) % m.q;
  u32* la = lds0;
  u32* lb = lds1;
  for (int r = 0; r < reps; r++) {
    const u32* fwk = fw;
    int tk = tau;
    asm volatile("" : "+s"(fwk));
    asm volatile("" : "+v"(tk));
    ntt_fwd_block_m<M>(v, tk, la, lb, fwk, fwk + N, m.q, m.two_q);
    u32* t = la;
    la = lb;
    lb = t;
  }
  u32 acc = 0;
#pragma unroll
  for (int mm = 0; mm < M; mm++)
#pragma unroll
    for (int k = 0; k < 8; k++) acc ^= v[mm][k];
  out[blockIdx.x * 256 + tau] = acc;
}
float bench_ntt_core(const DevTables& T, int M, int blocks, int reps, u32* scratch, hipStream_t s) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  auto go = [&] {
    switch (M) {
      case 1: hipLaunchKernelGGL(k_ntt_core_bench<1>, dim3(blocks), dim3(256), 0, s, T, scratch, reps); break;
      case 2: hipLaunchKernelGGL(k_ntt_core_bench<2>, dim3(blocks), dim3(256), 0, s, T, scratch, reps); break;
      default: hipLaunchKernelGGL(k_ntt_core_bench<4>, dim3(blocks), dim3(256), 0, s, T, scratch, reps); break;
    }
  };
  go();
  (void)hipEventRecord(a, s);
  go();
  (void)hipEventRecord(b, s);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return ms;
}

// ------------------------------------------------------------------------------------------------
// forward NTT kernel: grid (n_out, 2 crt)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ntt_fwd(DevTables T, FwdDesc d) {
  __shared__ u32 ldsA[LDS_WORDS];
  __shared__ u32 ldsB[LDS_WORDS];
  ntt_fwd_body(T, d, blockIdx.x, blockIdx.y, ldsA, ldsB);
}
__global__ __launch_bounds__(256) void k_ntt_fwd3(DevTables T, FwdDesc d0, FwdDesc d1, FwdDesc d2) {
  __shared__ u32 ldsA[LDS_WORDS];
  __shared__ u32 ldsB[LDS_WORDS];
  int o = blockIdx.x;
  if (o < d0.n_out) {
    ntt_fwd_body(T, d0, o, blockIdx.y, ldsA, ldsB);
  } else if (o < d0.n_out + d1.n_out) {
    ntt_fwd_body(T, d1, o - d0.n_out, blockIdx.y, ldsA, ldsB);
  } else {
    ntt_fwd_body(T, d2, o - d0.n_out - d1.n_out, blockIdx.y, ldsA, ldsB);
  }
}
// grouped form (kernels.hpp, GroupOff): grid.z = query
__global__ __launch_bounds__(256) void k_ntt_fwd3_group(DevTables T, FwdDesc d0, FwdDesc d1, FwdDesc d2, GroupOff g) {
  __shared__ u32 ldsA[LDS_WORDS];
  __shared__ u32 ldsB[LDS_WORDS];
  const int qi = blockIdx.z;
  int o = blockIdx.x;
  FwdDesc d = d0;
  long long dst_off = g.dig[qi];
  if (o >= d0.n_out + d1.n_out) {
    d = d2;
    o -= d0.n_out + d1.n_out;
    dst_off = g.ct1[qi];
  } else if (o >= d0.n_out) {
    d = d1;
    o -= d0.n_out;
  }
  d.src = group_rebase(d.src, g.raw[qi]);
  d.dst = group_rebase(d.dst, dst_off);
  ntt_fwd_body(T, d, o, blockIdx.y, ldsA, ldsB);
}
void launch_ntt_fwd3_group(const DevTables& T, const FwdDesc& d0, const FwdDesc& d1, const FwdDesc& d2, const GroupOff& g, int B,
                           hipStream_t s) {
  const int total = std::max(d0.n_out, 0) + std::max(d1.n_out, 0) + std::max(d2.n_out, 0);
  if (total <= 0 || B <= 0) return;
  FwdDesc a = d0, b = d1, c = d2;
  a.n_out = std::max(a.n_out, 0);
  b.n_out = std::max(b.n_out, 0);
  c.n_out = std::max(c.n_out, 0);
  hipLaunchKernelGGL(k_ntt_fwd3_group, dim3(total, 2, B), dim3(256), 0, s, T, a, b, c, g);
  launched(PATH_EXPAND_GROUP, "k_ntt_fwd3_group");
}
void launch_ntt_fwd(const DevTables& T, const FwdDesc& d, hipStream_t s) {
  if (d.n_out <= 0) return;
  hipLaunchKernelGGL(k_ntt_fwd, dim3(d.n_out, 2), dim3(256), 0, s, T, d);
  launched(0, "k_ntt_fwd");
}
void launch_ntt_fwd3(const DevTables& T, const FwdDesc& d0, const FwdDesc& d1, const FwdDesc& d2, hipStream_t s) {
  const int total = std::max(d0.n_out, 0) + std::max(d1.n_out, 0) + std::max(d2.n_out, 0);
  if (total <= 0) return;
  FwdDesc a = d0, b = d1, c = d2;
  a.n_out = std::max(a.n_out, 0);
  b.n_out = std::max(b.n_out, 0);
  c.n_out = std::max(c.n_out, 0);
  hipLaunchKernelGGL(k_ntt_fwd3, dim3(total, 2), dim3(256), 0, s, T, a, b, c);
  launched(0, "k_ntt_fwd3");
}

// ------------------------------------------------------------------------------------------------
// One expansion round's digit transforms and products (kernels.hpp, ExpandSideDesc): grid (left.cnt + right.cnt, 2 moduli).
// ------------------------------------------------------------------------------------------------
constexpr int EXPAND_GD = 4;   // digit polynomials per transform pass
static __device__ __forceinline__ void expand_round_body(const DevTables& T, const ExpandSideDesc& d, int b, int c, u32* lds0, u32* lds1) {
  constexpr int GD = EXPAND_GD;
  const int tau = threadIdx.x;
  const ModConst m = T.c.mod[c];
  const u32* fw = T.tw + (size_t)c * 4 * N;
  const u64* src = d.raw + (size_t)d.pos[b] * 2 * N;
  u64 x[8];
#pragma unroll
  for (int k = 0; k < 8; k++) x[k] = src[tau + 256 * k];
  const u64 mask = (1ULL << d.bits) - 1ULL;   // (bits <= 28: launch_expand_round)
  u64 acc0[8], acc1[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc0[k] = acc1[k] = 0;
#pragma unroll 1
  for (int k0 = 0; k0 < d.t; k0 += GD) {
    u32 v[GD][8];
#pragma unroll
    for (int mm = 0; mm < GD; mm++) {
      const int sh = (k0 + mm) * d.bits;
      const bool live = k0 + mm < d.t && sh < 64;
#pragma unroll
      for (int k = 0; k < 8; k++) v[mm][k] = live ? (u32)((x[k] >> (sh & 63)) & mask) : 0u;   // gadget.rs:48-53
    }
    int tk = tau;
    const u32* fwk = fw;
    asm volatile("" : "+v"(tk));    // (addresses recomputed per pass instead of hoisted and spilled)
    asm volatile("" : "+s"(fwk));
    ntt_fwd_block_m<GD>(v, tk, lds0, lds1, fwk, fwk + N, m.q, m.two_q);   // -> element 8 tau + k, canonical
#pragma unroll
    for (int mm = 0; mm < GD; mm++) {
      const int kk = k0 + mm;
      if (kk < d.t) {
        const uint4* a0 = reinterpret_cast<const uint4*>(d.A + ((size_t)kk * 2 + c) * N + 8 * tk);
        const uint4* a1 = reinterpret_cast<const uint4*>(d.A + ((size_t)(d.t + kk) * 2 + c) * N + 8 * tk);
        const uint4 p0 = a0[0], p1 = a0[1], r0 = a1[0], r1 = a1[1];
        acc0[0] += (u64)p0.x * v[mm][0]; acc0[1] += (u64)p0.y * v[mm][1]; acc0[2] += (u64)p0.z * v[mm][2]; acc0[3] += (u64)p0.w * v[mm][3];
        acc0[4] += (u64)p1.x * v[mm][4]; acc0[5] += (u64)p1.y * v[mm][5]; acc0[6] += (u64)p1.z * v[mm][6]; acc0[7] += (u64)p1.w * v[mm][7];
        acc1[0] += (u64)r0.x * v[mm][0]; acc1[1] += (u64)r0.y * v[mm][1]; acc1[2] += (u64)r0.z * v[mm][2]; acc1[3] += (u64)r0.w * v[mm][3];
        acc1[4] += (u64)r1.x * v[mm][4]; acc1[5] += (u64)r1.y * v[mm][5]; acc1[6] += (u64)r1.z * v[mm][6]; acc1[7] += (u64)r1.w * v[mm][7];
      }
    }
    if (((k0 / GD) & 7) == 7) {   // at most 32 products of < 2^56 between Barrett folds
#pragma unroll
      for (int k = 0; k < 8; k++) {
        acc0[k] = reduce64(acc0[k], m);
        acc1[k] = reduce64(acc1[k], m);
      }
    }
    __syncthreads();   // the LDS buffers are reused by the next pass
  }
  // to_ntt of the ciphertext's second row (poly.rs:613-623: reduced mod q first), added to output row 1
  u32 e1[8];
#pragma unroll
  for (int k = 0; k < 8; k++) e1[k] = reduce64(src[(size_t)N + tau + 256 * k], m);
  ntt_fwd_block(e1, tau, lds0, lds1, fw, fw + N, m.q, m.two_q);
  u32* o0 = d.v + ((size_t)d.out_idx[b] * 2 + c) * N + 8 * tau;
  u32* o1 = o0 + (size_t)2 * N;
  const uint4 g0 = reinterpret_cast<const uint4*>(o0)[0], g1 = reinterpret_cast<const uint4*>(o0)[1];
  const uint4 h0 = reinterpret_cast<const uint4*>(o1)[0], h1 = reinterpret_cast<const uint4*>(o1)[1];
  const u32 ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const u32 ha[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
  u32 w0[8], w1[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    w0[k] = reduce64(acc0[k] + ga[k], m);
    w1[k] = reduce64(acc1[k] + (u64)ha[k] + e1[k], m);
  }
  reinterpret_cast<uint4*>(o0)[0] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
  reinterpret_cast<uint4*>(o0)[1] = make_uint4(w0[4], w0[5], w0[6], w0[7]);
  reinterpret_cast<uint4*>(o1)[0] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
  reinterpret_cast<uint4*>(o1)[1] = make_uint4(w1[4], w1[5], w1[6], w1[7]);
}
__global__ __launch_bounds__(256) void k_expand_round(DevTables T, ExpandSideDesc dl, ExpandSideDesc dr) {
  __shared__ u32 lds0[EXPAND_GD * LDS_WORDS];
  __shared__ u32 lds1[EXPAND_GD * LDS_WORDS];
  const bool right = (int)blockIdx.x >= dl.cnt;
  expand_round_body(T, right ? dr : dl, (int)blockIdx.x - (right ? dl.cnt : 0), blockIdx.y, lds0, lds1);
}
// grouped form (kernels.hpp, GroupOff): grid.z = query.  The RIGHT side's workgroups come first: a right-hand ciphertext has 56
// one-bit digits (15 transform passes) against 8 (3) on the left, so the long workgroups start first and the short ones fill in
__global__ __launch_bounds__(256) void k_expand_round_group(DevTables T, ExpandSideDesc dl, ExpandSideDesc dr, GroupOff g) {
  __shared__ u32 lds0[EXPAND_GD * LDS_WORDS];
  __shared__ u32 lds1[EXPAND_GD * LDS_WORDS];
  const int qi = blockIdx.z;
  const bool right = (int)blockIdx.x < dr.cnt;
  ExpandSideDesc d = right ? dr : dl;
  d.raw = group_rebase(d.raw, g.raw[qi]);
  d.A = group_rebase(d.A, g.pp[qi]);
  d.v = group_rebase(d.v, g.v[qi]);
  expand_round_body(T, d, (int)blockIdx.x - (right ? 0 : dr.cnt), blockIdx.y, lds0, lds1);
}
void launch_expand_round_group(const DevTables& T, const ExpandSideDesc& left, const ExpandSideDesc& right, const GroupOff& g, int B,
                               hipStream_t s) {
  const int total = std::max(left.cnt, 0) + std::max(right.cnt, 0);
  if (total <= 0 || B <= 0) return;
  ExpandSideDesc a = left, b = right;
  a.cnt = std::max(a.cnt, 0);
  b.cnt = std::max(b.cnt, 0);
  hipLaunchKernelGGL(k_expand_round_group, dim3(total, 2, B), dim3(256), 0, s, T, a, b, g);
  launched(PATH_EXPAND_FUSED | PATH_EXPAND_GROUP, "k_expand_round_group");
}
void launch_expand_round(const DevTables& T, const ExpandSideDesc& left, const ExpandSideDesc& right, hipStream_t s) {
  const int total = std::max(left.cnt, 0) + std::max(right.cnt, 0);
  if (total <= 0) return;
  ExpandSideDesc a = left, b = right;
  a.cnt = std::max(a.cnt, 0);
  b.cnt = std::max(b.cnt, 0);
  hipLaunchKernelGGL(k_expand_round, dim3(total, 2), dim3(256), 0, s, T, a, b);
  launched(PATH_EXPAND_FUSED, "k_expand_round");
}

// ------------------------------------------------------------------------------------------------
// inverse NTT (both moduli) + Garner CRT -> raw u64.  grid (n_polys)
// The composed value is the unique v in [0, Q) with v = x mod q0, v = y mod q1, i.e. exactly
// (x*q1*(q1^-1 mod q0) + y*q0*(q0^-1 mod q1)) mod Q of params.rs:207-214.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ntt_inv(DevTables T, InvDesc d) {
  __shared__ u32 ldsA[LDS_WORDS];
  __shared__ u32 ldsB[LDS_WORDS];
  ntt_inv_body(T, d, blockIdx.x, ldsA, ldsB);
}
// grouped form (kernels.hpp, GroupOff): grid.y = query
__global__ __launch_bounds__(256) void k_ntt_inv_group(DevTables T, InvDesc d, GroupOff g) {
  __shared__ u32 ldsA[LDS_WORDS];
  __shared__ u32 ldsB[LDS_WORDS];
  const int qi = blockIdx.y;
  d.src = group_rebase(d.src, g.v[qi]);
  d.scal_dst = group_rebase(d.scal_dst, g.v[qi]);
  d.dst = group_rebase(d.dst, g.raw[qi]);
  ntt_inv_body(T, d, blockIdx.x, ldsA, ldsB);
}
void launch_ntt_inv_group(const DevTables& T, const InvDesc& d, const GroupOff& g, int B, hipStream_t s) {
  const int blocks = d.n_polys + (d.scal ? 2 * d.n_scalar_only : 0);
  if (blocks <= 0 || B <= 0) return;
  hipLaunchKernelGGL(k_ntt_inv_group, dim3(blocks, B), dim3(256), 0, s, T, d, g);
  launched(PATH_EXPAND_GROUP, "k_ntt_inv_group");
}
void launch_ntt_inv(const DevTables& T, const InvDesc& d, hipStream_t s) {
  const int blocks = d.n_polys + (d.scal ? 2 * d.n_scalar_only : 0);
  if (blocks <= 0) return;
  hipLaunchKernelGGL(k_ntt_inv, dim3(blocks), dim3(256), 0, s, T, d);
  launched(d.sweep_np > 0 ? PATH_FROM_SWEEP1 : 0, "k_ntt_inv");
}

// ------------------------------------------------------------------------------------------------
// from_ntt of four adjacent sweep-output columns per workgroup.  grid (np/4 * 2 * planes)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_from_sweep4(DevTables T, const u32* src, int np, int premod, u64* dst, int xcd_map) {
  __shared__ u32 lds0[4 * LDS_WORDS];
  __shared__ u32 lds1[4 * LDS_WORDS];
  const int tau = threadIdx.x;
  int g = blockIdx.x;                    // (plane, r, ii/4)
  // The 8 column groups that share one 128-byte line of the [z][ii] source should go through ONE XCD's L2, back
  // to back (workgroups are dealt to the 8 XCDs round-robin): block b -> XCD b % 8 handles line (b/64)*8 + b%8,
  // group (b/8) % 8 of that line.  Otherwise every line is fetched from HBM by up to 8 L2s.
  if (xcd_map & 1) {
    const int xcd = g & 7, t = g >> 3;
    g = ((t >> 3) * 8 + xcd) * 8 + (t & 7);
  }
  const int gpr = np / 4;  // groups per (plane, r)
  const int groups_per_plane = gpr * 2;
  const int plane = g / groups_per_plane, rem = g % groups_per_plane;
  const int r = rem / gpr, gl = rem % gpr;
  const int ii0 = gl * 4;
  const size_t base = ((size_t)plane * 4 + r * 2) * N * np + ii0;
  u32 res0[4][8];
  // both moduli's operands are requested up front (16 strided 16-byte loads per thread: the loads are 43 % of this kernel's time
  // when each modulus waits for its own, profiles/r05_from_sweep_dissection.md); modulus 1's arrive under modulus 0's transforms
  uint4 xin[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int k = 0; k < 8; k++)
      xin[c][k] = *reinterpret_cast<const uint4*>(src + base + (size_t)c * N * np + (size_t)(8 * tau + k) * np);
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    u32 v[4][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint4 x = xin[c][k];
      if (premod) {
        x.x %= m.q; x.y %= m.q; x.z %= m.q; x.w %= m.q;
      }
      v[0][k] = x.x; v[1][k] = x.y; v[2][k] = x.z; v[3][k] = x.w;
    }
    const u32* iw = inv_tables(T.tw, c);
    if (c == 1) __syncthreads();
    ntt_inv_block_m<4>(v, tau, lds0, lds1, iw, iw + N, m.q, m.two_q);
    if (c == 0) {
#pragma unroll
      for (int mm = 0; mm < 4; mm++)
#pragma unroll
        for (int k = 0; k < 8; k++) res0[mm][k] = v[mm][k];
    } else {
      const u32 q0 = T.c.mod[0].q, q1 = T.c.mod[1].q;
#pragma unroll
      for (int mm = 0; mm < 4; mm++) {
        u64* out = dst + (((size_t)plane * np + ii0 + mm) * 2 + r) * N;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          u32 x = res0[mm][k], y = v[mm][k];
          u32 xm = x >= q1 ? x - q1 : x;
          u32 dd = y >= xm ? y - xm : y + q1 - xm;
          u32 qt = __umulhi(dd, T.c.q0_inv_q1_sh);
          u32 e = dd * T.c.q0_inv_q1 - qt * q1;
          e = e >= q1 ? e - q1 : e;
          const u64 val = (u64)x + (u64)q0 * (u64)e;
          out[tau + 256 * k] = val;
        }
      }
    }
  }
}
void launch_from_sweep4(const DevTables& T, const u32* src, int np, int n_planes, int premod, u64* dst, hipStream_t s) {
  if (n_planes <= 0) return;
  const unsigned groups = (unsigned)((np / 4) * 2 * n_planes);
  // (r06, both measured and removed, profiles/r06_from_sweep8.md: the same loads as non-temporal ones, which bypass the L1 -- 16 of
  // each line's 128 bytes are this workgroup's -- are slower, 340 against 301 us per launch; EIGHT columns per 512-thread workgroup,
  // two sets of threads asking for neighbouring pieces of the same lines in lockstep, is no faster alone -- 278 against 267 us --
  // and, at 147 KiB of LDS, does not fit beside the sweep's resident wave: the pipelined query lost 6 %)
  const int xcd_map = tunable("from_sweep_xcd", 1) != 0 && (np % 32) == 0 && (groups % 64) == 0;
  // (r04: a persistent form of this kernel that pulled each workgroup's next operand into the L2 ahead of its transform was
  // measured on one allocation and changes nothing -- 85.8 vs 85.5 queries/s, 2.08 vs 2.05 ms of un-pipelined from_ntt +
  // fold, 30.4 vs 30.1 ms per 8-query step -- and is gone again: profiles/r04_fold_from_ntt_ab.md)
  hipLaunchKernelGGL(k_from_sweep4, dim3(groups), dim3(256), 0, s, T, src, np, premod, dst, xcd_map);
  launched(PATH_FROM_SWEEP4 | (xcd_map ? PATH_SWEEP_XCD_FROM : 0), "k_from_sweep4");
}

}  // namespace spiral
