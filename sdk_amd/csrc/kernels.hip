// gfx950 (MI355X, CDNA4) kernels for the Spiral PIR answer path.  Integer mod-q arithmetic on the
// vector ALU (u32 mul_lo/mul_hi, v_mad_u64_u32); no MFMA (this is not a floating-point contraction).
// wave = 64 lanes; 256-thread workgroups unless stated.
//
// Reference semantics reproduced here (lib/spiral-rs/src): ntt.rs:67-113, 212-258 (negacyclic NTT,
// Harvey lazy butterflies), poly.rs:351-663, gadget.rs:34-60, util.rs:323-355, server.rs:155-221.
// Residues leave every kernel canonical (< q), so all intermediates equal the scalar reference's.
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
// Gentleman-Sande butterfly with the 1/2 folded in (ntt.rs:236-249): x,y in [0,2q) -> [0,2q)
__device__ __forceinline__ void gs_bfly(u32& x, u32& y, u32 w, u32 wp, u32 q, u32 q2) {
  u32 tt = q2 - y + x;
  u32 s = x + y;
  u32 cx = s - (s >= q2 ? q2 : 0u);
  u32 ht = __umulhi(tt, wp);
  x = (cx + ((tt & 1u) ? q : 0u)) >> 1;
  y = w * tt - ht * q;
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
#pragma unroll
    for (int k = 0; k < 4; k++) gs_bfly(v[k], v[k + 4], w, wp, q, q2);
  }
}

// Forward 2048-point negacyclic NTT of the 8 values per thread held in pattern {tau + 256k}
// (natural order in), leaving the result in pattern {8 tau + k} (reference output order).
__device__ __forceinline__ void ntt_fwd_block(u32 (&v)[8], int tau, u32* ldsA, u32* ldsB, const u32* __restrict__ fw,
                                              const u32* __restrict__ fwp, u32 q, u32 q2) {
  fwd_pass<256, true>(v, 0, fw, fwp, q, q2);
#pragma unroll
  for (int k = 0; k < 8; k++) ldsA[PAD_A(tau + 256 * k)] = v[k];
  __syncthreads();
  {
    int b = tau >> 5, o = tau & 31;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = ldsA[PAD_A(256 * b + o + 32 * k)];
    fwd_pass<32, true>(v, b, fw, fwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) ldsB[PAD_A(256 * b + o + 32 * k)] = v[k];
  }
  __syncthreads();
  {
    int b = tau >> 2, o = tau & 3;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = ldsB[PAD_A(32 * b + o + 4 * k)];
    fwd_pass<4, true>(v, b, fw, fwp, q, q2);
#pragma unroll
    for (int k = 0; k < 8; k++) ldsA[PAD_B(32 * b + o + 4 * k)] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = ldsA[PAD_B(8 * tau + k)];
  fwd_pass<1, false>(v, tau, fw, fwp, q, q2);
#pragma unroll
  for (int k = 0; k < 8; k++) {  // ntt.rs:107-111
    u32 x = v[k];
    x -= (x >= q2 ? q2 : 0u);
    x -= (x >= q ? q : 0u);
    v[k] = x;
  }
}

// Inverse: values in pattern {8 tau + k} (< 2q) -> pattern {tau + 256k}, canonical.
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
  for (int k = 0; k < 8; k++) {  // ntt.rs:253-256
    u32 x = v[k];
    x -= (x >= q2 ? q2 : 0u);
    x -= (x >= q ? q : 0u);
    v[k] = x;
  }
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
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int k = 0; k < 4; k++) gs_bfly(v[m][k], v[m][k + 4], w, wp, q, q2);
  }
}
template <int M>
__device__ __forceinline__ void ntt_fwd_block_m(u32 (&v)[M][8], int tau, u32* la, u32* lb,
                                                const u32* __restrict__ fw, const u32* __restrict__ fwp, u32 q, u32 q2) {
  fwd_pass_m<256, true, M>(v, 0, fw, fwp, q, q2);
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
    fwd_pass_m<32, true, M>(v, b, fw, fwp, q, q2);
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
    fwd_pass_m<4, true, M>(v, b, fw, fwp, q, q2);
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
  fwd_pass_m<1, false, M>(v, tau, fw, fwp, q, q2);
#pragma unroll
  for (int m = 0; m < M; m++)
#pragma unroll
    for (int k = 0; k < 8; k++) {
      u32 x = v[m][k];
      x -= (x >= q2 ? q2 : 0u);
      x -= (x >= q ? q : 0u);
      v[m][k] = x;
    }
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
    for (int k = 0; k < 8; k++) {
      u32 x = v[m][k];
      x -= (x >= q2 ? q2 : 0u);
      x -= (x >= q ? q : 0u);
      v[m][k] = x;
    }
}

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
void launch_ntt_fwd(const DevTables& T, const FwdDesc& d, hipStream_t s) {
  if (d.n_out <= 0) return;
  hipLaunchKernelGGL(k_ntt_fwd, dim3(d.n_out, 2), dim3(256), 0, s, T, d);
}
void launch_ntt_fwd3(const DevTables& T, const FwdDesc& d0, const FwdDesc& d1, const FwdDesc& d2, hipStream_t s) {
  const int total = std::max(d0.n_out, 0) + std::max(d1.n_out, 0) + std::max(d2.n_out, 0);
  if (total <= 0) return;
  FwdDesc a = d0, b = d1, c = d2;
  a.n_out = std::max(a.n_out, 0);
  b.n_out = std::max(b.n_out, 0);
  c.n_out = std::max(c.n_out, 0);
  hipLaunchKernelGGL(k_ntt_fwd3, dim3(total, 2), dim3(256), 0, s, T, a, b, c);
}

// ------------------------------------------------------------------------------------------------
// inverse NTT (both moduli) + Garner CRT -> raw u64.  grid (n_polys)
// The composed value is the unique v in [0, Q) with v = x mod q0, v = y mod q1, i.e. exactly
// (x*q1*(q1^-1 mod q0) + y*q0*(q0^-1 mod q1)) mod Q of params.rs:207-214.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ntt_inv(DevTables T, InvDesc d) {
  __shared__ u32 ldsA[LDS_WORDS];
  __shared__ u32 ldsB[LDS_WORDS];
  const int tau = threadIdx.x;
  int p = blockIdx.x;
  long base;
  long crt_stride = d.crt_stride, z_stride = d.z_stride;
  if (d.sweep_np > 0) {
    const long np = d.sweep_np;
    // The 16 columns ii that share a 64-byte line of the [z][ii] source should be read through ONE XCD's
    // L2 (blocks are dealt to XCDs round-robin): block b -> XCD b % 8 handles group (b/8/16)*8 + b%8.
    if ((np % 16) == 0 && ((long)d.n_polys % 128) == 0) {
      const int b = blockIdx.x;
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
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int z = 8 * tau + k;
          d.scal_dst[dp + c * N + z] = reduce64((u64)d.src[sp + c * N + z] * (u64)d.scal[c * N + z], T.c.mod[c]);
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
    const u32* iw = T.tw + ((size_t)c * 4 + 2) * N;
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
void launch_ntt_inv(const DevTables& T, const InvDesc& d, hipStream_t s) {
  const int blocks = d.n_polys + (d.scal ? 2 * d.n_scalar_only : 0);
  if (blocks <= 0) return;
  hipLaunchKernelGGL(k_ntt_inv, dim3(blocks), dim3(256), 0, s, T, d);
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
  if (xcd_map) {
    const int xcd = g & 7, t = g >> 3;
    g = ((t >> 3) * 8 + xcd) * 8 + (t & 7);
  }
  const int groups_per_plane = (np / 4) * 2;
  const int plane = g / groups_per_plane, rem = g % groups_per_plane;
  const int r = rem / (np / 4), ii0 = (rem % (np / 4)) * 4;
  const size_t base = ((size_t)plane * 4 + r * 2) * N * np + ii0;
  u32 res0[4][8];
#pragma unroll 1
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    const u32* sp = src + base + (size_t)c * N * np;
    u32 v[4][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint4 x = *reinterpret_cast<const uint4*>(sp + (size_t)(8 * tau + k) * np);
      if (premod) {
        x.x %= m.q; x.y %= m.q; x.z %= m.q; x.w %= m.q;
      }
      v[0][k] = x.x; v[1][k] = x.y; v[2][k] = x.z; v[3][k] = x.w;
    }
    const u32* iw = T.tw + ((size_t)c * 4 + 2) * N;
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
          out[tau + 256 * k] = (u64)x + (u64)q0 * (u64)e;
        }
      }
    }
  }
}
void launch_from_sweep4(const DevTables& T, const u32* src, int np, int n_planes, int premod, u64* dst, hipStream_t s) {
  if (n_planes <= 0) return;
  const unsigned groups = (unsigned)((np / 4) * 2 * n_planes);
  static const int want = [] { const char* e = getenv("SPIRAL_FROM_SWEEP_XCD"); return e ? atoi(e) : 1; }();
  const int xcd_map = want && (np % 32) == 0 && (groups % 64) == 0;
  hipLaunchKernelGGL(k_from_sweep4, dim3(groups), dim3(256), 0, s, T, src, np, premod, dst, xcd_map);
}

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
template <bool HOIST_TW>
__global__ __launch_bounds__(256, 2) void k_fold_fused(DevTables T, FoldDesc d) {
  __shared__ u32 lds0[LDS_WORDS];
  __shared__ u32 lds1[LDS_WORDS];
  const int tau = threadIdx.x;
  const int i = blockIdx.x, plane = blockIdx.y;
  const int two_t = 2 * d.t, four_t = 4 * d.t;
  const u64* ct0 = d.X + ((size_t)plane * d.cur + i) * 2 * N;
  const u64* ct1 = ct0 + (size_t)d.half * 2 * N;
  const u64 mask = (1ULL << d.bits) - 1ULL;
  u32* la = lds0;
  u32* lb = lds1;
  u64* out = d.Y + ((size_t)plane * d.half + i) * 2 * N;
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
      for (int kd = 0; kd < d.t; kd++) {
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
        if (!HOIST_TW) {  // keep twiddle loads and LDS address arithmetic inside the loop (fewer live VGPRs)
          asm volatile("" : "+s"(fwk));
          asm volatile("" : "+v"(tk));
        }
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
    const u32* iw = T.tw + ((size_t)c * 4 + 2) * N;
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
// TW_LDS: the forward twiddles + Shoup quotients of the current modulus (16 KiB) are staged in LDS once per
// modulus, so the 4 x 14 twiddle reads of every digit transform are ds_reads (no vector-memory latency, and no
// queueing behind a concurrent sweep's load stream when the fold runs on the second stream).
template <bool TW_LDS>
__global__ __launch_bounds__(256, 2) void k_fold_fused2(DevTables T, FoldDesc d) {
  __shared__ u32 lds0[2 * LDS_WORDS];
  __shared__ u32 lds1[2 * LDS_WORDS];
  __shared__ u32 ltw[TW_LDS ? 2 * N : 4];
  const int tau = threadIdx.x;
  const int i = blockIdx.x, plane = blockIdx.y;
  const int two_t = 2 * d.t, four_t = 4 * d.t;
  const u64* ct0 = d.X + ((size_t)plane * d.cur + i) * 2 * N;
  const u64* ct1 = ct0 + (size_t)d.half * 2 * N;
  const u64 mask = (1ULL << d.bits) - 1ULL;
  u32* la = lds0;
  u32* lb = lds1;
  u64* out = d.Y + ((size_t)plane * d.half + i) * 2 * N;
#pragma unroll 1
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    const u32* fw = T.tw + (size_t)c * 4 * N;
    if (TW_LDS) {
      if (c == 1) __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; k++)  // [fw | fwp] = 2N words
        reinterpret_cast<uint4*>(ltw)[tau + 256 * k] = reinterpret_cast<const uint4*>(fw)[tau + 256 * k];
      __syncthreads();
    }
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
        const u32* fwk = fw;
        int tk = tau;
        asm volatile("" : "+s"(fwk));
        asm volatile("" : "+v"(tk));
        if (TW_LDS)
          ntt_fwd_block_m<2>(v, tk, la, lb, ltw, ltw + N, m.q, m.two_q);
        else
          ntt_fwd_block_m<2>(v, tk, la, lb, fwk, fwk + N, m.q, m.two_q);
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
    const u32* iw = T.tw + ((size_t)c * 4 + 2) * N;
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
void launch_fold_fused(const DevTables& T, const FoldDesc& d, hipStream_t s) {
  if (d.half <= 0 || d.planes <= 0) return;
  static const int variant = [] {
    const char* e = getenv("SPIRAL_FOLD_VARIANT");
    return e ? atoi(e) : 3;
  }();
  if (variant == 3 && (d.t % 2) == 0)
    hipLaunchKernelGGL(k_fold_fused2<true>, dim3(d.half, d.planes), dim3(256), 0, s, T, d);
  else if (variant == 2 && (d.t % 2) == 0)
    hipLaunchKernelGGL(k_fold_fused2<false>, dim3(d.half, d.planes), dim3(256), 0, s, T, d);
  else if (variant == 1)
    hipLaunchKernelGGL(k_fold_fused<true>, dim3(d.half, d.planes), dim3(256), 0, s, T, d);
  else
    hipLaunchKernelGGL(k_fold_fused<false>, dim3(d.half, d.planes), dim3(256), 0, s, T, d);
}

// ------------------------------------------------------------------------------------------------
// NTT-domain multiply-accumulate.  grid (2N/256, batch), one (crt, z) per thread.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mac_body(const DevTables& T, const MacDesc& d, int inner, int outer) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // index into [crt][z]
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
      for (; k + 28 <= k_hi; k += 28) {  // 56 loads in flight: the small expansion rounds are pure latency
        u32 a[28], bb[28];
#pragma unroll
        for (int u = 0; u < 28; u++) {
          a[u] = A[(size_t)(k + u) * PW];
          bb[u] = Bs[(size_t)(k + u) * PW];
        }
#pragma unroll
        for (int u = 0; u < 28; u++) acc += (u64)a[u] * (u64)bb[u];
        since += 28;
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
__global__ __launch_bounds__(256) void k_mac(DevTables T, MacDesc d) { mac_body(T, d, blockIdx.y, blockIdx.z); }
__global__ __launch_bounds__(256) void k_mac2(DevTables T, MacDesc d0, MacDesc d1) {
  const int y = blockIdx.y;
  if (y < d0.batch_inner)
    mac_body(T, d0, y, 0);
  else
    mac_body(T, d1, y - d0.batch_inner, 0);
}
void launch_mac(const DevTables& T, const MacDesc& d, hipStream_t s) {
  if (d.batch_inner <= 0 || d.batch_outer <= 0) return;
  hipLaunchKernelGGL(k_mac, dim3(2 * N / 256, d.batch_inner, d.batch_outer), dim3(256), 0, s, T, d);
}
void launch_mac2(const DevTables& T, const MacDesc& d0, const MacDesc& d1, hipStream_t s) {
  MacDesc a = d0, b = d1;
  a.batch_inner = std::max(a.batch_inner, 0);
  b.batch_inner = std::max(b.batch_inner, 0);
  if (a.batch_inner + b.batch_inner <= 0) return;
  hipLaunchKernelGGL(k_mac2, dim3(2 * N / 256, a.batch_inner + b.batch_inner, 1), dim3(256), 0, s, T, a, b);
}

__global__ __launch_bounds__(256) void k_add_poly_into(DevTables T, u32* dst, const int* idx, const u32* src) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const int b = blockIdx.y;
  const long dp = (long)idx[b] * 2 * N + e;
  dst[dp] = add_mod(dst[dp], src[(size_t)b * 2 * N + e], T.c.mod[c].q);
}
void launch_add_poly_into(const DevTables& T, u32* dst, const int* idx, const u32* src, int batch, hipStream_t s) {
  if (batch <= 0) return;
  hipLaunchKernelGGL(k_add_poly_into, dim3(2 * N / 256, batch), dim3(256), 0, s, T, dst, idx, src);
}

__global__ __launch_bounds__(256) void k_scalar_mul(DevTables T, u32* base, long dst_off, long src_off,
                                                    const u32* scalar) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const long b = blockIdx.y;
  const ModConst m = T.c.mod[c];
  base[(dst_off + b) * 2 * N + e] = reduce64((u64)base[(src_off + b) * 2 * N + e] * (u64)scalar[e], m);
}
void launch_scalar_mul(const DevTables& T, u32* base, long dst_off, long src_off, const u32* scalar, int n_polys,
                       hipStream_t s) {
  if (n_polys <= 0) return;
  hipLaunchKernelGGL(k_scalar_mul, dim3(2 * N / 256, n_polys), dim3(256), 0, s, T, base, dst_off, src_off, scalar);
}

__global__ __launch_bounds__(256) void k_add_polys_idx(DevTables T, u32* dst, const int* di, const u32* a, const int* ai,
                                                        const u32* b, const int* bi) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const int k = blockIdx.y;
  dst[(size_t)di[k] * 2 * N + e] = add_mod(a[(size_t)ai[k] * 2 * N + e], b[(size_t)bi[k] * 2 * N + e], T.c.mod[c].q);
}
void launch_add_polys_idx(const DevTables& T, u32* dst, const int* di, const u32* a, const int* ai, const u32* b,
                          const int* bi, int count, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_add_polys_idx, dim3(2 * N / 256, count), dim3(256), 0, s, T, dst, di, a, ai, b, bi);
}

__global__ __launch_bounds__(256) void k_interleave_query(u64* qv, const u32* row0_ntt, const u64* wire, int dim0) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // index over [z][j]
  if (i >= (size_t)N * dim0) return;
  const int j = (int)(i % dim0);
  const int z = (int)(i / dim0);
  const u32* p = row0_ntt + (size_t)j * 2 * N;
  qv[2 * i] = (u64)p[z] | ((u64)p[N + z] << 32);
  qv[2 * i + 1] = wire[i];
}
void launch_interleave_query(u64* qv, const u32* row0_ntt, const u64* wire, int dim0, hipStream_t s) {
  size_t total = (size_t)N * dim0;
  hipLaunchKernelGGL(k_interleave_query, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qv, row0_ntt, wire, dim0);
}

__global__ __launch_bounds__(256) void k_copy_polys(u32* dst, const int* dst_idx, int dst_row_stride, const u32* src,
                                                    const int* src_idx, int src_row_stride, int R) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y / R, r = blockIdx.y % R;
  dst[((long)dst_idx[b] + (long)r * dst_row_stride) * 2 * N + e] =
      src[((long)src_idx[b] + (long)r * src_row_stride) * 2 * N + e];
}
void launch_copy_polys(u32* dst, const int* dst_idx, int dst_row_stride, const u32* src, const int* src_idx,
                       int src_row_stride, int R, int batch, hipStream_t s) {
  if (batch <= 0) return;
  hipLaunchKernelGGL(k_copy_polys, dim3(2 * N / 256, batch * R), dim3(256), 0, s, dst, dst_idx, dst_row_stride, src,
                     src_idx, src_row_stride, R);
}

__global__ __launch_bounds__(256) void k_folding_neg(DevTables T, u32* mats, const u32* gadget_ntt, int two_t) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const int col = blockIdx.y % two_t, r = blockIdx.y / two_t;
  const int dd = blockIdx.z;
  const u32 q = T.c.mod[c].q;
  u32* row = mats + ((size_t)(dd * 2 + r) * 2 * two_t) * 2 * N;
  const u32 cv = row[(size_t)(two_t + col) * 2 * N + e];
  const u32 g = gadget_ntt[((size_t)r * two_t + col) * 2 * N + e];
  row[(size_t)col * 2 * N + e] = add_mod(g, cv ? q - cv : 0u, q);
}
void launch_folding_neg(const DevTables& T, u32* mats, const u32* gadget_ntt, int nu2, int two_t, hipStream_t s) {
  if (nu2 <= 0) return;
  hipLaunchKernelGGL(k_folding_neg, dim3(2 * N / 256, 2 * two_t, nu2), dim3(256), 0, s, T, mats, gadget_ntt, two_t);
}

__global__ __launch_bounds__(256) void k_add(DevTables T, u32* out, const u32* a, const u32* b) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)((i >> POLY_LEN_LOG2) & 1);
  out[i] = add_mod(a[i], b[i], T.c.mod[c].q);
}
void launch_add(const DevTables& T, u32* out, const u32* a, const u32* b, int n_polys, hipStream_t s) {
  if (n_polys <= 0) return;
  hipLaunchKernelGGL(k_add, dim3((unsigned)((size_t)n_polys * 2 * N / 256)), dim3(256), 0, s, T, out, a, b);
}

__global__ __launch_bounds__(256) void k_invert_raw(u64 Q, u64* out, const u64* a, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = Q - a[i];
}
void launch_invert_raw(const DevTables& T, u64* out, const u64* a, long n_words, hipStream_t s) {
  if (n_words <= 0) return;
  hipLaunchKernelGGL(k_invert_raw, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, T.c.Q, out, a, n_words);
}

__global__ __launch_bounds__(256) void k_automorph(u64 Q, u64* out, const u64* a, int t) {
  const int z = blockIdx.x * 256 + threadIdx.x;
  const size_t p = (size_t)blockIdx.y * N;
  unsigned zt = (unsigned)z * (unsigned)t;
  unsigned num = zt >> POLY_LEN_LOG2, rem = zt & (N - 1);
  u64 v = a[p + z];
  out[p + rem] = (num & 1u) ? Q - v : v;
}
void launch_automorph(const DevTables& T, u64* out, const u64* a, int n_polys, int t, hipStream_t s) {
  if (n_polys <= 0) return;
  hipLaunchKernelGGL(k_automorph, dim3(N / 256, n_polys), dim3(256), 0, s, T.c.Q, out, a, t);
}

__global__ __launch_bounds__(256) void k_gadget_raw(u64* out, const u64* inp, int cols, int rdim, int bits) {
  const int z = blockIdx.x * 256 + threadIdx.x;
  const int o = blockIdx.y;  // output poly index: row*cols + col
  const int row = o / cols, col = o - row * cols;
  const int k = row / rdim, j = row - k * rdim;
  const int sh = k * bits;
  const u64 mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL);
  u64 x = inp[((size_t)j * cols + col) * N + z];
  out[(size_t)o * N + z] = sh >= 64 ? 0ULL : ((x >> sh) & mask);
}
void launch_gadget_raw(u64* out, const u64* inp, int rows_in, int cols, int rows_out, int rdim, int bits,
                       hipStream_t s) {
  (void)rows_in;
  hipLaunchKernelGGL(k_gadget_raw, dim3(N / 256, rows_out * cols), dim3(256), 0, s, out, inp, cols, rdim, bits);
}

__global__ __launch_bounds__(256) void k_u64_to_u32(u32* out, const u64* in, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (u32)in[i];
}
__global__ __launch_bounds__(256) void k_u32_to_u64(u64* out, const u32* in, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (u64)in[i];
}
void launch_u64_to_u32(u32* out, const u64* in, long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_u64_to_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, in, n);
}
void launch_u32_to_u64(u64* out, const u32* in, long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_u32_to_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, in, n);
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
__global__ __launch_bounds__(256) void k_encode(EncodeDesc d) {
  const int per_inst_first = d.n * N, per_inst_rest = d.n * d.n * N;
  const int per_inst = per_inst_first + per_inst_rest;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
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
void launch_encode(const EncodeDesc& d, hipStream_t s) {
  const long total = (long)d.instances * ((long)d.n * N + (long)d.n * d.n * N);
  hipLaunchKernelGGL(k_encode, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
}

// out[z][j][r] = v[ct_j][r][0][z] | v[ct_j][r][1][z] << 32   (util.rs:343-350; residues already < q)
__global__ __launch_bounds__(256) void k_reorient(u64* out, const u32* v, int first, int step, int dim0) {
  __shared__ u64 tile[32][33];
  // tile over (z, j) for a fixed r: blockIdx.x -> j tile, blockIdx.y -> z tile, blockIdx.z -> r
  const int r = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int j0 = blockIdx.x * 32, z0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int j = j0 + ty + 8 * i;
    if (j < dim0) {
      const u32* p = v + ((size_t)(first + step * j) * 2 + r) * 2 * N;
      tile[ty + 8 * i][tx] = (u64)p[z0 + tx] | ((u64)p[N + z0 + tx] << 32);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int z = z0 + ty + 8 * i;
    int j = j0 + tx;
    if (j < dim0) out[((size_t)z * dim0 + j) * 2 + r] = tile[tx][ty + 8 * i];
  }
}
void launch_reorient(u64* out, const u32* v, int first, int step, int dim0, hipStream_t s) {
  hipLaunchKernelGGL(k_reorient, dim3((dim0 + 31) / 32, N / 32, 2), dim3(256), 0, s, out, v, first, step, dim0);
}

// ------------------------------------------------------------------------------------------------
// database sweep  (server.rs:155-221)
// ------------------------------------------------------------------------------------------------
// WIDE (num_per >= 128): one wave per (plane, z, 128-wide ii chunk).  Lane l owns output columns
// ii = chunk*128 + 2l, 2l+1 and streams its 16 bytes of every first-dimension row j: 1 KiB
// contiguous per wave per row.  The query words for (z, j) are wave-uniform (scalar loads, SGPR
// operands of v_mad_u64_u32); no cross-lane traffic at all.  Products are < 2^56, so 256 rows are
// accumulated in u64 between Barrett folds (the reference uses u128 and one % at the end; the
// residues are identical).
// flat index of output (plane, rc = r*2+crt, z, ii) in the (possibly column-interleaved) partial buffer
__device__ __forceinline__ size_t sweep_out_index(const SweepDesc& d, int plane, int rc, int z, int ii) {
  const int G = d.out_G > 1 ? d.out_G : 1;
  const int npl = d.num_per / G;
  const size_t chunk_words = (size_t)d.planes * 4 * N * npl;
  return (size_t)(ii % G) * chunk_words + (((size_t)plane * 4 + rc) * N + z) * npl + (ii / G);
}
__device__ __forceinline__ void sweep_store_pair(const SweepDesc& d, int plane, int z, int ii0, u32 r0c0_a, u32 r0c0_b,
                                                 u32 r0c1_a, u32 r0c1_b, u32 r1c0_a, u32 r1c0_b, u32 r1c1_a,
                                                 u32 r1c1_b) {
  if (d.out_G <= 1) {
    const size_t rc = (size_t)N * d.num_per;
    u32* o = d.out + (size_t)plane * 4 * rc + (size_t)z * d.num_per + ii0;
    *reinterpret_cast<uint2*>(o + 0 * rc) = make_uint2(r0c0_a, r0c0_b);
    *reinterpret_cast<uint2*>(o + 1 * rc) = make_uint2(r0c1_a, r0c1_b);
    *reinterpret_cast<uint2*>(o + 2 * rc) = make_uint2(r1c0_a, r1c0_b);
    *reinterpret_cast<uint2*>(o + 3 * rc) = make_uint2(r1c1_a, r1c1_b);
  } else {
    d.out[sweep_out_index(d, plane, 0, z, ii0)] = r0c0_a;
    d.out[sweep_out_index(d, plane, 0, z, ii0 + 1)] = r0c0_b;
    d.out[sweep_out_index(d, plane, 1, z, ii0)] = r0c1_a;
    d.out[sweep_out_index(d, plane, 1, z, ii0 + 1)] = r0c1_b;
    d.out[sweep_out_index(d, plane, 2, z, ii0)] = r1c0_a;
    d.out[sweep_out_index(d, plane, 2, z, ii0 + 1)] = r1c0_b;
    d.out[sweep_out_index(d, plane, 3, z, ii0)] = r1c1_a;
    d.out[sweep_out_index(d, plane, 3, z, ii0 + 1)] = r1c1_b;
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_sweep_wide(DevTables T, SweepDesc d) {
  const int lane = threadIdx.x & 63;
  const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int chunks = d.num_per >> 7;
  const int chunk = unit % chunks;
  const int zp = unit / chunks;  // plane * N + z
  const int z = zp & (N - 1);
  const int plane = zp >> POLY_LEN_LOG2;
  if (plane >= d.planes) return;
  const ulonglong2* p =
      reinterpret_cast<const ulonglong2*>(d.db + ((size_t)zp * d.nj) * d.num_per + (size_t)chunk * 128) + lane;
  const size_t stride = (size_t)(d.num_per >> 1);
  const uint4* __restrict__ qrow = reinterpret_cast<const uint4*>(d.qv) + ((size_t)z * d.dim0 + d.j0);
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];

  u64 a00 = 0, a01 = 0, a02 = 0, a03 = 0;  // word 0: n0_0, n0_1, n1_0, n1_1
  u64 a10 = 0, a11 = 0, a12 = 0, a13 = 0;  // word 1
  for (int jb = 0; jb < d.nj; jb += 256) {
    const int je = min(jb + 256, d.nj);
    int j = jb;
    for (; j + U <= je; j += U) {
      ulonglong2 w[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const ulonglong2* a = p + (size_t)(j + u) * stride;
        if (NT) {  // streamed once: do not keep the lines in L2 / MALL
          w[u].x = __builtin_nontemporal_load(&a->x);
          w[u].y = __builtin_nontemporal_load(&a->y);
        } else {
          w[u] = *a;
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint4 qa = qrow[j + u];  // (a0_lo, a0_hi, a1_lo, a1_hi)
        const u32 b0l = (u32)w[u].x, b0h = (u32)(w[u].x >> 32);
        const u32 b1l = (u32)w[u].y, b1h = (u32)(w[u].y >> 32);
        a00 += (u64)qa.x * b0l;
        a01 += (u64)qa.z * b0l;
        a02 += (u64)qa.y * b0h;
        a03 += (u64)qa.w * b0h;
        a10 += (u64)qa.x * b1l;
        a11 += (u64)qa.z * b1l;
        a12 += (u64)qa.y * b1h;
        a13 += (u64)qa.w * b1h;
      }
    }
    for (; j < je; j++) {
      const ulonglong2 w = p[(size_t)j * stride];
      const uint4 qa = qrow[j];
      const u32 b0l = (u32)w.x, b0h = (u32)(w.x >> 32);
      const u32 b1l = (u32)w.y, b1h = (u32)(w.y >> 32);
      a00 += (u64)qa.x * b0l;
      a01 += (u64)qa.z * b0l;
      a02 += (u64)qa.y * b0h;
      a03 += (u64)qa.w * b0h;
      a10 += (u64)qa.x * b1l;
      a11 += (u64)qa.z * b1l;
      a12 += (u64)qa.y * b1h;
      a13 += (u64)qa.w * b1h;
    }
    a00 = reduce64(a00, m0);
    a01 = reduce64(a01, m0);
    a02 = reduce64(a02, m1);
    a03 = reduce64(a03, m1);
    a10 = reduce64(a10, m0);
    a11 = reduce64(a11, m0);
    a12 = reduce64(a12, m1);
    a13 = reduce64(a13, m1);
  }
  // out[plane][r][crt][z][ii]
  // out[plane][r][crt][z][ii]: (r0,c0) = n0_0, (r0,c1) = n1_0, (r1,c0) = n0_1, (r1,c1) = n1_1
  sweep_store_pair(d, plane, z, chunk * 128 + 2 * lane, (u32)a00, (u32)a10, (u32)a02, (u32)a12, (u32)a01, (u32)a11,
                   (u32)a03, (u32)a13);
}

// PACKED wide sweep: as k_sweep_wide, but each lane streams 28 bytes per ROW PAIR (7 dwords = 8 limbs
// of 28 bits) instead of 32; limb extraction is 6 v_alignbit + 7 v_and per 16 multiply-accumulates.
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
typedef u32 u32x3_t __attribute__((ext_vector_type(3), aligned(4)));
__global__ __launch_bounds__(256) void k_sweep_packed(DevTables T, SweepDesc d) {
  const int lane = threadIdx.x & 63;
  const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int chunks = d.num_per >> 7;
  const int chunk = unit % chunks;
  const int zp = unit / chunks;  // plane * N + z
  const int z = zp & (N - 1);
  const int plane = zp >> POLY_LEN_LOG2;
  if (plane >= d.planes) return;
  const int npairs = d.nj >> 1;
  const u32* base = reinterpret_cast<const u32*>(d.db) + ((size_t)zp * npairs * chunks + chunk) * 448;  // 1792 B = 448 dwords
  const size_t ustride = (size_t)chunks * 448;
  const uint4* __restrict__ qrow = reinterpret_cast<const uint4*>(d.qv) + ((size_t)z * d.dim0 + d.j0);
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  const u32 M = 0x0FFFFFFFu;
  u64 a00 = 0, a01 = 0, a02 = 0, a03 = 0;  // ii = 2l  : n0_0, n0_1, n1_0, n1_1
  u64 a10 = 0, a11 = 0, a12 = 0, a13 = 0;  // ii = 2l+1
  for (int jb = 0; jb < npairs; jb += 128) {
    const int je = min(jb + 128, npairs);
    for (int jp = jb; jp < je; jp++) {
      const u32* u = base + (size_t)jp * ustride;
      const u32* p4 = u + lane * 4;
      const u32* p3 = u + 256 + lane * 3;
      const u32x4_t va = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p4));
      const u32x3_t vb = __builtin_nontemporal_load(reinterpret_cast<const u32x3_t*>(p3));
      const u32 d0 = va.x, d1 = va.y, d2 = va.z, d3 = va.w, d4 = vb.x, d5 = vb.y, d6 = vb.z;
      const uint4 qa = qrow[2 * jp];      // row 2jp   : (a0_lo, a0_hi, a1_lo, a1_hi)
      const uint4 qb = qrow[2 * jp + 1];  // row 2jp+1
      const u32 f0 = d0 & M;
      const u32 f1 = __builtin_amdgcn_alignbit(d1, d0, 28) & M;
      const u32 f2 = __builtin_amdgcn_alignbit(d2, d1, 24) & M;
      const u32 f3 = __builtin_amdgcn_alignbit(d3, d2, 20) & M;
      const u32 f4 = __builtin_amdgcn_alignbit(d4, d3, 16) & M;
      const u32 f5 = __builtin_amdgcn_alignbit(d5, d4, 12) & M;
      const u32 f6 = __builtin_amdgcn_alignbit(d6, d5, 8) & M;
      const u32 f7 = d6 >> 4;
      // (row 2jp, ii 2l) = (f0, f1); (2jp, 2l+1) = (f2, f3); (2jp+1, 2l) = (f4, f5); (2jp+1, 2l+1) = (f6, f7)
      a00 += (u64)qa.x * f0; a01 += (u64)qa.z * f0; a02 += (u64)qa.y * f1; a03 += (u64)qa.w * f1;
      a10 += (u64)qa.x * f2; a11 += (u64)qa.z * f2; a12 += (u64)qa.y * f3; a13 += (u64)qa.w * f3;
      a00 += (u64)qb.x * f4; a01 += (u64)qb.z * f4; a02 += (u64)qb.y * f5; a03 += (u64)qb.w * f5;
      a10 += (u64)qb.x * f6; a11 += (u64)qb.z * f6; a12 += (u64)qb.y * f7; a13 += (u64)qb.w * f7;
    }
    a00 = reduce64(a00, m0); a01 = reduce64(a01, m0); a02 = reduce64(a02, m1); a03 = reduce64(a03, m1);
    a10 = reduce64(a10, m0); a11 = reduce64(a11, m0); a12 = reduce64(a12, m1); a13 = reduce64(a13, m1);
  }
  // out[plane][r][crt][z][ii]: (r0,c0) = n0_0, (r0,c1) = n1_0, (r1,c0) = n0_1, (r1,c1) = n1_1
  sweep_store_pair(d, plane, z, chunk * 128 + 2 * lane, (u32)a00, (u32)a10, (u32)a02, (u32)a12, (u32)a01, (u32)a11,
                   (u32)a03, (u32)a13);
}

// Persistent form of k_sweep_packed for the intra-query pipeline: a fixed grid of `wgs_per_cu` workgroups per
// CU walks the (z, chunk) units, U row pairs in flight per lane, so that half of every CU's wave slots, VGPRs and
// LDS stay free for the fold kernels running concurrently on the second stream.
template <int U>
__global__ __launch_bounds__(256) void k_sweep_packed_persist(DevTables T, SweepDesc d, int units, int hi_prio) {
  // the sweep is a latency-bound load stream using ~20 % of the VALU slots: when fold kernels share the CU its
  // waves must win instruction arbitration or the loads in flight (and the HBM rate) drop
  if (hi_prio) __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int nwaves = gridDim.x * 4;
  const int chunks = d.num_per >> 7;
  const int npairs = d.nj >> 1;
  const size_t ustride = (size_t)chunks * 448;
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  const u32 M = 0x0FFFFFFFu;
  for (int unit = wave0; unit < units; unit += nwaves) {
    const int chunk = unit % chunks;
    const int zp = unit / chunks;
    const int z = zp & (N - 1);
    const int plane = zp >> POLY_LEN_LOG2;
    const u32* base = reinterpret_cast<const u32*>(d.db) + ((size_t)zp * npairs * chunks + chunk) * 448;
    const uint4* __restrict__ qrow = reinterpret_cast<const uint4*>(d.qv) + ((size_t)z * d.dim0 + d.j0);
    u64 a00 = 0, a01 = 0, a02 = 0, a03 = 0, a10 = 0, a11 = 0, a12 = 0, a13 = 0;
    for (int jb = 0; jb < npairs; jb += 128) {
      const int je = min(jb + 128, npairs);
      for (int jp0 = jb; jp0 < je; jp0 += U) {
        u32x4_t va[U];
        u32x3_t vb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int jp = min(jp0 + u, je - 1);
          const u32* uu = base + (size_t)jp * ustride;
          va[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(uu + lane * 4));
          vb[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x3_t*>(uu + 256 + lane * 3));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int jp = jp0 + u;
          if (jp < je) {
            const u32 d0 = va[u].x, d1 = va[u].y, d2 = va[u].z, d3 = va[u].w, d4 = vb[u].x, d5 = vb[u].y, d6 = vb[u].z;
            const uint4 qa = qrow[2 * jp];
            const uint4 qb = qrow[2 * jp + 1];
            const u32 f0 = d0 & M;
            const u32 f1 = __builtin_amdgcn_alignbit(d1, d0, 28) & M;
            const u32 f2 = __builtin_amdgcn_alignbit(d2, d1, 24) & M;
            const u32 f3 = __builtin_amdgcn_alignbit(d3, d2, 20) & M;
            const u32 f4 = __builtin_amdgcn_alignbit(d4, d3, 16) & M;
            const u32 f5 = __builtin_amdgcn_alignbit(d5, d4, 12) & M;
            const u32 f6 = __builtin_amdgcn_alignbit(d6, d5, 8) & M;
            const u32 f7 = d6 >> 4;
            a00 += (u64)qa.x * f0; a01 += (u64)qa.z * f0; a02 += (u64)qa.y * f1; a03 += (u64)qa.w * f1;
            a10 += (u64)qa.x * f2; a11 += (u64)qa.z * f2; a12 += (u64)qa.y * f3; a13 += (u64)qa.w * f3;
            a00 += (u64)qb.x * f4; a01 += (u64)qb.z * f4; a02 += (u64)qb.y * f5; a03 += (u64)qb.w * f5;
            a10 += (u64)qb.x * f6; a11 += (u64)qb.z * f6; a12 += (u64)qb.y * f7; a13 += (u64)qb.w * f7;
          }
        }
      }
      a00 = reduce64(a00, m0); a01 = reduce64(a01, m0); a02 = reduce64(a02, m1); a03 = reduce64(a03, m1);
      a10 = reduce64(a10, m0); a11 = reduce64(a11, m0); a12 = reduce64(a12, m1); a13 = reduce64(a13, m1);
    }
    sweep_store_pair(d, plane, z, chunk * 128 + 2 * lane, (u32)a00, (u32)a10, (u32)a02, (u32)a12, (u32)a01, (u32)a11,
                     (u32)a03, (u32)a13);
  }
}
void launch_sweep_persist(const DevTables& T, const SweepDesc& d, int wgs_per_cu, int unroll, hipStream_t s) {
  const int units = d.planes * N * (d.num_per >> 7);
  const dim3 grid((unsigned)std::min(256 * wgs_per_cu, (units + 3) / 4));
  static const int prio = [] { const char* e = getenv("SPIRAL_SWEEP_PRIO"); return e ? atoi(e) : 1; }();
  switch (unroll) {
    case 1: hipLaunchKernelGGL(k_sweep_packed_persist<1>, grid, dim3(256), 0, s, T, d, units, prio); break;
    case 2: hipLaunchKernelGGL(k_sweep_packed_persist<2>, grid, dim3(256), 0, s, T, d, units, prio); break;
    case 8: hipLaunchKernelGGL(k_sweep_packed_persist<8>, grid, dim3(256), 0, s, T, d, units, prio); break;
    default: hipLaunchKernelGGL(k_sweep_packed_persist<4>, grid, dim3(256), 0, s, T, d, units, prio); break;
  }
}

// Multi-query PACKED sweep: B queries share one pass over the database (BASELINE configs[4]).  Per row
// pair: one 28-byte load, B x 2 scalar query rows, 16 B multiply-accumulates.  HBM-bound up to B ~ 4,
// integer-ALU-bound beyond.
// U row pairs (28 B per lane each) are in flight per wave (U = 4 needs npairs % 4 == 0: no ragged tail, straight-line
// code): with B queries' accumulators the kernel runs at 2-5 waves per SIMD, and one load per wave leaves the HBM
// pipe mostly empty (B = 8: 21 ms per pass instead of ~10).
// QLDS: the B queries' rows for this workgroup's z (B * nj * 16 B, 64 KiB at B = 8) are staged in LDS once and
// read back as broadcast ds_reads.  The scalar-load form keeps them in the 16 KiB scalar cache, which B > 4
// overflows: every query word then costs an L2 round trip and the pass takes 25 ms instead of ~10 at B = 8.
// Needs all four waves of the workgroup on the same z (chunks % 4 == 0).
template <int B, int U, bool QLDS>
__global__ __launch_bounds__(256) void k_sweep_packed_batch(DevTables T, SweepBatchDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
  const int lane = threadIdx.x & 63;
  const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int chunks = d.num_per >> 7;
  const int chunk = unit % chunks;
  const int zp = unit / chunks;
  const int z = zp & (N - 1);
  const int plane = zp >> POLY_LEN_LOG2;
  const uint4* qs = reinterpret_cast<const uint4*>(smem_q);  // [B][nj]
  if (QLDS) {
    const int zp0 = (blockIdx.x * 4) / chunks;  // the whole workgroup shares one (plane, z)
    const int z0 = zp0 & (N - 1);
    uint4* qw = reinterpret_cast<uint4*>(smem_q);
#pragma unroll
    for (int b = 0; b < B; b++) {
      const uint4* src = reinterpret_cast<const uint4*>(d.qv[b]) + ((size_t)z0 * d.dim0 + d.j0);
      for (int j = threadIdx.x; j < d.nj; j += 256) qw[b * d.nj + j] = src[j];
    }
    __syncthreads();
  }
  if (plane >= d.planes) return;
  const int npairs = d.nj >> 1;
  const u32* base = reinterpret_cast<const u32*>(d.db) + ((size_t)zp * npairs * chunks + chunk) * 448;
  const size_t ustride = (size_t)chunks * 448;
  const size_t qoff = (size_t)z * d.dim0 + d.j0;
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  const u32 M = 0x0FFFFFFFu;
  u64 acc[B][8];
#pragma unroll
  for (int b = 0; b < B; b++)
#pragma unroll
    for (int k = 0; k < 8; k++) acc[b][k] = 0;
  // software pipeline: the loads of the next U row pairs are issued before the current U are consumed
  u32x4_t va[U], na[U];
  u32x3_t vb[U], nb[U];
#pragma unroll
  for (int uu = 0; uu < U; uu++) {
    const u32* u = base + (size_t)uu * ustride;
    va[uu] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(u + lane * 4));
    vb[uu] = __builtin_nontemporal_load(reinterpret_cast<const u32x3_t*>(u + 256 + lane * 3));
  }
  for (int jp0 = 0; jp0 < npairs; jp0 += U) {
    const bool more = jp0 + U < npairs;
    if (more) {
#pragma unroll
      for (int uu = 0; uu < U; uu++) {
        const u32* u = base + (size_t)(jp0 + U + uu) * ustride;
        na[uu] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(u + lane * 4));
        nb[uu] = __builtin_nontemporal_load(reinterpret_cast<const u32x3_t*>(u + 256 + lane * 3));
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the multiply-accumulates
#pragma unroll
    for (int uu = 0; uu < U; uu++) {
      const int jp = jp0 + uu;
      const u32 d0 = va[uu].x, d1 = va[uu].y, d2 = va[uu].z, d3 = va[uu].w, d4 = vb[uu].x, d5 = vb[uu].y, d6 = vb[uu].z;
      const u32 f0 = d0 & M;
      const u32 f1 = __builtin_amdgcn_alignbit(d1, d0, 28) & M;
      const u32 f2 = __builtin_amdgcn_alignbit(d2, d1, 24) & M;
      const u32 f3 = __builtin_amdgcn_alignbit(d3, d2, 20) & M;
      const u32 f4 = __builtin_amdgcn_alignbit(d4, d3, 16) & M;
      const u32 f5 = __builtin_amdgcn_alignbit(d5, d4, 12) & M;
      const u32 f6 = __builtin_amdgcn_alignbit(d6, d5, 8) & M;
      const u32 f7 = d6 >> 4;
#pragma unroll
      for (int b = 0; b < B; b++) {
        const uint4* __restrict__ qrow = QLDS ? qs + b * d.nj : reinterpret_cast<const uint4*>(d.qv[b]) + qoff;
        const uint4 qa = qrow[2 * jp];
        const uint4 qb = qrow[2 * jp + 1];
        acc[b][0] += (u64)qa.x * f0; acc[b][1] += (u64)qa.z * f0; acc[b][2] += (u64)qa.y * f1; acc[b][3] += (u64)qa.w * f1;
        acc[b][4] += (u64)qa.x * f2; acc[b][5] += (u64)qa.z * f2; acc[b][6] += (u64)qa.y * f3; acc[b][7] += (u64)qa.w * f3;
        acc[b][0] += (u64)qb.x * f4; acc[b][1] += (u64)qb.z * f4; acc[b][2] += (u64)qb.y * f5; acc[b][3] += (u64)qb.w * f5;
        acc[b][4] += (u64)qb.x * f6; acc[b][5] += (u64)qb.z * f6; acc[b][6] += (u64)qb.y * f7; acc[b][7] += (u64)qb.w * f7;
      }
    }
    if (!more || ((jp0 + U) & 127) == 0) {  // at most 256 rows of < 2^56 products between Barrett folds
#pragma unroll
      for (int b = 0; b < B; b++) {
        acc[b][0] = reduce64(acc[b][0], m0); acc[b][1] = reduce64(acc[b][1], m0);
        acc[b][2] = reduce64(acc[b][2], m1); acc[b][3] = reduce64(acc[b][3], m1);
        acc[b][4] = reduce64(acc[b][4], m0); acc[b][5] = reduce64(acc[b][5], m0);
        acc[b][6] = reduce64(acc[b][6], m1); acc[b][7] = reduce64(acc[b][7], m1);
      }
    }
#pragma unroll
    for (int uu = 0; uu < U; uu++) {
      va[uu] = na[uu];
      vb[uu] = nb[uu];
    }
  }
  const size_t rc = (size_t)N * d.num_per;
  const size_t zi = (size_t)plane * 4 * rc + (size_t)z * d.num_per + (size_t)chunk * 128 + 2 * lane;
#pragma unroll
  for (int b = 0; b < B; b++) {
    u32* o = d.out[b] + zi;
    *reinterpret_cast<uint2*>(o + 0 * rc) = make_uint2((u32)acc[b][0], (u32)acc[b][4]);  // r=0, crt=0
    *reinterpret_cast<uint2*>(o + 1 * rc) = make_uint2((u32)acc[b][2], (u32)acc[b][6]);  // r=0, crt=1
    *reinterpret_cast<uint2*>(o + 2 * rc) = make_uint2((u32)acc[b][1], (u32)acc[b][5]);  // r=1, crt=0
    *reinterpret_cast<uint2*>(o + 3 * rc) = make_uint2((u32)acc[b][3], (u32)acc[b][7]);  // r=1, crt=1
  }
}
void launch_sweep_batch(const DevTables& T, const SweepBatchDesc& d, hipStream_t s) {
  const long units = (long)d.planes * N * (d.num_per >> 7);
  const dim3 grid((unsigned)((units + 3) / 4));
  const bool unroll = ((d.nj >> 1) % 4) == 0;
  static const int lds_min_b = [] { const char* e = getenv("SPIRAL_BATCH_QLDS_MIN"); return e ? atoi(e) : 5; }();
  const bool qlds = unroll && ((d.num_per >> 7) % 4) == 0 && d.batch >= lds_min_b && (size_t)d.batch * d.nj * 16 <= 65536;
  const size_t lds = qlds ? (size_t)d.batch * d.nj * 16 : 0;
#define SP_BATCH_CASE(B)                                                                              \
  case B:                                                                                             \
    if (qlds)                                                                                         \
      hipLaunchKernelGGL((k_sweep_packed_batch<B, (B >= 7 ? 2 : 4), true>), grid, dim3(256), lds, s, T, d); \
    else if (unroll)                                                                                  \
      hipLaunchKernelGGL((k_sweep_packed_batch<B, 4, false>), grid, dim3(256), 0, s, T, d);           \
    else                                                                                              \
      hipLaunchKernelGGL((k_sweep_packed_batch<B, 1, false>), grid, dim3(256), 0, s, T, d);           \
    break;
  switch (d.batch) {
    SP_BATCH_CASE(1) SP_BATCH_CASE(2) SP_BATCH_CASE(3) SP_BATCH_CASE(4)
    SP_BATCH_CASE(5) SP_BATCH_CASE(6) SP_BATCH_CASE(7) SP_BATCH_CASE(8)
    default: break;
  }
#undef SP_BATCH_CASE
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

// NARROW (num_per <= 64): one workgroup per (plane, z).  The nj*num_per words of the row block are
// contiguous; thread tau reads word tau + 256*s, i.e. fixed ii = tau % num_per and rows
// j = tau/num_per + s*(256/num_per).  Query limbs for this z are staged in LDS (16 B per row);
// partial sums are Barrett-folded to < 2^28 and combined through LDS.
__global__ __launch_bounds__(256) void k_sweep_narrow(DevTables T, SweepDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* qs = reinterpret_cast<uint4*>(smem);                          // [nj]
  u32* red = reinterpret_cast<u32*>(smem + (size_t)d.nj * sizeof(uint4));  // [256][4]
  const int tau = threadIdx.x;
  const int zp = blockIdx.x;
  const int z = zp & (N - 1);
  const int plane = zp >> POLY_LEN_LOG2;
  const uint4* qrow = reinterpret_cast<const uint4*>(d.qv) + ((size_t)z * d.dim0 + d.j0);
  for (int j = tau; j < d.nj; j += 256) qs[j] = qrow[j];
  __syncthreads();
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  const u64* p = d.db + (size_t)zp * d.nj * d.num_per;
  const int L = d.nj * d.num_per;
  const int np_log = __ffs(d.num_per) - 1;
  u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int cnt = 0;
  for (int f = tau; f < L; f += 256) {
    const u64 w = p[f];
    const uint4 qa = qs[f >> np_log];
    const u32 bl = (u32)w, bh = (u32)(w >> 32);
    a0 += (u64)qa.x * bl;
    a1 += (u64)qa.z * bl;
    a2 += (u64)qa.y * bh;
    a3 += (u64)qa.w * bh;
    if (++cnt == 255) {
      cnt = 0;
      a0 = reduce64(a0, m0);
      a1 = reduce64(a1, m0);
      a2 = reduce64(a2, m1);
      a3 = reduce64(a3, m1);
    }
  }
  red[tau * 4 + 0] = reduce64(a0, m0);
  red[tau * 4 + 1] = reduce64(a1, m0);
  red[tau * 4 + 2] = reduce64(a2, m1);
  red[tau * 4 + 3] = reduce64(a3, m1);
  __syncthreads();
  // 4*num_per outputs; thread t < 4*num_per: which = t / num_per, ii = t % num_per
  if (tau < 4 * d.num_per) {
    const int which = tau >> np_log, ii = tau & (d.num_per - 1);
    u64 sacc = 0;
    for (int t2 = ii; t2 < 256; t2 += d.num_per) sacc += red[t2 * 4 + which];
    const u32 r = reduce64(sacc, which < 2 ? m0 : m1);
    // which: 0 n0_0 (r0,c0)  1 n0_1 (r1,c0)  2 n1_0 (r0,c1)  3 n1_1 (r1,c1)
    const int rr = which & 1, cc = which >> 1;
    d.out[sweep_out_index(d, plane, rr * 2 + cc, z, ii)] = r;
  }
}

// NARROW with 16-byte non-temporal loads (2 <= num_per <= 64): thread tau reads words 2 tau, 2 tau + 1 (+512 s):
// the same row j, columns ii0 = (2 tau) % num_per and ii0 + 1.
__global__ __launch_bounds__(256) void k_sweep_narrow2(DevTables T, SweepDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* qs = reinterpret_cast<uint4*>(smem);                              // [nj]
  u32* red = reinterpret_cast<u32*>(smem + (size_t)d.nj * sizeof(uint4));  // [256][8]
  const int tau = threadIdx.x;
  const int zp = blockIdx.x;
  const int z = zp & (N - 1);
  const int plane = zp >> POLY_LEN_LOG2;
  const uint4* qrow = reinterpret_cast<const uint4*>(d.qv) + ((size_t)z * d.dim0 + d.j0);
  for (int j = tau; j < d.nj; j += 256) qs[j] = qrow[j];
  __syncthreads();
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  const u64* p = d.db + (size_t)zp * d.nj * d.num_per;
  const int L = d.nj * d.num_per;
  const int np_log = __ffs(d.num_per) - 1;
  u64 a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int cnt = 0;
  for (int f = 2 * tau; f < L; f += 512) {
    ulonglong2 w;
    w.x = __builtin_nontemporal_load(p + f);
    w.y = __builtin_nontemporal_load(p + f + 1);
    const uint4 qa = qs[f >> np_log];
    const u32 b0l = (u32)w.x, b0h = (u32)(w.x >> 32), b1l = (u32)w.y, b1h = (u32)(w.y >> 32);
    a[0] += (u64)qa.x * b0l; a[1] += (u64)qa.z * b0l; a[2] += (u64)qa.y * b0h; a[3] += (u64)qa.w * b0h;
    a[4] += (u64)qa.x * b1l; a[5] += (u64)qa.z * b1l; a[6] += (u64)qa.y * b1h; a[7] += (u64)qa.w * b1h;
    if (++cnt == 255) {
      cnt = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = reduce64(a[i], (i & 2) ? m1 : m0);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) red[tau * 8 + i] = reduce64(a[i], (i & 2) ? m1 : m0);
  __syncthreads();
  if (tau < 4 * d.num_per) {
    const int which = tau >> np_log, ii = tau & (d.num_per - 1);
    const int slot = (ii & 1) * 4 + which, first = ii >> 1, step = d.num_per >> 1;
    u64 sacc = 0;
    for (int t2 = first; t2 < 256; t2 += step) sacc += red[t2 * 8 + slot];
    const u32 r = reduce64(sacc, which < 2 ? m0 : m1);
    const int rr = which & 1, cc = which >> 1;
    d.out[sweep_out_index(d, plane, rr * 2 + cc, z, ii)] = r;
  }
}

const char* sweep_kernel_name(int num_per) { return num_per >= 128 ? "k_sweep_packed" : "k_sweep_narrow"; }

void launch_sweep_persist(const DevTables& T, const SweepDesc& d, int wgs_per_cu, int unroll, hipStream_t s);
void launch_sweep(const DevTables& T, const SweepDesc& d, hipStream_t s) {
  // default: persistent grid of 4 workgroups per CU, 4 row pairs in flight per lane (profiles/r01_sweep_variants.md);
  // SPIRAL_SWEEP_PERSIST_WGS=0 selects the one-wave-per-unit grid
  static const int persist_wgs = [] { const char* e = getenv("SPIRAL_SWEEP_PERSIST_WGS"); return e ? atoi(e) : 4; }();
  static const int persist_unr = [] { const char* e = getenv("SPIRAL_SWEEP_PERSIST_UNROLL"); return e ? atoi(e) : 4; }();
  if (d.packed && persist_wgs > 0) {
    launch_sweep_persist(T, d, persist_wgs, persist_unr, s);
    return;
  }
  if (d.packed) {
    const long units = (long)d.planes * N * (d.num_per >> 7);
    hipLaunchKernelGGL(k_sweep_packed, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, T, d);
  } else if (d.num_per >= 128) {
    const long units = (long)d.planes * N * (d.num_per >> 7);
    static const int variant = [] {
      const char* e = getenv("SPIRAL_SWEEP_VARIANT");
      return e ? atoi(e) : 0;
    }();
    const dim3 grid((unsigned)((units + 3) / 4));
    // measured on MI355X, C2 (profiles/r01_sweep_variants.md): non-temporal loads + no manual unroll is
    // the fastest form (6.8 TB/s); the others stay selectable for A/B runs
    switch (variant) {
      case 1: hipLaunchKernelGGL((k_sweep_wide<8, true>), grid, dim3(256), 0, s, T, d); break;
      case 2: hipLaunchKernelGGL((k_sweep_wide<8, false>), grid, dim3(256), 0, s, T, d); break;
      case 4: hipLaunchKernelGGL((k_sweep_wide<4, true>), grid, dim3(256), 0, s, T, d); break;
      case 5: hipLaunchKernelGGL((k_sweep_wide<2, true>), grid, dim3(256), 0, s, T, d); break;
      default: hipLaunchKernelGGL((k_sweep_wide<1, true>), grid, dim3(256), 0, s, T, d); break;
    }
  } else {
    if (d.num_per >= 2 && !getenv("SPIRAL_NARROW1")) {
      size_t sh = (size_t)d.nj * sizeof(uint4) + 256 * 8 * sizeof(u32);
      hipLaunchKernelGGL(k_sweep_narrow2, dim3((unsigned)(d.planes * N)), dim3(256), sh, s, T, d);
    } else {
      size_t sh = (size_t)d.nj * sizeof(uint4) + 256 * 4 * sizeof(u32);
      hipLaunchKernelGGL(k_sweep_narrow, dim3((unsigned)(d.planes * N)), dim3(256), sh, s, T, d);
    }
  }
}

// reference [z][ii][j] -> device [z][j - j0][ii] for nz rows starting at z0 (32x32 LDS tile transpose)
__global__ __launch_bounds__(256) void k_db_relayout(u64* dst_plane, const u64* src, int z0, int num_per, int dim0,
                                                     int j0, int nj, ColMap cm) {
  __shared__ u64 tile[32][33];
  const int zl = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int jt = blockIdx.x * 32, it = blockIdx.y * 32;
  const u64* s = src + (size_t)zl * cm.np_global * dim0;
  u64* dpl = dst_plane + (size_t)(z0 + zl) * nj * num_per;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int ii = it + ty + 8 * i, j = jt + tx;
    if (ii < num_per && j < nj) tile[ty + 8 * i][tx] = s[(size_t)(cm.off + cm.stride * ii) * dim0 + j0 + j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int j = jt + ty + 8 * i, ii = it + tx;
    if (ii < num_per && j < nj) dpl[(size_t)j * num_per + ii] = tile[tx][ty + 8 * i];
  }
}
// reference layout -> PACKED: one thread per (z, jp, chunk, lane); gathers its 4 words (one-time cost)
__global__ __launch_bounds__(256) void k_db_relayout_packed(u32* dst, int plane, const u64* src, int z0, int nz,
                                                            int num_per, int dim0, int j0, int nj, ColMap cm) {
  const int chunks = num_per >> 7, npairs = nj >> 1;
  const size_t total = (size_t)nz * npairs * chunks * 64;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    size_t t = i >> 6;
    const int chunk = (int)(t % chunks);
    t /= chunks;
    const int jp = (int)(t % npairs);
    const int zl = (int)(t / npairs);
    const int ii = chunk * 128 + 2 * lane;
    const u64* s = src + ((size_t)zl * cm.np_global + cm.off + cm.stride * ii) * dim0 + j0 + 2 * jp;
    const size_t nx = (size_t)cm.stride * dim0;  // next local column
    const u64 w00 = s[0], w10 = s[1], w01 = s[nx], w11 = s[nx + 1];
    u32* unit = dst + ((((size_t)plane * N + (z0 + zl)) * npairs + jp) * chunks + chunk) * 448;
    pack_unit_lane(unit, lane, w00, w01, w10, w11);
  }
}
void launch_db_relayout(u64* dst, int plane, const u64* src, int z0, int nz, int num_per, int dim0, int j0, int nj,
                        int packed, ColMap cm, hipStream_t s) {
  if (cm.np_global == 0) cm.np_global = num_per;
  if (nz <= 0) return;
  if (packed) {
    hipLaunchKernelGGL(k_db_relayout_packed, dim3(4096), dim3(256), 0, s, reinterpret_cast<u32*>(dst), plane, src, z0,
                       nz, num_per, dim0, j0, nj, cm);
  } else {
    u64* dst_plane = dst + (size_t)plane * N * nj * num_per;
    hipLaunchKernelGGL(k_db_relayout, dim3((nj + 31) / 32, (num_per + 31) / 32, nz), dim3(256), 0, s, dst_plane, src,
                       z0, num_per, dim0, j0, nj, cm);
  }
}

__global__ __launch_bounds__(256) void k_db_synth(u64* dst, u64 seed, int num_per, int dim0, int j0, int nj,
                                                  size_t total, ColMap cm) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    // device index i = ((zp * nj) + jl) * num_per + ii
    size_t ii = i % num_per;
    size_t t = i / num_per;
    size_t jl = t % nj;
    size_t zp = t / nj;
    size_t ref = (zp * cm.np_global + cm.off + cm.stride * ii) * dim0 + j0 + jl;
    dst[i] = synth_word(seed, ref);
  }
}
__global__ __launch_bounds__(256) void k_db_synth_packed(u32* dst, u64 seed, int num_per, int dim0, int j0, int nj,
                                                         size_t total_lanes, ColMap cm) {
  const int chunks = num_per >> 7, npairs = nj >> 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_lanes; i += (size_t)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    size_t t = i >> 6;  // unit index = (zp * npairs + jp) * chunks + chunk
    const int chunk = (int)(t % chunks);
    const size_t t2 = t / chunks;
    const int jp = (int)(t2 % npairs);
    const size_t zp = t2 / npairs;
    const size_t ii = (size_t)chunk * 128 + 2 * lane;
    const size_t r0 = (zp * cm.np_global + cm.off + cm.stride * ii) * dim0 + j0 + 2 * jp;  // (row 2jp, ii)
    const size_t r1 = r0 + (size_t)cm.stride * dim0;                                         // (row 2jp, ii+1)
    pack_unit_lane(dst + t * 448, lane, synth_word(seed, r0), synth_word(seed, r1), synth_word(seed, r0 + 1),
                   synth_word(seed, r1 + 1));
  }
}
void launch_db_synth(u64* dst, u64 seed, int planes, int num_per, int dim0, int j0, int nj, int packed, ColMap cm,
                     hipStream_t s) {
  if (cm.np_global == 0) cm.np_global = num_per;
  if (packed) {
    size_t lanes = (size_t)planes * N * (nj >> 1) * (num_per >> 7) * 64;
    hipLaunchKernelGGL(k_db_synth_packed, dim3(256 * 32), dim3(256), 0, s, reinterpret_cast<u32*>(dst), seed, num_per,
                       dim0, j0, nj, lanes, cm);
  } else {
    size_t total = (size_t)planes * N * nj * num_per;
    hipLaunchKernelGGL(k_db_synth, dim3(256 * 32), dim3(256), 0, s, dst, seed, num_per, dim0, j0, nj, total, cm);
  }
}

__global__ void k_db_read(u64* out, const u64* db, int plane, int z, int ii, int jl0, int count, int num_per, int nj,
                          int packed) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int jl = jl0 + t;
  if (packed) {
    const int chunks = num_per >> 7, npairs = nj >> 1;
    const int chunk = ii >> 7, lane = (ii & 127) >> 1, iiofs = ii & 1;
    const u32* unit = reinterpret_cast<const u32*>(db) +
                      ((((size_t)plane * N + z) * npairs + (jl >> 1)) * chunks + chunk) * 448;
    out[t] = unpack_word(unit, lane, (jl & 1) * 2 + iiofs);
  } else {
    out[t] = db[(((size_t)plane * N + z) * nj + jl) * num_per + ii];
  }
}
void launch_db_read(u64* out, const u64* db, int plane, int z, int ii, int jl0, int count, int num_per, int nj,
                    int packed, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_db_read, dim3((count + 63) / 64), dim3(64), 0, s, out, db, plane, z, ii, jl0, count, num_per, nj,
                     packed);
}

// ------------------------------------------------------------------------------------------------
// database preprocessing (server.rs:277-357) -- grid (max(num_per/2,1), njp, planes)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 item_coeff(const DbEncodeDesc& d, size_t item, int chunk_idx, int z) {
  // load_item_from_seek: chunk bytes at item*db_item_size + chunk_idx*bytes_per_chunk, logp bits per coefficient
  const size_t pos = item * (size_t)d.db_item_size + (size_t)chunk_idx * d.bytes_per_chunk;
  if (pos >= d.file_len) return 0u;
  const size_t avail = d.file_len - pos;
  const int bytes_read = (int)(avail < (size_t)d.bytes_per_chunk ? avail : (size_t)d.bytes_per_chunk);
  const int words_read = (bytes_read * 8 + d.logp - 1) / d.logp;
  if (z >= words_read) return 0u;
  const size_t wpos = pos - d.win_item0 * (size_t)d.db_item_size;
  const int bit = z * d.logp;
  const int b0 = bit >> 3, sh = bit & 7;
  u64 acc = 0;
  const int nb = (sh + d.logp + 7) >> 3;
  for (int i = 0; i < nb; i++) {
    const int bi = b0 + i;
    const u64 byte = (bi < bytes_read && wpos + bi < d.win_bytes) ? (u64)d.win[wpos + bi] : 0ULL;
    acc |= byte << (8 * i);
  }
  return (u32)((acc >> sh) & ((1ULL << d.logp) - 1ULL));
}

__global__ __launch_bounds__(256) void k_db_encode(DevTables T, DbEncodeDesc d) {
  __shared__ u32 lds0[LDS_WORDS];
  __shared__ u32 lds1[LDS_WORDS];
  const int tau = threadIdx.x;
  const int qd = d.only_item >= 0 ? d.only_q : blockIdx.x;
  const int jp = d.jp0 + blockIdx.y, plane = blockIdx.z;
  u64 w[4][8];  // [row a * 2 + ii b][k]: words at z = 8 tau + k
  u32* la = lds0;
  u32* lb = lds1;
#pragma unroll 1
  for (int ab = 0; ab < 4; ab++) {
    const int a = ab >> 1, b = ab & 1;
    const int jl = 2 * jp + a, ii = 2 * qd + b;
    const bool valid = jl < d.nj && ii < d.num_per;
    // i = j * num_per + ii (server.rs:332-333), ii = global column of local column `ii`
    const size_t item = (size_t)(d.j0 + jl) * d.cm.np_global + d.cm.off + (size_t)d.cm.stride * ii;
    if (d.only_item >= 0 && (long)item != d.only_item) {
      // keep the neighbour's resident words
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int z = 8 * tau + k;
        u64 cur = 0;
        if (valid) {
          if (d.packed) {
            const int chunks = d.num_per >> 7, npairs = d.nj >> 1;
            const u32* unit = reinterpret_cast<const u32*>(d.db) +
                              ((((size_t)plane * N + z) * npairs + jp) * chunks + (ii >> 7)) * 448;
            cur = unpack_word(unit, (ii & 127) >> 1, a * 2 + b);
          } else {
            cur = d.db[(((size_t)plane * N + z) * d.nj + jl) * d.num_per + ii];
          }
        }
        w[ab][k] = cur;
      }
      continue;
    }
    u32 lo[8], hi[8];
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
      const ModConst m = T.c.mod[c];
      u32 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        u32 x = valid ? item_coeff(d, item, plane, tau + 256 * k) : 0u;
        // recenter_mod(x, p, Q) reduced mod q_c: values above p/2 are negative
        v[k] = x > d.pt_modulus / 2 ? m.q - (d.pt_modulus - x) : x;
      }
      const u32* fw = T.tw + (size_t)c * 4 * N;
      ntt_fwd_block(v, tau, la, lb, fw, fw + N, m.q, m.two_q);
      {
        u32* tmp = la;
        la = lb;
        lb = tmp;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (c == 0)
          lo[k] = v[k];
        else
          hi[k] = v[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) w[ab][k] = (u64)lo[k] | ((u64)hi[k] << 32);
  }
  const int jl0 = 2 * jp, ii0 = 2 * qd;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int z = 8 * tau + k;
    if (d.packed) {
      const int chunks = d.num_per >> 7, npairs = d.nj >> 1;
      u32* unit = reinterpret_cast<u32*>(d.db) +
                  ((((size_t)plane * N + z) * npairs + jp) * chunks + (ii0 >> 7)) * 448;
      pack_unit_lane(unit, (ii0 & 127) >> 1, w[0][k], w[1][k], w[2][k], w[3][k]);
    } else {
#pragma unroll
      for (int ab = 0; ab < 4; ab++) {
        const int jl = jl0 + (ab >> 1), ii = ii0 + (ab & 1);
        if (jl < d.nj && ii < d.num_per) d.db[(((size_t)plane * N + z) * d.nj + jl) * d.num_per + ii] = w[ab][k];
      }
    }
  }
}
void launch_db_encode(const DevTables& T, const DbEncodeDesc& d, hipStream_t s) {
  if (d.njp <= 0) return;
  const unsigned gx = d.only_item >= 0 ? 1u : (unsigned)((d.num_per + 1) / 2);
  hipLaunchKernelGGL(k_db_encode, dim3(gx, d.njp, d.planes), dim3(256), 0, s, T, d);
}

__global__ __launch_bounds__(256) void k_sweep_out_to_ref(u64* out, const u32* in, int num_per) {
  // out[ii][r][crt][z] <- in[r][crt][z][ii]
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)4 * N * num_per;
  if (i >= total) return;
  size_t z = i & (N - 1);
  size_t rc = (i >> POLY_LEN_LOG2) & 3;
  size_t ii = i >> (POLY_LEN_LOG2 + 2);
  out[i] = (u64)in[(rc * N + z) * num_per + ii];
}
void launch_sweep_out_to_ref(u64* out, const u32* in, int num_per, hipStream_t s) {
  size_t total = (size_t)4 * N * num_per;
  hipLaunchKernelGGL(k_sweep_out_to_ref, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, in, num_per);
}

}  // namespace spiral
