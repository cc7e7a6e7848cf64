// One round of coefficient_expansion (server.rs:19-121) for one ciphertext per workgroup, in ONE launch:
//   ct      = v[i]                      (second half of the round: neg1[r] * v[i - num_in], server.rs:105-110)
//   ct_auto = automorph(from_ntt(ct), t)                                              (poly.rs:393-405)
//   v'[i]   = ct + W * to_ntt(G^-1(ct_auto row 0)) + [0 ; to_ntt(ct_auto row 1)]      (server.rs:89-102)
// The three-kernel form (k_ntt_inv -> k_ntt_fwd3 -> k_mac2) costs three dependent launches per round and the rounds
// are a chain: ten rounds of a C2 query took 0.7 ms although the arithmetic of the first six is a few microseconds.
// Here a workgroup of eight waves keeps everything of its ciphertext on chip (wave-per-transform NTT, wave_ntt.hpp):
//   phase 1  waves 0-3 inverse-transform (row, modulus) = (w >> 1, w & 1) -- waves 4-7 stage the forward tables of both
//            moduli meanwhile -- Garner + the automorphism scatter the 64-bit coefficients of both rows into LDS
//   phase 2  waves 0-3 work modulus 0, waves 4-7 modulus 1: the t digit polynomials of row 0 and the residues of row 1
//            are dealt to the four waves of a modulus (transform dg -> wave dg % 4); each wave transforms its
//            polynomials one after the other and multiply-accumulates them with W's two rows into private 64-bit sums
//            (the row-1 polynomial is simply added to row 1's sums)
//   phase 3  the four partial sums of a modulus are combined through LDS (as in k_fold_wave); two waves per modulus add
//            the input ciphertext and store the result
// Reads v from `src` and writes to `dst` (the caller ping-pongs two buffers): a round's second-half workgroup reads
// v[i - num_in] while the first-half workgroup of the same launch replaces v[i - num_in].
// W: this round's 2 x t key-switching matrix in wave layout (wave_layout_word), polynomial (row * t + k).
#include "kernels.hpp"
#include "wave_ntt.hpp"

namespace spiral {

constexpr int EXP_WBUF = 8 * WBUF_WORDS;          // words: eight transpose buffers
constexpr int EXP_LDS_WORDS = EXP_WBUF + 4 * N + 4 * N;  // + forward tables of both moduli + two raw rows (u64)

struct ExpandMac {  // hooks into wntt_fwd
  u64 (&acc0)[32];
  u64 (&acc1)[32];
  const u32x4w_t* a0;
  const u32x4w_t* a1;
  bool unit;  // the row-1 polynomial: + [0 ; poly] instead of W * poly
  u32x4w_t m0[8], m1[8];
  __device__ __forceinline__ void fetch(int g) {
    if (!unit) {
      m0[g] = a0[64 * g];
      m1[g] = a1[64 * g];
    }
  }
  __device__ __forceinline__ void mac(int g, const u32 (&v)[32]) {
    if (unit) {
#pragma unroll
      for (int e = 0; e < 4; e++) acc1[4 * g + e] += v[4 * g + e];
    } else {
      acc0[4 * g] += (u64)m0[g].x * v[4 * g]; acc0[4 * g + 1] += (u64)m0[g].y * v[4 * g + 1];
      acc0[4 * g + 2] += (u64)m0[g].z * v[4 * g + 2]; acc0[4 * g + 3] += (u64)m0[g].w * v[4 * g + 3];
      acc1[4 * g] += (u64)m1[g].x * v[4 * g]; acc1[4 * g + 1] += (u64)m1[g].y * v[4 * g + 1];
      acc1[4 * g + 2] += (u64)m1[g].z * v[4 * g + 2]; acc1[4 * g + 3] += (u64)m1[g].w * v[4 * g + 3];
    }
  }
  __device__ __forceinline__ void before_t4() {
    fetch(0); fetch(1); fetch(2); fetch(3);
  }
  __device__ __forceinline__ void before_t1() {
    fetch(4); fetch(5);
  }
  __device__ __forceinline__ void after_quarter(int qq, u32 (&v)[32]) {
    if (qq == 0) {
      fetch(6); fetch(7);
      SP_SB();
    }
    mac(2 * qq, v);
    mac(2 * qq + 1, v);
  }
};

__global__ __launch_bounds__(512) void k_expand_round(DevTables T, ExpandDesc d) {
  __shared__ __attribute__((aligned(16))) u32 smem[EXP_LDS_WORDS];
  u32* wbuf = smem;
  u32* ltw = smem + EXP_WBUF;                                   // [modulus][w | w'] (swizzled, wtw_stage)
  u64* raw = reinterpret_cast<u64*>(smem + EXP_WBUF + 4 * N);   // [row][N] automorphed raw ciphertext
  u32* exch = wbuf + 4 * WBUF_WORDS;                            // phase 1 only: modulus-1 residues [row][N] (waves 4-7 idle)
  const int tau = threadIdx.x, lane = tau & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tau >> 6);
  const int grp = (int)blockIdx.x < d.n[0] ? 0 : 1;
  const int e = (int)blockIdx.x - (grp ? d.n[0] : 0);
  const int ct = d.ct_idx[grp][e];
  const int t = d.t[grp], bits = d.bits[grp];
  const bool second = ct >= d.num_in;
  u32* mybuf = wbuf + wv * WBUF_WORDS;
  // ---- phase 1
  u32 x[32];
  if (wv < 4) {
    const int row = wv >> 1, c = wv & 1;
    const ModConst m = T.c.mod[c];
    const u32* sp = d.src + (((size_t)(second ? ct - d.num_in : ct) * 2 + row) * 2 + c) * N + 32 * lane;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      const u32x4w_t t4 = *reinterpret_cast<const u32x4w_t*>(sp + 4 * g);
      x[4 * g] = t4.x; x[4 * g + 1] = t4.y; x[4 * g + 2] = t4.z; x[4 * g + 3] = t4.w;
    }
    if (second) {  // v[ct] = neg1 * v[ct - num_in]; parked in the output slot (phase 3 adds it back in)
      const u32* sc = d.neg1 + (size_t)c * N + 32 * lane;
      u32* pk = d.dst + (((size_t)ct * 2 + row) * 2 + c) * N + 32 * lane;
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const u32x4w_t s4 = *reinterpret_cast<const u32x4w_t*>(sc + 4 * g);
        u32x4w_t o;
        o.x = x[4 * g] = reduce64((u64)x[4 * g] * s4.x, m);
        o.y = x[4 * g + 1] = reduce64((u64)x[4 * g + 1] * s4.y, m);
        o.z = x[4 * g + 2] = reduce64((u64)x[4 * g + 2] * s4.z, m);
        o.w = x[4 * g + 3] = reduce64((u64)x[4 * g + 3] * s4.w, m);
        *reinterpret_cast<u32x4w_t*>(pk + 4 * g) = o;
      }
    }
    wntt_inv(x, lane, mybuf, T.tw + ((size_t)c * 4 + 2) * N, m.q, m.two_q);  // -> coefficient 64k + lane
    if (c == 1) {
#pragma unroll
      for (int k = 0; k < 32; k++) exch[row * N + 64 * k + lane] = x[k];
    }
  } else {
    wtw_stage(ltw, T.tw, tau - 256);
    wtw_stage(ltw + 2 * N, T.tw + 4 * N, tau - 256);
  }
  __syncthreads();
  if (wv < 4 && (wv & 1) == 0) {
    const int row = wv >> 1;
    const u32 q0 = T.c.mod[0].q, q1 = T.c.mod[1].q;
#pragma unroll
    for (int k = 0; k < 32; k++) {
      const int z = 64 * k + lane;
      const u32 xx = x[k], y = exch[row * N + z];
      const u32 xm = xx >= q1 ? xx - q1 : xx;  // q0 < 2*q1
      const u32 dd = y >= xm ? y - xm : y + q1 - xm;
      const u32 qt = __umulhi(dd, T.c.q0_inv_q1_sh);
      u32 ee = dd * T.c.q0_inv_q1 - qt * q1;
      ee = ee >= q1 ? ee - q1 : ee;
      const u64 val = (u64)xx + (u64)q0 * (u64)ee;
      const unsigned zt = (unsigned)z * (unsigned)d.t_auto;  // poly.rs:393-405
      const unsigned num = zt >> POLY_LEN_LOG2, rem = zt & (N - 1);
      raw[row * N + rem] = (num & 1u) ? T.c.Q - val : val;
    }
  }
  __syncthreads();
  // ---- phase 2
  const int c = wv >> 2, sub = wv & 3;
  const ModConst m = T.c.mod[c];
  const u32* fw = T.tw + (size_t)c * 4 * N;
  const u32* ltwc = ltw + c * 2 * N;
  const u64 mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL);
  u64 acc0[32], acc1[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc0[k] = acc1[k] = 0;
#pragma unroll 1
  for (int dg = sub; dg <= t; dg += 4) {
    int ln = lane;  // see k_fold_wave: keeps loop-invariant addresses and twiddles from being hoisted (and spilled)
    const u32* fwi = fw;
    asm volatile("" : "+v"(ln));
    asm volatile("" : "+s"(fwi));
    WaveScalarTw stw;
    wntt_scalar_tw(stw, fwi);
    const bool unit = dg == t;
    const u64* rp = raw + (unit ? N : 0) + ln;
    const int sh = dg * bits;
    u32 v[32];
    if (unit || bits > 28) {  // to_ntt (residue of the whole word) / wide digits: reduce  (gadget.rs:48-53)
      const u64 dmask = unit ? ~0ULL : (sh >= 64 ? 0ULL : mask);
      const int shc = unit ? 0 : (sh & 63);
#pragma unroll
      for (int k = 0; k < 32; k++) v[k] = reduce64((rp[64 * k] >> shc) & dmask, m);
    } else {
      const u64 dmask = sh >= 64 ? 0ULL : mask;
      const int shc = sh & 63;
#pragma unroll
      for (int k = 0; k < 32; k++) v[k] = (u32)((rp[64 * k] >> shc) & dmask);
    }
    ExpandMac hk{acc0, acc1, reinterpret_cast<const u32x4w_t*>(d.W[grp] + ((size_t)dg * 2 + c) * N) + ln,
                 reinterpret_cast<const u32x4w_t*>(d.W[grp] + ((size_t)(t + dg) * 2 + c) * N) + ln, unit};
    wntt_fwd<false>(v, ln, mybuf, fwi, stw, ltwc, m.q, m.two_q, hk);
  }
  // ---- phase 3: the four partial sums of this modulus -> wave sub 0 (row 0) and sub 1 (row 1)
  u32 r0[32], r1[32];
  int lt = lane;
  asm volatile("" : "+v"(lt));
#pragma unroll
  for (int k = 0; k < 32; k++) {
    r0[k] = reduce64(acc0[k], m);
    r1[k] = reduce64(acc1[k], m);
  }
  u32x4w_t* sc = reinterpret_cast<u32x4w_t*>(smem + EXP_WBUF) + c * 2048;  // 4 regions of 512 vectors per modulus
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
  __syncthreads();  // tables and raw rows are dead: their 64 KiB are the reduction scratch now
  if (sub == 2) { SP_PUT(r0, 0) SP_PUT(r1, 1) }
  if (sub == 3) { SP_PUT(r0, 2) SP_PUT(r1, 3) }
  __syncthreads();
  if (sub == 0) { SP_ADD(r0, 0) SP_ADD(r0, 2) }
  if (sub == 1) { SP_ADD(r1, 1) SP_ADD(r1, 3) }
  __syncthreads();
  if (sub == 0) { SP_PUT(r1, 0) }
  if (sub == 1) { SP_PUT(r0, 1) }
  __syncthreads();
  if (sub == 0) { SP_ADD(r0, 1) }
  if (sub == 1) { SP_ADD(r1, 0) }
#undef SP_PUT
#undef SP_ADD
  if (sub < 2) {  // rows in the reference's coefficient order: lane holds 32 lane .. 32 lane + 31
    const size_t slot = (((size_t)ct * 2 + sub) * 2 + c) * N + 32 * lt;
    const u32* ad = (second ? d.dst : d.src) + slot;
    u32* op = d.dst + slot;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      const u32x4w_t a4 = *reinterpret_cast<const u32x4w_t*>(ad + 4 * g);
      u32x4w_t o;
      o.x = add_mod(sub == 0 ? r0[4 * g] : r1[4 * g], a4.x, m.q);
      o.y = add_mod(sub == 0 ? r0[4 * g + 1] : r1[4 * g + 1], a4.y, m.q);
      o.z = add_mod(sub == 0 ? r0[4 * g + 2] : r1[4 * g + 2], a4.z, m.q);
      o.w = add_mod(sub == 0 ? r0[4 * g + 3] : r1[4 * g + 3], a4.w, m.q);
      *reinterpret_cast<u32x4w_t*>(op + 4 * g) = o;
    }
  }
}

void launch_expand_round(const DevTables& T, const ExpandDesc& d, hipStream_t s) {
  const int blocks = std::max(d.n[0], 0) + std::max(d.n[1], 0);
  if (blocks <= 0) return;
  ExpandDesc dd = d;
  dd.n[0] = std::max(d.n[0], 0);
  dd.n[1] = std::max(d.n[1], 0);
  hipLaunchKernelGGL(k_expand_round, dim3(blocks), dim3(512), 0, s, T, dd);
  launched(PATH_EXPAND_FUSED, "k_expand_round");
}

}  // namespace spiral
