// Wave-per-transform 2048-point negacyclic NTT for gfx950: ONE wavefront holds a whole polynomial, 32 coefficients
// per lane, and runs all 11 stages without a workgroup barrier.
//
//   layout A  lane l, register k  =  x[64 k + l]     stages t = 1024 .. 64 pair registers k, k + t/64: in-register, the
//                                                     twiddle index m + (64k + l) / 2t = m + k / (t/32) is wave-uniform
//                                                     (scalar loads, SGPR operands)
//   stage t = 32 pairs lane l with lane l + 32 of the SAME register: v_permlane32_swap_b32 on the register pair
//             (2p, 2p+1) puts both halves of a butterfly into one lane -- lanes 0-31 then work on register 2p's 32
//             butterflies, lanes 32-63 on register 2p+1's: no redundant multiplies
//   one transpose through a wave-private LDS buffer (no s_barrier: a wave's LDS operations execute in order): element
//             n lives at word n + 4 (n >> 5); the writes (b32, from the swapped layout) and the reads (b128) are both
//             bank-conflict-free
//   layout B  lane L, register k  =  x[32 L + k]     stages t = 16 .. 1 in-register; the twiddles of a lane are
//                                                     contiguous table ranges m + L (16/t) + k/(2t)
//
// 80 butterflies per lane sit between two LDS exchanges (12 in the workgroup-cooperative layout of device_common.hpp,
// which also needs three s_barriers per transform).  The arithmetic is the reference's (ntt.rs:67-113, 212-258: Harvey
// butterflies with Shoup quotients, same tables, same in-place index order), so the results are identical.
// Index mapping, swap semantics and LDS addressing were first checked in numpy against the oracle.
#pragma once
#include "device_common.hpp"

namespace spiral {

typedef const __attribute__((address_space(4))) u32 cu32_t;  // constant address space: uniform loads become s_load
typedef u32 u32x4w_t __attribute__((ext_vector_type(4)));
typedef u32 u32x2w_t __attribute__((ext_vector_type(2)));
constexpr int WBUF_WORDS = 64 * 36;  // padded transpose buffer of one wave (9216 bytes)

__device__ __forceinline__ int wave_h(int lane) { return lane < 32 ? lane : lane + 40; }
// a (lanes 0-31) or b (lanes 32-63) of two wave-uniform values: bit select on a lane mask, so that the compiler keeps two
// scalar loads and one v_bfi instead of branching around a per-lane load
__device__ __forceinline__ u32 half_select(u32 lom, u32 a, u32 b) { return (a & lom) | (b & ~lom); }

// physical word of twiddle-table entry idx in the LDS copy: the ranges a lane reads as b128 vectors in stages t = 2
// (entries 512 + 8L + 0..7) and t = 1 (1024 + 16L + 0..15) are rotated per lane group so that the 16 lanes of a b128
// service group hit 16 different bank quads
__device__ __forceinline__ int wtw_phys(int idx) {
  if (idx >= 1024) {
    const int L = (idx - 1024) >> 4, r = (idx - 1024) & 15;
    return 1024 + 16 * L + 4 * (((r >> 2) + (L >> 2)) & 3) + (r & 3);
  }
  if (idx >= 512) {
    const int L = (idx - 512) >> 3, r = (idx - 512) & 7;
    return 512 + 8 * L + 4 * (((r >> 2) + (L >> 3)) & 1) + (r & 3);
  }
  return idx;
}
// stage the forward tables [w | w'] (2N words at tw) of one modulus into ltw (2N words) -- whole workgroup, 256 threads
__device__ __forceinline__ void wtw_stage(u32* ltw, const u32* __restrict__ tw, int tau) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int idx = tau + 256 * k;
    const int ph = wtw_phys(idx);
    ltw[ph] = tw[idx];
    ltw[N + ph] = tw[N + idx];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: v[k] = x[64 k + lane] (values < q)  ->  v[k] = X[32 lane + k], canonical (the reference's output order)
// tw: global tables [w | w'] of this modulus (uniform pointer); ltw: their LDS copy (wtw_stage); buf: this wave's buffer
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wntt_fwd(u32 (&v)[32], int lane, u32* buf, const u32* tw, const u32* ltw, u32 q, u32 q2) {
  const cu32_t* sw = (const cu32_t*)tw;
#pragma unroll
  for (int mm = 0; mm < 5; mm++) {
    const int Tk = 16 >> mm;
#pragma unroll
    for (int k = 0; k < 32; k++) {
      if (((k / Tk) & 1) == 0) {
        const int ti = (1 << mm) + k / (2 * Tk);
        ct_bfly(v[k], v[k + Tk], sw[ti], sw[N + ti], q, q2);
      }
    }
  }
  const int hl = wave_h(lane);
  const u32 lom = lane < 32 ? 0xffffffffu : 0u;
#pragma unroll
  for (int p = 0; p < 16; p++) {
    const auto r = __builtin_amdgcn_permlane32_swap(v[2 * p], v[2 * p + 1], false, false);
    u32 a = r[0], b = r[1];  // lanes 0-31: (x, y) of register 2p; lanes 32-63: (x, y) of register 2p+1
    const u32 w = half_select(lom, sw[32 + 2 * p], sw[32 + 2 * p + 1]);
    const u32 wp = half_select(lom, sw[N + 32 + 2 * p], sw[N + 32 + 2 * p + 1]);
    ct_bfly(a, b, w, wp, q, q2);
    buf[144 * p + hl] = a;        // element 128p + (lo ? l : l + 32)        at n + 4 (n >> 5)
    buf[144 * p + 36 + hl] = b;   // the element 32 above it
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const u32x4w_t t4 = *reinterpret_cast<const u32x4w_t*>(buf + 36 * lane + 4 * g);
    v[4 * g] = t4.x; v[4 * g + 1] = t4.y; v[4 * g + 2] = t4.z; v[4 * g + 3] = t4.w;
  }
  __builtin_amdgcn_wave_barrier();
  {  // t = 16: one twiddle per lane
    const u32 w = ltw[64 + lane], wp = ltw[N + 64 + lane];
#pragma unroll
    for (int k = 0; k < 16; k++) ct_bfly(v[k], v[k + 16], w, wp, q, q2);
  }
  {  // t = 8: two
    const u32x2w_t w2 = *reinterpret_cast<const u32x2w_t*>(ltw + 128 + 2 * lane);
    const u32x2w_t p2 = *reinterpret_cast<const u32x2w_t*>(ltw + N + 128 + 2 * lane);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      ct_bfly(v[k], v[k + 8], w2.x, p2.x, q, q2);
      ct_bfly(v[16 + k], v[24 + k], w2.y, p2.y, q, q2);
    }
  }
  {  // t = 4: four
    const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(ltw + 256 + 4 * lane);
    const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(ltw + N + 256 + 4 * lane);
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int k = 0; k < 4; k++) ct_bfly(v[8 * g + k], v[8 * g + 4 + k], ww[g], pp[g], q, q2);
  }
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {  // t = 2: eight twiddles, one (rotated) b128 per half
    const int ph = 512 + 8 * lane + 4 * ((hh + (lane >> 3)) & 1);
    const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(ltw + ph);
    const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(ltw + N + ph);
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int k0 = 16 * hh + 4 * g;
      ct_bfly(v[k0], v[k0 + 2], ww[g], pp[g], q, q2);
      ct_bfly(v[k0 + 1], v[k0 + 3], ww[g], pp[g], q, q2);
    }
  }
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {  // t = 1: sixteen twiddles, four (rotated) b128
    const int ph = 1024 + 16 * lane + 4 * ((qq + (lane >> 2)) & 3);
    const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(ltw + ph);
    const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(ltw + N + ph);
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int g = 0; g < 4; g++) ct_bfly(v[8 * qq + 2 * g], v[8 * qq + 2 * g + 1], ww[g], pp[g], q, q2);
  }
#pragma unroll
  for (int k = 0; k < 32; k++) {  // ntt.rs:107-111
    u32 x = v[k];
    x -= (x >= q2 ? q2 : 0u);
    x -= (x >= q ? q : 0u);
    v[k] = x;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// inverse: v[k] = X[32 lane + k] (values < 2q)  ->  v[k] = x[64 k + lane], canonical
// itw: global inverse tables [w | w'] of this modulus (ntt_tables which = 2, 3); read per lane (two transforms per
// modulus and fold step only)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wntt_inv(u32 (&v)[32], int lane, u32* buf, const u32* __restrict__ itw, u32 q, u32 q2) {
  const cu32_t* sw = (const cu32_t*)itw;
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {  // t = 1
    const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(itw + 1024 + 16 * lane + 4 * qq);
    const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(itw + N + 1024 + 16 * lane + 4 * qq);
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int g = 0; g < 4; g++) gs_bfly(v[8 * qq + 2 * g], v[8 * qq + 2 * g + 1], ww[g], pp[g], q, q2);
  }
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {  // t = 2
    const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(itw + 512 + 8 * lane + 4 * hh);
    const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(itw + N + 512 + 8 * lane + 4 * hh);
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int k0 = 16 * hh + 4 * g;
      gs_bfly(v[k0], v[k0 + 2], ww[g], pp[g], q, q2);
      gs_bfly(v[k0 + 1], v[k0 + 3], ww[g], pp[g], q, q2);
    }
  }
  {  // t = 4
    const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(itw + 256 + 4 * lane);
    const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(itw + N + 256 + 4 * lane);
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int k = 0; k < 4; k++) gs_bfly(v[8 * g + k], v[8 * g + 4 + k], ww[g], pp[g], q, q2);
  }
  {  // t = 8
    const u32x2w_t w2 = *reinterpret_cast<const u32x2w_t*>(itw + 128 + 2 * lane);
    const u32x2w_t p2 = *reinterpret_cast<const u32x2w_t*>(itw + N + 128 + 2 * lane);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      gs_bfly(v[k], v[k + 8], w2.x, p2.x, q, q2);
      gs_bfly(v[16 + k], v[24 + k], w2.y, p2.y, q, q2);
    }
  }
  {  // t = 16
    const u32 w = itw[64 + lane], wp = itw[N + 64 + lane];
#pragma unroll
    for (int k = 0; k < 16; k++) gs_bfly(v[k], v[k + 16], w, wp, q, q2);
  }
#pragma unroll
  for (int g = 0; g < 8; g++) {
    u32x4w_t t4;
    t4.x = v[4 * g]; t4.y = v[4 * g + 1]; t4.z = v[4 * g + 2]; t4.w = v[4 * g + 3];
    *reinterpret_cast<u32x4w_t*>(buf + 36 * lane + 4 * g) = t4;
  }
  __builtin_amdgcn_wave_barrier();
  const int hl = wave_h(lane);
  const u32 lom = lane < 32 ? 0xffffffffu : 0u;
#pragma unroll
  for (int p = 0; p < 16; p++) {  // t = 32 in the swapped layout, then back to layout A
    u32 a = buf[144 * p + hl], b = buf[144 * p + 36 + hl];
    const u32 w = half_select(lom, sw[32 + 2 * p], sw[32 + 2 * p + 1]);
    const u32 wp = half_select(lom, sw[N + 32 + 2 * p], sw[N + 32 + 2 * p + 1]);
    gs_bfly(a, b, w, wp, q, q2);
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    v[2 * p] = r[0];
    v[2 * p + 1] = r[1];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int mm = 4; mm >= 0; mm--) {
    const int Tk = 16 >> mm;
#pragma unroll
    for (int k = 0; k < 32; k++) {
      if (((k / Tk) & 1) == 0) {
        const int ti = (1 << mm) + k / (2 * Tk);
        gs_bfly(v[k], v[k + Tk], sw[ti], sw[N + ti], q, q2);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 32; k++) {  // ntt.rs:253-256
    u32 x = v[k];
    x -= (x >= q2 ? q2 : 0u);
    x -= (x >= q ? q : 0u);
    v[k] = x;
  }
}

// word of coefficient n = 32 L + k of a polynomial stored in "wave layout": lane L reads its 32 coefficients as eight
// coalesced 16-byte vectors (vector g of all lanes is one contiguous KiB)
__host__ __device__ inline int wave_layout_word(int n) { return ((n & 31) >> 2) * 256 + (n >> 5) * 4 + (n & 3); }

}  // namespace spiral
