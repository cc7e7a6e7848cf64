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
//             bank-conflict-free.  It runs in two halves of 1024 elements (register pairs 0-7 feed lanes 0-31, pairs
//             8-15 lanes 32-63), so the buffer is 4.5 KiB per wave
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
constexpr int WBUF_WORDS = 32 * 36;  // padded transpose buffer of one wave: HALF a polynomial at a time (4608 bytes)

__device__ __forceinline__ int wave_h(int lane) { return lane < 32 ? lane : lane + 40; }

// (wtw_phys -- where entry idx of the tables lives in the LDS copy -- is in kernels.hpp: the host builds the image with it)
// stage the forward tables of one modulus into ltw (2N words) -- whole workgroup, 256 threads; img = wave_fwd_image(tw, c),
// already negated and rotated: four 16-byte copies per thread
__device__ __forceinline__ void wtw_stage(u32* ltw, const u32* __restrict__ img, int tau) {
#pragma unroll
  for (int k = 0; k < 4; k++)
    reinterpret_cast<u32x4w_t*>(ltw)[tau + 256 * k] = reinterpret_cast<const u32x4w_t*>(img)[tau + 256 * k];
}

// (SP_SB and ct_bfly_batch -- the lazy five-instruction Cooley-Tukey butterfly, issued phase by phase -- live in device_common.hpp
// since round 5: the cooperative forward transform uses them too.)
// butterflies (v[ia], v[ib]) with twiddles (w, wp), ia / ib / twiddle index given by the functors, in batches of 8
#define SP_BFLY_STAGE(COUNT, IA, IB, W, WP, CORR)                                                \
  _Pragma("unroll") for (int b0_ = 0; b0_ < (COUNT); b0_ += 8) {                                 \
    u32 xa_[8], ya_[8], wa_[8], pa_[8];                                                          \
    _Pragma("unroll") for (int b_ = 0; b_ < 8; b_++) {                                           \
      const int j_ = b0_ + b_;                                                                   \
      xa_[b_] = v[IA(j_)]; ya_[b_] = v[IB(j_)]; wa_[b_] = W(j_); pa_[b_] = WP(j_);               \
    }                                                                                            \
    ct_bfly_batch<8, CORR>(xa_, ya_, wa_, pa_, q, q2);                                           \
    _Pragma("unroll") for (int b_ = 0; b_ < 8; b_++) {                                           \
      const int j_ = b0_ + b_;                                                                   \
      v[IA(j_)] = xa_[b_]; v[IB(j_)] = ya_[b_];                                                  \
    }                                                                                            \
  }

struct WaveScalarTw {  // table entries 0..15 of [w | w'] (stages t = 1024 .. 128): wave-uniform, live in SGPRs
  u32 w[16], wp[16];
};
__device__ __forceinline__ void wntt_scalar_tw(WaveScalarTw& s, const u32* tw) {
  const cu32_t* sw = (const cu32_t*)tw;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    s.w[i] = 0u - sw[i];      // negated (ct_bfly_batch)
    s.wp[i] = sw[N + i];
  }
}
struct WaveNoHooks {
  __device__ __forceinline__ void before_t4() {}
  __device__ __forceinline__ void before_t1() {}
  __device__ __forceinline__ void after_quarter(int, u32 (&)[32]) {}
};

// ------------------------------------------------------------------------------------------------------------------
// forward: v[k] = x[64 k + lane] (values < 2q)  ->  v[k] = X[32 lane + k] (the reference's output order); canonical if
// CANON, else only < 12q < 2^32 (enough for a multiply-accumulate that is reduced afterwards).
// Bounds of the lazy reduction (ct_bfly_batch): < 2q in; stages 1-6 add 2q each: < 14q; stage 7 (t = 16) reduces x by
// 8q: < 10q out; stages 8, 9: < 14q; stage 10 (t = 2) reduces again: < 10q; stage 11: < 12q.  x + 2q < 16q < 2^32.
// tw: global tables [w | w'] of this modulus (uniform pointer), s: their first 16 entries (wntt_scalar_tw, issued by the
// caller as early as it can); ltw: LDS copy (wtw_stage); buf: this wave's transpose buffer
// ------------------------------------------------------------------------------------------------------------------
template <bool CANON, class Hooks>
__device__ __forceinline__ void wntt_fwd(u32 (&v)[32], int lane, u32* buf, const u32* tw, const WaveScalarTw& s,
                                         const u32* ltw, u32 q, u32 q2, Hooks& hk) {
  const cu32_t* sw = (const cu32_t*)tw;
  u32 w5[16], p5[16];  // stage t = 64: entries 16..31, in flight during the first four stages
#pragma unroll
  for (int i = 0; i < 16; i++) {
    w5[i] = 0u - sw[16 + i];  // negated (ct_bfly_batch)
    p5[i] = sw[N + 16 + i];
  }
  SP_SB();
#pragma unroll
  for (int mm = 0; mm < 4; mm++) {  // butterfly j of the stage: registers k, k + Tk with k = (j / Tk) * 2 Tk + j % Tk
    const int Tk = 16 >> mm;
#define SP_IA(j) (((j) / Tk) * 2 * Tk + (j) % Tk)
#define SP_IB(j) (SP_IA(j) + Tk)
#define SP_W(j) s.w[(1 << mm) + (j) / Tk]
#define SP_WP(j) s.wp[(1 << mm) + (j) / Tk]
    SP_BFLY_STAGE(16, SP_IA, SP_IB, SP_W, SP_WP, false)
#undef SP_IA
#undef SP_IB
#undef SP_W
#undef SP_WP
  }
  // stage t = 32: lanes 0-31 use entry 32 + 2p, lanes 32-63 entry 33 + 2p -- read per lane from the LDS copy, four
  // register pairs ahead
  const u32* lw = ltw + 32 + (lane >> 5);
  u32 gw[5][4], gp[5][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    gw[0][i] = lw[2 * i];
    gp[0][i] = lw[N + 2 * i];
  }
  SP_SB();
#define SP_IA(j) (2 * (j))
#define SP_IB(j) (2 * (j) + 1)
#define SP_W(j) w5[j]
#define SP_WP(j) p5[j]
  SP_BFLY_STAGE(16, SP_IA, SP_IB, SP_W, SP_WP, false)
#undef SP_IA
#undef SP_IB
#undef SP_W
#undef SP_WP
  const int hl = wave_h(lane);
  const bool lo = lane < 32;
  const u32* rd = buf + 36 * (lane & 31);
  u32 up[16];        // registers 16..31 of the transposed layout while the second half is still being written
  u32 w16 = 0, p16 = 0;
#pragma unroll
  for (int grp = 0; grp < 4; grp++) {
    if (grp < 3) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        gw[grp + 1][i] = lw[2 * (4 * (grp + 1) + i)];
        gp[grp + 1][i] = lw[N + 2 * (4 * (grp + 1) + i)];
      }
    } else {
      w16 = ltw[64 + lane];
      p16 = ltw[N + 64 + lane];
    }
    SP_SB();
    {
      u32 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int p = 4 * grp + i;
        const auto r = __builtin_amdgcn_permlane32_swap(v[2 * p], v[2 * p + 1], false, false);
        a[i] = r[0];  // lanes 0-31: (x, y) of register 2p; lanes 32-63: (x, y) of register 2p+1
        b[i] = r[1];
      }
      ct_bfly_batch<4, false>(a, b, gw[grp], gp[grp], q, q2);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int pp = (4 * grp + i) & 7;
        buf[144 * pp + hl] = a[i];       // element 128pp + (lane < 32 ? lane : lane + 32) of this half, at n + 4 (n >> 5)
        buf[144 * pp + 36 + hl] = b[i];  // the element 32 above it
      }
    }
    if (grp & 1) {
      // half (grp >> 1) is complete: lanes 0-31 (first half) / 32-63 (second) pick up their 32 values.  The reads of
      // the first half travel while the second half's butterflies run (LDS operations of a wave execute in order)
      __builtin_amdgcn_wave_barrier();
      if (lo == (grp == 1)) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
          const u32x4w_t t4 = *reinterpret_cast<const u32x4w_t*>(rd + 4 * g);
          if (g < 4) {
            v[4 * g] = t4.x; v[4 * g + 1] = t4.y; v[4 * g + 2] = t4.z; v[4 * g + 3] = t4.w;
          } else {
            up[4 * g - 16] = t4.x; up[4 * g - 15] = t4.y; up[4 * g - 14] = t4.z; up[4 * g - 13] = t4.w;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) v[16 + k] = up[k];
  // layout B.  t = 16: one twiddle per lane (already here); t = 8: two; t = 4: four; t = 2: eight; t = 1: sixteen
  const u32x2w_t w8 = *reinterpret_cast<const u32x2w_t*>(ltw + 128 + 2 * lane);
  const u32x2w_t p8 = *reinterpret_cast<const u32x2w_t*>(ltw + N + 128 + 2 * lane);
  SP_SB();
#define SP_IA(j) (j)
#define SP_IB(j) ((j) + 16)
#define SP_W(j) w16
#define SP_WP(j) p16
  SP_BFLY_STAGE(16, SP_IA, SP_IB, SP_W, SP_WP, true)   // stage 7: < 14q in, < 10q out
#undef SP_IA
#undef SP_IB
#undef SP_W
#undef SP_WP
  const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(ltw + 256 + 4 * lane);
  const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(ltw + N + 256 + 4 * lane);
  hk.before_t4();
  SP_SB();
#define SP_IA(j) (((j) >> 3) * 16 + ((j) & 7))
#define SP_IB(j) (SP_IA(j) + 8)
#define SP_W(j) ((j) < 8 ? w8.x : w8.y)
#define SP_WP(j) ((j) < 8 ? p8.x : p8.y)
  SP_BFLY_STAGE(16, SP_IA, SP_IB, SP_W, SP_WP, false)
#undef SP_IA
#undef SP_IB
#undef SP_W
#undef SP_WP
  u32x4w_t w2[2], p2[2];
  {
    const int ph = 512 + 8 * lane + 4 * ((lane >> 3) & 1);
    w2[0] = *reinterpret_cast<const u32x4w_t*>(ltw + ph);
    p2[0] = *reinterpret_cast<const u32x4w_t*>(ltw + N + ph);
  }
  SP_SB();
  {
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#define SP_IA(j) (((j) >> 2) * 8 + ((j) & 3))
#define SP_IB(j) (SP_IA(j) + 4)
#define SP_W(j) ww[(j) >> 2]
#define SP_WP(j) pp[(j) >> 2]
    SP_BFLY_STAGE(16, SP_IA, SP_IB, SP_W, SP_WP, false)
#undef SP_IA
#undef SP_IB
#undef SP_W
#undef SP_WP
  }
  u32x4w_t w1[4], p1[4];
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {  // t = 2: eight twiddles, one (rotated) b128 per half
    if (hh == 0) {
      const int ph = 512 + 8 * lane + 4 * ((1 + (lane >> 3)) & 1);
      w2[1] = *reinterpret_cast<const u32x4w_t*>(ltw + ph);
      p2[1] = *reinterpret_cast<const u32x4w_t*>(ltw + N + ph);
    } else {
      const int ph = 1024 + 16 * lane + 4 * ((lane >> 2) & 3);
      w1[0] = *reinterpret_cast<const u32x4w_t*>(ltw + ph);
      p1[0] = *reinterpret_cast<const u32x4w_t*>(ltw + N + ph);
      hk.before_t1();
    }
    SP_SB();
    const u32 ww[4] = {w2[hh].x, w2[hh].y, w2[hh].z, w2[hh].w}, pp[4] = {p2[hh].x, p2[hh].y, p2[hh].z, p2[hh].w};
#define SP_IA(j) (16 * hh + ((j) >> 1) * 4 + ((j) & 1))
#define SP_IB(j) (SP_IA(j) + 2)
#define SP_W(j) ww[(j) >> 1]
#define SP_WP(j) pp[(j) >> 1]
    SP_BFLY_STAGE(8, SP_IA, SP_IB, SP_W, SP_WP, true)   // stage 10: < 14q in, < 10q out
#undef SP_IA
#undef SP_IB
#undef SP_W
#undef SP_WP
  }
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {  // t = 1: sixteen twiddles, four (rotated) b128
    if (qq < 3) {
      const int ph = 1024 + 16 * lane + 4 * ((qq + 1 + (lane >> 2)) & 3);
      w1[qq + 1] = *reinterpret_cast<const u32x4w_t*>(ltw + ph);
      p1[qq + 1] = *reinterpret_cast<const u32x4w_t*>(ltw + N + ph);
    }
    SP_SB();
    const u32 ww[4] = {w1[qq].x, w1[qq].y, w1[qq].z, w1[qq].w}, pp[4] = {p1[qq].x, p1[qq].y, p1[qq].z, p1[qq].w};
    {
      u32 a[4], b[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        a[g] = v[8 * qq + 2 * g];
        b[g] = v[8 * qq + 2 * g + 1];
      }
      ct_bfly_batch<4, false>(a, b, ww, pp, q, q2);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        v[8 * qq + 2 * g] = a[g];
        v[8 * qq + 2 * g + 1] = b[g];
      }
    }
    if (CANON) {
#pragma unroll
      for (int k = 8 * qq; k < 8 * qq + 8; k++) {  // ntt.rs:107-111, from < 12q
        u32 x = v[k];
        x -= (x >= 4 * q2 ? 4 * q2 : 0u);
        x -= (x >= 2 * q2 ? 2 * q2 : 0u);
        x -= (x >= q2 ? q2 : 0u);
        x -= (x >= q ? q : 0u);
        v[k] = x;
      }
    }
    hk.after_quarter(qq, v);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// inverse: v[k] = X[32 lane + k] (values < 2q)  ->  v[k] = x[64 k + lane], canonical
// itw: global inverse tables [w | w'] of this modulus (inv_tables: unhalved, N^-1 in entries 0 and 1).  The per-lane twiddles are all fetched
// up front (94 registers: the caller's accumulators are gone by the time it transforms back)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wntt_inv(u32 (&v)[32], int lane, u32* buf, const u32* __restrict__ itw, u32 q, u32 q2) {
  const cu32_t* sw = (const cu32_t*)itw;
  u32x4w_t w1[4], p1[4], w2[2], p2[2];
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    w1[qq] = *reinterpret_cast<const u32x4w_t*>(itw + 1024 + 16 * lane + 4 * qq);
    p1[qq] = *reinterpret_cast<const u32x4w_t*>(itw + N + 1024 + 16 * lane + 4 * qq);
  }
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {
    w2[hh] = *reinterpret_cast<const u32x4w_t*>(itw + 512 + 8 * lane + 4 * hh);
    p2[hh] = *reinterpret_cast<const u32x4w_t*>(itw + N + 512 + 8 * lane + 4 * hh);
  }
  const u32x4w_t w4 = *reinterpret_cast<const u32x4w_t*>(itw + 256 + 4 * lane);
  const u32x4w_t p4 = *reinterpret_cast<const u32x4w_t*>(itw + N + 256 + 4 * lane);
  const u32x2w_t w8 = *reinterpret_cast<const u32x2w_t*>(itw + 128 + 2 * lane);
  const u32x2w_t p8 = *reinterpret_cast<const u32x2w_t*>(itw + N + 128 + 2 * lane);
  const u32 w16 = itw[64 + lane], p16 = itw[N + 64 + lane];
  u32 w32[16], p32[16];
#pragma unroll
  for (int p = 0; p < 16; p++) {
    w32[p] = itw[32 + 2 * p + (lane >> 5)];
    p32[p] = itw[N + 32 + 2 * p + (lane >> 5)];
  }
  SP_SB();
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {  // t = 1
    const u32 ww[4] = {w1[qq].x, w1[qq].y, w1[qq].z, w1[qq].w}, pp[4] = {p1[qq].x, p1[qq].y, p1[qq].z, p1[qq].w};
#pragma unroll
    for (int g = 0; g < 4; g++) gs_bfly(v[8 * qq + 2 * g], v[8 * qq + 2 * g + 1], ww[g], pp[g], q, q2);
  }
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {  // t = 2
    const u32 ww[4] = {w2[hh].x, w2[hh].y, w2[hh].z, w2[hh].w}, pp[4] = {p2[hh].x, p2[hh].y, p2[hh].z, p2[hh].w};
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int k0 = 16 * hh + 4 * g;
      gs_bfly(v[k0], v[k0 + 2], ww[g], pp[g], q, q2);
      gs_bfly(v[k0 + 1], v[k0 + 3], ww[g], pp[g], q, q2);
    }
  }
  {  // t = 4
    const u32 ww[4] = {w4.x, w4.y, w4.z, w4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int k = 0; k < 4; k++) gs_bfly(v[8 * g + k], v[8 * g + 4 + k], ww[g], pp[g], q, q2);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {  // t = 8
    gs_bfly(v[k], v[k + 8], w8.x, p8.x, q, q2);
    gs_bfly(v[16 + k], v[24 + k], w8.y, p8.y, q, q2);
  }
#pragma unroll
  for (int k = 0; k < 16; k++) gs_bfly(v[k], v[k + 16], w16, p16, q, q2);  // t = 16
  // back through the half-size buffer: lanes 0-31 hand over their 32 values, everyone computes register pairs 0-7
  // (t = 32 in the swapped layout), then the same for lanes 32-63 and pairs 8-15
  const int hl = wave_h(lane);
  const bool lo = lane < 32;
  u32* wr = buf + 36 * (lane & 31);
  u32 dn[16];  // registers 0..15 of layout A while lanes 32-63 still hold their layout-B values in v
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if (lo == (half == 0)) {
#pragma unroll
      for (int g = 0; g < 8; g++) {
        u32x4w_t t4;
        t4.x = v[4 * g]; t4.y = v[4 * g + 1]; t4.z = v[4 * g + 2]; t4.w = v[4 * g + 3];
        *reinterpret_cast<u32x4w_t*>(wr + 4 * g) = t4;
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pp = 0; pp < 8; pp++) {
      const int p = 8 * half + pp;
      u32 a = buf[144 * pp + hl], b = buf[144 * pp + 36 + hl];
      gs_bfly(a, b, w32[p], p32[p], q, q2);
      const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
      if (half == 0) {
        dn[2 * pp] = r[0];
        dn[2 * pp + 1] = r[1];
      } else {
        v[2 * p] = r[0];
        v[2 * p + 1] = r[1];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = dn[k];
#pragma unroll
  for (int mm = 4; mm >= 0; mm--) {
    const int Tk = 16 >> mm;
#pragma unroll
    for (int k = 0; k < 32; k++) {
      if (((k / Tk) & 1) == 0) {
        const int ti = (1 << mm) + k / (2 * Tk);
        if (mm == 0)  // distance N/2: the last stage carries N^-1
          gs_bfly_last(v[k], v[k + Tk], sw[0], sw[N], sw[1], sw[N + 1], q, q2);
        else
          gs_bfly(v[k], v[k + Tk], sw[ti], sw[N + ti], q, q2);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 32; k++) v[k] -= (v[k] >= q ? q : 0u);  // ntt.rs:253-256, from < 2q
}

// (wave_layout_word -- the word of coefficient n = 32 L + k of a polynomial stored in "wave layout", where lane L reads its
// 32 coefficients as eight coalesced 16-byte vectors -- lives in bodies.hpp beside the re-layout kernel's body)

}  // namespace spiral
