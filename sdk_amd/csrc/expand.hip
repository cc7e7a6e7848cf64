// One round of coefficient_expansion (server.rs:19-121) for one ciphertext per workgroup, in ONE launch:
//   ct      = v[i]                      (second half of the round: neg1[r] * v[i - num_in], server.rs:105-110)
//   ct_auto = automorph(from_ntt(ct), t)                                              (poly.rs:393-405)
//   v'[i]   = ct + W * to_ntt(G^-1(ct_auto row 0)) + [0 ; to_ntt(ct_auto row 1)]      (server.rs:89-102)
// The three-kernel form (k_ntt_inv -> k_ntt_fwd3 -> k_mac2) costs three dependent launches per round and the rounds
// are a chain: ten rounds of a C2 query take 0.7 ms although the arithmetic of the first six is a few microseconds.
// SPIRAL_EXPAND_FUSED=1 (with SPIRAL_EXPAND_SPLIT=1) selects this kernel instead; it is byte-identical but NOT faster
// (profiles/r02_expand_experiments.md): the 22 transforms of a ciphertext are ~35,000 wave instructions, 10 us of ONE
// CU's issue capacity, while the three launches spread them over 22 CUs -- a round costs 32 us here against ~28 us.
// Reads v from `src` and writes to `dst` (the caller ping-pongs two buffers): a round's second-half workgroup reads
// v[i - num_in] while the first-half workgroup of the same launch replaces v[i - num_in].
#include "device_common.hpp"
#include "kernels.hpp"

namespace spiral {

// A workgroup of SIXTEEN waves = four teams of 256 threads works on one ciphertext with the workgroup-cooperative transform
// (device_common.hpp); every team owns two LDS exchange buffers and all teams run their transforms in lockstep
// (ntt_*_block synchronise with __syncthreads, i.e. across the whole workgroup):
//   phase 1  team T inverse-transforms (row, modulus) = (T >> 1, T & 1); Garner + automorphism into LDS
//   phase 2  teams 0-1 work modulus 0, teams 2-3 modulus 1: in step s team (c, sub) transforms polynomial dg = 2 s + sub
//            (dg < t: digit dg of row 0; dg == t: residues of row 1) and multiply-accumulates it with W's two rows
//   phase 3  the two partial sums of a modulus are combined through LDS; + the input ciphertext; store
// (A first version on the wave-per-transform NTT of wave_ntt.hpp, 8 waves per ciphertext, took 50 us per round: that
// transform has a latency of 5.5 us.)
// ------------------------------------------------------------------------------------------------------------------
constexpr int EXT_TEAM_WORDS = 2 * LDS_WORDS;                         // two exchange buffers per team
constexpr int EXT_LDS_WORDS = 4 * EXT_TEAM_WORDS + 4 * N + 4 * N;     // + two raw rows (u64) + forward tables of both moduli

__global__ __launch_bounds__(1024) void k_expand_round_teams(DevTables T, ExpandDesc d) {
  __shared__ __attribute__((aligned(16))) u32 smem[EXT_LDS_WORDS];
  const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int tau = threadIdx.x & 255;
  u32* la = smem + team * EXT_TEAM_WORDS;
  u32* lb = la + LDS_WORDS;
  u64* raw = reinterpret_cast<u64*>(smem + 4 * EXT_TEAM_WORDS);   // [row][N] automorphed raw ciphertext
  u32* ltw = smem + 4 * EXT_TEAM_WORDS + 4 * N;                   // [modulus][w | w']: the transforms' twiddles come from
                                                                  // LDS (~100 cycles) instead of L2 (~700) four times per step
  u32* exch = la;                                                 // phase 1: modulus-1 residues of the team's row, in the
                                                                  // team's own buffers (free once the inverse is done)
  u32* red = smem + 4 * EXT_TEAM_WORDS;                           // phase 3, over the raw rows: [modulus][row][N]
  const int grp = (int)blockIdx.x < d.n[0] ? 0 : 1;
  const int e = (int)blockIdx.x - (grp ? d.n[0] : 0);
  const int ct = d.ct_idx[grp][e];
  const int t = d.t[grp], bits = d.bits[grp];
  const bool second = ct >= d.num_in;
  // ---- phase 1
  {
    const int row = team >> 1, c = team & 1;
    const ModConst m = T.c.mod[c];
    const u32* sp = d.src + (((size_t)(second ? ct - d.num_in : ct) * 2 + row) * 2 + c) * N + 8 * tau;
    u32 x[8];
    {
      const uint4 a = reinterpret_cast<const uint4*>(sp)[0], b = reinterpret_cast<const uint4*>(sp)[1];
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
    if (second) {  // v[ct] = neg1 * v[ct - num_in]; parked in the output slot (phase 3 adds it back in)
      const u32* sc = d.neg1 + (size_t)c * N + 8 * tau;
      u32* pk = d.dst + (((size_t)ct * 2 + row) * 2 + c) * N + 8 * tau;
#pragma unroll
      for (int k = 0; k < 8; k++) x[k] = reduce64((u64)x[k] * sc[k], m);
      reinterpret_cast<uint4*>(pk)[0] = make_uint4(x[0], x[1], x[2], x[3]);
      reinterpret_cast<uint4*>(pk)[1] = make_uint4(x[4], x[5], x[6], x[7]);
    }
    {  // forward tables of both moduli -> LDS (1024 threads, 8 words each)
      const int i8 = (int)threadIdx.x;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        ltw[i8 + 1024 * k] = T.tw[i8 + 1024 * k];                      // modulus 0: [w | w'] = tables 0, 1
        ltw[2 * N + i8 + 1024 * k] = T.tw[4 * N + i8 + 1024 * k];      // modulus 1: tables 4, 5
      }
    }
    const u32* iw = T.tw + ((size_t)c * 4 + 2) * N;
    ntt_inv_block(x, tau, la, lb, iw, iw + N, m.q, m.two_q);  // -> coefficient tau + 256 k
    __syncthreads();  // everybody is done with the exchange buffers
    if (c == 1) {     // hand the modulus-1 residues to the modulus-0 team of the same row (team - 1)
#pragma unroll
      for (int k = 0; k < 8; k++) exch[tau + 256 * k] = x[k];
    }
    __syncthreads();
    if (c == 0) {
      const u32* yb = smem + (team + 1) * EXT_TEAM_WORDS;
      const u32 q0 = T.c.mod[0].q, q1 = T.c.mod[1].q;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int z = tau + 256 * k;
        const u32 xx = x[k], y = yb[z];
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
  }
  // ---- phase 2
  const int c = team >> 1, sub = team & 1;
  const ModConst m = T.c.mod[c];
  const u32* fw = ltw + c * 2 * N;
  const u64 mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL);
  u64 acc0[8], acc1[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc0[k] = acc1[k] = 0;
  const int steps = (t + 2) / 2;  // polynomials 0 .. t over two teams
#pragma unroll 1
  for (int s = 0; s < steps; s++) {
    const int dg = 2 * s + sub;
    const bool active = dg <= t, unit = dg == t;
    const int sh = dg * bits;
    const u64* rp = raw + (unit ? N : 0) + tau;
    u32 v[8];
    if (!active) {
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = 0;
    } else if (unit || bits > 28) {  // to_ntt (residue of the whole word) / wide digits: reduce  (gadget.rs:48-53)
      const u64 dmask = unit ? ~0ULL : (sh >= 64 ? 0ULL : mask);
      const int shc = unit ? 0 : (sh & 63);
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = reduce64((rp[256 * k] >> shc) & dmask, m);
    } else {
      const u64 dmask = sh >= 64 ? 0ULL : mask;
      const int shc = sh & 63;
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (u32)((rp[256 * k] >> shc) & dmask);
    }
    // W's two rows for this polynomial: fetched before the transform, used after it
    const int dgc = active && !unit ? dg : 0;
    const uint4* a0 = reinterpret_cast<const uint4*>(d.W[grp] + ((size_t)dgc * 2 + c) * N + 8 * tau);
    const uint4* a1 = reinterpret_cast<const uint4*>(d.W[grp] + ((size_t)(t + dgc) * 2 + c) * N + 8 * tau);
    const uint4 p0 = a0[0], p1 = a0[1], r0 = a1[0], r1 = a1[1];
    ntt_fwd_block(v, tau, la, lb, fw, fw + N, m.q, m.two_q);  // -> coefficient 8 tau + k, canonical
    {
      u32* tmp = la;
      la = lb;
      lb = tmp;
    }
    if (active && !unit) {
      acc0[0] += (u64)p0.x * v[0]; acc0[1] += (u64)p0.y * v[1]; acc0[2] += (u64)p0.z * v[2]; acc0[3] += (u64)p0.w * v[3];
      acc0[4] += (u64)p1.x * v[4]; acc0[5] += (u64)p1.y * v[5]; acc0[6] += (u64)p1.z * v[6]; acc0[7] += (u64)p1.w * v[7];
      acc1[0] += (u64)r0.x * v[0]; acc1[1] += (u64)r0.y * v[1]; acc1[2] += (u64)r0.z * v[2]; acc1[3] += (u64)r0.w * v[3];
      acc1[4] += (u64)r1.x * v[4]; acc1[5] += (u64)r1.y * v[5]; acc1[6] += (u64)r1.z * v[6]; acc1[7] += (u64)r1.w * v[7];
    } else if (unit) {
#pragma unroll
      for (int k = 0; k < 8; k++) acc1[k] += v[k];
    }
  }
  // ---- phase 3 (the raw rows are dead: the last transform's first barrier came after everybody's reads)
  u32 r0[8], r1[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r0[k] = reduce64(acc0[k], m);
    r1[k] = reduce64(acc1[k], m);
  }
  __syncthreads();
  if (sub == 1) {
    uint4* o0 = reinterpret_cast<uint4*>(red + (c * 2 + 0) * N + 8 * tau);
    uint4* o1 = reinterpret_cast<uint4*>(red + (c * 2 + 1) * N + 8 * tau);
    o0[0] = make_uint4(r0[0], r0[1], r0[2], r0[3]); o0[1] = make_uint4(r0[4], r0[5], r0[6], r0[7]);
    o1[0] = make_uint4(r1[0], r1[1], r1[2], r1[3]); o1[1] = make_uint4(r1[4], r1[5], r1[6], r1[7]);
  }
  __syncthreads();
  if (sub == 0) {
#pragma unroll
    for (int row = 0; row < 2; row++) {
      const uint4* pr = reinterpret_cast<const uint4*>(red + (c * 2 + row) * N + 8 * tau);
      const size_t slot = (((size_t)ct * 2 + row) * 2 + c) * N + 8 * tau;
      const uint4* ad = reinterpret_cast<const uint4*>((second ? d.dst : d.src) + slot);
      const uint4 pa = pr[0], pb = pr[1], aa = ad[0], ab = ad[1];
      const u32 part[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
      const u32 add[8] = {aa.x, aa.y, aa.z, aa.w, ab.x, ab.y, ab.z, ab.w};
      u32 o[8];
#pragma unroll
      for (int k = 0; k < 8; k++) o[k] = add_mod(add_mod(row == 0 ? r0[k] : r1[k], part[k], m.q), add[k], m.q);
      uint4* op = reinterpret_cast<uint4*>(d.dst + slot);
      op[0] = make_uint4(o[0], o[1], o[2], o[3]);
      op[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
  }
}

void launch_expand_round(const DevTables& T, const ExpandDesc& d, hipStream_t s) {
  const int blocks = std::max(d.n[0], 0) + std::max(d.n[1], 0);
  if (blocks <= 0) return;
  ExpandDesc dd = d;
  dd.n[0] = std::max(d.n[0], 0);
  dd.n[1] = std::max(d.n[1], 0);
  hipLaunchKernelGGL(k_expand_round_teams, dim3(blocks), dim3(1024), 0, s, T, dd);
  launched(PATH_EXPAND_FUSED, "k_expand_round_teams");
}

}  // namespace spiral
