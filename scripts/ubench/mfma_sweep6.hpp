// Six-wave variant of k_sweep_mfma_batch (sdk_amd/csrc/sweep_mfma.hpp), kept here as the record of a negative result:
// 3 waves per SIMD (168 VGPRs, 8 spilled) instead of 2 is SLOWER, 3.07 ms against 2.75 ms per plane-pass at C2
// (profiles/r03_mfma_batch_sweep.md), because the pass is bound by issue slots, not by memory latency.  Not in the library.
#pragma once
#include "sweep_mfma.hpp"

namespace spiral {
// Six-wave form: a workgroup = 6 waves on ONE (plane, z) with `cpw` chunks = 4 cpw quarter-chunk tasks (chunk, slot group g);
// wave w takes tasks w, w + 6, w + 12, ...  Two workgroups per CU (the 64 KiB table each) = 12 waves = 3 per SIMD instead
// of 2: one more wave to issue from while the others wait for memory.  Needs <= 168 VGPRs (512 / 3).  Addresses are a
// wave-uniform base (task, step) + a per-lane 32-bit offset.  (cpw * 4) tasks; 16 chunks -> 11 / 11 / 11 / 11 / 10 / 10.
template <int NB, int DIAG = 0>
__global__ __launch_bounds__(384, 3) void k_sweep_mfma_batch6(DevTables T, SweepMfmaDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rq[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave of the workgroup: first task
  const int kb = lane >> 4, mp = lane & 15;
  const int chunks = d.num_per >> 7;
  const int wgs_per_zp = chunks / d.cpw;
  const int zp = blockIdx.x / wgs_per_zp;  // plane * N + z
  const int chunk0 = (blockIdx.x - zp * wgs_per_zp) * d.cpw;
  const int z = zp & (N - 1), plane = zp >> POLY_LEN_LOG2;
  const int steps = d.nj >> 4, npairs = d.nj >> 1;
  {  // this z's digit table -> LDS (steps * 2 KiB): up to 16 loads per thread in flight, then the LDS writes
    const mf_u32x4_t* src = reinterpret_cast<const mf_u32x4_t*>(d.rq) + (size_t)z * steps * 128;
    mf_u32x4_t* dst = reinterpret_cast<mf_u32x4_t*>(smem_rq);
    const int n16 = steps * 128;
    int i0 = 0;
    for (; i0 + 8 * 384 <= n16; i0 += 8 * 384) {  // no bounds checks inside: a guarded load costs a branch + vmcnt(0)
      mf_u32x4_t t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = src[i0 + k * 384 + threadIdx.x];
#pragma unroll
      for (int k = 0; k < 8; k++) dst[i0 + k * 384 + threadIdx.x] = t[k];
    }
    for (int i = i0 + threadIdx.x; i < n16; i += 384) dst[i] = src[i];
    __syncthreads();
  }
  const mf_u32x4_t* rql = reinterpret_cast<const mf_u32x4_t*>(smem_rq) + lane;
  // dword offset of this lane inside a step of slot group 0: row pairs 2 kb (+1), slot mp; group g adds 64 / 48 dwords
  const u32 off4 = (u32)((2 * kb) * 448 + mp * 4), off3 = (u32)((2 * kb) * 448 + 256 + mp * 3);
  const u32* const dbw = reinterpret_cast<const u32*>(d.db);
  const size_t zrow = packed_unit_offset((size_t)zp, 0, chunk0, npairs, chunks);  // first unit of this workgroup's chunks
  const int ntasks = 4 * d.cpw;
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  const u32 M = 0x0FFFFFFFu;
  // The QUERY digits are the MFMA's A operand (rows of D = query columns n = 2 b + r), the database digits its B operand
  // (columns of D = the wave's 16 slots): lane (kb, mp) then holds, in register i, query column 4 kb + i of slot mp, so
  // the 16 lanes of a group store 128 contiguous bytes of ONE output array per instruction (the other way round every
  // store instruction wrote 16-byte pieces of 16 arrays: +38 % on the pass).  Queries b = 2 kb and 2 kb + 1 of this lane
  // group: their output arrays are picked from the kernel-argument pointers with scalar loads + v_cndmask -- a VECTOR
  // load d.out[b] in the epilogue would need s_waitcnt vmcnt(0), i.e. drain the prefetch ring at every chunk end.
  u32* out_b0 = d.out[0];
  u32* out_b1 = d.out[1];
#pragma unroll
  for (int k2 = 1; k2 < SWEEP_BATCH_MAX / 2; k2++) {
    out_b0 = kb == k2 ? d.out[2 * k2] : out_b0;
    out_b1 = kb == k2 ? d.out[2 * k2 + 1] : out_b1;
  }
  const mf_u32x4_t off0 = reinterpret_cast<const mf_u32x4_t*>(d.rq_off)[(size_t)z * 8 + kb];
  const mf_u32x4_t off1 = reinterpret_cast<const mf_u32x4_t*>(d.rq_off)[(size_t)z * 8 + 4 + kb];
  mf_u32x4_t va[NB][2];
  mf_u32x3_t vb[NB][2];

// the four loads of a step are pinned in program order (sched_barrier): the s_waitcnt pass counts loads in flight in
// issue order, and only with the same order in the prologue and in every unrolled step does it wait for the oldest
// buffer alone (vmcnt(4 (NB - 1))) instead of draining the whole ring at the loop head
// TASK / STEP wave-uniform: base = the chunk's first unit + step * 8 row pairs (+ the slot group's 64 / 48 dwords)
#define SPM_LOAD(BUF, TASK, STEP)                                                                    \
  {                                                                                                  \
    const size_t sb = zrow + ((size_t)((TASK) >> 2) * npairs + (size_t)(STEP) * 8) * 448;            \
    const u32* q4 = dbw + sb + 64 * ((TASK) & 3) + off4;                                             \
    const u32* q3 = dbw + sb + 48 * ((TASK) & 3) + off3;                                             \
    va[BUF][0] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x4_t*>(q4));               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    vb[BUF][0] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x3_t*>(q3));               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    va[BUF][1] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x4_t*>(q4 + 448));         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    vb[BUF][1] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x3_t*>(q3 + 448));         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  }
// one step (16 rows x 32 columns x 16 query columns x 2 moduli): 28 MFMAs
#define SPM_STEP(BUF, SL)                                                                            \
  {                                                                                                  \
    u32 f[2][8];                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 2; u++) {                                                  \
      const u32 d0 = va[BUF][u].x, d1 = va[BUF][u].y, d2 = va[BUF][u].z, d3 = va[BUF][u].w;          \
      const u32 d4 = vb[BUF][u].x, d5 = vb[BUF][u].y, d6 = vb[BUF][u].z;                             \
      f[u][0] = d0 & M;                                                                              \
      f[u][1] = __builtin_amdgcn_alignbit(d1, d0, 28) & M;                                           \
      f[u][2] = __builtin_amdgcn_alignbit(d2, d1, 24) & M;                                           \
      f[u][3] = __builtin_amdgcn_alignbit(d3, d2, 20) & M;                                           \
      f[u][4] = __builtin_amdgcn_alignbit(d4, d3, 16) & M;                                           \
      f[u][5] = __builtin_amdgcn_alignbit(d5, d4, 12) & M;                                           \
      f[u][6] = __builtin_amdgcn_alignbit(d6, d5, 8) & M;                                            \
      f[u][7] = d6 >> 4;                                                                             \
    }                                                                                                \
    v4i_t A[2][2];                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 2; e++) _Pragma("unroll") for (int c = 0; c < 2; c++) {    \
      if (DIAG == 3) {                                                                               \
        A[e][c] = __builtin_bit_cast(v4i_t, (e ^ c) ? va[BUF][0] : va[BUF][1]);                      \
      } else {                                                                                       \
        A[e][c][0] = (int)offset_digits(f[0][2 * e + c]);                                            \
        A[e][c][1] = (int)offset_digits(f[0][4 + 2 * e + c]);                                        \
        A[e][c][2] = (int)offset_digits(f[1][2 * e + c]);                                            \
        A[e][c][3] = (int)offset_digits(f[1][4 + 2 * e + c]);                                        \
      }                                                                                              \
    }                                                                                                \
    _Pragma("unroll") for (int c = 0; c < 2; c++) {                                                  \
      const mf_u32x4_t R = rql[((SL) * 2 + c) * 64];                                                 \
      _Pragma("unroll") for (int s = 0; s < 7; s++) {                                                \
        const u32 sh = (u32)(8 * (s < 3 ? 3 - s : s - 3));                                           \
        const mf_u32x4_t Bs = DIAG == 3 ? R : (s < 3 ? R >> sh : R << sh);                           \
        const v4i_t Bi = __builtin_bit_cast(v4i_t, Bs);                                              \
        if (DIAG == 2) {                                                                             \
          acc[0][c][s] ^= A[0][c] + Bi;                                                              \
          acc[1][c][s] ^= A[1][c] - Bi;                                                              \
        } else {                                                                                     \
          acc[0][c][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Bi, A[0][c], acc[0][c][s], 0, 0, 0);  \
          acc[1][c][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Bi, A[1][c], acc[1][c][s], 0, 0, 0);  \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
  }

  if (wv >= ntasks) return;
#pragma unroll
  for (int k = 0; k < NB - 1; k++) SPM_LOAD(k, wv, k)
  // every step issues the load NB - 1 steps ahead; past the end of a task the prefetch continues in the wave's NEXT task
  // (6 tasks on), and behind its last task it re-reads that task's last step (cache hits) instead of leaving the range
  for (int task = wv; task < ntasks; task += 6) {
    const int g = task & 3, ch = task >> 2;
    const bool has_next = task + 6 < ntasks;
    v4i_t acc[2][2][7];  // [tile e][crt][shift]
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int s = 0; s < 7; s++) acc[e][c][s] = v4i_t{0, 0, 0, 0};
    for (int s0 = 0; s0 < steps; s0 += NB) {
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const int a = s0 + k + NB - 1;
        const bool over = a >= steps;
        const int atask = over && has_next ? task + 6 : task;
        const int astep = over ? (has_next ? a - steps : steps - 1) : a;
        if (DIAG != 1) SPM_LOAD((k + NB - 1) % NB, atask, astep)
        __builtin_amdgcn_sched_barrier(0);
        SPM_STEP(k, s0 + k)
      }
    }
    // chunk done: recombine the digit sums, reduce, store: register i = query column 4 kb + i (b = 2 kb + i / 2,
    // r = i % 2), lane mp = slot 16 g + mp = columns 2 (16 g + mp) + e
    const size_t rcw = (size_t)N * d.num_per;
    const size_t col = DIAG == 5 ? (size_t)(32 * g + 2 * mp)
                                 : (size_t)z * d.num_per + (size_t)(chunk0 + ch) * 128 + 32 * g + 2 * mp;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (2 * kb + (i >> 1) < d.batch) {
        u32* ob = ((i >> 1) ? out_b1 : out_b0) + ((size_t)plane * 4 + (i & 1) * 2) * rcw + col;
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const ModConst mc = c ? m1 : m0;
          const u32 v0 = combine_digit_sums(acc[0][c][0][i], acc[0][c][1][i], acc[0][c][2][i], acc[0][c][3][i],
                                            acc[0][c][4][i], acc[0][c][5][i], acc[0][c][6][i], mc, d.c4[c], d.c5[c], d.c6[c], c ? off1[i] : off0[i]);
          const u32 v1 = combine_digit_sums(acc[1][c][0][i], acc[1][c][1][i], acc[1][c][2][i], acc[1][c][3][i],
                                            acc[1][c][4][i], acc[1][c][5][i], acc[1][c][6][i], mc, d.c4[c], d.c5[c], d.c6[c], c ? off1[i] : off0[i]);
          if (DIAG == 4 && (v0 ^ v1) != 0xDEADBEEFu) continue;  // (practically) no stores
          if (DIAG == 7) {
            u32* o7 = d.out[0] + ((((size_t)zp * chunks + chunk0 + ch) * 4 + g) * 8 + (i * 2 + c)) * 128 + 2 * lane;
            *reinterpret_cast<uint2*>(o7) = make_uint2(v0, v1);
          } else if (DIAG == 6) {
            *reinterpret_cast<uint2*>(ob + (size_t)c * rcw) = make_uint2(v0, v1);
          } else {
            // non-temporal: HBM writes mixed into the read stream are expensive on this part (537 MB per plane cost
            // 0.27 ms as streaming stores, 0.37 ms as plain ones, in a pure read + write kernel: scripts/ubench/rw_mix.hip)
            __builtin_nontemporal_store(mf_u32x2_t{v0, v1}, reinterpret_cast<mf_u32x2_t*>(ob + (size_t)c * rcw));
          }
        }
      }
    }
  }
#undef SPM_LOAD
#undef SPM_STEP
}


}  // namespace spiral
