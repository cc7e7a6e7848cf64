// Batched database sweep on the matrix cores: up to 8 queries per pass over the PACKED database (BASELINE configs[4],
// the per-request loop lib/server/src/bin/server.rs:152-158 served by one pass; arithmetic = multiply_reg_by_database,
// server.rs:155-221, per query).
//
// With B >= 4 queries per pass the first-dimension multiply is, for every (plane, z, modulus), an exact integer GEMM
// [num_per x dim0] . [dim0 x 2B] whose VALU form (k_sweep_packed_batch, 32 v_mad_u64_u32 per database word at B = 8)
// is issue-bound at 1.9x the HBM time of the pass.  Here the 28-bit residues are split into four signed bytes and
// multiplied on v_mfma_i32_16x16x64_i8.  The QUERY side y uses carry-propagated signed base-256 digits (y + 0x808080 ^
// 0x808080: bytes 0-2 in [-128, 127], byte 3 in [0, 16], sum digit_b 256^b = y; done once per pass by k_query_digits); the
// DATABASE side x, converted inside the sweep for every word, uses OFFSET digits x ^ 0x80808080 = bytes u_a - 128 (one
// instruction, no carries), which leaves a term that does not depend on the database column:
//   sum_j x_j y_j = sum_{s=0..6} 256^s D_s + 128 (1 + 2^8 + 2^16 + 2^24) sum_j y_j,
//   D_s = sum_j sum_{a+b=s} (u_a(j) - 128) ydigit_b(j),   |D_s| <= 4 nj 2^14
// exact in i32 for nj <= 2^15 rows; the second term, mod q, comes from k_query_offset_terms per (z, query column, modulus)
// and is added when the digit sums are recombined.  One MFMA covers K = 64 = (16 rows) x (4 database digits); its B operand for shift
// s carries the query digit y_{s-a} in byte a of each row's dword (zero where s - a is outside 0..3), i.e. the
// byte-reversed digit dword shifted by whole bytes -- so the seven operands of a row block are six VALU shifts of one
// LDS read.  Seven i32 accumulators per (output, modulus) are recombined and reduced mod q once per 128-column chunk.
//
// Work split: a workgroup = 4 waves walks `cpw` consecutive 128-column chunks of one (plane, z); wave g owns the 16
// lane slots 16g .. 16g+15 of every unit = 32 columns = two 16-column MFMA tiles (even / odd column of each slot).
// Lane (kb = lane / 16, m = lane % 16) loads slot 16g + m of the row pairs 8s + 2kb, 8s + 2kb + 1 of step s (16 rows):
// exactly the four rows x four digits its k-block of the A operand needs, so no cross-lane exchange at all.  Per wave
// and load instruction that is 4 x 256 B (dwordx4 piece) or 4 x 192 B (dwordx3 piece); the four waves of the
// workgroup together read every unit contiguously.
#pragma once
#include "device_common.hpp"

namespace spiral {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef u32 mf_u32x4_t __attribute__((ext_vector_type(4)));
typedef u32 mf_u32x3_t __attribute__((ext_vector_type(3), aligned(4)));
typedef u32 mf_u32x2_t __attribute__((ext_vector_type(2)));

constexpr u32 DIGIT_BIAS = 0x00808080u;
// four signed base-256 digits of x < 2^28 in the four bytes of the result (sum digit_i 256^i = x)
__host__ __device__ __forceinline__ u32 signed_digits(u32 x) { return (x + DIGIT_BIAS) ^ DIGIT_BIAS; }
// four offset digits: byte a = (byte a of x) - 128 as a signed byte (sum digit_a 256^a = x - DIGIT_OFFSET_SUM)
constexpr u32 DIGIT_OFFSET = 0x80808080u;  // = 128 (1 + 2^8 + 2^16 + 2^24), also the value the offsets add up to
__host__ __device__ __forceinline__ u32 offset_digits(u32 x) { return x ^ DIGIT_OFFSET; }

constexpr int SWEEP_MFMA_MAX = 16;  // two query tiles of 8 queries (16 query columns = one MFMA M dimension) per pass
struct SweepMfmaDesc {
  const u64* db;                  // PACKED database: plane 0 of the launch
  const u32* rq;                  // query digit table [tile][N][nj / 16][2][64][4] (k_query_digits, one call per tile of 8 queries)
  const u32* rq_off;              // offset terms [tile][N][2][16]: DIGIT_OFFSET * sum_j y_j mod q per (z, crt, query column)
  u32* out[SWEEP_MFMA_MAX];       // per query: sweep-native [plane][r][crt][z][ii]
  int batch;                      // 1 .. 8 QT (unused query columns of the tables are zero)
  int planes, num_per, nj;        // nj % 16 == 0, nj <= 512 (LDS-staged table), num_per % 128 == 0
  int cpw;                        // chunks per workgroup, divides num_per / 128
  u32 c4[2], c5[2], c6[2];        // 2^32, 2^40, 2^48 mod q_crt
};

// Query digit table for one group of queries: dword i of entry (z, step, crt, lane = kb * 16 + n) = byte-reversed signed
// digits of residue crt of qv[b][z][j0 + 16 step + 4 kb + i][r], n = 2 b + r (zero for b >= batch).
struct QueryDigitsDesc {
  const u64* qv[SWEEP_BATCH_MAX];  // reoriented queries [N][dim0][2]
  u32* rq;
  int batch, dim0, j0, nj;
};
// the same table in the order of the digit-planar pass (sweep_planar.hpp / sweep_planar.hip)
void launch_query_digits_planar(const QueryDigitsDesc& q, size_t entries, hipStream_t s);
// ... both tiles' tables and offset terms of a 9 .. 16-query group, one launch each (sweep_planar.hip)
void launch_query_tables_planar2(const DevTables& T, const u64* const* qv, int batch, int dim0, int j0, int nj, u32* rq, hipStream_t s);
// (internal linkage: this header is included by sweep.hip and, through sweep_planar.hpp, by sweep_planar.hip)
static __global__ __launch_bounds__(256) void k_query_digits(QueryDigitsDesc d) {
  const int steps = d.nj >> 4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)N * steps * 128) return;
  const int lane = (int)(idx & 63), crt = (int)((idx >> 6) & 1);
  const size_t zs = idx >> 7;
  const int step = (int)(zs % steps), z = (int)(zs / steps);
  const int n = lane & 15, kb = lane >> 4, b = n >> 1, r = n & 1;
  mf_u32x4_t o = {0u, 0u, 0u, 0u};
  if (b < d.batch) {
    const u64* q = d.qv[b] + ((size_t)z * d.dim0 + d.j0 + 16 * step + 4 * kb) * 2 + r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u64 w = q[2 * i];
      const u32 x = crt ? (u32)(w >> 32) : (u32)w;
      o[i] = __builtin_bswap32(signed_digits(x));
    }
  }
  reinterpret_cast<mf_u32x4_t*>(d.rq)[idx] = o;
}

// Offset terms of the database-side digit form: off[z][crt][n] = DIGIT_OFFSET * (sum_j y_j) mod q over the nj rows of query
// column n = 2 b + r (zero for b >= batch).  One workgroup per z: 8 threads per (crt, n) add up strided rows, then a
// shuffle reduction; the row sum (< 2^37) and the product with DIGIT_OFFSET mod q (< 2^28) are exact in 64 bits.
static __global__ __launch_bounds__(256) void k_query_offset_terms(DevTables T, QueryDigitsDesc d, u32* off) {
  const int z = blockIdx.x, t = threadIdx.x;
  const int combo = t >> 3, part = t & 7;      // combo = crt * 16 + n
  const int crt = combo >> 4, n = combo & 15, b = n >> 1, r = n & 1;
  u64 sum = 0;
  if (b < d.batch) {
    const u64* q = d.qv[b] + ((size_t)z * d.dim0 + d.j0) * 2 + r;
    for (int j = part; j < d.nj; j += 8) {
      const u64 w = q[2 * (size_t)j];
      sum += crt ? (u32)(w >> 32) : (u32)w;
    }
  }
#pragma unroll
  for (int sft = 1; sft < 8; sft <<= 1) sum += __shfl_xor(sum, sft, 8);
  if (part == 0) {
    const ModConst m = T.c.mod[crt];
    const u32 sy = reduce64(sum, m);
    off[(size_t)z * 32 + combo] = reduce64((u64)sy * (u64)(DIGIT_OFFSET % m.q), m);
  }
}

// sum_s 256^s D[s] mod q for |D[s]| < 2^26 (exact: the sum is the non-negative integer sum_j x_j y_j).  The three high
// digit sums are multiplied by 256^s mod q (c4 = 2^32, c5 = 2^40, c6 = 2^48 mod q, each < 2^28: |products| < 2^54) and added
// to the low part as signed 64-bit integers; q 2^29 makes the total positive, one Barrett fold finishes.
__device__ __forceinline__ u32 combine_digit_sums(int d0, int d1, int d2, int d3, int d4, int d5, int d6, const ModConst m,
                                                  u32 c4, u32 c5, u32 c6, u32 off) {
  long long v = (long long)d0 + ((long long)d1 << 8) + ((long long)d2 << 16) + ((long long)d3 << 24);  // |.| < 2^51
  v += (long long)d4 * (long long)c4 + (long long)d5 * (long long)c5 + (long long)d6 * (long long)c6;   // |.| < 2^56
  return reduce64((u64)(v + ((long long)m.q << 29) + (long long)off), m);  // off < q: the offset term (k_query_offset_terms)
}

// DIAG (microbenchmark only, scripts/ubench/mfma_sweep.hip; 0 in the library): 1 = no database loads after the
// prologue (compute only), 2 = MFMAs replaced by one XOR each (loads + VALU only), 3 = raw dwords fed to the MFMAs (no
// digit extraction, no operand shifts: loads + MFMA only), 4 = no output stores, 5 = every workgroup stores to the first
// z-row and chunk (writes stay in L2), 6 = plain instead of non-temporal stores (results valid), 7 = the 4 KiB a wave produces per chunk
// stored as one contiguous block of out[0] (16 KiB per workgroup and chunk, 256 KiB per z-row: layout experiment).  Results are meaningless for DIAG 1-5.
// QT = query tiles per pass (r04): with QT = 2 sixteen queries share ONE pass -- the database words are loaded and split
// into digits once (60 of the 139 VALU instructions of a 16-row step) and multiplied against two digit tables; the
// accumulators (2 x 112 registers) then need the whole register file: one workgroup per CU (MINWG = 1), accumulators in
// AGPRs, and a deeper load ring (NB) to keep the HBM pipe full with one wave per SIMD.
template <int NB, int MINWG, int DIAG = 0, int QT = 1>
__global__ __launch_bounds__(256, MINWG) void k_sweep_mfma_batch(DevTables T, SweepMfmaDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rq[];
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // slot group of this wave
  const int kb = lane >> 4, mp = lane & 15;
  const int chunks = d.num_per >> 7;
  const int wgs_per_zp = chunks / d.cpw;
  const int zp = blockIdx.x / wgs_per_zp;  // plane * N + z
  const int chunk0 = (blockIdx.x - zp * wgs_per_zp) * d.cpw;
  const int z = zp & (N - 1), plane = zp >> POLY_LEN_LOG2;
  const int steps = d.nj >> 4, npairs = d.nj >> 1;
  const int n16 = steps * 128;   // 16-byte entries of one tile's z-row
  {  // this z's digit table(s) -> LDS (steps * 2 KiB per tile): up to 16 loads per thread in flight, then the LDS writes
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
      const mf_u32x4_t* src = reinterpret_cast<const mf_u32x4_t*>(d.rq) + ((size_t)qt * N + z) * n16;
      mf_u32x4_t* dst = reinterpret_cast<mf_u32x4_t*>(smem_rq) + (size_t)qt * n16;
      int i0 = 0;
      for (; i0 + 16 * 256 <= n16; i0 += 16 * 256) {  // no bounds checks inside: a guarded load costs a branch + vmcnt(0)
        mf_u32x4_t t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = src[i0 + k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 16; k++) dst[i0 + k * 256 + threadIdx.x] = t[k];
      }
      for (int i = i0 + threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
  }
  const mf_u32x4_t* rql = reinterpret_cast<const mf_u32x4_t*>(smem_rq) + lane;
  const u32* pu = reinterpret_cast<const u32*>(d.db) + packed_unit_offset((size_t)zp, 0, chunk0, npairs, chunks) +
                  (size_t)(2 * kb) * 448;
  const u32* p4 = pu + (16 * g + mp) * 4;
  const u32* p3 = pu + 256 + (16 * g + mp) * 3;
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  const u32 M = 0x0FFFFFFFu;
  const int total = d.cpw * steps;
  // The QUERY digits are the MFMA's A operand (rows of D = query columns n = 2 b + r), the database digits its B operand
  // (columns of D = the wave's 16 slots): lane (kb, mp) then holds, in register i, query column 4 kb + i of slot mp, so
  // the 16 lanes of a group store 128 contiguous bytes of ONE output array per instruction (the other way round every
  // store instruction wrote 16-byte pieces of 16 arrays: +38 % on the pass).  Queries b = 2 kb and 2 kb + 1 of this lane
  // group: their output arrays are picked from the kernel-argument pointers with scalar loads + v_cndmask -- a VECTOR
  // load d.out[b] in the epilogue would need s_waitcnt vmcnt(0), i.e. drain the prefetch ring at every chunk end.
  u32* out_b0[QT];
  u32* out_b1[QT];
  mf_u32x4_t off0[QT], off1[QT];
#pragma unroll
  for (int qt = 0; qt < QT; qt++) {
    out_b0[qt] = d.out[8 * qt];
    out_b1[qt] = d.out[8 * qt + 1];
#pragma unroll
    for (int k2 = 1; k2 < 4; k2++) {
      out_b0[qt] = kb == k2 ? d.out[8 * qt + 2 * k2] : out_b0[qt];
      out_b1[qt] = kb == k2 ? d.out[8 * qt + 2 * k2 + 1] : out_b1[qt];
    }
    // offset terms of this lane's four query columns n = 4 kb + i of the tile, both moduli (the same for every chunk)
    if (QT == 1) {
      off0[qt] = reinterpret_cast<const mf_u32x4_t*>(d.rq_off)[((size_t)qt * N + z) * 8 + kb];
      off1[qt] = reinterpret_cast<const mf_u32x4_t*>(d.rq_off)[((size_t)qt * N + z) * 8 + 4 + kb];
    }
  }
  // two query tiles: the 16 offset registers would sit in scratch through the chunk loop; they wait in LDS behind the digit
  // tables instead (256 bytes; the launch adds them to the dynamic size) and are read per chunk epilogue (lgkmcnt, not vmcnt)
  mf_u32x4_t* const smem_off = reinterpret_cast<mf_u32x4_t*>(smem_rq) + (size_t)QT * n16;
  if (QT == 2) {
    if (threadIdx.x < 16)
      smem_off[threadIdx.x] = reinterpret_cast<const mf_u32x4_t*>(d.rq_off)[((size_t)(threadIdx.x >> 3) * N + z) * 8 + (threadIdx.x & 7)];
    __syncthreads();
  }
  mf_u32x4_t va[NB][2];
  mf_u32x3_t vb[NB][2];

// the four loads of a step are pinned in program order (sched_barrier): the s_waitcnt pass counts loads in flight in
// issue order, and only with the same order in the prologue and in every unrolled step does it wait for the oldest
// buffer alone (vmcnt(4 (NB - 1))) instead of draining the whole ring at the loop head
#define SPM_LOAD(BUF, S)                                                                             \
  {                                                                                                  \
    const u32* q4 = p4 + (size_t)(S) * 3584;                                                         \
    const u32* q3 = p3 + (size_t)(S) * 3584;                                                         \
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
    _Pragma("unroll") for (int qt = 0; qt < QT; qt++) _Pragma("unroll") for (int c = 0; c < 2; c++) { \
      const mf_u32x4_t R = rql[(size_t)qt * n16 + ((SL) * 2 + c) * 64];                              \
      _Pragma("unroll") for (int s = 0; s < 7; s++) {                                                \
        const u32 sh = (u32)(8 * (s < 3 ? 3 - s : s - 3));                                           \
        const mf_u32x4_t Bs = DIAG == 3 ? R : (s < 3 ? R >> sh : R << sh);                           \
        const v4i_t Bi = __builtin_bit_cast(v4i_t, Bs);                                              \
        if (DIAG == 2) {                                                                             \
          acc[qt][0][c][s] ^= A[0][c] + Bi;                                                          \
          acc[qt][1][c][s] ^= A[1][c] - Bi;                                                          \
        } else {                                                                                     \
          acc[qt][0][c][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Bi, A[0][c], acc[qt][0][c][s], 0, 0, 0);  \
          acc[qt][1][c][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Bi, A[1][c], acc[qt][1][c][s], 0, 0, 0);  \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
  }

#pragma unroll
  for (int k = 0; k < NB - 1; k++) SPM_LOAD(k, k)
  // Control flow kept to two plain nested loops (accumulators live inside the chunk loop, no second code path for the
  // tail): every step issues the load NB - 1 steps ahead; the prefetch runs on into the workgroup's next chunk (the
  // stream is contiguous) and, for the last NB - 1 steps of the workgroup, re-reads its own last step (cache hits)
  // instead of running past the range.
  for (int ch = 0; ch < d.cpw; ch++) {
    v4i_t acc[QT][2][2][7];  // [query tile][column tile e][crt][shift]
#pragma unroll
    for (int qt = 0; qt < QT; qt++)
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int s = 0; s < 7; s++) acc[qt][e][c][s] = v4i_t{0, 0, 0, 0};
    for (int s0 = 0; s0 < steps; s0 += NB) {
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const int ahead = min(ch * steps + s0 + k + NB - 1, total - 1);
        if (DIAG != 1) SPM_LOAD((k + NB - 1) % NB, ahead)
        __builtin_amdgcn_sched_barrier(0);
        SPM_STEP(k, s0 + k)
      }
    }
    // chunk done: recombine the digit sums, reduce, store: register i = query column 4 kb + i (b = 2 kb + i / 2,
    // r = i % 2), lane mp = slot 16 g + mp = columns 2 (16 g + mp) + e
    const size_t rcw = (size_t)N * d.num_per;
    // (two query tiles: the epilogue derives its lane-dependent values from ONE opaque copy of the lane number, so that the
    // chunk loop keeps no query pair, column or address of it alive -- each was a scratch slot)
    int le = lane;
    if (QT == 2) asm volatile("" : "+v"(le));
    const int kbe = le >> 4, mpe = le & 15;
    const size_t col = DIAG == 5 ? (size_t)(32 * g + 2 * mpe)
                                 : (size_t)z * d.num_per + (size_t)(chunk0 + ch) * 128 + 32 * g + 2 * mpe;
#pragma unroll
    for (int qt = 0; qt < QT; qt++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (8 * qt + 2 * kbe + (i >> 1) < d.batch) {
        u32* sel = (i >> 1) ? out_b1[qt] : out_b0[qt];
        if (QT == 2) {
          // two query tiles: every register counts (224 accumulators + the load ring).  Left alone, the compiler forms the 24
          // loop-invariant store addresses of a chunk's epilogue before the chunk loop and parks them in scratch (49 dwords, r04-r05);
          // from the opaque lane copy above they are per-chunk work: a dozen selects from the kernel arguments
          sel = d.out[8 * qt + (i >> 1)];
#pragma unroll
          for (int k2 = 1; k2 < 4; k2++) sel = kbe == k2 ? d.out[8 * qt + 2 * k2 + (i >> 1)] : sel;
        }
        u32* ob = sel + ((size_t)plane * 4 + (i & 1) * 2) * rcw + col;
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const ModConst mc = c ? m1 : m0;
          u32 oc = QT == 2 ? smem_off[qt * 8 + c * 4 + kbe][i] : (c ? off1[qt][i] : off0[qt][i]);
          if (QT == 2) asm volatile("" : "+v"(oc));   // (as above: keeps (q << 29) + offset, 16 64-bit values, out of the chunk loop's live set)
          const u32 v0 = combine_digit_sums(acc[qt][0][c][0][i], acc[qt][0][c][1][i], acc[qt][0][c][2][i], acc[qt][0][c][3][i],
                                            acc[qt][0][c][4][i], acc[qt][0][c][5][i], acc[qt][0][c][6][i], mc, d.c4[c], d.c5[c], d.c6[c], oc);
          const u32 v1 = combine_digit_sums(acc[qt][1][c][0][i], acc[qt][1][c][1][i], acc[qt][1][c][2][i], acc[qt][1][c][3][i],
                                            acc[qt][1][c][4][i], acc[qt][1][c][5][i], acc[qt][1][c][6][i], mc, d.c4[c], d.c5[c], d.c6[c], oc);
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
