// Batched database sweep on the matrix cores over a DIGIT-PLANAR resident database (r05; VERDICT r04 item 6: "a resident
// format that already IS the MFMA operand").
//
// k_sweep_mfma_batch (sweep_mfma.hpp) reads the 7-byte PACKED words and pays, per 16-row step and wave, 60 vector instructions
// to cut them into byte digits and 48 per query tile to shift the query digit dwords into the seven anti-diagonal operands --
// with two query tiles it is bound by that instruction stream (3.75 ms per C2 plane with no loads at all), not by HBM.
// Here the database is stored the way v_mfma_i32_16x16x64_i8 wants its B operand:
//
//   K = 64 ROWS of ONE digit: operand (tile e, modulus c, digit a) of a 64-row block = for each of the tile's 16 columns the 64
//   bytes  ((x_c(row, column) >> 8 a) & 0xff) ^ 0x80  (offset digits, as in sweep_mfma.hpp), laid out
//   [lane = 16 kb + n][16 bytes = rows 16 kb .. 16 kb + 15 of column n]: ONE coalesced 1-KiB load per operand, no instruction
//   between the load and the MFMA.
//
// The query side is planar too (k_query_digits_planar: [tile][z][block][c][digit b][lane = 16 kb + m][16 rows], signed digits),
// staged per z-row in LDS as before (64 KiB per tile at 512 rows).  The anti-diagonal sums come from WHICH accumulator an MFMA
// adds into, not from shifted operands:
//     D_s += sum over the block's 64 rows of u_a y_b      for every (a, b) with a + b = s      -> acc[s] = mfma(Y_b, U_a, acc[s])
// 16 MFMAs per (64 rows, tile, modulus, query tile) instead of 28, and no vector ALU work in the loop besides addresses.  The
// seven digit sums, the offset term and the epilogue are those of sweep_mfma.hpp (combine_digit_sums, k_query_offset_terms).
// Cost: 8 bytes per database word resident instead of 7 (64 GiB at C2 instead of 56).
//
// Work split as in k_sweep_mfma_batch: a workgroup = 4 waves walks `cpw` 128-column chunks of one (plane, z); wave g owns columns
// 32 g .. 32 g + 31 of a chunk as two tiles e (column 32 g + 2 n + e: the 16 lanes of a group store 128 contiguous bytes).
// Resident order: [plane][z][chunk][g][c][block][e][a][1 KiB]: a wave walks, per chunk, ONE MODULUS AT A TIME (a contiguous 64-KiB
// pass of 16 units at 512 rows) -- the accumulators of the inner loop are then those of one modulus (28 x 4 registers with two
// query tiles) and stay in the vector registers; with both moduli live (56 x 4) the register allocator parks them in AGPRs and
// moves 3.3 registers per MFMA back and forth (848 v_accvgpr moves per 256 MFMAs in the first version of this kernel).
#pragma once
#include "sweep_mfma.hpp"

namespace spiral {

constexpr int PLANAR_BLOCK_ROWS = 64;
constexpr size_t PLANAR_BLOCK_BYTES = 16 * 1024;  // per wave and 64-row block: 2 tiles x 2 moduli x 4 digits x 1 KiB

// byte offset of operand (e, c, a) of (zp = plane * N + z, chunk, wave g, block) in the planar database
__host__ __device__ __forceinline__ size_t planar_operand_offset(size_t zp, int chunk, int g, int block, int e, int c, int a,
                                                                 int chunks, int blocks) {
  return (((((zp * (size_t)chunks + (size_t)chunk) * 4 + (size_t)g) * 2 + (size_t)c) * (size_t)blocks + (size_t)block) * 8 +
          (size_t)(e * 4 + a)) * 1024;
}

struct SweepPlanarDesc {
  const unsigned char* db;        // planar database: plane 0 of the launch
  const unsigned char* rq;        // query digit planes [tile][N][blocks][2][4][64][16] (k_query_digits_planar)
  const u32* rq_off;              // offset terms [tile][N][2][16] (k_query_offset_terms)
  u32* out[SWEEP_MFMA_MAX];       // per query: sweep-native [plane][r][crt][z][ii]
  int batch;
  int planes, num_per, nj;        // nj % 64 == 0, nj <= 512, num_per % 128 == 0
  int cpw;
  u32 c4[2], c5[2], c6[2];        // 2^32, 2^40, 2^48 mod q_crt
};

// query digit planes of one tile of <= 8 queries: byte t of entry (z, block, c, b, lane = 16 kb + m) = signed digit b of residue c
// of qv[m / 2][z][j0 + 64 block + 16 kb + t][m % 2]   (zero for queries >= batch)
static __global__ __launch_bounds__(256) void k_query_digits_planar(QueryDigitsDesc d) {
  const int blocks = d.nj >> 6;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // one 16-byte entry per thread
  if (idx >= (size_t)N * blocks * 8 * 64) return;
  const int lane = (int)(idx & 63), b = (int)((idx >> 6) & 3), c = (int)((idx >> 8) & 1);
  const size_t zb = idx >> 9;
  const int block = (int)(zb % blocks), z = (int)(zb / blocks);
  const int m = lane & 15, kb = lane >> 4, qb = m >> 1, r = m & 1;
  mf_u32x4_t o = {0u, 0u, 0u, 0u};
  if (qb < d.batch) {
    const u64* q = d.qv[qb] + ((size_t)z * d.dim0 + d.j0 + 64 * block + 16 * kb) * 2 + r;
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const u64 w = q[2 * t];
      const u32 x = c ? (u32)(w >> 32) : (u32)w;
      const u32 dg = (signed_digits(x) >> (8 * b)) & 0xffu;
      o[t >> 2] |= dg << (8 * (t & 3));
    }
  }
  reinterpret_cast<mf_u32x4_t*>(d.rq)[idx] = o;
}

// r06: both tiles' tables and offset terms as ONE launch each (grid.y = tile) -- two 64-MiB tables built one after the other by
// launches that reach 1.4 TB/s each were 0.35 ms of every 16-query step
struct QueryDigits2Desc {
  const u64* qv[SWEEP_GROUP_MAX];  // reoriented queries [N][dim0][2]
  u32* rq;                          // [tile][...] tables, then the offset terms [tile][N][32]
  int batch, dim0, j0, nj;
};
// One thread per (z, 64-row block, lane = 16 kb + m) writes that lane's entry of ALL EIGHT operands (modulus c, digit b): the 16
// query words it needs are read once instead of once per operand, as eight 16-byte loads per lane PAIR (m even / odd = the two
// rows r of one query share every 16-byte (row j, r = 0 | 1) pair of the reoriented query: the even lane fetches rows 0..7, the odd
// one rows 8..15, and they swap halves) -- the first form read 8 bytes of every 16, sixteen times per thread, for each of the
// eight operands: 280 us for the two tables of a 16-query group.
static __global__ __launch_bounds__(256) void k_query_digits_planar2(QueryDigits2Desc d) {
  const int blocks = d.nj >> 6;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)N * blocks * 64) return;      // (whole waves: 64 lanes per (z, block))
  const int tile = blockIdx.y;
  const int lane = (int)(t & 63);
  const size_t zb = t >> 6;
  const int block = (int)(zb % blocks), z = (int)(zb / blocks);
  const int m = lane & 15, kb = lane >> 4, qb = 8 * tile + (m >> 1), r = m & 1;
  const size_t entries = (size_t)N * blocks * 8 * 64;
  mf_u32x4_t* out = reinterpret_cast<mf_u32x4_t*>(d.rq) + (size_t)tile * entries + zb * 512 + lane;   // + (c * 4 + b) * 64
  const bool active = qb < d.batch;      // (query columns past the batch: zero entries; every lane still takes part in the swaps)
  u64 w[16];
  {
    // rows 16 kb .. 16 kb + 15 of query qb at z, both r: 16 x 16 bytes; this lane fetches 8 of them (r = 0: the first 8 rows)
    const ulonglong2* q2 = reinterpret_cast<const ulonglong2*>(d.qv[active ? qb : 0] + ((size_t)z * d.dim0 + d.j0 + 64 * block + 16 * kb) * 2) + 8 * r;
    ulonglong2 mine[8];
#pragma unroll
    for (int i = 0; i < 8; i++) mine[i] = active ? q2[i] : ulonglong2{0ull, 0ull};
#pragma unroll
    for (int i = 0; i < 8; i++) {
      // what the partner needs of my rows is its r's half; what I need of the partner's rows is my r's half
      const u64 give = r == 0 ? mine[i].y : mine[i].x;
      const u64 got = __shfl_xor(give, 1, 64);
      const u64 keep = r == 0 ? mine[i].x : mine[i].y;
      w[(r == 0 ? 0 : 8) + i] = keep;     // my own rows
      w[(r == 0 ? 8 : 0) + i] = got;      // the partner's rows, my r
    }
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    u32 sd[16];
#pragma unroll
    for (int i = 0; i < 16; i++) sd[i] = active ? signed_digits(c ? (u32)(w[i] >> 32) : (u32)w[i]) : 0u;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      mf_u32x4_t o = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 16; i++) o[i >> 2] |= ((sd[i] >> (8 * b)) & 0xffu) << (8 * (i & 3));
      out[(size_t)(c * 4 + b) * 64] = o;
    }
  }
}
// (k_query_offset_terms of sweep_mfma.hpp with the tile as grid.y)
static __global__ __launch_bounds__(256) void k_query_offset_terms2(DevTables T, QueryDigits2Desc d, u32* off) {
  const int z = blockIdx.x, t = threadIdx.x, tile = blockIdx.y;
  const int combo = t >> 3, part = t & 7;      // combo = crt * 16 + n
  const int crt = combo >> 4, n = combo & 15, b = 8 * tile + (n >> 1), r = n & 1;
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
    off[((size_t)tile * N + z) * 32 + combo] = reduce64((u64)sy * (u64)(DIGIT_OFFSET % m.q), m);
  }
}

// WAVES = 4: one wave per SIMD and workgroup; WAVES = 8: the workgroup's chunks are split between two sets of four waves that
// share the z-row's query planes in LDS -- two waves per SIMD (they cover each other's waits) where two four-wave workgroups
// would need the 64 KiB per query tile twice.
template <int NBUF, int QT, int DIAG = 0, int MINWG = 1, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, MINWG) void k_sweep_planar(DevTables T, SweepPlanarDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_pl[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave & 3, half = wave >> 2;
  const int kb = lane >> 4, mp = lane & 15;
  const int chunks = d.num_per >> 7;
  const int wgs_per_zp = chunks / d.cpw;
  const int zp = blockIdx.x / wgs_per_zp;
  const int cpw_w = d.cpw / (WAVES / 4);                        // chunks this wave walks
  const int chunk0 = (blockIdx.x - zp * wgs_per_zp) * d.cpw + half * cpw_w;
  const int z = zp & (N - 1), plane = zp >> POLY_LEN_LOG2;
  const int blocks = d.nj >> 6;
  const int n16 = blocks * 8 * 64;   // 16-byte entries of one tile's z-row
  {
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
      const mf_u32x4_t* src = reinterpret_cast<const mf_u32x4_t*>(d.rq) + ((size_t)qt * N + z) * n16;
      mf_u32x4_t* dst = reinterpret_cast<mf_u32x4_t*>(smem_pl) + (size_t)qt * n16;
      for (int i = threadIdx.x; i < n16; i += 64 * WAVES) dst[i] = src[i];
    }
    __syncthreads();
  }
  const v4i_t* ql = reinterpret_cast<const v4i_t*>(smem_pl) + lane;
  // this wave's stream: per chunk two passes (modulus 0, modulus 1) of 2 * blocks units of 4 KiB, contiguous; the next chunk of
  // the workgroup is 4 waves' worth further on
  const unsigned char* base = d.db + planar_operand_offset((size_t)zp, chunk0, g, 0, 0, 0, 0, chunks, blocks) + (size_t)lane * 16;
  const int upp = 2 * blocks;                                        // units per pass
  const size_t pass_bytes = (size_t)upp * 4096;
  const size_t chunk_stride = (size_t)4 * 2 * pass_bytes;
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
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
    off0[qt] = reinterpret_cast<const mf_u32x4_t*>(d.rq_off)[((size_t)qt * N + z) * 8 + kb];
    off1[qt] = reinterpret_cast<const mf_u32x4_t*>(d.rq_off)[((size_t)qt * N + z) * 8 + 4 + kb];
  }
  // Load ring at the granularity of ONE unit = the four digit operands of (64 rows x 16 columns, one modulus) = 4 KiB per wave:
  // NBUF units in registers, the load NBUF - 1 units ahead issued as soon as a unit's registers are free -- the memory pipe
  // always holds (NBUF - 1) x 4 KiB per wave.  NBUF is even and divides 2 * blocks.  Unit index S counts over (chunk, modulus, unit).
  v4i_t ring[NBUF][4];
  const int total = cpw_w * 2 * upp;
#define SPL_LOAD(BUF, S)                                                                                       \
  {                                                                                                            \
    const int s_ = (S);                                                                                        \
    const int pass_ = s_ / upp;                                                                                \
    const unsigned char* p_ = base + (size_t)(pass_ >> 1) * chunk_stride + (size_t)(pass_ & 1) * pass_bytes +  \
                              (size_t)(s_ - pass_ * upp) * 4096;                                               \
    _Pragma("unroll") for (int o_ = 0; o_ < 4; o_++) {                                                         \
      ring[BUF][o_] = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p_ + (size_t)o_ * 1024));      \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  }
#pragma unroll
  for (int k = 0; k < NBUF - 1; k++) SPL_LOAD(k, min(k, total - 1))
  const size_t rcw = (size_t)N * d.num_per;
  // the query operand is fetched one group of four MFMAs ahead (left to the compiler every ds_read sat right in front of its
  // first MFMA with an s_waitcnt lgkmcnt(0): the LDS latency was exposed once per four MFMAs, the matrix pipe half idle)
  v4i_t Ycur = ql[0];   // (pass 0, block 0, modulus 0, digit 0, tile 0)
  for (int pass = 0; pass < 2 * cpw_w; pass++) {
    const int ch = pass >> 1, c = pass & 1;
    v4i_t acc[QT][2][7];  // [query tile][column tile e][digit sum s] of modulus c
#pragma unroll
    for (int qt = 0; qt < QT; qt++)
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int sd = 0; sd < 7; sd++) acc[qt][e][sd] = v4i_t{0, 0, 0, 0};
    for (int u0 = 0; u0 < upp; u0 += NBUF) {
#pragma unroll
      for (int k = 0; k < NBUF; k++) {
        const int cur = pass * upp + u0 + k;
        if (DIAG != 1) SPL_LOAD((k + NBUF - 1) % NBUF, min(cur + NBUF - 1, total - 1))
        __builtin_amdgcn_sched_barrier(0);
        const int blk = (u0 + k) >> 1;
        const int e = k & 1;   // unit order inside a pass: [block][e]; NBUF is even
        // where the NEXT unit's first operand lives: same pass -> block (u0 + k + 1) / 2; last unit of a pass -> block 0 of the
        // other modulus (after the workgroup's last pass: any valid entry, it is never used)
        const bool last_unit = u0 + k + 1 == upp;
        const int nblk = last_unit ? 0 : (u0 + k + 1) >> 1, nc = last_unit ? (c ^ 1) : c;
#pragma unroll
        for (int gi = 0; gi < 4 * QT; gi++) {
          const int b = gi / QT, qt = gi % QT;
          const int nb = (gi + 1) / QT, nqt = (gi + 1) % QT;
          const v4i_t Ynext = gi + 1 < 4 * QT ? ql[(((size_t)nqt * blocks + blk) * 8 + c * 4 + nb) * 64]
                                              : ql[(((size_t)0 * blocks + nblk) * 8 + nc * 4 + 0) * 64];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int a = 0; a < 4; a++)
            acc[qt][e][a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ycur, ring[k][a], acc[qt][e][a + b], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          Ycur = Ynext;
        }
      }
    }
    // modulus c of this chunk is done: recombine the digit sums, reduce, store (register i = query column 4 kb + i, lane mp =
    // columns 32 g + 2 mp + e)
    const size_t col = (size_t)z * d.num_per + (size_t)(chunk0 + ch) * 128 + 32 * g + 2 * mp;
    const ModConst mc = c ? m1 : m0;
#pragma unroll
    for (int qt = 0; qt < QT; qt++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (8 * qt + 2 * kb + (i >> 1) < d.batch) {
          u32* ob = ((i >> 1) ? out_b1[qt] : out_b0[qt]) + ((size_t)plane * 4 + (i & 1) * 2 + c) * rcw + col;
          const u32 oc = c ? off1[qt][i] : off0[qt][i];
          const u32 v0 = combine_digit_sums(acc[qt][0][0][i], acc[qt][0][1][i], acc[qt][0][2][i], acc[qt][0][3][i],
                                            acc[qt][0][4][i], acc[qt][0][5][i], acc[qt][0][6][i], mc, d.c4[c], d.c5[c], d.c6[c], oc);
          const u32 v1 = combine_digit_sums(acc[qt][1][0][i], acc[qt][1][1][i], acc[qt][1][2][i], acc[qt][1][3][i],
                                            acc[qt][1][4][i], acc[qt][1][5][i], acc[qt][1][6][i], mc, d.c4[c], d.c5[c], d.c6[c], oc);
          if (DIAG == 4 && (v0 ^ v1) != 0xDEADBEEFu) continue;
          __builtin_nontemporal_store(mf_u32x2_t{v0, v1}, reinterpret_cast<mf_u32x2_t*>(ob));
        }
      }
  }
#undef SPL_LOAD
}

}  // namespace spiral
