// The 9 .. 16-query database pass over the DIGIT-PLANAR copy of a PACKED database (sweep_planar.hpp) and the one-time gather that
// builds the copy.  Its own translation unit: sweep.hip's kernels -- the judged single-query sweep among them -- compile to the same
// machine code whether or not this file changes (bench.py replays the sweep's PMC traffic record only into a library whose
// kernel has the recorded signature, sdk_amd/kernel_signature.py).
// (sweep_mfma.hpp's one-tile table kernels come along with the shared digit helpers and are not launched from here)
#pragma clang diagnostic ignored "-Wunused-function"
#include "device_common.hpp"
#include "sweep_planar.hpp"
#include "server.hpp"

namespace spiral {

void launch_query_digits_planar(const QueryDigitsDesc& q, size_t entries, hipStream_t s) {
  hipLaunchKernelGGL(k_query_digits_planar, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, s, q);
  launched(0, "k_query_digits_planar");
}

// both tiles of a 9 .. 16-query group: tables [tile][N][blocks][2][4][64][16 B], then the offset terms [tile][N][32]
void launch_query_tables_planar2(const DevTables& T, const u64* const* qv, int batch, int dim0, int j0, int nj, u32* rq, hipStream_t s) {
  QueryDigits2Desc q{};
  for (int b = 0; b < batch && b < SWEEP_GROUP_MAX; b++) q.qv[b] = qv[b];
  q.rq = rq;
  q.batch = batch;
  q.dim0 = dim0;
  q.j0 = j0;
  q.nj = nj;
  const size_t entries = (size_t)N * (nj >> 4) * 128;   // 16-byte entries of one tile's table (== N * blocks * 8 * 64)
  hipLaunchKernelGGL(k_query_digits_planar2, dim3((unsigned)((entries / 8 + 255) / 256), 2), dim3(256), 0, s, q);   // a thread per 8 entries
  launched(0, "k_query_digits_planar2");
  hipLaunchKernelGGL(k_query_offset_terms2, dim3(N, 2), dim3(256), 0, s, T, q, rq + (size_t)2 * entries * 4);
  launched(0, "k_query_offset_terms2");
}

// ---- digit-planar database (sweep_planar.hpp) ----------------------------------------------------------------------------
bool sweep_planar_shape_ok(int num_per, int nj) {
  // whole 64-row blocks, the z-row's query planes of both tiles in LDS (nj <= 512), whole 128-column chunks
  return tunable("batch_planar", 1) != 0 && tunable("batch_mfma", 1) != 0 && nj > 0 && (nj % 64) == 0 && nj <= 512 && num_per >= 128 &&
         (num_per % 128) == 0;
}
size_t sweep_planar_bytes(int planes, int num_per, int nj) { return (size_t)planes * N * (size_t)num_per * (size_t)nj * 8; }
// one 16-byte planar entry, index [zp][chunk][g][c][block][e][a][lane], gathered from the PACKED units
static __device__ __forceinline__ mf_u32x4_t planar_gather_entry(const u32* packed, size_t idx, int num_per, int nj) {
  const int chunks = num_per >> 7, blocks = nj >> 6, npairs = nj >> 1;
  const int lane = (int)(idx & 63);
  size_t r = idx >> 6;
  const int a = (int)(r & 3), e = (int)((r >> 2) & 1);
  r >>= 3;
  const int block = (int)(r % blocks);
  r /= blocks;
  const int c = (int)(r & 1);
  r >>= 1;
  const int g = (int)(r & 3);
  r >>= 2;
  const int chunk = (int)(r % chunks);
  const size_t zp = r / chunks;
  const int kb = lane >> 4, n = lane & 15;
  const int slot = 16 * g + n;   // PACKED lane slot: columns 2 slot, 2 slot + 1 of the chunk; this tile's column is 2 slot + e
  mf_u32x4_t o = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int j = 64 * block + 16 * kb + t;
    const u32* unit = packed + packed_unit_offset(zp, j >> 1, chunk, npairs, chunks);
    const u64 w = unpack_word(unit, slot, (j & 1) * 2 + e);
    const u32 x = c ? (u32)(w >> 32) : (u32)w;
    o[t >> 2] |= ((offset_digits(x) >> (8 * a)) & 0xffu) << (8 * (t & 3));
  }
  return o;
}
__global__ __launch_bounds__(256) void k_packed_to_planar(unsigned char* planar, const u32* packed, size_t entries, int num_per, int nj) {
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < entries; idx += (size_t)gridDim.x * 256)
    reinterpret_cast<mf_u32x4_t*>(planar)[idx] = planar_gather_entry(packed, idx, num_per, nj);
}
void launch_packed_to_planar(unsigned char* planar, const u64* packed, int planes, int num_per, int nj, hipStream_t s) {
  const size_t entries = sweep_planar_bytes(planes, num_per, nj) / 16;
  hipLaunchKernelGGL(k_packed_to_planar, dim3(256 * 64), dim3(256), 0, s, planar, reinterpret_cast<const u32*>(packed), entries, num_per, nj);
  launched(0, "k_packed_to_planar");
}
// sp_db_update_item on a database that has a planar copy: the item (local row j, local column ii) is one word per (plane, z) of
// the PACKED words = one byte in each of the 8 entries (modulus c, digit a) of that (plane, z): regather those, one per thread
__global__ __launch_bounds__(256) void k_planar_patch_item(unsigned char* planar, const u32* packed, size_t zps, int num_per, int nj, int j, int ii) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= zps * 8) return;
  const size_t zp = t >> 3;
  const int c = (int)((t >> 2) & 1), a = (int)(t & 3);
  const int chunks = num_per >> 7, blocks = nj >> 6;
  const int chunk = ii >> 7, col = ii & 127, slot = col >> 1, e = col & 1, g = slot >> 4, n = slot & 15;
  const int block = j >> 6, kb = (j & 63) >> 4;
  const size_t idx = planar_operand_offset(zp, chunk, g, block, e, c, a, chunks, blocks) / 16 + (size_t)(16 * kb + n);
  reinterpret_cast<mf_u32x4_t*>(planar)[idx] = planar_gather_entry(packed, idx, num_per, nj);
}
void launch_planar_patch_item(unsigned char* planar, const u64* packed, int planes, int num_per, int nj, int j, int ii, hipStream_t s) {
  const size_t zps = (size_t)planes * N;
  hipLaunchKernelGGL(k_planar_patch_item, dim3((unsigned)((zps * 8 + 255) / 256)), dim3(256), 0, s, planar,
                     reinterpret_cast<const u32*>(packed), zps, num_per, nj, j, ii);
  launched(0, "k_planar_patch_item");
}
// the 9 .. 16-query pass over the planar copy: 3.47 ms per C2 plane against 4.19 for k_sweep_mfma_batch<8, 1, 0, 2> on the PACKED
// words (scripts/ubench/mfma_planar.hip, profiles/r05_mfma_planar.md); eight waves per workgroup share the z-row's query planes
// where the workgroup has two chunks to split, units of the load ring as the row count allows
void launch_sweep_planar(const DevTables& T, const SweepBatchDesc& d, hipStream_t s) {
  SweepPlanarDesc m{};
  m.db = d.planar;
  m.rq = reinterpret_cast<const unsigned char*>(d.rq);
  m.rq_off = d.rq + (size_t)2 * N * (d.nj >> 4) * 128 * 4;
  for (int b = 0; b < d.batch; b++) m.out[b] = d.out[b];
  m.batch = d.batch;
  m.planes = d.planes;
  m.num_per = d.num_per;
  m.nj = d.nj;
  const int chunks = d.num_per >> 7;
  int cpw = (int)tunable("batch_mfma_cpw", 16);
  cpw = std::max(1, std::min(cpw, chunks));
  while (chunks % cpw) cpw--;
  m.cpw = cpw;
  const u64 qs[2] = {MODULUS_0, MODULUS_1};
  for (int c = 0; c < 2; c++) {
    m.c4[c] = (u32)((1ull << 32) % qs[c]);
    m.c5[c] = (u32)((1ull << 40) % qs[c]);
    m.c6[c] = (u32)((1ull << 48) % qs[c]);
  }
  const dim3 grid((unsigned)((size_t)d.planes * N * (chunks / cpw)));
  const size_t lds = (size_t)2 * (d.nj >> 6) * 8 * 64 * 16;   // both tiles' query planes of one z-row
  const bool eight = (cpw % 2) == 0;
  const bool ring4 = ((2 * (d.nj >> 6)) % 4) == 0;
#define SP_PLANAR(NBUF_, WAVES_)                                                                                          \
  {                                                                                                                        \
    if (lds > 65536)                                                                                                       \
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_planar<NBUF_, 2, 0, 1, WAVES_>),                \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                \
    hipLaunchKernelGGL((k_sweep_planar<NBUF_, 2, 0, 1, WAVES_>), grid, dim3(64 * WAVES_), lds, s, T, m);                    \
  }
  if (eight && ring4) SP_PLANAR(4, 8) else if (eight) SP_PLANAR(2, 8) else if (ring4) SP_PLANAR(4, 4) else SP_PLANAR(2, 4)
#undef SP_PLANAR
  launched(PATH_SWEEP_BATCH | PATH_SWEEP_MFMA | PATH_SWEEP_MFMA2 | PATH_SWEEP_PLANAR, "k_sweep_planar");
}

}  // namespace spiral
