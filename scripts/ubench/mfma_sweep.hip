// Microbenchmark + self-check of the MFMA batched sweep (sdk_amd/csrc/sweep_mfma.hpp) on one plane of a synthetic
// PACKED database, outside the library:
//   1. layout probe of v_mfma_i32_16x16x64_i8 (which (lane, register) holds D[m][n]; A/B k-blocks pair up lane-group-wise)
//   2. read-only kernels with the two candidate access patterns (whole units per wave / 16-slot quarters per wave)
//   3. k_query_digits + k_sweep_mfma_batch<NB, MINWG, DIAG> variants: time per plane pass, sampled outputs against a CPU u128 sum
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../sdk_amd/csrc -I../../include mfma_sweep.hip -o mfma_sweep
// Run:    ./mfma_sweep [nz = 2048] [reps = 3]
#include "sweep_mfma.hpp"
#include "mfma_sweep6.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using namespace spiral;

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

__host__ __device__ inline u32 hash32(u64 i, u64 seed) {
  u64 z = seed + 0x9E3779B97F4A7C15ULL * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (u32)(z ^ (z >> 31));
}
constexpr u64 SEED_DB = 0x1234, SEED_Q = 0x9876;

__global__ void k_fill_db(u32* db, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) db[i] = hash32(i, SEED_DB);
}
// qv[b][z][j][r] = lo | hi << 32, residues < q
__host__ __device__ inline u64 synth_q(int b, size_t idx) {
  const u32 h0 = hash32(idx * 2, SEED_Q + b), h1 = hash32(idx * 2 + 1, SEED_Q + b);
  return (u64)(h0 % (u32)MODULUS_0) | ((u64)(h1 % (u32)MODULUS_1) << 32);
}
__global__ void k_fill_q(u64* qv, int b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) qv[i] = synth_q(b, i);
}

// ---- 1. layout probe ------------------------------------------------------------------------------------------------
__global__ void k_probe(const int* a, const int* b, int* dout) {
  const int l = threadIdx.x;
  v4i_t A = {a[l * 4], a[l * 4 + 1], a[l * 4 + 2], a[l * 4 + 3]};
  v4i_t B = {b[l * 4], b[l * 4 + 1], b[l * 4 + 2], b[l * 4 + 3]};
  v4i_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) dout[l * 4 + i] = c[i];
}
static void probe() {
  std::mt19937 rng(7);
  std::vector<int8_t> a(64 * 16), b(64 * 16);
  for (auto& x : a) x = (int8_t)(rng() & 0xFF);
  for (auto& x : b) x = (int8_t)(rng() & 0xFF);
  int *da, *db_, *dd;
  CK(hipMalloc(&da, 1024));
  CK(hipMalloc(&db_, 1024));
  CK(hipMalloc(&dd, 1024));
  CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(db_, b.data(), 1024, hipMemcpyHostToDevice));
  k_probe<<<1, 64>>>(da, db_, dd);
  std::vector<int> d(256);
  CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
  // hypothesis: lane l of A = row l % 16, lane l of B = column l % 16, bytes of lane group l / 16 pair up; D[m][n] in
  // lane (m / 4) * 16 + n, register m % 4
  long ref[16][16];
  for (int m = 0; m < 16; m++)
    for (int n = 0; n < 16; n++) {
      long s = 0;
      for (int kb = 0; kb < 4; kb++)
        for (int t = 0; t < 16; t++) s += (long)a[(kb * 16 + m) * 16 + t] * (long)b[(kb * 16 + n) * 16 + t];
      ref[m][n] = s;
    }
  int bad_std = 0, bad_tr = 0;
  for (int l = 0; l < 64; l++)
    for (int i = 0; i < 4; i++) {
      const int m = (l >> 4) * 4 + i, n = l & 15;
      if (d[l * 4 + i] != ref[m][n]) bad_std++;
      if (d[l * 4 + i] != ref[n][m]) bad_tr++;
    }
  printf("[probe] v_mfma_i32_16x16x64_i8: D[m = 4 (lane / 16) + reg][n = lane %% 16] mismatches %d / 256; transposed hypothesis %d / 256\n",
         bad_std, bad_tr);
  CK(hipFree(da));
  CK(hipFree(db_));
  CK(hipFree(dd));
}

// ---- 2. read-only kernels -------------------------------------------------------------------------------------------
// (a) whole units: wave = (zp, chunk) stream, 4 row pairs in flight (the single-query sweep's pattern)
__global__ __launch_bounds__(256) void k_read_units(const u32* db, u32* sink, int chunks, int npairs, long units) {
  const int lane = threadIdx.x & 63;
  const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const u32* base = db + (size_t)unit * npairs * 448;
  u32 x = 0;
  for (int jp = 0; jp < npairs; jp += 4) {
    mf_u32x4_t a[4];
    mf_u32x3_t b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const u32* p = base + (size_t)(jp + u) * 448;
      a[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x4_t*>(p + lane * 4));
      b[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x3_t*>(p + 256 + lane * 3));
    }
#pragma unroll
    for (int u = 0; u < 4; u++) x ^= a[u].x ^ a[u].y ^ a[u].z ^ a[u].w ^ b[u].x ^ b[u].y ^ b[u].z;
  }
  if (x == 0x12345u) sink[0] = x;
}
// (b) quarters: workgroup = (zp, chunk) stream, wave g reads slots 16g..16g+15 of every unit, lane (kb, m) the row
// pairs 8s + 2kb + {0, 1}; NB - 1 steps ahead
template <int NB>
__global__ __launch_bounds__(256) void k_read_quarters(const u32* db, u32* sink, int npairs) {
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, kb = lane >> 4, mp = lane & 15;
  const u32* pu = db + (size_t)blockIdx.x * npairs * 448 + (size_t)(2 * kb) * 448;
  const u32* p4 = pu + (16 * g + mp) * 4;
  const u32* p3 = pu + 256 + (16 * g + mp) * 3;
  const int steps = npairs >> 3;
  u32 x = 0;
  for (int s0 = 0; s0 < steps; s0 += NB) {
    mf_u32x4_t a[NB][2];
    mf_u32x3_t b[NB][2];
#pragma unroll
    for (int k = 0; k < NB; k++) {
      const u32* q4 = p4 + (size_t)(s0 + k) * 3584;
      const u32* q3 = p3 + (size_t)(s0 + k) * 3584;
      a[k][0] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x4_t*>(q4));
      b[k][0] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x3_t*>(q3));
      a[k][1] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x4_t*>(q4 + 448));
      b[k][1] = __builtin_nontemporal_load(reinterpret_cast<const mf_u32x3_t*>(q3 + 448));
    }
#pragma unroll
    for (int k = 0; k < NB; k++)
#pragma unroll
      for (int u = 0; u < 2; u++) x ^= a[k][u].x ^ a[k][u].y ^ a[k][u].z ^ a[k][u].w ^ b[k][u].x ^ b[k][u].y ^ b[k][u].z;
  }
  if (x == 0x12345u) sink[0] = x;
}

// ---- 3. the sweep ---------------------------------------------------------------------------------------------------
struct Ctx {
  int nz, num_per, nj, chunks, npairs, batch;
  u32* db;
  u64* qv[16];
  u32* rq;
  u32* out[16];
  DevTables T;
  SweepMfmaDesc d;
};

static u32 host_limb(const Ctx& c, int zp, int chunk, int j, int slot, int e, int crt) {
  const int jp = j >> 1, rowp = j & 1;
  const size_t U = (((size_t)zp * c.chunks + chunk) * c.npairs + jp) * 448;
  u32 dd[8];
  for (int k = 0; k < 4; k++) dd[k] = hash32(U + slot * 4 + k, SEED_DB);
  for (int k = 0; k < 3; k++) dd[4 + k] = hash32(U + 256 + slot * 3 + k, SEED_DB);
  dd[7] = 0;
  const int L = (rowp * 2 + e) * 2 + crt, bit = 28 * L, w = bit >> 5, sh = bit & 31;
  const u64 two = (u64)dd[w] | ((u64)dd[w + 1 < 8 ? w + 1 : 7] << 32);
  return (u32)(two >> sh) & 0x0FFFFFFFu;
}

static int verify(const Ctx& c, const char* tag, int samples) {
  std::mt19937_64 rng(99);
  const size_t rcw = (size_t)N * c.num_per;
  int bad = 0;
  for (int s = 0; s < samples; s++) {
    const int z = (int)(rng() % c.nz), ii = (int)(rng() % c.num_per), n = (int)(rng() % (2 * c.batch)), crt = (int)(rng() & 1);
    const int b = n >> 1, r = n & 1, chunk = ii >> 7, slot = (ii & 127) >> 1, e = ii & 1;
    const u64 q = crt ? MODULUS_1 : MODULUS_0;
    unsigned __int128 sum = 0;
    for (int j = 0; j < c.nj; j++) {
      const u64 w = synth_q(b, ((size_t)z * c.nj + j) * 2 + r);
      const u32 y = crt ? (u32)(w >> 32) : (u32)w;
      sum += (unsigned __int128)host_limb(c, z, chunk, j, slot, e, crt) * y;
    }
    const u32 want = (u32)(sum % q);
    u32 got = 0;
    CK(hipMemcpy(&got, c.out[b] + ((size_t)(r * 2 + crt)) * rcw + (size_t)z * c.num_per + ii, 4, hipMemcpyDeviceToHost));
    if (got != want) {
      if (bad < 8) printf("  [%s] MISMATCH z=%d ii=%d n=%d crt=%d: got %u want %u\n", tag, z, ii, n, crt, got, want);
      bad++;
    }
  }
  printf("[verify %s] %d / %d sampled outputs wrong\n", tag, bad, samples);
  return bad;
}

template <int NB, int MINWG, int DIAG = 0>
static void run_variant(Ctx& c, int cpw, size_t lds_pad, int reps, const char* tag) {
  c.d.cpw = cpw;
  const size_t lds = (size_t)c.nj * 128 + lds_pad;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_mfma_batch<NB, MINWG, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int b = 0; b < c.batch; b++) CK(hipMemset(c.out[b], 0xEE, (size_t)4 * N * c.num_per * 4));
  const dim3 grid((unsigned)((size_t)c.nz * (c.chunks / cpw)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_sweep_mfma_batch<NB, MINWG, DIAG>), grid, dim3(256), lds, 0, c.T, c.d);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_sweep_mfma_batch<NB, MINWG, DIAG>), grid, dim3(256), lds, 0, c.T, c.d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes = (double)c.nz * c.chunks * c.npairs * 1792.0;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_sweep_mfma_batch<NB, MINWG, DIAG>)));
  printf("[mfma %s] NB=%d MINWG=%d cpw=%d lds=%zu regs=%d: best %.3f ms avg %.3f ms per plane-pass of %d z-rows = %.0f GB/s of database (x4 planes: %.2f ms per B=8 pass)\n",
         tag, NB, MINWG, cpw, lds, fa.numRegs, best, sum / reps, c.nz, bytes / (best * 1e-3) / 1e9, 4.0 * best * 2048.0 / c.nz);
  if (DIAG == 0) verify(c, tag, 400);
}

// QT = 2: sixteen queries per pass (two digit tables in LDS: one workgroup per CU)
template <int NB, int DIAG = 0>
static void run_variant16(Ctx& c, int cpw, int reps, const char* tag) {
  c.d.cpw = cpw;
  const int batch_was = c.batch;
  c.batch = 16;
  c.d.batch = 16;
  const size_t lds = (size_t)c.nj * 128 * 2 + 256;   // + the tiles' offset terms (r06)
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_mfma_batch<NB, 1, DIAG, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int b = 0; b < 16; b++) CK(hipMemset(c.out[b], 0xEE, (size_t)4 * N * c.num_per * 4));
  const dim3 grid((unsigned)((size_t)c.nz * (c.chunks / cpw)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_sweep_mfma_batch<NB, 1, DIAG, 2>), grid, dim3(256), lds, 0, c.T, c.d);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_sweep_mfma_batch<NB, 1, DIAG, 2>), grid, dim3(256), lds, 0, c.T, c.d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes = (double)c.nz * c.chunks * c.npairs * 1792.0;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_sweep_mfma_batch<NB, 1, DIAG, 2>)));
  printf("[mfma16 %s] QT=2 NB=%d cpw=%d lds=%zu regs=%d scratch=%zu: best %.3f ms avg %.3f ms per plane-pass of %d z-rows = %.0f GB/s of database (x4 planes: %.2f ms per B=16 pass = %.3f ms per query)\n",
         tag, NB, cpw, lds, fa.numRegs, (size_t)fa.localSizeBytes, best, sum / reps, c.nz, bytes / (best * 1e-3) / 1e9, 4.0 * best * 2048.0 / c.nz,
         4.0 * best * 2048.0 / c.nz / 16.0);
  if (DIAG == 0) verify(c, tag, 400);
  c.batch = batch_was;
  c.d.batch = batch_was;
}

template <int NB, int DIAG = 0>
static void run_variant6(Ctx& c, int cpw, int reps, const char* tag) {
  c.d.cpw = cpw;
  const size_t lds = (size_t)c.nj * 128;
  for (int b = 0; b < c.batch; b++) CK(hipMemset(c.out[b], 0xEE, (size_t)4 * N * c.num_per * 4));
  const dim3 grid((unsigned)((size_t)c.nz * (c.chunks / cpw)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_sweep_mfma_batch6<NB, DIAG>), grid, dim3(384), lds, 0, c.T, c.d);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_sweep_mfma_batch6<NB, DIAG>), grid, dim3(384), lds, 0, c.T, c.d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes = (double)c.nz * c.chunks * c.npairs * 1792.0;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_sweep_mfma_batch6<NB, DIAG>)));
  printf("[mfma6 %s] 6 waves per workgroup, NB=%d cpw=%d regs=%d scratch=%zu: best %.3f ms avg %.3f ms per plane-pass = %.0f GB/s of database (x4 planes: %.2f ms per B=8 pass)\n",
         tag, NB, cpw, fa.numRegs, (size_t)fa.localSizeBytes, best, sum / reps, bytes / (best * 1e-3) / 1e9, 4.0 * best * 2048.0 / c.nz);
  if (DIAG == 0) verify(c, tag, 400);
}

// ---- 4. issue rates: independent MFMAs alone and with VALU work between them (inline asm: the compiler's own version of
// this loop shuffled the accumulators through v_accvgpr moves and measured that instead) ----------------------------------
template <int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void k_mfma_rate(int* out, int iters) {
  v4i_t acc[8];
  v4i_t a = {(int)threadIdx.x * 0x01010101, 0x7F3C2D1E, (int)0x80FF7F01, 0x11223344};
  v4i_t b = {0x05060708, (int)blockIdx.x, (int)0xF1E2D3C4, 0x0A0B0C0D};
  u32 x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7;
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = v4i_t{i, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (VALU_PER_MFMA >= 1) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x0));
      if (VALU_PER_MFMA >= 2) asm volatile("v_xor_b32 %0, 0x808080, %0" : "+v"(x1));
      if (VALU_PER_MFMA >= 3) asm volatile("v_add_u32 %0, 0x808080, %0" : "+v"(x2));
      if (VALU_PER_MFMA >= 4) asm volatile("v_and_b32 %0, 0xfffffff, %0" : "+v"(x3));
      if (VALU_PER_MFMA >= 5) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x0));
      if (VALU_PER_MFMA >= 6) asm volatile("v_xor_b32 %0, 0x808080, %0" : "+v"(x1));
      if (VALU_PER_MFMA >= 7) asm volatile("v_add_u32 %0, 0x808080, %0" : "+v"(x2));
      if (VALU_PER_MFMA >= 8) asm volatile("v_and_b32 %0, 0xfffffff, %0" : "+v"(x3));
    }
  }
  int sacc = (int)(x0 + x1 + x2 + x3);
#pragma unroll
  for (int i = 0; i < 8; i++) sacc += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}
template <int V>
static void mfma_rate(int wgs_per_cu, int* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 40000;
  // co-residency forced by dynamic LDS: 160 KiB / wgs_per_cu per workgroup -> exactly wgs_per_cu workgroups per CU
  const size_t lds = (size_t)(160 * 1024 / wgs_per_cu) - 512;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mfma_rate<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_mfma_rate<V>), dim3(256 * wgs_per_cu), dim3(256), lds, 0, out, 100);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mfma_rate<V>), dim3(256 * wgs_per_cu), dim3(256), lds, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfmas_per_simd = 8.0 * iters * wgs_per_cu;
  printf("[mfma rate] %d VALU per MFMA, %d waves/SIMD: %.3f ms; %.2f ns per MFMA per SIMD = %.0f TOPS (dense i8), %.2f ns per (MFMA + %d VALU) per wave\n",
         V, wgs_per_cu, ms, ms * 1e6 / mfmas_per_simd, 2.0 * 16384 * mfmas_per_simd * 1024 / (ms * 1e-3) / 1e12, ms * 1e6 / (8.0 * iters), V);
}
int main(int argc, char** argv) {
  const int nz = argc > 1 ? atoi(argv[1]) : 2048;
  const int reps = argc > 2 ? atoi(argv[2]) : 3;
  probe();
  Ctx c{};
  c.nz = nz;
  c.num_per = 2048;
  c.nj = 512;
  c.chunks = c.num_per >> 7;
  c.npairs = c.nj >> 1;
  c.batch = 8;
  const size_t db_dwords = (size_t)nz * c.chunks * c.npairs * 448;
  CK(hipMalloc(&c.db, db_dwords * 4 + 65536));
  k_fill_db<<<256 * 32, 256>>>(c.db, db_dwords + 16384);
  const size_t qn = (size_t)N * c.nj * 2;
  for (int b = 0; b < 16; b++) {
    CK(hipMalloc(&c.qv[b], qn * 8));
    k_fill_q<<<(unsigned)((qn + 255) / 256), 256>>>(c.qv[b], b, qn);
    CK(hipMalloc(&c.out[b], (size_t)4 * N * c.num_per * 4 * (b == 0 ? 8 : 1)));
  }
  CK(hipMalloc(&c.rq, 2 * ((size_t)N * (c.nj / 16) * 128 * 16 + (size_t)N * 32 * 4)));  // two digit tables + two sets of offset terms
  CK(hipDeviceSynchronize());
  printf("database: %d z-rows x %d columns x %d rows = %.2f GB packed\n", nz, c.num_per, c.nj, db_dwords * 4 / 1e9);

  // constants
  DevConsts dc{};
  const u64 qs[2] = {MODULUS_0, MODULUS_1};
  for (int i = 0; i < 2; i++) {
    dc.mod[i].q = (u32)qs[i];
    dc.mod[i].two_q = (u32)(2 * qs[i]);
    dc.mod[i].m64 = (u64)(((unsigned __int128)1 << 64) / qs[i]);
  }
  c.T.tw = nullptr;
  c.T.c = dc;

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  u32* sink;
  CK(hipMalloc(&sink, 64));
  const double bytes = (double)db_dwords * 4;
  {  // read-only patterns
    const long units = (long)nz * c.chunks;
    for (int r = 0; r < reps + 1; r++) {
      CK(hipEventRecord(e0));
      k_read_units<<<(unsigned)((units + 3) / 4), 256>>>(c.db, sink, c.chunks, c.npairs, units);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) printf("[read whole units, 1 wave per stream]      %.3f ms = %.0f GB/s\n", ms, bytes / (ms * 1e-3) / 1e9);
    }
    for (int r = 0; r < reps + 1; r++) {
      CK(hipEventRecord(e0));
      k_read_quarters<2><<<(unsigned)units, 256>>>(c.db, sink, c.npairs);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) printf("[read 16-slot quarters, NB=2, 4 waves/stream] %.3f ms = %.0f GB/s\n", ms, bytes / (ms * 1e-3) / 1e9);
    }
    for (int r = 0; r < reps + 1; r++) {
      CK(hipEventRecord(e0));
      k_read_quarters<4><<<(unsigned)units, 256>>>(c.db, sink, c.npairs);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) printf("[read 16-slot quarters, NB=4, 4 waves/stream] %.3f ms = %.0f GB/s\n", ms, bytes / (ms * 1e-3) / 1e9);
    }
  }

  // digit table
  QueryDigitsDesc qd{};
  for (int b = 0; b < 8; b++) qd.qv[b] = c.qv[b];
  qd.rq = c.rq;
  qd.batch = 8;
  qd.dim0 = c.nj;
  qd.j0 = 0;
  qd.nj = c.nj;
  const size_t entries = (size_t)N * (c.nj / 16) * 128;
  CK(hipEventRecord(e0));
  k_query_digits<<<(unsigned)((entries + 255) / 256), 256>>>(qd);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("[k_query_digits] %.3f ms for 8 queries\n", ms);
  CK(hipEventRecord(e0));
  k_query_offset_terms<<<N, 256>>>(c.T, qd, c.rq + 2 * entries * 4);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("[k_query_offset_terms] %.3f ms for 8 queries\n", ms);

  {  // second tile: queries 8 .. 15 -> table 1 / offset terms 1 (layout [tile][N][...], tables first, then the offset terms)
    QueryDigitsDesc qd2 = qd;
    for (int b = 0; b < 8; b++) qd2.qv[b] = c.qv[8 + b];
    qd2.rq = c.rq + entries * 4;
    k_query_digits<<<(unsigned)((entries + 255) / 256), 256>>>(qd2);
    // offset terms: [tile 0][N][32] right after BOTH tables, then [tile 1][N][32]
    k_query_offset_terms<<<N, 256>>>(c.T, qd, c.rq + 2 * entries * 4);
    k_query_offset_terms<<<N, 256>>>(c.T, qd2, c.rq + 2 * entries * 4 + (size_t)N * 32);
    CK(hipDeviceSynchronize());
  }
  c.d.db = reinterpret_cast<const u64*>(c.db);
  c.d.rq = c.rq;
  c.d.rq_off = c.rq + 2 * entries * 4;
  for (int b = 0; b < 16; b++) c.d.out[b] = c.out[b];
  c.d.batch = 8;
  c.d.planes = 1;
  c.d.num_per = c.num_per;
  c.d.nj = c.nj;
  {
    const u64 qs[2] = {MODULUS_0, MODULUS_1};
    for (int k = 0; k < 2; k++) {
      c.d.c4[k] = (u32)((1ull << 32) % qs[k]);
      c.d.c5[k] = (u32)((1ull << 40) % qs[k]);
      c.d.c6[k] = (u32)((1ull << 48) % qs[k]);
    }
  }

  if (!getenv("MFMA_UBENCH_NORATE")) {
    int* o;
    CK(hipMalloc(&o, 256 * 8 * 256 * 4));
    mfma_rate<0>(1, o);
    mfma_rate<0>(2, o);
    mfma_rate<0>(4, o);
    mfma_rate<2>(1, o);
    mfma_rate<4>(1, o);
    mfma_rate<4>(2, o);
    mfma_rate<8>(1, o);
    mfma_rate<8>(2, o);
    mfma_rate<8>(3, o);
  }
  run_variant<2, 2>(c, 16, 0, reps, "a");
  if (getenv("MFMA_UBENCH_B16")) {  // round 4: sixteen queries per pass
    run_variant16<2>(c, 16, reps, "QT2 NB=2");
    run_variant16<4>(c, 16, reps, "QT2 NB=4");
    run_variant16<8>(c, 16, reps, "QT2 NB=8");
    run_variant16<4>(c, 4, reps, "QT2 NB=4 cpw=4");
    run_variant16<4, 4>(c, 16, reps, "QT2 NB=4 DIAG4 no stores");
    run_variant16<4, 1>(c, 16, reps, "QT2 NB=4 DIAG1 compute only");
    run_variant<4, 1>(c, 16, 32768, reps, "B=8 forced 1 WG/CU NB=4 (reference)");
    run_variant<2, 2>(c, 16, 0, reps, "a again");
    return 0;
  }
  run_variant6<2>(c, 16, reps, "six-wave NB=2");
  run_variant6<1>(c, 16, reps, "six-wave NB=1");
  run_variant6<2>(c, 8, reps, "six-wave NB=2 cpw=8");
  run_variant6<2, 1>(c, 16, reps, "six-wave NB=2 DIAG1 compute only");
  run_variant6<2, 4>(c, 16, reps, "six-wave NB=2 DIAG4 no stores");
  if (getenv("MFMA_UBENCH_SIX_ONLY")) return 0;
  run_variant<2, 2, 1>(c, 16, 0, reps, "a DIAG1 compute only (no loads)");
  run_variant<2, 2, 2>(c, 16, 0, reps, "a DIAG2 no MFMA (loads + VALU)");
  run_variant<2, 2, 3>(c, 16, 0, reps, "a DIAG3 no digit VALU (loads + MFMA)");
  run_variant<2, 2, 4>(c, 16, 0, reps, "a DIAG4 no output stores");
  run_variant<2, 2, 5>(c, 16, 0, reps, "a DIAG5 stores hit one L2-resident region");
  run_variant<2, 2, 7>(c, 16, 0, reps, "a DIAG7 contiguous 4 KiB per wave and chunk");
  run_variant<2, 2, 6>(c, 16, 0, reps, "a DIAG6 plain (cached) stores");
  verify(c, "a DIAG6", 400);
  run_variant<4, 2>(c, 16, 0, reps, "b");
  run_variant<4, 2, 1>(c, 16, 0, reps, "b DIAG1 compute only (no loads)");
  run_variant<4, 2, 4>(c, 16, 0, reps, "b DIAG4 no output stores");
  if (getenv("MFMA_UBENCH_SHORT")) return 0;
  run_variant<4, 2>(c, 16, 0, reps, "b");
  run_variant<4, 2>(c, 4, 0, reps, "c");
  run_variant<2, 2>(c, 4, 0, reps, "d");
  run_variant<4, 2>(c, 1, 0, reps, "e");
  run_variant<4, 1>(c, 16, 0, reps, "f (1 WG/CU budget, may still run 2)");
  run_variant<4, 1>(c, 16, 32768, reps, "g (forced 1 WG/CU by LDS)");
  run_variant<8, 1>(c, 16, 32768, reps, "h (forced 1 WG/CU, 7 steps ahead)");
  run_variant<2, 2>(c, 16, 0, reps, "a again");
  return 0;
}
