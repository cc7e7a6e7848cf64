// NTT-domain multiply-accumulate and the small elementwise kernels of the answer path (poly.rs:351-663, gadget.rs:34-60,
// util.rs:323-355, server.rs:470-517).
#include "device_common.hpp"
#include "bodies.hpp"

namespace spiral {

// ------------------------------------------------------------------------------------------------
// NTT-domain multiply-accumulate.  grid (batch, 2N/256[, outer]), one (crt, z) per thread.  The batch runs along
// grid.x: it is unbounded (num_per * planes on the unfused fold path), grid.y / grid.z are limited to 65535.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mac(DevTables T, MacDesc d) { mac_body(T, d, blockIdx.x, blockIdx.z, blockIdx.y); }
__global__ __launch_bounds__(256) void k_mac2(DevTables T, MacDesc d0, MacDesc d1) {
  const int y = blockIdx.x;
  if (y < d0.batch_inner)
    mac_body(T, d0, y, 0, blockIdx.y);
  else
    mac_body(T, d1, y - d0.batch_inner, 0, blockIdx.y);
}
// grouped form (kernels.hpp, GroupOff): grid.z = query
__global__ __launch_bounds__(256) void k_mac2_group(DevTables T, MacDesc d0, MacDesc d1, GroupOff g) {
  const int qi = blockIdx.z;
  int y = blockIdx.x;
  MacDesc d = d0;
  if (y >= d0.batch_inner) {
    d = d1;
    y -= d0.batch_inner;
  }
  d.A = group_rebase(d.A, g.pp[qi]);
  d.B = group_rebase(d.B, g.dig[qi]);
  d.out = group_rebase(d.out, g.v[qi]);
  d.addend = group_rebase(d.addend, g.v[qi]);
  d.extra = group_rebase(d.extra, g.ct1[qi]);
  mac_body(T, d, y, 0, blockIdx.y);
}
void launch_mac2_group(const DevTables& T, const MacDesc& d0, const MacDesc& d1, const GroupOff& g, int B, hipStream_t s) {
  MacDesc a = d0, b = d1;
  a.batch_inner = std::max(a.batch_inner, 0);
  b.batch_inner = std::max(b.batch_inner, 0);
  if (a.batch_inner + b.batch_inner <= 0 || B <= 0) return;
  hipLaunchKernelGGL(k_mac2_group, dim3(a.batch_inner + b.batch_inner, 2 * N / 256, B), dim3(256), 0, s, T, a, b, g);
  launched(PATH_EXPAND_GROUP, "k_mac2_group");
}
void launch_mac(const DevTables& T, const MacDesc& d, hipStream_t s) {
  if (d.batch_inner <= 0 || d.batch_outer <= 0) return;
  hipLaunchKernelGGL(k_mac, dim3(d.batch_inner, 2 * N / 256, d.batch_outer), dim3(256), 0, s, T, d);
  launched(0, "k_mac");
}
void launch_mac2(const DevTables& T, const MacDesc& d0, const MacDesc& d1, hipStream_t s) {
  MacDesc a = d0, b = d1;
  a.batch_inner = std::max(a.batch_inner, 0);
  b.batch_inner = std::max(b.batch_inner, 0);
  if (a.batch_inner + b.batch_inner <= 0) return;
  hipLaunchKernelGGL(k_mac2, dim3(a.batch_inner + b.batch_inner, 2 * N / 256, 1), dim3(256), 0, s, T, a, b);
  launched(0, "k_mac2");
}

__global__ __launch_bounds__(256) void k_add_poly_into(DevTables T, u32* dst, const int* idx, const u32* src) {
  add_poly_into_body(T, dst, idx, src, blockIdx.x, blockIdx.y);
}
void launch_add_poly_into(const DevTables& T, u32* dst, const int* idx, const u32* src, int batch, hipStream_t s) {
  if (batch <= 0) return;
  hipLaunchKernelGGL(k_add_poly_into, dim3(batch, 2 * N / 256), dim3(256), 0, s, T, dst, idx, src);
  launched(0, "k_add_poly_into");
}

__global__ __launch_bounds__(256) void k_scalar_mul(DevTables T, u32* base, long dst_off, long src_off,
                                                    const u32* scalar) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const long b = blockIdx.x;
  const ModConst m = T.c.mod[c];
  base[(dst_off + b) * 2 * N + e] = reduce64((u64)base[(src_off + b) * 2 * N + e] * (u64)scalar[e], m);
}
void launch_scalar_mul(const DevTables& T, u32* base, long dst_off, long src_off, const u32* scalar, int n_polys,
                       hipStream_t s) {
  if (n_polys <= 0) return;
  hipLaunchKernelGGL(k_scalar_mul, dim3(n_polys, 2 * N / 256), dim3(256), 0, s, T, base, dst_off, src_off, scalar);
  launched(0, "k_scalar_mul");
}

__global__ __launch_bounds__(256) void k_add_polys_idx(DevTables T, u32* dst, const int* di, const u32* a, const int* ai,
                                                        const u32* b, const int* bi) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  const int c = e >> POLY_LEN_LOG2;
  const int k = blockIdx.x;
  dst[(size_t)di[k] * 2 * N + e] = add_mod(a[(size_t)ai[k] * 2 * N + e], b[(size_t)bi[k] * 2 * N + e], T.c.mod[c].q);
}
void launch_add_polys_idx(const DevTables& T, u32* dst, const int* di, const u32* a, const int* ai, const u32* b,
                          const int* bi, int count, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_add_polys_idx, dim3(count, 2 * N / 256), dim3(256), 0, s, T, dst, di, a, ai, b, bi);
  launched(0, "k_add_polys_idx");
}

__global__ __launch_bounds__(256) void k_interleave_query(u64* qv, const u32* row0_ntt, const u64* wire, int dim0, u32 q0,
                                                          u32 q1) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // index over [z][j]
  if (i >= (size_t)N * dim0) return;
  const int j = (int)(i % dim0);
  const int z = (int)(i / dim0);
  const u32* p = row0_ntt + (size_t)j * 2 * N;
  qv[2 * i] = (u64)p[z] | ((u64)p[N + z] << 32);
  // wire words are used as they come in the reference's u128 sums followed by one % q (server.rs:196-217); the sweep
  // kernels accumulate 256 products in u64 and need limbs < q, so the limbs are reduced here: same residues.
  const u64 w = wire[i];
  qv[2 * i + 1] = (u64)((u32)w % q0) | ((u64)((u32)(w >> 32) % q1) << 32);
}
void launch_interleave_query(u64* qv, const u32* row0_ntt, const u64* wire, int dim0, hipStream_t s) {
  size_t total = (size_t)N * dim0;
  hipLaunchKernelGGL(k_interleave_query, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qv, row0_ntt, wire, dim0,
                     (u32)MODULUS_0, (u32)MODULUS_1);
  launched(0, "k_interleave_query");
}

__global__ __launch_bounds__(256) void k_copy_polys(CopyPolysDesc d) { copy_polys_body(d, blockIdx.x, blockIdx.y); }
void launch_copy_polys(u32* dst, const int* dst_idx, int dst_row_stride, const u32* src, const int* src_idx,
                       int src_row_stride, int R, int batch, hipStream_t s) {
  if (batch <= 0) return;
  const CopyPolysDesc d{dst, dst_idx, dst_row_stride, src, src_idx, src_row_stride, R, batch};
  hipLaunchKernelGGL(k_copy_polys, dim3(batch * R, 2 * N / 256), dim3(256), 0, s, d);
  launched(0, "k_copy_polys");
}

// grouped forms (kernels.hpp, GroupOff): one more grid dimension = the query
__global__ __launch_bounds__(256) void k_copy_polys_group(CopyPolysDesc d, GroupOff g) {
  const int qi = blockIdx.z;
  d.dst = group_rebase(d.dst, g.raw[qi]);
  d.src = group_rebase(d.src, g.v[qi]);
  copy_polys_body(d, blockIdx.x, blockIdx.y);
}
void launch_copy_polys_group(u32* dst, const int* dst_idx, int dst_row_stride, const u32* src, const int* src_idx, int src_row_stride,
                             int R, int batch, const GroupOff& g, int B, hipStream_t s) {
  if (batch <= 0 || B <= 0) return;
  const CopyPolysDesc d{dst, dst_idx, dst_row_stride, src, src_idx, src_row_stride, R, batch};
  hipLaunchKernelGGL(k_copy_polys_group, dim3(batch * R, 2 * N / 256, B), dim3(256), 0, s, d, g);
  launched(PATH_EXPAND_GROUP, "k_copy_polys_group");
}

// Diagnostics for resident data: checksum of a buffer as KERNELS see it (through the caches), and a kernel whose
// waves write back and invalidate the L2 of the XCD they run on (system-scope fence: buffer_wbl2 + buffer_inv).
__global__ __launch_bounds__(256) void k_checksum(const u32* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    acc += (unsigned long long)p[i] * (unsigned long long)((i << 1) | 1);
  atomicAdd(out, acc);
}
void launch_checksum(const u32* p, size_t n_words, unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(k_checksum, dim3(512), dim3(256), 0, s, p, n_words, out);
  launched(0, "k_checksum");
}
__global__ __launch_bounds__(64) void k_cache_sync(u32* sink) {
  __threadfence_system();
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1;
  __threadfence_system();
}
void launch_cache_sync(u32* sink, hipStream_t s) {
  hipLaunchKernelGGL(k_cache_sync, dim3(1024), dim3(64), 0, s, sink);  // blocks are dealt round-robin to the 8 XCDs
  launched(0, "k_cache_sync");
}

// plain word copy (grid-stride): the last hop of every host -> device upload of resident data (upload_words, server.cpp)
__global__ __launch_bounds__(256) void k_copy_words(u32* dst, const u32* src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void launch_copy_words(u32* dst, const u32* src, size_t n_words, hipStream_t s) {
  if (n_words == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((n_words + 255) / 256, 4096);
  hipLaunchKernelGGL(k_copy_words, dim3(blocks), dim3(256), 0, s, dst, src, n_words);
  launched(0, "k_copy_words");
}

__global__ __launch_bounds__(256) void k_folding_neg(DevTables T, FoldingNegDesc d) {
  folding_neg_body(T, d, blockIdx.x, blockIdx.y, blockIdx.z);
}
void launch_folding_neg(const DevTables& T, u32* mats, const u32* gadget_ntt, int nu2, int two_t, hipStream_t s) {
  if (nu2 <= 0) return;
  const FoldingNegDesc d{mats, gadget_ntt, two_t, nu2};
  hipLaunchKernelGGL(k_folding_neg, dim3(2 * N / 256, 2 * two_t, nu2), dim3(256), 0, s, T, d);
  launched(0, "k_folding_neg");
}

__global__ __launch_bounds__(256) void k_folding_neg_group(DevTables T, FoldingNegDesc d, GroupOff g) {
  const int qi = blockIdx.z / d.nu2;
  d.mats = group_rebase(d.mats, g.v[qi]);
  folding_neg_body(T, d, blockIdx.x, blockIdx.y, blockIdx.z - qi * d.nu2);
}
void launch_folding_neg_group(const DevTables& T, u32* mats, const u32* gadget_ntt, int nu2, int two_t, const GroupOff& g, int B,
                              hipStream_t s) {
  if (nu2 <= 0 || B <= 0) return;
  const FoldingNegDesc d{mats, gadget_ntt, two_t, nu2};
  hipLaunchKernelGGL(k_folding_neg_group, dim3(2 * N / 256, 2 * two_t, nu2 * B), dim3(256), 0, s, T, d, g);
  launched(PATH_EXPAND_GROUP, "k_folding_neg_group");
}

__global__ __launch_bounds__(256) void k_add(DevTables T, u32* out, const u32* a, const u32* b) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)((i >> POLY_LEN_LOG2) & 1);
  out[i] = add_mod(a[i], b[i], T.c.mod[c].q);
}
void launch_add(const DevTables& T, u32* out, const u32* a, const u32* b, int n_polys, hipStream_t s) {
  if (n_polys <= 0) return;
  hipLaunchKernelGGL(k_add, dim3((unsigned)((size_t)n_polys * 2 * N / 256)), dim3(256), 0, s, T, out, a, b);
  launched(0, "k_add");
}

__global__ __launch_bounds__(256) void k_invert_raw(u64 Q, u64* out, const u64* a, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = Q - a[i];
}
void launch_invert_raw(const DevTables& T, u64* out, const u64* a, long n_words, hipStream_t s) {
  if (n_words <= 0) return;
  hipLaunchKernelGGL(k_invert_raw, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, T.c.Q, out, a, n_words);
  launched(0, "k_invert_raw");
}

__global__ __launch_bounds__(256) void k_automorph(u64 Q, u64* out, const u64* a, int t) {
  const int z = blockIdx.y * 256 + threadIdx.x;
  const size_t p = (size_t)blockIdx.x * N;
  unsigned zt = (unsigned)z * (unsigned)t;
  unsigned num = zt >> POLY_LEN_LOG2, rem = zt & (N - 1);
  u64 v = a[p + z];
  out[p + rem] = (num & 1u) ? Q - v : v;
}
void launch_automorph(const DevTables& T, u64* out, const u64* a, int n_polys, int t, hipStream_t s) {
  if (n_polys <= 0) return;
  hipLaunchKernelGGL(k_automorph, dim3(n_polys, N / 256), dim3(256), 0, s, T.c.Q, out, a, t);
  launched(0, "k_automorph");
}

__global__ __launch_bounds__(256) void k_gadget_raw(u64* out, const u64* inp, int cols, int rdim, int bits) {
  const int z = blockIdx.y * 256 + threadIdx.x;
  const int o = blockIdx.x;  // output poly index: row*cols + col
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
  hipLaunchKernelGGL(k_gadget_raw, dim3(rows_out * cols, N / 256), dim3(256), 0, s, out, inp, cols, rdim, bits);
  launched(0, "k_gadget_raw");
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
  launched(0, "k_u64_to_u32");
}
void launch_u32_to_u64(u64* out, const u32* in, long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_u32_to_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, in, n);
  launched(0, "k_u32_to_u64");
}

__global__ __launch_bounds__(256) void k_encode(EncodeDesc d) { encode_body(d, blockIdx.x); }
void launch_encode(const EncodeDesc& d, hipStream_t s) {
  const long total = (long)d.instances * ((long)d.n * N + (long)d.n * d.n * N);
  hipLaunchKernelGGL(k_encode, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
  launched(0, "k_encode");
}

// out[z][j][r] = v[ct_j][r][0][z] | v[ct_j][r][1][z] << 32   (util.rs:343-350; residues already < q)
__global__ __launch_bounds__(256) void k_reorient(ReorientDesc d) {
  __shared__ u64 tile[2 * 32 * 33];
  reorient_body(d, blockIdx.x, blockIdx.y, tile);
}
void launch_reorient(u64* out, const u32* v, int first, int step, int dim0, hipStream_t s) {
  const ReorientDesc d{out, v, first, step, dim0};
  hipLaunchKernelGGL(k_reorient, dim3((dim0 + 31) / 32, N / 32), dim3(256), 0, s, d);
  launched(0, "k_reorient");
}

__global__ __launch_bounds__(256) void k_reorient_group(ReorientDesc d, GroupOff g) {
  __shared__ u64 tile[2 * 32 * 33];
  const int qi = blockIdx.z;
  d.out = group_rebase(d.out, g.raw[qi]);
  d.v = group_rebase(d.v, g.v[qi]);
  reorient_body(d, blockIdx.x, blockIdx.y, tile);
}
void launch_reorient_group(u64* out, const u32* v, int first, int step, int dim0, const GroupOff& g, int B, hipStream_t s) {
  if (B <= 0) return;
  const ReorientDesc d{out, v, first, step, dim0};
  hipLaunchKernelGGL(k_reorient_group, dim3((dim0 + 31) / 32, N / 32, B), dim3(256), 0, s, d, g);
  launched(PATH_EXPAND_GROUP, "k_reorient_group");
}

// ---- placement probe (diagnostics): every workgroup records which XCC / shader engine / CU it ran on -------------
__global__ __launch_bounds__(64) void k_cu_probe(u32* out) {
  u32 xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // burn a little time so that the grid spreads over every CU the queue may use
  u32 acc = threadIdx.x;
  for (int i = 0; i < 20000; i++) acc = acc * 1664525u + 1013904223u;
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hw | ((acc & 1u) << 31);
  }
}
void launch_cu_probe(u32* out, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(k_cu_probe, dim3(blocks), dim3(64), 0, s, out);
  launched(0, "k_cu_probe");
}

}  // namespace spiral
