// Sparse buckets: lib/server's SparseDb caller (lib/server/src/db/sparse_db.rs:5-48,
// db/loading.rs:278-359, compute/dot_product.rs:13-220).  Only present items are stored (one packed polynomial per
// (item, plane)) and only they are multiplied: the first-dimension sweep costs time in proportion to the occupancy.
#include "device_common.hpp"

namespace spiral {

// ---- update_item_raw (loading.rs:317-359): one item -> `planes` packed NTT polynomials in its slot -------------------
// grid (planes): chunk `plane` of the item's bytes -> log2(p)-bit coefficients (util.rs:289-301; lib/server asserts
// 8 bits, loading.rs:290) -> recenter_mod (arith.rs:415-427) -> forward NTT mod q0, q1 -> lo | hi << 32.
__global__ __launch_bounds__(256) void k_sparse_item_encode(DevTables T, const uint8_t* bytes, int item_bytes,
                                                            int bytes_per_chunk, int logp, u32 pt_modulus, u64* slot_polys) {
  __shared__ u32 lds0[LDS_WORDS];
  __shared__ u32 lds1[LDS_WORDS];
  const int tau = threadIdx.x, plane = blockIdx.x;
  const int pos = plane * bytes_per_chunk;
  const int avail = item_bytes - pos;
  const int bytes_read = avail < 0 ? 0 : (avail < bytes_per_chunk ? avail : bytes_per_chunk);
  const int words_read = (bytes_read * 8 + logp - 1) / logp;
  u32 coeff[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int z = tau + 256 * k;
    u32 x = 0;
    if (z < words_read) {
      const int bit = z * logp, b0 = bit >> 3, sh = bit & 7, nb = (sh + logp + 7) >> 3;
      u64 acc = 0;
      for (int i = 0; i < nb; i++) acc |= (u64)(b0 + i < bytes_read ? bytes[pos + b0 + i] : 0) << (8 * i);
      x = (u32)((acc >> sh) & ((1ULL << logp) - 1ULL));
    }
    coeff[k] = x;
  }
  u32 lo[8];
  u32* la = lds0;
  u32* lb = lds1;
#pragma unroll 1
  for (int c = 0; c < 2; c++) {
    const ModConst m = T.c.mod[c];
    u32 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = coeff[k] > pt_modulus / 2 ? m.q - (pt_modulus - coeff[k]) : coeff[k];
    const u32* fw = T.tw + (size_t)c * 4 * N;
    if (c == 1) __syncthreads();
    ntt_fwd_block(v, tau, la, lb, fw, fw + N, m.q, m.two_q);
    if (c == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) lo[k] = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) slot_polys[(size_t)plane * N + 8 * tau + k] = (u64)lo[k] | ((u64)v[k] << 32);
    }
  }
}
void launch_sparse_item_encode(const DevTables& T, const uint8_t* bytes, int item_bytes, int bytes_per_chunk, int logp,
                               u32 pt_modulus, u64* slot_polys, int planes, hipStream_t s) {
  hipLaunchKernelGGL(k_sparse_item_encode, dim3(planes), dim3(256), 0, s, T, bytes, item_bytes, bytes_per_chunk, logp,
                     pt_modulus, slot_polys);
  launched(0, "k_sparse_item_encode");
}

// ---- multiply_reg_by_sparse_database (dot_product.rs:13-220) ----------------------------------------------------------
// grid (num_per, planes); thread tau owns z = tau + 256 k.  Column ii's present items are col_rows / col_slots
// [col_ptr[ii], col_ptr[ii+1]); item polynomials: polys[slot][plane][z] (lo | hi << 32); the query is read from the
// expanded ciphertexts themselves, v[ct][r][crt][z] with ct = first + step * j (contiguous in z, no reorientation).
// Exact sums: products < 2^56, Barrett fold every 256 items.  Absent columns produce zeros (the fold relies on that).
__global__ __launch_bounds__(256) void k_sweep_sparse(DevTables T, const int* col_ptr, const int* col_rows, const int* col_slots,
                                                      const u64* polys, int planes, const u32* v, int first, int step,
                                                      u32* out, int num_per) {
  const int tau = threadIdx.x, ii = blockIdx.x, plane = blockIdx.y;
  const ModConst m0 = T.c.mod[0], m1 = T.c.mod[1];
  u64 a[8][4];
#pragma unroll
  for (int k = 0; k < 8; k++) a[k][0] = a[k][1] = a[k][2] = a[k][3] = 0;
  const int e0 = col_ptr[ii], e1 = col_ptr[ii + 1];
  int since = 0;
  for (int e = e0; e < e1; e++) {
    const int j = col_rows[e];
    const u64* b = polys + ((size_t)col_slots[e] * planes + plane) * N;
    const u32* q = v + (size_t)(first + step * j) * 4 * N;  // [r][crt][z]
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int z = tau + 256 * k;
      const u64 w = b[z];
      const u32 bl = (u32)w, bh = (u32)(w >> 32);
      a[k][0] += (u64)q[z] * bl;           // r0 crt0
      a[k][1] += (u64)q[N + z] * bh;       // r0 crt1
      a[k][2] += (u64)q[2 * N + z] * bl;   // r1 crt0
      a[k][3] += (u64)q[3 * N + z] * bh;   // r1 crt1
    }
    if (++since == 255) {
      since = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        a[k][0] = reduce64(a[k][0], m0); a[k][1] = reduce64(a[k][1], m1);
        a[k][2] = reduce64(a[k][2], m0); a[k][3] = reduce64(a[k][3], m1);
      }
    }
  }
  // out[plane][r][crt][z][ii]
  const size_t rc = (size_t)N * num_per;
  u32* o = out + (size_t)plane * 4 * rc + ii;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const size_t z = tau + 256 * k;
    o[0 * rc + z * num_per] = reduce64(a[k][0], m0);
    o[1 * rc + z * num_per] = reduce64(a[k][1], m1);
    o[2 * rc + z * num_per] = reduce64(a[k][2], m0);
    o[3 * rc + z * num_per] = reduce64(a[k][3], m1);
  }
}
void launch_sweep_sparse(const DevTables& T, const int* col_ptr, const int* col_rows, const int* col_slots, const u64* polys,
                         int planes, const u32* v, int first, int step, u32* out, int num_per, hipStream_t s) {
  hipLaunchKernelGGL(k_sweep_sparse, dim3(num_per, planes), dim3(256), 0, s, T, col_ptr, col_rows, col_slots, polys, planes, v,
                     first, step, out, num_per);
  launched(PATH_SWEEP_SPARSE, "k_sweep_sparse");
}

}  // namespace spiral
