// Device-side producers / readers of the resident database formats: re-layout of reference-order words, synthetic
// fill, read-back, and preprocessing of raw items (server.rs:223-357).
#include "device_common.hpp"

namespace spiral {

// reference [z][ii][j] -> device [z][j - j0][ii] for nz rows starting at z0 (32x32 LDS tile transpose)
__global__ __launch_bounds__(256) void k_db_relayout(u64* dst_plane, const u64* src, int z0, int num_per, int dim0,
                                                     int j0, int nj, ColMap cm) {
  __shared__ u64 tile[32][33];
  const int zl = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int jt = blockIdx.x * 32, it = blockIdx.y * 32;
  const u64* s = src + (size_t)zl * cm.np_global * dim0;
  u64* dpl = dst_plane + (size_t)(z0 + zl) * nj * num_per;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int ii = it + ty + 8 * i, j = jt + tx;
    if (ii < num_per && j < nj) tile[ty + 8 * i][tx] = canon_word(s[(size_t)(cm.off + cm.stride * ii) * dim0 + j0 + j]);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int j = jt + ty + 8 * i, ii = it + tx;
    if (ii < num_per && j < nj) dpl[(size_t)j * num_per + ii] = tile[tx][ty + 8 * i];
  }
}
// reference layout -> PACKED: one thread per (z, jp, chunk, lane); gathers its 4 words (one-time cost)
__global__ __launch_bounds__(256) void k_db_relayout_packed(u32* dst, int plane, const u64* src, int z0, int nz,
                                                            int num_per, int dim0, int j0, int nj, ColMap cm) {
  const int chunks = num_per >> 7, npairs = nj >> 1;
  const size_t total = (size_t)nz * npairs * chunks * 64;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    size_t t = i >> 6;
    const int chunk = (int)(t % chunks);
    t /= chunks;
    const int jp = (int)(t % npairs);
    const int zl = (int)(t / npairs);
    const int ii = chunk * 128 + 2 * lane;
    const u64* s = src + ((size_t)zl * cm.np_global + cm.off + cm.stride * ii) * dim0 + j0 + 2 * jp;
    const size_t nx = (size_t)cm.stride * dim0;  // next local column
    const u64 w00 = canon_word(s[0]), w10 = canon_word(s[1]), w01 = canon_word(s[nx]), w11 = canon_word(s[nx + 1]);
    u32* unit = dst + packed_unit_offset((size_t)plane * N + (z0 + zl), jp, chunk, npairs, chunks);
    pack_unit_lane(unit, lane, w00, w01, w10, w11);
  }
}
void launch_db_relayout(u64* dst, int plane, const u64* src, int z0, int nz, int num_per, int dim0, int j0, int nj,
                        int packed, ColMap cm, hipStream_t s) {
  if (cm.np_global == 0) cm.np_global = num_per;
  if (nz <= 0) return;
  if (packed) {
    hipLaunchKernelGGL(k_db_relayout_packed, dim3(4096), dim3(256), 0, s, reinterpret_cast<u32*>(dst), plane, src, z0,
                       nz, num_per, dim0, j0, nj, cm);
    launched(0, "k_db_relayout_packed");
  } else {
    u64* dst_plane = dst + (size_t)plane * N * nj * num_per;
    hipLaunchKernelGGL(k_db_relayout, dim3((nj + 31) / 32, (num_per + 31) / 32, nz), dim3(256), 0, s, dst_plane, src,
                       z0, num_per, dim0, j0, nj, cm);
    launched(0, "k_db_relayout");
  }
}

__global__ __launch_bounds__(256) void k_db_synth(u64* dst, u64 seed, int num_per, int dim0, int j0, int nj,
                                                  size_t total, ColMap cm) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    // device index i = ((zp * nj) + jl) * num_per + ii
    size_t ii = i % num_per;
    size_t t = i / num_per;
    size_t jl = t % nj;
    size_t zp = t / nj;
    size_t ref = (zp * cm.np_global + cm.off + cm.stride * ii) * dim0 + j0 + jl;
    dst[i] = synth_word(seed, ref);
  }
}
__global__ __launch_bounds__(256) void k_db_synth_packed(u32* dst, u64 seed, int num_per, int dim0, int j0, int nj,
                                                         size_t total_lanes, ColMap cm) {
  const int chunks = num_per >> 7, npairs = nj >> 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_lanes; i += (size_t)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    size_t t = i >> 6;  // unit index in memory order = (zp * chunks + chunk) * npairs + jp (packed_unit_offset)
    const int jp = (int)(t % npairs);
    const size_t t2 = t / npairs;
    const int chunk = (int)(t2 % chunks);
    const size_t zp = t2 / chunks;
    const size_t ii = (size_t)chunk * 128 + 2 * lane;
    const size_t r0 = (zp * cm.np_global + cm.off + cm.stride * ii) * dim0 + j0 + 2 * jp;  // (row 2jp, ii)
    const size_t r1 = r0 + (size_t)cm.stride * dim0;                                         // (row 2jp, ii+1)
    pack_unit_lane(dst + t * 448, lane, synth_word(seed, r0), synth_word(seed, r1), synth_word(seed, r0 + 1),
                   synth_word(seed, r1 + 1));
  }
}
void launch_db_synth(u64* dst, u64 seed, int planes, int num_per, int dim0, int j0, int nj, int packed, ColMap cm,
                     hipStream_t s) {
  if (cm.np_global == 0) cm.np_global = num_per;
  if (packed) {
    size_t lanes = (size_t)planes * N * (nj >> 1) * (num_per >> 7) * 64;
    hipLaunchKernelGGL(k_db_synth_packed, dim3(256 * 32), dim3(256), 0, s, reinterpret_cast<u32*>(dst), seed, num_per,
                       dim0, j0, nj, lanes, cm);
  } else {
    size_t total = (size_t)planes * N * nj * num_per;
    hipLaunchKernelGGL(k_db_synth, dim3(256 * 32), dim3(256), 0, s, dst, seed, num_per, dim0, j0, nj, total, cm);
  }
  launched(0, "k_db_synth");
}

__global__ void k_db_read(u64* out, const u64* db, int plane, int z, int ii, int jl0, int count, int num_per, int nj,
                          int packed) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int jl = jl0 + t;
  if (packed) {
    const int chunks = num_per >> 7, npairs = nj >> 1;
    const int chunk = ii >> 7, lane = (ii & 127) >> 1, iiofs = ii & 1;
    const u32* unit = reinterpret_cast<const u32*>(db) +
                      packed_unit_offset((size_t)plane * N + z, jl >> 1, chunk, npairs, chunks);
    out[t] = unpack_word(unit, lane, (jl & 1) * 2 + iiofs);
  } else {
    out[t] = db[(((size_t)plane * N + z) * nj + jl) * num_per + ii];
  }
}
void launch_db_read(u64* out, const u64* db, int plane, int z, int ii, int jl0, int count, int num_per, int nj,
                    int packed, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_db_read, dim3((count + 63) / 64), dim3(64), 0, s, out, db, plane, z, ii, jl0, count, num_per, nj,
                     packed);
  launched(0, "k_db_read");
}

// ------------------------------------------------------------------------------------------------
// database preprocessing (server.rs:277-357) -- grid (max(num_per/2,1), njp, planes)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 item_coeff(const DbEncodeDesc& d, size_t item, int chunk_idx, int z) {
  // load_item_from_seek: chunk bytes at item*db_item_size + chunk_idx*bytes_per_chunk, logp bits per coefficient
  const size_t pos = item * (size_t)d.db_item_size + (size_t)chunk_idx * d.bytes_per_chunk;
  if (pos >= d.file_len) return 0u;
  const size_t avail = d.file_len - pos;
  const int bytes_read = (int)(avail < (size_t)d.bytes_per_chunk ? avail : (size_t)d.bytes_per_chunk);
  const int words_read = (bytes_read * 8 + d.logp - 1) / d.logp;
  if (z >= words_read) return 0u;
  const size_t wpos = pos - d.win_item0 * (size_t)d.db_item_size;
  const int bit = z * d.logp;
  const int b0 = bit >> 3, sh = bit & 7;
  u64 acc = 0;
  const int nb = (sh + d.logp + 7) >> 3;
  for (int i = 0; i < nb; i++) {
    const int bi = b0 + i;
    const u64 byte = (bi < bytes_read && wpos + bi < d.win_bytes) ? (u64)d.win[wpos + bi] : 0ULL;
    acc |= byte << (8 * i);
  }
  return (u32)((acc >> sh) & ((1ULL << d.logp) - 1ULL));
}

__global__ __launch_bounds__(256) void k_db_encode(DevTables T, DbEncodeDesc d) {
  __shared__ u32 lds0[LDS_WORDS];
  __shared__ u32 lds1[LDS_WORDS];
  const int tau = threadIdx.x;
  // grid.x runs over (quad column, row pair): both are unbounded (nu_1 up to 20 on direct-upload configs), grid.y / z are not
  const int gx = d.only_item >= 0 ? 1 : (d.num_per + 1) / 2;
  const int qd = d.only_item >= 0 ? d.only_q : (int)(blockIdx.x % (unsigned)gx);
  const int jp = d.jp0 + (int)(blockIdx.x / (unsigned)gx), plane = blockIdx.z;
  u64 w[4][8];  // [row a * 2 + ii b][k]: words at z = 8 tau + k
  u32* la = lds0;
  u32* lb = lds1;
#pragma unroll 1
  for (int ab = 0; ab < 4; ab++) {
    const int a = ab >> 1, b = ab & 1;
    const int jl = 2 * jp + a, ii = 2 * qd + b;
    const bool valid = jl < d.nj && ii < d.num_per;
    // i = j * num_per + ii (server.rs:332-333), ii = global column of local column `ii`
    const size_t item = (size_t)(d.j0 + jl) * d.cm.np_global + d.cm.off + (size_t)d.cm.stride * ii;
    if (d.only_item >= 0 && (long)item != d.only_item) {
      // keep the neighbour's resident words
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int z = 8 * tau + k;
        u64 cur = 0;
        if (valid) {
          if (d.packed) {
            const int chunks = d.num_per >> 7, npairs = d.nj >> 1;
            const u32* unit = reinterpret_cast<const u32*>(d.db) +
                              packed_unit_offset((size_t)plane * N + z, jp, ii >> 7, npairs, chunks);
            cur = unpack_word(unit, (ii & 127) >> 1, a * 2 + b);
          } else {
            cur = d.db[(((size_t)plane * N + z) * d.nj + jl) * d.num_per + ii];
          }
        }
        w[ab][k] = cur;
      }
      continue;
    }
    u32 lo[8], hi[8];
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
      const ModConst m = T.c.mod[c];
      u32 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        u32 x = valid ? item_coeff(d, item, plane, tau + 256 * k) : 0u;
        // recenter_mod(x, p, Q) reduced mod q_c: values above p/2 are negative
        v[k] = x > d.pt_modulus / 2 ? m.q - (d.pt_modulus - x) : x;
      }
      const u32* fw = T.tw + (size_t)c * 4 * N;
      ntt_fwd_block(v, tau, la, lb, fw, fw + N, m.q, m.two_q);
      {
        u32* tmp = la;
        la = lb;
        lb = tmp;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (c == 0)
          lo[k] = v[k];
        else
          hi[k] = v[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) w[ab][k] = (u64)lo[k] | ((u64)hi[k] << 32);
  }
  const int jl0 = 2 * jp, ii0 = 2 * qd;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int z = 8 * tau + k;
    if (d.packed) {
      const int chunks = d.num_per >> 7, npairs = d.nj >> 1;
      u32* unit = reinterpret_cast<u32*>(d.db) +
                  packed_unit_offset((size_t)plane * N + z, jp, ii0 >> 7, npairs, chunks);
      pack_unit_lane(unit, (ii0 & 127) >> 1, w[0][k], w[1][k], w[2][k], w[3][k]);
    } else {
#pragma unroll
      for (int ab = 0; ab < 4; ab++) {
        const int jl = jl0 + (ab >> 1), ii = ii0 + (ab & 1);
        if (jl < d.nj && ii < d.num_per) d.db[(((size_t)plane * N + z) * d.nj + jl) * d.num_per + ii] = w[ab][k];
      }
    }
  }
}
void launch_db_encode(const DevTables& T, const DbEncodeDesc& d, hipStream_t s) {
  if (d.njp <= 0) return;
  const unsigned gx = d.only_item >= 0 ? 1u : (unsigned)((d.num_per + 1) / 2);
  hipLaunchKernelGGL(k_db_encode, dim3(gx * (unsigned)d.njp, 1, d.planes), dim3(256), 0, s, T, d);
  launched(0, "k_db_encode");
}

__global__ __launch_bounds__(256) void k_sweep_out_to_ref(u64* out, const u32* in, int num_per) {
  // out[ii][r][crt][z] <- in[r][crt][z][ii]
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)4 * N * num_per;
  if (i >= total) return;
  size_t z = i & (N - 1);
  size_t rc = (i >> POLY_LEN_LOG2) & 3;
  size_t ii = i >> (POLY_LEN_LOG2 + 2);
  out[i] = (u64)in[(rc * N + z) * num_per + ii];
}
void launch_sweep_out_to_ref(u64* out, const u32* in, int num_per, hipStream_t s) {
  size_t total = (size_t)4 * N * num_per;
  hipLaunchKernelGGL(k_sweep_out_to_ref, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, in, num_per);
  launched(0, "k_sweep_out_to_ref");
}

}  // namespace spiral
