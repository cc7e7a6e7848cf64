// Launch wrappers for the gfx950 kernels of the Spiral answer path (ntt.hip, fold.hip, elementwise.hip, sweep.hip, db.hip; shared device helpers in device_common.hpp).
// Device data model:
//   NTT-form poly   : u32[2][N]  ([crt][z], residues < q_crt)           -- "npoly", 16 KiB
//   raw poly        : u64[N]     (coefficients <= Q < 2^56)              -- "rpoly", 16 KiB
//   matrices        : row-major arrays of polys, as the reference's PolyMatrix (poly.rs:31-35)
// All launches are asynchronous on the given stream.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "params.hpp"

namespace spiral {

constexpr int N = (int)POLY_LEN;

struct DevTables {
  const u32* tw;  // [crt][4][N] twiddles as the reference's (0 fwd, 1 fwd', 2 inv, 3 inv'), then [crt][2][N] (inv_tables), then [crt][2][N] (wave_fwd_image)
  DevConsts c;
};
// [w | w'] of modulus c for the kernels' inverse transform: unhalved psi^-i, entries 0 and 1 times N^-1 (params.cpp, finish)
__host__ __device__ inline const u32* inv_tables(const u32* tw, int c) { return tw + (size_t)8 * N + (size_t)c * 2 * N; }
// The forward tables of modulus c as the wave-per-transform NTT keeps them in LDS (wave_ntt.hpp): [ -w | w' ] with the entries
// of the last two stages rotated per lane group (wtw_phys), 2N words -- a workgroup stages them with plain 16-byte copies
// (wtw_stage) instead of negating and permuting 16 KiB per fold step and modulus (7.7 % of k_fold_wave: profiles/r05_fold_dissection.md)
__host__ __device__ inline const u32* wave_fwd_image(const u32* tw, int c) { return tw + (size_t)12 * N + (size_t)c * 2 * N; }
// physical word of twiddle-table entry idx in that image: the ranges a lane reads as b128 vectors in stages t = 2
// (entries 512 + 8L + 0..7) and t = 1 (1024 + 16L + 0..15) are rotated per lane group so that the 16 lanes of a b128
// service group hit 16 different bank quads
__host__ __device__ inline int wtw_phys(int idx) {
  if (idx >= 1024) {
    const int L = (idx - 1024) >> 4, r = (idx - 1024) & 15;
    return 1024 + 16 * L + 4 * (((r >> 2) + (L >> 2)) & 3) + (r & 3);
  }
  if (idx >= 512) {
    const int L = (idx - 512) >> 3, r = (idx - 512) & 7;
    return 512 + 8 * L + 4 * (((r >> 2) + (L >> 3)) & 1) + (r & 3);
  }
  return idx;
}

// ---- which kernels / flows the calling thread's work went through (sp_paths_taken, include/spiral_hip.h) ----------
// Every launch wrapper reports here right after its hipLaunchKernelGGL: the bit is recorded (thread-local, tests assert
// which code path produced the bytes they compared) and the launch is checked -- a rejected launch (bad grid, too
// much LDS ...) throws HipError instead of leaving stale workspace contents to be packed into a "successful" response.
enum PathBit : u64 {
  PATH_SWEEP_PERSIST = 1ull << 0,     // k_sweep_packed_persist / k_sweep_packed_ring (PACKED database, persistent grid)
  PATH_SWEEP_PACKED = 1ull << 1,      // (retired: k_sweep_packed, one wave per unit)
  PATH_SWEEP_WIDE = 1ull << 2,        // k_sweep_wide (8-byte words, num_per >= 128)
  PATH_SWEEP_NARROW = 1ull << 3,      // k_sweep_narrow / k_sweep_narrow2 (num_per < 128, LDS-staged query)
  PATH_SWEEP_BATCH = 1ull << 4,       // k_sweep_packed_batch (several queries per database pass)
  PATH_FROM_SWEEP4 = 1ull << 5,       // k_from_sweep4 (4 columns per workgroup)
  PATH_FROM_SWEEP1 = 1ull << 6,       // k_ntt_inv in sweep-source mode
  PATH_FOLD_FUSED = 1ull << 7,        // k_fold_fused* (one workgroup per fold step)
  PATH_FOLD_TAIL = 1ull << 8,         // three-launch tree tail (delta form)
  PATH_FOLD_TAIL_LITERAL = 1ull << 9, // three-launch tree tail (two-matrix form)
  PATH_PIPELINED = 1ull << 10,        // per-plane sweep launches with the fold of plane p beside the sweep of p+1
  PATH_EXPAND_PRUNED = 1ull << 11,    // expansion pruned to a row shard's rows
  PATH_PACK_V1 = 1ull << 12,
  PATH_DIRECT_UPLOAD = 1ull << 13,
  PATH_SCATTER_OUT = 1ull << 14,      // column-interleaved sweep output (multi-GPU reduce-scatter layout)
  PATH_SWEEP_XCD_FROM = 1ull << 15,   // k_from_sweep4 with the XCD-aware block order
  PATH_FOLD_TAIL_PERSIST = 1ull << 16,// (retired: k_fold_tail of round 1)
  PATH_EXPAND_FUSED = 1ull << 17,     // k_expand_round: a large expansion round's digit transforms and products in one launch (r05)
  PATH_SWEEP_SPARSE = 1ull << 18,     // presence-aware sweep (absent units skipped)
  PATH_RCCL = 1ull << 19,             // RCCL collectives issued by the library itself (sp_comm_create)
  PATH_FOLD_WAVE = 1ull << 20,        // k_fold_wave (wave-per-transform NTT, no workgroup barriers inside a transform)
  PATH_CU_SPLIT = 1ull << 21,         // (retired: sweeps and overlapped folds on disjoint CU sets)
  PATH_EXPAND_SPLIT = 1ull << 22,     // odd expansion subtree + GSW side on the second stream, beside the even subtree
  PATH_PIPE_CLASS_SPLIT = 1ull << 23, // (retired: a plane swept and folded as two chunk-parity classes)
  PATH_SWEEP_MFMA = 1ull << 24,       // k_sweep_mfma_batch (batched sweep on the matrix cores, signed base-256 digits)
  PATH_CUSTOM_TRANSPORT = 1ull << 25, // sharded query whose collectives were the host's (sp_comm_create_custom), not RCCL
  PATH_FROM_SWEEP_WAVE = 1ull << 26,  // (retired: k_from_sweep_wave)
  PATH_FOLD_TAIL_BATCHED = 1ull << 27,// pipelined query: the planes' small fold levels deferred and run as one batch
  PATH_SWEEP_RING = 1ull << 28,       // k_sweep_packed_ring (persistent sweep, two buffers of row pairs per wave)
  PATH_SWEEP_MFMA2 = 1ull << 29,      // k_sweep_mfma_batch with two query tiles (9 .. 16 queries per database pass)
  PATH_FOLD_WAVE8 = 1ull << 30,       // (retired in the round that built it: k_fold_wave8, profiles/r05_fold_wave8.md)
  PATH_SWEEP_PLANAR = 1ull << 31,     // k_sweep_planar: the 9 .. 16-query pass over the digit-planar copy of the database
  PATH_EXPAND_GROUP = 1ull << 32,     // a group's expansions with every round's launches shared (grid dimension = query; r06)
  PATH_EXPAND_WAVE = 1ull << 33       // k_expand_wave: a round's many-digit side on the wave-per-transform NTT (r06)
};
// Run-time tunables (sp_debug_set / environment SPIRAL_<NAME>): read on every launch, so that variants can be A/B
// measured inside one process on ONE database allocation (HBM placement alone moves the sweep by +-5 %).
long tunable(const char* name, long dflt);  // server.cpp
void launched(u64 path_bits, const char* kernel);  // server.cpp
void note_path(u64 path_bits);
void launch_checksum(const u32* p, size_t n_words, unsigned long long* out, hipStream_t s);  // sum p[i] * (2i + 1)
void launch_cache_sync(u32* sink, hipStream_t s);  // every XCD: L2 write-back + invalidate (system-scope fences)
void launch_copy_words(u32* dst, const u32* src, size_t n_words, hipStream_t s);  // elementwise.hip
void debug_stage(int stage);  // 1 expansion, 2 sweep, 3 fold/pack/encode (debug_sync, server.cpp)

// ---- forward NTT family -------------------------------------------------------------------
// Generic source descriptor for a batch of forward NTTs: output poly `o` (0 <= o < n_out) is the
// NTT of digit `k` of raw poly `src`, where with out matrix (rdim*t x cols) per batch element
// (gadget.rs:34-60):  b = o / (rdim*t*cols), row = (o / cols) % (rdim*t), col = o % cols,
// k = row / rdim, j = row % rdim, src poly = b*src_batch_stride + (src_row0 + j)*src_cols + col.
// t = 1, bits = 64 gives plain to_ntt (poly.rs:613-623: value reduced mod q_crt first).
struct FwdDesc {
  const u64* src;      // raw polys
  const int* src_idx;  // optional: batch element b reads from src + src_idx[b]*src_batch_stride polys
  u32* dst;            // n_out NTT polys, dense
  int n_out;
  int rdim, cols, t, bits;
  int src_batch_stride;  // in polys
  int src_row0, src_cols;
  // delta mode (fold tree tail): output = NTT(digit_k(src2 poly) - digit_k(src poly) mod q); src2 poly =
  // src poly + delta_off polys, where batch element b = (outer, inner) with inner < delta_inner reads
  // src batch index outer * delta_outer_stride + inner
  long delta_off;
  int delta_inner, delta_outer_stride;
};
void launch_ntt_fwd(const DevTables& T, const FwdDesc& d, hipStream_t s);
// up to three descriptors in ONE launch (expansion rounds: left digits, right digits, row-1 NTT)
void launch_ntt_fwd3(const DevTables& T, const FwdDesc& d0, const FwdDesc& d1, const FwdDesc& d2, hipStream_t s);

// micro-benchmark of the transform core: returns milliseconds for blocks*reps*M transforms
float bench_ntt_core(const DevTables& T, int M, int blocks, int reps, u32* scratch, hipStream_t s);

// ---- inverse NTT + CRT compose (poly.rs:646-663) -------------------------------------------
// Source element (poly p, crt c, coefficient z) is read from
//   src[(idx ? idx[p / polys_per_idx] * idx_stride + (p % polys_per_idx) * poly_stride : p * poly_stride)
//       + c * crt_stride + z * z_stride]                      (all strides in u32 words)
// premod: source values are sums of up to 8 residues (multi-GPU partials) -> reduce mod q first.
struct InvDesc {
  const u32* src;
  const int* idx;
  int polys_per_idx;
  long idx_stride, poly_stride, crt_stride, z_stride;
  u64* dst;  // n_polys raw polys, dense
  int n_polys;
  int premod;
  // sweep_np > 0: source is the sweep-native [plane][r][crt][z][ii] buffer with num_per = sweep_np;
  // poly p = (plane*num_per + ii)*2 + r (idx/strides ignored)
  int sweep_np;
  // optional fused automorphism (poly.rs:393-405) applied to the raw result: dst[(z*t) % N] = +-v
  int automorph_t;  // 0 = none
  // optional fused scalar multiply (coefficient_expansion, server.rs:105-110): ciphertexts idx[e] >= scal_thresh
  // are first formed as scal * v[idx[e] - scal_thresh] (stored to scal_dst, the writable alias of src) and then
  // inverse-transformed; n_scalar_only further ciphertexts (scal_only_idx) are only formed and stored.
  const u32* scal;
  u32* scal_dst;
  int scal_thresh;
  const int* scal_only_idx;
  int n_scalar_only;
  // optional: dst = (result + addend poly) mod Q, addend poly index = (p / add_inner2) * add_outer_stride + p % add_inner2
  const u64* addend;
  int add_inner2, add_outer_stride;
};
void launch_ntt_inv(const DevTables& T, const InvDesc& d, hipStream_t s);
// from_ntt of the sweep-native buffer [plane][r][crt][z][ii] (num_per = np, np % 4 == 0), four adjacent
// columns per workgroup (16-byte loads: a quarter of the cache-line traffic of the one-column form);
// dst raw polys dense in the same order as InvDesc's sweep mode: poly (plane*np + ii)*2 + r.
void launch_from_sweep4(const DevTables& T, const u32* src, int np, int n_planes, int premod, u64* dst, hipStream_t s);

// ---- NTT-domain multiply-accumulate (poly.rs:437-481) --------------------------------------
// out[b][r] = (addend ? addend[b][r] : 0) + sum_k A[r][k] * B[b][k]   (pointwise, per crt), r < R
//   A: R x K polys, shared by the batch.
//   B operand k of batch element (outer, inner): poly
//     outer*B_outer_stride + inner*B_inner_stride + (k < split_k ? k : split_off + (k - split_k))
//   out / addend poly index: (out_idx ? out_idx[b] : b * out_batch_stride) + r * out_row_stride,
//   b = outer*batch_inner + inner.
struct MacDesc {
  const u32* A;
  const u32* B;
  u32* out;
  const u32* addend;  // may alias out
  const int* out_idx;
  int R, K;
  int A_row_stride;  // polys between consecutive rows of A; 0 = K (dense)
  int batch_inner, batch_outer;
  long B_inner_stride, B_outer_stride;  // in polys
  int split_k;
  long split_off;  // in polys
  int out_batch_stride, out_row_stride;
  // optional second addend for output row `extra_row`: NTT poly extra[extra_idx ? extra_idx[b] : b]
  const u32* extra;
  const int* extra_idx;
  int extra_row;
};
void launch_mac(const DevTables& T, const MacDesc& d, hipStream_t s);
// two descriptors in ONE launch (both must have batch_outer == 1)
void launch_mac2(const DevTables& T, const MacDesc& d0, const MacDesc& d1, hipStream_t s);

// ---- one expansion round's digit transforms AND their products in one launch (server.rs:89-102) ------------------------------
// For each ciphertext b of a side: v[out_idx[b] + r] += sum_k A[r][k] * NTT(digit_k(raw[pos[b] * 2])) (+ NTT(raw[pos[b] * 2 + 1])
// for r = 1), r = 0, 1 -- what launch_ntt_fwd3 + launch_mac2 compute through the digit buffer, as ONE workgroup per (ciphertext,
// modulus): the source polynomial is read once, its digits are transformed four at a time and multiplied into the two rows' sums
// in registers; nothing is written but the two result polynomials.  Same residues (exact sums), so the same bytes.
// A round's latency is then t / 4 + 1 transform passes in sequence: for rounds large enough to fill the chip (expand_round_min).
struct ExpandSideDesc {
  const u64* raw;      // the round's automorphed ciphertexts, raw: polys [pos * 2 + row]
  const int* pos;      // [cnt] position of ciphertext b in that list
  const int* out_idx;  // [cnt] poly index of the output row 0 in v (row 1 follows)
  const u32* A;        // 2 x t NTT polys (the side's expansion key of this round)
  u32* v;              // NTT polys, read (addend) and written
  int cnt, t, bits;
};
void launch_expand_round(const DevTables& T, const ExpandSideDesc& left, const ExpandSideDesc& right, hipStream_t s);
constexpr long EXPAND_ROUND_MIN_DEFAULT = 2048;

// ---- the expansions of a GROUP of queries with every round's launches shared (batched steps, r06) ------------------------------
// The queries of a group have the same Params (same index lists, same schedule) but each its own workspace buffers and its own
// public parameters.  A grouped launch is the single-query launch with one more grid dimension = the query; its descriptors are
// query 0's, and query qi's pointers are query 0's plus a BYTE offset per class of buffer: v (the ciphertext tree, NTT form), raw
// (the round's automorphed ciphertexts), dig (their digit transforms), ct1 (the transforms of their second rows), pp (the public
// parameters' polynomials).  A round of 16 queries then fills the chip from round 2 or 3 on, where sixteen separate chains each
// left most of it idle behind four hardware queues (profiles/r06_group_expansion.md).
constexpr int GROUP_MAX = 16;   // == SWEEP_GROUP_MAX
struct GroupOff {
  long long v[GROUP_MAX], raw[GROUP_MAX], dig[GROUP_MAX], ct1[GROUP_MAX], pp[GROUP_MAX];
};
void launch_ntt_inv_group(const DevTables& T, const InvDesc& d, const GroupOff& g, int B, hipStream_t s);      // src, scal_dst: v; dst: raw
void launch_ntt_fwd3_group(const DevTables& T, const FwdDesc& d0, const FwdDesc& d1, const FwdDesc& d2, const GroupOff& g, int B,
                           hipStream_t s);                                                                         // src: raw; dst: dig, dig, ct1
void launch_mac2_group(const DevTables& T, const MacDesc& d0, const MacDesc& d1, const GroupOff& g, int B, hipStream_t s);   // A: pp; B: dig; out, addend: v; extra: ct1
void launch_expand_round_group(const DevTables& T, const ExpandSideDesc& left, const ExpandSideDesc& right, const GroupOff& g, int B,
                               hipStream_t s);                                                                     // raw: raw; A: pp; v: v
// ... and what follows the rounds (v_reg, regev_to_gsw, G - C, wave layout of the fold operands).  The transform and multiply
// launches above serve again with other buffers behind their offset classes; four small kernels get grouped forms of their own
// (classes named per launcher: "v" = the source side, "raw" = the destination side):
void launch_reorient_group(u64* out, const u32* v, int first, int step, int dim0, const GroupOff& g, int B, hipStream_t s);       // out: raw; v: v
void launch_copy_polys_group(u32* dst, const int* dst_idx, int dst_row_stride, const u32* src, const int* src_idx, int src_row_stride,
                             int R, int batch, const GroupOff& g, int B, hipStream_t s);                                            // dst: raw; src: v
void launch_folding_neg_group(const DevTables& T, u32* mats, const u32* gadget_ntt, int nu2, int two_t, const GroupOff& g, int B,
                              hipStream_t s);                                                                                       // mats: v
void launch_mats_to_wave_group(u32* dst, const u32* src, size_t n_words, int half_polys, const GroupOff& g, int B, hipStream_t s);                  // dst: raw; src: v
// ... and the many-digit (right-hand) side of a large round on the wave-per-transform NTT (fold.hip, k_expand_wave): ciphertexts
// pos[0 .. cnt) of `raw`, t digits of `bits` bits, expansion key of the round in WAVE layout (A_w: the polynomials of
// ExpandSideDesc::A through k_mats_to_wave), const_w: the constant polynomials 0 and 1 (N words each)
struct ExpandWaveDesc {
  const u64* raw;
  const int* pos;
  const int* out_idx;
  const u32* A_w;
  const u32* const_w;
  u32* v;
  int cnt, t, bits;
};
void launch_expand_wave(const DevTables& T, const ExpandWaveDesc& d, const GroupOff& g, int B, hipStream_t s);   // raw: raw; A_w: pp; v: v
constexpr long EXPAND_WAVE_MIN_DIGITS_DEFAULT = 16;     // sides with at least this many digits per ciphertext take the wave kernel
constexpr long EXPAND_GROUP_ROUND_MIN_DEFAULT = 4096;   // digit transforms per modulus of the WHOLE group from which a round is one launch

// ---- fused fold step (server.rs:407-424) ----------------------------------------------------
// One workgroup per (pair i, plane): out[plane][i] = from_ntt( [G-C | C] * NTT(G^-1([ct_i ; ct_{i+half}])) ),
// digits -> NTT -> multiply-accumulate -> iNTT -> CRT entirely in registers/LDS.
//   X: raw cts dense [plane][cur][2][N];  Y: raw cts dense [plane][half][2][N];  mats: this level's 2 x 4t polys
struct FoldDesc {
  const u64* X;
  u64* Y;
  const u32* mats;
  int cur, half, planes;
  int t, bits;
  // digits that can be non-zero: a coefficient below Q < 2^modulus_log2 has only ceil(modulus_log2 / bits) of them (8-bit digits
  // of a 56-bit Q: 7 of t_gsw = 8 -- the top digit polynomial of G^-1 (gadget.rs:34-60) is identically zero, its transform is
  // zero and adds nothing to the products).  The query flows, whose ciphertexts are from_ntt / fold outputs and therefore < Q,
  // set it; with caller-supplied ciphertexts (stage-level entry points) it equals t.  0 means t.
  int t_live;
  // lib/server semantics (lib/server/src/compute/fold.rs:38-44): an all-zero ct_i is replaced by ct_{i+half}, an
  // all-zero ct_{i+half} leaves ct_i as it is -- no external product in either case
  int zero_shortcuts;
  // the level's operands in wave layout (k_fold_wave); nullptr: not available
  const u32* mats_w;
};
// fold_mats -> wave layout (wave_ntt.hpp wave_layout_word), n_words = polynomials * 2 * N
void launch_mats_to_wave(u32* dst, const u32* src, size_t n_words, hipStream_t s, int half_polys = 0);
// SPIRAL_FOLD_VARIANT: 5 = k_fold_wave (wave-per-transform NTT; used while two workgroups fit a CU's LDS, else falls
// through), 3 = k_fold_fused2 (two cooperative transforms per pass, even t_gsw, twiddles in LDS), 0 = k_fold_fused.
// profiles/r02_fold_batch_experiments.md has the measurements behind the default.
constexpr long FOLD_VARIANT_DEFAULT = 5;
void launch_fold_fused(const DevTables& T, const FoldDesc& d, hipStream_t s);

// dst poly idx[b] += src poly b   (NTT polys, mod q)
void launch_add_poly_into(const DevTables& T, u32* dst, const int* idx, const u32* src, int batch, hipStream_t s);
// polys [dst_off + b] = scalar (1 poly) * polys [src_off + b], b < n_polys   (multiply_poly, poly.rs:351-358)
void launch_scalar_mul(const DevTables& T, u32* base, long dst_off, long src_off, const u32* scalar, int n_polys,
                       hipStream_t s);
// dst poly di[k] = a poly ai[k] + b poly bi[k]  (mod q), k < count; a/b/dst may alias
void launch_add_polys_idx(const DevTables& T, u32* dst, const int* di, const u32* a, const int* ai, const u32* b,
                          const int* bi, int count, hipStream_t s);
// direct-upload queries (client.rs:105-128): qv[z][j][0] = lo | hi << 32 of NTT poly j (dense [dim0] polys),
// qv[z][j][1] = wire[z*dim0 + j]
void launch_interleave_query(u64* qv, const u32* row0_ntt, const u64* wire, int dim0, hipStream_t s);
// gather copy of NTT polys: dst poly (dst_idx[b] + r*dst_row_stride) = src poly (src_idx[b] + r*src_row_stride), r < R
void launch_copy_polys(u32* dst, const int* dst_idx, int dst_row_stride, const u32* src, const int* src_idx,
                       int src_row_stride, int R, int batch, hipStream_t s);
// v_folding_neg = G - C in the NTT domain (== server.rs:505-523): for each GSW ct d,
// mats[d][r][0..2t) = gadget_ntt[r][col] + q - mats[d][r][2t + col]; mats rows are 4t polys wide
void launch_folding_neg(const DevTables& T, u32* mats, const u32* gadget_ntt, int nu2, int two_t, hipStream_t s);
// out = a + b (mod q) over n_polys NTT polys (poly.rs:483-498)
void launch_add(const DevTables& T, u32* out, const u32* a, const u32* b, int n_polys, hipStream_t s);
// raw: out[i] = Q - a[i]  (poly.rs:387-391, gives Q for 0)
void launch_invert_raw(const DevTables& T, u64* out, const u64* a, long n_words, hipStream_t s);
// raw automorphism on n_polys polys (poly.rs:393-405)
void launch_automorph(const DevTables& T, u64* out, const u64* a, int n_polys, int t, hipStream_t s);
// raw digits (gadget.rs:34-60) without NTT: out[rows_out x cols] from inp[rows_in x cols]
void launch_gadget_raw(u64* out, const u64* inp, int rows_in, int cols, int rows_out, int rdim, int bits,
                       hipStream_t s);
// ref-layout conversions: u64 NTT poly words <-> u32 npolys
void launch_u64_to_u32(u32* out, const u64* in, long n, hipStream_t s);
void launch_u32_to_u64(u64* out, const u32* in, long n, hipStream_t s);

// diagnostics: out[2b] = HW_REG_XCC_ID, out[2b+1] = HW_REG_HW_ID of workgroup b
void launch_cu_probe(u32* out, int blocks, hipStream_t s);

// ---- encode (server.rs:470-503) on the device: rescale (arith.rs:429-444) + LSB-first bit packing -------------
// packed: [instances][(n+1) x n raw polys]; out: zeroed buffer of response_bytes/8 u64 words (atomicOr packing)
struct EncodeDesc {
  const u64* packed;
  unsigned long long* out;
  int instances, n;
  u64 Q, q1, q2;
  int q1_bits, q2_bits;
};
void launch_encode(const EncodeDesc& d, hipStream_t s);

// ---- query reorientation (util.rs:323-355) --------------------------------------------------
// v (NTT cts, 2 polys each) at ct indices first + step*j, j < dim0  ->  out[z][j][r] = lo | hi << 32
void launch_reorient(u64* out, const u32* v, int first, int step, int dim0, hipStream_t s);

// ---- descriptors of the small elementwise kernels ------------------------------------------------------------------
struct CopyPolysDesc {  // launch_copy_polys
  u32* dst;
  const int* dst_idx;
  int dst_row_stride;
  const u32* src;
  const int* src_idx;
  int src_row_stride, R, batch;
};
struct FoldingNegDesc {  // launch_folding_neg
  u32* mats;
  const u32* gadget_ntt;
  int two_t, nu2;
};
struct ReorientDesc {  // launch_reorient
  u64* out;
  const u32* v;
  int first, step, dim0;
};
struct MatsToWaveDesc {  // launch_mats_to_wave
  u32* dst;
  const u32* src;
  size_t n_words;
  // > 0: the polynomials are rows of [ G - C | C ] with half_polys polynomials per half; only the C halves are converted
  // (the query path's fold kernels read nothing else: server.cpp run_fold_operands)
  int half_polys;
};
struct AddPolyIntoDesc {  // launch_add_poly_into
  u32* dst;
  const int* idx;
  const u32* src;
  int batch;
};
struct CopyWordsDesc {  // launch_copy_words
  u32* dst;
  const u32* src;
  size_t n_words;
};

// ---- database sweep (server.rs:155-221) -----------------------------------------------------
// Device DB layout per plane: [z][j_local][ii] u64 (word = lo28 | hi28<<32): the reference's
// [z][ii][j] with the two inner axes swapped so that lanes run along ii.
// qv: reoriented query [z][dim0][2] u64 (reference layout); rows j0 .. j0+nj of it are used.
// out: u32 [plane][r][crt][z][ii] residues < q.
// PACKED device format (num_per >= 128 and nj even): the 56 significant bits of each word only.
// Unit = (plane, z, row pair jp, 128-wide ii chunk) = 1792 B: lane l of 64 owns the 4 words
// (row 2jp, ii 2l), (2jp, 2l+1), (2jp+1, 2l), (2jp+1, 2l+1); their 8 28-bit limbs
// (lo, hi of each, in that order) form a 224-bit little-endian string = 7 dwords; dwords 0-3 of all 64
// lanes are stored first (16 B per lane, one global_load_dwordx4), then dwords 4-6 (12 B per lane).
// Units are ordered [plane][z][chunk][jp] (device_common.hpp packed_unit_offset): the row pairs one sweep wave
// reads are one sequential stream.  12.5 % less HBM traffic than the 8-byte words.
struct SweepDesc {
  const u64* db;  // plane 0 of this shard
  const u64* qv;
  u32* out;
  int planes, num_per, dim0, j0, nj;
  int packed;
  // out_G > 1: column-interleaved output for the multi-GPU reduce-scatter -- chunk g = ii % out_G holds
  // [plane][r][crt][z][ii / out_G]; chunks are contiguous (chunk g goes to rank g)
  int out_G;
  // non-temporal (streaming) output stores: HBM writes mixed into the read stream cost 3-4x a read byte on this part, a
  // quarter less as streaming stores (scripts/ubench/rw_mix.hip); switch sweep_nt_store
  int nt_store;
};
// Column sharding (multi-GPU alternative to row sharding): a shard holds the columns ii = off + stride*i,
// i < num_per_local, of every row; kernels see the local column count, loaders map to the global index.
struct ColMap {
  int off = 0, stride = 1, np_global = 0;
};
inline bool db_can_pack(int num_per, int nj) { return num_per >= 128 && (nj % 2) == 0; }
inline size_t db_bytes(int planes, int num_per, int nj, bool packed) {
  return (size_t)planes * N * nj * num_per * (packed ? 7 : 8);
}
void launch_sweep(const DevTables& T, const SweepDesc& d, hipStream_t s);
// persistent PACKED sweep: the ring form on one workgroup per CU where the row-pair count allows, else the plain form on
// wgs_per_cu workgroups per CU
void launch_sweep_persist(const DevTables& T, const SweepDesc& d, int wgs_per_cu, hipStream_t s, int n_cus = 256);
// B queries against ONE pass over the (PACKED) database: every database word is multiplied into B
// accumulator sets.  qv[b] / out[b] as in SweepDesc.  B <= SWEEP_BATCH_MAX = 8 for the vector kernel and for one query
// TILE of the matrix-core kernel (16 query columns = the M dimension of one MFMA); the matrix-core kernel takes two tiles
// (r04): up to SWEEP_GROUP_MAX = 16 queries share the loads AND the digit extraction of every database word.
constexpr int SWEEP_BATCH_MAX = 8;
constexpr int SWEEP_GROUP_MAX = 16;
struct SweepBatchDesc {
  const u64* db;
  const u64* qv[SWEEP_GROUP_MAX];
  u32* out[SWEEP_GROUP_MAX];
  int batch;
  int planes, num_per, dim0, j0, nj;
  // matrix-core form (sweep_mfma.hpp): rq = scratch for the group's query digit table (sweep_batch_rq_words words),
  // filled by sweep_batch_prepare; use_mfma is set by it.  rq == nullptr: VALU kernel.
  u32* rq;
  int use_mfma;
  // digit-planar copy of the same database (sweep_planar.hpp; sp_db::planar, built on the first group of more than 8 queries):
  // the two-tile pass then runs k_sweep_planar.  nullptr: the PACKED kernels.
  const unsigned char* planar;
};
// does this shape have a digit-planar form (whole 64-row blocks, the z-row's query planes in LDS, whole 128-column chunks)?
bool sweep_planar_shape_ok(int num_per, int nj);
size_t sweep_planar_bytes(int planes, int num_per, int nj);
// PACKED database -> digit-planar copy (one-time, per database; planes * N * num_per * nj * 8 bytes)
void launch_packed_to_planar(unsigned char* planar, const u64* packed, int planes, int num_per, int nj, hipStream_t s);
// ... and the 8 entries per (plane, z) that hold one item (local row j, local column ii), after sp_db_update_item
void launch_planar_patch_item(unsigned char* planar, const u64* packed, int planes, int num_per, int nj, int j, int ii, hipStream_t s);
void launch_sweep_planar(const DevTables& T, const SweepBatchDesc& d, hipStream_t s);   // sweep_planar.hip
// does this shape / group size run on the matrix cores (switch batch_mfma, default on from batch_mfma_min = 4 queries)?
// Groups of more than SWEEP_BATCH_MAX queries exist only there.
bool sweep_batch_wants_mfma(const SweepBatchDesc& d);
// largest group one pass takes for this shape: SWEEP_GROUP_MAX where the two-tile matrix-core form applies (switch
// batch_mfma_tiles, default 2), else SWEEP_BATCH_MAX
int sweep_batch_group_max(int num_per, int nj);
inline int sweep_batch_tiles(int batch) { return batch > SWEEP_BATCH_MAX ? 2 : 1; }
// per tile: digit table [N][nj / 16][2][64][4]; then per tile the offset-correction table [N][2][16]
// (k_query_digits / k_query_offset_terms)
inline size_t sweep_batch_rq_words(int nj, int tiles = 1) {
  return (size_t)tiles * ((size_t)N * (size_t)(nj / 16) * 128 * 4 + (size_t)N * 32);
}
// once per group of queries, before the pass (all planes share the table): builds the digit table when the matrix-core
// form applies and d.rq is set
void sweep_batch_prepare(const DevTables& T, SweepBatchDesc& d, hipStream_t s);
void launch_sweep_batch(const DevTables& T, const SweepBatchDesc& d, hipStream_t s);
// reference layout -> device layout for a z-range of one plane: src [nz][num_per][dim0] (host-order
// words already on the device), dst plane base; keeps rows j0..j0+nj
void launch_db_relayout(u64* dst, int plane, const u64* src, int z0, int nz, int num_per, int dim0, int j0, int nj,
                        int packed, ColMap cm, hipStream_t s);
void launch_db_synth(u64* dst, u64 seed, int planes, int num_per, int dim0, int j0, int nj, int packed, ColMap cm,
                     hipStream_t s);
// read back words (plane, z, ii, j_local0 .. +count) of either device format into out[count] (device)
void launch_db_read(u64* out, const u64* db, int plane, int z, int ii, int jl0, int count, int num_per, int nj,
                    int packed, hipStream_t s);
// ---- database preprocessing on the device (server.rs:277-357 load_item_from_seek / load_db_from_seek) ----
// One workgroup per 2x2 quad of items {rows 2jp, 2jp+1} x {ii 2q, 2q+1} of one plane: plaintext bytes ->
// logp-bit coefficients (util.rs:289-301) -> recenter_mod (arith.rs:415-427) -> forward NTT mod q0, q1 ->
// lo | hi << 32 words written straight into the resident format (8-byte or PACKED).
struct DbEncodeDesc {
  const uint8_t* win;     // device window of the raw item file
  size_t win_item0;       // index of the first item in the window
  size_t win_bytes;       // bytes present in the window
  size_t file_len;        // total length of the raw file (reads past it yield zeros, server.rs:300-309)
  u64* db;
  int db_item_size, bytes_per_chunk, logp;
  u32 pt_modulus;
  int planes, num_per, dim0, j0, nj, packed;
  int jp0, njp;           // local row pairs [jp0, jp0 + njp) to encode
  // single-item update (lib/server/src/db/loading.rs:317-359 update_item_raw): only item `only_item`
  // (global index) is re-encoded, its three quad neighbours keep their resident words; -1 = encode all
  long only_item;
  int only_q;             // quad column (local ii / 2) of that item
  ColMap cm;              // num_per above is the LOCAL column count
};
void launch_db_encode(const DevTables& T, const DbEncodeDesc& d, hipStream_t s);

// ---- sparse buckets (lib/server SparseDb caller; sparse.hip) ----------------------------------------------------------
// one item's bytes (device) -> `planes` packed NTT polynomials at slot_polys[plane][z]
void launch_sparse_item_encode(const DevTables& T, const uint8_t* bytes, int item_bytes, int bytes_per_chunk, int logp,
                               u32 pt_modulus, u64* slot_polys, int planes, hipStream_t s);
// first-dimension multiply over the present items only (CSR by column); v = expanded NTT ciphertexts, row j at
// ct index first + step * j; out = sweep-native [plane][r][crt][z][ii]
void launch_sweep_sparse(const DevTables& T, const int* col_ptr, const int* col_rows, const int* col_slots, const u64* polys,
                         int planes, const u32* v, int first, int step, u32* out, int num_per, hipStream_t s);

// sweep-native out [plane][r][crt][z][ii] -> reference out[ii].data[r*2N + crt*N + z] (u64) for one plane
void launch_sweep_out_to_ref(u64* out, const u32* in, int num_per, hipStream_t s);

__host__ __device__ inline u64 synth_word(u64 seed, u64 idx) {
  u64 z = seed + 0x9E3779B97F4A7C15ULL * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  u64 lo = (u32)z % (u32)MODULUS_0;
  u64 hi = (u32)(z >> 32) % (u32)MODULUS_1;
  return lo | (hi << 32);
}

}  // namespace spiral
