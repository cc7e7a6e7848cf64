// TEST INFRASTRUCTURE: the device helper functions of sdk_amd/csrc (device_common.hpp, wave_ntt.hpp, bodies.hpp), compiled
// UNCHANGED for the host and run as emulated workgroups (emu_runtime.cpp), behind a small C interface for
// tests/test_device_bodies_emulated.py.  What it buys: the CPU suite checks the kernels' arithmetic, index patterns, LDS
// exchanges and wave swaps against the oracle before anything reaches a GPU.  What it cannot do: run the product (the
// sweep kernels, the kernels' cross-wave reductions and every launch wrapper need gfx950), or say anything about speed.
#include <cstring>
#include <stdexcept>
#include <string>

#include "emu_runtime.hpp"
// product headers, as they are
#include "bodies.hpp"
#include "wave_ntt.hpp"

using namespace spiral;

namespace {
struct EmuParams {
  Params p;
  DevTables T;
};
thread_local std::string g_err;
template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}
// LDS of the emulated workgroup (one copy per host thread of the emulation = per workgroup in flight)
thread_local u32 g_ldsA[2 * LDS_WORDS], g_ldsB[2 * LDS_WORDS];
thread_local u32 g_wbuf[4][WBUF_WORDS];
thread_local u32 g_ltw[2 * N];
}  // namespace

extern "C" {

const char* emu_last_error() { return g_err.c_str(); }

void* emu_params_new(const char* json) {
  EmuParams* e = nullptr;
  if (guarded([&] {
        e = new EmuParams{Params::from_json(json), {}};
        e->T.tw = e->p.ntt_tables.data();
        e->T.c = e->p.dc;
      }))
    return nullptr;
  return e;
}
void emu_params_free(void* h) { delete (EmuParams*)h; }

// ntt_fwd_block / ntt_inv_block on data[N] of modulus c, indexed by coefficient (forward: natural in, reference order out)
int emu_ntt_block(void* h, int c, int inverse, uint32_t* data) {
  return guarded([&] {
    const EmuParams& E = *(EmuParams*)h;
    const ModConst m = E.T.c.mod[c];
    const u32* tw = inverse ? inv_tables(E.T.tw, c) : E.T.tw + (size_t)c * 4 * N;  // the kernels' inverse tables are unhalved (kernels.hpp)
    emu::run_block(256, [&] {
      const int tau = threadIdx.x;
      u32 v[8];
      if (!inverse) {
        for (int k = 0; k < 8; k++) v[k] = data[tau + 256 * k];
        ntt_fwd_block(v, tau, g_ldsA, g_ldsB, tw, tw + N, m.q, m.two_q);
        __syncthreads();
        for (int k = 0; k < 8; k++) data[8 * tau + k] = v[k];
      } else {
        for (int k = 0; k < 8; k++) v[k] = data[8 * tau + k];
        ntt_inv_block(v, tau, g_ldsA, g_ldsB, tw, tw + N, m.q, m.two_q);
        __syncthreads();
        for (int k = 0; k < 8; k++) data[tau + 256 * k] = v[k];
      }
    });
  });
}

// the two-at-a-time forms (ntt_fwd_block_m<2> / ntt_inv_block_m<2>) on data[2][N]
int emu_ntt_block_m2(void* h, int c, int inverse, uint32_t* data) {
  return guarded([&] {
    const EmuParams& E = *(EmuParams*)h;
    const ModConst m = E.T.c.mod[c];
    const u32* tw = inverse ? inv_tables(E.T.tw, c) : E.T.tw + (size_t)c * 4 * N;  // the kernels' inverse tables are unhalved (kernels.hpp)
    emu::run_block(256, [&] {
      const int tau = threadIdx.x;
      u32 v[2][8];
      if (!inverse) {
        for (int mm = 0; mm < 2; mm++)
          for (int k = 0; k < 8; k++) v[mm][k] = data[mm * N + tau + 256 * k];
        ntt_fwd_block_m<2>(v, tau, g_ldsA, g_ldsB, tw, tw + N, m.q, m.two_q);
        __syncthreads();
        for (int mm = 0; mm < 2; mm++)
          for (int k = 0; k < 8; k++) data[mm * N + 8 * tau + k] = v[mm][k];
      } else {
        for (int mm = 0; mm < 2; mm++)
          for (int k = 0; k < 8; k++) v[mm][k] = data[mm * N + 8 * tau + k];
        ntt_inv_block_m<2>(v, tau, g_ldsA, g_ldsB, tw, tw + N, m.q, m.two_q);
        __syncthreads();
        for (int mm = 0; mm < 2; mm++)
          for (int k = 0; k < 8; k++) data[mm * N + tau + 256 * k] = v[mm][k];
      }
    });
  });
}

// wntt_inv: four waves, each its own polynomial of modulus c: data[4][N] by coefficient index
int emu_wave_ntt_inv(void* h, int c, uint32_t* data) {
  return guarded([&] {
    const EmuParams& E = *(EmuParams*)h;
    const ModConst m = E.T.c.mod[c];
    const u32* itw = inv_tables(E.T.tw, c);
    emu::run_block(256, [&] {
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
      u32* poly = data + (size_t)wv * N;
      u32 v[32];
      for (int k = 0; k < 32; k++) v[k] = poly[32 * lane + k];
      __syncthreads();
      wntt_inv(v, lane, g_wbuf[wv], itw, m.q, m.two_q);
      for (int k = 0; k < 32; k++) poly[64 * k + lane] = v[k];
    });
  });
}

// wntt_fwd: four waves, each its own polynomial of modulus c (values < 2q allowed in); canon = the CANON form (canonical out),
// otherwise the lazy form (< 12q out, same residues)
int emu_wave_ntt_fwd(void* h, int c, int canon, uint32_t* data) {
  return guarded([&] {
    const EmuParams& E = *(EmuParams*)h;
    const ModConst m = E.T.c.mod[c];
    const u32* tw = E.T.tw + (size_t)c * 4 * N;
    emu::run_block(256, [&] {
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
      u32* poly = data + (size_t)wv * N;
      wtw_stage(g_ltw, wave_fwd_image(E.T.tw, c), threadIdx.x);
      WaveScalarTw s;
      wntt_scalar_tw(s, tw);
      u32 v[32];
      for (int k = 0; k < 32; k++) v[k] = poly[64 * k + lane];
      __syncthreads();
      WaveNoHooks hk;
      if (canon)
        wntt_fwd<true>(v, lane, g_wbuf[wv], tw, s, g_ltw, m.q, m.two_q, hk);
      else
        wntt_fwd<false>(v, lane, g_wbuf[wv], tw, s, g_ltw, m.q, m.two_q, hk);
      for (int k = 0; k < 32; k++) poly[32 * lane + k] = v[k];
    });
  });
}

// from_ntt (poly.rs:646-663) through ntt_inv_body: n polys, dense [poly][crt][N] u32 source, optional automorphism
int emu_from_ntt(void* h, const uint32_t* src, int n, int automorph_t, uint64_t* dst) {
  return guarded([&] {
    const EmuParams& E = *(EmuParams*)h;
    InvDesc d{};
    d.src = src;
    d.poly_stride = 2 * N;
    d.crt_stride = N;
    d.z_stride = 1;
    d.dst = dst;
    d.n_polys = n;
    d.automorph_t = automorph_t;
    for (int blk = 0; blk < n; blk++) emu::run_block(256, [&] { ntt_inv_body(E.T, d, blk, g_ldsA, g_ldsB); });
  });
}
// the same from the sweep-native buffer [plane][r][crt][z][ii] (num_per = np), sums of residues allowed when premod
int emu_from_sweep(void* h, const uint32_t* src, int np, int planes, int premod, uint64_t* dst) {
  return guarded([&] {
    const EmuParams& E = *(EmuParams*)h;
    InvDesc d{};
    d.src = src;
    d.dst = dst;
    d.n_polys = planes * np * 2;
    d.premod = premod;
    d.sweep_np = np;
    for (int blk = 0; blk < d.n_polys; blk++) emu::run_block(256, [&] { ntt_inv_body(E.T, d, blk, g_ldsA, g_ldsB); });
  });
}

// gadget digits + to_ntt (gadget.rs:34-60, poly.rs:613-638) through ntt_fwd_body: src = `batch` matrices of rdim x cols raw
// polys; dst = batch * rdim*t * cols NTT polys [poly][crt][N]; bits = 64: plain to_ntt (t = 1)
int emu_digits_to_ntt(void* h, const uint64_t* src, int batch, int rdim, int cols, int t, int bits, uint32_t* dst) {
  return guarded([&] {
    const EmuParams& E = *(EmuParams*)h;
    FwdDesc d{};
    d.src = src;
    d.dst = dst;
    d.n_out = batch * rdim * t * cols;
    d.rdim = rdim;
    d.cols = cols;
    d.t = t;
    d.bits = bits;
    d.src_batch_stride = rdim * cols;
    d.src_row0 = 0;
    d.src_cols = cols;
    for (int o = 0; o < d.n_out; o++)
      for (int c = 0; c < 2; c++) emu::run_block(256, [&] { ntt_fwd_body(E.T, d, o, c, g_ldsA, g_ldsB); });
  });
}

// scalar helpers, n values each
void emu_reduce64(void* h, int c, const uint64_t* x, int n, uint32_t* out) {
  const EmuParams& E = *(EmuParams*)h;
  for (int i = 0; i < n; i++) out[i] = reduce64(x[i], E.T.c.mod[c]);
}
void emu_canon_word(const uint64_t* x, int n, uint64_t* out) {
  for (int i = 0; i < n; i++) out[i] = canon_word(x[i]);
}
void emu_rescale(const uint64_t* a, int n, uint64_t Q, uint64_t out_mod, uint64_t* out) {
  for (int i = 0; i < n; i++) out[i] = rescale_dev(a[i], Q, out_mod);
}
// PACKED database unit (device_common.hpp): 64 lanes x four words (row r, column offset) -> 448 dwords -> back
void emu_pack_unpack(const uint64_t* words /*[64][4]*/, uint32_t* unit /*[448]*/, uint64_t* back /*[64][4]*/) {
  std::memset(unit, 0, 448 * sizeof(uint32_t));
  for (int lane = 0; lane < 64; lane++)
    pack_unit_lane(unit, lane, words[lane * 4 + 0], words[lane * 4 + 1], words[lane * 4 + 2], words[lane * 4 + 3]);
  for (int lane = 0; lane < 64; lane++)
    for (int w = 0; w < 4; w++) back[lane * 4 + w] = unpack_word(unit, lane, w);
}

}  // extern "C"
