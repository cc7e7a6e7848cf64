// Per-call device workspace + the stage functions of the answer path (server.cpp).
#pragma once
#include <vector>

#include "server.hpp"

namespace spiral {

struct Workspace {
  const Params* P;
  DeviceState* D;
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;            // fold of plane p runs here while plane p+1 is swept on `stream`
  std::vector<hipEvent_t> ev_plane;         // sweep of plane p done
  hipEvent_t ev_fold = nullptr;             // all folds done (stream2)
  hipEvent_t ev_round0 = nullptr, ev_right = nullptr;  // expansion round 0 done (main) / odd subtree + GSW side done (stream2)
  bool right_pending = false;               // this query's fold operands are produced on stream2: join_right before use
  bool long_sweep_follows = false;          // hint for run_begin (set by the caller that knows the database)
  hipEvent_t ev_sw[2] = {nullptr, nullptr};  // first sweep launch begins / last sweep launch done (timing)
  bool have_sweep_span = false;
  bool pipelined = false;                   // set by run_sweep_pipelined, consumed by run_finish
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // begin, after expand, after sweep, after fold, end
  // expansion
  DevBuf<u64> q_raw;      // query ct, raw 2x1
  DevBuf<u32> v;          // [2^g] 2x1 NTT cts
  DevBuf<u64> exp_raw;    // [active] automorphed raw cts
  DevBuf<u32> exp_dig;    // digit NTTs of one group
  DevBuf<u32> exp_ct1;    // NTT of row 1 of the automorphed cts
  DevBuf<u64> exp_raw_r;  // the same three for the odd subtree (runs concurrently on stream2)
  DevBuf<u32> exp_dig_r, exp_ct1_r;
  DevBuf<u64> qv;         // reoriented first-dimension query [N][dim0][2]
  DevBuf<u32> fold_mats;  // [nu_2][2 rows][ G - C | C ] (2 x 4 t_gsw NTT polys per GSW ct)
  DevBuf<u32> fold_mats_w;  // the same polynomials in wave layout (k_fold_wave), filled by run_fold_operands (query path: the C halves only)
  bool mats_w_ready = false;
  DevBuf<u64> gsw_raw;
  DevBuf<u32> gsw_dig;
  // sweep
  DevBuf<u32> sweep_out;  // [plane][r][crt][z][ii]
  DevBuf<u64> fold_tail;     // pipe_tail_defer: [2][plane][cts parked per plane][2][N]
  DevBuf<u32> batch_rq;   // query digit table of a batched pass on the matrix cores (first workspace of a group; on first use)
  // fold / pack
  DevBuf<u64> foldX, foldY, final_cts, pack_raw;
  DevBuf<u32> fold_dig, fold_ntt, pack_dig, pack_ct2, pack_res, pack_v1_P, pack_v1_P2;
  DevBuf<u64> pack_v1_raw, du_raw, du_wire;
  DevBuf<u32> du_ntt;
  // host staging (pinned)
  u64* h_query = nullptr;
  u64* h_group_query = nullptr;   // a group leader's pinned staging of all its queries' ciphertexts (run_begin_group), GROUP_MAX x 2 polys
  DevBuf<u64> group_q_raw;        // ... and its device image
  u64* h_packed = nullptr;
  size_t h_packed_words = 0;
  DevBuf<u64> enc_out;       // response bits built on the device
  uint8_t* h_response = nullptr;
  bool host_pinned = true;  // the three staging buffers above come from hipHostMalloc (false: malloc, debug switch ws_pinned)
  bool delta_tail = true;  // unfused fold levels use the delta form too (false: literal two-matrix form)
  // true only while run_fold works on ciphertexts the library produced itself (from_ntt / fold outputs: below Q; set by
  // run_fold_canonical, server.cpp) or on a caller's ciphertexts the stage-level entry point has checked: the fused kernels may
  // then skip the gadget digits that are identically zero (FoldDesc::t_live).  Default false: raw buffers from a caller or a peer
  // rank are decomposed in full.
  bool fold_inputs_below_q = false;
  int out_G = 1;  // column interleave of the sweep output (multi-GPU reduce-scatter path)
  bool zero_shortcuts = false;  // lib/server fold semantics (sparse buckets): set per query, every level fused
  long fused_min_pairs = 256;  // fold levels with at least this many (pair, plane) units use k_fold_fused

  Workspace(const Params& P, DeviceState& D);
  ~Workspace();
  Workspace(const Workspace&) = delete;
  Workspace& operator=(const Workspace&) = delete;
  void ensure_expand();
  void ensure_sweep();
  void ensure_finish();
  size_t plane_group() const;
};

// tree: 0 = whole schedule, 1 = even subtree, 2 = round 0 + odd subtree; rounds [r_begin, g_rounds)
void run_coefficient_expansion(Workspace& W, const sp_pp& pp, size_t g_rounds, const DeviceState::PrunedPlan* plan = nullptr,
                               int tree = 0, size_t r_begin = 0);
void join_right(Workspace& W);
void run_regev_to_gsw(Workspace& W, const sp_pp& pp, const u32* v_src, const int* src_ct, const int* src_poly);
void run_folding_neg(Workspace& W);
void run_mats_to_wave(Workspace& W, size_t levels, bool c_only = false);
void run_fold_operands(Workspace& W);
void run_begin_direct(Workspace& W, const uint8_t* query);
// j0 / nj > 0: only the first-dimension rows [j0, j0 + nj) of the expanded query will be used (row shards)
// B queries of one group (same params, un-pruned): the expansions' rounds as shared launches (kernels.hpp, GroupOff)
void run_begin_group(Workspace* const* Ws, const sp_pp* const* pps, const uint8_t* const* queries, const size_t* query_lens, int B);
void run_begin(Workspace& W, const sp_pp& pp, const uint8_t* query, size_t query_len, int j0 = 0, int nj = 0,
               const DeviceState::PrunedPlan* plan = nullptr);
void run_sweep_sparse(Workspace& W, const sp_db& db, const int* col_ptr, const int* col_rows, const int* col_slots);
// expansion schedule pruned to an arbitrary set of first-dimension rows (rows[j] != 0), lists uploaded
std::unique_ptr<DeviceState::PrunedPlan> build_pruned_plan_rows(const Params& P, const std::vector<char>& rows);
void run_sweep(Workspace& W, const sp_db& db);
void run_sweep_pipelined(Workspace& W, const sp_db& db);
bool sweep_is_pipelined(const Params& p, const sp_db& db);
void launch_plane_sweep(Workspace& W, const sp_db& db, size_t plane);
bool fused_fold_supported(const Params& p);
u64* run_fold(Workspace& W, u64* X, u64* Y, int np, int num_cts, int top, int d_begin = 0, int d_end = -1);
void run_fold_local(Workspace& W, const u32* reduced_chunk, int G);
void run_fold_local_plane(Workspace& W, const u32* reduced_plane_chunk, int G, int plane);
void run_fold_local_join(Workspace& W);
void run_finish_gathered(Workspace& W, const sp_pp& pp, const u64* gathered, int G);
void run_fold_all(Workspace& W, bool premod);
void run_pack(Workspace& W, const sp_pp& pp);
void run_finish(Workspace& W, const sp_pp& pp, bool premod);
size_t encode_response(const Params& p, const u64* packed, uint8_t* out);
void run_encode_device(Workspace& W);  // pack_raw -> enc_out -> h_response (async on W.stream)

}  // namespace spiral
