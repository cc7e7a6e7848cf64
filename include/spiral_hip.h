/*
 * spiral_hip.h -- C ABI of libspiral_hip.so: the MI355X (gfx950) implementation of the Spiral PIR
 * server answer path of blyssprivacy/sdk `lib/spiral-rs` (feature `server`).
 *
 * This is the drop-in boundary.  Every entry point names the reference interface it replaces
 * (file:line under /root/reference/lib/spiral-rs/src/).  Plain pointers and sizes only; all
 * multi-byte data is little-endian / native-endian exactly as the reference's `to_ne_bytes`
 * wire formats on x86-64 (client.rs:58,77,292; util.rs:294-319).
 *
 * Conventions
 *   - Opaque handles, freed by the matching *_free.  The library never frees caller memory.
 *   - Functions returning `int` return SP_OK (0) or a negative SP_E_* code; the message is in
 *     sp_last_error() (thread-local).  Nothing panics/aborts across the ABI; a Rust shim turns a
 *     non-zero status into the `panic!`/`assert!` the reference would have raised
 *     (client.rs:213,304; server.rs:131-132,434-439).
 *   - "ref layout": PolyMatrixRaw = rows*cols*N u64 (poly.rs:31-35,92-94); PolyMatrixNTT =
 *     rows*cols*crt_count*N u64, each < moduli[crt] (poly.rs:263-265).  N = poly_len = 2048,
 *     crt_count = 2, moduli {268369921, 249561089} (util.rs:245-248).
 *   - Every compute entry point runs on the HIP device; there is no CPU fallback.  If no gfx950
 *     device is usable the call fails with SP_E_HIP.
 *   - Handles are immutable after creation and may be shared by host threads; each call takes a
 *     private workspace + HIP stream from an internal pool.
 */
#ifndef SPIRAL_HIP_H
#define SPIRAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_E_ARG (-1)    /* bad argument / length mismatch (reference: assert_eq! on sizes) */
#define SP_E_HIP (-2)    /* HIP runtime error or no device */
#define SP_E_OOM (-3)    /* device or host allocation failed */
#define SP_E_STATE (-4)  /* call sequence error */
#define SP_E_NOTFOUND (-5) /* unknown client uuid (lib/server Error::NotFound -> HTTP 404, lib/server/src/error.rs:7-34) */

/* Row shards per database: each shard emits partial residues < q < 2^28 that the exchange step sums element-wise in
 * 32 bits (ncclSum on ncclUint32); 8 * (q - 1) < 2^31, so 8 is the largest shard count that can never overflow.
 * sp_db_create and the sp_query_sweep_scatter / fold_local family reject larger counts with SP_E_ARG. */
#define SP_MAX_ROW_SHARDS 8

typedef struct sp_params sp_params_t; /* spiral_rs::params::Params           params.rs:49-82   */
typedef struct sp_pp sp_pp_t;         /* spiral_rs::client::PublicParameters client.rs:146-152 */
typedef struct sp_db sp_db_t;         /* the `db: &[u64]` argument of process_query, device-resident */
typedef struct sp_query sp_query_t;   /* one in-flight query (expanded), for the multi-GPU split */

const char* sp_last_error(void);
/* Diagnostics: bit mask of the kernels / flows the CALLING THREAD's library calls went through since the last reset
 * (bit b is named by sp_path_name(b): sweep_packed_persist, from_sweep4, fold_fused, fold_tail_delta,
 * pipelined_fold_overlap, expand_pruned, ...).  Tests use it to assert that the bytes they compared were produced by
 * the production code path (e.g. the persistent PACKED sweep + fused fold + fold/sweep overlap at full size) and not
 * by a silently selected fallback kernel.  reset != 0 clears the mask after reading it. */
uint64_t sp_paths_taken(int reset);
const char* sp_path_name(int bit); /* NULL past the last defined bit */
/* Run-time tunables for A/B measurements inside one process (the same switches as the SPIRAL_<NAME> environment
 * variables of DESIGN.md section 8, e.g. "pipe_ring", "pipe_tail_defer", "sweep_prio", "batch_group"): takes effect on the next launch / next workspace. */
int sp_debug_set(const char* name, long value);
/* Number of visible HIP devices (0 if none); selects `device` for this thread's subsequent calls. */
int sp_device_count(void);
int sp_set_device(int device);

/* ------------------------------------------------------------------ Params
 * util.rs:219-263 params_from_json(): keys n, nu_1, nu_2, p, q2_bits, t_gsw, t_conv, t_exp_left,
 * t_exp_right, instances, db_item_size, version; presence of "direct_upload" disables query
 * expansion.  Single or double quotes are accepted (the reference's presets use single quotes and
 * `.replace("'", "\"")`, util.rs:104-120). */
sp_params_t* sp_params_from_json(const char* json);
/* The params handle owns the device tables and the pool of workspaces every other handle made from it borrows (spiral-rs:
 * `&'a Params` in PublicParameters, Query, the database slice): free those first -- sp_query_t, sp_pp_t, sp_db_t, sp_server_t,
 * a reserved sp_comm_t -- then the params. */
void sp_params_free(sp_params_t*);
/* Field / derived-size accessor: poly_len, crt_count, modulus, moduli0, moduli1, n, pt_modulus,
 * q2_bits, t_conv, t_exp_left, t_exp_right, t_gsw, expand_queries, db_dim_1, db_dim_2, instances,
 * db_item_size, version, g (params.rs:129), stop_round (:134), setup_bytes (:146), query_bytes
 * (:169), num_items (:120), db_words (instances*n*n*num_items*N), response_bytes
 * (server.rs:476-480).  Returns UINT64_MAX for an unknown name. */
uint64_t sp_params_get(const sp_params_t*, const char* name);
/* ntt_tables[crt][which][0..N) (params.rs:85-96, ntt.rs:39-65); which: 0 fwd, 1 fwd', 2 inv, 3 inv' */
int sp_params_ntt_table(const sp_params_t*, int crt, int which, uint64_t* out_n);

/* --------------------------------------------------------------------- DB
 * The reference passes the whole preprocessed database as `db: &[u64]` on every call
 * (server.rs:650-655) in the layout [instance][trial][z][ii][j], word = lo28 | hi28 << 32
 * (server.rs:262-270).  A per-call host pointer cannot be streamed at HBM rate, so residency is the
 * one deliberate API change: register once, query many times.
 * Row sharding (multi-GPU): shard `s` of `S` holds first-dimension rows j in [s*dim0/S, (s+1)*dim0/S). */
sp_db_t* sp_db_create(const sp_params_t*, int shard, int num_shards);
/* Column sharding (the zero-reduction alternative): shard s of S (a power of two) holds the
 * output columns ii = s (mod S) of every row.  Each shard's sweep yields complete first-dimension outputs
 * for its columns, so the flow is sp_query_sweep -> sp_query_fold_local(q, sp_query_partial_ptr(q), S) ->
 * gather of the local results -> sp_query_finish_gathered; no partial sums cross GPUs. */
sp_db_t* sp_db_create_columns(const sp_params_t*, int shard, int num_shards);
/* A sparse bucket: lib/server's SparseDb (lib/server/src/db/sparse_db.rs:5-48).  Starts empty; items arrive through
 * sp_db_update_item (= update_item_raw + upsert, db/loading.rs:317-359) and only they are stored (one packed NTT
 * polynomial per (item, plane)) and multiplied, so the first-dimension step costs time in proportion to the occupancy.
 * sp_process_query / sp_query_begin_for_db + sweep + finish on such a handle follow lib/server's process_query
 * (lib/server/src/server.rs:17-99): expansion pruned to the rows that hold items (compute/query_expansion.rs:213-357),
 * multiply_reg_by_sparse_database (compute/dot_product.rs:13-220; exact sums mod q, not the reference's wrapping u64),
 * and fold_ciphertexts with the all-zero shortcuts of compute/fold.rs:38-44 -- so the response bytes are lib/server's,
 * which differ from spiral-rs's dense process_query over the zero-filled database wherever a shortcut fires (both
 * decode to the item).  Updates must not run concurrently with queries on the same handle (the reference holds a
 * RwLock write guard, bin/server.rs:33,48).  Needs expand_queries and 3 <= t_gsw <= 32. */
sp_db_t* sp_db_create_sparse(const sp_params_t*);
size_t sp_db_sparse_items(const sp_db_t*); /* items present */
void sp_db_free(sp_db_t*);
/* Upload (a z-range of) one (instance,trial) plane given in the reference layout [z][ii][j] with
 * the FULL dim0 rows per (z,ii); the shard keeps only its rows.  `words` points at row z0.
 * Equivalent of handing generate_random_db_and_get_item / load_db_from_seek /
 * load_preprocessed_db_from_file output (server.rs:223-275, 320-386) to process_query.
 * The handle keeps its device-side upload window (at most 64 MiB, not counted by sp_db_device_bytes) from the first call on:
 * nothing is allocated or freed between the kernels of a load (allocation rule). */
int sp_db_load_plane(sp_db_t*, int plane, int z0, int nz, const uint64_t* words);
/* Whole database in one call: `words` = instances*n*n*N*num_per*dim0 u64 in the reference layout. */
int sp_db_load(sp_db_t*, const uint64_t* words, size_t n_words);
/* Build the database on the device from the raw item file: the GPU form of load_db_from_seek
 * (server.rs:320-357) over an in-memory file image.  `file` holds num_items records of db_item_size
 * bytes (a shorter file reads as zeros past its end, as the reference's failed seek/short read do);
 * each record is split into instances*n*n chunks of bytes_per_chunk bytes, log2(p)-bit coefficients,
 * recenter_mod, forward NTT, packed words -- written straight into the resident layout. */
int sp_db_load_items(sp_db_t*, const uint8_t* file, size_t file_len);
/* Upsert one item in place (the /write path: lib/server/src/db/loading.rs:317-359 update_item_raw over a
 * densified bucket; sp_db_create starts from the empty bucket = all-zero polynomials, which is what the
 * reference's SparseDb yields for absent rows).  `data` is zero-padded to db_item_size.  On a row shard
 * that does not hold the item's row the call is a no-op. */
int sp_db_update_item(sp_db_t*, size_t item_idx, const uint8_t* data, size_t len);
/* Synthetic benchmark database generated on the device: reference-layout word index i holds
 * sp_synth_word(seed, i).  (Roofline runs at sizes no host buffer can hold.) */
int sp_db_fill_synthetic(sp_db_t*, uint64_t seed);
/* Build now what batched calls would otherwise build inside the first sp_process_query_batch of 9 .. 16 queries: the
 * digit-planar copy of an unsharded PACKED database (8 more bytes per word) that the two-tile matrix-core pass reads --
 * when the shape has one and the device has the room (call it after the load, e.g. where bin/server.rs:98-141 has loaded
 * the database).  *built (may be null): 1 = the copy stands, 0 = this database keeps to the PACKED kernels.  The copy
 * follows sp_db_update_item in place (8 sixteen-byte entries per (plane, z)); the bulk loaders drop it. */
int sp_db_prepare_batch(sp_db_t*, int* built);
uint64_t sp_synth_word(uint64_t seed, uint64_t ref_index);
/* Read back words of the reference-layout view (debug / tests): plane, z, ii, j0..j0+count within the shard's rows */
int sp_db_read_ref(const sp_db_t*, int plane, int z, int ii, int j0, int count, uint64_t* out);
size_t sp_db_device_bytes(const sp_db_t*);      /* the resident database words */
size_t sp_db_batch_copy_bytes(const sp_db_t*);  /* the digit-planar copy beside them while it stands (sp_db_prepare_batch), else 0 */

/* ------------------------------------------------------- PublicParameters
 * client.rs:212-259 PublicParameters::deserialize(params, data): 32-byte seed, row 0 of every
 * matrix regenerated as Q - (ChaCha20Rng(seed).gen::<u64>() % Q) (client.rs:47-49, 68-75), other
 * rows read as LE u64, everything NTT'd and kept device-resident.  len must equal setup_bytes. */
sp_pp_t* sp_pp_deserialize(const sp_params_t*, const uint8_t* data, size_t len);
void sp_pp_free(sp_pp_t*);
/* NTT-form matrices in wire order, ref layout (tests): v_packing[n], v_expansion_left[g],
 * v_expansion_right[stop_round+1] (if present on the wire), v_conversion[1]. */
int sp_pp_export(const sp_pp_t*, uint64_t* out, size_t cap_words, size_t* n_words);

/* ---------------------------------------------------------- process_query
 * server.rs:650-741 process_query(params, public_params, query, db) -> Vec<u8>, with
 * Query::deserialize (client.rs:303-329) folded in.  `query` is the 32-byte seed followed by the
 * serialized query body (query_bytes long).  Requires a db created with num_shards == 1.
 * out must hold response_bytes. */
int sp_process_query(const sp_params_t*, const sp_pp_t*, const uint8_t* query, size_t query_len,
                     const sp_db_t*, uint8_t* out, size_t out_cap, size_t* out_len);
/* The per-request loop of lib/server (bin/server.rs:152-158: process_query for every query of the list) as one call: the
 * responses are those of `batch` sp_process_query calls, out + i * out_stride each.  On an unsharded PACKED database the
 * queries share database passes in groups -- up to 8 per pass, up to 16 where the two-tile matrix-core pass applies (which
 * reads the digit-planar copy of the database when one stands: sp_db_prepare_batch) -- with the next group's expansions
 * queued behind the current group's folds; on 8-byte (narrow) databases it is one pass per query with up to three queries
 * in flight on their own streams.  Out of memory is handled inside the call (the planar copy is given back, then groups of
 * 8 one at a time, then one query at a time) before it is reported. */
int sp_process_query_batch(const sp_params_t*, const sp_pp_t* const* pps, const uint8_t* const* queries,
                           const size_t* query_lens, int batch, const sp_db_t*, uint8_t* out,
                           size_t out_stride, size_t* out_len);

/* Multi-GPU split of process_query around the one exchange step (sum of per-shard partial
 * first-dimension outputs):
 *   begin  : Query::deserialize + expand_query + get_v_folding_neg     (server.rs:664-680; G - C itself is not materialised:
 *            every fold step of the library is the delta form, which reads C only -- DESIGN.md section 3)
 *   sweep  : multiply_reg_by_database over this shard's rows             (server.rs:698-705)
 *            -> partial residues (u32, < q) at sp_query_partial_ptr(), layout
 *               [plane][r][crt][z][ii], sp_query_partial_words() u32 words; DEVICE memory
 *   ...    : caller sums the partial buffers of all shards element-wise (RCCL ncclSum on u32;
 *            8 shards * (q-1) < 2^31) and, on the rank that finishes, writes the sum back.
 *   finish : % q, from_ntt, fold_ciphertexts, pack, encode                (server.rs:707-740)
 * All three enqueue on the query's HIP stream; sp_query_sync() waits for it. */
/* Distributed-fold variant (what bench.py uses for N > 1), G = number of row shards = ranks, a power of two:
 *   sweep_scatter : as sweep, but the partial buffer is column-interleaved: chunk g (contiguous,
 *                   partial_words/G u32) holds columns ii = g mod G as [plane][r][crt][z][ii / G]
 *   ...           : caller reduce-scatters the partial buffers (RCCL ncclSum, u32): rank g receives chunk g
 *   fold_local    : % q, from_ntt and the top nu_2 - log2(G) fold levels on this rank's columns
 *                   -> planes 2x1 raw cts at sp_query_local_cts_ptr() (DEVICE, local_cts_words u64)
 *   ...           : caller gathers the G local results on the finishing rank: [g][plane][2][N]
 *   finish_gathered: the last log2(G) fold levels (leaf g = rank g), pack, encode. */
int sp_query_sweep_scatter(sp_query_t*, const sp_db_t*, int G);
/* The same sweep one (instance, trial) plane per launch, planes in order 0 .. planes-1.  Plane p's region of the
 * partial buffer ([p * partial_words/planes, (p+1) * partial_words/planes)) is laid out [g][r][crt][z][ii / G], so
 * each plane can be reduce-scattered on its own, overlapping the exchange of plane p with the sweep of plane p+1;
 * the concatenation of the received chunks is the [plane][r][crt][z][ii / G] buffer sp_query_fold_local takes. */
int sp_query_sweep_scatter_plane(sp_query_t*, const sp_db_t*, int G, int plane);
int sp_query_fold_local(sp_query_t*, const void* reduced_chunk_dev, int G);
/* The local fold one plane at a time on the query's SECOND stream (sp_query_stream2), planes in order: plane p may
 * be folded as soon as it has been swept and its reduced chunk ([r][crt][z][ii / G]) is complete — the caller orders
 * stream2 after that exchange — so the fold of plane p runs beside the sweep / exchange of the later planes.
 * sp_query_fold_local_join() orders the main stream after all of them (replaces sp_query_fold_local). */
int sp_query_fold_local_plane(sp_query_t*, const void* reduced_plane_chunk_dev, int G, int plane);
int sp_query_fold_local_join(sp_query_t*);
void* sp_query_stream2(sp_query_t*);
void* sp_query_local_cts_ptr(sp_query_t*);
size_t sp_query_local_cts_words(const sp_query_t*);
int sp_query_finish_gathered(sp_query_t*, const void* gathered_dev, int G, uint8_t* out, size_t out_cap, size_t* out_len);
sp_query_t* sp_query_begin(const sp_params_t*, const sp_pp_t*, const uint8_t* query, size_t query_len);
/* The same for a query that will sweep `db`: on a row shard only the first-dimension ciphertexts of the shard's
 * rows are expanded (the even subtree of coefficient_expansion is pruned to their ancestors; server.rs:58-111 is
 * otherwise unchanged, the GSW side is complete).  The sweep calls then insist on a database with those rows.
 * db == NULL, unsharded or column-sharded: identical to sp_query_begin. */
sp_query_t* sp_query_begin_for_db(const sp_params_t*, const sp_pp_t*, const uint8_t* query, size_t query_len,
                                  const sp_db_t* db);
int sp_query_sweep(sp_query_t*, const sp_db_t*);
void* sp_query_partial_ptr(sp_query_t*);
size_t sp_query_partial_words(const sp_query_t*);
int sp_query_sync(sp_query_t*);
int sp_query_finish(sp_query_t*, uint8_t* out, size_t out_cap, size_t* out_len);
void sp_query_free(sp_query_t*);
/* HIP stream (hipStream_t) the query's work is enqueued on -- for event timing by the caller. */
void* sp_query_stream(sp_query_t*);
/* Stage timings of the last finish()ed query in milliseconds (HIP events on the query stream):
 * [0] expand+conversion+folding_neg, [1] db sweep, [2] from_ntt+fold, [3] pack+encode(+D2H). */
int sp_query_timings(const sp_query_t*, float* ms4);

/* ----------------------------------------------------------------- multi-GPU, collectives inside the library
 * One process per GPU, the database row-sharded (sp_db_create(params, rank, world)).  The reference has no
 * multi-node / multi-device code (it is CPU-only); the caller this serves is lib/server's request handler
 * (lib/server/src/bin/server.rs:98-141), which would call sp_process_query_sharded where it calls process_query
 * (server.rs:650-655) today -- on every rank, with the same query bytes; rank 0 gets the response.
 * Built-in transport: RCCL (librccl linked directly): ncclReduceScatter(ncclUint32, ncclSum) of each plane's partial
 * Regev ciphertexts on the communicator's own stream while the next plane is swept, then ncclAllGather(ncclUint64)
 * of one locally folded ciphertext per plane per rank.  world must be a power of two <= SP_MAX_ROW_SHARDS.
 *   rank 0   : sp_comm_unique_id(id)  -> hand the SP_COMM_ID_BYTES bytes to every rank (any side channel)
 *   all ranks: sp_comm_create(rank, world, id)   (collective: returns when every rank has joined)
 * One sharded query at a time per communicator (calls are serialised inside; every rank must issue them in the
 * same order).  *out_len = 0 on ranks other than 0. */
typedef struct sp_comm sp_comm_t;
#define SP_COMM_ID_BYTES 128
int sp_comm_unique_id(uint8_t* id128);
sp_comm_t* sp_comm_create(int rank, int world, const uint8_t* id128);
/* Custom transport: the host supplies the two collectives (its own RCCL/MPI setup, or an in-process loopback for
 * tests).  Both are ENQUEUE operations on `hip_stream` (hipStream_t): they must order their device work after what
 * is already on that stream and must not block the host on it.  Return 0 on success.
 *   reduce_scatter_u32: send = world chunks of recv_count u32 each; recv (recv_count u32) = element-wise sum over
 *                       ranks of chunk[rank]                      (ncclReduceScatter semantics)
 *   all_gather_u64    : recv = world blocks of send_count u64, block g = rank g's send   (ncclAllGather semantics) */
typedef struct sp_comm_ops {
  int (*reduce_scatter_u32)(void* user, const void* send, void* recv, size_t recv_count, void* hip_stream);
  int (*all_gather_u64)(void* user, const void* send, void* recv, size_t send_count, void* hip_stream);
  void* user;
} sp_comm_ops_t;
sp_comm_t* sp_comm_create_custom(int rank, int world, const sp_comm_ops_t* ops);
void sp_comm_free(sp_comm_t*);
int sp_comm_rank(const sp_comm_t*);
int sp_comm_world(const sp_comm_t*);
void* sp_comm_stream(sp_comm_t*); /* hipStream_t the collectives are enqueued on */
int sp_comm_barrier(sp_comm_t*);  /* one tiny all-gather + host wait: every rank has reached this call */
/* process_query (server.rs:650-741) over a row-sharded database; `shard` = this rank's sp_db_create(params, rank,
 * world).  Byte-identical to sp_process_query over the unsharded database. */
int sp_process_query_sharded(sp_comm_t*, const sp_params_t*, const sp_pp_t*, const uint8_t* query, size_t query_len,
                             const sp_db_t* shard, uint8_t* out, size_t out_cap, size_t* out_len);
/* The same for a LIST of queries (what a /private-read request carries, lib/server/src/bin/server.rs:152-158),
 * software-pipelined: query k + 1 is deserialised and expanded while query k's planes are swept and exchanged, so the
 * replicated expansion leaves the per-query critical path.  Every rank passes the same list; response k is written at
 * out + k * out_stride on rank 0 (out_stride >= response_bytes), *out_len = response_bytes there and 0 elsewhere.
 * Byte-identical to n calls of sp_process_query_sharded. */
int sp_process_queries_sharded(sp_comm_t*, const sp_params_t*, const sp_pp_t* const* pps, const uint8_t* const* queries,
                               const size_t* query_lens, int n, const sp_db_t* shard, uint8_t* out, size_t out_stride,
                               size_t* out_len);
/* Allocates the communicator's exchange buffers and events for `params` now, so that no sharded query allocates
 * anything (otherwise the first query with new params does).  Not a collective. */
int sp_comm_reserve(sp_comm_t*, const sp_params_t* params);
/* milliseconds of the last sharded query on this rank (HIP events): [0] the per-plane sweep launches (the exchanges of
 * the earlier planes run beside them), [1] exchange tail + local fold + all-gather, [2] the part of the exchange that is NOT
 * hidden behind the sweeps: last sweep launch done -> last plane's reduce-scatter done (exchange stream) */
int sp_comm_timings(const sp_comm_t*, float* ms3);
/* One line of JSON into buf: transport ("rccl" / "custom"), RCCL version, rank / world / device, bytes one rank sends and
 * receives per plane reduce-scatter and in the all-gather, the three sp_comm_timings values.  For the logs of a first run on a
 * multi-GPU node (the reference has no multi-device code; bench.py --gpus N prints it per rank). */
int sp_comm_describe(const sp_comm_t*, char* buf, size_t cap);

/* ------------------------------------------------------ request layer (lib/server's binary without the HTTP transport)
 * ServerState of lib/server/src/bin/server.rs:21-28: the params, the resident database and the
 * RwLock<HashMap<uuid, PublicParameters>> that POST /setup fills and POST /private-read consults.  `params` and `db` are
 * borrowed and must outlive the handle.  All entry points may be called concurrently from several host threads. */
typedef struct sp_server sp_server_t;
sp_server_t* sp_server_create(const sp_params_t* params, const sp_db_t* db);
void sp_server_free(sp_server_t*);
size_t sp_server_clients(const sp_server_t*); /* registered public-parameter sets */
/* POST /setup (bin/server.rs:71-94) after base64 decoding: deserialize (length must be setup_bytes), NTT, keep resident
 * under a fresh UUIDv4; uuid_out37 receives the 36 characters + NUL. */
int sp_server_setup(sp_server_t*, const uint8_t* pp_bytes, size_t len, char* uuid_out37);
/* The same on the HTTP body: a JSON string holding base64(public parameters); out = {"uuid":"..."} (NUL-terminated). */
int sp_server_setup_json(sp_server_t*, const char* body, size_t body_len, char* out, size_t out_cap, size_t* out_len);
/* Drop a client's public parameters (not in the reference, which never evicts). SP_E_NOTFOUND for an unknown uuid. */
int sp_server_forget(sp_server_t*, const char* uuid);
/* POST /private-read (bin/server.rs:98-164) on decoded requests: request i is uuid (36 bytes) || serialized query when
 * the params expand queries (bin/server.rs:107-121), serialized public parameters || query otherwise (:123-138).  The
 * reference answers the list one query at a time (:152-158); here the whole list goes through the batch scheduler
 * (sp_process_query_batch: groups of <= 16 queries (two tiles of 8 on the matrix cores; <= 8 elsewhere) share one pass over the database, each with its own client's public
 * parameters).  Response i (response_bytes long) is written at out + i * out_stride.  A malformed length is SP_E_ARG
 * (the reference asserts), an unknown uuid SP_E_NOTFOUND; nothing is answered in either case. */
int sp_server_private_read(sp_server_t*, const uint8_t* const* requests, const size_t* request_lens, int n, uint8_t* out,
                           size_t out_stride, size_t* out_lens);
/* The same on the HTTP body: JSON list of base64 strings in, JSON list of base64 strings out (NUL-terminated; the text
 * serde_json / the base64 crate produce: no spaces, standard alphabet, padding). */
int sp_server_private_read_json(sp_server_t*, const char* body, size_t body_len, char* out, size_t out_cap, size_t* out_len);
size_t sp_server_private_read_json_bound(const sp_server_t*, int n_queries); /* out_cap that always suffices */

/* Stand-alone timed sweep for the roofline measurement: issues the db-sweep launches of one query over `db`
 * (the same kernel and launch shapes sp_process_query uses) `iters` times with the query slice of `q`, HIP events
 * on the launch stream around the whole batch; returns average milliseconds per kernel launch in *ms_per_launch.
 * sp_sweep_launches() = kernel launches per query sweep (a wide single-GPU database is swept one launch per plane, so
 * that the from_ntt and the fold of what has been swept overlap the rest of the sweep; 1 otherwise). */
int sp_sweep_launches(const sp_params_t*, const sp_db_t*);
int sp_bench_sweep(sp_query_t* q, const sp_db_t* db, int iters, float* ms_per_launch);
/* per_plane_launches: 1 = one launch per plane (what sp_query_sweep_scatter_plane issues), 0 = one launch,
 * -1 = what sp_query_sweep would do for this db (sp_bench_sweep). */
int sp_bench_sweep_ex(sp_query_t* q, const sp_db_t* db, int iters, int per_plane_launches, float* ms_per_launch);
/* The same for the BATCHED pass of sp_process_query_batch (BASELINE configs[4]): the `batch` (<= 16 where the two-tile pass applies, else <= 8) begun queries `qs`
 * share database passes exactly as a group of sp_process_query_batch does (query digit table + k_sweep_mfma_batch on
 * the matrix cores from 4 queries, k_sweep_packed_batch below; one launch over all planes); returns average
 * milliseconds per PASS.  The queries' partial buffers hold the pass's outputs afterwards. */
int sp_bench_sweep_batch(sp_query_t* const* qs, int batch, const sp_db_t* db, int iters, float* ms_per_pass);

/* Diagnostics: where workgroups land.  `blocks` 64-thread workgroups are launched on a stream whose CU mask has bits
 * [bit_lo, bit_hi) set (no mask when bit_hi <= bit_lo); out2[2b] = HW_REG_XCC_ID, out2[2b+1] = HW_REG_HW_ID of
 * workgroup b.  Used to verify the CU partition of the fold / sweep overlap (SPIRAL_CU_SPLIT). */
int sp_debug_cu_probe(int bit_lo, int bit_hi, int blocks, uint32_t* out2);
/* Diagnostic: checksums of the resident tables / index lists / public parameters as seen by a kernel, by a
 * device-to-host copy, by a kernel after an L2 write-back + invalidate, and of the host original (24 values). */
int sp_debug_resident_check(const sp_params_t* params, const sp_pp_t* pp, uint64_t* out, int cap);
/* The library's ChaCha20 keystream as u64 words (rand_chacha 0.3.1 ChaCha20Rng::from_seed(seed).gen::<u64>(), client.rs:47-49:
 * what regenerates row 0 of the public parameters and of a query; no GPU needed): tests compare it with the RFC 8439-pinned
 * restatement word for word, on lengths that exercise the eight-block AVX2 body and the one-block tail. */
int sp_debug_chacha20_u64(const uint8_t seed[32], uint64_t* out, size_t count);

/* Profiling aid: nanoseconds per 2048-point forward NTT of the transform core alone (M = 1, 2 or 4 coefficient
 * vectors per thread, `blocks` workgroups each chaining `reps` transforms, no memory traffic but twiddles). */
int sp_bench_ntt(const sp_params_t*, int M, int blocks, int reps, float* ns_per_ntt);

/* ------------------------------------------------------------ stage level
 * 1:1 with the reference's pub functions, operating on caller-owned host arrays in ref layout.
 * These exist for parity tests and for callers (lib/server) that drive stages themselves. */

/* ntt.rs:67-113 / 212-258: in-place forward / inverse negacyclic NTT of `count` polys (crt*N each) */
int sp_ntt_forward(const sp_params_t*, uint64_t* data, size_t count);
int sp_ntt_inverse(const sp_params_t*, uint64_t* data, size_t count);
/* poly.rs:613-638 to_ntt / to_ntt_no_reduce: raw[count][N] -> ntt[count][crt*N] */
int sp_to_ntt(const sp_params_t*, const uint64_t* raw, uint64_t* out, size_t count);
/* poly.rs:646-663 from_ntt: ntt[count][crt*N] -> raw[count][N] (iNTT + CRT compose, params.rs:207-214) */
int sp_from_ntt(const sp_params_t*, const uint64_t* ntt, uint64_t* out, size_t count);
/* poly.rs:437-458 multiply: res[ar x bc] = a[ar x ac] * b[ac x bc] (NTT form) */
int sp_multiply(const sp_params_t*, const uint64_t* a, size_t ar, size_t ac, const uint64_t* b, size_t bc,
                uint64_t* res);
/* poly.rs:483-498 add, :500-512 add_into, :575-588 scalar_multiply over `count` NTT polynomials (2 * 2048 words each;
 * `scalar` is one polynomial): res = a + b; res += a; res = scalar * b, all pointwise mod the two primes.  The reference's
 * add does not re-reduce a sum below 2 q of canonical inputs differently from these: inputs are reduced mod q first. */
int sp_add(const sp_params_t*, const uint64_t* a, const uint64_t* b, size_t count, uint64_t* res);
int sp_add_into(const sp_params_t*, uint64_t* res, const uint64_t* a, size_t count);
int sp_scalar_multiply(const sp_params_t*, const uint64_t* scalar, const uint64_t* b, size_t count, uint64_t* res);
/* poly.rs:539-551 automorph on `count` raw polys (x -> x^t, sign flip Q - a, Q for a == 0) */
int sp_automorph(const sp_params_t*, const uint64_t* a, size_t count, size_t t, uint64_t* res);
/* gadget.rs:34-60 gadget_invert_rdim followed by to_ntt_no_reduce is what the pipeline uses; this
 * export returns the raw digits: inp[rows_in x cols] -> out[rows_out x cols] */
int sp_gadget_invert_rdim(const sp_params_t*, const uint64_t* inp, size_t rows_in, size_t cols, uint64_t* out,
                          size_t rows_out, size_t rdim);
/* util.rs:323-355 reorient_reg_ciphertexts: v_reg[dim0] 2x1 NTT -> out[N*dim0*2] packed lo|hi<<32 */
int sp_reorient_reg_ciphertexts(const sp_params_t*, const uint64_t* v_reg, uint64_t* out);
/* server.rs:155-221 multiply_reg_by_database: db = one plane [N][num_per][dim0] (ref layout),
 * v_firstdim = [N][dim0][2]; out = num_per 2x1 NTT cts (ref layout, out[i].data[r*2N + crt*N + z]) */
int sp_multiply_reg_by_database(const sp_params_t*, const uint64_t* db, const uint64_t* v_firstdim, size_t dim0,
                                size_t num_per, uint64_t* out);
/* server.rs:19-121 coefficient_expansion on v[2^g] 2x1 NTT cts in place (v_w_* from pp; v_neg1 internal) */
int sp_coefficient_expansion(const sp_params_t*, const sp_pp_t*, uint64_t* v, size_t g, size_t stop_round,
                             size_t max_bits_to_gen_right);
/* server.rs:123-151 regev_to_gsw (idx_factor 1, idx_offset 0): v_inp[t_gsw*num_gsw] 2x1 NTT ->
 * v_gsw[num_gsw] 2 x 2t_gsw NTT, V = pp.v_conversion[0] */
int sp_regev_to_gsw(const sp_params_t*, const sp_pp_t*, const uint64_t* v_inp, uint64_t* v_gsw, size_t num_gsw);
/* server.rs:505-523 get_v_folding_neg: v_folding[nu_2] 2 x 2t_gsw NTT -> same shape */
int sp_get_v_folding_neg(const sp_params_t*, const uint64_t* v_folding, uint64_t* out);
/* server.rs:525-591 expand_query: v_reg_reoriented[N*dim0*2], v_folding[nu_2] 2 x 2t_gsw NTT */
int sp_expand_query(const sp_params_t*, const sp_pp_t*, const uint8_t* query, size_t query_len,
                    uint64_t* v_reg_reoriented, uint64_t* v_folding);
/* server.rs:388-427 fold_ciphertexts: cts[num_per] raw 2x1 in/out (result in cts[0]; the other
 * entries are scratch in the reference too and are returned unmodified here) */
int sp_fold_ciphertexts(const sp_params_t*, uint64_t* cts, size_t num_per, const uint64_t* v_folding,
                        const uint64_t* v_folding_neg);
/* The same fold the way process_query runs it -- every step in the delta form  ct_i + C (*) (G^-1(ct_{i+half}) - G^-1(ct_i)),
 * which equals (G - C) (*) ct_i + C (*) ct_{i+half} exactly (this export still forms G - C = get_v_folding_neg(v_folding) on the
 * device, server.rs:505-523; nothing reads it) -- and levels with at least `fused_min_pairs` pairs go through the fused
 * one-workgroup-per-step kernel (0: the library's default threshold, 1: every level fused); the remaining levels use
 * the three-launch tree tail.  Result in cts[0 .. 2N); byte-identical to
 * fold_ciphertexts(cts, v_folding, get_v_folding_neg(v_folding)). */
int sp_fold_ciphertexts_fused(const sp_params_t*, uint64_t* cts, size_t num_per, const uint64_t* v_folding,
                              long fused_min_pairs);
/* server.rs:429-468 pack: v_ct[n*n] raw 2x1, v_w = pp.v_packing -> (n+1) x n NTT */
int sp_pack(const sp_params_t*, const sp_pp_t*, const uint64_t* v_ct, uint64_t* out);
/* server.rs:470-503 encode: v_packed[instances] raw (n+1) x n -> response bytes */
int sp_encode(const sp_params_t*, const uint64_t* v_packed, uint8_t* out, size_t out_cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* SPIRAL_HIP_H */
