//! Rust binding of include/spiral_hip.h: the spiral-rs server surface (lib/spiral-rs/src/server.rs) backed by the
//! MI355X library.  `sys` mirrors the C header one to one (every entry point, same order); the safe layer keeps the
//! reference's names and argument meaning: `process_query(params, public_params, query, db) -> Vec<u8>`
//! (server.rs:650-655), `PublicParameters::deserialize` (client.rs:212-259), the stage functions of
//! server.rs:19-591 on `&[u64]` in the reference's own layouts, and the multi-GPU entry point.
//! Where the reference panics (`assert!` / `unwrap`, e.g. client.rs:213,304) the shim panics with the library's
//! message; `try_*` variants return `Result<_, HipError>` for callers (lib/server) that map errors to HTTP bodies
//! (lib/server/src/error.rs:7-34).
#![allow(non_camel_case_types, clippy::missing_safety_doc, clippy::too_many_arguments)]

use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int, c_long, c_void};

pub mod sys {
    use super::*;

    pub const SP_OK: c_int = 0;
    pub const SP_E_ARG: c_int = -1;
    pub const SP_E_HIP: c_int = -2;
    pub const SP_E_OOM: c_int = -3;
    pub const SP_E_STATE: c_int = -4;
    pub const SP_E_NOTFOUND: c_int = -5;
    pub const SP_MAX_ROW_SHARDS: c_int = 8;
    pub const SP_COMM_ID_BYTES: usize = 128;

    #[repr(C)]
    pub struct sp_params_t {
        _p: [u8; 0],
    }
    #[repr(C)]
    pub struct sp_pp_t {
        _p: [u8; 0],
    }
    #[repr(C)]
    pub struct sp_db_t {
        _p: [u8; 0],
    }
    #[repr(C)]
    pub struct sp_query_t {
        _p: [u8; 0],
    }
    #[repr(C)]
    pub struct sp_comm_t {
        _p: [u8; 0],
    }
    #[repr(C)]
    pub struct sp_server_t {
        _p: [u8; 0],
    }
    /// Host-supplied collectives for `sp_comm_create_custom` (enqueue on `hip_stream`, never block on it).
    #[repr(C)]
    pub struct sp_comm_ops_t {
        pub reduce_scatter_u32: Option<
            unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, recv_count: usize, hip_stream: *mut c_void) -> c_int,
        >,
        pub all_gather_u64: Option<
            unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, send_count: usize, hip_stream: *mut c_void) -> c_int,
        >,
        pub user: *mut c_void,
    }

    extern "C" {
        pub fn sp_last_error() -> *const c_char;
        pub fn sp_paths_taken(reset: c_int) -> u64;
        pub fn sp_path_name(bit: c_int) -> *const c_char;
        pub fn sp_debug_set(name: *const c_char, value: c_long) -> c_int;
        pub fn sp_device_count() -> c_int;
        pub fn sp_set_device(device: c_int) -> c_int;
        // ---- Params (util.rs:219-263, params.rs:49-200)
        pub fn sp_params_from_json(json: *const c_char) -> *mut sp_params_t;
        pub fn sp_params_free(p: *mut sp_params_t);
        pub fn sp_params_get(p: *const sp_params_t, name: *const c_char) -> u64;
        pub fn sp_params_ntt_table(p: *const sp_params_t, crt: c_int, which: c_int, out_n: *mut u64) -> c_int;
        // ---- database (`db: &[u64]` of server.rs:650-655, resident)
        pub fn sp_db_create(p: *const sp_params_t, shard: c_int, num_shards: c_int) -> *mut sp_db_t;
        pub fn sp_db_create_columns(p: *const sp_params_t, shard: c_int, num_shards: c_int) -> *mut sp_db_t;
        pub fn sp_db_create_sparse(p: *const sp_params_t) -> *mut sp_db_t;
        pub fn sp_db_sparse_items(db: *const sp_db_t) -> usize;
        pub fn sp_db_free(db: *mut sp_db_t);
        pub fn sp_db_load_plane(db: *mut sp_db_t, plane: c_int, z0: c_int, nz: c_int, words: *const u64) -> c_int;
        pub fn sp_db_load(db: *mut sp_db_t, words: *const u64, n_words: usize) -> c_int;
        pub fn sp_db_load_items(db: *mut sp_db_t, file: *const u8, file_len: usize) -> c_int;
        pub fn sp_db_update_item(db: *mut sp_db_t, item_idx: usize, data: *const u8, len: usize) -> c_int;
        pub fn sp_db_fill_synthetic(db: *mut sp_db_t, seed: u64) -> c_int;
        pub fn sp_db_prepare_batch(db: *mut sp_db_t, built: *mut c_int) -> c_int;
        pub fn sp_db_batch_copy_bytes(db: *const sp_db_t) -> usize;
        pub fn sp_synth_word(seed: u64, ref_index: u64) -> u64;
        pub fn sp_db_read_ref(db: *const sp_db_t, plane: c_int, z: c_int, ii: c_int, j0: c_int, count: c_int, out: *mut u64) -> c_int;
        pub fn sp_db_device_bytes(db: *const sp_db_t) -> usize;
        // ---- PublicParameters (client.rs:146-259)
        pub fn sp_pp_deserialize(p: *const sp_params_t, data: *const u8, len: usize) -> *mut sp_pp_t;
        pub fn sp_pp_free(pp: *mut sp_pp_t);
        pub fn sp_pp_export(pp: *const sp_pp_t, out: *mut u64, cap_words: usize, n_words: *mut usize) -> c_int;
        // ---- process_query (server.rs:650-741)
        pub fn sp_process_query(p: *const sp_params_t, pp: *const sp_pp_t, query: *const u8, query_len: usize, db: *const sp_db_t,
                                out: *mut u8, out_cap: usize, out_len: *mut usize) -> c_int;
        pub fn sp_process_query_batch(p: *const sp_params_t, pps: *const *const sp_pp_t, queries: *const *const u8,
                                      query_lens: *const usize, batch: c_int, db: *const sp_db_t, out: *mut u8,
                                      out_stride: usize, out_len: *mut usize) -> c_int;
        // ---- the same call split around the exchange step (multi-GPU driven by the caller)
        pub fn sp_query_sweep_scatter(q: *mut sp_query_t, db: *const sp_db_t, g: c_int) -> c_int;
        pub fn sp_query_sweep_scatter_plane(q: *mut sp_query_t, db: *const sp_db_t, g: c_int, plane: c_int) -> c_int;
        pub fn sp_query_fold_local(q: *mut sp_query_t, reduced_chunk_dev: *const c_void, g: c_int) -> c_int;
        pub fn sp_query_fold_local_plane(q: *mut sp_query_t, reduced_plane_chunk_dev: *const c_void, g: c_int, plane: c_int) -> c_int;
        pub fn sp_query_fold_local_join(q: *mut sp_query_t) -> c_int;
        pub fn sp_query_stream2(q: *mut sp_query_t) -> *mut c_void;
        pub fn sp_query_local_cts_ptr(q: *mut sp_query_t) -> *mut c_void;
        pub fn sp_query_local_cts_words(q: *const sp_query_t) -> usize;
        pub fn sp_query_finish_gathered(q: *mut sp_query_t, gathered_dev: *const c_void, g: c_int, out: *mut u8, out_cap: usize,
                                        out_len: *mut usize) -> c_int;
        pub fn sp_query_begin(p: *const sp_params_t, pp: *const sp_pp_t, query: *const u8, query_len: usize) -> *mut sp_query_t;
        pub fn sp_query_begin_for_db(p: *const sp_params_t, pp: *const sp_pp_t, query: *const u8, query_len: usize,
                                     db: *const sp_db_t) -> *mut sp_query_t;
        pub fn sp_query_sweep(q: *mut sp_query_t, db: *const sp_db_t) -> c_int;
        pub fn sp_query_partial_ptr(q: *mut sp_query_t) -> *mut c_void;
        pub fn sp_query_partial_words(q: *const sp_query_t) -> usize;
        pub fn sp_query_sync(q: *mut sp_query_t) -> c_int;
        pub fn sp_query_finish(q: *mut sp_query_t, out: *mut u8, out_cap: usize, out_len: *mut usize) -> c_int;
        pub fn sp_query_free(q: *mut sp_query_t);
        pub fn sp_query_stream(q: *mut sp_query_t) -> *mut c_void;
        pub fn sp_query_timings(q: *const sp_query_t, ms4: *mut f32) -> c_int;
        // ---- multi-GPU with the collectives inside the library (RCCL linked directly)
        pub fn sp_comm_unique_id(id128: *mut u8) -> c_int;
        pub fn sp_comm_create(rank: c_int, world: c_int, id128: *const u8) -> *mut sp_comm_t;
        pub fn sp_comm_create_custom(rank: c_int, world: c_int, ops: *const sp_comm_ops_t) -> *mut sp_comm_t;
        pub fn sp_comm_free(c: *mut sp_comm_t);
        pub fn sp_comm_rank(c: *const sp_comm_t) -> c_int;
        pub fn sp_comm_world(c: *const sp_comm_t) -> c_int;
        pub fn sp_comm_stream(c: *mut sp_comm_t) -> *mut c_void;
        pub fn sp_comm_barrier(c: *mut sp_comm_t) -> c_int;
        pub fn sp_process_query_sharded(c: *mut sp_comm_t, p: *const sp_params_t, pp: *const sp_pp_t, query: *const u8,
                                        query_len: usize, shard: *const sp_db_t, out: *mut u8, out_cap: usize,
                                        out_len: *mut usize) -> c_int;
        pub fn sp_process_queries_sharded(c: *mut sp_comm_t, p: *const sp_params_t, pps: *const *const sp_pp_t, queries: *const *const u8,
                                          query_lens: *const usize, n: c_int, shard: *const sp_db_t, out: *mut u8, out_stride: usize,
                                          out_len: *mut usize) -> c_int;
        pub fn sp_comm_reserve(c: *mut sp_comm_t, p: *const sp_params_t) -> c_int;
        pub fn sp_comm_timings(c: *const sp_comm_t, ms3: *mut f32) -> c_int;
        pub fn sp_comm_describe(c: *const sp_comm_t, buf: *mut c_char, cap: usize) -> c_int;
        // ---- request layer: lib/server/src/bin/server.rs ServerState, /setup, /private-read (no HTTP)
        pub fn sp_server_create(params: *const sp_params_t, db: *const sp_db_t) -> *mut sp_server_t;
        pub fn sp_server_free(s: *mut sp_server_t);
        pub fn sp_server_clients(s: *const sp_server_t) -> usize;
        pub fn sp_server_setup(s: *mut sp_server_t, pp_bytes: *const u8, len: usize, uuid_out37: *mut c_char) -> c_int;
        pub fn sp_server_setup_json(s: *mut sp_server_t, body: *const c_char, body_len: usize, out: *mut c_char, out_cap: usize,
                                    out_len: *mut usize) -> c_int;
        pub fn sp_server_forget(s: *mut sp_server_t, uuid: *const c_char) -> c_int;
        pub fn sp_server_private_read(s: *mut sp_server_t, requests: *const *const u8, request_lens: *const usize, n: c_int,
                                      out: *mut u8, out_stride: usize, out_lens: *mut usize) -> c_int;
        pub fn sp_server_private_read_json(s: *mut sp_server_t, body: *const c_char, body_len: usize, out: *mut c_char,
                                           out_cap: usize, out_len: *mut usize) -> c_int;
        pub fn sp_server_private_read_json_bound(s: *const sp_server_t, n_queries: c_int) -> usize;
        // ---- measurement aids
        pub fn sp_sweep_launches(p: *const sp_params_t, db: *const sp_db_t) -> c_int;
        pub fn sp_bench_sweep(q: *mut sp_query_t, db: *const sp_db_t, iters: c_int, ms_per_launch: *mut f32) -> c_int;
        pub fn sp_bench_sweep_ex(q: *mut sp_query_t, db: *const sp_db_t, iters: c_int, per_plane_launches: c_int,
                                 ms_per_launch: *mut f32) -> c_int;
        pub fn sp_bench_sweep_batch(qs: *const *mut sp_query_t, batch: c_int, db: *const sp_db_t, iters: c_int,
                                    ms_per_pass: *mut f32) -> c_int;
        pub fn sp_debug_cu_probe(bit_lo: c_int, bit_hi: c_int, blocks: c_int, out2: *mut u32) -> c_int;
        pub fn sp_debug_resident_check(params: *const sp_params_t, pp: *const sp_pp_t, out: *mut u64, cap: c_int) -> c_int;
        pub fn sp_debug_chacha20_u64(seed: *const u8, out: *mut u64, count: usize) -> c_int;
        pub fn sp_bench_ntt(p: *const sp_params_t, m: c_int, blocks: c_int, reps: c_int, ns_per_ntt: *mut f32) -> c_int;
        // ---- stage level, 1:1 with the reference's pub functions (host arrays in the reference layouts)
        pub fn sp_ntt_forward(p: *const sp_params_t, data: *mut u64, count: usize) -> c_int;
        pub fn sp_ntt_inverse(p: *const sp_params_t, data: *mut u64, count: usize) -> c_int;
        pub fn sp_to_ntt(p: *const sp_params_t, raw: *const u64, out: *mut u64, count: usize) -> c_int;
        pub fn sp_from_ntt(p: *const sp_params_t, ntt: *const u64, out: *mut u64, count: usize) -> c_int;
        pub fn sp_multiply(p: *const sp_params_t, a: *const u64, ar: usize, ac: usize, b: *const u64, bc: usize, res: *mut u64) -> c_int;
        pub fn sp_add(p: *const sp_params_t, a: *const u64, b: *const u64, count: usize, res: *mut u64) -> c_int;
        pub fn sp_add_into(p: *const sp_params_t, res: *mut u64, a: *const u64, count: usize) -> c_int;
        pub fn sp_scalar_multiply(p: *const sp_params_t, scalar: *const u64, b: *const u64, count: usize, res: *mut u64) -> c_int;
        pub fn sp_automorph(p: *const sp_params_t, a: *const u64, count: usize, t: usize, res: *mut u64) -> c_int;
        pub fn sp_gadget_invert_rdim(p: *const sp_params_t, inp: *const u64, rows_in: usize, cols: usize, out: *mut u64,
                                     rows_out: usize, rdim: usize) -> c_int;
        pub fn sp_reorient_reg_ciphertexts(p: *const sp_params_t, v_reg: *const u64, out: *mut u64) -> c_int;
        pub fn sp_multiply_reg_by_database(p: *const sp_params_t, db: *const u64, v_firstdim: *const u64, dim0: usize,
                                           num_per: usize, out: *mut u64) -> c_int;
        pub fn sp_coefficient_expansion(p: *const sp_params_t, pp: *const sp_pp_t, v: *mut u64, g: usize, stop_round: usize,
                                        max_bits_to_gen_right: usize) -> c_int;
        pub fn sp_regev_to_gsw(p: *const sp_params_t, pp: *const sp_pp_t, v_inp: *const u64, v_gsw: *mut u64, num_gsw: usize) -> c_int;
        pub fn sp_get_v_folding_neg(p: *const sp_params_t, v_folding: *const u64, out: *mut u64) -> c_int;
        pub fn sp_expand_query(p: *const sp_params_t, pp: *const sp_pp_t, query: *const u8, query_len: usize,
                               v_reg_reoriented: *mut u64, v_folding: *mut u64) -> c_int;
        pub fn sp_fold_ciphertexts(p: *const sp_params_t, cts: *mut u64, num_per: usize, v_folding: *const u64,
                                   v_folding_neg: *const u64) -> c_int;
        pub fn sp_fold_ciphertexts_fused(p: *const sp_params_t, cts: *mut u64, num_per: usize, v_folding: *const u64,
                                         fused_min_pairs: c_long) -> c_int;
        pub fn sp_pack(p: *const sp_params_t, pp: *const sp_pp_t, v_ct: *const u64, out: *mut u64) -> c_int;
        pub fn sp_encode(p: *const sp_params_t, v_packed: *const u64, out: *mut u8, out_cap: usize, out_len: *mut usize) -> c_int;
    }
}

/// A non-zero status from the library with its message (`sp_last_error`).
#[derive(Debug, Clone)]
pub struct HipError {
    pub code: c_int,
    pub message: String,
}
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "spiral_hip rc={}: {}", self.code, self.message)
    }
}
impl std::error::Error for HipError {}

fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::sp_last_error()) }.to_string_lossy().into_owned()
}
fn check(rc: c_int) -> Result<(), HipError> {
    if rc == sys::SP_OK {
        Ok(())
    } else {
        Err(HipError { code: rc, message: last_error() })
    }
}
/// The reference panics on malformed input (client.rs:213,304; server.rs:131-132,434-439): keep that contract.
fn must<T>(r: Result<T, HipError>) -> T {
    r.unwrap_or_else(|e| panic!("{}", e))
}

const N: usize = 2048; // poly_len (util.rs:245)
const CRT: usize = 2;

/// `spiral_rs::params::Params` (params.rs:49-82) as built by `util::params_from_json` (util.rs:219-263).
pub struct Params(*mut sys::sp_params_t);
/// `spiral_rs::client::PublicParameters` (client.rs:146-152), NTT'd and device resident.
pub struct PublicParameters(*mut sys::sp_pp_t);
/// The `db: &[u64]` argument of `process_query`, registered once and resident in HBM.
pub struct Database(*mut sys::sp_db_t);
/// One RCCL rank (one process per GPU).
pub struct Comm(*mut sys::sp_comm_t);
// handles are immutable after creation and safe to share between host threads (include/spiral_hip.h)
unsafe impl Send for Params {}
unsafe impl Sync for Params {}
unsafe impl Send for PublicParameters {}
unsafe impl Sync for PublicParameters {}
unsafe impl Send for Database {}
unsafe impl Sync for Database {}
unsafe impl Send for Comm {}

impl Params {
    pub fn try_from_json(cfg: &str) -> Result<Self, HipError> {
        let c = CString::new(cfg).map_err(|_| HipError { code: sys::SP_E_ARG, message: "NUL in JSON".into() })?;
        let p = unsafe { sys::sp_params_from_json(c.as_ptr()) };
        if p.is_null() {
            Err(HipError { code: sys::SP_E_ARG, message: last_error() })
        } else {
            Ok(Params(p))
        }
    }
    /// util.rs:219-222
    pub fn from_json(cfg: &str) -> Self {
        must(Self::try_from_json(cfg))
    }
    pub fn get(&self, name: &str) -> u64 {
        let c = CString::new(name).unwrap();
        let v = unsafe { sys::sp_params_get(self.0, c.as_ptr()) };
        assert!(v != u64::MAX, "unknown Params field {}", name);
        v
    }
    pub fn setup_bytes(&self) -> usize { self.get("setup_bytes") as usize }        // params.rs:146
    pub fn query_bytes(&self) -> usize { self.get("query_bytes") as usize }        // params.rs:169
    pub fn num_items(&self) -> usize { self.get("num_items") as usize }            // params.rs:120
    pub fn response_bytes(&self) -> usize { self.get("response_bytes") as usize }  // server.rs:476-480
    pub fn db_words(&self) -> usize { self.get("db_words") as usize }
    pub fn dim0(&self) -> usize { 1usize << self.get("db_dim_1") }
    pub fn num_per(&self) -> usize { 1usize << self.get("db_dim_2") }
    pub fn planes(&self) -> usize { (self.get("instances") * self.get("n") * self.get("n")) as usize }
    pub fn t_gsw(&self) -> usize { self.get("t_gsw") as usize }
    pub fn n(&self) -> usize { self.get("n") as usize }
    pub fn as_ptr(&self) -> *const sys::sp_params_t { self.0 }
}
impl Drop for Params {
    fn drop(&mut self) {
        unsafe { sys::sp_params_free(self.0) }
    }
}

impl PublicParameters {
    pub fn try_deserialize(params: &Params, data: &[u8]) -> Result<Self, HipError> {
        let h = unsafe { sys::sp_pp_deserialize(params.0, data.as_ptr(), data.len()) };
        if h.is_null() {
            Err(HipError { code: sys::SP_E_ARG, message: last_error() })
        } else {
            Ok(PublicParameters(h))
        }
    }
    /// client.rs:212-259 (`assert_eq!(params.setup_bytes(), data.len())`)
    pub fn deserialize(params: &Params, data: &[u8]) -> Self {
        must(Self::try_deserialize(params, data))
    }
}
impl Drop for PublicParameters {
    fn drop(&mut self) {
        unsafe { sys::sp_pp_free(self.0) }
    }
}

impl Database {
    /// An empty bucket (all-zero polynomials = what lib/server's SparseDb yields for absent rows), unsharded.
    pub fn new(params: &Params) -> Self {
        Self::shard(params, 0, 1)
    }
    /// lib/server's `SparseDb` (db/sparse_db.rs:5-48): only the items written through `update_item` are stored and
    /// multiplied; `process_query` on it follows lib/server/src/server.rs:17-99.
    pub fn sparse(params: &Params) -> Self {
        let h = unsafe { sys::sp_db_create_sparse(params.0) };
        assert!(!h.is_null(), "sp_db_create_sparse: {}", last_error());
        Database(h)
    }
    /// Row shard `rank` of `world` (multi-GPU): first-dimension rows [rank*dim0/world, (rank+1)*dim0/world).
    pub fn shard(params: &Params, rank: usize, world: usize) -> Self {
        let h = unsafe { sys::sp_db_create(params.0, rank as c_int, world as c_int) };
        assert!(!h.is_null(), "sp_db_create: {}", last_error());
        Database(h)
    }
    /// `db` is exactly what `generate_random_db_and_get_item` / `load_db_from_seek` /
    /// `load_preprocessed_db_from_file` return (server.rs:223-386).
    pub fn register(params: &Params, db: &[u64]) -> Self {
        let d = Self::new(params);
        must(check(unsafe { sys::sp_db_load(d.0, db.as_ptr(), db.len()) }));
        d
    }
    /// GPU form of `load_db_from_seek` (server.rs:320-357) over an in-memory image of the item file.
    pub fn load_items(&mut self, file: &[u8]) {
        must(check(unsafe { sys::sp_db_load_items(self.0, file.as_ptr(), file.len()) }))
    }
    /// `/write`: lib/server/src/db/loading.rs:317-359 `update_item_raw`.
    pub fn update_item(&mut self, item_idx: usize, data: &[u8]) {
        must(check(unsafe { sys::sp_db_update_item(self.0, item_idx, data.as_ptr(), data.len()) }))
    }
    pub fn fill_synthetic(&mut self, seed: u64) {
        must(check(unsafe { sys::sp_db_fill_synthetic(self.0, seed) }))
    }
    /// Builds the digit-planar copy that lists of 9..16 queries read, at load time instead of inside the first such
    /// list; `true` when the copy stands (unsharded PACKED database with room for 8 more bytes per word).
    pub fn prepare_batch(&mut self) -> bool {
        let mut built: c_int = 0;
        must(check(unsafe { sys::sp_db_prepare_batch(self.0, &mut built) }));
        built != 0
    }
    pub fn device_bytes(&self) -> usize {
        unsafe { sys::sp_db_device_bytes(self.0) }
    }
    /// Bytes of the digit-planar copy while it stands (0 otherwise).
    pub fn batch_copy_bytes(&self) -> usize {
        unsafe { sys::sp_db_batch_copy_bytes(self.0) }
    }
}
impl Drop for Database {
    fn drop(&mut self) {
        unsafe { sys::sp_db_free(self.0) }
    }
}

/// Drop-in for `spiral_rs::server::process_query(params, public_params, query, db)` (server.rs:650-655);
/// `query` is the serialized query, i.e. what `Query::deserialize` (client.rs:303-329) takes.
pub fn try_process_query(params: &Params, public_params: &PublicParameters, query: &[u8], db: &Database) -> Result<Vec<u8>, HipError> {
    let mut out = vec![0u8; params.response_bytes()];
    let mut n = 0usize;
    check(unsafe {
        sys::sp_process_query(params.0, public_params.0, query.as_ptr(), query.len(), db.0, out.as_mut_ptr(), out.len(), &mut n)
    })?;
    out.truncate(n);
    Ok(out)
}
pub fn process_query(params: &Params, public_params: &PublicParameters, query: &[u8], db: &Database) -> Vec<u8> {
    must(try_process_query(params, public_params, query, db))
}

/// The per-request query loop of lib/server/src/bin/server.rs:152-158 in one call: groups of <= 16 queries share one
/// pass over the database.  `items[i] = (public parameters of the query's client, serialized query)`.
pub fn process_query_batch(params: &Params, items: &[(&PublicParameters, &[u8])], db: &Database) -> Vec<Vec<u8>> {
    let n = params.response_bytes();
    let pps: Vec<*const sys::sp_pp_t> = items.iter().map(|(pp, _)| pp.0 as *const _).collect();
    let qs: Vec<*const u8> = items.iter().map(|(_, q)| q.as_ptr()).collect();
    let lens: Vec<usize> = items.iter().map(|(_, q)| q.len()).collect();
    let mut out = vec![0u8; n * items.len()];
    let mut len = 0usize;
    must(check(unsafe {
        sys::sp_process_query_batch(params.0, pps.as_ptr(), qs.as_ptr(), lens.as_ptr(), items.len() as c_int, db.0,
                                    out.as_mut_ptr(), n, &mut len)
    }));
    out.chunks(n).map(|c| c[..len].to_vec()).collect()
}

impl Comm {
    /// Rank 0 calls this and hands the bytes to every rank (any side channel: file, env, the HTTP control plane).
    pub fn unique_id() -> [u8; sys::SP_COMM_ID_BYTES] {
        let mut id = [0u8; sys::SP_COMM_ID_BYTES];
        must(check(unsafe { sys::sp_comm_unique_id(id.as_mut_ptr()) }));
        id
    }
    /// Collective: returns when all `world` ranks have joined (ncclCommInitRank on the current HIP device).
    pub fn create(rank: usize, world: usize, id: &[u8; sys::SP_COMM_ID_BYTES]) -> Self {
        let h = unsafe { sys::sp_comm_create(rank as c_int, world as c_int, id.as_ptr()) };
        assert!(!h.is_null(), "sp_comm_create: {}", last_error());
        Comm(h)
    }
    pub fn rank(&self) -> usize { unsafe { sys::sp_comm_rank(self.0) as usize } }
    pub fn world(&self) -> usize { unsafe { sys::sp_comm_world(self.0) as usize } }
    pub fn barrier(&mut self) { must(check(unsafe { sys::sp_comm_barrier(self.0) })) }
    /// `process_query` over a row-sharded database: every rank calls it with the same query and its own shard
    /// (`Database::shard(params, rank, world)`); rank 0 gets the response, the others an empty vector.
    pub fn process_query(&mut self, params: &Params, public_params: &PublicParameters, query: &[u8], shard: &Database) -> Vec<u8> {
        let mut out = vec![0u8; params.response_bytes()];
        let mut n = 0usize;
        must(check(unsafe {
            sys::sp_process_query_sharded(self.0, params.0, public_params.0, query.as_ptr(), query.len(), shard.0,
                                          out.as_mut_ptr(), out.len(), &mut n)
        }));
        out.truncate(n);
        out
    }
}
impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { sys::sp_comm_free(self.0) }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage functions with the reference's names (server.rs:19-591, poly.rs, ntt.rs, gadget.rs, util.rs) on flat slices in
// the reference's layouts: PolyMatrixRaw = rows*cols*N u64, PolyMatrixNTT = rows*cols*crt_count*N u64.
// ---------------------------------------------------------------------------------------------------------------
pub fn ntt_forward(params: &Params, operand_overall: &mut [u64]) {
    // ntt.rs:67-113
    must(check(unsafe { sys::sp_ntt_forward(params.0, operand_overall.as_mut_ptr(), operand_overall.len() / (CRT * N)) }))
}
pub fn ntt_inverse(params: &Params, operand_overall: &mut [u64]) {
    // ntt.rs:212-258
    must(check(unsafe { sys::sp_ntt_inverse(params.0, operand_overall.as_mut_ptr(), operand_overall.len() / (CRT * N)) }))
}
pub fn to_ntt(params: &Params, raw: &[u64]) -> Vec<u64> {
    // poly.rs:613-623
    let count = raw.len() / N;
    let mut out = vec![0u64; count * CRT * N];
    must(check(unsafe { sys::sp_to_ntt(params.0, raw.as_ptr(), out.as_mut_ptr(), count) }));
    out
}
pub fn from_ntt(params: &Params, ntt: &[u64]) -> Vec<u64> {
    // poly.rs:646-663
    let count = ntt.len() / (CRT * N);
    let mut out = vec![0u64; count * N];
    must(check(unsafe { sys::sp_from_ntt(params.0, ntt.as_ptr(), out.as_mut_ptr(), count) }));
    out
}
pub fn multiply(params: &Params, a: &[u64], a_rows: usize, a_cols: usize, b: &[u64], b_cols: usize) -> Vec<u64> {
    // poly.rs:437-458
    let mut res = vec![0u64; a_rows * b_cols * CRT * N];
    must(check(unsafe { sys::sp_multiply(params.0, a.as_ptr(), a_rows, a_cols, b.as_ptr(), b_cols, res.as_mut_ptr()) }));
    res
}
pub fn automorph(params: &Params, a: &[u64], t: usize) -> Vec<u64> {
    // poly.rs:393-405, 539-551
    let mut res = vec![0u64; a.len()];
    must(check(unsafe { sys::sp_automorph(params.0, a.as_ptr(), a.len() / N, t, res.as_mut_ptr()) }));
    res
}
pub fn gadget_invert_rdim(params: &Params, inp: &[u64], rows_in: usize, cols: usize, rows_out: usize, rdim: usize) -> Vec<u64> {
    // gadget.rs:34-60
    let mut out = vec![0u64; rows_out * cols * N];
    must(check(unsafe { sys::sp_gadget_invert_rdim(params.0, inp.as_ptr(), rows_in, cols, out.as_mut_ptr(), rows_out, rdim) }));
    out
}
pub fn reorient_reg_ciphertexts(params: &Params, v_reg: &[u64]) -> Vec<u64> {
    // util.rs:323-355
    let mut out = vec![0u64; params.dim0() * 2 * N];
    must(check(unsafe { sys::sp_reorient_reg_ciphertexts(params.0, v_reg.as_ptr(), out.as_mut_ptr()) }));
    out
}
pub fn multiply_reg_by_database(params: &Params, db: &[u64], v_firstdim: &[u64], dim0: usize, num_per: usize) -> Vec<u64> {
    // server.rs:155-221: one (instance, trial) plane of the database
    let mut out = vec![0u64; num_per * 2 * CRT * N];
    must(check(unsafe { sys::sp_multiply_reg_by_database(params.0, db.as_ptr(), v_firstdim.as_ptr(), dim0, num_per, out.as_mut_ptr()) }));
    out
}
pub fn coefficient_expansion(params: &Params, public_params: &PublicParameters, v: &mut [u64], g: usize, stop_round: usize,
                             max_bits_to_gen_right: usize) {
    // server.rs:19-121 (v_w_left / v_w_right come from public_params, v_neg1 is internal)
    must(check(unsafe { sys::sp_coefficient_expansion(params.0, public_params.0, v.as_mut_ptr(), g, stop_round, max_bits_to_gen_right) }))
}
pub fn regev_to_gsw(params: &Params, public_params: &PublicParameters, v_inp: &[u64], num_gsw: usize) -> Vec<u64> {
    // server.rs:123-151 with idx_factor 1, idx_offset 0
    let mut out = vec![0u64; num_gsw * 2 * 2 * params.t_gsw() * CRT * N];
    must(check(unsafe { sys::sp_regev_to_gsw(params.0, public_params.0, v_inp.as_ptr(), out.as_mut_ptr(), num_gsw) }));
    out
}
pub fn get_v_folding_neg(params: &Params, v_folding: &[u64]) -> Vec<u64> {
    // server.rs:505-523
    let mut out = vec![0u64; v_folding.len()];
    must(check(unsafe { sys::sp_get_v_folding_neg(params.0, v_folding.as_ptr(), out.as_mut_ptr()) }));
    out
}
/// server.rs:525-591 -> (v_reg_reoriented, v_folding)
pub fn expand_query(params: &Params, public_params: &PublicParameters, query: &[u8]) -> (Vec<u64>, Vec<u64>) {
    let nu2 = params.get("db_dim_2") as usize;
    let mut v_reg = vec![0u64; params.dim0() * 2 * N];
    let mut v_fold = vec![0u64; nu2.max(1) * 2 * 2 * params.t_gsw() * CRT * N];
    must(check(unsafe {
        sys::sp_expand_query(params.0, public_params.0, query.as_ptr(), query.len(), v_reg.as_mut_ptr(), v_fold.as_mut_ptr())
    }));
    v_fold.truncate(nu2 * 2 * 2 * params.t_gsw() * CRT * N);
    (v_reg, v_fold)
}
pub fn fold_ciphertexts(params: &Params, v_cts: &mut [u64], v_folding: &[u64], v_folding_neg: &[u64]) {
    // server.rs:388-427; the folded ciphertext ends up in v_cts[0 .. 2N)
    must(check(unsafe {
        sys::sp_fold_ciphertexts(params.0, v_cts.as_mut_ptr(), v_cts.len() / (2 * N), v_folding.as_ptr(), v_folding_neg.as_ptr())
    }))
}
pub fn pack(params: &Params, public_params: &PublicParameters, v_ct: &[u64]) -> Vec<u64> {
    // server.rs:429-468 with v_w = public_params.v_packing
    let n = params.n();
    let mut out = vec![0u64; (n + 1) * n * CRT * N];
    must(check(unsafe { sys::sp_pack(params.0, public_params.0, v_ct.as_ptr(), out.as_mut_ptr()) }));
    out
}
pub fn encode(params: &Params, v_packed_ct: &[u64]) -> Vec<u8> {
    // server.rs:470-503
    let mut out = vec![0u8; params.response_bytes()];
    let mut n = 0usize;
    must(check(unsafe { sys::sp_encode(params.0, v_packed_ct.as_ptr(), out.as_mut_ptr(), out.len(), &mut n) }));
    out.truncate(n);
    out
}

/// `ServerState` of lib/server/src/bin/server.rs:21-28 minus the HTTP transport: `/setup` and `/private-read` bodies in,
/// response bodies out; the per-uuid public parameters stay device resident.
pub struct Server<'a> {
    h: *mut sys::sp_server_t,
    _params: &'a Params,
    _db: &'a Database,
}
unsafe impl<'a> Send for Server<'a> {}
unsafe impl<'a> Sync for Server<'a> {}
impl<'a> Server<'a> {
    pub fn new(params: &'a Params, db: &'a Database) -> Self {
        let h = unsafe { sys::sp_server_create(params.0, db.0) };
        assert!(!h.is_null(), "sp_server_create: {}", last_error());
        Server { h, _params: params, _db: db }
    }
    /// POST /setup (bin/server.rs:71-94): body in, `{"uuid":"..."}` out.
    pub fn setup(&self, body: &str) -> Result<String, HipError> {
        let mut out = vec![0u8; 64];
        let mut n = 0usize;
        check(unsafe { sys::sp_server_setup_json(self.h, body.as_ptr() as *const c_char, body.len(), out.as_mut_ptr() as *mut c_char, out.len(), &mut n) })?;
        out.truncate(n);
        Ok(String::from_utf8(out).unwrap())
    }
    /// POST /private-read (bin/server.rs:143-164): JSON list of base64 requests in, JSON list of base64 responses out.
    /// `Err` with `code == SP_E_NOTFOUND` maps to `Error::NotFound`.
    pub fn private_read(&self, body: &[u8]) -> Result<String, HipError> {
        let n_max = body.iter().filter(|&&c| c == b',').count() as c_int + 1;
        let cap = unsafe { sys::sp_server_private_read_json_bound(self.h, n_max) };
        let mut out = vec![0u8; cap];
        let mut n = 0usize;
        check(unsafe { sys::sp_server_private_read_json(self.h, body.as_ptr() as *const c_char, body.len(), out.as_mut_ptr() as *mut c_char, out.len(), &mut n) })?;
        out.truncate(n);
        Ok(String::from_utf8(out).unwrap())
    }
}
impl<'a> Drop for Server<'a> {
    fn drop(&mut self) {
        unsafe { sys::sp_server_free(self.h) }
    }
}

/// Names of the kernels / flows the calling thread went through since the last call (diagnostics).
pub fn paths_taken() -> Vec<String> {
    let mask = unsafe { sys::sp_paths_taken(1) };
    let mut out = Vec::new();
    let mut bit = 0;
    loop {
        let p = unsafe { sys::sp_path_name(bit) };
        if p.is_null() {
            break;
        }
        if mask >> bit & 1 == 1 {
            out.push(unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned());
        }
        bit += 1;
    }
    out
}
