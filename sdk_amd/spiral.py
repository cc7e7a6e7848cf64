"""Host-side mirror of the spiral-rs server API over the C ABI of libspiral_hip.so.

Names, argument meaning and error behaviour follow lib/spiral-rs/src (file:line cited per item); data
crosses the boundary in the reference's own layouts (numpy uint64 arrays / bytes).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class SpiralError(RuntimeError):
    """Raised where the reference would panic (assert!/unwrap) or on a HIP failure."""


def library_path():
    # SPIRAL_HIP_LIB: load another build of the same C ABI (A/B measurements of kernel or layout changes)
    return os.environ.get("SPIRAL_HIP_LIB") or os.path.join(_HERE, "libspiral_hip.so")


def build_library(force=False):
    """Compile libspiral_hip.so for gfx950 with hipcc, in-tree (sdk_amd/libspiral_hip.so)."""
    src = os.path.join(_HERE, "csrc")
    so = library_path()
    deps = [os.path.join(src, f) for f in os.listdir(src) if f.endswith((".cpp", ".hip", ".hpp"))]
    deps.append(os.path.join(_HERE, "..", "include", "spiral_hip.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["make", "-C", src, "-s", "-j8"])
    return so


def lib():
    """The loaded C-ABI library.  Fails loudly if it has not been built: there is no fallback path."""
    global _LIB
    if _LIB is None:
        so = library_path()
        if not os.path.exists(so):
            raise SpiralError(f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); sdk_amd has no CPU fallback")
        # torch-rocm bundles its own libamdhip64; two HIP runtimes in one process do not share devices
        # ("No HIP GPUs are available").  Importing torch first makes the loader reuse its copy for us.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(so)
        L.sp_last_error.restype = C.c_char_p
        L.sp_params_from_json.restype = C.c_void_p
        L.sp_params_from_json.argtypes = [C.c_char_p]
        L.sp_params_free.argtypes = [C.c_void_p]
        L.sp_params_get.restype = C.c_uint64
        L.sp_params_get.argtypes = [C.c_void_p, C.c_char_p]
        L.sp_db_create.restype = C.c_void_p
        L.sp_db_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.sp_db_free.argtypes = [C.c_void_p]
        L.sp_db_device_bytes.restype = C.c_size_t
        L.sp_db_device_bytes.argtypes = [C.c_void_p]
        L.sp_db_batch_copy_bytes.restype = C.c_size_t
        L.sp_db_batch_copy_bytes.argtypes = [C.c_void_p]
        L.sp_synth_word.restype = C.c_uint64
        L.sp_synth_word.argtypes = [C.c_uint64, C.c_uint64]
        L.sp_pp_deserialize.restype = C.c_void_p
        L.sp_pp_deserialize.argtypes = [C.c_void_p, u8p, C.c_size_t]
        L.sp_pp_free.argtypes = [C.c_void_p]
        L.sp_query_begin.restype = C.c_void_p
        L.sp_query_begin.argtypes = [C.c_void_p, C.c_void_p, u8p, C.c_size_t]
        L.sp_query_free.argtypes = [C.c_void_p]
        L.sp_query_partial_ptr.restype = C.c_void_p
        L.sp_query_partial_ptr.argtypes = [C.c_void_p]
        L.sp_query_partial_words.restype = C.c_size_t
        L.sp_query_partial_words.argtypes = [C.c_void_p]
        L.sp_query_stream.restype = C.c_void_p
        L.sp_query_stream.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _finalizing(_is=__import__("sys").is_finalizing):
    """Interpreter shutdown destroys objects in no particular order (a Params before the QueryRun that returns a workspace
    to it): handles are left to the process exit then instead of being freed in the wrong order.  (Bound as a default
    argument: module globals are already gone when the last destructors run.)"""
    return _is()


def _err():
    return lib().sp_last_error().decode(errors="replace")


class NotFound(SpiralError):
    """SP_E_NOTFOUND: unknown client uuid (lib/server Error::NotFound)"""


def _chk(rc):
    if rc == -5:
        raise NotFound(f"libspiral_hip rc={rc}: {_err()}")
    if rc != 0:
        raise SpiralError(f"libspiral_hip rc={rc}: {_err()}")


def _p(a, typ=u64p):
    return a.ctypes.data_as(typ)


def _vp(h):
    return C.c_void_p(h)


def _u64arr(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def _bytes(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


# ------------------------------------------------------------------------------------------------
class Params:
    """spiral_rs::params::Params (params.rs:49-82), constructed by params_from_json (util.rs:219-263)."""

    def __init__(self, cfg):
        if isinstance(cfg, dict):
            cfg = json.dumps(cfg)
        self.json = cfg
        self.h = lib().sp_params_from_json(cfg.encode())
        if not self.h:
            raise SpiralError(_err())

    def __del__(self):
        try:
            if getattr(self, "h", None) and not _finalizing():
                lib().sp_params_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    def get(self, name):
        v = lib().sp_params_get(_vp(self.h), name.encode())
        if v == 0xFFFFFFFFFFFFFFFF:
            raise KeyError(name)
        return int(v)

    def __getattr__(self, name):
        if name.startswith("_") or name in ("h", "json"):
            raise AttributeError(name)
        try:
            return self.get(name)
        except KeyError:
            raise AttributeError(name)

    # derived helpers with the reference's method names (params.rs:116-200)
    def num_expanded(self):
        return 1 << self.db_dim_1

    def num_items(self):
        return self.get("num_items")

    def setup_bytes(self):
        return self.get("setup_bytes")

    def query_bytes(self):
        return self.get("query_bytes")

    @property
    def dim0(self):
        return 1 << self.db_dim_1

    @property
    def num_per(self):
        return 1 << self.db_dim_2

    @property
    def ntt_words(self):
        return self.crt_count * self.poly_len

    def ntt_table(self, crt, which):
        out = np.zeros(self.poly_len, dtype=np.uint64)
        _chk(lib().sp_params_ntt_table(_vp(self.h), C.c_int(crt), C.c_int(which), _p(out)))
        return out


def params_from_json(cfg):
    """util.rs:219-222"""
    return Params(cfg)


# ------------------------------------------------------------------------------------------------
class PolyMatrixRaw:
    """poly.rs:59-64: rows x cols polynomials, N u64 coefficients each (row-major)."""

    def __init__(self, params, rows, cols, data=None):
        self.params, self.rows, self.cols = params, rows, cols
        n = rows * cols * params.poly_len
        self.data = np.zeros(n, dtype=np.uint64) if data is None else _u64arr(data).reshape(n)

    @classmethod
    def zero(cls, params, rows, cols):
        return cls(params, rows, cols)

    def get_poly(self, r, c):
        n = self.params.poly_len
        o = (r * self.cols + c) * n
        return self.data[o:o + n]

    def ntt(self):
        """poly.rs:188-190"""
        return PolyMatrixNTT(self.params, self.rows, self.cols, to_ntt(self.params, self.data))


class PolyMatrixNTT:
    """poly.rs:66-71: rows x cols polynomials in NTT form, crt_count*N u64 words each."""

    def __init__(self, params, rows, cols, data=None):
        self.params, self.rows, self.cols = params, rows, cols
        n = rows * cols * params.ntt_words
        self.data = np.zeros(n, dtype=np.uint64) if data is None else _u64arr(data).reshape(n)

    @classmethod
    def zero(cls, params, rows, cols):
        return cls(params, rows, cols)

    def get_poly(self, r, c):
        n = self.params.ntt_words
        o = (r * self.cols + c) * n
        return self.data[o:o + n]

    def raw(self):
        """poly.rs:335-337"""
        return PolyMatrixRaw(self.params, self.rows, self.cols, from_ntt(self.params, self.data))

    def __mul__(self, rhs):
        """poly.rs:681-689"""
        assert self.cols == rhs.rows
        return PolyMatrixNTT(self.params, self.rows, rhs.cols,
                             multiply(self.params, self.data, self.rows, self.cols, rhs.data, rhs.cols))


def ntt_forward(params, operand_overall):
    """ntt.rs:67-113, in place semantics returned as a new array"""
    a = _u64arr(operand_overall).copy()
    _chk(lib().sp_ntt_forward(_vp(params.h), _p(a), C.c_size_t(a.size // params.ntt_words)))
    return a


def ntt_inverse(params, operand_overall):
    """ntt.rs:212-258"""
    a = _u64arr(operand_overall).copy()
    _chk(lib().sp_ntt_inverse(_vp(params.h), _p(a), C.c_size_t(a.size // params.ntt_words)))
    return a


def to_ntt(params, raw):
    """poly.rs:613-623 (to_ntt_no_reduce, :625-638, gives the same result for its in-range inputs)"""
    raw = _u64arr(raw)
    cnt = raw.size // params.poly_len
    out = np.zeros(cnt * params.ntt_words, dtype=np.uint64)
    _chk(lib().sp_to_ntt(_vp(params.h), _p(raw), _p(out), C.c_size_t(cnt)))
    return out


def from_ntt(params, ntt):
    """poly.rs:646-663"""
    ntt = _u64arr(ntt)
    cnt = ntt.size // params.ntt_words
    out = np.zeros(cnt * params.poly_len, dtype=np.uint64)
    _chk(lib().sp_from_ntt(_vp(params.h), _p(ntt), _p(out), C.c_size_t(cnt)))
    return out


def multiply(params, a, ar, ac, b, bc):
    """poly.rs:437-458"""
    a, b = _u64arr(a), _u64arr(b)
    res = np.zeros(ar * bc * params.ntt_words, dtype=np.uint64)
    _chk(lib().sp_multiply(_vp(params.h), _p(a), C.c_size_t(ar), C.c_size_t(ac), _p(b), C.c_size_t(bc), _p(res)))
    return res


def add(params, a, b):
    """poly.rs:483-498 over flat NTT polynomials"""
    a, b = _u64arr(a), _u64arr(b)
    res = np.zeros(a.size, dtype=np.uint64)
    _chk(lib().sp_add(_vp(params.h), _p(a), _p(b), C.c_size_t(a.size // params.ntt_words), _p(res)))
    return res


def add_into(params, res, a):
    """poly.rs:500-512 (returns the updated copy)"""
    res, a = _u64arr(res).copy(), _u64arr(a)
    _chk(lib().sp_add_into(_vp(params.h), _p(res), _p(a), C.c_size_t(a.size // params.ntt_words)))
    return res


def scalar_multiply(params, scalar, b):
    """poly.rs:575-588: one NTT polynomial times every polynomial of b"""
    scalar, b = _u64arr(scalar), _u64arr(b)
    res = np.zeros(b.size, dtype=np.uint64)
    _chk(lib().sp_scalar_multiply(_vp(params.h), _p(scalar), _p(b), C.c_size_t(b.size // params.ntt_words), _p(res)))
    return res


def automorph(params, a, t):
    """poly.rs:539-551"""
    a = _u64arr(a)
    res = np.zeros_like(a)
    _chk(lib().sp_automorph(_vp(params.h), _p(a), C.c_size_t(a.size // params.poly_len), C.c_size_t(t), _p(res)))
    return res


def gadget_invert_rdim(params, inp, rows_in, cols, rows_out, rdim):
    """gadget.rs:34-60"""
    inp = _u64arr(inp)
    out = np.zeros(rows_out * cols * params.poly_len, dtype=np.uint64)
    _chk(lib().sp_gadget_invert_rdim(_vp(params.h), _p(inp), C.c_size_t(rows_in), C.c_size_t(cols), _p(out),
                                     C.c_size_t(rows_out), C.c_size_t(rdim)))
    return out


def reorient_reg_ciphertexts(params, v_reg):
    """util.rs:323-355"""
    v_reg = _u64arr(v_reg)
    out = np.zeros(params.dim0 * 2 * params.poly_len, dtype=np.uint64)
    _chk(lib().sp_reorient_reg_ciphertexts(_vp(params.h), _p(v_reg), _p(out)))
    return out


# ------------------------------------------------------------------------------------------------
class PublicParameters:
    """client.rs:146-152; only the server-side constructor `deserialize` (client.rs:212-259) exists here."""

    def __init__(self, params, h):
        self.params, self.h = params, h

    @classmethod
    def deserialize(cls, params, data):
        d = _bytes(data)
        h = lib().sp_pp_deserialize(_vp(params.h), _p(d, u8p), C.c_size_t(d.size))
        if not h:
            raise SpiralError(_err())  # reference: assert_eq!(params.setup_bytes(), data.len())
        return cls(params, h)

    def __del__(self):
        try:
            if getattr(self, "h", None) and not _finalizing():
                lib().sp_pp_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    def export(self):
        p = self.params
        cap = (p.setup_bytes() // 8 + 4096) * 4 * 2
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t(0)
        _chk(lib().sp_pp_export(_vp(self.h), _p(out), C.c_size_t(cap), C.byref(n)))
        return out[:n.value].copy()


class Query:
    """client.rs:262-267; holds the serialized form (seed || body) that `Query::deserialize` consumes."""

    def __init__(self, params, data):
        self.params, self.data = params, bytes(data)

    @classmethod
    def deserialize(cls, params, data):
        if len(data) != params.query_bytes():
            raise SpiralError(f"query length {len(data)} != query_bytes {params.query_bytes()}")  # client.rs:304
        return cls(params, data)


class Database:
    """The `db: &[u64]` argument of process_query (server.rs:650-655), resident in HBM.

    shard/num_shards row-shard the first dimension across GPUs (one Database per process/GPU)."""

    def __init__(self, params, shard=0, num_shards=1, by_columns=False):
        self.params = params
        self.shard, self.num_shards, self.by_columns = shard, num_shards, by_columns
        lib().sp_db_create_columns.restype = C.c_void_p
        create = lib().sp_db_create_columns if by_columns else lib().sp_db_create
        self.h = create(_vp(params.h), C.c_int(shard), C.c_int(num_shards))
        if not self.h:
            raise SpiralError(_err())

    def __del__(self):
        try:
            if getattr(self, "h", None) and not _finalizing():
                lib().sp_db_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    @classmethod
    def sparse(cls, params):
        """lib/server's SparseDb (db/sparse_db.rs:5-48): an empty bucket that stores (and multiplies) only the items
        written through update_item; process_query on it follows lib/server/src/server.rs:17-99."""
        self = cls.__new__(cls)
        self.params, self.shard, self.num_shards, self.by_columns = params, 0, 1, False
        lib().sp_db_create_sparse.restype = C.c_void_p
        lib().sp_db_sparse_items.restype = C.c_size_t
        self.h = lib().sp_db_create_sparse(_vp(params.h))
        if not self.h:
            raise SpiralError(_err())
        return self

    def sparse_items(self):
        return int(lib().sp_db_sparse_items(_vp(self.h)))

    def load(self, words):
        """words: the reference-layout array produced by generate_random_db_and_get_item /
        load_db_from_seek / load_preprocessed_db_from_file (server.rs:223-386)."""
        w = _u64arr(words)
        _chk(lib().sp_db_load(_vp(self.h), _p(w), C.c_size_t(w.size)))
        return self

    def load_plane(self, plane, z0, nz, words):
        w = _u64arr(words)
        _chk(lib().sp_db_load_plane(_vp(self.h), C.c_int(plane), C.c_int(z0), C.c_int(nz), _p(w)))
        return self

    def load_items(self, blob):
        """GPU form of load_db_from_seek (server.rs:320-357): `blob` = num_items records of db_item_size bytes."""
        b = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else np.ascontiguousarray(blob, dtype=np.uint8)
        _chk(lib().sp_db_load_items(_vp(self.h), _p(b, u8p), C.c_size_t(b.size)))
        return self

    def update_item(self, item_idx, data):
        """upsert one item in place (lib/server update_item_raw, loading.rs:317-359)"""
        b = np.frombuffer(bytes(data), dtype=np.uint8)
        _chk(lib().sp_db_update_item(_vp(self.h), C.c_size_t(item_idx), _p(b, u8p), C.c_size_t(b.size)))
        return self

    def fill_synthetic(self, seed):
        _chk(lib().sp_db_fill_synthetic(_vp(self.h), C.c_uint64(seed)))
        return self

    def prepare_batch(self):
        """build what lists of 9..16 queries read (the digit-planar copy of an unsharded PACKED database) now, at load time;
        True when the copy stands, False when this database keeps to the PACKED kernels (shape, or no room)"""
        built = C.c_int(0)
        _chk(lib().sp_db_prepare_batch(_vp(self.h), C.byref(built)))
        return bool(built.value)

    def read_ref(self, plane, z, ii, j0, count):
        out = np.zeros(count, dtype=np.uint64)
        _chk(lib().sp_db_read_ref(_vp(self.h), C.c_int(plane), C.c_int(z), C.c_int(ii), C.c_int(j0), C.c_int(count),
                                  _p(out)))
        return out

    def device_bytes(self):
        return int(lib().sp_db_device_bytes(_vp(self.h)))

    def batch_copy_bytes(self):
        """bytes of the digit-planar copy beside the resident words while it stands (prepare_batch), else 0"""
        return int(lib().sp_db_batch_copy_bytes(_vp(self.h)))


def synth_word(seed, ref_index):
    return int(lib().sp_synth_word(C.c_uint64(seed), C.c_uint64(ref_index)))


def synth_words(seed, ref_indices):
    """Vectorised sp_synth_word (numpy), for sampled parity on synthetic databases."""
    M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.asarray(ref_indices, dtype=np.uint64)
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))) & M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        z = z ^ (z >> np.uint64(31))
        lo = (z & np.uint64(0xFFFFFFFF)) % np.uint64(268369921)
        hi = (z >> np.uint64(32)) % np.uint64(249561089)
    return lo | (hi << np.uint64(32))


# ------------------------------------------------------------------------------------------------
class QueryRun:
    """One in-flight query split around the exchange step (see include/spiral_hip.h)."""

    def __init__(self, params, pp, query, db=None):
        """db: the database this query will sweep; on a row shard the expansion is pruned to the shard's rows"""
        q = query.data if isinstance(query, Query) else bytes(query)
        d = _bytes(q)
        self.params, self.pp = params, pp
        L = lib()
        L.sp_query_begin_for_db.restype = C.c_void_p
        self.h = L.sp_query_begin_for_db(_vp(params.h), _vp(pp.h), _p(d, u8p), C.c_size_t(d.size),
                                         _vp(db.h) if db is not None else None)
        if not self.h:
            raise SpiralError(_err())

    def __del__(self):
        try:
            if not _finalizing():
                self.free()
        except Exception:
            pass

    def free(self):
        try:
            if getattr(self, "h", None):
                lib().sp_query_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    def sweep(self, db):
        _chk(lib().sp_query_sweep(_vp(self.h), _vp(db.h)))
        return self

    def sweep_scatter(self, db, G):
        """sweep with the column-interleaved partial layout of the distributed-fold path"""
        _chk(lib().sp_query_sweep_scatter(_vp(self.h), _vp(db.h), C.c_int(G)))
        return self

    def sweep_scatter_plane(self, db, G, plane):
        """one plane of sweep_scatter; plane region of the partial buffer = [g][r][crt][z][ii / G]"""
        _chk(lib().sp_query_sweep_scatter_plane(_vp(self.h), _vp(db.h), C.c_int(G), C.c_int(plane)))
        return self

    def fold_local(self, reduced_chunk_ptr, G):
        _chk(lib().sp_query_fold_local(_vp(self.h), C.c_void_p(reduced_chunk_ptr), C.c_int(G)))
        return self

    def fold_local_plane(self, reduced_plane_chunk_ptr, G, plane):
        _chk(lib().sp_query_fold_local_plane(_vp(self.h), C.c_void_p(reduced_plane_chunk_ptr), C.c_int(G), C.c_int(plane)))
        return self

    def fold_local_join(self):
        _chk(lib().sp_query_fold_local_join(_vp(self.h)))
        return self

    def stream2(self):
        lib().sp_query_stream2.restype = C.c_void_p
        return int(lib().sp_query_stream2(_vp(self.h)) or 0)

    def local_cts_ptr(self):
        lib().sp_query_local_cts_ptr.restype = C.c_void_p
        return int(lib().sp_query_local_cts_ptr(_vp(self.h)))

    def local_cts_words(self):
        lib().sp_query_local_cts_words.restype = C.c_size_t
        return int(lib().sp_query_local_cts_words(_vp(self.h)))

    def finish_gathered(self, gathered_ptr, G):
        n = self.params.get("response_bytes")
        out = np.zeros(n, dtype=np.uint8)
        ln = C.c_size_t(0)
        _chk(lib().sp_query_finish_gathered(_vp(self.h), C.c_void_p(gathered_ptr), C.c_int(G), _p(out, u8p),
                                            C.c_size_t(n), C.byref(ln)))
        return out[:ln.value].tobytes()

    def sync(self):
        _chk(lib().sp_query_sync(_vp(self.h)))

    def partial_ptr(self):
        ptr = lib().sp_query_partial_ptr(_vp(self.h))
        if not ptr:
            raise SpiralError(_err() or "no partial buffer")
        return int(ptr)

    def partial_words(self):
        return int(lib().sp_query_partial_words(_vp(self.h)))

    def stream(self):
        return int(lib().sp_query_stream(_vp(self.h)) or 0)

    def finish(self):
        n = self.params.get("response_bytes")
        out = np.zeros(n, dtype=np.uint8)
        ln = C.c_size_t(0)
        _chk(lib().sp_query_finish(_vp(self.h), _p(out, u8p), C.c_size_t(n), C.byref(ln)))
        return out[:ln.value].tobytes()

    def timings(self):
        t = (C.c_float * 4)()
        _chk(lib().sp_query_timings(_vp(self.h), t))
        return list(t)

    def bench_sweep(self, db, iters, per_plane=-1):
        """average milliseconds per db-sweep kernel launch (sweep_launches(db) launches per query by default;
        per_plane = 1 / 0 forces one launch per plane / one launch)"""
        ms = C.c_float(0)
        _chk(lib().sp_bench_sweep_ex(_vp(self.h), _vp(db.h), C.c_int(iters), C.c_int(per_plane), C.byref(ms)))
        return ms.value

    def sweep_launches(self, db):
        return int(lib().sp_sweep_launches(_vp(self.params.h), _vp(db.h)))


def bench_sweep_batch(runs, db, iters):
    """average milliseconds per batched database PASS of the begun queries `runs` (QueryRun objects, <= 16) -- the pass
    sp_process_query_batch issues for such a group (sp_bench_sweep_batch)"""
    ms = C.c_float(0)
    arr = (C.c_void_p * len(runs))(*[r.h for r in runs])
    _chk(lib().sp_bench_sweep_batch(arr, C.c_int(len(runs)), _vp(db.h), C.c_int(iters), C.byref(ms)))
    return ms.value


def process_query(params, public_params, query, db):
    """server.rs:650-741: process_query(params, public_params, query, db) -> Vec<u8>"""
    q = query.data if isinstance(query, Query) else bytes(query)
    d = _bytes(q)
    n = params.get("response_bytes")
    out = np.zeros(n, dtype=np.uint8)
    ln = C.c_size_t(0)
    _chk(lib().sp_process_query(_vp(params.h), _vp(public_params.h), _p(d, u8p), C.c_size_t(d.size), _vp(db.h),
                                _p(out, u8p), C.c_size_t(n), C.byref(ln)))
    return out[:ln.value].tobytes()


def process_query_batch(params, public_params_list, queries, db):
    """B queries against one database pass per group of <= 16 (sp_process_query_batch; BASELINE configs[4]).
    Equivalent to [process_query(params, pp_i, q_i, db) for i in ...] -- the reference's per-request loop
    (lib/server/src/bin/server.rs:152-158)."""
    B = len(queries)
    if not isinstance(public_params_list, (list, tuple)):
        public_params_list = [public_params_list] * B
    qs = [(_q.data if isinstance(_q, Query) else bytes(_q)) for _q in queries]
    bufs = [_bytes(q) for q in qs]
    n = params.get("response_bytes")
    out = np.zeros(B * n, dtype=np.uint8)
    ln = C.c_size_t(0)
    pp_arr = (C.c_void_p * B)(*[pp.h for pp in public_params_list])
    q_arr = (u8p * B)(*[_p(b, u8p) for b in bufs])
    l_arr = (C.c_size_t * B)(*[b.size for b in bufs])
    _chk(lib().sp_process_query_batch(_vp(params.h), pp_arr, q_arr, l_arr, C.c_int(B), _vp(db.h), _p(out, u8p),
                                      C.c_size_t(n), C.byref(ln)))
    return [out[i * n:(i + 1) * n].tobytes() for i in range(B)]


def expand_query(params, public_params, query):
    """server.rs:525-591 -> (v_reg_reoriented, v_folding)"""
    q = query.data if isinstance(query, Query) else bytes(query)
    d = _bytes(q)
    v_reg = np.zeros(params.dim0 * 2 * params.poly_len, dtype=np.uint64)
    v_fold = np.zeros(max(params.db_dim_2, 1) * 2 * 2 * params.t_gsw * params.ntt_words, dtype=np.uint64)
    _chk(lib().sp_expand_query(_vp(params.h), _vp(public_params.h), _p(d, u8p), C.c_size_t(d.size), _p(v_reg),
                               _p(v_fold)))
    return v_reg, v_fold[:params.db_dim_2 * 2 * 2 * params.t_gsw * params.ntt_words]


def coefficient_expansion(params, public_params, v, g, stop_round, max_bits_to_gen_right):
    """server.rs:19-121 (v_w_left / v_w_right come from public_params, v_neg1 is internal)"""
    v = _u64arr(v).copy()
    _chk(lib().sp_coefficient_expansion(_vp(params.h), _vp(public_params.h), _p(v), C.c_size_t(g),
                                        C.c_size_t(stop_round), C.c_size_t(max_bits_to_gen_right)))
    return v


def regev_to_gsw(params, public_params, v_inp, num_gsw):
    """server.rs:123-151 with V = public_params.v_conversion[0], idx_factor 1, idx_offset 0"""
    v_inp = _u64arr(v_inp)
    out = np.zeros(num_gsw * 2 * 2 * params.t_gsw * params.ntt_words, dtype=np.uint64)
    _chk(lib().sp_regev_to_gsw(_vp(params.h), _vp(public_params.h), _p(v_inp), _p(out), C.c_size_t(num_gsw)))
    return out


def get_v_folding_neg(params, v_folding):
    """server.rs:505-523"""
    v_folding = _u64arr(v_folding)
    out = np.zeros_like(v_folding)
    _chk(lib().sp_get_v_folding_neg(_vp(params.h), _p(v_folding), _p(out)))
    return out


def multiply_reg_by_database(params, db, v_firstdim, dim0=None, num_per=None):
    """server.rs:155-221: one plane `db` in reference layout -> num_per 2x1 NTT ciphertexts"""
    dim0 = dim0 or params.dim0
    num_per = num_per or params.num_per
    db, v_firstdim = _u64arr(db), _u64arr(v_firstdim)
    out = np.zeros(num_per * 2 * params.ntt_words, dtype=np.uint64)
    _chk(lib().sp_multiply_reg_by_database(_vp(params.h), _p(db), _p(v_firstdim), C.c_size_t(dim0),
                                           C.c_size_t(num_per), _p(out)))
    return out


def fold_ciphertexts(params, v_cts, v_folding, v_folding_neg):
    """server.rs:388-427: returns v_cts with the folded ciphertext in slot 0"""
    cts = _u64arr(v_cts).copy()
    num_per = cts.size // (2 * params.poly_len)
    _chk(lib().sp_fold_ciphertexts(_vp(params.h), _p(cts), C.c_size_t(num_per), _p(_u64arr(v_folding)),
                                   _p(_u64arr(v_folding_neg))))
    return cts


def fold_ciphertexts_fused(params, v_cts, v_folding, fused_min_pairs=0):
    """fold_ciphertexts the way process_query runs it (server.rs:388-427 with v_folding_neg = G - v_folding formed
    on the device): levels with >= fused_min_pairs pairs use the fused kernel (0 = library default, 1 = every level)."""
    cts = _u64arr(v_cts).copy()
    num_per = cts.size // (2 * params.poly_len)
    _chk(lib().sp_fold_ciphertexts_fused(_vp(params.h), _p(cts), C.c_size_t(num_per), _p(_u64arr(v_folding)),
                                         C.c_long(fused_min_pairs)))
    return cts


def paths_taken(reset=True):
    """Names of the kernels / flows this thread's library calls went through since the last reset
    (sp_paths_taken / sp_path_name, include/spiral_hip.h)."""
    L = lib()
    L.sp_paths_taken.restype = C.c_uint64
    L.sp_path_name.restype = C.c_char_p
    mask = int(L.sp_paths_taken(C.c_int(1 if reset else 0)))
    names, bit = set(), 0
    while True:
        nm = L.sp_path_name(C.c_int(bit))
        if nm is None:
            break
        if mask >> bit & 1:
            names.add(nm.decode())
        bit += 1
    return names


def pack(params, public_params, v_ct):
    """server.rs:429-468 with v_w = public_params.v_packing"""
    v_ct = _u64arr(v_ct)
    out = np.zeros((params.n + 1) * params.n * params.ntt_words, dtype=np.uint64)
    _chk(lib().sp_pack(_vp(params.h), _vp(public_params.h), _p(v_ct), _p(out)))
    return out


def encode(params, v_packed_ct):
    """server.rs:470-503"""
    v = _u64arr(v_packed_ct)
    n = params.get("response_bytes")
    out = np.zeros(n, dtype=np.uint8)
    ln = C.c_size_t(0)
    _chk(lib().sp_encode(_vp(params.h), _p(v), _p(out, u8p), C.c_size_t(n), C.byref(ln)))
    return out[:ln.value].tobytes()


# ------------------------------------------------------------------------------------------------
class Server:
    """ServerState of lib/server/src/bin/server.rs:21-28 without the HTTP transport (sp_server_*): POST /setup and POST
    /private-read bodies in, response bodies out; public parameters stay device resident per client uuid and a list of
    queries goes through the batch scheduler (<= 16 queries per pass over the database)."""

    def __init__(self, params, db):
        L = lib()
        L.sp_server_create.restype = C.c_void_p
        L.sp_server_clients.restype = C.c_size_t
        L.sp_server_private_read_json_bound.restype = C.c_size_t
        self.params, self.db = params, db
        self.h = L.sp_server_create(_vp(params.h), _vp(db.h))
        if not self.h:
            raise SpiralError(_err())

    def __del__(self):
        try:
            if getattr(self, "h", None) and not _finalizing():
                lib().sp_server_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    def clients(self):
        return int(lib().sp_server_clients(_vp(self.h)))

    def setup(self, pp_bytes):
        """/setup on the decoded bytes -> uuid string"""
        d = _bytes(pp_bytes)
        out = C.create_string_buffer(37)
        _chk(lib().sp_server_setup(_vp(self.h), _p(d, u8p), C.c_size_t(d.size), out))
        return out.value.decode()

    def setup_json(self, body):
        """/setup on the HTTP body (JSON string of base64) -> '{"uuid":"..."}'"""
        b = body.encode() if isinstance(body, str) else bytes(body)
        out = C.create_string_buffer(128)
        n = C.c_size_t(0)
        _chk(lib().sp_server_setup_json(_vp(self.h), b, C.c_size_t(len(b)), out, C.c_size_t(128), C.byref(n)))
        return out.raw[:n.value].decode()

    def forget(self, uuid):
        _chk(lib().sp_server_forget(_vp(self.h), uuid.encode()))

    def private_read(self, requests):
        """/private-read on decoded requests (uuid || query, or pp || query for direct-upload params) -> list of responses"""
        n = len(requests)
        bufs = [_bytes(r) for r in requests]
        rb = self.params.get("response_bytes")
        out = np.zeros(max(n, 1) * rb, dtype=np.uint8)
        lens = (C.c_size_t * max(n, 1))()
        r_arr = (u8p * max(n, 1))(*[_p(b, u8p) for b in bufs])
        l_arr = (C.c_size_t * max(n, 1))(*[b.size for b in bufs])
        _chk(lib().sp_server_private_read(_vp(self.h), r_arr, l_arr, C.c_int(n), _p(out, u8p), C.c_size_t(rb), lens))
        return [out[i * rb:i * rb + lens[i]].tobytes() for i in range(n)]

    def private_read_json(self, body):
        """/private-read on the HTTP body (JSON list of base64 strings) -> JSON list of base64 strings"""
        b = body.encode() if isinstance(body, str) else bytes(body)
        cap = int(lib().sp_server_private_read_json_bound(_vp(self.h), C.c_int(b.count(b",") + 1)))
        out = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        _chk(lib().sp_server_private_read_json(_vp(self.h), b, C.c_size_t(len(b)), out, C.c_size_t(cap), C.byref(n)))
        return out.raw[:n.value].decode()
