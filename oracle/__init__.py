"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes front-end over ``oracle/liboracle.so``, the scalar CPU restatement of
``lib/spiral-rs`` (see ``oracle/spiral_oracle.h`` for the pinning status).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; ``sdk_amd`` never does.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile liboracle.so with g++ (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("spiral_oracle.cpp", "oracle_capi.cpp", "sparse_server.cpp", "spiral_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class avx2_bodies:
    """Context manager: inside it the restatement runs the reference's `cfg(target_feature = "avx2")` bodies (ntt.rs:115-210,
    260-365; poly.rs:407-426, 460-481) instead of the scalar ones.  Process-wide switch, restored on exit."""

    def __init__(self, on=True):
        self.on = 1 if on else 0

    def __enter__(self):
        self.prev = lib().orc_set_avx2_bodies(self.on)
        return self

    def __exit__(self, *exc):
        lib().orc_set_avx2_bodies(self.prev)
        return False


def lib():
    global _LIB
    if _LIB is None:
        # libgomp reads this when it is loaded; a 256-thread team on a shared host makes the small
        # parallel loops of the restatement slower, not faster
        os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        # (thread binding -- OMP_PLACES=cores, OMP_PROC_BIND=spread, what keeps a team's first-touched pages local on a
        # two-socket host -- is the CALLER's choice: bench.py's cpu_baseline sets it for a single-rank run; binding pins the
        # loading thread to one core, which a multi-rank launch or a test process must not inherit)
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        L = _LIB
        if os.environ.get("ORACLE_TUNE_ALLOCATOR", "1") != "0":
            L.orc_tune_allocator()       # large temporaries from the arenas, not mmap / munmap per call (oracle_capi.cpp)
        L.orc_last_error.restype = C.c_char_p
        L.orc_params_new.restype = C.c_void_p
        L.orc_params_new.argtypes = [C.c_uint64] * 12 + [C.c_int]
        L.orc_params_init.restype = C.c_void_p
        L.orc_params_init.argtypes = [C.c_uint64, u64p, C.c_uint64] + [C.c_uint64] * 7 + [C.c_int] + [C.c_uint64] * 5
        L.orc_params_free.argtypes = [C.c_void_p]
        L.orc_params_get.restype = C.c_uint64
        L.orc_params_get.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_client_new.restype = C.c_void_p
        L.orc_client_new.argtypes = [C.c_void_p]
        L.orc_client_free.argtypes = [C.c_void_p]
        for name in ("orc_div2_uint_mod", "orc_barrett_raw_u64", "orc_barrett_reduction_u128_raw", "orc_rescale",
                     "orc_recenter_mod", "orc_recenter", "orc_reverse_bits", "orc_exponentiate_uint_mod",
                     "orc_invert_uint_mod", "orc_get_minimal_primitive_root", "orc_crt_compose_2", "orc_calc_index",
                     "orc_read_arbitrary_bits", "orc_get_bits_per"):
            getattr(L, name).restype = C.c_uint64
        for name in ("orc_client_generate_keys", "orc_client_generate_query", "orc_client_decode_response",
                     "orc_item_to_vec", "orc_process_query", "orc_pp_deserialize_flat", "orc_encode",
                     "orc_process_query_timed", "orc_process_query_synth"):
            getattr(L, name).restype = C.c_int64
    return _LIB


def _p(a, typ=u64p):
    return a.ctypes.data_as(typ)


def _u64(v):
    return C.c_uint64(int(v))


def _vp(h):
    return C.c_void_p(h)


class OracleError(RuntimeError):
    pass


def _chk(rc):
    if rc < 0:
        raise OracleError(lib().orc_last_error().decode() or f"oracle rc={rc}")
    return rc


class Params:
    """Mirror of spiral-rs ``Params`` (params.rs:49-82) built via ``params_from_json`` (util.rs:219-263)."""

    FIELDS = ("n", "nu_1", "nu_2", "p", "q2_bits", "t_gsw", "t_conv", "t_exp_left", "t_exp_right", "instances",
              "db_item_size", "version")

    def __init__(self, cfg):
        if isinstance(cfg, str):
            cfg = json.loads(cfg.replace("'", '"'))
        self.cfg = dict(cfg)
        a = [int(cfg.get(k, 0)) for k in self.FIELDS]
        self.h = lib().orc_params_new(*[_u64(x) for x in a], C.c_int(1 if "direct_upload" in cfg else 0))
        if not self.h:
            raise OracleError(lib().orc_last_error().decode())

    @classmethod
    def init_raw(cls, poly_len, moduli, n, p, q2_bits, t_conv, t_exp_left, t_exp_right, t_gsw, expand_queries, nu_1,
                 nu_2, instances, db_item_size, version):
        self = cls.__new__(cls)
        self.cfg = None
        m = np.array(moduli, dtype=np.uint64)
        self.h = lib().orc_params_init(_u64(poly_len), _p(m), _u64(len(moduli)), _u64(n), _u64(p), _u64(q2_bits),
                                       _u64(t_conv), _u64(t_exp_left), _u64(t_exp_right), _u64(t_gsw),
                                       C.c_int(int(expand_queries)), _u64(nu_1), _u64(nu_2), _u64(instances),
                                       _u64(db_item_size), _u64(version))
        if not self.h:
            raise OracleError(lib().orc_last_error().decode())
        return self

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_params_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    def get(self, name):
        v = lib().orc_params_get(_vp(self.h), name.encode())
        if v == 0xFFFFFFFFFFFFFFFF:
            raise KeyError(name)
        return int(v)

    def __getattr__(self, name):
        if name.startswith("_") or name in ("h", "cfg"):
            raise AttributeError(name)
        try:
            return self.get(name)
        except KeyError:
            raise AttributeError(name)

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
        lib().orc_ntt_table(_vp(self.h), _u64(crt), _u64(which), _p(out))
        return out

    def db_words(self):
        return self.instances * self.n * self.n * self.num_items * self.poly_len

    # ---- L0
    def ntt_forward(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint64).copy()
        lib().orc_ntt_forward(_vp(self.h), _p(a), _u64(a.size // self.ntt_words))
        return a

    def ntt_inverse(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint64).copy()
        lib().orc_ntt_inverse(_vp(self.h), _p(a), _u64(a.size // self.ntt_words))
        return a

    def to_ntt(self, raw, no_reduce=False):
        raw = np.ascontiguousarray(raw, dtype=np.uint64)
        cnt = raw.size // self.poly_len
        out = np.zeros(cnt * self.ntt_words, dtype=np.uint64)
        lib().orc_to_ntt(_vp(self.h), _p(raw), _p(out), _u64(cnt), C.c_int(int(no_reduce)))
        return out

    def from_ntt(self, ntt):
        ntt = np.ascontiguousarray(ntt, dtype=np.uint64)
        cnt = ntt.size // self.ntt_words
        out = np.zeros(cnt * self.poly_len, dtype=np.uint64)
        lib().orc_from_ntt(_vp(self.h), _p(ntt), _p(out), _u64(cnt))
        return out

    def multiply(self, a, ar, ac, b, bc):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        res = np.zeros(ar * bc * self.ntt_words, dtype=np.uint64)
        lib().orc_multiply(_vp(self.h), _p(a), _u64(ar), _u64(ac), _p(b), _u64(bc), _p(res))
        return res

    def add(self, a, b):
        """poly.rs:483-498 over flat NTT polynomials"""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        res = np.zeros(a.size, dtype=np.uint64)
        lib().orc_add(_vp(self.h), _p(a), _p(b), _u64(a.size // self.ntt_words), _p(res))
        return res

    def add_into(self, res, a):
        """poly.rs:500-512"""
        res = np.ascontiguousarray(res, dtype=np.uint64).copy()
        a = np.ascontiguousarray(a, dtype=np.uint64)
        lib().orc_add_into(_vp(self.h), _p(res), _p(a), _u64(a.size // self.ntt_words))
        return res

    def scalar_multiply(self, scalar, b):
        """poly.rs:575-588: (1x1 NTT polynomial) * every polynomial of b"""
        scalar = np.ascontiguousarray(scalar, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        res = np.zeros(b.size, dtype=np.uint64)
        lib().orc_scalar_multiply(_vp(self.h), _p(scalar), _p(b), _u64(b.size // self.ntt_words), _p(res))
        return res

    def automorph(self, a, t):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        res = np.zeros_like(a)
        lib().orc_automorph(_vp(self.h), _p(a), _u64(a.size // self.poly_len), _u64(t), _p(res))
        return res

    def gadget_invert_rdim(self, inp, rows_in, cols, rows_out, rdim):
        inp = np.ascontiguousarray(inp, dtype=np.uint64)
        out = np.zeros(rows_out * cols * self.poly_len, dtype=np.uint64)
        lib().orc_gadget_invert_rdim(_vp(self.h), _p(inp), _u64(rows_in), _u64(cols), _p(out), _u64(rows_out),
                                     _u64(rdim))
        return out

    def build_gadget(self, rows, cols):
        out = np.zeros(rows * cols * self.poly_len, dtype=np.uint64)
        lib().orc_build_gadget(_vp(self.h), _u64(rows), _u64(cols), _p(out))
        return out

    def get_bits_per(self, dim):
        return int(lib().orc_get_bits_per(_vp(self.h), _u64(dim)))

    def crt_compose_2(self, x, y):
        return int(lib().orc_crt_compose_2(_vp(self.h), _u64(x), _u64(y)))

    # ---- server.rs
    def generate_random_db_and_get_item(self, item_idx, seed=0x123456789):
        db = np.zeros(self.db_words(), dtype=np.uint64)
        item = np.zeros(self.instances * self.n * self.n * self.poly_len, dtype=np.uint64)
        _chk(lib().orc_generate_random_db_and_get_item(_vp(self.h), _u64(item_idx), _u64(seed), _p(db), _p(item)))
        return item, db

    def load_db_from_bytes(self, blob):
        b = np.frombuffer(bytes(blob), dtype=np.uint8)
        db = np.zeros(self.db_words(), dtype=np.uint64)
        _chk(lib().orc_load_db_from_bytes(_vp(self.h), _p(b, u8p), _u64(b.size), _p(db)))
        return db

    def item_to_vec(self, item):
        item = np.ascontiguousarray(item, dtype=np.uint64)
        out = np.zeros(self.instances * self.n * self.n * self.poly_len * 2 + 64, dtype=np.uint8)
        n = _chk(lib().orc_item_to_vec(_vp(self.h), _p(item), _p(out, u8p), _u64(out.size)))
        return out[:n].tobytes()

    def response_bytes(self):
        q1_bits = int(np.ceil(np.log2(4 * self.pt_modulus)))
        bits = self.instances * (self.q2_bits * self.n * self.poly_len + q1_bits * self.n * self.n * self.poly_len)
        return ((bits + 63) // 64) * 8

    def process_query(self, pp_bytes, q_bytes, db, timed=False):
        pp = np.frombuffer(pp_bytes, dtype=np.uint8)
        q = np.frombuffer(q_bytes, dtype=np.uint8)
        db = np.ascontiguousarray(db, dtype=np.uint64)
        out = np.zeros(self.response_bytes() + 64, dtype=np.uint8)
        if timed:
            t = (C.c_double * 4)()
            n = _chk(lib().orc_process_query_timed(_vp(self.h), _p(pp, u8p), _u64(pp.size), _p(q, u8p), _u64(q.size),
                                                   _p(db), _p(out, u8p), _u64(out.size), t))
            return out[:n].tobytes(), list(t)
        n = _chk(lib().orc_process_query(_vp(self.h), _p(pp, u8p), _u64(pp.size), _p(q, u8p), _u64(q.size), _p(db),
                                         _p(out, u8p), _u64(out.size)))
        return out[:n].tobytes()

    def process_query_synth(self, pp_bytes, q_bytes, seed, fold_classes=None, timed=False):
        """process_query (server.rs:650-741) over the synthetic database word(i) = synth_word(seed, i) (reference-layout
        index), generated row by row: response-byte parity at sizes (64-256 GiB encoded) no host buffer holds."""
        pp = np.frombuffer(pp_bytes, dtype=np.uint8)
        q = np.frombuffer(q_bytes, dtype=np.uint8)
        out = np.zeros(self.response_bytes() + 64, dtype=np.uint8)
        t = (C.c_double * 4)()
        if fold_classes is None:
            fold_classes = int(os.environ.get("OMP_NUM_THREADS", "1"))
        n = _chk(lib().orc_process_query_synth(_vp(self.h), _p(pp, u8p), _u64(pp.size), _p(q, u8p), _u64(q.size),
                                               _u64(seed), _u64(fold_classes), _p(out, u8p), _u64(out.size), t))
        return (out[:n].tobytes(), list(t)) if timed else out[:n].tobytes()

    def sweep_synth_row(self, seed, plane, z, v_reg, j0=0, nj=None):
        """(num_per, 4) partial first-dimension residues of rows [j0, j0 + nj) for one (plane, z) of the synthetic db:
        columns n0_0 (r0,crt0), n0_1 (r1,crt0), n1_0 (r0,crt1), n1_1 (r1,crt1)"""
        nj = self.dim0 - j0 if nj is None else nj
        v = np.ascontiguousarray(v_reg, dtype=np.uint64)
        out = np.zeros(self.num_per * 4, dtype=np.uint64)
        lib().orc_sweep_synth_row(_vp(self.h), _u64(seed), _u64(plane), _u64(z), _p(v), _u64(j0), _u64(nj), _p(out))
        return out.reshape(self.num_per, 4)

    def pp_poly_count(self):
        n_pack = self.n * (self.n + 1) * self.t_conv
        if not self.expand_queries:
            return n_pack
        right = (self.stop_round + 1) * 2 * self.t_exp_right
        if self.version > 0 and self.t_exp_left == self.t_exp_right:
            right = 0
        return n_pack + self.g * 2 * self.t_exp_left + right + 2 * 2 * self.t_conv

    def pp_deserialize_flat(self, pp_bytes):
        pp = np.frombuffer(pp_bytes, dtype=np.uint8)
        out = np.zeros(self.pp_poly_count() * self.ntt_words, dtype=np.uint64)
        n = _chk(lib().orc_pp_deserialize_flat(_vp(self.h), _p(pp, u8p), _u64(pp.size), _p(out), _u64(out.size)))
        assert n == out.size
        return out

    def query_deserialize_ct(self, q_bytes):
        q = np.frombuffer(q_bytes, dtype=np.uint8)
        out = np.zeros(2 * self.poly_len, dtype=np.uint64)
        _chk(lib().orc_query_deserialize_ct(_vp(self.h), _p(q, u8p), _u64(q.size), _p(out)))
        return out

    def expand_query(self, pp_bytes, q_bytes):
        pp = np.frombuffer(pp_bytes, dtype=np.uint8)
        q = np.frombuffer(q_bytes, dtype=np.uint8)
        v_reg = np.zeros(self.dim0 * 2 * self.poly_len, dtype=np.uint64)
        v_fold = np.zeros(self.db_dim_2 * 2 * 2 * self.t_gsw * self.ntt_words, dtype=np.uint64)
        _chk(lib().orc_expand_query(_vp(self.h), _p(pp, u8p), _u64(pp.size), _p(q, u8p), _u64(q.size), _p(v_reg),
                                    _p(v_fold)))
        return v_reg, v_fold

    def coefficient_expansion(self, pp_bytes, v, g, stop_round, max_bits_to_gen_right):
        pp = np.frombuffer(pp_bytes, dtype=np.uint8)
        v = np.ascontiguousarray(v, dtype=np.uint64).copy()
        _chk(lib().orc_coefficient_expansion(_vp(self.h), _p(pp, u8p), _u64(pp.size), _p(v), _u64(g), _u64(stop_round),
                                             _u64(max_bits_to_gen_right)))
        return v

    def regev_to_gsw(self, v_inp, v_conv, num_gsw):
        v_inp = np.ascontiguousarray(v_inp, dtype=np.uint64)
        v_conv = np.ascontiguousarray(v_conv, dtype=np.uint64)
        out = np.zeros(num_gsw * 2 * 2 * self.t_gsw * self.ntt_words, dtype=np.uint64)
        _chk(lib().orc_regev_to_gsw(_vp(self.h), _p(v_inp), _u64(v_inp.size // (2 * self.ntt_words)), _p(v_conv),
                                    _p(out), _u64(num_gsw)))
        return out

    def get_v_folding_neg(self, v_folding):
        v_folding = np.ascontiguousarray(v_folding, dtype=np.uint64)
        out = np.zeros_like(v_folding)
        _chk(lib().orc_get_v_folding_neg(_vp(self.h), _p(v_folding), _p(out)))
        return out

    def multiply_reg_by_database(self, db, v_firstdim, dim0=None, num_per=None):
        dim0 = dim0 or self.dim0
        num_per = num_per or self.num_per
        db = np.ascontiguousarray(db, dtype=np.uint64)
        v_firstdim = np.ascontiguousarray(v_firstdim, dtype=np.uint64)
        out = np.zeros(num_per * 2 * self.ntt_words, dtype=np.uint64)
        _chk(lib().orc_multiply_reg_by_database(_vp(self.h), _p(db), _p(v_firstdim), _u64(dim0), _u64(num_per), _p(out)))
        return out

    def fold_ciphertexts(self, cts, v_folding, v_folding_neg, nu=None):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).copy()
        num_per = cts.size // (2 * self.poly_len)
        nu = self.db_dim_2 if nu is None else nu
        v_folding = np.ascontiguousarray(v_folding, dtype=np.uint64)
        v_folding_neg = np.ascontiguousarray(v_folding_neg, dtype=np.uint64)
        _chk(lib().orc_fold_ciphertexts(_vp(self.h), _p(cts), _u64(num_per), _p(v_folding), _p(v_folding_neg), _u64(nu)))
        return cts

    def from_ntt_fold_parallel(self, cts_ntt, v_folding, v_folding_neg, nu=None, classes=1):
        """from_ntt of every ciphertext + fold_ciphertexts with the tree split over `classes` threads (results are
        those of the sequential fold); returns the folded raw ciphertext (2N words)"""
        cts = np.ascontiguousarray(cts_ntt, dtype=np.uint64)
        nu = self.db_dim_2 if nu is None else nu
        out = np.zeros(2 * self.poly_len, dtype=np.uint64)
        _chk(lib().orc_from_ntt_fold_parallel(_vp(self.h), _p(cts), _u64(cts.size // (2 * self.ntt_words)),
                                              _p(np.ascontiguousarray(v_folding, dtype=np.uint64)),
                                              _p(np.ascontiguousarray(v_folding_neg, dtype=np.uint64)), _u64(nu),
                                              _u64(classes), _p(out)))
        return out

    def pack(self, v_ct, v_w):
        v_ct = np.ascontiguousarray(v_ct, dtype=np.uint64)
        v_w = np.ascontiguousarray(v_w, dtype=np.uint64)
        out = np.zeros((self.n + 1) * self.n * self.ntt_words, dtype=np.uint64)
        _chk(lib().orc_pack(_vp(self.h), _p(v_ct), _p(v_w), _p(out)))
        return out

    def encode(self, v_packed):
        v_packed = np.ascontiguousarray(v_packed, dtype=np.uint64)
        out = np.zeros(self.response_bytes() + 64, dtype=np.uint8)
        n = _chk(lib().orc_encode(_vp(self.h), _p(v_packed), _p(out, u8p), _u64(out.size)))
        return out[:n].tobytes()

    def reorient_reg_ciphertexts(self, v_reg):
        v_reg = np.ascontiguousarray(v_reg, dtype=np.uint64)
        out = np.zeros(self.dim0 * 2 * self.poly_len, dtype=np.uint64)
        lib().orc_reorient_reg_ciphertexts(_vp(self.h), _p(v_reg), _p(out))
        return out


def sweep_rows(db_zslice, v_firstdim_zslice, nz, dim0, num_per, q0=268369921, q1=249561089):
    db_zslice = np.ascontiguousarray(db_zslice, dtype=np.uint64)
    v = np.ascontiguousarray(v_firstdim_zslice, dtype=np.uint64)
    out = np.zeros(nz * num_per * 4, dtype=np.uint64)
    lib().orc_sweep_rows(_p(db_zslice), _p(v), _u64(nz), _u64(dim0), _u64(num_per), _u64(q0), _u64(q1), _p(out))
    return out.reshape(nz, num_per, 4)


def sweep_rows_avx2(db_zslice, v_firstdim_zslice, nz, dim0, num_per, q0=268369921, q1=249561089):
    """all-core AVX2 form of sweep_rows (same results), for bench.py's cpu_baseline"""
    db_zslice = np.ascontiguousarray(db_zslice, dtype=np.uint64)
    v = np.ascontiguousarray(v_firstdim_zslice, dtype=np.uint64)
    out = np.zeros(nz * num_per * 4, dtype=np.uint64)
    lib().orc_sweep_rows_avx2(_p(db_zslice), _p(v), _u64(nz), _u64(dim0), _u64(num_per), _u64(q0), _u64(q1), _p(out))
    return out.reshape(nz, num_per, 4)


class SparseDb:
    """lib/server's SparseDb (db/sparse_db.rs:5-48) filled through update_item_raw (db/loading.rs:317-359), and
    lib/server's process_query over it (server.rs:17-99): the sparse caller the GPU library serves."""

    def __init__(self, params):
        L = lib()
        L.orc_sparse_db_new.restype = C.c_void_p
        L.orc_sparse_last_error.restype = C.c_char_p
        L.orc_process_query_sparse.restype = C.c_int64
        L.orc_sparse_db_polys.restype = C.c_uint64
        self.params = params
        self.h = L.orc_sparse_db_new()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_sparse_db_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    def update_item_raw(self, db_idx, data):
        d = np.frombuffer(bytes(data), dtype=np.uint8)
        if lib().orc_sparse_update_item_raw(_vp(self.params.h), _vp(self.h), _u64(db_idx), _p(d, u8p), _u64(d.size)) != 0:
            raise OracleError(lib().orc_sparse_last_error().decode())
        return self

    def polys(self):
        return int(lib().orc_sparse_db_polys(_vp(self.h)))

    def process_query(self, pp_bytes, q_bytes):
        pp = np.frombuffer(pp_bytes, dtype=np.uint8)
        q = np.frombuffer(q_bytes, dtype=np.uint8)
        out = np.zeros(self.params.response_bytes() + 64, dtype=np.uint8)
        n = lib().orc_process_query_sparse(_vp(self.params.h), _p(pp, u8p), _u64(pp.size), _p(q, u8p), _u64(q.size),
                                           _vp(self.h), _p(out, u8p), _u64(out.size))
        if n < 0:
            raise OracleError(lib().orc_sparse_last_error().decode())
        return out[:n].tobytes()

    def to_dense(self):
        """the dense database with zero polynomials for absent items (reference layout)"""
        out = np.zeros(self.params.db_words(), dtype=np.uint64)
        lib().orc_sparse_to_dense(_vp(self.params.h), _vp(self.h), _p(out))
        return out


def synth_word(seed, idx):
    lib().orc_synth_word.restype = C.c_uint64
    return int(lib().orc_synth_word(_u64(seed), _u64(idx)))


class Client:
    """Mirror of spiral-rs ``Client`` (client.rs:361-810); secrets from caller-provided 32-byte seeds."""

    def __init__(self, params):
        self.params = params
        self.h = lib().orc_client_new(_vp(params.h))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_client_free(_vp(self.h))
                self.h = None
        except Exception:
            pass

    @staticmethod
    def _seed(seed):
        if isinstance(seed, int):
            seed = seed.to_bytes(32, "little")
        assert len(seed) == 32
        return np.frombuffer(bytes(seed), dtype=np.uint8)

    def generate_keys(self, seed=1):
        p = self.params
        out = np.zeros(p.setup_bytes, dtype=np.uint8)
        s = self._seed(seed)
        n = _chk(lib().orc_client_generate_keys(_vp(self.h), _p(s, u8p), _p(out, u8p), _u64(out.size)))
        assert n == p.setup_bytes, (n, p.setup_bytes)
        return out.tobytes()

    def generate_query(self, idx, seed=2):
        p = self.params
        out = np.zeros(p.query_bytes, dtype=np.uint8)
        s = self._seed(seed)
        n = _chk(lib().orc_client_generate_query(_vp(self.h), _u64(idx), _p(s, u8p), _p(out, u8p), _u64(out.size)))
        assert n == p.query_bytes, (n, p.query_bytes)
        return out.tobytes()

    def decode_response(self, data):
        d = np.frombuffer(data, dtype=np.uint8)
        out = np.zeros(len(data) * 2 + 4096, dtype=np.uint8)
        n = _chk(lib().orc_client_decode_response(_vp(self.h), _p(d, u8p), _u64(d.size), _p(out, u8p), _u64(out.size)))
        return out[:n].tobytes()

    def decrypt_reg(self, cts_ntt):
        p = self.params
        cts = np.ascontiguousarray(cts_ntt, dtype=np.uint64)
        cnt = cts.size // (2 * p.ntt_words)
        out = np.zeros(cnt * p.poly_len, dtype=np.uint64)
        lib().orc_client_decrypt_reg(_vp(self.h), _p(cts), _u64(cnt), _p(out))
        return out.reshape(cnt, p.poly_len)

    def encrypt_reg(self, pt_raw, seed=3, seed_pub=4):
        p = self.params
        pt = np.ascontiguousarray(pt_raw, dtype=np.uint64)
        out = np.zeros(2 * p.ntt_words, dtype=np.uint64)
        s, sp = self._seed(seed), self._seed(seed_pub)
        lib().orc_client_encrypt_reg(_vp(self.h), _p(pt), _p(s, u8p), _p(sp, u8p), _p(out))
        return out


# ---- free functions used by the KAT tests
def words_first_touch(nz, words_per_z, seed=5):
    """nz * words_per_z database words (residues < 2^28), written by the OpenMP team with the z-row schedule of
    sweep_rows_avx2: pages land on the NUMA node of the thread that will read them"""
    out = np.empty(nz * words_per_z, dtype=np.uint64)
    lib().orc_fill_words_first_touch(_p(out), _u64(nz), _u64(words_per_z), _u64(seed))
    return out


def set_threads(n):
    """OpenMP team size of the parallel sections (returns the previous maximum)"""
    return int(lib().orc_set_threads(C.c_int(n)))


def get_barrett_crs(m):
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_get_barrett_crs(_u64(m), _p(o))
    return int(o[0]), int(o[1])


def divide_uint192(num, den):
    n = np.array(num, dtype=np.uint64)
    r = np.zeros(3, dtype=np.uint64)
    q = np.zeros(3, dtype=np.uint64)
    lib().orc_divide_uint192(_p(n), _u64(den), _p(r), _p(q))
    return [int(x) for x in r], [int(x) for x in q]


def chacha20_block(state16):
    i = np.array(state16, dtype=np.uint32)
    o = np.zeros(16, dtype=np.uint32)
    lib().orc_chacha20_block(_p(i, u32p), _p(o, u32p))
    return o


def chacha20_rng_u64(seed32, count):
    s = np.frombuffer(bytes(seed32), dtype=np.uint8)
    o = np.zeros(count, dtype=np.uint64)
    lib().orc_chacha20_rng_u64(_p(s, u8p), _p(o), _u64(count))
    return o


def scalar(name, *args):
    return int(getattr(lib(), "orc_" + name)(*[_u64(a) for a in args]))


def read_arbitrary_bits(data, off, nb):
    d = np.frombuffer(bytes(data), dtype=np.uint8)
    return int(lib().orc_read_arbitrary_bits(_p(d, u8p), _u64(off), _u64(nb)))


def write_arbitrary_bits(buf, val, off, nb):
    """buf: writable np.uint8 array"""
    lib().orc_write_arbitrary_bits(_p(buf, u8p), _u64(val), _u64(off), _u64(nb))


def calc_index(ind, lens):
    a = np.array(ind, dtype=np.uint64)
    b = np.array(lens, dtype=np.uint64)
    return int(lib().orc_calc_index(_p(a), _p(b), _u64(len(ind))))
