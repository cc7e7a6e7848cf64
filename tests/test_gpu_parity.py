"""GPU parity: every stage of the HIP answer path, called through the C ABI, against the CPU oracle on
identical seeded inputs -- bit-exact (integer arithmetic).  Mirrors the reference's test ladder
(SURVEY.md App. F): L0 kernels, stage functions, end-to-end response bytes, then decrypt."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import C1, FAST, FAST56, P2, ROOT, SERVER_DEFAULT, SMALL_INST2

pytestmark = pytest.mark.gpu

Q0, Q1 = 268369921, 249561089
Q = Q0 * Q1


@pytest.fixture(scope="module")
def sp():
    import sdk_amd
    assert sdk_amd.lib().sp_device_count() >= 1, "no HIP device visible"
    return sdk_amd


def _pair(sp, oracle_mod, cfg):
    return sp.Params(cfg), oracle_mod.Params(cfg)


# ---------------------------------------------------------------------------------------- L0
def test_params_tables_match(sp, oracle_mod):
    p, o = _pair(sp, oracle_mod, P2)
    for c in range(2):
        for w in range(4):
            assert (p.ntt_table(c, w) == o.ntt_table(c, w)).all()
    for k in ("setup_bytes", "query_bytes", "g", "stop_round", "num_items", "modulus"):
        assert p.get(k) == o.get(k)


def _edge_polys(rng, n_random=6):
    N = 2048
    polys = [np.zeros(2 * N, dtype=np.uint64)]
    a = np.zeros(2 * N, dtype=np.uint64); a[0] = 100; a[N] = 100; polys.append(a)
    polys.append(np.concatenate([np.full(N, Q0 - 1, dtype=np.uint64), np.full(N, Q1 - 1, dtype=np.uint64)]))
    polys.append(np.full(2 * N, 100, dtype=np.uint64))
    for _ in range(n_random):
        polys.append(np.concatenate([rng.integers(0, Q0, N, dtype=np.uint64), rng.integers(0, Q1, N, dtype=np.uint64)]))
    return np.concatenate(polys)


def test_ntt_forward_inverse(sp, oracle_mod):
    p, o = _pair(sp, oracle_mod, FAST)
    x = _edge_polys(np.random.default_rng(1))
    f_gpu = sp.ntt_forward(p, x)
    assert (f_gpu == o.ntt_forward(x)).all()
    assert (sp.ntt_inverse(p, f_gpu) == x).all()
    assert (sp.ntt_inverse(p, x) == o.ntt_inverse(x)).all()
    # ntt.rs:400-423 KATs through the GPU
    assert (f_gpu[2 * 2048:4 * 2048] == 100).all()


def test_to_ntt_from_ntt(sp, oracle_mod):
    p, o = _pair(sp, oracle_mod, FAST)
    rng = np.random.default_rng(2)
    raw = rng.integers(0, Q, 12 * 2048, dtype=np.uint64)
    raw[:2048] = 0
    raw[2048:4096] = Q           # the Q-for-zero values of automorph/invert
    raw[4096:6144] = Q - 1
    raw[6144] = (1 << 64) - 1    # Barrett on the full u64 range (arith.rs:122-134)
    ntt_gpu = sp.to_ntt(p, raw)
    assert (ntt_gpu == o.to_ntt(raw)).all()
    back = sp.from_ntt(p, ntt_gpu)
    assert (back == o.from_ntt(ntt_gpu)).all()
    assert (back[6145:] == raw[6145:] % np.uint64(Q)).all()


def test_from_ntt_small_and_boundary_coefficients(sp, oracle_mod):
    """from_ntt (poly.rs:646-663 -> params.rs:207-214 crt_compose_2 -> arith.rs barrett_reduction_u128) on polynomials whose
    coefficients sit where the reference's add_u64 quirk could matter if it ever did: 0, 1, every value below 2^14, the
    values just below Q, powers of two.  GPU (exact Garner form) == oracle (literal restatement) == the raw input."""
    p, o = _pair(sp, oracle_mod, FAST)
    N = 2048
    rng = np.random.default_rng(17)
    polys = [np.arange(k * N, (k + 1) * N, dtype=np.uint64) for k in range(8)]                     # 0 .. 2^14 - 1
    polys += [np.uint64(Q - 1) - np.arange(k * N, (k + 1) * N, dtype=np.uint64) for k in range(2)]  # Q - 1 downwards
    polys.append(np.array([(1 << (i % 56)) % Q for i in range(N)], dtype=np.uint64))
    polys.append(np.array([((1 << (i % 56)) - 1) % Q for i in range(N)], dtype=np.uint64))
    polys.append(rng.integers(0, 1 << 13, N, dtype=np.uint64))
    polys.append(np.uint64(Q) - rng.integers(1, 1 << 13, N, dtype=np.uint64))
    raw = np.concatenate(polys)
    ntt = o.to_ntt(raw)
    assert (sp.to_ntt(p, raw) == ntt).all()
    back = sp.from_ntt(p, ntt)
    assert (back == o.from_ntt(ntt)).all()
    assert (back == raw).all()


def test_add_and_scalar_multiply(sp, oracle_mod):
    """poly.rs:483-512 add / add_into and :575-588 scalar_multiply as stage-level exports (the kernels the expansion and
    the packing use), against the oracle: edge polynomials (0, q - 1 in every slot) and random ones."""
    p, o = _pair(sp, oracle_mod, FAST)
    rng = np.random.default_rng(23)
    a = _edge_polys(rng, 4)
    b = np.roll(_edge_polys(np.random.default_rng(24), 4).reshape(-1, 2 * 2048), 3, axis=0).reshape(-1)
    assert (sp.add(p, a, b) == o.add(a, b)).all()
    assert (sp.add_into(p, a, b) == o.add_into(a, b)).all()
    assert (sp.add(p, a, a) == o.add(a, a)).all()
    scal = np.concatenate([rng.integers(0, Q0, 2048, dtype=np.uint64), rng.integers(0, Q1, 2048, dtype=np.uint64)])
    assert (sp.scalar_multiply(p, scal, a) == o.scalar_multiply(scal, a)).all()
    top = np.concatenate([np.full(2048, Q0 - 1, dtype=np.uint64), np.full(2048, Q1 - 1, dtype=np.uint64)])
    assert (sp.scalar_multiply(p, top, a) == o.scalar_multiply(top, a)).all()


def test_reorient_reg_ciphertexts_export(sp, oracle_mod):
    """util.rs:323-355 through the stand-alone export sp_reorient_reg_ciphertexts (so far only checked inside
    expand_query): dim0 expanded NTT ciphertexts -> [z][j][r] packed words, canonical residues."""
    for cfg in (FAST, dict(FAST, nu_1=4), C1):
        p, o = _pair(sp, oracle_mod, cfg)
        rng = np.random.default_rng(31 + cfg["nu_1"])
        n = o.dim0 * 2 * 2 * 2048
        v = np.empty(n, dtype=np.uint64)
        vv = v.reshape(o.dim0, 2, 2, 2048)
        vv[:, :, 0, :] = rng.integers(0, Q0, (o.dim0, 2, 2048), dtype=np.uint64)
        vv[:, :, 1, :] = rng.integers(0, Q1, (o.dim0, 2, 2048), dtype=np.uint64)
        vv[0, 0, 0, :] = Q0 - 1
        vv[0, 1, 1, :] = Q1 - 1
        vv[1 % o.dim0, 0, :, :5] = 0
        assert (sp.reorient_reg_ciphertexts(p, v) == o.reorient_reg_ciphertexts(v)).all()


def test_multiply(sp, oracle_mod):  # poly.rs:731-743 + random 2x16 * 16x1 (the fold shape)
    p, o = _pair(sp, oracle_mod, FAST)
    m1 = np.zeros(2048, dtype=np.uint64); m1[1] = 100
    m2 = np.zeros(2048, dtype=np.uint64); m2[1] = 7
    m3 = sp.from_ntt(p, sp.multiply(p, sp.to_ntt(p, m1), 1, 1, sp.to_ntt(p, m2), 1))
    assert int(m3[2]) == 700 and int(m3.sum()) == 700
    rng = np.random.default_rng(3)
    a = _rand_ntt(rng, 2 * 16)
    b = _rand_ntt(rng, 16 * 3)
    assert (sp.multiply(p, a, 2, 16, b, 3) == o.multiply(a, 2, 16, b, 3)).all()


def _rand_ntt(rng, n_polys):
    N = 2048
    out = np.zeros((n_polys, 2, N), dtype=np.uint64)
    out[:, 0] = rng.integers(0, Q0, (n_polys, N), dtype=np.uint64)
    out[:, 1] = rng.integers(0, Q1, (n_polys, N), dtype=np.uint64)
    return out.reshape(-1)


def test_automorph_and_gadget(sp, oracle_mod):
    import sdk_amd.spiral as S
    p, o = _pair(sp, oracle_mod, FAST56)
    rng = np.random.default_rng(4)
    a = rng.integers(0, Q, 3 * 2048, dtype=np.uint64)
    a[:2048:7] = 0
    for t in (2049, 1025, 5, 3):
        assert (S.automorph(p, a, t) == o.automorph(a, t)).all()
    inp = rng.integers(0, Q, 2 * 2048, dtype=np.uint64)
    inp[5] = Q
    for rows_out, rdim in ((16, 2), (8, 1), (56, 1), (4, 1), (8, 2)):
        rows_in = 2
        g = S.gadget_invert_rdim(p, inp, rows_in, 1, rows_out, rdim)
        assert (g == o.gadget_invert_rdim(inp, rows_in, 1, rows_out, rdim)).all(), (rows_out, rdim)


def test_resident_data_survive_database_allocation(sp, oracle_mod):
    """The public parameters and tables of a handle are device buffers that exist BEFORE its database is allocated:
    they must be the same bytes afterwards.  (A physically contiguous database allocation made the driver move them
    and changed their contents on some machines, profiles/r02_stale_reads.md -- the switch db_contiguous is off.)
    Kernels (through the caches) and a device-to-host copy must also see the same resident data."""
    import ctypes as C
    for cfg, seed in ((FAST, 41), (dict(FAST, nu_2=1), 42), (P2, 43)):
        o, cl, pp, q = _session(oracle_mod, cfg, 5, seed)
        flat = o.pp_deserialize_flat(pp)
        p = sp.Params(cfg)
        gpp = sp.PublicParameters.deserialize(p, pp)
        assert (gpp.export() == flat).all()
        gdb = sp.Database(p).fill_synthetic(0x1234 + seed)
        assert (gpp.export() == flat).all(), "public parameters changed while the database was allocated"
        out = (C.c_uint64 * 24)()
        assert sp.lib().sp_debug_resident_check(C.c_void_p(p.h), C.c_void_p(gpp.h), out, 24) == 0
        for i, name in enumerate(("tw", "neg1", "gadget", "lists", "pp.all", "pp.pack_cat")):
            kernel_view, copy_view, after_sync, host = out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]
            assert kernel_view == copy_view == after_sync, name
            assert host == 0 or host == copy_view, name
        del gdb


# ------------------------------------------------------------------------------------ stages
def _session(oracle_mod, cfg, idx, seed):
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(seed)
    q = cl.generate_query(idx, seed + 1)
    return o, cl, pp, q


_CASE = {}


def _oracle_case(oracle_mod, cfg, idx, key_seed, query_seed):
    """The CPU side of one end-to-end case -- client, keys, query, random database, the oracle's response -- computed once
    and shared by the consecutive parametrizations that only vary a switch of the GPU library (the oracle's CPU time, not
    the GPU's, is most of this suite).  One case is kept at a time: the databases are gigabytes."""
    import json
    key = (json.dumps(cfg, sort_keys=True), idx, key_seed, query_seed)
    if key not in _CASE:
        _CASE.clear()
        o = oracle_mod.Params(cfg)
        cl = oracle_mod.Client(o)
        pp = cl.generate_keys(key_seed)
        q = cl.generate_query(idx, query_seed)
        item, db = o.generate_random_db_and_get_item(idx)
        _CASE[key] = {"o": o, "cl": cl, "pp": pp, "q": q, "item": item, "db": db, "resp": o.process_query(pp, q, db)}
    return _CASE[key]


@pytest.mark.parametrize("cfg", [FAST, FAST56], ids=["fast", "fast56"])
def test_pp_deserialize(sp, oracle_mod, cfg):
    o, cl, pp, q = _session(oracle_mod, cfg, 5, 40)
    p = sp.Params(cfg)
    gpu = sp.PublicParameters.deserialize(p, pp).export()
    assert (gpu == o.pp_deserialize_flat(pp)).all()
    with pytest.raises(sp.SpiralError):
        sp.PublicParameters.deserialize(p, pp[:-8])   # client.rs:213 assert_eq!


@pytest.mark.parametrize("cfg,idx", [(FAST, 77), (FAST56, 300), (C1, 9999)], ids=["fast", "fast56", "c1"])
def test_expand_query(sp, oracle_mod, cfg, idx):
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 50)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    v_reg, v_fold = sp.expand_query(p, gpp, q)
    o_reg, o_fold = o.expand_query(pp, q)
    assert (v_reg == o_reg).all()
    assert (v_fold == o_fold).all()
    assert (sp.get_v_folding_neg(p, v_fold) == o.get_v_folding_neg(o_fold)).all()


def test_coefficient_expansion_and_regev_to_gsw(sp, oracle_mod):
    cfg = FAST56
    o, cl, pp, q = _session(oracle_mod, cfg, 11, 60)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    g, sr = o.g, o.stop_round
    v = np.zeros((1 << g) * 2 * o.ntt_words, dtype=np.uint64)
    v[:2 * o.ntt_words] = o.to_ntt(o.query_deserialize_ct(q))
    mb = o.t_gsw * o.db_dim_2
    v_gpu = sp.coefficient_expansion(p, gpp, v, g, sr, mb)
    v_cpu = o.coefficient_expansion(pp, v, g, sr, mb)
    assert (v_gpu == v_cpu).all()
    w = 2 * o.ntt_words
    v_gsw_inp = np.concatenate([v_cpu[(2 * i + 1) * w:(2 * i + 2) * w] for i in range(mb)])
    flat = o.pp_deserialize_flat(pp)
    v_conv = flat[-2 * 2 * o.t_conv * o.ntt_words:]
    assert (sp.regev_to_gsw(p, gpp, v_gsw_inp, o.db_dim_2) == o.regev_to_gsw(v_gsw_inp, v_conv, o.db_dim_2)).all()


@pytest.mark.parametrize("dim0,num_per", [(64, 4), (512, 32), (16, 64), (300, 128), (64, 256), (512, 1), (700, 2),
                                          (1024, 2), (1024, 64), (1024, 128), (2048, 4)])
def test_multiply_reg_by_database_shapes(sp, oracle_mod, dim0, num_per):
    """db sweep on random (not NTT-of-plaintext) words incl. ragged dim0 and the >255-row fold path; 1024 rows = nu_1 = 10 of
    CFG_16_100000 (util.rs:21-34: (1024, 64) is its plane shape; (1024, 128) the same depth on the PACKED ring kernel), 2048 rows
    beyond every shipped configuration."""
    p, o = _pair(sp, oracle_mod, FAST)
    rng = np.random.default_rng(dim0 * 1000 + num_per)
    N = 2048
    db = rng.integers(0, Q0, N * num_per * dim0, dtype=np.uint64) | (rng.integers(0, Q1, N * num_per * dim0, dtype=np.uint64) << np.uint64(32))
    qv = rng.integers(0, Q0, N * dim0 * 2, dtype=np.uint64) | (rng.integers(0, Q1, N * dim0 * 2, dtype=np.uint64) << np.uint64(32))
    # worst case for the u64 accumulators: all operands maximal
    db[:num_per * dim0] = np.uint64((Q0 - 1) | ((Q1 - 1) << 32))
    qv[:dim0 * 2] = np.uint64((Q0 - 1) | ((Q1 - 1) << 32))
    got = sp.multiply_reg_by_database(p, db, qv, dim0, num_per)
    exp = o.multiply_reg_by_database(db, qv, dim0, num_per)
    assert (got == exp).all()


def test_fold_pack_encode_stages(sp, oracle_mod):
    cfg = FAST56
    idx = 123
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 70)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    item, db = o.generate_random_db_and_get_item(idx)
    v_reg, v_fold = o.expand_query(pp, q)
    v_neg = o.get_v_folding_neg(v_fold)
    N = 2048
    slice_words = o.dim0 * o.num_per * N
    cts = []
    for trial in range(4):
        out_cpu = o.multiply_reg_by_database(db[trial * slice_words:(trial + 1) * slice_words], v_reg)
        out_gpu = sp.multiply_reg_by_database(p, db[trial * slice_words:(trial + 1) * slice_words], v_reg)
        assert (out_gpu == out_cpu).all()
        raw = o.from_ntt(out_cpu)
        f_cpu = o.fold_ciphertexts(raw, v_fold, v_neg)[:2 * N]
        f_gpu = sp.fold_ciphertexts(p, raw, v_fold, v_neg)[:2 * N]
        assert (f_gpu == f_cpu).all()
        cts.append(f_cpu)
    v_ct = np.concatenate(cts)
    flat = o.pp_deserialize_flat(pp)
    v_w = flat[:o.n * (o.n + 1) * o.t_conv * o.ntt_words]
    packed_cpu = o.pack(v_ct, v_w)
    assert (sp.pack(p, gpp, v_ct) == packed_cpu).all()
    praw = o.from_ntt(packed_cpu)
    assert sp.encode(p, praw) == o.encode(praw)


# -------------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("cfg,idx", [(FAST, 0), (FAST, 255), (FAST56, 301), (SMALL_INST2, 123),
                                     (dict(FAST, nu_2=0, db_item_size=8192), 17),
                                     (dict(FAST, nu_2=1), 99)],
                         ids=["fast-0", "fast-255", "fast56", "inst2", "nu2_0", "nu2_1"])
def test_process_query_bytes_and_decode(sp, oracle_mod, cfg, idx):
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 80 + idx)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    resp = sp.process_query(p, gpp, sp.Query.deserialize(p, q), gdb)
    assert resp == o.process_query(pp, q, db)                 # byte-identical to the CPU restatement
    assert cl.decode_response(resp) == o.item_to_vec(item)    # and it decodes (server.rs:1029-1042)
    # a second query against the same registered DB (handles are reusable)
    q2 = cl.generate_query((idx + 1) % o.num_items, 999)
    assert sp.process_query(p, gpp, q2, gdb) == o.process_query(pp, q2, db)


NO_EXPANSION = {"direct_upload": 1, "n": 5, "nu_1": 6, "nu_2": 3, "p": 65536, "q2_bits": 27, "t_gsw": 3, "t_conv": 56,
                "t_exp_left": 56, "t_exp_right": 56}   # get_no_expansion_testing_params, util.rs:139-153


@pytest.mark.parametrize("cfg,idx", [(dict(FAST, version=1), 200), (dict(FAST56, version=1), 300),
                                     (dict(FAST, version=1, instances=2, db_item_size=16384), 5),
                                     (dict(FAST, direct_upload=1), 100), (NO_EXPANSION, 400),
                                     (dict(FAST, direct_upload=1, version=1, nu_2=0), 33)],
                         ids=["v1", "v1-t56", "v1-inst2", "direct", "direct-n5-p65536", "direct-v1-nu2_0"])
def test_process_query_next_rows(sp, oracle_mod, cfg, idx):
    """packing version 1 (lib/server/src/compute/pack.rs:46-99, e2e-tests/params/v1.json) and
    non-expanded 'direct_upload' queries (server.rs:666-679, client.rs:105-128, 315-327)."""
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 17)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    resp = sp.process_query(p, gpp, sp.Query.deserialize(p, q), gdb)
    assert resp == o.process_query(pp, q, db)
    assert cl.decode_response(resp) == o.item_to_vec(item)


def test_process_query_lib_server_default_gadgets(sp, oracle_mod):
    """The lib/server default parameter family (bin/server.rs:191-203; server.rs:1052-1066): odd gadget sizes
    t = (7, 3, 5, 5) -> 9/19/12-bit digits, q2_bits 22, 4 instances x 32 KiB items; nu_1 shortened to 6."""
    cfg = dict(SERVER_DEFAULT, nu_1=6)
    idx = 1500
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 3)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    resp = sp.process_query(p, gpp, q, gdb)
    assert len(resp) == 86016
    assert resp == o.process_query(pp, q, db)
    assert cl.decode_response(resp) == o.item_to_vec(item)


def test_concurrent_host_threads(sp, oracle_mod):
    """Handles are immutable and shareable: 4 host threads issue queries against one registered database at
    once (each call takes its own workspace + stream, SURVEY 8(b) threading row)."""
    import threading
    cfg = FAST56
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(4)
    item, db = o.generate_random_db_and_get_item(9)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    qs = [cl.generate_query((17 * i) % o.num_items, 50 + i) for i in range(8)]
    exp = [o.process_query(pp, q, db) for q in qs]
    got = [None] * 8
    errs = []

    def work(t):
        try:
            for i in range(t, 8, 4):
                for _ in range(3):
                    got[i] = sp.process_query(p, gpp, qs[i], gdb)
        except Exception as e:  # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    assert got == exp


def test_query_list_in_flight_errors_and_threads(sp, oracle_mod):
    """Lists on a narrow database keep up to three queries in flight (sp_process_query_batch): a malformed query in the
    middle of a list fails the call without wedging the workspaces of the queries around it (the next list is answered
    correctly), and two host threads may submit lists against one database at the same time."""
    import threading
    cfg = FAST56
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(4)
    item, db = o.generate_random_db_and_get_item(9)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    qs = [cl.generate_query((29 * i + 1) % o.num_items, 70 + i) for i in range(7)]
    exp = [o.process_query(pp, q, db) for q in qs]
    bad = list(qs)
    bad[3] = qs[3][:-8]  # wrong length: rejected when the list reaches it, three good queries already in flight
    with pytest.raises(sp.SpiralError):
        sp.process_query_batch(p, gpp, bad, gdb)
    assert sp.process_query_batch(p, gpp, qs, gdb) == exp
    got, errs = [None, None], []

    def work(t):
        try:
            for _ in range(3):
                got[t] = sp.process_query_batch(p, gpp, qs[t:] + qs[:t], gdb)
        except Exception as e:  # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    assert got[0] == exp and got[1] == exp[1:] + exp[:1]


def test_grouped_lists_errors_and_threads(sp, oracle_mod):
    """Lists on a PACKED database run as groups with shared expansion launches (r06, run_begin_group): a malformed query in the
    middle of a group fails the call before anything of the group is enqueued and leaves its workspaces usable (the next list
    is answered correctly); two host threads submit lists of 9 (two query tiles, planar copy) and 6 against one database at the
    same time, three times each."""
    import threading
    cfg = {"n": 2, "nu_1": 6, "nu_2": 7, "p": 256, "q2_bits": 20, "t_gsw": 4, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 56,
           "instances": 1, "db_item_size": 256}
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(14)
    item, db = o.generate_random_db_and_get_item(9)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    qs = [cl.generate_query((389 * i + 1) % o.num_items, 170 + i) for i in range(9)]
    exp = [o.process_query(pp, q, db) for q in qs]
    bad = list(qs)
    bad[4] = qs[4][:-8]
    with pytest.raises(sp.SpiralError):
        sp.process_query_batch(p, gpp, bad, gdb)
    sp.paths_taken()
    assert sp.process_query_batch(p, gpp, qs, gdb) == exp
    assert {"expand_group", "sweep_batch_planar"} <= sp.paths_taken()
    lists = [qs, qs[3:]]
    want = [exp, exp[3:]]
    got, errs = [None, None], []

    def work(t):
        try:
            for _ in range(3):
                got[t] = sp.process_query_batch(p, gpp, lists[t], gdb)
        except Exception as e:  # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert got == want


@pytest.mark.parametrize("split", ["0", "1"], ids=["one-stream", "split"])
def test_query_path_without_folding_neg(sp, oracle_mod, monkeypatch, split):
    """r06: the query path no longer computes G - C (get_v_folding_neg, server.rs:505-523): every fold step is the delta form, which
    reads C only (server.cpp run_fold_operands).  Same bytes as the oracle -- whose process_query does compute and use it -- with
    the switch either way, for a single query (split and un-split expansion), a direct-upload query, 56-bit gadget digits and a
    grouped list; the stage export sp_get_v_folding_neg keeps computing it (test_coefficient_expansion_and_regev_to_gsw)."""
    import ctypes as C
    monkeypatch.setenv("SPIRAL_EXPAND_SPLIT", split)
    cases = [({"n": 2, "nu_1": 6, "nu_2": 7, "p": 256, "q2_bits": 20, "t_gsw": 4, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 56,
               "instances": 1, "db_item_size": 256}, 4),
             ({"n": 2, "nu_1": 5, "nu_2": 3, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 8,
               "instances": 2, "db_item_size": 2048, "direct_upload": 1}, 0),
             ({"n": 2, "nu_1": 4, "nu_2": 4, "p": 256, "q2_bits": 20, "t_gsw": 1, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 8,
               "instances": 1, "db_item_size": 256}, 0)]     # (56-bit gadget digits: no fused fold, the delta tail on every level)
    try:
        for cfg, B in cases:
            o = oracle_mod.Params(cfg)
            cl = oracle_mod.Client(o)
            pp = cl.generate_keys(21)
            item, db = o.generate_random_db_and_get_item(3)
            qs = [cl.generate_query((37 * i + 3) % o.num_items, 40 + i) for i in range(max(B, 1))]
            exp = [o.process_query(pp, q, db) for q in qs]
            p = sp.Params(cfg)
            gpp = sp.PublicParameters.deserialize(p, pp)
            gdb = sp.Database(p).load(db)
            for materialise in (0, 1, 0):
                sp.lib().sp_debug_set(b"fold_neg_materialise", C.c_long(materialise))
                assert sp.process_query(p, gpp, qs[0], gdb) == exp[0], (cfg, materialise)
                if B:
                    assert sp.process_query_batch(p, gpp, qs, gdb) == exp, (cfg, materialise)
    finally:
        sp.lib().sp_debug_set(b"fold_neg_materialise", C.c_long(0))


def test_database_reload_beside_other_processes(sp):
    """r06 (profiles/r06_stale_staging.md): with several PROCESSES on one GPU, a device buffer that is freed and allocated again
    between two kernels can be read by one XCD's workgroups as what its address held before -- sp_db_load_plane used to allocate
    its staging buffer per plane, and about one load in 500 then came out with one XCD's share of one plane wrong (found by the
    GPU fuzz: a wrong response from a handle whose database differed from what had been loaded).  The loaders keep one staging
    buffer per handle now.  Eight processes reload a database 1000 times each and check a query after every load (with the old
    loader, SPIRAL_DB_STAGE_KEEP=0, about ten of these 8000 loads go wrong)."""
    import subprocess
    import sys
    worker = os.path.join(ROOT, "tests", "_gpu_reload_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, "1000"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for _ in range(8)]
    outs = [pr.communicate(timeout=900)[0] for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), "\n".join(o[-600:] for o in outs)


def test_process_query_c1(sp, oracle_mod):
    """BASELINE.json configs[0]: 2^14 items x 256 B (nu = (9,5)), full DB, bytes-exact."""
    idx = 12345
    o, cl, pp, q = _session(oracle_mod, C1, idx, 5)
    p = sp.Params(C1)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    resp = sp.process_query(p, gpp, q, gdb)
    assert resp == o.process_query(pp, q, db)
    assert cl.decode_response(resp) == o.item_to_vec(item)


@pytest.mark.parametrize("variant", ["0", "3", "5"])
def test_fused_fold_kernel(sp, oracle_mod, monkeypatch, variant):
    """The fused fold kernels (k_fold_fused / k_fold_fused2: the form used when a level has >= 256 (pair, plane)
    units, i.e. at C2 scale) forced on for a small tree, every kernel variant, through the STAGE export that runs
    them (sp_fold_ciphertexts_fused: G - C formed on the device as in process_query) and through process_query.
    sp_fold_ciphertexts itself honours a caller-supplied v_folding_neg and therefore always takes the literal
    two-matrix tail; both must equal the oracle's fold_ciphertexts.  The path mask proves which kernels ran."""
    monkeypatch.setenv("SPIRAL_FOLD_VARIANT", variant)
    cfg, idx = dict(FAST56, nu_2=4), 777
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 33)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    v_reg, v_fold = o.expand_query(pp, q)
    v_neg = o.get_v_folding_neg(v_fold)
    N = 2048
    sw = o.dim0 * o.num_per * N
    raw = o.from_ntt(o.multiply_reg_by_database(db[:sw], v_reg))
    expect = o.fold_ciphertexts(raw, v_fold, v_neg)[:2 * N]
    sp.paths_taken()
    assert (sp.fold_ciphertexts(p, raw, v_fold, v_neg)[:2 * N] == expect).all()
    taken = sp.paths_taken()
    assert "fold_tail_literal" in taken and "fold_fused" not in taken, taken
    assert (sp.fold_ciphertexts_fused(p, raw, v_fold, fused_min_pairs=1)[:2 * N] == expect).all()
    taken = sp.paths_taken()
    assert "fold_fused" in taken and not ({"fold_tail_delta", "fold_tail_literal", "fold_tail_persistent"} & taken), taken
    assert ("fold_wave" in taken) == (variant == "5"), taken     # the wave-per-transform kernel (wave_ntt.hpp)
    # mixed: the first levels fused, the tail of the tree through the non-fused form
    assert (sp.fold_ciphertexts_fused(p, raw, v_fold, fused_min_pairs=4)[:2 * N] == expect).all()
    taken = sp.paths_taken()
    assert "fold_fused" in taken and ({"fold_tail_delta", "fold_tail_persistent"} & taken), taken
    # ragged tree depth (fewer levels than nu_2) through the stage API
    sub = raw[:4 * 2 * N]
    e2 = o.fold_ciphertexts(sub, v_fold[:2 * 32 * 2 * N], v_neg[:2 * 32 * 2 * N], nu=2)[:2 * N]
    assert (sp.fold_ciphertexts(p, sub, v_fold[:2 * 32 * 2 * N], v_neg[:2 * 32 * 2 * N])[:2 * N] == e2).all()
    assert (sp.fold_ciphertexts_fused(p, sub, v_fold[:2 * 32 * 2 * N], fused_min_pairs=1)[:2 * N] == e2).all()
    monkeypatch.setenv("SPIRAL_FUSED_MIN_PAIRS", "1")
    p2 = sp.Params(cfg)          # workspaces of this handle read the env at creation
    gpp2 = sp.PublicParameters.deserialize(p2, pp)
    gdb = sp.Database(p2).load(db)
    sp.paths_taken()
    resp = sp.process_query(p2, gpp2, q, gdb)
    assert "fold_fused" in sp.paths_taken()
    assert resp == o.process_query(pp, q, db)
    assert cl.decode_response(resp) == o.item_to_vec(item)


@pytest.mark.parametrize("num_per_log,dim0_log,shards", [(7, 3, 1), (8, 4, 2), (5, 4, 1)])
def test_db_device_formats_roundtrip(sp, oracle_mod, num_per_log, dim0_log, shards):
    """8-byte and 7-byte PACKED device formats: what was loaded (reference layout) / synthesised is what
    sp_db_read_ref returns, for every shard."""
    from sdk_amd.spiral import synth_words
    cfg = dict(FAST, nu_1=dim0_log, nu_2=num_per_log, t_gsw=2)
    p = sp.Params(cfg)
    N, dim0, num_per = 2048, 1 << dim0_log, 1 << num_per_log
    rng = np.random.default_rng(num_per_log)
    words = (rng.integers(0, Q0, 4 * N * num_per * dim0, dtype=np.uint64) |
             (rng.integers(0, Q1, 4 * N * num_per * dim0, dtype=np.uint64) << np.uint64(32)))
    ref = words.reshape(4, N, num_per, dim0)
    for s_ in range(shards):
        db = sp.Database(p, s_, shards).load(words)
        nj = dim0 // shards
        for _ in range(40):
            pl, z, ii = int(rng.integers(4)), int(rng.integers(N)), int(rng.integers(num_per))
            got = db.read_ref(pl, z, ii, 0, nj)
            assert (got == ref[pl, z, ii, s_ * nj:(s_ + 1) * nj]).all()
        db.fill_synthetic(99)
        for _ in range(20):
            pl, z, ii = int(rng.integers(4)), int(rng.integers(N)), int(rng.integers(num_per))
            idx = ((pl * N + z) * num_per + ii) * dim0 + s_ * nj + np.arange(nj, dtype=np.uint64)
            assert (db.read_ref(pl, z, ii, 0, nj) == synth_words(99, idx)).all()


@pytest.mark.parametrize("cfg,short", [(dict(FAST, db_item_size=256), 0), (dict(FAST, nu_1=2, nu_2=7, t_gsw=2, db_item_size=1000), 0),
                                       (dict(FAST, nu_1=3, nu_2=1, db_item_size=8192), 0),
                                       (dict(FAST, nu_1=2, nu_2=7, t_gsw=2, db_item_size=512), 3000),
                                       (dict(FAST, nu_1=3, nu_2=0, db_item_size=300, p=16), 100)],
                         ids=["narrow", "packed-ragged-chunk", "full-poly", "packed-short-file", "p16-nu2_0"])
def test_db_preprocessing_on_gpu(sp, oracle_mod, cfg, short):
    """sp_db_load_items == load_db_from_seek (server.rs:320-357): every word of the resident database equals
    the CPU restatement's, incl. ragged chunks, a file shorter than the database, and p != 256."""
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    rng = np.random.default_rng(o.db_item_size)
    blob = rng.integers(0, 256, o.num_items * o.db_item_size - short, dtype=np.uint8).tobytes()
    exp = o.load_db_from_bytes(blob).reshape(4, 2048, o.num_per, o.dim0)
    db = sp.Database(p).load_items(blob)
    for pl in range(4):
        for z in (0, 1, 777, 2047):
            for ii in range(0, o.num_per, max(1, o.num_per // 8)):
                assert (db.read_ref(pl, z, ii, 0, o.dim0) == exp[pl, z, ii]).all(), (pl, z, ii)
    # and sharded
    db2 = sp.Database(p, 1, 2).load_items(blob)
    assert (db2.read_ref(3, 5, o.num_per - 1, 0, o.dim0 // 2) == exp[3, 5, o.num_per - 1, o.dim0 // 2:]).all()


@pytest.mark.parametrize("item_size", [1001, 5, 9], ids=["1001B", "5B-spill-3", "9B-spill-3"])
def test_db_preprocessing_item_spill_across_windows(sp, oracle_mod, monkeypatch, item_size):
    """db_item_size not a multiple of the chunk count: the chunks of an item cover chunks * bytes_per_chunk bytes, i.e.
    they read into the NEXT item(s) (load_item_from_seek, server.rs:300-309).  The upload windows carry that spill as a
    tail so that this also holds for the last item of a window (here: every second row pair starts a new window); with
    5-byte items in 4 chunks the spill (3 bytes) is longer than a chunk (2 bytes) and crosses into the item after next."""
    cfg = dict(FAST, nu_1=4, nu_2=1, db_item_size=item_size)
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    monkeypatch.setenv("SPIRAL_DB_LOAD_WINDOW", str(2 * o.num_per * o.db_item_size))
    rng = np.random.default_rng(5)
    blob = rng.integers(1, 256, o.num_items * o.db_item_size, dtype=np.uint8).tobytes()
    exp = o.load_db_from_bytes(blob).reshape(4, 2048, o.num_per, o.dim0)
    db = sp.Database(p).load_items(blob)
    for pl in range(4):
        for z in (0, 500, 501, 2047):
            for ii in range(o.num_per):
                assert (db.read_ref(pl, z, ii, 0, o.dim0) == exp[pl, z, ii]).all(), (pl, z, ii)


@pytest.mark.parametrize("cfg", [dict(FAST, db_item_size=256), dict(FAST, nu_1=2, nu_2=7, t_gsw=2, db_item_size=600)],
                         ids=["narrow", "packed"])
def test_db_update_item_sparse_bucket(sp, oracle_mod, cfg):
    """an initially empty bucket (absent rows = zero polynomials, as lib/server's SparseDb)
    upserted item by item equals the bulk-preprocessed database; overwriting an item changes only it."""
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    rng = np.random.default_rng(5)
    isz = o.db_item_size
    present = sorted(set(int(x) for x in rng.integers(0, o.num_items, 24)))
    blob = np.zeros(o.num_items * isz, dtype=np.uint8)
    db = sp.Database(p)
    assert not db.read_ref(0, 0, 0, 0, o.dim0).any()
    for i in present:
        rec = rng.integers(0, 256, isz - (i % 3), dtype=np.uint8)   # short records are zero padded
        blob[i * isz:i * isz + rec.size] = rec
        db.update_item(i, rec.tobytes())
    exp = o.load_db_from_bytes(blob.tobytes()).reshape(4, 2048, o.num_per, o.dim0)
    for pl in range(4):
        for z in (0, 9, 2047):
            for ii in range(o.num_per):
                assert (db.read_ref(pl, z, ii, 0, o.dim0) == exp[pl, z, ii]).all()
    i = present[0]
    blob[i * isz:(i + 1) * isz] = 7
    db.update_item(i, bytes([7]) * isz)
    exp = o.load_db_from_bytes(blob.tobytes()).reshape(4, 2048, o.num_per, o.dim0)
    for ii in range(o.num_per):
        assert (db.read_ref(2, 100, ii, 0, o.dim0) == exp[2, 100, ii]).all()
    with pytest.raises(sp.SpiralError):
        db.update_item(o.num_items, b"x")


def test_db_preprocessing_then_query_decodes(sp, oracle_mod):
    cfg = dict(FAST, nu_1=6, nu_2=7, db_item_size=256)
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    with pytest.raises(sp.SpiralError):
        sp.Params(dict(FAST, nu_1=3, nu_2=7))   # the reference would index out of bounds (server.rs:566-571)
    rng = np.random.default_rng(3)
    blob = rng.integers(0, 256, o.num_items * o.db_item_size, dtype=np.uint8).tobytes()
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(8)
    gpp = sp.PublicParameters.deserialize(p, pp)
    db = sp.Database(p).load_items(blob)
    for idx in (0, 513, o.num_items - 1):
        q = cl.generate_query(idx, 9 + idx)
        resp = sp.process_query(p, gpp, q, db)
        if idx == 513:
            assert resp == o.process_query(pp, q, o.load_db_from_bytes(blob))
        got = cl.decode_response(resp)
        item = blob[idx * 256:(idx + 1) * 256]
        assert all(got[t * 64:(t + 1) * 64] == item[t * 64:(t + 1) * 64] for t in range(4))


def test_c2_full_size_decodes_planted_items(sp, oracle_mod):
    """BASELINE.json configs[1] end to end at full size: 2^20 random 256-byte items preprocessed on the GPU
    (56 GiB resident), a real query from the oracle client, the HIP response decoded by the oracle client
    must be the queried item's bytes."""
    import torch
    from conftest import C2
    if torch.cuda.mem_get_info()[0] < 70 * 2**30:
        pytest.skip("needs ~60 GiB of free HBM")
    o = oracle_mod.Params(C2)
    p = sp.Params(C2)
    rng = np.random.default_rng(2024)
    blob = rng.integers(0, 256, o.num_items * 256, dtype=np.uint8)
    db = sp.Database(p).load_items(blob)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(77)
    gpp = sp.PublicParameters.deserialize(p, pp)
    for idx in (0, 123456, o.num_items - 1):
        resp = sp.process_query(p, gpp, cl.generate_query(idx, 1000 + idx), db)
        got = cl.decode_response(resp)
        item = blob[idx * 256:(idx + 1) * 256].tobytes()
        assert all(got[t * 64:(t + 1) * 64] == item[t * 64:(t + 1) * 64] for t in range(4)), idx


def test_c2_full_size_sampled_parity(sp, oracle_mod):
    """BASELINE.json configs[1] at full size (2^20 x 256 B, 64 GiB encoded, 56 GiB packed in HBM):
    the synthetic DB is a pure function of the reference-layout index, so any first-dimension output
    can be recomputed on the CPU from sp_synth_word and the query slice.  64 sampled (plane, z, ii)
    outputs x 4 residues against exact integer arithmetic."""
    import torch
    from sdk_amd.sharding import partial_tensor
    from sdk_amd.spiral import synth_words
    from conftest import C2
    import bench
    p = sp.Params(C2)
    free, total = torch.cuda.mem_get_info()
    if free < 70 * 2**30:
        pytest.skip("needs ~60 GiB of free HBM")
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    q = bench.synthetic_wire_bytes(p.query_bytes(), 2)
    db = sp.Database(p).fill_synthetic(0x123456789)
    v_reg, _ = sp.expand_query(p, pp, q)
    run = sp.QueryRun(p, pp, q).sweep(db)
    run.sync()
    part = partial_tensor(run)
    N, dim0, num_per = 2048, 512, 2048
    vr = v_reg.reshape(N, dim0, 2)
    rng = np.random.default_rng(11)
    for _ in range(64):
        pl, z, ii = int(rng.integers(4)), int(rng.integers(N)), int(rng.integers(num_per))
        idx = ((pl * N + z) * num_per + ii) * dim0 + np.arange(dim0, dtype=np.uint64)
        w = synth_words(0x123456789, idx)
        blo, bhi = [int(x) for x in (w & np.uint64(0xFFFFFFFF))], [int(x) for x in (w >> np.uint64(32))]
        for r in range(2):
            alo = [int(x) & 0xFFFFFFFF for x in vr[z, :, r]]
            ahi = [int(x) >> 32 for x in vr[z, :, r]]
            e0 = sum(a * b for a, b in zip(alo, blo)) % Q0
            e1 = sum(a * b for a, b in zip(ahi, bhi)) % Q1
            g0 = int(part[(((pl * 2 + r) * 2 + 0) * N + z) * num_per + ii])
            g1 = int(part[(((pl * 2 + r) * 2 + 1) * N + z) * num_per + ii])
            assert (g0, g1) == (e0, e1), (pl, z, ii, r)
    # the whole path still runs to a response of the right size at this scale
    assert len(run.finish()) == p.get("response_bytes")


@pytest.mark.parametrize("fused_min", ["256", "1"], ids=["default-threshold", "fused-all-levels"])
@pytest.mark.parametrize("per_plane", [False, True], ids=["one-launch", "per-plane"])
@pytest.mark.parametrize("cfg,G", [(dict(FAST56, nu_2=4), 2), (dict(FAST56, nu_2=4), 8), (dict(FAST, nu_1=6, nu_2=7, db_item_size=256), 4),
                                   (dict(FAST56, nu_2=5, t_gsw=3, t_conv=3, t_exp_left=5), 4)],
                         ids=["narrow-G2", "narrow-G8", "packed-G4", "odd-gadgets-G4"])
def test_distributed_fold_single_gpu_emulation(sp, oracle_mod, monkeypatch, cfg, G, per_plane, fused_min):
    """The N > 1 bench path (sweep_scatter -> reduce-scatter -> fold_local -> gather -> finish_gathered) with
    the G ranks played one after another on one GPU; the collective is replaced by a torch sum/slice.
    per-plane: the sweep one plane per launch (sp_query_sweep_scatter_plane), reduce-scattered plane by plane."""
    import torch
    from sdk_amd.sharding import local_cts_tensor, partial_tensor, scatter_layout_index, scatter_plane_layout_index
    monkeypatch.setenv("SPIRAL_FUSED_MIN_PAIRS", fused_min)   # the fused fold kernels in fold_local / finish_gathered too
    idx = 77
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 91)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    expect = o.process_query(pp, q, db)
    shards = [sp.Database(p, s, G).load(db) for s in range(G)]
    planes = 4
    if per_plane:
        # begun for their shard: only the shard's first-dimension rows of the query are expanded
        runs = [sp.QueryRun(p, gpp, q, db=shards[s]) for s in range(G)]
        with pytest.raises(sp.SpiralError):
            runs[0].sweep_scatter_plane(shards[0], G, 1)       # planes go in order
        with pytest.raises(sp.SpiralError):
            runs[0].sweep_scatter_plane(shards[1], G, 0)       # expanded for shard 0's rows only
        for s in range(G):
            for pl in range(planes):
                runs[s].sweep_scatter_plane(shards[s], G, pl)
    else:
        runs = [sp.QueryRun(p, gpp, q).sweep_scatter(shards[s], G) for s in range(G)]
    total = None
    for r in runs:
        r.sync()
        t = partial_tensor(r)
        total = t.clone() if total is None else total + t
    chunk = total.numel() // G
    if per_plane:
        # the two layouts hold the same values: [plane][g][...] vs [g][plane][...]
        ref_runs = [sp.QueryRun(p, gpp, q).sweep_scatter(shards[s], G) for s in range(G)]
        ref_total = None
        for r in ref_runs:
            r.sync()
            t = partial_tensor(r)
            ref_total = t.clone() if ref_total is None else ref_total + t
        rng = np.random.default_rng(3)
        for _ in range(200):
            pl, rr, c, z, ii = (int(rng.integers(planes)), int(rng.integers(2)), int(rng.integers(2)),
                                int(rng.integers(2048)), int(rng.integers(o.num_per)))
            assert int(total[scatter_plane_layout_index(o.num_per, G, pl, rr, c, z, ii)]) == \
                int(ref_total[scatter_layout_index(o.num_per, planes, G, pl, rr, c, z, ii)])
        pw, pc = total.numel() // planes, total.numel() // planes // G
    locals_ = []
    for g, r in enumerate(runs):
        if per_plane:
            mine = torch.cat([total[pl * pw + g * pc:pl * pw + (g + 1) * pc] for pl in range(planes)]).contiguous()
        else:
            mine = total[g * chunk:(g + 1) * chunk].contiguous()
        torch.cuda.synchronize()
        if per_plane:   # one plane at a time on the second stream, then the join (what the overlapped flow issues)
            with pytest.raises(sp.SpiralError):
                r.fold_local_plane(mine.data_ptr(), G, 1)    # planes go in order
            for pl in range(planes):
                r.fold_local_plane(mine[pl * pc:].data_ptr(), G, pl)
            r.fold_local_join()
        else:
            r.fold_local(mine.data_ptr(), G)
        r.sync()
        locals_.append(local_cts_tensor(r).clone())
    gathered = torch.cat(locals_).contiguous()
    torch.cuda.synchronize()
    resp = runs[0].finish_gathered(gathered.data_ptr(), G)
    assert resp == expect
    if cfg["t_gsw"] >= 8:   # the narrow test gadgets are too noisy to decode; byte parity is the claim there
        assert cl.decode_response(resp) == o.item_to_vec(item)


@pytest.mark.parametrize("cfg,G,loader", [(dict(FAST56, nu_2=4), 4, "load"), (dict(FAST, nu_1=6, nu_2=8, db_item_size=256), 2, "items"),
                                          (dict(FAST, nu_1=6, nu_2=8, db_item_size=256), 8, "load")],
                         ids=["narrow-G4", "packed-G2-items", "narrow-from-packed-G8"])
def test_column_sharded_single_gpu_emulation(sp, oracle_mod, monkeypatch, cfg, G, loader):
    """column shards (ii = g mod G), complete per-shard outputs, local fold, gather, final levels."""
    import torch
    from sdk_amd.sharding import local_cts_tensor
    if loader == "items":
        monkeypatch.setenv("SPIRAL_FUSED_MIN_PAIRS", "1")
    idx = 1234 % (1 << (cfg["nu_1"] + cfg["nu_2"]))
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 55)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    if loader == "items":
        rng = np.random.default_rng(1)
        blob = rng.integers(0, 256, o.num_items * o.db_item_size, dtype=np.uint8).tobytes()
        db = o.load_db_from_bytes(blob)
        shards = [sp.Database(p, g, G, by_columns=True).load_items(blob) for g in range(G)]
    else:
        item, db = o.generate_random_db_and_get_item(idx)
        shards = [sp.Database(p, g, G, by_columns=True).load(db) for g in range(G)]
    expect = o.process_query(pp, q, db)
    ref = db.reshape(4, 2048, o.num_per, o.dim0)
    assert (shards[1].read_ref(2, 7, 1 + G, 0, o.dim0) == ref[2, 7, 1 + G]).all()
    with pytest.raises(sp.SpiralError):
        shards[1].read_ref(2, 7, 0, 0, o.dim0)       # column 0 lives on shard 0
    runs = [sp.QueryRun(p, gpp, q).sweep(shards[g]) for g in range(G)]
    locals_ = []
    for r in runs:
        r.fold_local(r.partial_ptr(), G)
        r.sync()
        locals_.append(local_cts_tensor(r).clone())
    gathered = torch.cat(locals_).contiguous()
    torch.cuda.synchronize()
    assert runs[0].finish_gathered(gathered.data_ptr(), G) == expect


@pytest.mark.parametrize("cfg,B,in_flight", [(dict(FAST, nu_1=6, nu_2=7, db_item_size=256), 11, "2"), (FAST56, 3, "2"),
                                             (FAST56, 6, "2"), (FAST56, 4, "1")],
                         ids=["packed-11", "narrow-3", "narrow-6", "narrow-4-one-at-a-time"])
def test_process_query_batch(sp, oracle_mod, monkeypatch, cfg, B, in_flight):
    """BASELINE configs[4]: batched queries share database passes (groups of <= 8 on the PACKED layout; 11 = 8 + 3
    exercises two group sizes); on narrow (8-byte) databases the list runs one pass per query with two queries in flight
    (SPIRAL_BATCH_IN_FLIGHT=1: one at a time).  Two different clients' keys in one batch.  Byte-identical per query."""
    monkeypatch.setenv("SPIRAL_BATCH_IN_FLIGHT", in_flight)
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    cls = [oracle_mod.Client(o), oracle_mod.Client(o)]
    pps = [cls[0].generate_keys(1), cls[1].generate_keys(2)]
    gpps = [sp.PublicParameters.deserialize(p, x) for x in pps]
    item, db = o.generate_random_db_and_get_item(3)
    gdb = sp.Database(p).load(db)
    idxs = [(37 * i + 5) % o.num_items for i in range(B)]
    qs = [cls[i % 2].generate_query(idxs[i], 100 + i) for i in range(B)]
    resp = sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb)
    for i in range(B):
        assert resp[i] == o.process_query(pps[i % 2], qs[i], db), i
    assert cls[1].decode_response(resp[1]) == o.item_to_vec(o.generate_random_db_and_get_item(idxs[1])[0])


def test_bad_lengths_raise(sp, oracle_mod):
    o, cl, pp, q = _session(oracle_mod, FAST, 1, 7)
    p = sp.Params(FAST)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p)
    with pytest.raises(sp.SpiralError):
        sp.process_query(p, gpp, q[:-1], gdb)          # client.rs:304 assert_eq!
    with pytest.raises(sp.SpiralError):
        gdb.load(np.zeros(10, dtype=np.uint64))


def test_row_sharded_partials_sum_to_full(sp, oracle_mod):
    """Multi-GPU split on one GPU: G shards' partial buffers summed element-wise == unsharded result."""
    import torch
    from sdk_amd.sharding import partial_tensor
    cfg, idx = FAST56, 200
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 91)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    expect = o.process_query(pp, q, db)
    for G in (2, 4, 8):
        shards = [sp.Database(p, s, G).load(db) for s in range(G)]
        runs = [sp.QueryRun(p, gpp, q, db=shards[s] if G != 4 else None).sweep(shards[s]) for s in range(G)]
        total = None
        for r in runs:
            r.sync()
            t = partial_tensor(r)
            total = t.clone() if total is None else total + t
        assert int(total.max()) < 2**31 - 1
        partial_tensor(runs[0]).copy_(total)
        torch.cuda.synchronize()
        assert runs[0].finish() == expect
        assert cl.decode_response(expect) == o.item_to_vec(item)


def test_rccl_world1_paths():
    """The N > 1 bench paths through a real nccl (RCCL) process group (world size 1; a 1-GPU box cannot host
    two ranks): collectives on the library's own device buffers, byte-identical responses."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_world1.py")], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("env,extra,mode", [({"SPIRAL_FORCE_DIST": "1"}, [], "lib"),
                                            ({"SPIRAL_FORCE_DIST": "1", "SPIRAL_MULTIGPU": "torch"}, [], "torch"),
                                            ({"SPIRAL_FORCE_DIST": "1"}, ["--mode", "replicas"], "replicas"),
                                            ({}, ["--mode", "replicas", "--batch", "3"], "replicas"),
                                            ({}, [], "single")],
                         ids=["shard-lib-rccl", "shard-torch", "replicas-dist", "replicas-1gpu", "single"])
def test_bench_modes_first_run_safe(env, extra, mode):
    """bench.py's N > 1 code paths (process group, library-side RCCL communicator, self-check, per-rank gathers,
    replicas mode) executed end to end at world size 1 on a small configuration: the JSON line must carry the
    fields the driver and the judge read, and the multi-GPU self-check must have passed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, MASTER_PORT="29611", **env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "p2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=280, env=e)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["mode"] == mode and j["value"] > 0 and j["n_gpus"] == 1 and j["rccl_ranks"] == 1
    assert j["unit"] == "queries/s" and j["steps"] == 3 and j["warmup"] == 1
    assert 0 < j["roofline"]["frac"] < 1 and j["roofline"]["bound"] == "hbm"
    if mode in ("lib", "torch"):
        assert j["overlap_selfcheck"] == "ok", j["overlap_selfcheck"]
        assert j["per_rank"] and len(j["per_rank"]["step_ms"]) == 1
        assert j["scaling"] == "strong"
    if mode == "replicas":
        assert j["scaling"] == "weak" and j["config"]["queries_per_step"] == (3 if "--batch" in extra else 8)


@pytest.mark.parametrize("cfg,G", [(dict(FAST56, nu_2=4), 2), (dict(FAST56, nu_2=4), 8), (dict(FAST, nu_1=6, nu_2=7, db_item_size=256), 4),
                                   (dict(FAST56, nu_2=5, t_gsw=3, t_conv=3, t_exp_left=5), 4),
                                   (dict(FAST, nu_1=6, nu_2=10, t_gsw=2, db_item_size=256), 8)],
                         ids=["narrow-G2", "narrow-G8", "packed-G4", "odd-gadgets-G4", "wide-G8"])
def test_process_query_sharded_c_abi_loopback(sp, oracle_mod, cfg, G):
    """sp_process_query_sharded (sdk_amd/csrc/comm.cpp) -- the whole N > 1 answer path inside the library: G ranks
    as G host threads on this one GPU, each with its own row shard, workspace and streams; the two collectives are
    the in-process loopback transport (sp_comm_create_custom) with ncclReduceScatter / ncclAllGather semantics.
    Rank 0's response must equal the oracle's process_query on the unsharded database; twice, to cover buffer reuse."""
    from sdk_amd.sharding import LoopbackWorld
    idx = 77
    o, cl, pp, q = _session(oracle_mod, cfg, idx, 91)
    q2 = cl.generate_query(idx + 1, 17)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    expect = [o.process_query(pp, q, db), o.process_query(pp, q2, db)]
    shards = [sp.Database(p, s, G).load(db) for s in range(G)]
    world = LoopbackWorld(G)

    def rank_main(r):
        sp.lib().sp_set_device(0)
        sp.paths_taken()
        out = [world.comm(r).process_query(p, gpp, qq, shards[r]) for qq in (q, q2)]
        return out, sp.paths_taken()
    res = world.run(rank_main)
    assert res[0][0] == expect
    for r in range(1, G):
        assert res[r][0] == [b"", b""]
    for r in range(G):
        assert {"scatter_out", "custom_transport"} <= res[r][1] and "rccl_in_library" not in res[r][1], res[r][1]
        if G > 1 and o.num_per >= 2:
            assert "expand_pruned" in res[r][1]
    if cfg.get("t_gsw") == 8:       # the reduced gadget widths of the other configs are not decodable (nor need be)
        assert cl.decode_response(expect[0]) == o.item_to_vec(item)
    with pytest.raises(sp.SpiralError):
        world.comm(0).process_query(p, gpp, q[:-8], shards[0])      # bad query length: no collective is entered


@pytest.mark.parametrize("G", [2, 4])
def test_process_queries_sharded_pipelined_list(sp, oracle_mod, G):
    """sp_process_queries_sharded: a list of queries through the sharded flow, query k + 1 expanding while query k's planes
    are swept and exchanged (loopback transport, G ranks on this GPU).  Rank 0's responses must equal the oracle's and the
    one-at-a-time entry point's; the exchange buffers come from sp_comm_reserve."""
    from sdk_amd.sharding import LoopbackWorld
    cfg = dict(FAST, nu_1=5, nu_2=8, t_gsw=4, db_item_size=1024)
    o, cl, pp, q0 = _session(oracle_mod, cfg, 5, 91)
    qs = [q0] + [cl.generate_query((311 * k + 9) % o.num_items, 40 + k) for k in range(1, 5)]
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(5)
    gpp = sp.PublicParameters.deserialize(p, pp)
    expect = [o.process_query(pp, q, db) for q in qs]
    shards = [sp.Database(p, s, G).load(db) for s in range(G)]
    world = LoopbackWorld(G)

    def rank_main(r):
        sp.lib().sp_set_device(0)
        world.comm(r).reserve(p)
        sp.paths_taken()
        out = world.comm(r).process_queries(p, gpp, qs, shards[r])
        single = world.comm(r).process_query(p, gpp, qs[2], shards[r])
        return out, single, sp.paths_taken()
    res = world.run(rank_main)
    assert res[0][0] == expect and res[0][1] == expect[2]
    for r in range(1, G):
        assert res[r][0] == [] and res[r][1] == b""
    for r in range(G):
        assert {"scatter_out", "custom_transport", "expand_pruned"} <= res[r][2], res[r][2]
    assert world.comm(0).process_queries(p, gpp, [], shards[0]) == []


def test_overlapped_fold_direct_upload_parity(sp, oracle_mod):
    """Non-expanded ('direct_upload') query on a wide packed database: the overlapped per-plane path with the
    fold matrices coming straight from the wire (server.rs:666-679) instead of regev_to_gsw."""
    cfg = {"n": 2, "nu_1": 4, "nu_2": 10, "p": 256, "q2_bits": 20, "t_gsw": 2, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 8192, "direct_upload": 1}
    o, cl, pp, q = _session(oracle_mod, cfg, 1234, 6)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(1234)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    assert sp.process_query(p, gpp, q, gdb) == o.process_query(pp, q, db)


def test_overlapped_fold_many_planes_parity(sp, oracle_mod):
    """The default path of wide packed databases (one sweep launch per plane, from_ntt + fold of plane p on the
    second stream under the sweep of plane p+1) with 8 planes (instances = 2), byte-identical to the oracle.
    t_gsw = 2 keeps the oracle fast (the response is not decodable at this noise level and need not be) and makes
    the gadget digits 28 bits wide, i.e. possibly >= q: the fused fold has to canonicalise them."""
    cfg = {"n": 2, "nu_1": 4, "nu_2": 10, "p": 256, "q2_bits": 20, "t_gsw": 2, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 2, "db_item_size": 16384}
    o, cl, pp, q = _session(oracle_mod, cfg, 1234, 6)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(1234)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    run = sp.QueryRun(p, gpp, q)
    assert run.sweep_launches(gdb) == (1 if os.environ.get("SPIRAL_PIPELINE") == "0" else 8)   # t_gsw = 2: no class split
    resp = run.sweep(gdb).finish()
    run.free()
    assert resp == o.process_query(pp, q, db)
    assert sp.process_query(p, gpp, q, gdb) == resp


@pytest.mark.parametrize("nu_1,nu_2,t_gsw,ring,wgs,defer", [
    (5, 10, 4, "8", "1", "256"),   # the defaults: two buffers of 8 row pairs, one workgroup per CU, tails batched
    (5, 10, 4, "4", "2", "256"),
    (5, 10, 4, "2", "1", "64"),    # two more levels per plane before parking
    (5, 10, 4, "0", "1", "256"),   # plain persistent sweep, batched tails
    (5, 10, 4, "8", "1", "0"),     # ring sweep, every plane folded to the end under the next sweep
    (5, 11, 4, "8", "1", "256"),
    (4, 10, 2, "8", "1", "256"),   # 8 row pairs per stream: the ring falls back to buffers of 4
    (3, 10, 3, "8", "1", "256"),   # 4 row pairs per stream: buffers of 2
])
def test_ring_sweep_and_batched_tails_parity(sp, oracle_mod, monkeypatch, nu_1, nu_2, t_gsw, ring, wgs, defer):
    """The pipelined query's persistent sweep in ring form (k_sweep_packed_ring: two buffers of row pairs per wave, the next
    stream's first buffer requested before this one's sums are stored) and the deferred fold tails (every plane folded to
    256 / 64 ciphertexts under the next sweep, the remaining levels of all planes as one batch): response bytes equal the
    oracle's for every buffer size and with either piece switched off."""
    monkeypatch.setenv("SPIRAL_PIPE_RING", ring)
    monkeypatch.setenv("SPIRAL_PIPE_RING_WGS", wgs)
    monkeypatch.setenv("SPIRAL_PIPE_TAIL_DEFER", defer)
    cfg = {"n": 2, "nu_1": nu_1, "nu_2": nu_2, "p": 256, "q2_bits": 20, "t_gsw": t_gsw, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 8192}
    c = _oracle_case(oracle_mod, cfg, 97, 11, 12)
    o, cl, pp, q, item, db = c["o"], c["cl"], c["pp"], c["q"], c["item"], c["db"]
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    sp.paths_taken()
    resp = sp.process_query(p, gpp, q, gdb)
    taken = sp.paths_taken()
    assert "pipelined_fold_overlap" in taken, taken
    assert ("sweep_ring" in taken) == (ring != "0"), taken
    # (t_gsw = 2: 29-bit gadget digits, which the fused fold kernels do not take -- nothing is deferred there)
    assert ("fold_tail_batched" in taken) == (defer != "0" and t_gsw > 2), taken
    assert resp == c["resp"]
    if t_gsw >= 4:  # (fewer gadget digits: the noise is too large to decode, the bytes still have to agree)
        assert cl.decode_response(resp) == o.item_to_vec(item)
    if (nu_1, nu_2, ring, defer) == (5, 10, "8", "256"):
        # two and three queries in flight on separate workspaces (begin + sweep of the next before the previous one is waited
        # for; their sweeps may overlap on the device): each answer is the single query's
        qs = [q, cl.generate_query(5, 77), cl.generate_query(1234 % o.num_items, 78)]
        single = [resp] + [sp.process_query(p, gpp, x, gdb) for x in qs[1:]]
        for depth in (2, 3):
            runs, got = [], []
            for x in qs + qs:
                r = sp.QueryRun(p, gpp, x, db=gdb)
                r.sweep(gdb)
                runs.append(r)
                if len(runs) == depth:
                    got.append(runs[0].finish())
                    runs.pop(0).free()
            while runs:
                got.append(runs[0].finish())
                runs.pop(0).free()
            assert got == single + single, depth


def test_process_query_batch_lds_staged(sp, oracle_mod):
    """Batched sweep with the queries' rows staged in LDS (groups of >= 5 queries on databases with >= 512 columns):
    14 queries = one group of 8 (two row pairs in flight) + one of 6 (four in flight); byte-identical per query."""
    cfg = {"n": 2, "nu_1": 4, "nu_2": 9, "p": 256, "q2_bits": 20, "t_gsw": 2, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 8192}
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(21)
    gpp = sp.PublicParameters.deserialize(p, pp)
    item, db = o.generate_random_db_and_get_item(9)
    gdb = sp.Database(p).load(db)
    B = 14
    qs = [cl.generate_query((523 * i + 9) % o.num_items, 300 + i) for i in range(B)]
    resp = sp.process_query_batch(p, [gpp] * B, qs, gdb)
    for i in (0, 3, 7, 8, 13):   # both groups, first and last slot of each
        assert resp[i] == o.process_query(pp, qs[i], db), i
    single = sp.process_query(p, gpp, qs[5], gdb)
    assert resp[5] == single
    for i in range(B):
        assert resp[i] == sp.process_query(p, gpp, qs[i], gdb), i


@pytest.mark.parametrize("nu_1,nu_2,B,check", [(6, 7, 8, "all"), (5, 8, 5, "all"), (7, 7, 4, "all"), (8, 8, 7, "some"), (9, 7, 8, "some")],
                         ids=["64x128-B8", "32x256-B5", "128x128-B4", "256x256-B7", "512x128-B8"])
def test_process_query_batch_matrix_core_sweep(sp, oracle_mod, nu_1, nu_2, B, check):
    """Batched pass on the matrix cores (k_sweep_mfma_batch: signed base-256 digits on v_mfma_i32_16x16x64_i8, groups of
    >= 4 queries, first dimension a multiple of 32 rows): byte-identical to the oracle, to the VALU batch kernel
    (SPIRAL_BATCH_MFMA=0) and to one-at-a-time queries; the path bit proves which kernel ran.  Two clients' keys."""
    import ctypes as C
    cfg = {"n": 2, "nu_1": nu_1, "nu_2": nu_2, "p": 256, "q2_bits": 20, "t_gsw": 4, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 256 << max(0, 16 - nu_1 - nu_2)}
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    cls = [oracle_mod.Client(o), oracle_mod.Client(o)]
    pps = [cls[0].generate_keys(41), cls[1].generate_keys(42)]
    gpps = [sp.PublicParameters.deserialize(p, x) for x in pps]
    if check == "all":
        item, db = o.generate_random_db_and_get_item(3)
    else:   # the larger shapes: uniformly random words (byte parity does not need a decodable database)
        rng = np.random.default_rng(nu_1 * 16 + nu_2)
        n_words = 4 * 2048 * o.num_per * o.dim0
        db = rng.integers(0, Q0, n_words, dtype=np.uint64) | (rng.integers(0, Q1, n_words, dtype=np.uint64) << np.uint64(32))
    gdb = sp.Database(p).load(db)
    idxs = [(1009 * i + 3) % o.num_items for i in range(B)]
    qs = [cls[i % 2].generate_query(idxs[i], 500 + i) for i in range(B)]
    sp.paths_taken()
    resp = sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb)
    taken = sp.paths_taken()
    assert "sweep_batch_mfma" in taken
    # r06: the group's expansions share every round's launches (grid dimension = query, two clients' public parameters behind
    # per-query offsets); one query at a time on sixteen streams -- the r02-r05 form -- must give the same bytes, and so must
    # both extremes of the shared rounds' dispatch (every round as three launches / every round as k_expand_round)
    assert "expand_group" in taken, taken
    # ... and the right-hand side of the large rounds (56 one-bit digits per ciphertext) on the wave-per-transform engine
    # (k_expand_wave: the expansion key in wave layout, the second row's transform as a 57th "digit" times the constants 0 | 1)
    assert "expand_wave" in taken, taken
    for switch, value in ((b"expand_group", 0), (b"expand_group_round_min", 1 << 40), (b"expand_group_round_min", 1), (b"expand_wave_min_digits", 0),
                          (b"expand_wave_min_digits", 1)):
        sp.lib().sp_debug_set(switch, C.c_long(value))
        try:
            sp.paths_taken()
            assert sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb) == resp, (switch, value)
            taken = sp.paths_taken()
            assert ("expand_group" in taken) == (switch != b"expand_group"), (switch, taken)
            if switch == b"expand_group_round_min":
                assert ("expand_round_one_launch" in taken) == (value == 1), (value, taken)
            if switch == b"expand_wave_min_digits":      # 0: never; 1: the 8-digit left-hand side too
                assert ("expand_wave" in taken) == (value == 1), (value, taken)
        finally:
            sp.lib().sp_debug_set(switch, C.c_long({b"expand_group": 1, b"expand_group_round_min": 4096, b"expand_wave_min_digits": 16}[switch]))
    sp.lib().sp_debug_set(b"batch_mfma", C.c_long(0))
    try:
        sp.paths_taken()
        valu = sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb)
        assert "sweep_batch_mfma" not in sp.paths_taken()
    finally:
        sp.lib().sp_debug_set(b"batch_mfma", C.c_long(1))
    assert resp == valu
    for i in (range(B) if check == "all" else (0, B - 1)):
        assert resp[i] == o.process_query(pps[i % 2], qs[i], db), i
    for i in range(B):
        assert resp[i] == sp.process_query(p, gpps[i % 2], qs[i], gdb), i
    if check == "all":
        assert cls[1].decode_response(resp[1]) == o.item_to_vec(o.generate_random_db_and_get_item(idxs[1])[0])


@pytest.mark.parametrize("nu_1,nu_2,B,check", [(6, 7, 16, "all"), (5, 8, 12, "all"), (7, 7, 9, "all"), (8, 8, 16, "some"), (9, 7, 16, "some"),
                                               (5, 7, 19, "all")],
                         ids=["64x128-B16", "32x256-B12", "128x128-B9", "256x256-B16", "512x128-B16", "32x128-B19"])
def test_process_query_batch_two_query_tiles(sp, oracle_mod, nu_1, nu_2, B, check):
    """Nine to sixteen queries share ONE database pass (k_sweep_mfma_batch with two query tiles, r04: the database words are
    loaded and split into digits once for both tiles; ring depths 2, 4 and 8 by row count): byte-identical to the oracle, to
    groups of at most 8 (SPIRAL_BATCH_GROUP=8, the one-tile pass) and to one-at-a-time queries; unused query columns of the
    second tile (B = 9, 12) stay out of the way; a list of 19 is cut into a group of 16 and a group of 3 (vector kernel).
    Two clients' keys."""
    import ctypes as C
    cfg = {"n": 2, "nu_1": nu_1, "nu_2": nu_2, "p": 256, "q2_bits": 20, "t_gsw": 4, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 256 << max(0, 16 - nu_1 - nu_2)}
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    cls = [oracle_mod.Client(o), oracle_mod.Client(o)]
    pps = [cls[0].generate_keys(43), cls[1].generate_keys(44)]
    gpps = [sp.PublicParameters.deserialize(p, x) for x in pps]
    if check == "all":
        item, db = o.generate_random_db_and_get_item(3)
    else:
        rng = np.random.default_rng(nu_1 * 16 + nu_2 + 1)
        n_words = 4 * 2048 * o.num_per * o.dim0
        db = rng.integers(0, Q0, n_words, dtype=np.uint64) | (rng.integers(0, Q1, n_words, dtype=np.uint64) << np.uint64(32))
    gdb = sp.Database(p).load(db)
    idxs = [(1013 * i + 5) % o.num_items for i in range(B)]
    qs = [cls[i % 2].generate_query(idxs[i], 700 + i) for i in range(B)]
    sp.paths_taken()
    resp = sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb)
    taken = sp.paths_taken()
    assert "sweep_batch_mfma_two_tiles" in taken
    # r05: where the first dimension is whole 64-row blocks the two-tile pass reads the DIGIT-PLANAR copy of the database
    # (k_sweep_planar, built from the PACKED words on this first call); the same call on the PACKED words must agree
    planar = (1 << nu_1) % 64 == 0
    assert ("sweep_batch_planar" in taken) == planar, taken
    if planar:
        sp.lib().sp_debug_set(b"batch_planar", C.c_long(0))
        try:
            gdb2 = sp.Database(p).load(db)          # (a database that has never been given a planar copy)
            sp.paths_taken()
            packed = sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb2)
            assert "sweep_batch_planar" not in sp.paths_taken()
        finally:
            sp.lib().sp_debug_set(b"batch_planar", C.c_long(1))
        assert resp == packed
        # a writer drops the planar copy; the next group rebuilds it from the new words
        gdb.load(db)
        sp.paths_taken()
        assert sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb) == resp
        assert "sweep_batch_planar" in sp.paths_taken()
    sp.lib().sp_debug_set(b"batch_group", C.c_long(8))
    try:
        sp.paths_taken()
        one_tile = sp.process_query_batch(p, [gpps[i % 2] for i in range(B)], qs, gdb)
        assert "sweep_batch_mfma_two_tiles" not in sp.paths_taken()
    finally:
        sp.lib().sp_debug_set(b"batch_group", C.c_long(0))
    assert resp == one_tile
    # against the oracle: both ends of both tiles (+ the second group's last query); everything against one-at-a-time queries,
    # which the other tests tie to the oracle
    for i in sorted({0, 7, 8, min(B, 16) - 1, B - 1}):
        assert resp[i] == o.process_query(pps[i % 2], qs[i], db), i
    for i in range(B):
        assert resp[i] == sp.process_query(p, gpps[i % 2], qs[i], gdb), i
    if check == "all":
        assert cls[1].decode_response(resp[9 if B > 9 else 1]) == o.item_to_vec(o.generate_random_db_and_get_item(idxs[9 if B > 9 else 1])[0])


@pytest.mark.parametrize("nu_1,nu_2", [(6, 7), (7, 8)], ids=["64x128", "128x256"])
def test_planar_copy_lifecycle(sp, oracle_mod, nu_1, nu_2):
    """The digit-planar copy of a PACKED database (what groups of 9-16 queries read) over the life of a bucket:
    built at load time by sp_db_prepare_batch; sp_db_update_item PATCHES it -- the 8 sixteen-byte entries per (plane, z) that hold
    the item -- instead of invalidating 8 bytes per database word, so the batch after a series of upserts (first / last row and
    column, both columns of a lane slot, both rows of a row pair, an overwrite) reads the copy and returns the oracle's bytes on
    the edited file; a bulk writer drops it and gives the memory back; so does the switch."""
    import ctypes as C
    cfg = {"n": 2, "nu_1": nu_1, "nu_2": nu_2, "p": 256, "q2_bits": 20, "t_gsw": 4, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 256}
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(61)
    gpp = sp.PublicParameters.deserialize(p, pp)
    rng = np.random.default_rng(nu_1 + nu_2)
    isz = o.db_item_size
    blob = rng.integers(0, 256, o.num_items * isz, dtype=np.uint8)
    gdb = sp.Database(p).load_items(blob)
    copy = 4 * 2048 * o.num_per * o.dim0 * 8                                      # 8 bytes per database word
    assert gdb.batch_copy_bytes() == 0
    assert gdb.prepare_batch() is True
    assert gdb.batch_copy_bytes() == copy
    npr, d0 = o.num_per, o.dim0
    edits = [0, 1, npr, npr + 1, (d0 - 1) * npr + npr - 1, (d0 // 2) * npr + 77, 3 * npr + 126, 3 * npr + 127, 64 * npr % o.num_items + 5, 1]
    for k, it in enumerate(edits):
        rec = rng.integers(0, 256, isz - (k % 4), dtype=np.uint8)
        blob[it * isz:(it + 1) * isz] = 0
        blob[it * isz:it * isz + rec.size] = rec
        gdb.update_item(it, rec.tobytes())
    assert gdb.batch_copy_bytes() == copy                                        # the copy survived the upserts
    exp = o.load_db_from_bytes(blob.tobytes())
    B = 11
    idxs = [edits[i] if i < 9 else (977 * i + 3) % o.num_items for i in range(B)]
    qs = [cl.generate_query(idxs[i], 900 + i) for i in range(B)]
    sp.paths_taken()
    resp = sp.process_query_batch(p, [gpp] * B, qs, gdb)
    assert "sweep_batch_planar" in sp.paths_taken()
    for i in range(B):
        assert resp[i] == o.process_query(pp, qs[i], exp), (i, idxs[i])
    got = cl.decode_response(resp[4])
    item = blob[idxs[4] * isz:(idxs[4] + 1) * isz].tobytes()
    assert all(got[t * 64:(t + 1) * 64] == item[t * 64:(t + 1) * 64] for t in range(4))
    # bulk writer: the copy and its memory go; the next group builds it again
    gdb.load_items(blob)
    assert gdb.batch_copy_bytes() == 0
    sp.paths_taken()
    assert sp.process_query_batch(p, [gpp] * B, qs, gdb) == resp
    assert "sweep_batch_planar" in sp.paths_taken() and gdb.batch_copy_bytes() == copy
    # switched off: PACKED two-tile kernel, memory released
    sp.lib().sp_debug_set(b"batch_planar", C.c_long(0))
    try:
        sp.paths_taken()
        assert sp.process_query_batch(p, [gpp] * B, qs, gdb) == resp
        taken = sp.paths_taken()
        assert "sweep_batch_planar" not in taken and "sweep_batch_mfma_two_tiles" in taken
        assert gdb.batch_copy_bytes() == 0
        assert gdb.prepare_batch() is False
    finally:
        sp.lib().sp_debug_set(b"batch_planar", C.c_long(1))
    # a narrow (8-byte) database has no planar form
    assert sp.Database(sp.Params(dict(cfg, nu_2=3))).prepare_batch() is False


def test_process_query_batch_matrix_core_extreme_digits(sp, oracle_mod):
    """Database words whose residues sit at the edges of the signed-digit split (every byte 0x80 / 0x7f, q - 1, 0, the
    largest top digit) in every row: the i32 digit sums of k_sweep_mfma_batch reach their largest magnitudes; 256 rows.
    Responses are compared byte for byte with the oracle's exact u128 sums (server.rs:196-217)."""
    cfg = {"n": 2, "nu_1": 8, "nu_2": 7, "p": 256, "q2_bits": 20, "t_gsw": 4, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 512}
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(51)
    gpp = sp.PublicParameters.deserialize(p, pp)
    rng = np.random.default_rng(5)
    lo = np.array([0x00808080, 0x007F7F7F, Q0 - 1, 0, 0x0F7F7F7F, 0x0F808080, 0x0080807F, 1], dtype=np.uint64)
    hi = np.array([0x00808080, 0x007F7F7F, Q1 - 1, 0, 0x0E7F7F7F, 0x0E808080, 0x0E80807F, 1], dtype=np.uint64)
    n_words = 4 * 2048 * o.num_per * o.dim0
    pick = rng.integers(0, 8, n_words)
    db = (lo[pick] | (hi[(pick + 3) % 8] << np.uint64(32))).astype(np.uint64)
    gdb = sp.Database(p).load(db)
    B = 8
    qs = [cl.generate_query((977 * i + 1) % o.num_items, 600 + i) for i in range(B)]
    sp.paths_taken()
    resp = sp.process_query_batch(p, [gpp] * B, qs, gdb)
    assert "sweep_batch_mfma" in sp.paths_taken()
    for i in (0, 3, 7):
        assert resp[i] == o.process_query(pp, qs[i], db), i
    for i in range(B):
        assert resp[i] == sp.process_query(p, gpp, qs[i], gdb), i


def _valid_cfg(c):
    dim0, right = 1 << c["nu_1"], c["t_gsw"] * c["nu_2"]
    g = max(1, int(np.ceil(np.log2(right + dim0))))
    return 2 * max(dim0, right) <= (1 << g) and (c.get("version", 0) == 0 or c["n"] == 2)


_FUZZ = [
    # odd digit counts (one-transform fused kernel), 28-bit digits, 19-bit digits, small plaintext moduli,
    # several instances, packing version 1, ragged item sizes, nu_2 from 1 to 7
    dict(n=2, nu_1=6, nu_2=4, p=256, q2_bits=20, t_gsw=3, t_conv=4, t_exp_left=8, t_exp_right=56, instances=1, db_item_size=8192),
    dict(n=2, nu_1=6, nu_2=5, p=256, q2_bits=22, t_gsw=5, t_conv=3, t_exp_left=5, t_exp_right=28, instances=1, db_item_size=5000),
    dict(n=2, nu_1=5, nu_2=3, p=16, q2_bits=18, t_gsw=7, t_conv=2, t_exp_left=4, t_exp_right=14, instances=2, db_item_size=4096),
    dict(n=2, nu_1=5, nu_2=6, p=256, q2_bits=20, t_gsw=2, t_conv=4, t_exp_left=8, t_exp_right=56, instances=1, db_item_size=3000),
    dict(n=2, nu_1=6, nu_2=7, p=4, q2_bits=14, t_gsw=4, t_conv=7, t_exp_left=16, t_exp_right=8, instances=1, db_item_size=1024),
    dict(n=2, nu_1=7, nu_2=2, p=256, q2_bits=27, t_gsw=10, t_conv=4, t_exp_left=8, t_exp_right=56, instances=3, db_item_size=20000),
    dict(n=2, nu_1=6, nu_2=1, p=64, q2_bits=21, t_gsw=6, t_conv=5, t_exp_left=7, t_exp_right=9, instances=1, db_item_size=6000),
    dict(n=2, nu_1=6, nu_2=4, p=256, q2_bits=20, t_gsw=8, t_conv=4, t_exp_left=8, t_exp_right=56, instances=2, db_item_size=16384, version=1),
    dict(n=2, nu_1=5, nu_2=5, p=256, q2_bits=20, t_gsw=9, t_conv=4, t_exp_left=6, t_exp_right=19, instances=1, db_item_size=8192),
    dict(n=2, nu_1=6, nu_2=3, p=256, q2_bits=20, t_gsw=14, t_conv=14, t_exp_left=14, t_exp_right=14, instances=1, db_item_size=8192),
    # extreme gadgets: one 57-bit digit (no fused fold), 28-bit digits everywhere (digits can exceed q), t_conv = 1,
    # 2-bit digits (t = 28)
    dict(n=2, nu_1=5, nu_2=3, p=256, q2_bits=20, t_gsw=1, t_conv=4, t_exp_left=8, t_exp_right=56, instances=1, db_item_size=8192),
    dict(n=2, nu_1=5, nu_2=4, p=256, q2_bits=20, t_gsw=4, t_conv=2, t_exp_left=2, t_exp_right=2, instances=1, db_item_size=8192),
    dict(n=2, nu_1=4, nu_2=5, p=256, q2_bits=20, t_gsw=2, t_conv=1, t_exp_left=3, t_exp_right=4, instances=1, db_item_size=2048),
    dict(n=2, nu_1=6, nu_2=4, p=256, q2_bits=20, t_gsw=28, t_conv=28, t_exp_left=28, t_exp_right=56, instances=1, db_item_size=8192),
]


@pytest.mark.parametrize("fused_min", ["1", "256"], ids=["fused-all-levels", "default-threshold"])
@pytest.mark.parametrize("ci", range(len(_FUZZ)))
def test_config_sweep_response_parity(sp, oracle_mod, monkeypatch, ci, fused_min):
    """End-to-end response bytes over a spread of gadget widths / moduli / shapes, with the fused fold kernels forced
    onto every tree level and with the default level threshold."""
    cfg = _FUZZ[ci]
    assert _valid_cfg(cfg), cfg
    monkeypatch.setenv("SPIRAL_FUSED_MIN_PAIRS", fused_min)
    idx = (977 * (ci + 1)) % oracle_mod.Params(cfg).num_items
    c = _oracle_case(oracle_mod, cfg, idx, 60 + ci, 160 + ci)   # (shared by the two thresholds of a configuration)
    p = sp.Params(cfg)          # workspaces of this handle read the env at creation
    gpp = sp.PublicParameters.deserialize(p, c["pp"])
    gdb = sp.Database(p).load(c["db"])
    assert sp.process_query(p, gpp, c["q"], gdb) == c["resp"]


@pytest.mark.parametrize("mode", ["split", "launches", "one_launch_rounds"])
@pytest.mark.parametrize("ci", [0, 1, 2, 4, 6, 8, 11, 12, 13])
def test_expansion_variants_response_parity(sp, oracle_mod, monkeypatch, ci, mode):
    """The expansion schedules over gadget widths from 2 to 56 digits, 28-bit digits included: everything on one stream, and
    the odd subtree + GSW side on the second stream (SPIRAL_EXPAND_SPLIT, profiles/r02_expand_experiments.md); expand_query
    and the response must not change, twice in a row on the same workspace (join of the previous odd side).
    `one_launch_rounds`: every round's digit transforms and products in ONE launch (k_expand_round, which the library takes
    for rounds of expand_round_min digit transforms or more -- here for every round)."""
    cfg = _FUZZ[ci]
    # launches (the default of narrow databases): one stream; split (the default before a pipelined sweep): the odd subtree
    # + GSW side on the second stream
    monkeypatch.setenv("SPIRAL_EXPAND_SPLIT", "0" if mode == "launches" else "1")
    monkeypatch.setenv("SPIRAL_EXPAND_ROUND_MIN", "1" if mode == "one_launch_rounds" else str(1 << 30))
    monkeypatch.setenv("SPIRAL_EXPAND_ROUND_ODD", "1")   # (the odd subtree of a split expansion too; off by default)
    idx = (613 * (ci + 1)) % oracle_mod.Params(cfg).num_items
    c = _oracle_case(oracle_mod, cfg, idx, 70 + ci, 170 + ci)   # (shared by the two schedules of a configuration)
    o, pp, q, db = c["o"], c["pp"], c["q"], c["db"]
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    sp.paths_taken()
    v_reg, v_fold = sp.expand_query(p, gpp, q)
    taken = sp.paths_taken()
    assert ("expand_split" in taken) == (mode != "launches"), taken
    # (k_expand_round keeps its digits in 32-bit words: gadget digits of more than 28 bits -- t_exp = 2 -- stay on the three-launch form)
    narrow = all(56 // cfg[k] + 1 <= 28 for k in ("t_exp_left", "t_exp_right"))
    assert ("expand_round_one_launch" in taken) == (mode == "one_launch_rounds" and narrow), taken
    e_reg, e_fold = o.expand_query(pp, q)
    assert (v_reg == e_reg).all() and (v_fold == e_fold).all()
    expect = c["resp"]
    assert sp.process_query(p, gpp, q, gdb) == expect
    assert sp.process_query(p, gpp, q, gdb) == expect
    # row shard: pruned even subtree through the same schedules
    shard = sp.Database(p, shard=1, num_shards=2).load(db)
    run = sp.QueryRun(p, gpp, q, db=shard)
    run.sweep(shard)
    full = sp.QueryRun(p, gpp, q, db=gdb)
    full.sweep(gdb)
    assert "expand_pruned" in sp.paths_taken()
    run.free()
    full.free()


@pytest.mark.parametrize("ci", [0, 1, 2, 4, 5, 8, 9, 11, 13])
def test_wave_fold_kernel_gadget_widths(sp, oracle_mod, monkeypatch, ci):
    """k_fold_wave (SPIRAL_FOLD_VARIANT=5) on every tree level over odd and even t_gsw from 2 to 28 (byte, short and
    word digit planes: 2..28-bit digits), against the oracle's response."""
    cfg = _FUZZ[ci]
    monkeypatch.setenv("SPIRAL_FOLD_VARIANT", "5")
    monkeypatch.setenv("SPIRAL_FUSED_MIN_PAIRS", "1")
    o = oracle_mod.Params(cfg)
    idx = (331 * (ci + 1)) % o.num_items
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(90 + ci)
    q = cl.generate_query(idx, 190 + ci)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(idx)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    sp.paths_taken()
    resp = sp.process_query(p, gpp, q, gdb)
    # the kernel is used while two workgroups fit a CU: 34 KiB + 2 t_live (2048 ES + 256) bytes of LDS <= 80 KiB, t_live = the
    # digits that can be non-zero below Q (56 bits): 7 of t_gsw = 8, 8 of 9, 12 of 14 (FoldDesc::t_live)
    bits = o.get_bits_per(cfg["t_gsw"])
    es = 1 if bits <= 8 else 2 if bits <= 16 else 4
    t_live = min(cfg["t_gsw"], -(-56 // bits))
    fits = 34816 + 2 * t_live * (2048 * es + 256) <= 80 * 1024
    taken = sp.paths_taken()
    assert ("fold_wave" in taken) == fits, (taken, es, fits)
    expect = o.process_query(pp, q, db)
    assert resp == expect
    if t_live < cfg["t_gsw"]:        # the same query with the dead digits transformed as well (what rounds 2-4 did)
        monkeypatch.setenv("SPIRAL_FOLD_SKIP_DEAD_DIGITS", "0")
        p0 = sp.Params(cfg)
        assert sp.process_query(p0, sp.PublicParameters.deserialize(p0, pp), q, sp.Database(p0).load(db)) == expect
