"""Pins the CPU oracle's L1 (server.rs stage functions + process_query) the way the reference's own tests do:
decrypt-and-compare after each stage (server.rs:788-993) and the full protocol round trip
(server.rs:995-1043), plus serialise/deserialise identities (client.rs:848-955)."""
import numpy as np
import pytest

from conftest import FAST, FAST56, SMALL_INST2

Q = 66974689739603969


def _full_protocol(oracle_mod, cfg, idx, seed=1):
    p = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(p)
    pp = cl.generate_keys(seed)
    assert len(pp) == p.setup_bytes
    q = cl.generate_query(idx, seed + 100)
    item, db = p.generate_random_db_and_get_item(idx)
    resp = p.process_query(pp, q, db)
    assert len(resp) == p.response_bytes()
    got = cl.decode_response(resp)
    exp = p.item_to_vec(item)
    assert got == exp
    return p, cl, pp, q, db, item, resp


@pytest.mark.parametrize("idx", [0, 77, 255])
def test_full_protocol_is_correct(oracle_mod, idx):  # server.rs:1045-1048
    _full_protocol(oracle_mod, FAST, idx)


def test_full_protocol_t_exp_right_56(oracle_mod):
    _full_protocol(oracle_mod, FAST56, 301)


@pytest.mark.parametrize("nu2", [0, 1])
def test_full_protocol_shallow_second_dimension(oracle_mod, nu2):
    _full_protocol(oracle_mod, dict(FAST, nu_2=nu2), 17)


def test_full_protocol_two_instances(oracle_mod):
    _full_protocol(oracle_mod, SMALL_INST2, 123)


def test_full_protocol_version1_keys_share_expansion(oracle_mod):
    # version > 0 with t_exp_left == t_exp_right omits v_expansion_right on the wire (client.rs:229, params.rs:158)
    cfg = dict(FAST, version=1)
    p = oracle_mod.Params(cfg)
    assert p.setup_bytes < oracle_mod.Params(FAST).setup_bytes
    # server.rs pack() is version 0 only -- only keys/serialisation are checked here
    cl = oracle_mod.Client(p)
    assert len(cl.generate_keys(5)) == p.setup_bytes


def test_wrong_index_does_not_decode(oracle_mod):
    p, cl, pp, q, db, item, resp = _full_protocol(oracle_mod, FAST, 10)
    item2, _ = p.generate_random_db_and_get_item(11)
    assert cl.decode_response(resp) != p.item_to_vec(item2)


def test_multiply_reg_by_database_is_correct(oracle_mod):  # server.rs:870-925
    p = oracle_mod.Params(FAST)
    cl = oracle_mod.Client(p)
    cl.generate_keys(9)
    idx = 201
    item, db = p.generate_random_db_and_get_item(idx)
    scale_k = Q // p.pt_modulus
    cts = []
    for i in range(p.dim0):
        pt = np.zeros(p.poly_len, dtype=np.uint64)
        pt[0] = scale_k if i == idx // p.num_per else 0
        cts.append(cl.encrypt_reg(pt, seed=1000 + i, seed_pub=5000 + i))
    v_reg = p.reorient_reg_ciphertexts(np.concatenate(cts))
    slice_words = p.dim0 * p.num_per * p.poly_len
    for trial in range(4):
        out = p.multiply_reg_by_database(db[trial * slice_words:(trial + 1) * slice_words], v_reg)
        dec = cl.decrypt_reg(out.reshape(p.num_per, -1)[idx % p.num_per])[0]
        resc = np.array([oracle_mod.scalar("rescale", int(x), Q, p.pt_modulus) for x in dec], dtype=np.uint64)
        exp = item.reshape(4, p.poly_len)[trial]
        assert (resc == exp).all()


def test_expand_then_stages_decrypt(oracle_mod):  # server.rs:788-868, 928-993 rolled into the real pipeline
    p = oracle_mod.Params(FAST)
    cl = oracle_mod.Client(p)
    pp = cl.generate_keys(11)
    idx = 150
    q = cl.generate_query(idx, 12)
    v_reg, v_fold = p.expand_query(pp, q)
    # v_reg decrypts to the one-hot over dim0: re-assemble ct j from the reoriented buffer
    N, d0 = p.poly_len, p.dim0
    vr = v_reg.reshape(N, d0, 2)
    scale_k = Q // p.pt_modulus
    for j in (idx // p.num_per, (idx // p.num_per + 1) % d0):
        ct = np.zeros(2 * 2 * N, dtype=np.uint64)
        for r in range(2):
            ct[r * 2 * N:r * 2 * N + N] = vr[:, j, r] & 0xFFFFFFFF
            ct[r * 2 * N + N:(r + 1) * 2 * N] = vr[:, j, r] >> 32
        dec = cl.decrypt_reg(ct)[0]
        c0 = int(dec[0])
        c0 = c0 - Q if c0 >= Q // 2 else c0
        assert round(c0 / scale_k) == (1 if j == idx // p.num_per else 0)
    # fold over the sweep output decrypts to the item
    item, db = p.generate_random_db_and_get_item(idx)
    slice_words = d0 * p.num_per * N
    out = p.multiply_reg_by_database(db[:slice_words], v_reg)
    raw = p.from_ntt(out)
    v_neg = p.get_v_folding_neg(v_fold)
    folded = p.fold_ciphertexts(raw, v_fold, v_neg)[:2 * N]
    dec = cl.decrypt_reg(p.to_ntt(folded))[0]
    resc = np.array([oracle_mod.scalar("rescale", int(x), Q, p.pt_modulus) for x in dec], dtype=np.uint64)
    assert (resc == item.reshape(4, N)[0]).all()


def test_pp_and_query_wire_identity(oracle_mod):
    """deserialize regenerates row 0 from the seed; the NTT-form pp must equal what the client built."""
    p = oracle_mod.Params(FAST)
    cl = oracle_mod.Client(p)
    pp = cl.generate_keys(21)
    flat = p.pp_deserialize_flat(pp)
    assert flat.size == p.pp_poly_count() * p.ntt_words
    # every residue canonical
    f = flat.reshape(-1, 2, p.poly_len)
    assert (f[:, 0] < 268369921).all() and (f[:, 1] < 249561089).all()
    q = cl.generate_query(3, 22)
    ct = p.query_deserialize_ct(q)
    assert (ct[:p.poly_len] <= Q).all() and (ct[:p.poly_len] > 0).all()   # get_inv_from_rng in (0, Q]
    assert (ct[p.poly_len:] < Q).all()
    # determinism
    assert cl.generate_query(3, 22) == q
    assert oracle_mod.Client(p).generate_keys(21) == pp


def test_load_db_from_bytes_roundtrip(oracle_mod):  # server.rs:277-357
    cfg = dict(FAST, db_item_size=256)
    p = oracle_mod.Params(cfg)
    rng = np.random.default_rng(8)
    blob = rng.integers(0, 256, p.num_items * p.db_item_size, dtype=np.uint8).tobytes()
    db = p.load_db_from_bytes(blob)
    cl = oracle_mod.Client(p)
    pp = cl.generate_keys(31)
    idx = 99
    resp = p.process_query(pp, cl.generate_query(idx, 32), db)
    got = cl.decode_response(resp)
    # to_vec lays each of the n*n chunks at a byte boundary: bytes_per_chunk = 64 here
    bpc = p.bytes_per_chunk
    item = blob[idx * p.db_item_size:(idx + 1) * p.db_item_size]
    for t in range(4):
        assert got[t * bpc:(t + 1) * bpc] == item[t * bpc:(t + 1) * bpc]
