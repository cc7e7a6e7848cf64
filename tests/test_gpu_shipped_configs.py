"""The parameter sets the reference SHIPS, verbatim, on the GPU (VERDICT r05 missing #2):

  CFG_16_100000        lib/spiral-rs/src/util.rs:21-34   p = 512 (9-bit plaintext words through read_arbitrary_bits in
                       load_item_from_seek, server.rs:277-318), nu_1 = 10 (1024-row first dimension), 11 instances = 44 planes,
                       t_gsw = 10, t_exp_left = 16, q2_bits = 21, 100000-byte items (not a multiple of the 44 chunks)
  e2e-tests/params/v0.json   n = 4 with query expansion (16 planes per instance, 4 x 4 packing), 32 KiB items
  e2e-tests/params/v1.json   packing version 1, t = (7, 3, 5, 5), four instances (the lib/server default family)

Each at a shrunk size where the oracle holds the whole real database (resident words, response bytes, decode) and at the
full size the reference ships it: real items preprocessed on the GPU + decode, and -- where no host buffer holds the encoded
database -- the synthetic database against oracle.process_query_synth.  Everything goes through the C ABI; the expected side is
always the oracle.
"""
import gc
import hashlib
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x123456789
CFG_16_100000 = {"n": 2, "nu_1": 10, "nu_2": 6, "p": 512, "q2_bits": 21, "t_gsw": 10, "t_conv": 4, "t_exp_left": 16,
                 "t_exp_right": 56, "instances": 11, "db_item_size": 100000}                     # util.rs:21-34, verbatim
E2E_V0 = {"n": 4, "nu_1": 9, "nu_2": 5, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
          "t_exp_right": 56, "instances": 1, "db_item_size": 32768}                              # e2e-tests/params/v0.json, verbatim
E2E_V1 = {"n": 2, "nu_1": 9, "nu_2": 5, "p": 256, "q2_bits": 22, "t_gsw": 7, "t_conv": 3, "t_exp_left": 5,
          "t_exp_right": 5, "instances": 4, "db_item_size": 32768, "version": 1}                 # e2e-tests/params/v1.json, verbatim
N = 2048


@pytest.fixture(scope="module")
def sp():
    import sdk_amd
    assert sdk_amd.lib().sp_device_count() >= 1, "no HIP device visible"
    return sdk_amd


def _need_hbm(gib):
    import torch
    gc.collect()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info()[0]
    if free < gib * 2**30:
        pytest.skip("needs %d GiB of free HBM, %.1f available" % (gib, free / 2**30))


def _chunks(o):
    """(chunks, bytes per chunk in the item file, bytes per polynomial in a decoded response): load_item_from_seek,
    server.rs:286-292; decode_response writes modp_words_per_chunk words of log2(p) bits per polynomial and rounds the bit
    offset DOWN to a byte after each (client.rs:809, poly.rs:213-235, params.rs:195-200)"""
    chunks = o.instances * o.n * o.n
    logp = int(np.ceil(np.log2(o.pt_modulus)))
    bpc = -(-o.db_item_size // chunks)
    words = -(-bpc * 8 // logp)
    return chunks, bpc, words * logp // 8


def _assert_decodes_to_item(o, got, blob, idx):
    """polynomial c of the decoded response starts with the bytes_per_chunk bytes the loader read for chunk c of item idx --
    which run on into the next item(s) when db_item_size is not a multiple of the chunk count, and are zero past the end
    of the file (server.rs:296-309)"""
    chunks, bpc, stride = _chunks(o)
    blob = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    got = np.frombuffer(got, dtype=np.uint8)
    for c in range(chunks):
        start = idx * o.db_item_size + c * bpc
        want = np.zeros(bpc, dtype=np.uint8)
        avail = blob[start:start + bpc]
        want[:avail.size] = avail
        assert (got[c * stride:c * stride + bpc] == want).all(), (idx, c)


# --------------------------------------------------------------------------------------------- CFG_16_100000
@pytest.mark.parametrize("shrink", [dict(nu_1=6, nu_2=3, instances=2, db_item_size=18000),
                                    dict(nu_2=2)],
                         ids=["p512-small", "p512-nu1_10-inst11"])
def test_cfg_16_100000_real_database(sp, oracle_mod, shrink):
    """CFG_16_100000 with only the database shortened (second case: nu_1 = 10, 11 instances, 100000-byte items and every gadget
    verbatim, nu_2 = 2: 2.75 GiB encoded): sp_db_load_items == load_db_from_seek word for word at p = 512, sp_db_update_item on
    top of it, response bytes == the oracle's on that database, and the response decodes to the item's bytes."""
    cfg = dict(CFG_16_100000, **shrink)
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    planes = o.instances * o.n * o.n
    rng = np.random.default_rng(o.db_item_size + o.instances)
    blob = rng.integers(0, 256, o.num_items * o.db_item_size, dtype=np.uint8)
    exp = o.load_db_from_bytes(blob.tobytes())
    exp4 = exp.reshape(planes, N, o.num_per, o.dim0)
    db = sp.Database(p).load_items(blob)
    for pl in sorted({0, 1, planes // 2, planes - 1}):
        for z in (0, 1, 1000, N - 1):
            for ii in range(o.num_per):
                assert (db.read_ref(pl, z, ii, 0, o.dim0) == exp4[pl, z, ii]).all(), (pl, z, ii)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(1601)
    gpp = sp.PublicParameters.deserialize(p, pp)
    idxs = (0, o.num_items // 2 + 3, o.num_items - 1)
    for k, idx in enumerate(idxs):
        q = cl.generate_query(idx, 1602 + k)
        resp = sp.process_query(p, gpp, q, db)
        assert len(resp) == o.response_bytes()
        if k < 2:
            assert resp == o.process_query(pp, q, exp), idx
        _assert_decodes_to_item(o, cl.decode_response(resp), blob, idx)
    # upsert: a short record is zero padded, the neighbours keep their bytes (lib/server loading.rs:317-359)
    idx = idxs[1]
    rec = rng.integers(0, 256, o.db_item_size - 5, dtype=np.uint8)
    db.update_item(idx, rec.tobytes())
    blob2 = blob.copy()
    blob2[idx * o.db_item_size:(idx + 1) * o.db_item_size] = 0
    blob2[idx * o.db_item_size:idx * o.db_item_size + rec.size] = rec
    chunks, bpc, _ = _chunks(o)
    if chunks * bpc == o.db_item_size:      # no spill: the bulk loader on the edited file is the expected database
        exp2 = o.load_db_from_bytes(blob2.tobytes()).reshape(planes, N, o.num_per, o.dim0)
        j, ii = idx // o.num_per, idx % o.num_per
        for pl in (0, planes - 1):
            assert (db.read_ref(pl, 77, ii, 0, o.dim0) == exp2[pl, 77, ii]).all()
            assert db.read_ref(pl, 77, ii, j, 1)[0] != exp4[pl, 77, ii, j]      # and the item did change
    q = cl.generate_query(idx, 1610)
    got = np.frombuffer(cl.decode_response(sp.process_query(p, gpp, q, db)), dtype=np.uint8)
    _, _, stride = _chunks(o)
    for c in range(chunks):                 # an upserted item is its own record, zero padded: no spill into the neighbour
        want = np.zeros(bpc, dtype=np.uint8)
        part = rec[c * bpc:(c + 1) * bpc]
        want[:part.size] = part
        assert (got[c * stride:c * stride + bpc] == want).all(), c
    del db, exp, exp4
    gc.collect()


def test_cfg_16_100000_verbatim_full_size(sp, oracle_mod):
    """CFG_16_100000 exactly as util.rs:21-34 ships it: 2^16 items x 100000 B, 44 planes of 1024 x 64 polynomials = 44 GiB
    resident (8-byte form: num_per = 64 < 128).  (i) response bytes on the synthetic database == the oracle's
    (process_query_synth: the u128 sweep over regenerated words, then the unmodified restatement); (ii) the REAL database:
    6.1 GiB of random item bytes preprocessed on the GPU at p = 512, three queries decode to their items."""
    _need_hbm(52)
    cfg = CFG_16_100000
    o = oracle_mod.Params(cfg)
    assert (o.dim0, o.num_per, o.instances * o.n * o.n) == (1024, 64, 44)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(1621)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    q = cl.generate_query(43210, 1622)
    t0 = time.time()
    expect = o.process_query_synth(pp, q, SEED)
    print("oracle: %.1f s for the CFG_16_100000 query" % (time.time() - t0))
    db = sp.Database(p).fill_synthetic(SEED)
    assert db.device_bytes() >= 44 * 2**30
    sp.paths_taken()
    got = sp.process_query(p, gpp, q, db)
    taken = sp.paths_taken()
    assert "sweep_narrow" in taken, taken
    assert got == expect, "CFG_16_100000 response differs from the oracle (sha %s vs %s)" % (
        hashlib.sha256(got).hexdigest()[:16], hashlib.sha256(expect).hexdigest()[:16])
    # the list entry point on the same database (narrow: one pass per query, queries in flight)
    q2 = cl.generate_query(0, 1623)
    outs = sp.process_query_batch(p, gpp, [q, q2], db)
    assert outs[0] == expect and outs[1] == o.process_query_synth(pp, q2, SEED)
    del db
    gc.collect()
    rng = np.random.default_rng(1624)
    blob = rng.integers(0, 256, o.num_items * o.db_item_size, dtype=np.uint8)
    db = sp.Database(p).load_items(blob)
    for k, idx in enumerate((0, 40000, o.num_items - 1)):
        resp = sp.process_query(p, gpp, cl.generate_query(idx, 1630 + k), db)
        _assert_decodes_to_item(o, cl.decode_response(resp), blob, idx)
    del db, blob
    gc.collect()


# --------------------------------------------------------------------------------------------- e2e-tests/params
@pytest.mark.parametrize("cfg", [dict(E2E_V0, nu_1=6, nu_2=2), E2E_V0, E2E_V1], ids=["v0-small", "v0-verbatim", "v1-verbatim"])
def test_e2e_params_real_database(sp, oracle_mod, cfg):
    """The two parameter files of the reference's end-to-end tests (e2e-tests/params/v0.json: n = 4 WITH query expansion;
    v1.json: packing version 1) on a real random database (generate_random_db_and_get_item, server.rs:223-275; 4 GiB encoded at
    full size): response bytes == oracle, decode == the planted item (server.rs:1029-1042)."""
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(1701)
    idx = (o.num_items * 5) // 7
    item, db = o.generate_random_db_and_get_item(idx)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    q = cl.generate_query(idx, 1702)
    resp = sp.process_query(p, gpp, q, gdb)
    assert len(resp) == o.response_bytes()
    assert resp == o.process_query(pp, q, db)
    assert cl.decode_response(resp) == o.item_to_vec(item)
    q2 = cl.generate_query(o.num_items - 1, 1703)
    q3 = cl.generate_query(0, 1704)
    assert sp.process_query_batch(p, gpp, [q2, q, q3], gdb) == [o.process_query(pp, q2, db), resp, o.process_query(pp, q3, db)]
    # and through the item loader: 32 KiB items, chunked over the n * n * instances polynomials
    rng = np.random.default_rng(1705)
    blob = rng.integers(0, 256, o.num_items * o.db_item_size, dtype=np.uint8)
    del gdb, db
    gc.collect()
    gdb = sp.Database(p).load_items(blob)
    exp = o.load_db_from_bytes(blob.tobytes())
    resp = sp.process_query(p, gpp, q, gdb)
    assert resp == o.process_query(pp, q, exp)
    _assert_decodes_to_item(o, cl.decode_response(resp), blob, idx)
    del gdb, exp
    gc.collect()


# --------------------------------------------------------------------------------------------- the same sets, row-sharded
@pytest.mark.parametrize("cfg,G", [(E2E_V0, 4), (dict(CFG_16_100000, nu_2=3), 8), (E2E_V1, 2)],
                         ids=["v0-verbatim-G4", "p512-inst11-G8", "v1-verbatim-G2"])
def test_shipped_configs_row_sharded_loopback(sp, oracle_mod, cfg, G):
    """The multi-GPU answer path (sp_process_query_sharded: per-plane reduce-scatter, distributed fold, all-gather; G ranks as host
    threads on this one GPU over the loopback transport) on the reference's shipped parameter sets: e2e v0.json verbatim (n = 4: 16
    planes, 4 x 4 packing) over 4 ranks, CFG_16_100000 with every gadget and its 44 planes verbatim (nu_1 = 10: 128 rows per rank,
    p = 512; nu_2 shortened to 3) over 8, e2e v1.json verbatim (packing version 1) over 2.  Rank 0's response == the oracle's
    process_query on the unsharded real database, a single query and a pipelined list."""
    from sdk_amd.sharding import LoopbackWorld
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(1801)
    idx = o.num_items // 3
    item, db = o.generate_random_db_and_get_item(idx)
    qs = [cl.generate_query(idx, 1802), cl.generate_query(o.num_items - 1, 1803), cl.generate_query(0, 1804)]
    expect = [o.process_query(pp, q, db) for q in qs]
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    shards = [sp.Database(p, s, G).load(db) for s in range(G)]
    del db
    gc.collect()
    world = LoopbackWorld(G)

    def rank_main(r):
        sp.lib().sp_set_device(0)
        sp.paths_taken()
        one = world.comm(r).process_query(p, gpp, qs[0], shards[r])
        lst = world.comm(r).process_queries(p, gpp, qs, shards[r])
        return one, lst, sp.paths_taken()
    res = world.run(rank_main)
    assert res[0][0] == expect[0]
    assert res[0][1] == expect
    for r in range(G):
        assert {"scatter_out", "custom_transport", "expand_pruned"} <= res[r][2], res[r][2]
    assert cl.decode_response(res[0][0]) == o.item_to_vec(item)
    del shards, world
    gc.collect()
