"""Full-size byte parity on the BASELINE.json configurations: the GPU response must equal the
oracle's response byte for byte, on the production code path (asserted through sp_paths_taken).

The databases are synthetic -- reference-layout word i = synth_word(seed, i) -- because 64-256 GiB of encoded
database fits no host buffer; the oracle regenerates the same words row by row inside its u128 multiply-accumulate
(oracle.Params.process_query_synth, ~10 s of CPU for C2) and runs the unmodified restatement for everything after
the sweep.  Queries and public parameters are REAL ones from the oracle client, so expansion, GSW conversion and
the fold operate on genuine ciphertexts.  Decode checks use a zero-initialised bucket with planted items
(sp_db_update_item), the reference's SparseDb semantics.

  C2 = configs[1]  2^20 x 256 B           64 GiB encoded / 56 GiB resident     (the headline config)
  P2 = CFG_20_256  2^15 x 8 KiB            the reference's own "2^20 x 256 B" preset (util.rs:7-20)
  C4 = configs[3]  2^20 x 32 KiB           256 GiB encoded / 224 GiB resident, 16 planes
  C3 = configs[2]  2^22 x 256 B            256 GiB encoded: unsharded on one GPU, and one row shard of 8
"""
import gc
import hashlib
import time

import numpy as np
import pytest

from conftest import C1, C2, P2

pytestmark = pytest.mark.gpu

SEED = 0x123456789          # util.rs:171-173 get_static_seed; bench.py fills its database with the same seed
C3 = dict(C1, nu_2=13)
C4 = dict(C1, nu_2=11, instances=4, db_item_size=32768)
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


def _client(oracle_mod, cfg, seed):
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    return o, cl, cl.generate_keys(seed)


PRODUCTION_C2 = {"sweep_packed_persist", "sweep_ring", "from_sweep4", "from_sweep4_xcd_order", "fold_fused", "pipelined_fold_overlap",
                 "fold_tail_batched"}


def test_synth_word_copies_agree(sp, oracle_mod):
    """the oracle's restatement of the synthetic-word hash == the product's (host and device copies)"""
    from sdk_amd.spiral import synth_word
    for i in (0, 1, 12345, 2**31 + 7, 2**35 - 1):
        assert oracle_mod.synth_word(SEED, i) == synth_word(SEED, i)
    p = sp.Params(dict(C1, nu_2=7))
    db = sp.Database(p).fill_synthetic(SEED)
    got = db.read_ref(2, 1000, 77, 100, 8)
    base = ((2 * N + 1000) * 128 + 77) * 512 + 100
    assert [int(x) for x in got] == [oracle_mod.synth_word(SEED, base + k) for k in range(8)]


def test_c2_full_size_response_bytes(sp, oracle_mod):
    """BASELINE.json configs[1] (the headline config) at full size, byte for byte against the oracle, through the
    kernels bench.py times: persistent PACKED sweep one plane per launch, k_from_sweep4 with the XCD-aware block
    order at num_per = 2048, three fused fold levels + the tree tail, the fold of plane p overlapped with the
    sweep of plane p+1 on the second stream.  lib/spiral-rs/src/server.rs:1029-1042 (full-protocol equality)."""
    _need_hbm(62)
    o, cl, pp = _client(oracle_mod, C2, 501)
    p = sp.Params(C2)
    gpp = sp.PublicParameters.deserialize(p, pp)
    db = sp.Database(p).fill_synthetic(SEED)
    queries = [cl.generate_query(idx, 900 + k) for k, idx in enumerate((0, 777777, o.num_items - 1))]
    t0 = time.time()
    expect = [o.process_query_synth(pp, q, SEED) for q in queries[:2]]
    print("oracle: %.1f s per C2 query" % ((time.time() - t0) / 2))
    sp.paths_taken()
    got = sp.process_query(p, gpp, queries[0], db)
    taken = sp.paths_taken()
    assert PRODUCTION_C2 <= taken, taken
    assert {"fold_tail_delta", "fold_tail_persistent"} & taken, taken
    assert not ({"sweep_packed", "sweep_wide", "sweep_narrow", "from_sweep1", "fold_tail_literal"} & taken), taken
    assert got == expect[0], "C2 response differs from the oracle (sha %s vs %s)" % (
        hashlib.sha256(got).hexdigest()[:16], hashlib.sha256(expect[0]).hexdigest()[:16])
    # the split API (what bench.py's step drives) and a second query
    run = sp.QueryRun(p, gpp, queries[1], db=db).sweep(db)
    assert run.finish() == expect[1]
    run.free()
    # the same two queries sharing ONE database pass (k_sweep_packed_batch), and the non-pipelined single launch
    sp.paths_taken()
    assert sp.process_query_batch(p, gpp, queries[:2], db) == expect
    assert "sweep_batch" in sp.paths_taken()
    # every stage still decodes: responses are ciphertexts of SOMETHING only for real item encodings, so decode
    # is checked on the planted-item database (test_c2_full_size_decodes_planted_items in test_gpu_parity.py)


def test_c2_batch8_matrix_core_pass_at_full_size(sp, oracle_mod):
    """BASELINE.json configs[4] at one GPU, full size: EIGHT queries share one database pass, which from four queries
    on runs on the matrix cores (k_sweep_mfma_batch, v_mfma_i32_16x16x64_i8) -- the batch of two in the test above is below
    batch_mfma_min and takes the vector kernel.  First and last response of the group byte for byte against the oracle
    (lib/server/src/bin/server.rs:152-158: the per-request loop over process_query), the six in between against the
    one-at-a-time path that the test above ties to the oracle."""
    _need_hbm(96)
    o, cl, pp = _client(oracle_mod, C2, 501)
    p = sp.Params(C2)
    gpp = sp.PublicParameters.deserialize(p, pp)
    db = sp.Database(p).fill_synthetic(SEED)
    idxs = (5, 1 << 19, 777777, 123456, o.num_items - 1, 0, 999999, 31337)
    queries = [cl.generate_query(idx, 950 + k) for k, idx in enumerate(idxs)]
    sp.paths_taken()
    outs = sp.process_query_batch(p, gpp, queries, db)
    taken = sp.paths_taken()
    assert {"sweep_batch", "sweep_batch_mfma"} <= taken, taken
    for k in (0, 7):
        want = o.process_query_synth(pp, queries[k], SEED)
        assert outs[k] == want, "query %d of the batched pass differs from the oracle (sha %s vs %s)" % (
            k, hashlib.sha256(outs[k]).hexdigest()[:16], hashlib.sha256(want).hexdigest()[:16])
    for k in range(1, 7):
        assert outs[k] == sp.process_query(p, gpp, queries[k], db), k
    del db
    gc.collect()


def test_c2_batch16_two_query_tiles_at_full_size(sp, oracle_mod):
    """Sixteen queries per database pass at full size.  The production pass of a 9-16-query group is k_sweep_planar over the
    digit-planar copy of the database (sweep_planar.hpp: +64 GiB beside the 56 GiB PACKED words, built on the first group) --
    asserted through sp_paths_taken, so that a silent fall-back to the PACKED two-tile kernel (no room for the copy) fails here
    instead of passing: one response of each query tile byte for byte against the oracle.  Then the same sixteen through the
    fall-back itself (`batch_planar` = 0: k_sweep_mfma_batch with two query tiles, the path of databases too large for a copy,
    C3 / C4) and through groups of 8 (one tile, the path the test above ties to the oracle): identical bytes."""
    import ctypes as C
    _need_hbm(130)
    o, cl, pp = _client(oracle_mod, C2, 501)
    p = sp.Params(C2)
    gpp = sp.PublicParameters.deserialize(p, pp)
    db = sp.Database(p).fill_synthetic(SEED)
    queries = [cl.generate_query((65537 * k + 11) % o.num_items, 970 + k) for k in range(16)]
    sp.paths_taken()
    outs = sp.process_query_batch(p, gpp, queries, db)
    taken = sp.paths_taken()
    assert {"sweep_batch", "sweep_batch_mfma", "sweep_batch_mfma_two_tiles", "sweep_batch_planar"} <= taken, taken
    for k in (3, 12):
        want = o.process_query_synth(pp, queries[k], SEED)
        assert outs[k] == want, "query %d of the sixteen differs from the oracle (sha %s vs %s)" % (
            k, hashlib.sha256(outs[k]).hexdigest()[:16], hashlib.sha256(want).hexdigest()[:16])
    for switch, value, must, must_not in ((b"batch_planar", 0, "sweep_batch_mfma_two_tiles", "sweep_batch_planar"),
                                          (b"batch_group", 8, "sweep_batch_mfma", "sweep_batch_mfma_two_tiles")):
        sp.lib().sp_debug_set(switch, C.c_long(value))
        try:
            sp.paths_taken()
            assert sp.process_query_batch(p, gpp, queries, db) == outs, switch
            taken = sp.paths_taken()
            assert must in taken and must_not not in taken, (switch, taken)
        finally:
            sp.lib().sp_debug_set(switch, C.c_long(1 if switch == b"batch_planar" else 0))
    del db
    gc.collect()


def test_c2_fold_thresholds_agree_at_full_size(sp, oracle_mod, monkeypatch):
    """The two extremes of the fold dispatch produce the oracle's bytes at full size as well: every level through the
    fused kernel (threshold 1), and no level through it (threshold 2048 > the 1024 pairs of a plane's first level:
    the whole tree runs as the non-fused tail, 1024-pair launches included)."""
    _need_hbm(62)
    o, cl, pp = _client(oracle_mod, C2, 502)
    q = cl.generate_query(424242, 77)
    expect = o.process_query_synth(pp, q, SEED)
    for fused_min in ("1", "2048"):
        monkeypatch.setenv("SPIRAL_FUSED_MIN_PAIRS", fused_min)
        p = sp.Params(C2)           # workspaces of this handle read the env at creation
        gpp = sp.PublicParameters.deserialize(p, pp)
        db = sp.Database(p).fill_synthetic(SEED)
        sp.paths_taken()
        assert sp.process_query(p, gpp, q, db) == expect, fused_min
        taken = sp.paths_taken()
        if fused_min == "1":
            assert "fold_fused" in taken and not ({"fold_tail_delta", "fold_tail_persistent"} & taken), taken
        else:
            assert "fold_fused" not in taken and ({"fold_tail_delta", "fold_tail_persistent"} & taken), taken
        del db, gpp, p
        gc.collect()


def test_p2_cfg_20_256_end_to_end(sp, oracle_mod):
    """The reference's own "2^20 x 256 B" preset CFG_20_256 (util.rs:7-20; 2^15 elements x 8 KiB, 2 GiB encoded):
    response bytes equal the oracle's on a real random database and decode to the queried element
    (server.rs:1029-1042 on the preset of util.rs:7-20)."""
    o, cl, pp = _client(oracle_mod, P2, 601)
    p = sp.Params(P2)
    gpp = sp.PublicParameters.deserialize(p, pp)
    idx = 31337 % o.num_items
    item, db = o.generate_random_db_and_get_item(idx)
    gdb = sp.Database(p).load(db)
    q = cl.generate_query(idx, 602)
    sp.paths_taken()
    resp = sp.process_query(p, gpp, q, gdb)
    assert "sweep_narrow" in sp.paths_taken()
    assert resp == o.process_query(pp, q, db)
    assert cl.decode_response(resp) == o.item_to_vec(item)
    # a second element through the batched entry point
    q2 = cl.generate_query(0, 603)
    assert sp.process_query_batch(p, gpp, [q, q2], gdb) == [resp, o.process_query(pp, q2, db)]


def test_c4_full_size_response_bytes_and_decode(sp, oracle_mod):
    """BASELINE.json configs[3]: 2^20 items x 32 KiB, 16 planes, 224 GiB resident.  (i) response bytes on the
    synthetic database equal the oracle's; (ii) a zero-initialised bucket with three planted 32 KiB items decodes
    to those items (and to zeros for an absent one)."""
    _need_hbm(236)
    o, cl, pp = _client(oracle_mod, C4, 701)
    p = sp.Params(C4)
    gpp = sp.PublicParameters.deserialize(p, pp)
    q = cl.generate_query(555555, 702)
    t0 = time.time()
    expect = o.process_query_synth(pp, q, SEED)
    print("oracle: %.1f s for the C4 query" % (time.time() - t0))
    db = sp.Database(p).fill_synthetic(SEED)
    sp.paths_taken()
    got = sp.process_query(p, gpp, q, db)
    taken = sp.paths_taken()
    assert {"sweep_packed_persist", "from_sweep4", "fold_fused", "pipelined_fold_overlap"} <= taken, taken
    assert got == expect
    del db
    gc.collect()
    db = sp.Database(p)                      # the empty bucket: all-zero polynomials
    rng = np.random.default_rng(703)
    planted = {}
    for idx in (0, 555555, o.num_items - 1):
        planted[idx] = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
        db.update_item(idx, planted[idx])
    chunk = 32768 // 16
    for idx in (555555, 0, o.num_items - 1, 12345):
        resp = sp.process_query(p, gpp, cl.generate_query(idx, 800 + idx % 97), db)
        got = cl.decode_response(resp)
        want = planted.get(idx, bytes(32768))
        assert all(got[t * chunk:(t + 1) * chunk] == want[t * chunk:(t + 1) * chunk] for t in range(16)), idx
    del db
    gc.collect()


def test_c3_full_size_unsharded_and_row_shard(sp, oracle_mod):
    """BASELINE.json configs[2]: 2^22 items x 256 B (nu = (9, 13): 13 fold levels, 1 GiB first-dimension output).
    (i) the whole database on ONE GPU (224 GiB resident): response bytes equal the oracle's; (ii) row shard 0 of 8
    (what one rank of the 8-GPU run holds, 28 GiB): sampled rows of its partial buffer equal the oracle's partial
    sums over rows j in [0, 64), in the column-interleaved layout the reduce-scatter consumes; timing of the
    shard's sweep is printed (per-rank sweep time of the 8-GPU run)."""
    from sdk_amd.sharding import partial_tensor, scatter_plane_layout_index
    _need_hbm(240)
    o, cl, pp = _client(oracle_mod, C3, 801)
    p = sp.Params(C3)
    gpp = sp.PublicParameters.deserialize(p, pp)
    q = cl.generate_query(3000001, 802)
    t0 = time.time()
    expect = o.process_query_synth(pp, q, SEED)
    print("oracle: %.1f s for the C3 query" % (time.time() - t0))
    db = sp.Database(p).fill_synthetic(SEED)
    sp.paths_taken()
    got = sp.process_query(p, gpp, q, db)
    taken = sp.paths_taken()
    assert {"sweep_packed_persist", "from_sweep4", "fold_fused", "pipelined_fold_overlap"} <= taken, taken
    assert got == expect
    del db
    gc.collect()
    # ---- one row shard of 8
    G, planes, num_per = 8, 4, 1 << 13
    shard = sp.Database(p, 0, G).fill_synthetic(SEED)
    assert shard.device_bytes() == 28 * 2**30
    v_reg, _ = o.expand_query(pp, q)
    run = sp.QueryRun(p, gpp, q, db=shard)
    assert "expand_pruned" in sp.paths_taken()
    for pl in range(planes):
        run.sweep_scatter_plane(shard, G, pl)
    run.sync()
    assert {"sweep_packed_persist", "scatter_out"} <= sp.paths_taken()
    part = partial_tensor(run).cpu().numpy()
    rng = np.random.default_rng(803)
    ii = np.arange(num_per)
    for _ in range(6):
        pl, z = int(rng.integers(planes)), int(rng.integers(N))
        want = o.sweep_synth_row(SEED, pl, z, v_reg, 0, 512 // G)          # (num_per, 4): n0_0 n0_1 n1_0 n1_1
        for which, (r, crt) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
            idx = scatter_plane_layout_index(num_per, G, pl, r, crt, z, ii)
            assert (part[idx].astype(np.uint64) == want[:, which]).all(), (pl, z, r, crt)
    ms = run.bench_sweep(shard, 3, per_plane=1)
    print("C3 row shard 0/8: %.3f ms per plane launch, %.3f ms per query sweep" % (ms, ms * planes))
    run.free()


@pytest.mark.parametrize("G", [8, 4, 2])
def test_c2_row_sharded_flow_on_one_gpu(sp, oracle_mod, G):
    """The N > 1 benchmark path at the headline size, all G ranks on this one GPU: G row shards of the C2 database
    (56 GiB together), G workspaces, 2 G streams, sp_process_query_sharded per rank (host threads) with the
    loopback transport standing in for ncclReduceScatter / ncclAllGather -- the real stream structure (per-plane
    exchange overlapping the next plane's sweep, local fold, all-gather, final levels on rank 0).  Rank 0's
    response must equal the oracle's for the unsharded database (== the single-GPU response checked above)."""
    from sdk_amd.sharding import LoopbackWorld
    _need_hbm(66)
    o, cl, pp = _client(oracle_mod, C2, 501)
    p = sp.Params(C2)
    gpp = sp.PublicParameters.deserialize(p, pp)
    q = cl.generate_query(777777, 901)
    expect = o.process_query_synth(pp, q, SEED)
    shards = [sp.Database(p, s, G).fill_synthetic(SEED) for s in range(G)]
    world = LoopbackWorld(G)

    def rank_main(r):
        sp.lib().sp_set_device(0)
        sp.paths_taken()
        out = world.comm(r).process_query(p, gpp, q, shards[r])
        return out, sp.paths_taken(), world.comm(r).timings()
    res = world.run(rank_main)
    assert res[0][0] == expect
    for r in range(G):
        assert {"sweep_packed_persist", "scatter_out", "expand_pruned", "from_sweep4", "custom_transport"} <= res[r][1], res[r][1]
    print("G=%d per-rank ms (all ranks sharing one GPU): sweep+exchange %s" % (G, ["%.2f" % t[0] for _, _, t in res]))
    del shards, world
    gc.collect()
