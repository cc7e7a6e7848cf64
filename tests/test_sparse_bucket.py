"""lib/server's sparse caller: a SparseDb bucket served by the GPU library, byte-identical to the
restatement of lib/server's process_query over the same SparseDb (oracle/sparse_server.cpp: pruned expansion,
multiply_reg_by_sparse_database, fold with the all-zero shortcuts, pack, encode)."""
import numpy as np
import pytest

from conftest import FAST


def _fill(o, oracle_mod, n_items, seed, item_bytes):
    rng = np.random.default_rng(seed)
    sdb = oracle_mod.SparseDb(o)
    items = {}
    for idx in rng.choice(o.num_items, n_items, replace=False):
        items[int(idx)] = rng.integers(0, 256, item_bytes, dtype=np.uint8).tobytes()
        sdb.update_item_raw(int(idx), items[int(idx)])
    return sdb, items


def test_sparse_oracle_semantics(oracle_mod):
    """CPU: the sparse restatement decodes present items; its bytes differ from spiral-rs's dense process_query over the
    zero-filled database exactly when fold.rs:38-44's shortcuts fire, i.e. when a whole column of the bucket is empty
    (with every column populated they coincide)."""
    cfg = dict(FAST, nu_1=4, nu_2=2, db_item_size=256)
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(3)
    sdb, items = _fill(o, oracle_mod, 2, 1, 256)    # at most 2 of the 4 columns hold anything: shortcuts fire
    idx = next(iter(items))
    q = cl.generate_query(idx, 4)
    r = sdb.process_query(pp, q)
    assert cl.decode_response(r)[:256] == items[idx]
    assert r != o.process_query(pp, q, sdb.to_dense())
    full, items_full = _fill(o, oracle_mod, o.num_items, 2, 256)      # nothing absent: no shortcut can fire
    q = cl.generate_query(5, 6)
    assert full.process_query(pp, q) == o.process_query(pp, q, full.to_dense())
    assert full.polys() == 4 * o.num_items


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n_items", [(dict(FAST, nu_1=9, nu_2=7, db_item_size=256), 655),
                                         (dict(FAST, nu_1=6, nu_2=4, db_item_size=16384, instances=2, version=1), 40),
                                         (dict(FAST, nu_1=3, nu_2=2, db_item_size=256, t_gsw=7, t_conv=3, t_exp_left=5, t_exp_right=5, q2_bits=22), 32)],
                         ids=["2^16-items-1pct", "two-instances-pack-v1", "full-bucket-server-gadgets"])
def test_sparse_bucket_matches_lib_server(oracle_mod, cfg, n_items):
    import sdk_amd as sp
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(11)
    gpp = sp.PublicParameters.deserialize(p, pp)
    sdb, items = _fill(o, oracle_mod, n_items, 5, cfg["db_item_size"])
    gdb = sp.Database.sparse(p)
    for idx, data in items.items():
        gdb.update_item(idx, data)
    assert gdb.sparse_items() == n_items == len(items)
    present = list(items)[:3]
    absent = next(i for i in range(o.num_items) if i not in items) if n_items < o.num_items else None
    for k, idx in enumerate(present + ([absent] if absent is not None else [])):
        q = cl.generate_query(idx, 70 + k)
        sp.paths_taken()
        resp = sp.process_query(p, gpp, q, gdb)
        taken = sp.paths_taken()
        assert {"sweep_sparse", "fold_fused"} <= taken and not ({"sweep_packed_persist", "sweep_narrow", "sweep_wide"} & taken), taken
        assert resp == sdb.process_query(pp, q), idx
        if idx in items and cfg.get("t_gsw", 8) == 8:
            chunks = cfg.get("instances", 1) * 4
            size = cfg["db_item_size"]
            got = cl.decode_response(resp)
            per = size // chunks
            assert all(got[t * per:(t + 1) * per] == items[idx][t * per:(t + 1) * per] for t in range(chunks)), idx
    # an upsert of an existing item and a new item; the next query sees both (index rebuilt)
    idx = present[0]
    newdata = bytes(reversed(items[idx]))
    gdb.update_item(idx, newdata)
    sdb.update_item_raw(idx, newdata)
    if absent is not None:
        gdb.update_item(absent, b"\x07" * 16)
        sdb.update_item_raw(absent, b"\x07" * 16)
        assert gdb.sparse_items() == n_items + 1
    q = cl.generate_query(idx, 99)
    assert sp.process_query(p, gpp, q, gdb) == sdb.process_query(pp, q)
    # the split API and the batch entry point on a sparse bucket
    run = sp.QueryRun(p, gpp, q, db=gdb).sweep(gdb)
    assert run.finish() == sdb.process_query(pp, q)
    run.free()
    assert sp.process_query_batch(p, gpp, [q, q], gdb) == [sdb.process_query(pp, q)] * 2
    with pytest.raises(sp.SpiralError):
        sp.QueryRun(p, gpp, q).sweep(gdb)          # begun without the bucket: its expansion was not pruned for it
    with pytest.raises(sp.SpiralError):
        gdb.fill_synthetic(1)


@pytest.mark.gpu
def test_sparse_sweep_time_scales_with_occupancy(oracle_mod):
    """the first-dimension step of a sparse bucket costs time in proportion to the items present"""
    import sdk_amd as sp
    import bench
    cfg = dict(FAST, nu_1=9, nu_2=7, db_item_size=256)
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    q = bench.synthetic_wire_bytes(p.query_bytes(), 2)
    rng = np.random.default_rng(3)
    order = rng.permutation(1 << 16)
    gdb = sp.Database.sparse(p)
    times, filled = [], 0
    for target in (655, 2621, 10485):                 # 1 %, 4 %, 16 %
        for idx in order[filled:target]:
            gdb.update_item(int(idx), b"\x01\x02\x03")
        filled = target
        ms = []
        for _ in range(4):
            run = sp.QueryRun(p, pp, q, db=gdb).sweep(gdb)
            run.finish()
            ms.append(run.timings()[1])
            run.free()
        times.append(min(ms))
    print("sparse sweep ms at 1 / 4 / 16 %% occupancy: %s" % ["%.3f" % t for t in times])
    assert times[2] > 2.0 * times[0] and times[2] < 40 * times[0]
