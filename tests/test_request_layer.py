"""The request layer around the answer path (lib/server/src/bin/server.rs:21-164 without HTTP):
per-uuid public-parameter cache, /setup and /private-read framing (JSON list of base64 strings), batch scheduling of
the query list.  CPU part: framing and error behaviour that need no device; GPU part: end to end against the oracle."""
import base64
import json

import numpy as np
import pytest

from conftest import FAST


def test_framing_and_errors_without_a_device():
    """request parsing happens before any device work: malformed bodies, bad lengths and unknown uuids are rejected
    with the reference's outcomes (assert -> SP_E_ARG, Error::NotFound -> SP_E_NOTFOUND)"""
    import sdk_amd as sp
    import ctypes as C
    p = sp.Params(FAST)
    L = sp.lib()
    L.sp_server_create.restype = C.c_void_p
    # a null database handle is refused
    assert not L.sp_server_create(C.c_void_p(p.h), None)


@pytest.mark.gpu
def test_setup_private_read_roundtrip(oracle_mod):
    import sdk_amd as sp
    cfg = dict(FAST, nu_2=7, db_item_size=256)
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(7)
    gdb = sp.Database(p).load(db)
    srv = sp.Server(p, gdb)
    clients = []
    for k in range(3):                       # three clients with their own keys
        cl = oracle_mod.Client(o)
        pp = cl.generate_keys(100 + k)
        body = json.dumps(base64.b64encode(pp).decode())          # what the SDKs POST to /setup
        resp = json.loads(srv.setup_json(body))
        assert set(resp) == {"uuid"} and len(resp["uuid"]) == 36 and resp["uuid"][14] == "4"
        clients.append((cl, pp, resp["uuid"]))
    assert srv.clients() == 3
    # 11 queries from the three clients in ONE /private-read body (groups of 8 + 3 in the batch scheduler)
    reqs, expect = [], []
    for i in range(11):
        cl, pp, uuid = clients[i % 3]
        idx = (911 * i + 7) % o.num_items
        q = cl.generate_query(idx, 500 + i)
        reqs.append(uuid.encode() + q)
        expect.append(o.process_query(pp, q, db))
    body = json.dumps([base64.b64encode(r).decode() for r in reqs])
    out = json.loads(srv.private_read_json(body))
    assert [base64.b64decode(x) for x in out] == expect
    assert srv.private_read_json(body) == json.dumps([base64.b64encode(e).decode() for e in expect], separators=(",", ":"))
    # decoded-bytes entry point, single request; the first client's answer decodes to the database item
    cl, pp, uuid = clients[0]
    q = cl.generate_query(7, 900)
    (resp,) = srv.private_read([uuid.encode() + q])
    assert resp == o.process_query(pp, q, db) and cl.decode_response(resp) == o.item_to_vec(item)
    assert srv.private_read([]) == [] and srv.private_read_json("[]") == "[]"
    # errors: unknown uuid -> NotFound (HTTP 404 in the reference), bad length -> SpiralError (assert_eq! in the reference)
    with pytest.raises(sp.NotFound):
        srv.private_read([b"00000000-0000-4000-8000-000000000000" + q])
    with pytest.raises(sp.SpiralError):
        srv.private_read([uuid.encode() + q[:-1]])
    with pytest.raises(sp.SpiralError):
        srv.private_read_json('["not base64!"]')
    with pytest.raises(sp.SpiralError):
        srv.private_read_json('{"a": 1}')
    with pytest.raises(sp.SpiralError):
        srv.setup(pp[:-8])
    srv.forget(uuid)
    with pytest.raises(sp.NotFound):
        srv.private_read([uuid.encode() + q])
    assert srv.clients() == 2


@pytest.mark.gpu
def test_private_read_direct_upload_params(oracle_mod):
    """params without query expansion: the public parameters travel with every query (bin/server.rs:123-138)"""
    import sdk_amd as sp
    cfg = {"n": 2, "nu_1": 4, "nu_2": 7, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 256, "direct_upload": 1}
    o = oracle_mod.Params(cfg)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(3)
    gdb = sp.Database(p).load(db)
    srv = sp.Server(p, gdb)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(5)
    qs = [cl.generate_query(i, 50 + i) for i in (3, 200)]
    out = srv.private_read([pp + q for q in qs])
    assert out == [o.process_query(pp, q, db) for q in qs]
    assert cl.decode_response(out[0]) == o.item_to_vec(item)
    assert srv.clients() == 0
