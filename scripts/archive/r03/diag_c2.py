"""Diagnostic: small configurations first (as test_golden_vectors does), then C2; are the database words and the public
parameters intact before / after the first query's workspace is created, and is the first response right?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402
from conftest import C2, FAST, FAST56, SMALL_INST2  # noqa: E402
from sdk_amd.spiral import synth_word  # noqa: E402

SEED = 0x123456789
N = 2048


def small(cfg, idx):
    o = oracle.Params(cfg)
    cl = oracle.Client(o)
    pp = cl.generate_keys(80 + idx)
    q = cl.generate_query(idx, 81 + idx)
    item, db = o.generate_random_db_and_get_item(idx)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    return sp.process_query(p, gpp, q, gdb) == o.process_query(pp, q, db)


def db_ok(o, gdb):
    ok = True
    for (pl, z, ii) in ((0, 0, 0), (0, 1, 5), (1, 777, 1234), (2, 2047, 2047), (3, 2047, 2047), (3, 0, 0), (1, 1024, 1), (2, 5, 2000)):
        got = gdb.read_ref(pl, z, ii, 0, o.dim0)
        base = ((pl * N + z) * o.num_per + ii) * o.dim0
        exp = np.array([synth_word(SEED, base + k) for k in range(o.dim0)], dtype=np.uint64)
        ok &= bool((got == exp).all())
    return ok


if __name__ == "__main__":
    if os.environ.get("SMALL_FIRST", "1") == "1":
        print("small first:", [small(c, i) for c, i in ((FAST, 3), (FAST56, 300), (SMALL_INST2, 12), (dict(FAST, version=1), 7),
                                                        (dict(FAST, direct_upload=1), 100))], flush=True)
    o = oracle.Params(C2)
    cl = oracle.Client(o)
    pp = cl.generate_keys(501)
    q = cl.generate_query(0, 900)
    exp = o.process_query_synth(pp, q, SEED)
    flat = o.pp_deserialize_flat(pp)
    p = sp.Params(C2)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).fill_synthetic(SEED)
    print("before the first query: db words ok", db_ok(o, gdb), "pp ok", bool((gpp.export() == flat).all()), flush=True)
    got = sp.process_query(p, gpp, q, gdb)
    print("first query ok:", got == exp, flush=True)
    print("after the first query: db words ok", db_ok(o, gdb), "pp ok", bool((gpp.export() == flat).all()), flush=True)
    print("second query ok:", sp.process_query(p, gpp, q, gdb) == exp, flush=True)
    r = sp.QueryRun(p, gpp, q, db=gdb)
    r.sweep(gdb)
    print("stepwise ok:", r.finish() == exp, flush=True)
