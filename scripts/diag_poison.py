"""Diagnostic: run small configurations on FRESH handles with every new device buffer poisoned (poison_ws), find the
buffer whose uninitialised contents reach the response (poison_skip bisection), tell a race from a logic error
(debug_sync)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402
from conftest import FAST, FAST56, SERVER_DEFAULT, SMALL_INST2  # noqa: E402

_cache = {}


def case(name, cfg, idx):
    if name not in _cache:
        o = oracle.Params(cfg)
        cl = oracle.Client(o)
        pp = cl.generate_keys(80 + idx)
        q = cl.generate_query(idx, 81 + idx)
        item, db = o.generate_random_db_and_get_item(idx)
        _cache[name] = (cfg, pp, q, db, o.process_query(pp, q, db))
    return _cache[name]


def fresh(name, cfg, idx, n_queries=1):
    cfg, pp, q, db, exp = case(name, cfg, idx)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    return [sp.process_query(p, gpp, q, gdb) == exp for _ in range(n_queries)]


def setd(name, v):
    sp.lib().sp_debug_set(name.encode(), C.c_long(v))


CASES = [("fast", FAST, 99), ("nu2_1", dict(FAST, nu_2=1), 99), ("nu2_3", dict(FAST, nu_2=3), 99), ("fast56", FAST56, 301),
         ("inst2", SMALL_INST2, 123), ("nu2_0", dict(FAST, nu_2=0, db_item_size=8192), 17),
         ("v1", dict(FAST, version=1), 200), ("direct", dict(FAST, direct_upload=1), 100),
         ("srvdef", dict(SERVER_DEFAULT, nu_1=6), 1500), ("nu2_5", dict(FAST, nu_2=5), 7), ("nu1_7", dict(FAST, nu_1=7, nu_2=1), 7)]

if __name__ == "__main__":
    print("baseline:", {n: fresh(n, c, i, 2) for n, c, i in CASES[:3]}, flush=True)
    for b in (255, 0xA5):
        setd("poison_ws", b)
        setd("poison_skip", -1)
        res = {}
        for n, c, i in CASES:
            try:
                res[n] = fresh(n, c, i, 2)
            except Exception as e:  # noqa: BLE001
                res[n] = "ERR %s" % e
        print("poison 0x%02x:" % b, res, flush=True)
    bad = [(n, c, i) for n, c, i in CASES if res.get(n) != [True, True]]
    print("failing under poison:", [n for n, _, _ in bad], flush=True)
    for n, c, i in bad[:3]:
        setd("debug_sync", 1)
        print("  %s with debug_sync:" % n, fresh(n, c, i, 2), flush=True)
        setd("debug_sync", 0)
        good_k = []
        for k in range(0, 90):
            setd("poison_skip", k)
            try:
                if fresh(n, c, i, 1) == [True]:
                    good_k.append(k)
            except Exception as e:  # noqa: BLE001
                print("   k", k, "ERR", e)
        print("  %s: correct when allocation #k is zero-filled instead, k in" % n, good_k, flush=True)
        os.environ["SPIRAL_ALLOC_DEBUG"] = "1"
        setd("poison_skip", -1)
        fresh(n, c, i, 1)
        os.environ.pop("SPIRAL_ALLOC_DEBUG")
    setd("poison_ws", 0)
