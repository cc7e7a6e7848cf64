"""Diagnostic: every device buffer between two poisoned guard regions (guard_ws): out-of-bounds reads become
deterministic, out-of-bounds writes are reported at release; poison_skip bisects the buffer being over-read."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402
from conftest import C1, FAST, FAST56, SERVER_DEFAULT, SMALL_INST2  # noqa: E402

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
    return "".join("T" if sp.process_query(p, gpp, q, gdb) == exp else "F" for _ in range(n_queries))


def setd(name, v):
    sp.lib().sp_debug_set(name.encode(), C.c_long(v))


CASES = [("fast", FAST, 99), ("nu2_1", dict(FAST, nu_2=1), 99), ("nu2_3", dict(FAST, nu_2=3), 99), ("fast56", FAST56, 301),
         ("inst2", SMALL_INST2, 123), ("nu2_0", dict(FAST, nu_2=0, db_item_size=8192), 17),
         ("v1", dict(FAST, version=1), 200), ("direct", dict(FAST, direct_upload=1), 100),
         ("srvdef", dict(SERVER_DEFAULT, nu_1=6), 1500), ("nu2_5", dict(FAST, nu_2=5), 7), ("nu1_7", dict(FAST, nu_1=7, nu_2=1), 7),
         ("nu2_8", dict(FAST, nu_2=8), 7)]

if __name__ == "__main__":
    only = sys.argv[1:]
    setd("guard_ws", 1 << 20)
    res = {}
    for b in (0xA5, 0xFF):
        setd("poison_ws", b)
        setd("poison_skip", -1)
        for n, c, i in CASES:
            if only and n not in only:
                continue
            try:
                res[n] = res.get(n, "") + fresh(n, c, i, 2)
            except Exception as e:  # noqa: BLE001
                res[n] = "ERR %s" % e
        print("guards 0x%02x:" % b, res, flush=True)
    bad = [(n, c, i) for n, c, i in CASES if "F" in res.get(n, "") or "ERR" in res.get(n, "")]
    print("failing with poisoned guards:", [n for n, _, _ in bad], flush=True)
    setd("poison_ws", 0xA5)
    for n, c, i in bad[:4]:
        good_k = []
        for k in range(0, 60):
            setd("poison_skip", k)
            try:
                if fresh(n, c, i, 1) == "T":
                    good_k.append(k)
            except Exception as e:  # noqa: BLE001
                print("   k", k, "ERR", e)
        print("  %s: correct when the guards of allocation #k are zero, k in" % n, good_k, flush=True)
        os.environ["SPIRAL_ALLOC_DEBUG"] = "1"
        setd("poison_skip", -1)
        fresh(n, c, i, 1)
        os.environ.pop("SPIRAL_ALLOC_DEBUG")
    setd("guard_ws", 0)
    setd("poison_ws", 0)
