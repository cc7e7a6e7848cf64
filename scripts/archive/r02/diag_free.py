"""Diagnostic: handles of the previous iteration are released between the creation of the new handles and the first
query (the pattern that failed in diag_race).  On a failing handle: is the device-side computation right?"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402
from conftest import FAST  # noqa: E402

N = 2048
_cache = {}


def case(name, cfg, idx):
    if name not in _cache:
        o = oracle.Params(cfg)
        cl = oracle.Client(o)
        pp = cl.generate_keys(80 + idx)
        q = cl.generate_query(idx, 81 + idx)
        item, db = o.generate_random_db_and_get_item(idx)
        o_reg, o_fold = o.expand_query(pp, q)
        sw = o.dim0 * o.num_per * N
        planes = o.instances * o.n * o.n
        outs = [o.multiply_reg_by_database(db[t * sw:(t + 1) * sw], o_reg) for t in range(planes)]
        _cache[name] = (o, cfg, pp, q, db, o.process_query(pp, q, db), o_reg, o_fold, outs)
    return _cache[name]


def handles(name, cfg, idx):
    o, cfg, pp, q, db = case(name, cfg, idx)[:5]
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    return p, gpp, gdb


def resident(p, gpp):
    out = (C.c_uint64 * 24)()
    rc = sp.lib().sp_debug_resident_check(C.c_void_p(p.h), C.c_void_p(gpp.h), out, 24)
    names = ["tw", "neg1", "gadget", "lists", "pp.all", "pp.pack_cat"]
    res = []
    for i, n in enumerate(names):
        k, c, k2, h = out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]
        flag = ""
        if k != c:
            flag += " KERNEL!=COPY"
        if k2 != k:
            flag += " changes-after-cache-sync"
        if h and h != c:
            flag += " COPY!=HOST"
        res.append("%s:%s" % (n, flag or "ok"))
    return "rc=%d " % rc + " ".join(res)


def probe_failed(name, cfg, idx, p, gpp, gdb, got):
    from sdk_amd import sharding
    o, cfg, pp, q, db, exp, o_reg, o_fold, outs = case(name, cfg, idx)
    planes = len(outs)
    print("      resident:", resident(p, gpp), flush=True)
    print("      pp export == oracle:", bool((gpp.export() == o.pp_deserialize_flat(pp)).all()), flush=True)
    print("      query again:", sp.process_query(p, gpp, q, gdb) == exp, flush=True)
    g = np.frombuffer(got, dtype=np.uint8)
    e = np.frombuffer(exp, dtype=np.uint8)
    info = "bytes differing %d/%d, zero bytes %d" % (int((g != e).sum()), g.size, int((g == 0).sum()))
    v_reg, v_fold = sp.expand_query(p, gpp, q)
    info += " | expand R=%s G=%s" % (bool((v_reg == o_reg).all()), bool((v_fold == o_fold).all()))
    r = sp.QueryRun(p, gpp, q, gdb)
    r.sweep(gdb)
    r.sync()
    part = sharding.partial_tensor(r).cpu().numpy().view(np.uint32).reshape(planes, 2, 2, N, o.num_per)
    sw_ok = all((np.transpose(part[t], (3, 0, 1, 2)).astype(np.uint64) == outs[t].reshape(o.num_per, 2, 2, N)).all()
                for t in range(planes))
    fin = r.finish()
    info += " | stepwise sweep ok=%s finish ok=%s finish==failing bytes %s" % (sw_ok, fin == exp, fin == got)
    return info


if __name__ == "__main__":
    for kv in os.environ.get("DIAG_TUNABLES", "").split(","):
        if "=" in kv:
            k, v = kv.split("=")
            sp.lib().sp_debug_set(k.encode(), C.c_long(int(v)))
    CASES = [("nu2_3", dict(FAST, nu_2=3), 99), ("nu2_1", dict(FAST, nu_2=1), 99), ("fast", FAST, 99)]
    reported = 0
    for rnd in range(2):
        for name, cfg, idx in CASES:
            exp = case(name, cfg, idx)[5]
            res = []
            for _ in range(10):
                p, gpp, gdb = handles(name, cfg, idx)      # the previous p, gpp, gdb are released HERE
                got = [sp.process_query(p, gpp, case(name, cfg, idx)[3], gdb) for _ in range(3)]
                res.append("".join("T" if g == exp else "F" for g in got))
                if "F" in res[-1] and reported < 3:
                    reported += 1
                    print("   failing handle:", name, res[-1], probe_failed(name, cfg, idx, p, gpp, gdb, got[res[-1].index("F")]), flush=True)
            print(rnd, name, res, "| last handle resident:", resident(p, gpp), flush=True)
