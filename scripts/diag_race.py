"""Diagnostic: failure statistics of small configurations on fresh handles, which stage the corruption enters, and which
synchronisation placement (debug_sync) removes it."""
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


def setd(name, v):
    sp.lib().sp_debug_set(name.encode(), C.c_long(v))


def handles(name, cfg, idx):
    o, cfg, pp, q, db, exp, o_reg, o_fold, outs = case(name, cfg, idx)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    return p, gpp, gdb


def pattern(name, cfg, idx, reps, nq=3):
    o, cfg, pp, q, db, exp = case(name, cfg, idx)[:6]
    res = []
    for _ in range(reps):
        p, gpp, gdb = handles(name, cfg, idx)
        res.append("".join("T" if sp.process_query(p, gpp, q, gdb) == exp else "F" for _ in range(nq)))
    return res


def stepwise(name, cfg, idx, reps):
    from sdk_amd import sharding
    o, cfg, pp, q, db, exp, o_reg, o_fold, outs = case(name, cfg, idx)
    planes = len(outs)
    res = []
    for _ in range(reps):
        p, gpp, gdb = handles(name, cfg, idx)
        r = sp.QueryRun(p, gpp, q, gdb)
        r.sweep(gdb)
        r.sync()
        part = sharding.partial_tensor(r).cpu().numpy().view(np.uint32).reshape(planes, 2, 2, N, o.num_per)
        sw_ok = all((np.transpose(part[t], (3, 0, 1, 2)).astype(np.uint64) == outs[t].reshape(o.num_per, 2, 2, N)).all()
                    for t in range(planes))
        fin_ok = r.finish() == exp
        res.append(("S" if sw_ok else "s") + ("F" if fin_ok else "f"))
    return res


def expand_only(name, cfg, idx, reps):
    o, cfg, pp, q, db, exp, o_reg, o_fold, outs = case(name, cfg, idx)
    res = []
    for _ in range(reps):
        p = sp.Params(cfg)
        gpp = sp.PublicParameters.deserialize(p, pp)
        v_reg, v_fold = sp.expand_query(p, gpp, q)
        res.append(("R" if (v_reg == o_reg).all() else "r") + ("G" if (v_fold == o_fold).all() else "g"))
    return res


if __name__ == "__main__":
    CASES = [("nu2_3", dict(FAST, nu_2=3), 99), ("nu2_1", dict(FAST, nu_2=1), 99), ("fast", FAST, 99)]
    for n, c, i in CASES:
        print(n, "fresh handles x 3 queries:", pattern(n, c, i, 12), flush=True)
    n, c, i = CASES[0]
    print(n, "stepwise (S/s sweep output ok/bad, F/f final ok/bad):", stepwise(n, c, i, 12), flush=True)
    print(n, "expand_query stage export (R/r v_reg, G/g v_folding):", expand_only(n, c, i, 12), flush=True)
    for ds in (1, 2, 11, 12, 13):
        setd("debug_sync", ds)
        print(n, "debug_sync", ds, pattern(n, c, i, 12, 1), flush=True)
    setd("debug_sync", 0)
    for name, v in (("narrow1", 1), ("from_sweep1", 1), ("fused_min_pairs", 1)):
        setd(name, v)
        print(n, name, v, pattern(n, c, i, 12, 1), flush=True)
        setd(name, 0 if name != "fused_min_pairs" else 256)
    os.environ["HIP_LAUNCH_BLOCKING"] = "1"
    print(n, "again default:", pattern(n, c, i, 12, 1), flush=True)
