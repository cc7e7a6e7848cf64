"""Diagnostic: drive the library into the state in which a small configuration fails on every query of a handle
(many handles of one configuration, then handles of another), and report which stage's output differs first."""
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
        _cache[name] = (o, cfg, pp, q, db, o.process_query(pp, q, db), o_reg, o_fold, outs, o.pp_deserialize_flat(pp))
    return _cache[name]


def probe(name, cfg, idx):
    from sdk_amd import sharding
    o, cfg, pp, q, db, exp, o_reg, o_fold, outs, pp_flat = case(name, cfg, idx)
    planes = len(outs)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    s = "Q" if sp.process_query(p, gpp, q, gdb) == exp else "q"
    s += "P" if (gpp.export() == pp_flat).all() else "p"
    ok = True
    for (pl, z, ii) in ((0, 0, 0), (1, 5, 1), (planes - 1, N - 1, o.num_per - 1)):
        ref = db[pl * N * o.num_per * o.dim0 + (z * o.num_per + ii) * o.dim0:][:o.dim0]
        ok &= bool((gdb.read_ref(pl, z, ii, 0, o.dim0) == ref).all())
    s += "D" if ok else "d"
    v_reg, v_fold = sp.expand_query(p, gpp, q)
    s += ("R" if (v_reg == o_reg).all() else "r") + ("G" if (v_fold == o_fold).all() else "g")
    r = sp.QueryRun(p, gpp, q, gdb)
    r.sweep(gdb)
    r.sync()
    part = sharding.partial_tensor(r).cpu().numpy().view(np.uint32).reshape(planes, 2, 2, N, o.num_per)
    sw_ok = all((np.transpose(part[t], (3, 0, 1, 2)).astype(np.uint64) == outs[t].reshape(o.num_per, 2, 2, N)).all()
                for t in range(planes))
    s += "S" if sw_ok else "s"
    s += "F" if r.finish() == exp else "f"
    s += "Q" if sp.process_query(p, gpp, q, gdb) == exp else "q"
    return s, (p, gpp, gdb)


if __name__ == "__main__":
    A = ("nu2_3", dict(FAST, nu_2=3), 99)
    B = ("nu2_1", dict(FAST, nu_2=1), 99)
    Cc = ("fast", FAST, 99)
    print("legend: Q process_query, P pp export, D db words, R v_reg, G v_folding, S sweep output, F stepwise finish, Q again; lower case = differs")
    keep = None
    for rnd in range(2):
        for name, cfg, idx in (A, B, Cc):
            out = []
            for _ in range(8):
                s, keep = probe(name, cfg, idx)   # the previous handle set is released after the new one exists
                out.append(s)
            print(rnd, name, out, flush=True)
