"""Diagnostic: where does process_query diverge from the oracle for FAST with nu_2 = 1 (num_per = 2)?"""
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


def session(cfg, idx, seed):
    o = oracle.Params(cfg)
    cl = oracle.Client(o)
    return o, cl, cl.generate_keys(seed), cl.generate_query(idx, seed + 1)


def run(cfg, idx, label):
    o, cl, pp, q = session(cfg, idx, 80 + idx)
    item, db = o.generate_random_db_and_get_item(idx)
    exp = o.process_query(pp, q, db)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    got = [sp.process_query(p, gpp, q, gdb) for _ in range(3)]
    sp.paths_taken()
    sp.process_query(p, gpp, q, gdb)
    print(label, "process_query == oracle:", [g == exp for g in got], "paths:", sorted(sp.paths_taken()), flush=True)
    return o, cl, pp, q, db, exp, p, gpp, gdb


def tunable_sweep(cfg, idx):
    o, cl, pp, q = session(cfg, idx, 80 + idx)
    item, db = o.generate_random_db_and_get_item(idx)
    exp = o.process_query(pp, q, db)
    for name, val in [("expand_split", 0), ("expand_split", 1), ("narrow1", 1), ("from_sweep1", 1), ("fold_variant", 3),
                      ("fused_min_pairs", 1), ("pipeline", 0), ("db_contiguous", 0), ("sweep_variant", 1)]:
        sp.lib().sp_debug_set(name.encode(), C.c_long(val))
        try:
            p = sp.Params(cfg)
            gpp = sp.PublicParameters.deserialize(p, pp)
            gdb = sp.Database(p).load(db)
            ok = sp.process_query(p, gpp, q, gdb) == exp
        except Exception as e:  # noqa: BLE001
            ok = "ERR %s" % e
        print("  tunable %s=%d ->" % (name, val), ok, flush=True)
        sp.lib().sp_debug_set(name.encode(), C.c_long(-1 if name == "expand_split" else 0))
    # restore defaults that are not 0
    for name, val in [("fold_variant", 5), ("fused_min_pairs", 256), ("pipeline", 1), ("db_contiguous", 1)]:
        sp.lib().sp_debug_set(name.encode(), C.c_long(val))


def stages(o, cl, pp, q, db, exp, p, gpp, gdb):
    import torch
    from sdk_amd import sharding
    v_reg, v_fold = sp.expand_query(p, gpp, q)
    o_reg, o_fold = o.expand_query(pp, q)
    print("  expand_query v_reg:", bool((v_reg == o_reg).all()), " v_fold:", bool((v_fold == o_fold).all()), flush=True)
    v_neg = o.get_v_folding_neg(o_fold)
    print("  folding_neg:", bool((sp.get_v_folding_neg(p, o_fold) == v_neg).all()), flush=True)
    slice_words = o.dim0 * o.num_per * N
    planes = o.instances * o.n * o.n
    outs = []
    for t in range(planes):
        oc = o.multiply_reg_by_database(db[t * slice_words:(t + 1) * slice_words], o_reg)
        og = sp.multiply_reg_by_database(p, db[t * slice_words:(t + 1) * slice_words], o_reg)
        print("  stage multiply_reg_by_database plane", t, bool((og == oc).all()), flush=True)
        outs.append(oc)
    # the resident database through the stepwise entry points
    r = sp.QueryRun(p, gpp, q, gdb)
    r.sweep(gdb)
    r.sync()
    part = sharding.partial_tensor(r).cpu().numpy().view(np.uint32).reshape(planes, 2, 2, N, o.num_per)
    for t in range(planes):
        oc = outs[t].reshape(o.num_per, 2, 2, N)  # [ii][r][crt][z]
        g = np.transpose(part[t], (3, 0, 1, 2)).astype(np.uint64)
        bad = np.argwhere(g != oc)
        print("  resident sweep plane", t, "equal:", len(bad) == 0, "mismatches:", len(bad), bad[:3].tolist(), flush=True)
    resp = r.finish()
    print("  stepwise finish == oracle:", resp == exp, flush=True)
    for t in range(planes):
        raw = o.from_ntt(outs[t])
        graw = sp.from_ntt(p, outs[t])
        f_cpu = o.fold_ciphertexts(raw, o_fold, v_neg)[:2 * N]
        print("  plane", t, "from_ntt:", bool((graw == raw).all()),
              " fold literal:", bool((sp.fold_ciphertexts(p, raw, o_fold, v_neg)[:2 * N] == f_cpu).all()),
              " fold default:", bool((sp.fold_ciphertexts_fused(p, raw, o_fold, 256)[:2 * N] == f_cpu).all()),
              " fold fused:", bool((sp.fold_ciphertexts_fused(p, raw, o_fold, 1)[:2 * N] == f_cpu).all()), flush=True)


if __name__ == "__main__":
    print("device count", sp.lib().sp_device_count(), flush=True)
    run(FAST, 99, "FAST nu_2=2 idx 99:")
    st = run(dict(FAST, nu_2=1), 99, "nu2_1 idx 99:")
    run(dict(FAST, nu_2=1), 5, "nu2_1 idx 5:")
    run(dict(FAST, nu_2=3), 99, "nu2_3 idx 99:")
    try:
        stages(*st)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
    tunable_sweep(dict(FAST, nu_2=1), 99)
