"""GPU fuzz seed 204, case 45 (scripts/r06/call26.sh): one response differed from the oracle on gfx950, equal on the emulated device.
Repeats that exact case: fresh handles every repetition (as the fuzzer does), response against the oracle's.
Usage: python scripts/r06/repro_fuzz_204.py REPS MATERIALISE [same_handles]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

import oracle
import sdk_amd as sp

cfg = {"n": 2, "nu_1": 6, "nu_2": 3, "p": 256, "q2_bits": 16, "t_gsw": 4, "t_conv": 1, "t_exp_left": 14, "t_exp_right": 14,
       "instances": 2, "db_item_size": 16384, "version": 1, "direct_upload": 1}
ks, qs, idx = 714463203, 808286267, 449
reps, mat = int(sys.argv[1]), int(sys.argv[2])
same = len(sys.argv) > 3 and sys.argv[3] == "same"
qmode = len(sys.argv) > 3 and sys.argv[3] == "queries"      # same handles, a DIFFERENT query every repetition (the query's H2D path)
same = same or qmode
reload = len(sys.argv) > 3 and sys.argv[3] == "reload"        # same handles; the SAME database handle is loaded again every repetition
same = same or reload
pponly = len(sys.argv) > 3 and sys.argv[3] == "pponly"        # fresh PUBLIC PARAMETERS every repetition, everything else kept
dbonly = len(sys.argv) > 3 and sys.argv[3] == "dbonly"      # fresh DATABASE handle per repetition, everything else kept
o = oracle.Params(cfg)
cl = oracle.Client(o)
pp = cl.generate_keys(ks)
q = cl.generate_query(idx, qs)
item, db = o.generate_random_db_and_get_item(idx)
want = o.process_query(pp, q, db)
if qmode:
    qlist = [q] + [cl.generate_query((idx + 97 * k) % o.num_items, qs + k) for k in range(1, 6)]
    wlist = [want] + [o.process_query(pp, x, db) for x in qlist[1:]]
sp.lib().sp_debug_set(b"fold_neg_materialise", C.c_long(mat))
bad = 0
h = None
for r in range(reps):
    if h is not None and pponly:
        h = (h[0], None, h[2])
        h = (h[0], sp.PublicParameters.deserialize(h[0], pp), h[2])
    elif h is not None and dbonly:
        h = (h[0], h[1], None)
        h = (h[0], h[1], sp.Database(h[0]).load(db))
    elif h is None or not same:
        p = sp.Params(cfg)
        gpp = sp.PublicParameters.deserialize(p, pp)
        gdb = sp.Database(p).load(db)
        h = (p, gpp, gdb)
    p, gpp, gdb = h
    if reload and r:
        gdb.load(db)
    if qmode:
        q, want = qlist[r % 6], wlist[r % 6]
    got = sp.process_query(p, gpp, q, gdb)
    if qmode and got != want:
        which = [k for k in range(6) if got == wlist[k]]
        print("rep %d: response of query %d %s" % (r, r % 6, ("equals the response of query %d" % which[0]) if which else "matches none of the six"), flush=True)
        bad += 1
        continue
    if got != want:
        bad += 1
        a, b = np.frombuffer(got, dtype=np.uint8), np.frombuffer(want, dtype=np.uint8)
        d = np.nonzero(a != b)[0]
        print("rep %d DIFFERS: %d of %d bytes, first %s last %s" % (r, len(d), len(a), d[:6], d[-3:]), flush=True)
        # the same handles again: is it the handles' state (database / parameters) or the run?
        again = sp.process_query(p, gpp, q, gdb)
        print("   same handles again: %s" % ("equal" if again == want else "differs (%s)" % ("same bytes" if again == got else "other bytes")), flush=True)
        # which handle holds wrong data?  the database read back in the reference layout against what was loaded; the public
        # parameters exported against a fresh deserialisation's export
        if (dbonly or reload or pponly) and bad > 3:
            continue
        dbh = np.asarray(db, dtype=np.uint64).reshape(o.get("instances") * 4, 2048, 8, 64)
        wrong = []
        for pl in range(dbh.shape[0]):
            for z in range(2048):
                for ii in range(8):
                    w = gdb.read_ref(pl, z, ii, 0, 64)
                    if not (w == dbh[pl, z, ii]).all():
                        nz = int((w != dbh[pl, z, ii]).sum())
                        wrong.append((pl, z, ii, nz, int((w[w != dbh[pl, z, ii]] == 0).sum())))
        print("   database: %d of %d (plane, z, column) rows differ from what was loaded" % (len(wrong), dbh.shape[0] * 2048 * 8), flush=True)
        if wrong:
            pls = sorted({w[0] for w in wrong})
            zs = [w[1] for w in wrong]
            print("      planes %s, z %d .. %d, words differing %d, of them read back as ZERO %d; first %s last %s" % (
                pls, min(zs), max(zs), sum(w[3] for w in wrong), sum(w[4] for w in wrong), wrong[0], wrong[-1]), flush=True)
        e0 = gpp.export()
        e1 = sp.PublicParameters.deserialize(p, pp).export()
        dpp = np.nonzero(e0 != e1)[0]
        print("   public parameters: %d of %d exported words differ from a fresh deserialisation%s" % (
            len(dpp), len(e0), "" if not len(dpp) else "; first %d last %d, zero in the loaded copy: %d" % (dpp[0], dpp[-1], int((e0[dpp] == 0).sum()))), flush=True)
print("materialise=%d same_handles=%s dbonly=%s env=%s: %d of %d repetitions differ" % (mat, same, dbonly, {k: v for k, v in os.environ.items() if k.startswith("SPIRAL_")}, bad, reps))
