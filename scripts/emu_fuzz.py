"""Differential fuzzing of the library's sources on the emulated device against the oracle (CPU only, test infrastructure).

Random VALID parameter sets (gadget widths, moduli sizes, instances, pack version, item sizes, nu_1 <= 7, nu_2 <= 8), random
item index and key / query seeds; response bytes of sp_process_query (and, on PACKED shapes, of a list through
sp_process_query_batch) must equal the oracle's.  Usage:
    SPIRAL_HIP_LIB=tests/emu/_build/libspiral_emu.so python scripts/emu_fuzz.py [--seed S] [--minutes M]
    python scripts/emu_fuzz.py --device [--seed S] [--minutes M]      (the same cases against the real library on a GPU box)
Prints one line per case; a mismatch prints the configuration as JSON (paste it into tests/test_gpu_parity.py::_FUZZ) and exits 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def valid(c):
    dim0, right = 1 << c["nu_1"], c["t_gsw"] * c["nu_2"]
    g = max(1, int(np.ceil(np.log2(right + dim0))))
    return 2 * max(dim0, right) <= (1 << g) and (c.get("version", 0) == 0 or c["n"] == 2)


def draw(rng):
    while True:
        c = dict(n=2, nu_1=int(rng.integers(2, 8)), nu_2=int(rng.integers(0, 9)), p=int(rng.choice([4, 16, 64, 256, 256, 256])),
                 q2_bits=int(rng.integers(14, 29)), t_gsw=int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 10, 14, 28])),
                 t_conv=int(rng.choice([1, 2, 3, 4, 5, 7, 14, 28])), t_exp_left=int(rng.choice([2, 3, 4, 5, 6, 7, 8, 14, 16, 28])),
                 t_exp_right=int(rng.choice([2, 4, 5, 8, 9, 14, 19, 28, 56])), instances=int(rng.choice([1, 1, 1, 2, 3])),
                 db_item_size=int(rng.choice([256, 600, 1000, 2048, 3000, 4096, 5000, 8192, 16384])))
        if rng.random() < 0.2:
            c["version"] = 1
        if rng.random() < 0.1:
            c["direct_upload"] = 1
        if c["nu_1"] + c["nu_2"] > 13:
            continue
        if valid(c):
            return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10)
    ap.add_argument("--device", action="store_true", help="the real library on a real GPU (gfx950) instead of the emulated build")
    a = ap.parse_args()
    import oracle
    import sdk_amd as sp
    if a.device:
        assert not hasattr(sp.lib(), "sp_emulated_device_marker"), "--device is for the real library"
    else:
        assert hasattr(sp.lib(), "sp_emulated_device_marker"), "set SPIRAL_HIP_LIB to the emulated build"
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + 60 * a.minutes
    n = 0
    while time.time() < t_end:
        cfg = draw(rng)
        try:
            o = oracle.Params(cfg)
        except Exception as e:  # a parameter set the reference would reject as well
            continue
        if o.num_items < 1 or cfg["db_item_size"] * 8 > o.get("instances") * o.get("n") ** 2 * 2048 * int(np.log2(cfg["p"])):
            continue
        t0 = time.time()
        try:
            p = sp.Params(cfg)        # the library's own validation first: what it rejects the reference cannot run either
        except sp.SpiralError:
            continue
        print("     try  %s" % json.dumps(cfg), flush=True)   # (a parameter set the oracle aborts on is the last line then)
        cl = oracle.Client(o)
        ks, qs = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30))
        idx = int(rng.integers(0, o.num_items))
        wire = ""
        try:
            pp = cl.generate_keys(ks)
            q = cl.generate_query(idx, qs)
            if rng.random() < 0.3:
                # arbitrary bit patterns in the wire formats (coefficients >= Q, up to 2^64 - 1): process_query is a function of
                # the bytes whatever a client would have sent, and the two sides must still agree
                pp = rng.integers(0, 256, len(pp), dtype=np.uint8).tobytes()
                q = rng.integers(0, 256, len(q), dtype=np.uint8).tobytes()
                wire = " random-wire"
            item, db = o.generate_random_db_and_get_item(idx)
            want = o.process_query(pp, q, db)
        except Exception as e:
            print("oracle rejects", json.dumps(cfg), repr(e)[:80], flush=True)
            continue
        gpp = sp.PublicParameters.deserialize(p, pp)
        gdb = sp.Database(p).load(db)
        sp.paths_taken()
        got = sp.process_query(p, gpp, q, gdb)
        paths = sorted(sp.paths_taken())
        ok = got == want
        extra = wire
        if ok and rng.random() < (0.8 if cfg["nu_2"] >= 7 else 0.3):
            # lists: DISTINCT queries, and on PACKED shapes (nu_2 >= 7) often more than 8 of them -- r06: a batched group's expansions
            # run as shared launches with per-query buffers behind byte offsets (run_begin_group), two query tiles from 9 queries
            B = int(rng.integers(2, 12 if cfg["nu_2"] >= 7 else 6))
            idxs = [int(rng.integers(0, o.num_items)) for _ in range(B)]
            lst = [q] + ([cl.generate_query(i, qs + 7 + k) for k, i in enumerate(idxs[1:])] if not wire else [q] * (B - 1))
            wants = [want] + ([o.process_query(pp, x, db) for x in lst[1:]] if not wire else [want] * (B - 1))
            sp.paths_taken()
            outs = sp.process_query_batch(p, gpp, lst, gdb)
            paths = sorted(set(paths) | sp.paths_taken())
            ok = outs == wants
            extra = wire + " list%d" % B
        if ok and rng.random() < 0.3 and not cfg.get("direct_upload") and cfg["p"] == 256:   # (lib/server stores bytes: p = 256)
            # lib/server's sparse bucket (SparseDb + update_item_raw; pruned expansion, present-items-only multiply, fold
            # shortcuts) against oracle/sparse_server.cpp, a random fraction of the items present
            sdb = oracle.SparseDb(o)
            try:
                gsp = sp.Database.sparse(p)
            except sp.SpiralError:      # (sparse buckets: 3 <= t_gsw <= 32)
                gsp = None
            frac = float(rng.choice([0.02, 0.2, 0.6, 1.0]))
            present = [int(i) for i in rng.choice(o.num_items, max(1, int(frac * o.num_items)), replace=False)]
            for i in present[:64] if gsp is not None else []:
                data = rng.integers(0, 256, cfg["db_item_size"], dtype=np.uint8).tobytes()
                sdb.update_item_raw(i, data)
                gsp.update_item(i, data)
            for target in (present[0], int(rng.integers(0, o.num_items))) if gsp is not None else ():
                qq = cl.generate_query(target, qs + 1)
                ok = ok and sp.process_query(p, gpp, qq, gsp) == sdb.process_query(pp, qq)
            if gsp is not None:
                extra += " sparse%d" % min(64, len(present))
        n += 1
        print("%4d %s %.1fs %s%s  %s" % (n, "ok  " if ok else "FAIL", time.time() - t0, json.dumps(cfg), extra, ",".join(paths)), flush=True)
        if not ok:
            print("MISMATCH seeds keys=%d query=%d idx=%d" % (ks, qs, idx))
            sys.exit(1)
    print("fuzz: %d cases, all equal" % n)


if __name__ == "__main__":
    main()
