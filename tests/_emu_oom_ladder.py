"""TEST INFRASTRUCTURE (run by tests/test_emulated_library.py in a child process with SPIRAL_HIP_LIB = the emulated build): the
out-of-memory ladder of sp_process_query_batch (capi.cpp), driven by the emulator's device-memory budget
(tests/emu/emu_streams.cpp: hipMalloc fails once the live device bytes would exceed it).

A list of nine queries against a PACKED 64 x 128 database with its digit-planar copy built, three times, every response against
the oracle:
  A  the budget runs out while the group's workspaces are acquired, but holds once the planar copy is given back: the call succeeds
     through the PACKED two-tile kernel and the copy's memory is released;
  B  without a planar copy, room for about ten workspaces: groups of 8, one at a time;
  C  room for two workspaces: one query at a time.
Usage: python tests/_emu_oom_ladder.py A|B|C   (one case per process; prints oom-ladder-ok)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402


def main():
    which = [a for a in sys.argv[1:] if a in ("A", "B", "C")] or ["A"]
    L = sp.lib()
    assert hasattr(L, "sp_emulated_device_marker"), "set SPIRAL_HIP_LIB to the emulated build"
    L.emu_device_live_bytes.restype = C.c_size_t
    L.emu_set_device_budget.argtypes = [C.c_size_t]
    cfg = {"n": 2, "nu_1": 6, "nu_2": 7, "p": 256, "q2_bits": 20, "t_gsw": 4, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 56,
           "instances": 1, "db_item_size": 256}
    o = oracle.Params(cfg)
    cl = oracle.Client(o)
    pp = cl.generate_keys(91)
    item, db = o.generate_random_db_and_get_item(3)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    gdb = sp.Database(p).load(db)
    case = ([a for a in sys.argv[1:] if a in ("A", "B", "C")] or ["A"])[0]
    # (A: eleven -- the call's own estimate of free memory, 32 x 3 x the first-dimension output, must still allow groups of 16 where
    # the real workspaces of this shape, ~4x that, do not fit: the out-of-memory has to strike inside the group)
    B = 11 if case == "A" else 9
    qs = [cl.generate_query((211 * i + 3) % o.num_items, 400 + i) for i in range(B)]
    want = [o.process_query(pp, q, db) for q in qs]
    # the size of ONE workspace: a single query with no budget leaves one in the pool
    live0 = L.emu_device_live_bytes()
    assert sp.process_query(p, gpp, qs[0], gdb) == want[0]
    ws = L.emu_device_live_bytes() - live0
    assert ws > 0
    print("one workspace: %.1f MiB" % (ws / 2**20), flush=True)
    assert len(which) == 1, "one case per process: the workspace pool of a Params handle only grows"
    case = which[0]
    # one workspace is pooled now; the queries of the list need one each
    if case == "A":
        assert gdb.prepare_batch() is True
        copy = gdb.batch_copy_bytes()
        assert 2 * ws < copy, "the shape no longer makes the planar copy worth two workspaces"
        L.emu_set_device_budget(L.emu_device_live_bytes() + 8 * ws + ws // 2)     # the tenth workspace does not fit ...
        sp.paths_taken()
        got = sp.process_query_batch(p, gpp, qs, gdb)
        taken = sp.paths_taken()
        assert got == want, "case A: responses differ"
        # ... until the copy is given back (2.5 workspaces' worth): then all eleven run as ONE group through the PACKED two-tile kernel
        assert "sweep_batch_mfma_two_tiles" in taken and "sweep_batch_planar" not in taken, taken
        assert "expand_group" in taken, taken        # (the group's expansions ran as shared launches, run_begin_group)
        assert gdb.batch_copy_bytes() == 0
        L.emu_set_device_budget(0)
        assert gdb.prepare_batch() is True and gdb.batch_copy_bytes() == copy          # a later call may build it again
    elif case == "B":
        L.sp_debug_set(b"batch_planar", C.c_long(0))
        L.emu_set_device_budget(L.emu_device_live_bytes() + 7 * ws + ws // 2)     # eight workspaces in all: no group of nine
        sp.paths_taken()
        got = sp.process_query_batch(p, gpp, qs, gdb)
        taken = sp.paths_taken()
        assert got == want, "case B: responses differ"
        # groups of 8, one at a time: the one-tile matrix-core pass for the eight, the vector kernel for the ninth
        assert "sweep_batch_mfma" in taken and "sweep_batch_mfma_two_tiles" not in taken, taken
    else:
        L.sp_debug_set(b"batch_planar", C.c_long(0))
        L.emu_set_device_budget(L.emu_device_live_bytes() + ws // 2)              # nothing but the pooled workspace
        sp.paths_taken()
        got = sp.process_query_batch(p, gpp, qs, gdb)
        taken = sp.paths_taken()
        assert got == want, "case C: responses differ"
        assert "sweep_batch" not in taken and "expand_group" not in taken, taken     # one query at a time
    print("case %s ok: %s" % (case, ",".join(sorted(t for t in taken if t.startswith(("sweep", "expand_group"))))), flush=True)
    L.emu_set_device_budget(0)
    L.sp_debug_set(b"batch_planar", C.c_long(1))
    print("oom-ladder-ok")


if __name__ == "__main__":
    main()
