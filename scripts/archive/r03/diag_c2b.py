"""When do the C2 public parameters change?  (SPIRAL_DB_CONTIGUOUS=1, small configurations first.)  pp is exported and
compared with the oracle's deserialisation after every step: deserialize, database allocation, database fill."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402
from conftest import C2, FAST, FAST56, SMALL_INST2  # noqa: E402
from diag_c2 import small  # noqa: E402

SEED = 0x123456789

if __name__ == "__main__":
    if os.environ.get("SMALL_FIRST", "1") == "1":
        print("small first:", [small(c, i) for c, i in ((FAST, 3), (FAST56, 300), (SMALL_INST2, 12), (dict(FAST, version=1), 7),
                                                        (dict(FAST, direct_upload=1), 100))], flush=True)
    o = oracle.Params(C2)
    cl = oracle.Client(o)
    pp = cl.generate_keys(501)
    flat = o.pp_deserialize_flat(pp)
    p = sp.Params(C2)
    gpp = sp.PublicParameters.deserialize(p, pp)

    def check(when):
        e = gpp.export()
        bad = np.nonzero(e != flat)[0]
        print("%-34s pp ok %s%s" % (when, bad.size == 0, "" if bad.size == 0 else "  (%d of %d words differ, first at word %d = poly %d, last at %d)" %
                                    (bad.size, e.size, bad[0], bad[0] // 4096, bad[-1])), flush=True)
        if bad.size and when == "after deserialize":
            # what do the wrong words hold, and how are they laid out?
            runs = np.split(bad, np.nonzero(np.diff(bad) != 1)[0] + 1)
            print("    %d runs of wrong words; run lengths (first 12): %s" % (len(runs), [len(r) for r in runs[:12]]))
            print("    run starts in bytes of the u32 device buffer (first 12): %s" % [hex(int(r[0]) * 4) for r in runs[:12]])
            vals = e[bad]
            print("    wrong values: %d zero, %d nonzero; first nonzero ones: %s; expected there: %s" %
                  (int((vals == 0).sum()), int((vals != 0).sum()), [hex(int(v)) for v in vals[vals != 0][:6]], [hex(int(v)) for v in flat[bad][:6]]))

    check("after deserialize")
    check("after deserialize (again)")
    gdb = sp.Database(p)
    check("after the database allocation")
    gdb.fill_synthetic(SEED)
    check("after the database fill")
    gpp2 = sp.PublicParameters.deserialize(p, pp)
    e2 = gpp2.export()
    print("a second deserialize now: pp ok", bool((e2 == flat).all()), flush=True)
