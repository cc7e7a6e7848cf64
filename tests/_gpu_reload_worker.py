"""Worker of test_gpu_parity.py::test_database_reload_beside_other_processes (test infrastructure): reloads ONE database handle
REPS times and checks a query against the oracle after every load.  Several of these run at the same time on one GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402

cfg = {"n": 2, "nu_1": 6, "nu_2": 3, "p": 256, "q2_bits": 16, "t_gsw": 4, "t_conv": 1, "t_exp_left": 14, "t_exp_right": 14,
       "instances": 2, "db_item_size": 16384, "version": 1, "direct_upload": 1}
reps = int(sys.argv[1])
o = oracle.Params(cfg)
cl = oracle.Client(o)
pp = cl.generate_keys(714463203)
q = cl.generate_query(449, 808286267)
item, db = o.generate_random_db_and_get_item(449)
want = o.process_query(pp, q, db)
p = sp.Params(cfg)
gpp = sp.PublicParameters.deserialize(p, pp)
gdb = sp.Database(p)
bad = 0
for r in range(reps):
    gdb.load(db)
    bad += sp.process_query(p, gpp, q, gdb) != want
print("reload-worker: %d of %d loads gave a wrong response" % (bad, reps))
sys.exit(1 if bad else 0)
