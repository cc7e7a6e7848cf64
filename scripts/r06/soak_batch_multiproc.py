"""Batched-call soak for several processes on one GPU (round 6): a PACKED database with its digit-planar copy, a pool of 48 distinct
queries whose single-query responses are checked against the ORACLE first; then REPS lists of 2 .. 16 random distinct queries
through sp_process_query_batch (grouped expansion launches, planar / two-tile / one-tile passes) -- every response must equal
the stored one.  Usage: python scripts/r06/soak_batch_multiproc.py REPS SEED"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

import oracle
import sdk_amd as sp

cfg = {"n": 2, "nu_1": 6, "nu_2": 7, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 56,
       "instances": 1, "db_item_size": 256}
reps, seed = int(sys.argv[1]), int(sys.argv[2])
o = oracle.Params(cfg)
p = sp.Params(cfg)
rng = np.random.default_rng(seed)
cl = oracle.Client(o)
pp = cl.generate_keys(seed)
gpp = sp.PublicParameters.deserialize(p, pp)
item, dbw = o.generate_random_db_and_get_item(7)
db = sp.Database(p).load(dbw)
db.prepare_batch()
qs = [cl.generate_query(int(rng.integers(0, o.num_items)), 500 + k) for k in range(48)]
want = [sp.process_query(p, gpp, q, db) for q in qs]
for k in (0, 17, 47):
    assert want[k] == o.process_query(pp, qs[k], dbw), "single-query path differs from the oracle"
bad = 0
for r in range(reps):
    B = int(rng.choice([2, 3, 5, 8, 9, 11, 12, 16]))
    pick = [int(x) for x in rng.choice(48, B, replace=False)]
    outs = sp.process_query_batch(p, gpp, [qs[i] for i in pick], db)
    for i, out in zip(pick, outs):
        if out != want[i]:
            bad += 1
            print("rep %d: list of %d, query %d differs" % (r, B, i), flush=True)
print("batch soak: %d wrong responses in %d lists; paths %s" % (bad, reps, sorted(sp.paths_taken())))
sys.exit(1 if bad else 0)
