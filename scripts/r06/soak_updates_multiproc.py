"""Upsert soak for several processes on one GPU (round 6, after profiles/r06_stale_staging.md): a dense database preprocessed on
the device, then REPS times: overwrite a random item with random bytes (sp_db_update_item: the handle's kept staging buffer),
query it, decode the response with the oracle's client -- must be the bytes just written -- and every 16th repetition also query an
item written long ago.  Usage: python scripts/r06/soak_updates_multiproc.py REPS SEED"""
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
blob = bytearray(rng.integers(0, 256, o.num_items * 256, dtype=np.uint8).tobytes())
cl = oracle.Client(o)
gpp = sp.PublicParameters.deserialize(p, cl.generate_keys(seed))
db = sp.Database(p).load_items(bytes(blob))
db.prepare_batch()          # (the digit-planar copy is patched by every upsert too)
idxs = [int(x) for x in rng.integers(0, o.num_items, 24)]
queries = {i: cl.generate_query(i, 100 + k) for k, i in enumerate(idxs)}
bad = 0


def check(i, what):
    global bad
    got = cl.decode_response(sp.process_query(p, gpp, queries[i], db))
    item = bytes(blob[i * 256:(i + 1) * 256])
    if not all(got[t * 64:(t + 1) * 64] == item[t * 64:(t + 1) * 64] for t in range(4)):
        bad += 1
        print("rep %d: item %d (%s) decodes to other bytes" % (r, i, what), flush=True)


for r in range(reps):
    i = idxs[int(rng.integers(0, len(idxs)))]
    data = rng.integers(0, 256, 256, dtype=np.uint8).tobytes()
    blob[i * 256:(i + 1) * 256] = data
    db.update_item(i, data)
    check(i, "just written")
    if r % 16 == 0:
        check(idxs[(r // 16) % len(idxs)], "written earlier")
print("update soak: %d wrong of %d upserts" % (bad, reps))
sys.exit(1 if bad else 0)
