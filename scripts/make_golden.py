#!/usr/bin/env python
"""Generate tests/golden/protocol_vectors.json from the CPU oracle (oracle/): fixed seeds -> SHA-256 of the public
parameters, the query, the database and the response, plus the decoded item.  The reference ships no golden
ciphertexts (its tests draw fresh entropy) and cannot be built here, so these vectors pin the *oracle* (and through
the GPU parity tests the HIP path) against silent drift; they are not reference outputs."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import FAST, FAST56, SMALL_INST2  # noqa: E402

CASES = [
    ("fast", FAST, 137, 1, 2),
    ("fast56", FAST56, 301, 3, 4),
    ("inst2", SMALL_INST2, 123, 5, 6),
    ("v1", dict(FAST, version=1), 200, 7, 8),
    ("direct", dict(FAST, direct_upload=1), 100, 9, 10),
]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    out = {"generator": "scripts/make_golden.py", "db_seed": 0x123456789, "cases": []}
    for name, cfg, idx, ks, qs in CASES:
        o = oracle.Params(cfg)
        cl = oracle.Client(o)
        pp = cl.generate_keys(ks)
        q = cl.generate_query(idx, qs)
        item, db = o.generate_random_db_and_get_item(idx)
        resp = o.process_query(pp, q, db)
        dec = cl.decode_response(resp)
        assert dec == o.item_to_vec(item)
        v_reg, v_fold = (o.expand_query(pp, q) if "direct_upload" not in cfg else (None, None))
        out["cases"].append({
            "name": name, "params": cfg, "idx": idx, "key_seed": ks, "query_seed": qs,
            "setup_bytes": len(pp), "query_bytes": len(q), "response_bytes": len(resp),
            "sha256_pp": sha(pp), "sha256_query": sha(q), "sha256_db": sha(db.tobytes()),
            "sha256_response": sha(resp), "response_head_hex": resp[:32].hex(),
            "sha256_v_reg_reoriented": sha(v_reg.tobytes()) if v_reg is not None else None,
            "sha256_v_folding": sha(v_fold.tobytes()) if v_fold is not None else None,
            "sha256_decoded": sha(dec),
        })
    path = os.path.join(ROOT, "tests", "golden", "protocol_vectors.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
