#!/usr/bin/env python
"""Close the loop to the REAL reference (blyssprivacy/sdk lib/spiral-rs): write, for the golden cases of
tests/golden/protocol_vectors.json (+ C1 = BASELINE configs[0] with --c1), the actual bytes a spiral-rs build can load,

    <out>/<case>/params.json     util::params_from_json                      (util.rs:219-263)
    <out>/<case>/pp.bin          PublicParameters::deserialize               (client.rs:212-259)
    <out>/<case>/query.bin       Query::deserialize                          (client.rs:303-329)
    <out>/<case>/db.bin          server::load_preprocessed_db_from_file      (server.rs:373-386; native-endian u64 words,
                                 layout [instance][trial][z][ii][j], the &[u64] process_query takes)
    <out>/<case>/response.bin    what server::process_query must return      (server.rs:650-655)
    <out>/<case>/v_reg.bin       STAGE: expand_query's first output, v_reg_reoriented as_slice()   (server.rs:525-591)
    <out>/<case>/v_folding.bin   STAGE: expand_query's second output, the nu_2 GSW matrices' as_slice() concatenated
    <out>/<case>/plane0.bin      STAGE: multiply_reg_by_database on the first (instance 0, trial 0) database slice: the
                                 num_per 2x1 PolyMatrixNTT outputs' as_slice() concatenated           (server.rs:155-221)
    <out>/manifest.json          sizes + SHA-256 of every file

The three STAGE files let the Rust side name the stage at which a build of the reference and this repository part ways
(query expansion incl. the ChaCha20 row-0 regeneration of Query::deserialize / the database multiply / everything after)
instead of reporting one differing response byte.

The bytes come from the CPU oracle with the fixed seeds of the golden file (their digests are checked against it); with
--gpu the HIP path must reproduce response.bin from the same files before they are written.  The Rust side that reads
them is scripts/ref_check/ref_check.rs (README.md in this directory has the cargo command).  No file of this kit is read
by the product or by the -m gpu tests; it exists so that anyone with cargo can compare against the reference itself."""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_check"))
    ap.add_argument("--c1", action="store_true", help="also C1 = 2^14 x 256 B (1 GiB db.bin)")
    ap.add_argument("--gpu", action="store_true", help="require the HIP path to reproduce every response first")
    args = ap.parse_args()
    import oracle
    oracle.build()
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "protocol_vectors.json")))
    cases = [dict(c, golden=True) for c in vec["cases"]]
    if args.c1:
        from conftest import C1
        cases.append({"name": "c1", "params": C1, "idx": 12345, "key_seed": 11, "query_seed": 12, "golden": False})
    manifest = {"generator": "scripts/ref_check/dump_cases.py", "db_seed": vec["db_seed"], "cases": []}
    for c in cases:
        o = oracle.Params(c["params"])
        cl = oracle.Client(o)
        pp = cl.generate_keys(c["key_seed"])
        q = cl.generate_query(c["idx"], c["query_seed"])
        item, db = o.generate_random_db_and_get_item(c["idx"], vec["db_seed"])
        resp = o.process_query(pp, q, db)
        assert cl.decode_response(resp) == o.item_to_vec(item), "oracle response does not decode"
        if c["golden"]:
            assert (sha(pp), sha(q), sha(db.tobytes()), sha(resp)) == (c["sha256_pp"], c["sha256_query"], c["sha256_db"],
                                                                       c["sha256_response"]), "golden digests moved"
        # stage outputs in the reference's own layouts (all native-endian u64 words)
        # (direct-upload configurations do not expand on the server: process_query takes v_buf / v_ct from the query as they
        # are, server.rs:666-679 -- no stage files for them)
        staged = not c["params"].get("direct_upload")
        slice_words = o.dim0 * o.num_per * o.poly_len
        if staged:
            v_reg, v_fold = o.expand_query(pp, q)
            plane0 = o.multiply_reg_by_database(db[:slice_words], v_reg)
        if args.gpu:
            import numpy as np
            import sdk_amd as sp
            p = sp.Params(c["params"])
            gpp = sp.PublicParameters.deserialize(p, pp)
            got = sp.process_query(p, gpp, q, sp.Database(p).load(db))
            assert got == resp, "HIP response differs from the oracle for " + c["name"]
            if staged:
                g_reg, g_fold = sp.expand_query(p, gpp, q)
                assert (np.asarray(g_reg) == v_reg).all() and (o.db_dim_2 == 0 or (np.asarray(g_fold) == v_fold).all()), \
                    "HIP expand_query differs from the oracle for " + c["name"]
                g_plane = sp.multiply_reg_by_database(p, db[:slice_words], v_reg)
                assert (np.asarray(g_plane) == plane0).all(), "HIP multiply_reg_by_database differs from the oracle for " + c["name"]
        d = os.path.join(args.out, c["name"])
        os.makedirs(d, exist_ok=True)
        files = {"params.json": json.dumps(c["params"], sort_keys=True).encode(), "pp.bin": bytes(pp), "query.bin": bytes(q),
                 "db.bin": db.tobytes(), "response.bin": bytes(resp)}
        if staged:
            files.update({"v_reg.bin": v_reg.tobytes(), "v_folding.bin": v_fold.tobytes(), "plane0.bin": plane0.tobytes()})
        for name, blob in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(blob)
        manifest["cases"].append({"name": c["name"], "params": c["params"], "idx": c["idx"],
                                  "files": {n: {"bytes": len(b), "sha256": sha(b)} for n, b in files.items()},
                                  "hip_checked": bool(args.gpu)})
        print("%-12s pp %d B, query %d B, db %d words, response %d B%s" %
              (c["name"], len(pp), len(q), db.size, len(resp), "  (HIP == oracle)" if args.gpu else ""))
    with open(os.path.join(args.out, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
