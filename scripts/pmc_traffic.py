#!/usr/bin/env python
"""HBM traffic of the db-sweep kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel
trace only) -> the JSON record bench.py reads for roofline.traffic.  gfx950 correction per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts half of a wide streaming read."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdk_amd.kernel_signature import SWEEP_C2, compiler_version, kernel_signature  # noqa: E402


def mean_counter(db, counter, kernel_substr):
    c = sqlite3.connect(db)
    rows = list(c.execute("select value from counters_collection where counter_name = ? and kernel_name like ?",
                          (counter, "%" + kernel_substr + "%")))
    vals = [r[0] for r in rows]
    return sum(vals) / len(vals), len(vals)


def main(fetch_db, write_db, out, kernel="k_sweep_packed_", launches_per_query=4, alg_bytes_per_query=69323456512):
    f, nf = mean_counter(fetch_db, "FETCH_SIZE", kernel)
    w, nw = mean_counter(write_db, "WRITE_SIZE", kernel)
    so = os.environ.get("SPIRAL_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdk_amd", "libspiral_hip.so")
    sig, mangled = kernel_signature(so, SWEEP_C2)
    rec = {
        # identity of the profiled kernel: bench.py replays this record only into a library whose kernel has the same machine code
        "kernel_signature": sig,
        "kernel_signature_of": mangled,
        "hipcc_version": compiler_version(),   # of the build that was profiled (tests/test_traffic_replay_guard.py)
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py "
                "--steps 1 --warmup 0 --sweep-iters 2 --no-cpu-baseline`, config c2, kernel %s (7-byte PACKED "
                "database, one launch per plane), %d / %d dispatches" % (kernel, nf, nw),
        "kernel": kernel,
        "launches_per_query": launches_per_query,
        "FETCH_SIZE_KB_mean": f,
        "WRITE_SIZE_KB_mean": w,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (calibrated on the "
                      "8-byte kernel, profiles/r01_pmc_sweep_c2.json) -> read bytes = 2 * FETCH_SIZE * 1024; "
                      "WRITE_SIZE * 1024 equals the kernel's own u32 stores",
        "read_bytes_per_launch": 2 * f * 1024,
        "write_bytes_per_launch": w * 1024,
        "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
        "algorithmic_bytes_per_launch": alg_bytes_per_query / launches_per_query,
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
