"""bench.py's `roofline.traffic` is a replay of rocprofv3 PMC passes made on the builder's box.  It must stop being reported the
day the sweep kernel changes: the record carries the SHA-256 of the profiled kernel's machine code and is only replayed into a
library whose kernel has the same.  CPU-only: the signature is read from the built .so."""
import json
import os

import pytest

from conftest import ROOT

import bench
from sdk_amd import library_path
from sdk_amd.kernel_signature import SWEEP_C2, compiler_version, kernel_signature

pytestmark = pytest.mark.skipif(not os.path.exists(library_path()), reason="libspiral_hip.so not built")


def test_signature_names_one_kernel_and_is_stable():
    sig, name = kernel_signature(library_path(), SWEEP_C2)
    assert len(sig) == 32 and "k_sweep_packed_ring" in name
    assert kernel_signature(library_path(), SWEEP_C2) == (sig, name)
    other, _ = kernel_signature(library_path(), "k_sweep_packed_ringILi4E")      # a different instantiation: different code
    assert other != sig
    with pytest.raises(LookupError):
        kernel_signature(library_path(), "k_no_such_kernel")


def test_newest_record_matches_the_built_library():
    """The tracked record bench.py would replay was measured on THIS sweep kernel (fails when sweep.hip's code changes without
    a new PMC pass -- which is the point)."""
    sig, _ = kernel_signature(library_path(), SWEEP_C2)
    traffic, source = bench.pmc_traffic("c2", 1, 4, library_path())
    if traffic is None:
        # a library built by ANOTHER hipcc has other machine code for the same source: that is not the event this test is for
        # (ADVICE r05).  Records carry the compiler of the profiled build since r06; older ones were made with the image's ROCm 7.2.0.
        names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("pmc_sweep_c2.json"))
        rec = json.load(open(os.path.join(ROOT, "profiles", names[-1])))
        recorded = rec.get("hipcc_version") or "HIP version: 7.2.26015"
        here = compiler_version()
        if not here or recorded.split(" | ")[0] not in here:
            pytest.skip("built with %r, newest record %s was profiled on a build of %r" % (here, names[-1], recorded))
    assert traffic is not None and sig in source, source
    assert 14.5e9 < traffic < 16.5e9          # one plane launch: 15.1 GB algorithmic in the resident format


def test_a_record_of_another_kernel_is_refused(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_final_pmc_sweep_c2.json")))
    rec["kernel_signature"] = "0" * 32
    json.dump(rec, open(prof / "r99_pmc_sweep_c2.json", "w"))
    unsigned = dict(rec)
    del unsigned["kernel_signature"]
    json.dump(unsigned, open(prof / "r98_pmc_sweep_c2.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    traffic, source = bench.pmc_traffic("c2", 1, 4, library_path())
    assert traffic is None and "not replayed" in source and "r99_pmc_sweep_c2.json" in source
