"""The kit for checking against a real spiral-rs build (scripts/ref_check) stays in step with the tree: the dumper, run on
the CPU oracle, must reproduce the committed manifest (sizes and SHA-256 of every params.json / pp.bin / query.bin / db.bin /
response.bin of the golden cases).  CPU only; nothing here touches the product."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dumper_reproduces_the_committed_manifest(tmp_path):
    out = tmp_path / "ref_check"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "ref_check", "dump_cases.py"), "--out", str(out)],
                          cwd=ROOT, stdout=subprocess.DEVNULL)
    got = json.load(open(out / "manifest.json"))
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_check_manifest.json")))
    want_cases = {c["name"]: c["files"] for c in want["cases"]}
    assert got["db_seed"] == want["db_seed"]
    assert len(got["cases"]) >= 5
    for c in got["cases"]:
        assert c["files"] == want_cases[c["name"]], c["name"]
        for name, meta in c["files"].items():
            assert os.path.getsize(out / c["name"] / name) == meta["bytes"]
