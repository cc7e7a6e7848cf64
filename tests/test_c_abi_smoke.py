"""tests/c_abi_smoke.c built with gcc against include/spiral_hip.h and run as a stand-alone process: the C ABI is
usable without Python (the host a maintainer links is Rust / C).  The CPU part links the program and runs the
host-only entry points; the -m gpu part runs one query (and the RCCL-sharded entry point at world size 1) and
compares the response bytes with the oracle's."""
import json
import os
import subprocess

import pytest

from conftest import FAST

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import sdk_amd
    so_dir = os.path.dirname(sdk_amd.library_path())
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c_abi_smoke.c"),
                           "-I", os.path.join(ROOT, "include"), "-L", so_dir, "-lspiral_hip",
                           "-Wl,-rpath," + so_dir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib",
                           "-o", exe])
    return exe


def test_c_program_links_and_host_entry_points_work(tmp_path):
    exe = _build(tmp_path)
    (tmp_path / "params.json").write_text(json.dumps(FAST))
    r = subprocess.run([exe, "host", str(tmp_path / "params.json")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host-ok" in r.stdout, (r.stdout, r.stderr)
    assert "response_bytes=20480" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["query", "sharded"])
def test_c_program_query_equals_oracle(tmp_path, oracle_mod, mode):
    cfg = dict(FAST, nu_2=7, db_item_size=256)          # PACKED database, fused + tail fold levels
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(41)
    idx = 4242 % o.num_items
    q = cl.generate_query(idx, 42)
    item, db = o.generate_random_db_and_get_item(idx)
    expect = o.process_query(pp, q, db)
    (tmp_path / "params.json").write_text(json.dumps(cfg))
    (tmp_path / "pp.bin").write_bytes(pp)
    (tmp_path / "query.bin").write_bytes(q)
    (tmp_path / "db.bin").write_bytes(db.tobytes())
    (tmp_path / "expected.bin").write_bytes(expect)
    exe = _build(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, mode] + [str(tmp_path / f) for f in ("params.json", "pp.bin", "query.bin", "db.bin", "expected.bin")],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "== the oracle's" in r.stdout
    if mode == "sharded":
        assert "rccl_in_library" in r.stdout
