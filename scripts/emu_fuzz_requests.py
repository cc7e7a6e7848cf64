"""Malformed /setup and /private-read bodies against the request layer (endpoint.cpp) of the emulated library, meant to be run
under AddressSanitizer: the parser of client-supplied JSON / base64 must answer garbage with an error, never with a memory error.

    RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
    LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \\
      SPIRAL_HIP_LIB=tests/emu/_build/libspiral_emu_asan.so python scripts/emu_fuzz_requests.py [--seed S] [--cases N]
Every body either raises SpiralError / NotFound or returns well-formed JSON; valid requests mixed in must keep answering
correctly afterwards (the uuid map and the workspaces survive the garbage)."""
import argparse
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def mutate(rng, body):
    b = bytearray(body.encode())
    kind = int(rng.integers(0, 9))
    if kind == 0 and b:
        for _ in range(int(rng.integers(1, 8))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    elif kind == 1 and b:
        del b[int(rng.integers(0, len(b))):]
    elif kind == 2:
        b = b[:int(rng.integers(0, len(b) + 1))] + bytes(rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8)) + b
    elif kind == 3:
        b = bytearray(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes())
    elif kind == 4:
        b = bytearray(b"[" * int(rng.integers(1, 2000)))
    elif kind == 5:
        b = bytearray(json.dumps(["=" * int(rng.integers(0, 50)), "A" * int(rng.integers(0, 50)), "****", ""]).encode())
    elif kind == 6:
        b = bytearray(json.dumps([base64.b64encode(rng.integers(0, 256, int(rng.integers(0, 80)), dtype=np.uint8).tobytes()).decode()
                                  for _ in range(int(rng.integers(0, 6)))]).encode())
    elif kind == 7 and len(b) > 10:
        i = int(rng.integers(0, len(b) - 4))
        b[i:i + 4] = b"\\u00"
    else:
        b = bytearray(b.replace(b'"', b"'", 3))
    return bytes(b).decode("latin-1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=3000)
    a = ap.parse_args()
    import oracle
    import sdk_amd as sp
    from conftest import FAST
    assert hasattr(sp.lib(), "sp_emulated_device_marker"), "set SPIRAL_HIP_LIB to the emulated build"
    cfg = dict(FAST, nu_1=4, nu_2=2, db_item_size=256)
    o = oracle.Params(cfg)
    p = sp.Params(cfg)
    item, db = o.generate_random_db_and_get_item(3)
    gdb = sp.Database(p).load(db)
    srv = sp.Server(p, gdb)
    cl = oracle.Client(o)
    pp = cl.generate_keys(7)
    setup_body = json.dumps(base64.b64encode(pp).decode())
    uuid = json.loads(srv.setup_json(setup_body))["uuid"]
    q = cl.generate_query(3, 8)
    want = o.process_query(pp, q, db)
    read_body = json.dumps([base64.b64encode(uuid.encode() + q).decode()] * 2)
    rng = np.random.default_rng(a.seed)
    errors = answers = 0
    for i in range(a.cases):
        base, fn = (setup_body, srv.setup_json) if rng.random() < 0.3 else (read_body, srv.private_read_json)
        body = mutate(rng, base)
        try:
            out = fn(body)
            json.loads(out)
            answers += 1
        except sp.SpiralError:
            errors += 1
        except UnicodeError:
            errors += 1
        if i % 200 == 199:   # the layer still answers valid requests
            got = json.loads(srv.private_read_json(read_body))
            assert [base64.b64decode(x) for x in got] == [want, want], "a valid request failed after malformed ones"
    print("request fuzz: %d bodies, %d rejected, %d answered, %d clients registered" % (a.cases, errors, answers, srv.clients()))


if __name__ == "__main__":
    main()
