"""The C-ABI library loads on a GPU-less host and exports every symbol include/spiral_hip.h declares;
host-only entry points (params, sizes, encode, synthetic-word hash) work without a device, and compute
entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "spiral_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    import sdk_amd
    lib = C.CDLL(sdk_amd.library_path())
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_rust_shim_declares_every_symbol():
    """rust/src/hip.rs (source only: no rustc in the image) keeps an extern declaration for every entry point of
    include/spiral_hip.h, with the same number of arguments"""
    hdr = open(os.path.join(ROOT, "include", "spiral_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs = open(os.path.join(ROOT, "rust", "src", "hip.rs")).read()
    for name in _declared():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, hdr, flags=re.S)
        assert m, name
        c_args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        r = re.search(r"pub fn %s\s*\(([^;]*?)\)\s*(->[^;]*)?;" % name, rs, flags=re.S)
        assert r, "rust/src/hip.rs lacks %s" % name
        r_args = [a for a in r.group(1).split(",") if a.strip()]
        assert len(r_args) == len(c_args), (name, c_args, r_args)


def test_host_only_entry_points(oracle_mod):
    import sdk_amd as sp
    from conftest import C2, FAST, P2
    for cfg in (FAST, P2, C2):
        p, o = sp.Params(cfg), oracle_mod.Params(cfg)
        for k in ("setup_bytes", "query_bytes", "g", "stop_round", "num_items", "modulus", "modulus_log2"):
            assert p.get(k) == o.get(k), (k, cfg)
        assert p.get("response_bytes") == o.response_bytes()
        for c in range(2):
            for w in range(4):
                assert (p.ntt_table(c, w) == o.ntt_table(c, w)).all()
    # single-quoted preset strings as the reference writes them (util.rs:7-20)
    p = sp.params_from_json("{'n': 2, 'nu_1': 9, 'nu_2': 6, 'p': 256, 'q2_bits': 20, 's_e': 87.6, 't_gsw': 8, "
                            "'t_conv': 4, 't_exp_left': 8, 't_exp_right': 56, 'instances': 1, 'db_item_size': 8192 }")
    assert p.setup_bytes() == 8126496 and p.query_bytes() == 16416
    with pytest.raises(sp.SpiralError):
        sp.params_from_json('{"n": 2}')
    with pytest.raises(sp.SpiralError):
        sp.params_from_json("not json")
    # encode (server.rs:470-503) is host code in the product too
    rng = np.random.default_rng(1)
    o = oracle_mod.Params(FAST)
    packed = rng.integers(0, 66974689739603969, 3 * 2 * 2048, dtype=np.uint64)
    packed[:5] = [0, 1, 66974689739603968, 33487344869801984, 33487344869801985]
    assert sp.encode(sp.Params(FAST), packed) == o.encode(packed)
    # synthetic DB word hash: C and numpy agree
    from sdk_amd.spiral import synth_word, synth_words
    idx = np.array([0, 1, 2**31, 2**40 + 12345], dtype=np.uint64)
    assert [int(x) for x in synth_words(77, idx)] == [synth_word(77, int(i)) for i in idx]


def test_compute_fails_loudly_without_gpu():
    import sdk_amd as sp
    from conftest import FAST
    if sp.lib().sp_device_count() > 0:
        pytest.skip("a GPU is present")
    p = sp.Params(FAST)
    with pytest.raises(sp.SpiralError):
        sp.to_ntt(p, np.zeros(2048, dtype=np.uint64))
    with pytest.raises(sp.SpiralError):
        sp.Database(p)


def test_params_validation_rejects_what_the_reference_cannot_run():
    """Params are validated on the host: a configuration the reference would index out of bounds with
    (coefficient_expansion writes 2*max(dim0, t_gsw*nu_2) ciphertexts into a 2^g vector, server.rs:525-591),
    unknown packing versions and version 1 with n != 2 (lib/server pack.rs:46-99) are errors, not crashes."""
    import sdk_amd as sp
    from conftest import FAST
    ok = dict(FAST)
    sp.Params(ok)
    assert sp.Params(dict(FAST, q2_bits=3)).get("q2_bits") == 14      # raised to MIN_Q2_BITS as util.rs:230 does
    for bad in (dict(FAST, nu_1=3, nu_2=7, t_gsw=8),      # 2*56 > 2^g = 64
                dict(FAST, nu_2=1, t_gsw=1),              # stop_round = 0: one right expansion matrix, g - 1 rounds that index further
                dict(FAST, version=2),
                dict(FAST, q2_bits=37),                   # past Q2_VALUES (params.rs:8-46)
                dict(FAST, version=1, n=3),
                dict(FAST, nu_1=0, nu_2=0, t_gsw=8)):
        try:
            sp.Params(bad)
        except sp.SpiralError:
            continue
        # nu_1 = 0 is legal if the reference accepts it; only the first five must raise
        assert bad.get("nu_1") == 0, bad


def test_library_chacha20_equals_the_pinned_restatement(oracle_mod):
    """sp_debug_chacha20_u64 (no GPU): the product's keystream -- eight blocks side by side with AVX2, one block at a time for short
    requests and tails -- against the oracle's, which tests/test_oracle_kat.py pins to RFC 8439 and to rand_chacha's own vectors."""
    import ctypes as C
    import numpy as np
    from sdk_amd import library_path
    L = C.CDLL(library_path())
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 2048, 2051):
        for seed in (bytes(range(32)), bytes(31) + b"\x01", b"\xff" * 32):
            out = np.zeros(max(n, 1), dtype=np.uint64)
            assert L.sp_debug_chacha20_u64((C.c_uint8 * 32)(*seed), out.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == 0
            assert (out[:n] == oracle_mod.chacha20_rng_u64(seed, n)[:n]).all(), (n, seed[:2])
