"""The DEVICE helper functions of the product headers, run on the CPU and compared with the oracle.

tests/emu/ compiles sdk_amd/csrc/{device_common,wave_ntt,bodies}.hpp unchanged with the host compiler (a stand-in for
<hip/hip_runtime.h> maps work-items to host threads, __syncthreads to a barrier, v_permlane32_swap to an exchange) and runs
them one workgroup at a time.  This is test infrastructure only: it checks arithmetic, index patterns, LDS exchanges and
wave swaps of the kernels' building blocks without a GPU; the kernels themselves, the launch wrappers and everything that
needs gfx950 instructions are covered by the -m gpu tests.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import FAST, ROOT

EMU_DIR = os.path.join(ROOT, "tests", "emu")
BUILD_DIR = os.path.join(EMU_DIR, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
N = 2048
Q0, Q1 = 268369921, 249561089
Q = Q0 * Q1

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def _p32(a):
    return a.ctypes.data_as(u32p)


def _p64(a):
    return a.ctypes.data_as(u64p)


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang (the wave helpers use clang vector extensions)")
    os.makedirs(BUILD_DIR, exist_ok=True)
    so = os.path.join(BUILD_DIR, "libdevice_bodies_emu.so")
    srcs = [os.path.join(EMU_DIR, "device_bodies_emu.cpp"), os.path.join(EMU_DIR, "emu_runtime.cpp"), os.path.join(EMU_DIR, "emu_streams.cpp"),
            os.path.join(ROOT, "sdk_amd", "csrc", "params.cpp")]
    deps = srcs + [os.path.join(EMU_DIR, "hip", "hip_runtime.h"), os.path.join(EMU_DIR, "emu_runtime.hpp")] + [
        os.path.join(ROOT, "sdk_amd", "csrc", f) for f in ("device_common.hpp", "wave_ntt.hpp", "bodies.hpp", "kernels.hpp",
                                                          "params.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run([CLANG, "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-I" + EMU_DIR,
                        "-I" + os.path.join(ROOT, "sdk_amd", "csrc"), "-I" + os.path.join(ROOT, "include")] + srcs +
                       ["-o", so], check=True)
    lib = C.CDLL(so)
    lib.emu_params_new.restype = C.c_void_p
    lib.emu_params_new.argtypes = [C.c_char_p]
    lib.emu_last_error.restype = C.c_char_p
    for name in ("emu_params_free", "emu_ntt_block", "emu_ntt_block_m2", "emu_wave_ntt_inv", "emu_wave_ntt_fwd", "emu_from_ntt", "emu_from_sweep",
                 "emu_digits_to_ntt", "emu_reduce64", "emu_rescale"):
        getattr(lib, name).argtypes = None
    h = lib.emu_params_new(json.dumps(FAST).encode())
    assert h, lib.emu_last_error()

    class Emu:
        pass

    e = Emu()
    e.lib, e.h = lib, C.c_void_p(h)

    def call(name, *args):
        rc = getattr(lib, name)(e.h, *args)
        assert rc == 0, lib.emu_last_error()

    e.call = call
    yield e
    lib.emu_params_free(e.h)


def _edge_polys(rng, q, count):
    """residue vectors < q: random ones plus the corners the reference's lazy butterflies care about"""
    polys = [rng.integers(0, q, N, dtype=np.uint64) for _ in range(count)]
    polys += [np.zeros(N, np.uint64), np.full(N, q - 1, np.uint64), np.full(N, 1, np.uint64)]
    d = np.zeros(N, np.uint64)
    d[0] = 100
    polys.append(d)                                    # ntt.rs:401-409
    polys.append(np.full(N, 100, np.uint64))           # ntt.rs:412-423
    alt = np.zeros(N, np.uint64)
    alt[::2] = q - 1
    polys.append(alt)
    return polys


def _oracle_ntt(op, poly, c):
    both = np.zeros(2 * N, np.uint64)
    both[c * N:(c + 1) * N] = poly
    return op(both)[c * N:(c + 1) * N]


@pytest.mark.parametrize("c", [0, 1])
def test_block_transforms_equal_the_oracle(emu, oracle_mod, c):   # ntt.rs:67-113, 212-258 vs ntt_fwd_block / ntt_inv_block
    o = oracle_mod.Params(FAST)
    q = (Q0, Q1)[c]
    rng = np.random.default_rng(11 + c)
    for poly in _edge_polys(rng, q, 3):
        for inverse, op in ((0, o.ntt_forward), (1, o.ntt_inverse)):
            data = poly.astype(np.uint32)
            emu.call("emu_ntt_block", C.c_int(c), C.c_int(inverse), _p32(data))
            assert np.array_equal(data.astype(np.uint64), _oracle_ntt(op, poly, c)), (c, inverse)


@pytest.mark.parametrize("c", [0, 1])
def test_two_at_a_time_block_transforms(emu, oracle_mod, c):      # ntt_fwd_block_m<2> / ntt_inv_block_m<2>
    o = oracle_mod.Params(FAST)
    q = (Q0, Q1)[c]
    rng = np.random.default_rng(21 + c)
    a, b = rng.integers(0, q, N, dtype=np.uint64), np.full(N, q - 1, np.uint64)
    for inverse, op in ((0, o.ntt_forward), (1, o.ntt_inverse)):
        data = np.concatenate([a, b]).astype(np.uint32)
        emu.call("emu_ntt_block_m2", C.c_int(c), C.c_int(inverse), _p32(data))
        want = np.concatenate([_oracle_ntt(op, a, c), _oracle_ntt(op, b, c)])
        assert np.array_equal(data.astype(np.uint64), want), (c, inverse)


@pytest.mark.parametrize("c", [0, 1])
def test_wave_inverse_transform(emu, oracle_mod, c):              # wntt_inv: 32 coefficients per lane, permlane swap, LDS transpose
    o = oracle_mod.Params(FAST)
    q = (Q0, Q1)[c]
    rng = np.random.default_rng(31 + c)
    polys = _edge_polys(rng, q, 2)[:8]
    for g in range(0, 8, 4):
        four = polys[g:g + 4]
        data = np.concatenate(four).astype(np.uint32)
        emu.call("emu_wave_ntt_inv", C.c_int(c), _p32(data))
        want = np.concatenate([_oracle_ntt(o.ntt_inverse, p, c) for p in four])
        assert np.array_equal(data.astype(np.uint64), want), (c, g)


@pytest.mark.parametrize("c", [0, 1])
def test_wave_forward_transform(emu, oracle_mod, c):              # wntt_fwd: lazy five-instruction butterflies, LDS twiddle copy
    o = oracle_mod.Params(FAST)
    q = (Q0, Q1)[c]
    rng = np.random.default_rng(35 + c)
    polys = _edge_polys(rng, q, 2)[:8]
    polys[1] = polys[1] + np.uint64(q)                  # inputs < 2q are allowed (the fold's digit differences)
    for g in range(0, 8, 4):
        four = polys[g:g + 4]
        want = np.concatenate([_oracle_ntt(o.ntt_forward, p % np.uint64(q), c) for p in four])
        data = np.concatenate(four).astype(np.uint32)
        emu.call("emu_wave_ntt_fwd", C.c_int(c), C.c_int(1), _p32(data))
        assert np.array_equal(data.astype(np.uint64), want), (c, g)
        data = np.concatenate(four).astype(np.uint32)
        emu.call("emu_wave_ntt_fwd", C.c_int(c), C.c_int(0), _p32(data))   # lazy form: < 12q, same residues
        assert int(data.max()) < 12 * q
        assert np.array_equal(data.astype(np.uint64) % np.uint64(q), want), (c, g)


def test_from_ntt_body(emu, oracle_mod):                          # poly.rs:646-663 (+ automorph, poly.rs:393-405) vs ntt_inv_body
    o = oracle_mod.Params(FAST)
    rng = np.random.default_rng(41)
    n = 3
    ntt = np.concatenate([np.concatenate([rng.integers(0, Q0, N, dtype=np.uint64), rng.integers(0, Q1, N, dtype=np.uint64)])
                          for _ in range(n)])
    ntt[:N] = Q0 - 1
    ntt[N:2 * N] = Q1 - 1
    want = o.from_ntt(ntt)
    dst = np.zeros(n * N, np.uint64)
    emu.call("emu_from_ntt", _p32(ntt.astype(np.uint32)), C.c_int(n), C.c_int(0), _p64(dst))
    assert np.array_equal(dst, want)
    for t in (3, N + 1, 2 * N - 1):
        dst = np.zeros(n * N, np.uint64)
        emu.call("emu_from_ntt", _p32(ntt.astype(np.uint32)), C.c_int(n), C.c_int(t), _p64(dst))
        assert np.array_equal(dst, o.automorph(want, t)), t


def test_from_sweep_body(emu, oracle_mod):                        # the sweep-native source [plane][r][crt][z][ii], sums of residues
    o = oracle_mod.Params(FAST)
    rng = np.random.default_rng(51)
    np_, planes = 16, 2      # 16 columns: the XCD-aware block order of ntt_inv_body is in play (64 polys = not % 128: plain order)
    for premod, hi in ((0, None), (1, 8)):
        src = np.zeros((planes, 2, 2, N, np_), np.uint32)
        for c, q in enumerate((Q0, Q1)):
            src[:, :, c] = rng.integers(0, q if not premod else hi * q, (planes, 2, N, np_), dtype=np.uint64).astype(np.uint32)
        dst = np.zeros(planes * np_ * 2 * N, np.uint64)
        emu.call("emu_from_sweep", _p32(np.ascontiguousarray(src)), C.c_int(np_), C.c_int(planes), C.c_int(premod), _p64(dst))
        # poly (plane*np + ii)*2 + r
        red = src.astype(np.uint64)
        red[:, :, 0] %= Q0
        red[:, :, 1] %= Q1
        ntt = np.ascontiguousarray(red.transpose(0, 4, 1, 2, 3)).reshape(-1)   # [plane][ii][r][crt][z]
        assert np.array_equal(dst, o.from_ntt(ntt)), premod


@pytest.mark.parametrize("rdim,cols,t,bits", [(2, 1, 8, 8), (2, 1, 4, 15), (1, 1, 56, 1), (2, 2, 3, 19), (1, 1, 1, 64)])
def test_gadget_digits_to_ntt_body(emu, oracle_mod, rdim, cols, t, bits):   # gadget.rs:34-60 + poly.rs:613-638 vs ntt_fwd_body
    o = oracle_mod.Params(FAST)
    rng = np.random.default_rng(61 + t)
    raw = rng.integers(0, Q, rdim * cols * N, dtype=np.uint64)
    raw[:4] = (0, Q - 1, 1, (1 << 56) - 1 if bits < 64 else Q - 2)
    out = np.zeros(rdim * t * cols * 2 * N, np.uint32)
    emu.call("emu_digits_to_ntt", _p64(raw), C.c_int(1), C.c_int(rdim), C.c_int(cols), C.c_int(t), C.c_int(bits), _p32(out))
    if bits == 64:
        want = o.to_ntt(raw)
    else:
        assert o.get_bits_per(t) == bits
        want = o.to_ntt(o.gadget_invert_rdim(raw, rdim, cols, rdim * t, rdim), no_reduce=True)
    assert np.array_equal(out.astype(np.uint64), want)


def test_scalar_helpers(emu):
    rng = np.random.default_rng(71)
    x = np.concatenate([rng.integers(0, 1 << 63, 4096, dtype=np.uint64) * 2 + rng.integers(0, 2, 4096, dtype=np.uint64),
                        np.array([0, 1, Q0, Q1, Q0 - 1, Q1 - 1, (1 << 64) - 1, (1 << 32) - 1, 1 << 32], np.uint64)])
    for c, q in enumerate((Q0, Q1)):
        out = np.zeros(x.size, np.uint32)
        emu.lib.emu_reduce64(emu.h, C.c_int(c), _p64(x), C.c_int(x.size), _p32(out))
        assert np.array_equal(out.astype(np.uint64), x % np.uint64(q))            # arith.rs:122-134
    out = np.zeros(x.size, np.uint64)
    emu.lib.emu_canon_word(_p64(x), C.c_int(x.size), _p64(out))
    lo, hi = x & np.uint64(0xFFFFFFFF), x >> np.uint64(32)
    assert np.array_equal(out, (lo % np.uint64(Q0)) | ((hi % np.uint64(Q1)) << np.uint64(32)))   # server.rs:196-217 operands


def test_rescale_helper(emu, oracle_mod):                        # arith.rs:429-444 vs rescale_dev
    rng = np.random.default_rng(81)
    a = np.concatenate([rng.integers(0, Q, 2048, dtype=np.uint64), np.array([0, 1, Q - 1, Q // 2, Q // 2 + 1, Q // 2 - 1], np.uint64)])
    for out_mod in (1 << 20, 1024, 4 * 256, (1 << 22) - 3):
        out = np.zeros(a.size, np.uint64)
        emu.lib.emu_rescale(_p64(a), C.c_int(a.size), C.c_uint64(Q), C.c_uint64(out_mod), _p64(out))
        want = np.array([oracle_mod.scalar("rescale", int(v), Q, out_mod) for v in a], np.uint64)
        assert np.array_equal(out, want), out_mod


def test_packed_unit_round_trip(emu):                            # the 7-byte PACKED database word (device_common.hpp)
    rng = np.random.default_rng(91)
    lo = rng.integers(0, Q0, 256, dtype=np.uint64)
    hi = rng.integers(0, Q1, 256, dtype=np.uint64)
    lo[:3] = (0, Q0 - 1, (1 << 28) - 1)
    hi[:3] = ((1 << 28) - 1, 0, Q1 - 1)
    words = lo | (hi << np.uint64(32))
    unit = np.zeros(448, np.uint32)
    back = np.zeros(256, np.uint64)
    emu.lib.emu_pack_unpack(_p64(words), _p32(unit), _p64(back))
    assert np.array_equal(back, words)
