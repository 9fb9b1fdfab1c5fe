"""The oracle's CKKS encoder restatement (oracle/o_encode.c) pinned by what encoding means:
decode inverts encode, the product of two encoded polynomials decodes to the slot-wise
product, and the automorphism X -> X^5 rotates the slots by one (the reference's
rotate_rows semantics, ckks/evaluationkey.cu:408)."""
import ctypes

import numpy as np
import pytest

from he_math import RLWE


@pytest.fixture(scope="module")
def ctx(oracle):
    n_power = 11
    arr = (ctypes.c_int * 4)(50, 40, 40, 50)
    out = (ctypes.c_uint64 * 4)()
    assert oracle.lib().o_generate_primes(1 << n_power, arr, 4, out) == 0
    primes = [int(v) for v in out]
    return oracle.OracleContext(oracle.CKKS, n_power, primes, 3, 1), primes


def test_decode_inverts_encode(ctx):
    o, primes = ctx
    slots = o.n // 2
    g = np.random.default_rng(1)
    x = g.uniform(-10, 10, slots)
    scale = 2.0 ** 40
    plain = o.ckks_encode(x, scale)
    assert np.max(np.abs(o.ckks_decode(plain, scale) - x)) < 1e-7
    # short message: the missing slots are zero
    y = o.ckks_decode(o.ckks_encode(x[:5], scale), scale)
    assert np.max(np.abs(y[:5] - x[:5])) < 1e-7 and np.max(np.abs(y[5:])) < 1e-7
    # lower level: the first l limbs of the same plaintext
    l = o.Q - 1
    sub = plain.reshape(o.Q, o.n)[:l].reshape(-1)
    assert np.max(np.abs(o.ckks_decode(sub, scale, depth=1) - x)) < 1e-7


def test_polynomial_product_is_slotwise_product(ctx):
    o, primes = ctx
    slots = o.n // 2
    g = np.random.default_rng(2)
    x, y = g.uniform(-4, 4, slots), g.uniform(-4, 4, slots)
    scale = 2.0 ** 30
    px = o.ckks_encode(x, scale).reshape(o.Q, o.n)
    py = o.ckks_encode(y, scale).reshape(o.Q, o.n)
    prod = np.stack([np.array([(int(a) * int(b)) % primes[j] for a, b in zip(px[j], py[j])], dtype=np.uint64)
                     for j in range(o.Q)]).reshape(-1)
    got = o.ckks_decode(prod, scale * scale)
    assert np.max(np.abs(got - x * y)) < 1e-5


def test_galois_5_rotates_slots(ctx):
    o, primes = ctx
    n, slots = o.n, o.n // 2
    he = RLWE(o, seed=0)
    g = np.random.default_rng(3)
    x = g.uniform(-4, 4, slots)
    scale = 2.0 ** 40
    plain = o.ckks_encode(x, scale).reshape(o.Q, n)
    ids = list(range(o.Q))
    coeff = he.ntt_limbs(plain, ids, inverse=True)
    rot = np.zeros_like(coeff)
    for j in ids:
        q = primes[j]
        for i in range(n):
            r = (i * 5) % (2 * n)
            if r >= n:
                rot[j][r - n] = (q - int(coeff[j][i])) % q
            else:
                rot[j][r] = coeff[j][i]
    back = he.ntt_limbs(rot, ids).reshape(-1)
    got = o.ckks_decode(back, scale)
    assert np.max(np.abs(got - np.roll(x, -1))) < 1e-7


def test_other_encodings(ctx):
    """complex slots, coefficient encoding and the scalar shortcut (ckks/encoder.cu:222-446),
    pinned by their meaning: complex decode inverts complex encode and its product is slot-wise;
    the coefficient encoding IS the polynomial (INTT of the plaintext = round(m * scale), negacyclic
    products); a scalar equals the constant vector."""
    o, primes = ctx
    n, slots = o.n, o.n // 2
    g = np.random.default_rng(4)
    scale = 2.0 ** 35
    z = g.uniform(-4, 4, slots) + 1j * g.uniform(-4, 4, slots)
    w = g.uniform(-4, 4, slots) + 1j * g.uniform(-4, 4, slots)
    pz, pw = o.ckks_encode_ex(1, z, scale), o.ckks_encode_ex(1, w, scale)
    assert np.max(np.abs(o.ckks_decode_ex(1, pz, scale) - z)) < 1e-6
    he = RLWE(o, seed=2)
    prod = np.concatenate([he.mulmod(pz.reshape(o.Q, n)[j], pw.reshape(o.Q, n)[j], primes[j]) for j in range(o.Q)])
    assert np.max(np.abs(o.ckks_decode_ex(1, prod, scale * scale) - z * w)) < 1e-5
    # a real vector is the complex vector with zero imaginary parts
    x = g.uniform(-4, 4, slots)
    assert np.array_equal(o.ckks_encode_ex(1, x + 0j, scale), o.ckks_encode(x, scale))
    # coefficient encoding
    m = g.uniform(-8, 8, n)
    pm = o.ckks_encode_ex(2, m, scale)
    coeff = he.ntt_limbs(pm.reshape(o.Q, n), list(range(o.Q)), inverse=True)
    for j in range(o.Q):
        want = np.array([int(round(v * scale)) % primes[j] for v in m], dtype=np.uint64)
        assert np.array_equal(coeff[j], want)
    assert np.max(np.abs(o.ckks_decode_ex(2, pm, scale) - m)) < 1e-9
    short = o.ckks_decode_ex(2, o.ckks_encode_ex(2, m[:3], scale), scale)
    assert np.max(np.abs(short[:3] - m[:3])) < 1e-9 and np.max(np.abs(short[3:])) == 0.0
    # scalar
    for v in (2.75, -1.5, 0.0):
        ps = o.ckks_encode_ex(3, [v], scale)
        assert np.array_equal(ps, o.ckks_encode(np.full(slots, v), scale))
        assert np.max(np.abs(o.ckks_decode(ps, scale) - v)) < 1e-9


def test_constant_operations_and_mult_i(ctx):
    """add / sub / multiply by a real constant and multiplication / division by i
    (ckks/operator.cuh:312-390, :812-925, :969-1050), pinned by their meaning on a trivial ciphertext
    (c0 = plaintext, c1 = 0): slots + c, slots - c, slots * c at scale^2, and i * slots."""
    o, primes = ctx
    n, slots, Q = o.n, o.n // 2, o.Q
    g = np.random.default_rng(6)
    scale = 2.0 ** 35
    z = g.uniform(-4, 4, slots) + 1j * g.uniform(-4, 4, slots)
    ct = np.concatenate([o.ckks_encode_ex(1, z, scale), np.zeros(Q * n, dtype=np.uint64)])
    for c in (2.5, -1.75, 0.0):
        add = o.ckks_constant_op(0, ct, c * scale, Q)
        assert np.max(np.abs(o.ckks_decode_ex(1, add[:Q * n], scale) - (z + c))) < 1e-6
        assert np.array_equal(add[Q * n:], ct[Q * n:])
        sub = o.ckks_constant_op(1, ct, c * scale, Q)
        assert np.max(np.abs(o.ckks_decode_ex(1, sub[:Q * n], scale) - (z - c))) < 1e-6
        mul = o.ckks_constant_op(2, ct, c * scale, Q)
        assert np.max(np.abs(o.ckks_decode_ex(1, mul[:Q * n], scale * scale) - z * c)) < 1e-5
    # at a lower level only the first l limbs take part
    l = Q - 1
    sub_ct = np.concatenate([ct[:l * n], ct[Q * n:Q * n + l * n]])
    add = o.ckks_constant_op(0, sub_ct, 3.0 * scale, l)
    assert np.max(np.abs(o.ckks_decode_ex(1, add[:l * n], scale, depth=1) - (z + 3.0))) < 1e-6
    mi = o.ckks_mult_i(ct, Q)
    assert np.max(np.abs(o.ckks_decode_ex(1, mi[:Q * n], scale) - 1j * z)) < 1e-6
    di = o.ckks_mult_i(ct, Q, divide=True)
    assert np.max(np.abs(o.ckks_decode_ex(1, di[:Q * n], scale) + 1j * z)) < 1e-6
    assert np.array_equal(o.ckks_mult_i(mi, Q, divide=True), ct)
