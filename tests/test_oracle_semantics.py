"""Semantic pin of the oracle (CPU): encrypt -> op -> decrypt round trips in
the shape of the reference's own gtest suite (BFV exact mod t,
test/test_bfv_multiplication.cpp; CKKS approximate, test_ckks_relinearization.cpp;
rotations test_*_rotation_method_1.cpp).  The reference holds no golden
vectors, so this is what ties the restated kernel sequences to FHE semantics:
a wrong constant, index table or rounding step makes decryption fail.
"""
import numpy as np
import pytest

from he_math import RLWE, negacyclic_mul


@pytest.fixture(scope="module")
def bfv(oracle):
    import ctypes
    t = 1032193
    primes = (ctypes.c_uint64 * 8)()
    cnt = oracle.lib().o_default_modulus_128(4096, primes)
    primes = [int(primes[i]) for i in range(cnt)]
    o = oracle.OracleContext(oracle.BFV, 12, primes, 2, 1, t)
    return o, RLWE(o, seed=1), t


def _bfv_encode(m, M, t):
    delta = M // t
    return [int(v) * delta for v in m]


def _bfv_decode(x, M, t):
    # round(t * x / M) mod t
    return np.array([((2 * t * int(v) + M) // (2 * M)) % t for v in x], dtype=np.int64)


def test_bfv_multiply_relinearize_decrypts(bfv, oracle):
    o, he, t = bfv
    n, Q = o.n, o.Q
    M = 1
    for j in range(Q):
        M *= o.primes[j]
    rng = np.random.default_rng(5)
    m1 = rng.integers(0, t, n)
    m2 = rng.integers(0, t, n)
    ct1 = he.encrypt(_bfv_encode(m1, M, t), Q, ntt_domain=False)
    ct2 = he.encrypt(_bfv_encode(m2, M, t), Q, ntt_domain=False)
    x, _ = he.decrypt(ct1, Q, 2, ntt_domain=False)
    assert np.array_equal(_bfv_decode(x, M, t), m1)
    want = np.array([int(v) % t for v in negacyclic_mul(m1, m2)], dtype=np.int64)
    ct3 = o.bfv_multiply(ct1, ct2)
    x, _ = he.decrypt(ct3, Q, 3, ntt_domain=False)
    assert np.array_equal(_bfv_decode(x, M, t), want), "3-part product decrypts to m1*m2 mod t"
    rk = he.relin_key()
    o.bfv_relinearize(ct3, rk)
    x, _ = he.decrypt(ct3[:2 * Q * n], Q, 2, ntt_domain=False)
    assert np.array_equal(_bfv_decode(x, M, t), want), "relinearized product decrypts to m1*m2 mod t"


@pytest.mark.parametrize("steps", [1, -2])
def test_bfv_rotate_decrypts(bfv, oracle, steps):
    o, he, t = bfv
    n, Q = o.n, o.Q
    M = o.primes[0] * o.primes[1]
    rng = np.random.default_rng(6)
    m = rng.integers(0, t, n)
    ct = he.encrypt(_bfv_encode(m, M, t), Q, ntt_domain=False)
    g = oracle.lib().o_steps_to_galois_elt(steps, n, 3)
    gk = he.galois_key(g)
    out = o.bfv_apply_galois(ct, gk, g)
    x, _ = he.decrypt(out, Q, 2, ntt_domain=False)
    want = np.array([int(v) % t for v in he.apply_galois_poly(m.astype(object), g)], dtype=np.int64)
    assert np.array_equal(_bfv_decode(x, M, t), want)


@pytest.fixture(scope="module")
def ckks(oracle):
    import ctypes
    bits = (ctypes.c_int * 4)(40, 30, 30, 40)
    out = (ctypes.c_uint64 * 4)()
    assert oracle.lib().o_generate_primes(4096, bits, 4, out) == 0
    primes = [int(v) for v in out]
    o = oracle.OracleContext(oracle.CKKS, 12, primes, 3, 1)
    return o, RLWE(o, seed=2)


def test_ckks_mul_relin_rescale_decrypts(ckks):
    o, he = ckks
    n, Q = o.n, o.Q
    scale = 1 << 30
    rng = np.random.default_rng(8)
    m1 = rng.integers(-8, 9, n)
    m2 = rng.integers(-8, 9, n)
    ct1 = he.encrypt([int(v) * scale for v in m1], Q, ntt_domain=True)
    ct2 = he.encrypt([int(v) * scale for v in m2], Q, ntt_domain=True)
    prod = negacyclic_mul(m1, m2)
    ct3 = o.ckks_multiply(ct1, ct2, 0)
    x, M = he.decrypt(ct3, Q, 3, ntt_domain=True)
    err = max(abs(int(a) - int(b) * scale * scale) for a, b in zip(x, prod))
    assert err < scale * scale // 2 ** 10
    o.ckks_relinearize(ct3, he.relin_key(), 0)
    x, M = he.decrypt(ct3[:2 * Q * n], Q, 2, ntt_domain=True)
    err = max(abs(int(a) - int(b) * scale * scale) for a, b in zip(x, prod))
    assert err < scale * scale // 2 ** 10, "relinearization keeps the message"
    ct = ct3[:2 * Q * n].copy()
    o.ckks_rescale(ct, 0)
    x, M = he.decrypt(ct[:2 * (Q - 1) * n], Q - 1, 2, ntt_domain=True)
    q_last = o.primes[Q - 1]
    err = max(abs(int(a) * q_last - int(b) * scale * scale) for a, b in zip(x, prod))
    assert err < scale * scale // 2 ** 10, "rescale divides by q_last with rounding"
    # leveled path (depth 1): fresh ciphertexts on Q-1 limbs, smaller scale so
    # that the product stays below q_0*q_1
    l = Q - 1
    sc = 1 << 20
    a1 = rng.integers(-2, 3, n)
    a2 = rng.integers(-2, 3, n)
    c1 = he.encrypt([int(v) * sc for v in a1], l, ntt_domain=True)
    c2 = he.encrypt([int(v) * sc for v in a2], l, ntt_domain=True)
    ct3 = o.ckks_multiply(c1, c2, 1)
    o.ckks_relinearize(ct3, he.relin_key(), 1)
    x, M = he.decrypt(ct3[:2 * l * n], l, 2, ntt_domain=True)
    pr = negacyclic_mul(a1, a2)
    err = max(abs(int(a) - int(b) * sc * sc) for a, b in zip(x, pr))
    assert err < sc * sc // 2 ** 6, "leveled (depth 1) multiply+relinearize keeps the message"
    ct = ct3[:2 * l * n].copy()
    o.ckks_rescale(ct, 1)
    x, M = he.decrypt(ct[:2 * (l - 1) * n], l - 1, 2, ntt_domain=True)
    q_last = o.primes[l - 1]
    err = max(abs(int(a) * q_last - int(b) * sc * sc) for a, b in zip(x, pr))
    # rescale adds a rounding term tau0 + tau1*s, |tau| <= 1/2 (times q_last here)
    assert err < 4 * n * q_last // 8, "leveled rescale"


@pytest.mark.parametrize("depth", [0, 1])
def test_ckks_rotate_decrypts(ckks, oracle, depth):
    o, he = ckks
    n, Q = o.n, o.Q
    l = Q - depth
    scale = 1 << 30
    rng = np.random.default_rng(9)
    m = rng.integers(-100, 101, n)
    ct = he.encrypt([int(v) * scale for v in m], l, ntt_domain=True)
    g = oracle.lib().o_steps_to_galois_elt(3, n, 5)
    out = o.ckks_apply_galois(ct, he.galois_key(g), g, depth)
    x, M = he.decrypt(out, l, 2, ntt_domain=True)
    want = he.apply_galois_poly(m.astype(object), g)
    err = max(abs(int(a) - int(b) * scale) for a, b in zip(x, want))
    assert err < scale // 2 ** 8


# ---------------------------------------------------------------- key-switching method II (P_size > 1)
@pytest.fixture(scope="module")
def ckks2(oracle):
    """reference test_ckks_relinearization.cpp:438 style: two special primes"""
    import ctypes
    bits = [40, 30, 30, 30, 30, 40, 40]
    arr = (ctypes.c_int * len(bits))(*bits)
    out = (ctypes.c_uint64 * len(bits))()
    assert oracle.lib().o_generate_primes(4096, arr, len(bits), out) == 0
    o = oracle.OracleContext(oracle.CKKS, 12, [int(v) for v in out], 5, 2)
    return o, RLWE(o, seed=3)


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_ckks_method_II_relin_and_rotate(ckks2, oracle, depth):
    o, he = ckks2
    n, Q = o.n, o.Q
    l = Q - depth
    sc = 1 << 25
    rng = np.random.default_rng(20 + depth)
    m1, m2 = rng.integers(-4, 5, n), rng.integers(-4, 5, n)
    c1 = he.encrypt([int(v) * sc for v in m1], l, ntt_domain=True)
    c2 = he.encrypt([int(v) * sc for v in m2], l, ntt_domain=True)
    ct3 = o.ckks_multiply(c1, c2, depth)
    o.ckks_relinearize_II(ct3, he.relin_key_II(o.P), depth)
    x, M = he.decrypt(ct3[:2 * l * n], l, 2, ntt_domain=True)
    pr = negacyclic_mul(m1, m2)
    err = max(abs(int(a) - int(b) * sc * sc) for a, b in zip(x, pr))
    assert err < sc * sc // 2 ** 6, "method II relinearization keeps the message"
    g = oracle.lib().o_steps_to_galois_elt(2, n, 5)
    out = o.ckks_apply_galois_II(c1, he.galois_key_II(g, o.P), g, depth)
    x, M = he.decrypt(out, l, 2, ntt_domain=True)
    want = he.apply_galois_poly(m1.astype(object), g)
    err = max(abs(int(a) - int(b) * sc) for a, b in zip(x, want))
    assert err < sc // 2 ** 6, "method II rotation"


def test_bfv_method_II_relin_and_rotate(oracle):
    """reference test_bfv_relinearization.cpp:434 style: Q={36,36,36} P={37,37}, digits of m=2"""
    import ctypes
    t = 1032193
    bits = [36, 36, 36, 37, 37]
    arr = (ctypes.c_int * len(bits))(*bits)
    out = (ctypes.c_uint64 * len(bits))()
    assert oracle.lib().o_generate_primes(4096, arr, len(bits), out) == 0
    o = oracle.OracleContext(oracle.BFV, 12, [int(v) for v in out], 3, 2, t)
    he = RLWE(o, seed=4)
    n, Q = o.n, o.Q
    M = 1
    for j in range(Q):
        M *= o.primes[j]
    rng = np.random.default_rng(31)
    m1, m2 = rng.integers(0, t, n), rng.integers(0, t, n)
    ct1 = he.encrypt(_bfv_encode(m1, M, t), Q, ntt_domain=False)
    ct2 = he.encrypt(_bfv_encode(m2, M, t), Q, ntt_domain=False)
    ct3 = o.bfv_multiply(ct1, ct2)
    o.bfv_relinearize_II(ct3, he.relin_key_II(2))
    x, _ = he.decrypt(ct3[:2 * Q * n], Q, 2, ntt_domain=False)
    want = np.array([int(v) % t for v in negacyclic_mul(m1, m2)], dtype=np.int64)
    assert np.array_equal(_bfv_decode(x, M, t), want)
    g = oracle.lib().o_steps_to_galois_elt(3, n, 3)
    out_ct = o.bfv_apply_galois_II(ct1, he.galois_key_II(g, 2), g)
    x, _ = he.decrypt(out_ct, Q, 2, ntt_domain=False)
    want = np.array([int(v) % t for v in he.apply_galois_poly(m1.astype(object), g)], dtype=np.int64)
    assert np.array_equal(_bfv_decode(x, M, t), want)
