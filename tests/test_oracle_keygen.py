"""Semantic checks of the oracle's key generation / CKKS encryption / decryption
restatement (oracle/o_keygen.c) in the shape of the reference's own tests
(encrypt -> operate -> decrypt): the generated keys must be valid RLWE samples under the
generated secret, the public-key encryption must decrypt to the message, and the generated
relinearisation / Galois keys must work with the operator restatement."""
import numpy as np
import pytest

from he_math import RLWE, negacyclic_mul


@pytest.fixture(scope="module")
def setup(oracle):
    import ctypes
    n_power = 10
    n = 1 << n_power
    bits = [40, 30, 30]
    arr = (ctypes.c_int * 4)(*(bits + [40]))
    out = (ctypes.c_uint64 * 4)()
    assert oracle.lib().o_generate_primes(n, arr, 4, out) == 0
    primes = [int(v) for v in out]
    o = oracle.OracleContext(oracle.CKKS, n_power, primes, len(bits), 1)
    rng = oracle.ORng(20260928)
    sk = o.gen_secret_key(rng)
    he = RLWE(o, seed=1)
    # adopt the generated secret: coefficients from the inverse NTT of limb 0
    s0 = np.ascontiguousarray(sk.reshape(o.Qp, n)[0].copy())
    o.ntt(s0, 1, 1, mod_offset=0, inverse=True)
    q0 = primes[0]
    he.s = np.array([int(v) - q0 if int(v) > q0 // 2 else int(v) for v in s0])
    he.s_ntt = sk.reshape(o.Qp, n).copy()
    return o, he, rng, sk, primes


def test_secret_key_shape(setup):
    o, he, rng, sk, primes = setup
    assert set(np.unique(he.s)) <= {-1, 0, 1}
    assert int(np.count_nonzero(he.s)) == o.n // 2          # secretkey.cu:23 default hamming weight
    # every limb is the NTT of the same ternary polynomial
    assert np.array_equal(he.to_ntt(he.s, range(o.Qp)), sk.reshape(o.Qp, o.n))


def test_public_key_is_rlwe_sample(setup):
    o, he, rng, sk, primes = setup
    pk = o.gen_public_key(rng, sk)
    # pk0 + pk1*s = -e: a "ciphertext" of zero whose decryption is the negated error
    x, M = he.decrypt(pk, o.Qp, 2, ntt_domain=True)
    e = np.array([int(v) for v in x])
    assert np.max(np.abs(e)) <= 19, "clipped at 6 sigma"
    assert 2.6 < np.std(e) < 3.8 and abs(np.mean(e)) < 0.5
    a = pk.reshape(2, o.Qp, o.n)[1]
    for j in range(o.Qp):
        assert int(a[j].max()) < primes[j]
        assert abs(float(np.mean(a[j].astype(np.float64))) / primes[j] - 0.5) < 0.05


def test_encrypt_decrypt_round_trip(setup):
    o, he, rng, sk, primes = setup
    n, Q = o.n, o.Q
    pk = o.gen_public_key(rng, sk)
    scale = 1 << 30
    m = np.random.default_rng(3).integers(-100, 101, n)
    plain = he.to_ntt([int(v) * scale for v in m], range(Q)).reshape(-1)
    ct = o.ckks_encrypt(rng, pk, plain)
    dec = o.ckks_decrypt(ct, sk)
    coeff = he.ntt_limbs(dec.reshape(Q, n), list(range(Q)), inverse=True)
    x, M = he.crt_centered(coeff, list(range(Q)))
    err = max(abs(int(a) - int(b) * scale) for a, b in zip(x, m))
    # u*e_pk + e0 + e1*s after the division by P: a few hundred at most for n = 1024
    assert err < 1 << 14
    # two encryptions of the same plaintext differ (fresh streams)
    ct2 = o.ckks_encrypt(rng, pk, plain)
    assert not np.array_equal(ct, ct2)


def test_generated_relin_key_works(setup):
    o, he, rng, sk, primes = setup
    n, Q = o.n, o.Q
    rk = o.gen_switch_key(rng, sk, 0)
    scale = 1 << 25
    g = np.random.default_rng(5)
    m1, m2 = g.integers(-8, 9, n), g.integers(-8, 9, n)
    ct1 = he.encrypt([int(v) * scale for v in m1], Q, ntt_domain=True)
    ct2 = he.encrypt([int(v) * scale for v in m2], Q, ntt_domain=True)
    ct3 = o.ckks_multiply(ct1, ct2, 0)
    o.ckks_relinearize(ct3, rk, 0)
    x, M = he.decrypt(ct3[:2 * Q * n], Q, 2, ntt_domain=True)
    prod = negacyclic_mul(m1, m2)
    err = max(abs(int(a) - int(b) * scale * scale) for a, b in zip(x, prod))
    assert err < scale * scale // 2 ** 10


def test_generated_galois_key_works(setup, oracle):
    o, he, rng, sk, primes = setup
    n, Q = o.n, o.Q
    gal = oracle.lib().o_steps_to_galois_elt(1, n, 5)
    gk = o.gen_switch_key(rng, sk, gal)
    scale = 1 << 30
    m = np.random.default_rng(6).integers(-50, 51, n)
    ct = he.encrypt([int(v) * scale for v in m], Q, ntt_domain=True)
    rot = o.ckks_apply_galois(ct, gk, gal, 0)
    x, M = he.decrypt(rot, Q, 2, ntt_domain=True)
    want = he.apply_galois_poly(np.array([int(v) * scale for v in m], dtype=object), gal)
    err = max(abs(int(a) - int(b)) for a, b in zip(x, want))
    assert err < scale // 2 ** 8


def _adopt(he_cls, o, sk, primes):
    """RLWE helper whose secret is a generated key (coefficients from the inverse NTT of limb 0)"""
    he = he_cls(o, seed=1)
    s0 = np.ascontiguousarray(sk.reshape(o.Qp, o.n)[0].copy())
    o.ntt(s0, 1, 1, mod_offset=0, inverse=True)
    q0 = primes[0]
    he.s = np.array([int(v) - q0 if int(v) > q0 // 2 else int(v) for v in s0])
    he.s_ntt = sk.reshape(o.Qp, o.n).copy()
    return he


def test_generated_switch_key_moves_ciphertext_to_the_new_secret(setup, oracle):
    """generate_switch_key + keyswitch (ckks/keygenerator.cu:996-1095, operator.cu
    switchkey_ckks_method_I = the Galois path with the identity permutation): a ciphertext under
    sk decrypts under sk2 after the switch, and no longer under sk."""
    o, he, rng, sk, primes = setup
    n, Q = o.n, o.Q
    sk2 = o.gen_secret_key(rng)
    he2 = _adopt(RLWE, o, sk2, primes)
    swk = o.gen_switch_key_new_old(rng, sk2, sk)
    scale = 1 << 30
    m = np.random.default_rng(8).integers(-50, 51, n)
    ct = he.encrypt([int(v) * scale for v in m], Q, ntt_domain=True)
    moved = o.ckks_apply_galois(ct, swk, 1, 0)
    x, M = he2.decrypt(moved, Q, 2, ntt_domain=True)
    assert max(abs(int(a) - int(b) * scale) for a, b in zip(x, m)) < scale // 2 ** 8
    y, M = he.decrypt(moved, Q, 2, ntt_domain=True)
    assert max(abs(int(a) - int(b) * scale) for a, b in zip(y, m)) > scale * 2 ** 8


@pytest.fixture(scope="module")
def setup_m2(oracle):
    import ctypes
    n_power = 10
    n = 1 << n_power
    bits = [40, 30, 30, 30, 30]
    arr = (ctypes.c_int * 7)(*(bits + [40, 40]))
    out = (ctypes.c_uint64 * 7)()
    assert oracle.lib().o_generate_primes(n, arr, 7, out) == 0
    primes = [int(v) for v in out]
    o = oracle.OracleContext(oracle.CKKS, n_power, primes, len(bits), 2)
    rng = oracle.ORng(31337)
    sk = o.gen_secret_key(rng)
    return o, _adopt(RLWE, o, sk, primes), rng, sk, primes


def test_method_II_generated_keys_work(setup_m2, oracle):
    """relinkey_gen_II_kernel / galoiskey_gen_II_kernel (keygeneration.cu:584-629, :807-858) with
    two special primes: 3 digits of 2 primes for Q = 5; the keys drive the method II operators."""
    o, he, rng, sk, primes = setup_m2
    n, Q = o.n, o.Q
    assert o.switch_key_digits() == 3
    rk = o.gen_switch_key(rng, sk, 0)
    assert rk.size == 3 * 2 * o.Qp * n
    scale = 1 << 25
    g = np.random.default_rng(15)
    m1, m2 = g.integers(-8, 9, n), g.integers(-8, 9, n)
    ct1 = he.encrypt([int(v) * scale for v in m1], Q, ntt_domain=True)
    ct2 = he.encrypt([int(v) * scale for v in m2], Q, ntt_domain=True)
    ct3 = o.ckks_multiply(ct1, ct2, 0)
    o.ckks_relinearize_II(ct3, rk, 0)
    x, M = he.decrypt(ct3[:2 * Q * n], Q, 2, ntt_domain=True)
    prod = negacyclic_mul(m1, m2)
    assert max(abs(int(a) - int(b) * scale * scale) for a, b in zip(x, prod)) < scale * scale // 2 ** 10
    gal = oracle.lib().o_steps_to_galois_elt(2, n, 5)
    gk = o.gen_switch_key(rng, sk, gal)
    sc = 1 << 30
    ct = he.encrypt([int(v) * sc for v in m1], Q, ntt_domain=True)
    rot = o.ckks_apply_galois_II(ct, gk, gal, 0)
    x, M = he.decrypt(rot, Q, 2, ntt_domain=True)
    want = he.apply_galois_poly(np.array([int(v) * sc for v in m1], dtype=object), gal)
    assert max(abs(int(a) - int(b)) for a, b in zip(x, want)) < sc // 2 ** 8
    # switch key, method II
    sk2 = o.gen_secret_key(rng)
    he2 = _adopt(RLWE, o, sk2, primes)
    swk = o.gen_switch_key_new_old(rng, sk2, sk)
    moved = o.ckks_apply_galois_II(ct, swk, 1, 0)
    x, M = he2.decrypt(moved, Q, 2, ntt_domain=True)
    assert max(abs(int(a) - int(b) * sc) for a, b in zip(x, m1)) < sc // 2 ** 8


def test_streams_are_reproducible(setup, oracle):
    o, he, rng, sk, primes = setup
    a = o.gen_secret_key(oracle.ORng(77))
    b = o.gen_secret_key(oracle.ORng(77))
    c = o.gen_secret_key(oracle.ORng(78))
    assert np.array_equal(a, b) and not np.array_equal(a, c)


# ---------------------------------------------------------------- BFV
@pytest.fixture(scope="module")
def bfv(oracle):
    import ctypes
    n_power, t = 10, 65537
    n = 1 << n_power
    arr = (ctypes.c_int * 3)(36, 36, 37)
    out = (ctypes.c_uint64 * 3)()
    assert oracle.lib().o_generate_primes(n, arr, 3, out) == 0
    primes = [int(v) for v in out]
    o = oracle.OracleContext(oracle.BFV, n_power, primes, 2, 1, t)
    rng = oracle.ORng(99)
    sk = o.gen_secret_key(rng)
    pk = o.gen_public_key(rng, sk)
    return o, rng, sk, pk, t


def test_bfv_encrypt_decrypt_exact(bfv):
    o, rng, sk, pk, t = bfv
    g = np.random.default_rng(2)
    for m in (g.integers(0, t, o.n).astype(np.uint64), np.zeros(o.n, dtype=np.uint64),
              np.full(o.n, t - 1, dtype=np.uint64)):
        ct = o.bfv_encrypt(rng, pk, m)
        assert np.array_equal(o.bfv_decrypt(ct, sk), m)


def test_bfv_homomorphic_multiply_with_generated_keys(bfv):
    """encrypt -> multiply -> relinearize (generated key) -> decrypt = m1*m2 mod (X^N+1, t)"""
    o, rng, sk, pk, t = bfv
    n = o.n
    rk = o.gen_switch_key(rng, sk, 0)
    g = np.random.default_rng(4)
    m1 = g.integers(0, t, n).astype(np.uint64)
    m2 = g.integers(0, t, n).astype(np.uint64)
    ct3 = o.bfv_multiply(o.bfv_encrypt(rng, pk, m1), o.bfv_encrypt(rng, pk, m2))
    o.bfv_relinearize(ct3, rk)
    got = o.bfv_decrypt(ct3[:2 * o.Q * n].copy(), sk)
    want = np.array([int(v) % t for v in negacyclic_mul(m1, m2)], dtype=np.uint64)
    assert np.array_equal(got, want)
    # rotation with a generated Galois key
    from he_math import RLWE
    gal = 3
    gk = o.gen_switch_key(rng, sk, gal)
    rot = o.bfv_apply_galois(o.bfv_encrypt(rng, pk, m1), gk, gal)
    he = RLWE(o, seed=0)
    want = np.array([int(v) % t for v in he.apply_galois_poly(m1.astype(object), gal)], dtype=np.uint64)
    assert np.array_equal(o.bfv_decrypt(rot, sk), want)


def test_bfv_batch_encoder_semantics(bfv):
    """The encoder is pinned by what batching means: decode inverts encode, polynomial
    multiplication mod (X^N+1, t) is slot-wise multiplication, and X -> X^3 rotates both rows
    one slot to the left (the reference's rotate_rows semantics, bfv/evaluationkey.cu:308)."""
    from he_math import RLWE
    o, rng, sk, pk, t = bfv
    n = o.n
    g = np.random.default_rng(10)
    a = g.integers(0, t, n)
    b = g.integers(0, t, n)
    pa, pb = o.bfv_encode(a), o.bfv_encode(b)
    assert np.array_equal(o.bfv_decode(pa), a.astype(np.uint64))
    prod = np.array([int(v) % t for v in negacyclic_mul(pa, pb)], dtype=np.uint64)
    assert np.array_equal(o.bfv_decode(prod), (a * b % t).astype(np.uint64))
    he = RLWE(o, seed=0)
    rot = np.array([int(v) % t for v in he.apply_galois_poly(pa.astype(object), 3)], dtype=np.uint64)
    want = np.concatenate([np.roll(a[:n // 2], -1), np.roll(a[n // 2:], -1)]).astype(np.uint64)
    assert np.array_equal(o.bfv_decode(rot), want)
    # short and negative messages
    short = np.array([-1, 5, -7], dtype=np.int64)
    dec = o.bfv_decode(o.bfv_encode(short))
    assert list(dec[:3]) == [t - 1, 5, t - 7] and not dec[3:].any()
    # end to end with encryption: encode -> encrypt -> decrypt -> decode
    ct = o.bfv_encrypt(rng, pk, pa)
    assert np.array_equal(o.bfv_decode(o.bfv_decrypt(ct, sk)), a.astype(np.uint64))


def test_bfv_multiply_plain_and_power_of_x(bfv):
    """multiply_plain (bfv/operator.cu:432-503), transform_to_ntt of a plaintext (:1398-1431) and
    multiply_power_of_X (switchkey.cu:1433-1457), pinned by decryption: Enc(m) * p decrypts to the
    negacyclic product mod t (centred lift of p), the NTT-domain route gives the same ciphertext, and
    X^k shifts the message polynomial negacyclically."""
    o, rng, sk, pk, t = bfv
    n, Q = o.n, o.Q
    g = np.random.default_rng(21)
    m = g.integers(0, t, n).astype(np.uint64)
    p = np.zeros(n, dtype=np.uint64)
    p[0], p[1], p[5] = 3, t - 2, 7          # 3 - 2X + 7X^5
    ct = o.bfv_encrypt(rng, pk, m)
    prod = o.bfv_multiply_plain(ct, p)
    want = np.array([(3 * int(m[i]) - 2 * (int(m[i - 1]) if i >= 1 else -int(m[n - 1]))
                      + 7 * (int(m[i - 5]) if i >= 5 else -int(m[n + i - 5]))) % t for i in range(n)], dtype=np.uint64)
    assert np.array_equal(o.bfv_decrypt(prod, sk), want)
    # NTT-domain route: transform both, multiply pointwise, transform back
    pn = o.bfv_plain_to_ntt(p)
    ctn = ct.copy()
    o.ntt(ctn, 2 * Q, Q)
    primes = o.primes
    for z in range(2):
        for j in range(Q):
            seg = slice((z * Q + j) * n, (z * Q + j + 1) * n)
            ctn[seg] = np.array([(int(a) * int(b)) % primes[j] for a, b in zip(ctn[seg], pn[j * n:(j + 1) * n])],
                                dtype=np.uint64)
    o.ntt(ctn, 2 * Q, Q, inverse=True)
    assert np.array_equal(ctn, prod)
    for k in (1, 17, n - 1, n + 3):
        sh = o.negacyclic_shift(ct, k, Q)
        dec = o.bfv_decrypt(sh, sk)
        want = np.zeros(n, dtype=np.uint64)
        for i in range(n):
            r = i + k
            want[r % n] = m[i] if (r // n) % 2 == 0 else (t - int(m[i])) % t
        assert np.array_equal(dec, want), k
