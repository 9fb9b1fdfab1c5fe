"""GPU parity of key generation, CKKS encryption and decryption through
the C ABI: every output is bit-identical to the CPU oracle's for the same DRBG seed and call
sequence, and the whole pipeline -- keygen -> encrypt -> multiply -> relinearize -> rescale ->
rotate -> decrypt -- run on the GPU alone returns the message."""
import numpy as np
import pytest

from he_math import RLWE, negacyclic_mul

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _pair(hg, oracle, n, log_q, log_p):
    c = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, log_p, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.CKKS, c.n_power, primes, len(log_q), len(log_p))
    c.upload()
    return c, o, primes


@pytest.mark.parametrize("n,log_q,log_p", [(4096, [50, 40, 40], [55]), (8192, [60, 45, 36, 50], [60])])
def test_keygen_encrypt_decrypt_bit_exact(hg, oracle, torch, n, log_q, log_p):
    c, o, primes = _pair(hg, oracle, n, log_q, log_p)
    Q = len(log_q)
    seed = 424242 + n
    rg, ro = hg.Rng(seed), oracle.ORng(seed)
    sk = c.generate_secret_key(rg)
    sk_o = o.gen_secret_key(ro)
    assert np.array_equal(hg.to_host(sk), sk_o), "secret key"
    pk = c.generate_public_key(rg, sk)
    pk_o = o.gen_public_key(ro, sk_o)
    assert np.array_equal(hg.to_host(pk), pk_o), "public key"
    rk = c.generate_relin_key(rg, sk)
    rk_o = o.gen_switch_key(ro, sk_o, 0)
    assert np.array_equal(hg.to_host(rk), rk_o), "relinearisation key"
    gal = hg.steps_to_galois_elt(3, n, 5)
    gk = c.generate_galois_key(rg, sk, gal)
    gk_o = o.gen_switch_key(ro, sk_o, gal)
    assert np.array_equal(hg.to_host(gk), gk_o), "galois key"
    plain = np.concatenate([oracle.fill_poly(9, j, n, primes[j]) for j in range(Q)])
    ct = c.ckks_encrypt(rg, pk, hg.to_device(plain))
    ct_o = o.ckks_encrypt(ro, pk_o, plain)
    assert np.array_equal(hg.to_host(ct), ct_o), "ciphertext"
    for depth in range(min(2, Q - 1) + 1):
        l = Q - depth
        ctl = np.concatenate([ct_o.reshape(2, Q, n)[p, :l].reshape(-1) for p in range(2)])
        dec = c.ckks_decrypt(hg.to_device(ctl), sk, depth)
        assert np.array_equal(hg.to_host(dec), o.ckks_decrypt(ctl, sk_o, depth)), f"decrypt depth {depth}"
    # a second generator with another seed gives other keys
    sk2 = c.generate_secret_key(hg.Rng(seed + 1))
    assert not np.array_equal(hg.to_host(sk2), sk_o)


def test_gpu_only_pipeline_returns_message(hg, oracle, torch):
    """keygen, encryption, multiply, relinearize, rescale, rotate, decrypt on the GPU; only the
    encoding of the test message (big-integer NTT) and the final CRT use the host helpers."""
    n = 4096
    c, o, primes = _pair(hg, oracle, n, [50, 30, 30, 30], [50])
    Q = 4
    rg = hg.Rng(7)
    sk = c.generate_secret_key(rg)
    pk = c.generate_public_key(rg, sk)
    rk = c.generate_relin_key(rg, sk)
    gal = hg.steps_to_galois_elt(1, n, 5)
    gk = c.generate_galois_key(rg, sk, gal)
    he = RLWE(o, seed=1)
    scale = 1 << 30
    g = np.random.default_rng(12)
    m1, m2 = g.integers(-8, 9, n), g.integers(-8, 9, n)
    p1 = he.to_ntt([int(v) * scale for v in m1], range(Q)).reshape(-1)
    p2 = he.to_ntt([int(v) * scale for v in m2], range(Q)).reshape(-1)
    ct1 = c.ckks_encrypt(rg, pk, hg.to_device(p1))
    ct2 = c.ckks_encrypt(rg, pk, hg.to_device(p2))

    def decode(dec, l):
        coeff = he.ntt_limbs(hg.to_host(dec).reshape(l, n), list(range(l)), inverse=True)
        return he.crt_centered(coeff, list(range(l)))[0]

    # fresh ciphertext
    x = decode(c.ckks_decrypt(ct1, sk, 0), Q)
    assert max(abs(int(a) - int(b) * scale) for a, b in zip(x, m1)) < 1 << 16
    # multiply + relinearize
    out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(ct1, 2 * Q * n, ct2, 2 * Q * n, out, 3 * Q * n, 0, 1)
    ws = c.workspace(hg.OP_CKKS_RELIN, 0, 1)
    c.ckks_relinearize_inplace(out, 3 * Q * n, rk, 0, 1, ws)
    prod = negacyclic_mul(m1, m2)
    x = decode(c.ckks_decrypt(out[:2 * Q * n].contiguous(), sk, 0), Q)
    assert max(abs(int(a) - int(b) * scale * scale) for a, b in zip(x, prod)) < scale * scale // 2 ** 8
    # rescale
    ws2 = c.workspace(hg.OP_CKKS_RESCALE, 0, 1)
    c.ckks_rescale_inplace(out, 3 * Q * n, 0, 1, ws2)
    l = Q - 1
    x = decode(c.ckks_decrypt(out[:2 * l * n].contiguous(), sk, 1), l)
    q_last = primes[Q - 1]
    assert max(abs(int(a) * q_last - int(b) * scale * scale) for a, b in zip(x, prod)) < scale * scale // 2 ** 8
    # rotate the fresh ciphertext
    rot = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
    ws3 = c.workspace(hg.OP_CKKS_GALOIS, 0, 1)
    c.ckks_apply_galois(ct1, 2 * Q * n, rot, 2 * Q * n, gk, gal, 0, 1, ws3)
    x = decode(c.ckks_decrypt(rot, sk, 0), Q)
    want = he.apply_galois_poly(np.array([int(v) * scale for v in m1], dtype=object), gal)
    assert max(abs(int(a) - int(b)) for a, b in zip(x, want)) < scale // 2 ** 8


def test_bfv_encrypt_decrypt_bit_exact_and_pipeline(hg, oracle, torch):
    """BFV (config C1 parameters): encryption/decryption bit-identical to the oracle for the same
    seed, then keygen -> encrypt -> multiply -> relinearize -> rotate -> decrypt on the GPU alone."""
    n, t = 4096, 1032193
    c = hg.Context.from_default(hg.BFV, n, 1, t)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.BFV, c.n_power, primes, c.Q_size, c.P_size, t)
    c.upload()
    Q = c.Q_size
    rg, ro = hg.Rng(31337), oracle.ORng(31337)
    sk, sk_o = c.generate_secret_key(rg), o.gen_secret_key(ro)
    pk, pk_o = c.generate_public_key(rg, sk), o.gen_public_key(ro, sk_o)
    assert np.array_equal(hg.to_host(pk), pk_o)
    g = np.random.default_rng(8)
    msgs = [g.integers(0, t, n).astype(np.uint64), np.full(n, t - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64)]
    for m in msgs:
        ct = c.bfv_encrypt(rg, pk, hg.to_device(m))
        ct_o = o.bfv_encrypt(ro, pk_o, m)
        assert np.array_equal(hg.to_host(ct), ct_o), "ciphertext"
        dec = c.bfv_decrypt(ct, sk)
        assert np.array_equal(hg.to_host(dec), o.bfv_decrypt(ct_o, sk_o)), "decryption == oracle"
        assert np.array_equal(hg.to_host(dec), m), "decryption == message"
    # GPU-only pipeline
    rk = c.generate_relin_key(rg, sk)
    gal = hg.steps_to_galois_elt(1, n, 3)
    gk = c.generate_galois_key(rg, sk, gal)
    m1, m2 = msgs[0], g.integers(0, t, n).astype(np.uint64)
    c1, c2 = c.bfv_encrypt(rg, pk, hg.to_device(m1)), c.bfv_encrypt(rg, pk, hg.to_device(m2))
    out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
    c.bfv_multiply(c1, 2 * Q * n, c2, 2 * Q * n, out, 3 * Q * n, 1, c.workspace(hg.OP_BFV_MULTIPLY, 0, 1))
    c.bfv_relinearize_inplace(out, 3 * Q * n, rk, 1, c.workspace(hg.OP_BFV_RELIN, 0, 1))
    got = hg.to_host(c.bfv_decrypt(out[:2 * Q * n].contiguous(), sk))
    want = np.array([int(v) % t for v in negacyclic_mul(m1, m2)], dtype=np.uint64)
    assert np.array_equal(got, want), "decrypt(relinearize(c1*c2)) = m1*m2 mod (X^N+1, t)"
    rot = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
    c.bfv_apply_galois(c1, 2 * Q * n, rot, 2 * Q * n, gk, gal, 1, c.workspace(hg.OP_BFV_GALOIS, 0, 1))
    he = RLWE(o, seed=0)
    want = np.array([int(v) % t for v in he.apply_galois_poly(m1.astype(object), gal)], dtype=np.uint64)
    assert np.array_equal(hg.to_host(c.bfv_decrypt(rot, sk)), want), "decrypt(rotate(c1)) = sigma_g(m1)"


@pytest.mark.parametrize("n,t", [(4096, 1032193), (8192, 65537)])
def test_bfv_batch_encoder_bit_exact(hg, oracle, torch, n, t):
    """encode / decode vs the oracle, decode(encode) = identity, short and negative messages,
    and the plain-modulus table set through hegpu_ntt."""
    c = hg.Context.from_default(hg.BFV, n, 1, t)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.BFV, c.n_power, primes, c.Q_size, c.P_size, t)
    c.upload()
    g = np.random.default_rng(3)
    for msg in (g.integers(0, t, n), g.integers(-t // 2, t // 2, n), np.array([-1, 5, -7, 3]), np.zeros(1)):
        msg = msg.astype(np.int64)
        plain = c.bfv_encode(torch.from_numpy(msg).cuda())
        want = o.bfv_encode(msg)
        assert np.array_equal(hg.to_host(plain), want), "encode"
        dec = c.bfv_decode(plain)
        assert np.array_equal(hg.to_host(dec), o.bfv_decode(want)), "decode"
        full = np.zeros(n, dtype=np.int64)
        full[:len(msg)] = msg
        assert np.array_equal(hg.to_host(dec), (full % t).astype(np.uint64)), "decode(encode(m)) = m mod t"


def test_switch_key_bit_exact_and_keyswitch(hg, oracle, torch):
    """generate_switch_key (ckks/keygenerator.cu:996-1095) against the oracle, then keyswitch
    (= the Galois path with the identity permutation, operator.cu switchkey_ckks_method_I) on the
    GPU against the oracle, and decryption under the new secret."""
    n = 4096
    c, o, primes = _pair(hg, oracle, n, [50, 30, 30, 30], [50])
    Q = 4
    rg, ro = hg.Rng(99), oracle.ORng(99)
    sk, sk_o = c.generate_secret_key(rg), o.gen_secret_key(ro)
    sk2, sk2_o = c.generate_secret_key(rg), o.gen_secret_key(ro)
    pk, pk_o = c.generate_public_key(rg, sk), o.gen_public_key(ro, sk_o)
    swk = c.generate_switch_key(rg, sk2, sk)
    swk_o = o.gen_switch_key_new_old(ro, sk2_o, sk_o)
    assert np.array_equal(hg.to_host(swk), swk_o), "switch key"
    he = RLWE(o, seed=1)
    scale = 1 << 30
    m = np.random.default_rng(4).integers(-50, 51, n)
    plain = he.to_ntt([int(v) * scale for v in m], range(Q)).reshape(-1)
    ct = c.ckks_encrypt(rg, pk, hg.to_device(plain))
    moved = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(ct, 2 * Q * n, moved, 2 * Q * n, swk, 1, 0, 1, c.workspace(hg.OP_CKKS_GALOIS, 0, 1))
    want = o.ckks_apply_galois(hg.to_host(ct), swk_o, 1, 0)
    assert np.array_equal(hg.to_host(moved), want), "keyswitch"
    dec = hg.to_host(c.ckks_decrypt(moved, sk2, 0))
    coeff = he.ntt_limbs(dec.reshape(Q, n), list(range(Q)), inverse=True)
    x = he.crt_centered(coeff, list(range(Q)))[0]
    assert max(abs(int(a) - int(b) * scale) for a, b in zip(x, m)) < 1 << 20


@pytest.mark.parametrize("scheme_name", ["ckks", "bfv"])
def test_method_II_key_generation_bit_exact(hg, oracle, torch, scheme_name):
    """relinkey_gen_II / galoiskey_gen_II / switchkey_gen_II (keygeneration.cu:584-629, :807-858,
    :941-989) with two special primes against the oracle; the generated relinearisation key then
    drives the method II relinearisation on both sides."""
    n = 4096
    if scheme_name == "ckks":
        c = hg.Context.from_bit_sizes(hg.CKKS, n, [50, 36, 36, 36, 36], [50, 50], sec=hg.SEC_NONE)
        sch = oracle.CKKS
    else:
        c = hg.Context.from_bit_sizes(hg.BFV, n, [40, 40, 40], [41, 41], plain_modulus=65537, sec=hg.SEC_NONE)
        sch = oracle.BFV
    primes = [int(x) for x in c.table("modulus")]
    Q, P = c.Q_size, c.P_size
    o = oracle.OracleContext(sch, c.n_power, primes, Q, P, 65537 if scheme_name == "bfv" else 0)
    c.upload()
    assert c.switch_key_digits() == o.switch_key_digits() == (3 if scheme_name == "ckks" else 2)
    rg, ro = hg.Rng(5150), oracle.ORng(5150)
    sk, sk_o = c.generate_secret_key(rg), o.gen_secret_key(ro)
    assert np.array_equal(hg.to_host(sk), sk_o)
    rk, rk_o = c.generate_relin_key(rg, sk), o.gen_switch_key(ro, sk_o, 0)
    assert np.array_equal(hg.to_host(rk), rk_o), "relinearisation key (method II)"
    gal = hg.steps_to_galois_elt(1, n, 5 if scheme_name == "ckks" else 3)
    gk, gk_o = c.generate_galois_key(rg, sk, gal), o.gen_switch_key(ro, sk_o, gal)
    assert np.array_equal(hg.to_host(gk), gk_o), "galois key (method II)"
    sk2, sk2_o = c.generate_secret_key(rg), o.gen_secret_key(ro)
    swk, swk_o = c.generate_switch_key(rg, sk2, sk), o.gen_switch_key_new_old(ro, sk2_o, sk_o)
    assert np.array_equal(hg.to_host(swk), swk_o), "switch key (method II)"
    pk, pk_o = c.generate_public_key(rg, sk), o.gen_public_key(ro, sk_o)
    assert np.array_equal(hg.to_host(pk), pk_o), "public key over two special primes"
    if scheme_name == "ckks":
        plain = np.concatenate([oracle.fill_poly(3, j, n, primes[j]) for j in range(Q)])
        ct = c.ckks_encrypt(rg, pk, hg.to_device(plain))
        ct_o = o.ckks_encrypt(ro, pk_o, plain)
        assert np.array_equal(hg.to_host(ct), ct_o), "encryption with P_size = 2"
        out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
        c.ckks_multiply(ct, 2 * Q * n, ct, 2 * Q * n, out, 3 * Q * n, 0, 1)
        c.ckks_relinearize_inplace(out, 3 * Q * n, rk, 0, 1, c.workspace(hg.OP_CKKS_RELIN, 0, 1))
        ct3 = o.ckks_multiply(ct_o, ct_o, 0)
        o.ckks_relinearize_II(ct3, rk_o, 0)
        assert np.array_equal(hg.to_host(out)[:2 * Q * n], ct3[:2 * Q * n]), "relinearize with the generated key"
        rot = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
        c.ckks_apply_galois(ct, 2 * Q * n, rot, 2 * Q * n, gk, gal, 0, 1, c.workspace(hg.OP_CKKS_GALOIS, 0, 1))
        assert np.array_equal(hg.to_host(rot), o.ckks_apply_galois_II(ct_o, gk_o, gal, 0)), "rotate with the generated key"


def test_bfv_plain_ops_bit_exact(hg, oracle, torch):
    """multiply_plain (bfv/operator.cu:432-503), transform_to_ntt of a plaintext (:1398-1431) and
    multiply_power_of_X (switchkey.cu:1433-1457) against the oracle."""
    n, t = 8192, 65537
    c = hg.Context.from_bit_sizes(hg.BFV, n, [54, 54, 54], [55], plain_modulus=t, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    Q = 3
    o = oracle.OracleContext(oracle.BFV, c.n_power, primes, Q, 1, t)
    c.upload()
    g = np.random.default_rng(2)
    plain = g.integers(0, t, n).astype(np.uint64)
    plain[:4] = [0, t - 1, (t + 1) // 2, (t + 1) // 2 - 1]       # both sides of the threshold
    ct = np.concatenate([oracle.fill_poly(40 + p, j, n, primes[j]) for p in range(2) for j in range(Q)])
    assert np.array_equal(hg.to_host(c.bfv_plain_to_ntt(hg.to_device(plain))), o.bfv_plain_to_ntt(plain))
    out = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
    ws = c.workspace(hg.OP_BFV_MULTIPLY_PLAIN, 0, 1)
    d_ct, d_plain = hg.to_device(ct), hg.to_device(plain)
    rc = hg._lib.load().hegpu_bfv_multiply_plain(c._h, d_ct.data_ptr(), d_plain.data_ptr(), out.data_ptr(),
                                                 ws.data_ptr(), ws.numel() * 8, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert np.array_equal(hg.to_host(out), o.bfv_multiply_plain(ct, plain))
    for k in (0, 1, 100, n - 1, n, 2 * n - 1):
        for parts, l in ((2, Q), (3, Q), (1, 2)):
            sub = ct[:parts * l * n] if parts * l * n <= ct.size else np.concatenate([ct, ct[:Q * n]])
            got = hg.to_host(c.negacyclic_shift(hg.to_device(sub), k, l, parts))
            assert np.array_equal(got, o.negacyclic_shift(sub, k, l, parts)), (k, parts, l)
