"""GPU parity of the CKKS encoder / decoder (special FFT + RNS conversion + CRT composition)
through the C ABI.  FP64 with the oracle's operation order and no FMA contraction: the encoded
residues are identical and the decoded doubles agree to the last bit; then the reference's
basic CKKS flow (encode -> encrypt -> multiply -> relinearize -> rescale -> rotate -> decrypt ->
decode) runs on the GPU alone and returns the slot-wise results."""
import numpy as np
import pytest

from helpers import synth_ct

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


@pytest.mark.parametrize("n,log_q", [(4096, [50, 40, 40]), (16384, [60, 50, 50, 50, 50])])
def test_encode_decode_match_oracle(hg, oracle, torch, n, log_q):
    c, o, primes = _pair(hg, oracle, n, log_q, [60])
    slots = n // 2
    g = np.random.default_rng(n)
    scale = 2.0 ** 40
    for msg in (g.uniform(-100, 100, slots), g.uniform(-1, 1, 7), np.array([0.0]), -g.uniform(0, 1e6, slots)):
        plain = c.ckks_encode(torch.from_numpy(np.ascontiguousarray(msg)).cuda(), scale)
        want = o.ckks_encode(msg, scale)
        assert np.array_equal(hg.to_host(plain), want), "encoded residues"
        for depth in (0, len(log_q) - 1):
            l = len(log_q) - depth
            sub = np.ascontiguousarray(want.reshape(len(log_q), n)[:l].reshape(-1))
            dec = c.ckks_decode(hg.to_device(sub), scale, depth).cpu().numpy()
            ref = o.ckks_decode(sub, scale, depth)
            assert np.array_equal(dec, ref), f"decoded doubles, depth {depth}"
        dec = c.ckks_decode(plain, scale, 0).cpu().numpy()   # full chain: no wrap-around
        full = np.zeros(slots)
        full[:len(msg)] = msg
        assert np.max(np.abs(dec - full)) < 1e-6 * max(1.0, np.max(np.abs(full)))


def test_basic_ckks_flow_on_gpu(hg, oracle, torch):
    n = 8192
    c, o, primes = _pair(hg, oracle, n, [60, 40, 40, 40], [60])
    Q, slots = 4, n // 2
    rg = hg.Rng(5)
    sk = c.generate_secret_key(rg)
    pk = c.generate_public_key(rg, sk)
    rk = c.generate_relin_key(rg, sk)
    gal = hg.steps_to_galois_elt(1, n, 5)
    gk = c.generate_galois_key(rg, sk, gal)
    g = np.random.default_rng(9)
    x, y = g.uniform(-3, 3, slots), g.uniform(-3, 3, slots)
    scale = 2.0 ** 40
    cx = c.ckks_encrypt(rg, pk, c.ckks_encode(torch.from_numpy(x).cuda(), scale))
    cy = c.ckks_encrypt(rg, pk, c.ckks_encode(torch.from_numpy(y).cuda(), scale))
    dec = c.ckks_decode(c.ckks_decrypt(cx, sk, 0), scale).cpu().numpy()
    assert np.max(np.abs(dec - x)) < 1e-6, "decode(decrypt(encrypt(encode(x)))) = x"
    out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(cx, 2 * Q * n, cy, 2 * Q * n, out, 3 * Q * n, 0, 1)
    c.ckks_relinearize_inplace(out, 3 * Q * n, rk, 0, 1, c.workspace(hg.OP_CKKS_RELIN, 0, 1))
    c.ckks_rescale_inplace(out, 3 * Q * n, 0, 1, c.workspace(hg.OP_CKKS_RESCALE, 0, 1))
    l = Q - 1
    new_scale = scale * scale / primes[Q - 1]
    prod = c.ckks_decode(c.ckks_decrypt(out[:2 * l * n].contiguous(), sk, 1), new_scale, 1).cpu().numpy()
    assert np.max(np.abs(prod - x * y)) < 1e-5, "x .* y after multiply + relinearize + rescale"
    rot = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(cx, 2 * Q * n, rot, 2 * Q * n, gk, gal, 0, 1, c.workspace(hg.OP_CKKS_GALOIS, 0, 1))
    r = c.ckks_decode(c.ckks_decrypt(rot, sk, 0), scale).cpu().numpy()
    assert np.max(np.abs(r - np.roll(x, -1))) < 1e-6, "rotate_rows(1)"


@pytest.mark.parametrize("n,log_q", [(4096, [50, 40, 40]), (32768, [59, 45, 45, 45, 45, 45])])
def test_other_encodings_match_oracle(hg, oracle, torch, n, log_q):
    """complex-slot, coefficient and scalar encodings (ckks/encoder.cu:222-446, :515-690) against the
    oracle: residues identical, decoded doubles identical."""
    c, o, primes = _pair(hg, oracle, n, log_q, [60])
    slots = n // 2
    g = np.random.default_rng(n + 1)
    scale = 2.0 ** 42
    z = g.uniform(-50, 50, slots) + 1j * g.uniform(-50, 50, slots)
    for msg in (z, z[:9]):
        plain = c.ckks_encode_ex(1, torch.from_numpy(np.ascontiguousarray(msg)).cuda(), scale)
        want = o.ckks_encode_ex(1, msg, scale)
        assert np.array_equal(hg.to_host(plain), want), "complex encode"
        for depth in (0, len(log_q) - 1):
            l = len(log_q) - depth
            sub = np.ascontiguousarray(want.reshape(len(log_q), n)[:l].reshape(-1))
            dec = c.ckks_decode_ex(1, hg.to_device(sub), scale, depth).cpu().numpy()
            assert np.array_equal(dec.view(np.float64), o.ckks_decode_ex(1, sub, scale, depth).view(np.float64)), "complex decode"
    m = g.uniform(-1000, 1000, n)
    for msg in (m, m[:5], -np.abs(m)):
        plain = c.ckks_encode_ex(2, torch.from_numpy(np.ascontiguousarray(msg)).cuda(), scale)
        want = o.ckks_encode_ex(2, msg, scale)
        assert np.array_equal(hg.to_host(plain), want), "coefficient encode"
        for depth in (0, 1):
            l = len(log_q) - depth
            sub = np.ascontiguousarray(want.reshape(len(log_q), n)[:l].reshape(-1))
            dec = c.ckks_decode_ex(2, hg.to_device(sub), scale, depth).cpu().numpy()
            assert np.array_equal(dec, o.ckks_decode_ex(2, sub, scale, depth)), "coefficient decode"
    for v in (3.141592653589793, -2.5e6, 0.0, -0.0, 1e-30):
        plain = c.ckks_encode_ex(3, v, scale)
        assert np.array_equal(hg.to_host(plain), o.ckks_encode_ex(3, [v], scale)), "scalar encode"
    with pytest.raises(hg.HEError):
        c.ckks_encode_ex(2, torch.zeros(n + 1, dtype=torch.float64, device="cuda"), scale)


def test_constant_operations_and_mult_i_match_oracle(hg, oracle, torch):
    """addition_constant_plain_ckks_poly / substraction_... / cipher_constant_plain_multiplication_kernel /
    cipher_{mult,div}_by_i_kernel against the oracle, 2- and 3-part ciphertexts, two levels."""
    n = 8192
    c, o, primes = _pair(hg, oracle, n, [59, 45, 45, 45], [59])
    Q = 4
    for parts, l in ((2, Q), (3, Q - 1), (2, 1)):
        ct = np.concatenate([oracle.fill_poly(20 + p, j, n, primes[j]) for p in range(parts) for j in range(l)])
        d = hg.to_device(ct)
        for op in (0, 1, 2):
            for v in (1234567.25 * 2.0 ** 40, -9.75 * 2.0 ** 45, 0.0, -0.4, 2.0 ** 100):
                got = hg.to_host(c.ckks_constant_op(op, d, v, l, parts))
                assert np.array_equal(got, o.ckks_constant_op(op, ct, v, l, parts)), (op, v, parts, l)
        for div in (False, True):
            assert np.array_equal(hg.to_host(c.ckks_mult_i(d, l, parts, div)), o.ckks_mult_i(ct, l, parts, div))
        # in place
        e = d.clone()
        c.ckks_constant_op(2, e, 3.0 * 2.0 ** 30, l, parts, out=e)
        assert np.array_equal(hg.to_host(e), o.ckks_constant_op(2, ct, 3.0 * 2.0 ** 30, l, parts))


@pytest.mark.parametrize("parts", [2, 3])
def test_gaussian_integer_constant_ops(hg, oracle, parts):
    """hegpu_ckks_gaussian_integer_op (add_constant_plain_ckks_v2 / multiply_const_plain_ckks_v2,
    ckks/operator.cu:567-724; kernels multiplication.cu:497-570): a complex constant in every slot, residues of
    the rounded doubles taken exactly (positive, negative, beyond 2^64, zero imaginary part)."""
    import torch
    n = 4096
    c = hg.Context.from_bit_sizes(hg.CKKS, n, [50, 40, 40], [50], sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.CKKS, 12, primes, 3, 1)
    c.upload()
    for depth in (0, 1):
        l = 3 - depth
        ct = synth_ct(primes, range(l), parts, n, 21 + depth)
        d = hg.to_device(ct)
        # (the last two: beyond 2^128 -- a constant times the scale of an un-rescaled ciphertext -- where the residue is
        # taken from mantissa and exponent; the reference's NTL conversion accepts any magnitude)
        for re, im in ((3.0, 0.0), (-7.49, 2.5), (1.5 * 2**40, -3.25 * 2**40), (2.0**70 + 12345.0, -(2.0**66)), (0.0, -1.0),
                       (1.37 * 2.0**130, -(2.0**200 + 2.0**160)), (-1.7976931348623157e308, 3.0 * 2.0**127)):
            for op in (0, 1):
                got = hg.to_host(c.ckks_gaussian_integer_op(op, d, re, im, l, parts))
                torch.cuda.synchronize()
                want = o.ckks_gaussian_integer_op(op, ct, re, im, l, parts)
                assert np.array_equal(got, want), (depth, re, im, op)
    # the oracle's mantissa / exponent path against Python's big integers (limb 0 of a zero ciphertext + constant)
    for v in (1.37 * 2.0**130, 2.0**200 + 2.0**160, 1.7976931348623157e308):
        got = o.ckks_gaussian_integer_op(0, np.zeros(2 * n, dtype=np.uint64), v, 0.0, 1, 2)
        assert int(got[0]) == int(v) % primes[0], v
    with pytest.raises(hg.HEError):
        c.ckks_gaussian_integer_op(0, d, float("inf"), 0.0, l, parts)
