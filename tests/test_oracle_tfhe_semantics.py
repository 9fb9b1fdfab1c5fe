"""Semantic pin of the TFHE oracle (CPU): with REAL keys built in the
reference's layouts (boot key [n][k+1][l][k+1][N] NTT domain,
bootstrapping.cu:1037-1041; key-switch key [N][ks_length][base-1][n],
bootstrapping.cu:1385-1412) the restated gate path must reproduce the truth
tables -- the shape of reference test/test_tfhe_gate_boot.cpp:64-86."""
import numpy as np
import pytest


def wrap32(x):
    return ((np.asarray(x, dtype=np.int64) + 2**31) % 2**32 - 2**31).astype(np.int32)


@pytest.fixture(scope="module")
def keys(oracle):
    o = oracle.OracleTfhe()
    rng = np.random.default_rng(42)
    n, N, l, bg_bit, ks_len, ks_bit = 512, 1024, 2, 10, 8, 2
    s = rng.integers(0, 2, n).astype(np.int64)           # LWE key
    S = rng.integers(0, 2, N).astype(np.int32)           # TRLWE key (k = 1)

    def noise(size, bits):
        return rng.integers(-(1 << bits), (1 << bits) + 1, size)

    # boot key: bk[i][y][z] = TRLWE_S(0) + s_i * h_z on component y
    bk = np.zeros((n, 2, l, 2, N), dtype=np.uint64)
    for i in range(n):
        for y in range(2):
            for z in range(l):
                a = wrap32(rng.integers(-2**31, 2**31, N))
                b = wrap32(o.polymul(a, S).astype(np.int64) + noise(N, 2))
                comp = [a, b]
                h = 1 << (32 - bg_bit * (z + 1))
                comp[y] = comp[y].copy()
                comp[y][0] = wrap32(int(comp[y][0]) + int(s[i]) * h)
                bk[i, y, z, 0] = o.to_ntt(comp[0])
                bk[i, y, z, 1] = o.to_ntt(comp[1])
    # key-switch key: ks[i][j][v-1] = LWE_s(v * S_i / base^(j+1))
    base = 1 << ks_bit
    A = rng.integers(-2**31, 2**31, (N, ks_len, base - 1, n))
    msg = np.zeros((N, ks_len, base - 1), dtype=np.int64)
    for j in range(ks_len):
        for v in range(1, base):
            msg[:, j, v - 1] = S.astype(np.int64) * v * (1 << (32 - ks_bit * (j + 1)))
    B = (A * s).sum(axis=3) + msg + noise((N, ks_len, base - 1), 2)
    ks_a = wrap32(A).reshape(-1)
    ks_b = wrap32(B).reshape(-1)
    return o, rng, s, bk.reshape(-1), ks_a, ks_b


def encrypt_bits(rng, s, bits):
    n = s.shape[0]
    mu = 1 << 29  # 1/8 on the 32-bit torus
    a = rng.integers(-2**31, 2**31, (len(bits), n))
    b = (a * s).sum(axis=1) + np.array([mu if x else -mu for x in bits]) + rng.integers(-2**10, 2**10, len(bits))
    return wrap32(a).reshape(-1), wrap32(b)


def decrypt_bits(s, a, b):
    n = s.shape[0]
    a = a.reshape(-1, n).astype(np.int64)
    phase = wrap32(b.astype(np.int64) - (a * s).sum(axis=1))
    return [int(p > 0) for p in phase]


TRUTH = {0: lambda x, y: 1 - (x & y), 1: lambda x, y: x & y, 2: lambda x, y: (1 - x) & y,
         3: lambda x, y: 1 - (x | y), 4: lambda x, y: x | y, 5: lambda x, y: 1 - (x ^ y), 6: lambda x, y: x ^ y}


@pytest.mark.parametrize("gate", [0, 1, 2, 3, 4, 5, 6])
def test_gate_truth_table(keys, gate):
    o, rng, s, bk, ks_a, ks_b = keys
    xs, ys = [0, 0, 1, 1], [0, 1, 0, 1]
    a1, b1 = encrypt_bits(rng, s, xs)
    a2, b2 = encrypt_bits(rng, s, ys)
    assert decrypt_bits(s, a1, b1) == xs
    oa, ob_ = o.gate(gate, a1, b1, a2, b2, bk, ks_a, ks_b)
    got = decrypt_bits(s, oa, ob_)
    assert got == [TRUTH[gate](x, y) for x, y in zip(xs, ys)]


def test_generated_keys_reproduce_truth_tables(oracle):
    """keys from the oracle's own generator (oracle/o_keygen.c: binary keys, TGSW boot key,
    key-switch key, Irwin-Hall torus noise) drive the gate path to the truth tables"""
    o = oracle.OracleTfhe()
    rng = oracle.ORng(2026)
    lwe, tlwe = o.gen_secret(rng)
    assert set(np.unique(lwe)) <= {0, 1} and 150 < lwe.sum() < 362
    bk, ks_a, ks_b = o.gen_bootkey(rng, lwe, tlwe)
    xs, ys = [0, 0, 1, 1], [0, 1, 0, 1]
    mu = 1 << 29
    a1, b1 = o.encrypt(rng, lwe, [mu if x else -mu for x in xs])
    a2, b2 = o.encrypt(rng, lwe, [mu if y else -mu for y in ys])
    ph = o.phase(lwe, a1, b1)
    assert [int(p > 0) for p in ph] == xs
    assert np.max(np.abs(ph.astype(np.int64) - np.array([mu if x else -mu for x in xs]))) < 1 << 22
    for gate in (0, 4, 6):  # NAND, OR, XOR
        oa, ob_ = o.gate(gate, a1, b1, a2, b2, bk, ks_a, ks_b)
        got = [int(p > 0) for p in o.phase(lwe, oa, ob_)]
        assert got == [TRUTH[gate](x, y) for x, y in zip(xs, ys)], gate
