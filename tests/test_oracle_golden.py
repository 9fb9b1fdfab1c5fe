"""CPU: the C oracle and the product's host-side parameter derivation against
the committed golden vectors (tests/golden/*.json, produced by the independent
pure-Python big-int restatement oracle/pyref.py -- the reference's own tests
hold no golden vectors)."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import synth_ct, synth_key

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def summ(arr):
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
    return {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "head": [int(v) for v in a[:4]], "count": int(a.size)}


def check(arr, want, what):
    got = summ(arr)
    assert got["count"] == want["count"], what
    assert got["head"] == want["head"], what
    assert got["sha256"] == want["sha256"], what


@pytest.fixture(scope="module")
def c1(oracle):
    g = load("c1_bfv_4096.json")
    o = oracle.OracleContext(oracle.BFV, 12, g["primes"], 2, 1, 1032193)
    return g, o


def test_reference_hardcoded_constants(oracle):
    """constants that ARE in the reference tree: default chain (defaultmodulus.cpp:18-20),
    psi probe values, TFHE prime/psi (tfhe/context.cu:23-24)."""
    import ctypes
    out = (ctypes.c_uint64 * 8)()
    assert oracle.lib().o_default_modulus_128(4096, out) == 3
    assert [int(out[i]) for i in range(3)] == [0x800004001, 0x800008001, 0x1000002001]
    for q, psi in ((0x800004001, 6071469), (0x800008001, 18291550), (0x1000002001, 28979647)):
        assert oracle.lib().o_min_primitive_root(8192, q) == psi
    # TFHE: psi hard-coded in the reference must be the minimal primitive 2N-th root
    assert oracle.lib().o_min_primitive_root(2048, 1152921504606877697) == 1689264667710614


def test_c1_parameters(c1, hg):
    g, o = c1
    for name in ("psi", "n_inverse", "last_q_modinv", "half", "half_mod", "factor", "base_Bsk"):
        assert [int(v) for v in o.table(name)] == g[name], name
    assert int(o.table("gamma")[0]) == g["gamma"]
    prod = hg.Context.from_default(hg.BFV, 4096, 1, 1032193)
    assert prod.bsk_modulus == g["bsk_modulus"]
    for name, want in g["tables"].items():
        for src, tab in (("oracle", o.table(name)), ("product", prod.table(name))):
            if isinstance(want, dict):
                check(tab, want, f"{src}:{name}")
            else:
                assert [int(v) for v in tab] == want, f"{src}:{name}"


def test_c1_ntt(c1, oracle):
    g, o = c1
    n, q = 4096, g["primes"][0]
    x = oracle.fill_poly(7, 0, n, q)
    check(o.ntt(x.copy(), 1, 1), g["ntt"]["forward"], "forward NTT")
    check(o.ntt(x.copy(), 1, 1, inverse=True), g["ntt"]["inverse_of_input"], "inverse NTT")


def test_c1_ops(c1, oracle):
    """config C1: BFV add + multiply(+relinearize) + rotate on the CPU path."""
    g, o = c1
    n, Q = 4096, 2
    primes = g["primes"]
    ct1 = synth_ct(primes, range(Q), 2, n, 1)
    ct2 = synth_ct(primes, range(Q), 2, n, 2)
    key = synth_key(primes, Q, 3, n, 3)
    out = np.zeros_like(ct1)
    o.L.o_addition(ct1.ctypes.data, ct2.ctypes.data, out.ctypes.data, o.qp_mods, 12, Q, 2)
    check(out, g["ops"]["add"], "add")
    o.L.o_substraction(ct1.ctypes.data, ct2.ctypes.data, out.ctypes.data, o.qp_mods, 12, Q, 2)
    check(out, g["ops"]["sub"], "sub")
    mul = o.bfv_multiply(ct1, ct2)
    check(mul, g["ops"]["multiply"], "multiply")
    rel = o.bfv_relinearize(mul.copy(), key)
    check(rel[:2 * Q * n], g["ops"]["multiply_relinearize"], "multiply+relinearize")
    assert oracle.lib().o_steps_to_galois_elt(1, n, 3) == g["ops"]["galois_elt"]
    check(o.bfv_apply_galois(ct1, key, g["ops"]["galois_elt"]), g["ops"]["rotate_rows_1"], "rotate")


def test_ckks_small(oracle, hg):
    g = load("ckks_4096.json")
    n, Q, Qp = 4096, 3, 4
    prod = hg.Context.from_bit_sizes(hg.CKKS, n, [40, 30, 30], [40], sec=hg.SEC_NONE)
    assert [int(v) for v in prod.table("modulus")] == g["primes"]
    o = oracle.OracleContext(oracle.CKKS, 12, g["primes"], Q, 1)
    for name in ("psi", "n_inverse", "last_q_modinv", "half", "half_mod", "factor", "rescaled_half",
                 "rescaled_half_mod", "rescaled_last_q_modinv", "new_prime_locations", "new_input_locations"):
        assert [int(v) for v in o.table(name)] == g[name], "oracle:" + name
        assert [int(v) for v in prod.table(name)] == g[name], "product:" + name
    for name in ("ntt_table", "intt_table"):
        check(o.table(name), g["tables"][name], "oracle:" + name)
        check(prod.table(name), g["tables"][name], "product:" + name)
    key = synth_key(g["primes"], Q, Qp, n, 3)
    for depth in (0, 1):
        want = g["ops"]["depth%d" % depth]
        l = Q - depth
        ct1 = synth_ct(g["primes"], range(l), 2, n, 1)
        ct2 = synth_ct(g["primes"], range(l), 2, n, 2)
        mul = o.ckks_multiply(ct1, ct2, depth)
        check(mul, want["multiply"], "multiply")
        rel = o.ckks_relinearize(mul.copy(), key, depth)[:2 * l * n].copy()
        check(rel, want["relinearize"], "relinearize")
        res = o.ckks_rescale(rel.copy(), depth)[:2 * (l - 1) * n]
        check(res, want["rescale"], "rescale")
        assert oracle.lib().o_steps_to_galois_elt(1, n, 5) == want["galois_elt"]
        check(o.ckks_apply_galois(ct1, key, want["galois_elt"], depth), want["rotate_1"], "rotate")


def test_benchmark_chains(oracle, hg):
    """prime chains of the C2 / C4 benchmark configs (deterministic SEAL-style search)."""
    g = load("params.json")
    for name, item in g.items():
        nq = len(item["bits"]) - 1
        prod = hg.Context.from_bit_sizes(hg.CKKS, item["n"], item["bits"][:nq], item["bits"][nq:])
        assert [int(v) for v in prod.table("modulus")] == item["primes"], name
        import ctypes
        bits = (ctypes.c_int * len(item["bits"]))(*item["bits"])
        out = (ctypes.c_uint64 * len(item["bits"]))()
        assert oracle.lib().o_generate_primes(item["n"], bits, len(item["bits"]), out) == 0
        assert [int(v) for v in out] == item["primes"], name
        psi = [int(v) for v in prod.table("psi")][:2]
        assert psi == item["psi"][:2]


def test_no_barrett_domain_violation(oracle):
    """every o_mult call of the suites above stayed inside Barrett's exact domain
    (a*b < 2^(2*bit)): outside it the reference's behaviour would be unpinned."""
    import ctypes
    v = ctypes.c_uint64.in_dll(oracle.lib(), "o_barrett_domain_violations").value
    assert v == 0


def test_method_II_tables_product_vs_oracle(oracle, hg):
    """KeySwitchParameterGenerator output (contextpool.cpp): product host code vs oracle."""
    for scheme, oscheme, bits_q, bits_p, t in ((hg.CKKS, oracle.CKKS, [40, 30, 30, 30, 30], [40, 40], 0),
                                               (hg.CKKS, oracle.CKKS, [50, 40, 40, 40, 40, 40, 40], [50, 50, 50], 0),
                                               (hg.BFV, oracle.BFV, [36, 36, 36], [37, 37], 1032193)):
        n = 4096
        prod = hg.Context.from_bit_sizes(scheme, n, bits_q, bits_p, plain_modulus=t, sec=hg.SEC_NONE)
        primes = [int(v) for v in prod.table("modulus")]
        o = oracle.OracleContext(oscheme, prod.n_power, primes, len(bits_q), len(bits_p), t)
        for name in ("m2_I_j", "m2_I_location", "m2_Mi_inv", "m2_matrix", "m2_prod"):
            assert np.array_equal(prod.table(name), o.table(name)), name


# ------------------------------------------------------------------ every constant the reference tree holds
# tests/golden/reference_constants.json is extracted from the reference by tools/extract_reference_constants.py
# (defaultmodulus.cpp:12-175, secstdparams.h:22-79, tfhe/context.cu:23-57, benchmark/*.cpp).  With no limb-level
# vectors in the reference and no way to build it here, this is the reference-held data there is: the oracle
# AND the product's own host-side parameter code (csrc/host_params.cpp, a separate implementation) are held to it.
REF = load("reference_constants.json")
DEGREES = (4096, 8192, 16384, 32768, 65536)


@pytest.mark.parametrize("level", [128, 192, 256])
def test_default_chains_match_the_reference(oracle, hg, level):
    import ctypes
    sec = {128: hg.SEC_128, 192: hg.SEC_192, 256: hg.SEC_256}[level]
    for n in DEGREES:
        want = REF["default_modulus"][str(level)][str(n)]
        out = (ctypes.c_uint64 * 64)()
        cnt = oracle.lib().o_default_modulus(n, level, out)
        assert [int(out[i]) for i in range(cnt)] == want, ("oracle", level, n)
        c = hg.Context.from_default(hg.CKKS, n, 1, sec=sec)
        assert [int(v) for v in c.table("modulus")] == want, ("product", level, n)
        assert (c.Q_size, c.P_size) == (len(want) - 1, 1)
        c.close()
        # what the chains must satisfy for the transform: NTT-friendly primes within the security budget
        bits = 0
        for q in want:
            assert oracle.lib().o_is_prime(q) and q % (2 * n) == 1, (level, n, hex(q))
            bits += q.bit_length()
        assert bits <= REF["max_logq"][str(level)][str(n)], (level, n, bits)
        assert oracle.lib().o_max_logq(n, level) == REF["max_logq"][str(level)][str(n)]
        assert c.__class__.from_default(hg.CKKS, n, 1, sec=sec).int("max_logq_%d" % level) == REF["max_logq"][str(level)][str(n)]


@pytest.mark.parametrize("level", [128, 192, 256])
def test_security_budget_is_enforced_like_the_reference(hg, level):
    """ckks/context.cu:94-119: a chain whose total bit count exceeds the level's table entry is refused"""
    sec = {128: hg.SEC_128, 192: hg.SEC_192, 256: hg.SEC_256}[level]
    for n in (8192, 16384):
        budget = REF["max_logq"][str(level)][str(n)]
        b = 30
        k = budget // b - 1
        assert k >= 1
        hg.Context.from_bit_sizes(hg.CKKS, n, [b] * k, [b], sec=sec).close()       # b (k + 1) <= budget
        with pytest.raises(hg.HEError) as e:
            hg.Context.from_bit_sizes(hg.CKKS, n, [b] * (k + 1), [b], sec=sec)     # one prime more
        assert e.value.code == hg.E_RUNTIME and "security recommendations" in str(e.value)


def test_minimal_roots_of_every_default_prime(oracle, hg):
    """psi of every prime of every default chain: the oracle's and the product's root selection agree, the root
    has order exactly 2N (psi^N = -1) -- what the reference's tables are built from (util.cu:312-380)."""
    for level in ("128", "192", "256"):
        for n in DEGREES:
            chain = REF["default_modulus"][level][str(n)]
            sec = {"128": hg.SEC_128, "192": hg.SEC_192, "256": hg.SEC_256}[level]
            c = hg.Context.from_default(hg.CKKS, n, 1, sec=sec)
            psi = [int(v) for v in c.table("psi")]
            c.close()
            for q, r in zip(chain, psi):
                assert oracle.lib().o_min_primitive_root(2 * n, q) == r, (level, n, hex(q))
                assert pow(r, n, q) == q - 1


def test_tfhe_parameter_set_matches_the_reference(oracle, hg):
    t = REF["tfhe"]
    o = oracle.OracleTfhe()
    assert o.prime == t["prime"] and oracle.lib().o_min_primitive_root(2 << t["ntt_log_size"], t["prime"]) == t["psi"]
    assert (o.n, o.N, o.k, o.l, o.ks_length, o.ks_base) == (t["n"], t["N"], t["k"], t["bk_l"], t["ks_length"], 1 << t["ks_base_bit"])
    p = hg.TfheContext()
    assert p.prime == t["prime"]
    for name in ("n", "N", "k", "bk_l", "bk_bg_bit", "ks_base_bit", "ks_length"):
        assert p.int(name) == t[name], name


def test_error_distribution_parameter():
    """secstdparams.h:22 error_std_dev = 3.2 -- the value both Gaussian CDTs are built for (csrc/context.cpp,
    oracle/o_keygen.c); the empirical sigma of generated keys is checked in tests/test_oracle_keygen.py"""
    assert REF["error_std_dev"] == 3.2


def test_galois_automorphism_is_a_slot_gather_in_the_ntt_domain(oracle, hg):
    """The index map behind the NTT-domain rotations (csrc/rns.hip k_permute_ntt, NttEpilogue::galois_inv), pinned
    on the CPU with the oracle's transform: for b(X) = a(X^g) -- the reference's coefficient permutation
    out[(i g) mod N] = +-a[i] (switchkey.cu:1689-1711) -- NTT(b)[j] = NTT(a)[j'] with
    2 br(j') + 1 = (2 br(j) + 1) g mod 2N; and the scatter form with g^-1 mod 2N is the same permutation."""
    n, np_ = 4096, 12
    prod = hg.Context.from_bit_sizes(hg.CKKS, n, [40, 30, 30], [40], sec=hg.SEC_NONE)
    primes = [int(v) for v in prod.table("modulus")]
    o = oracle.OracleContext(oracle.CKKS, np_, primes, 3, 1)
    rng = np.random.default_rng(5)
    br = np.array([int(format(j, "012b")[::-1], 2) for j in range(n)], dtype=np.int64)
    j = np.arange(n, dtype=np.int64)
    for g in (5, 25, 3, 2 * n - 1, oracle.lib().o_steps_to_galois_elt(-7, n, 5)):
        src = br[(((2 * br[j] + 1) * g) % (2 * n) - 1) // 2]                   # gather: out[j] = in[src[j]]
        ginv = pow(int(g), -1, 2 * n)
        dst = br[(((2 * br[j] + 1) * ginv) % (2 * n) - 1) // 2]                # scatter: out[dst[j]] = in[j]
        assert np.array_equal(dst[src], j) and np.array_equal(src[dst], j)
        for m in range(3):
            q = primes[m]
            a = rng.integers(0, q, n, dtype=np.uint64)
            b = np.zeros(n, dtype=np.uint64)
            e = (j * g) % (2 * n)
            b[e % n] = np.where(e >= n, (q - a) % q, a)
            A = o.ntt(a.copy(), 1, 1, mod_offset=m)
            B = o.ntt(b.copy(), 1, 1, mod_offset=m)
            assert np.array_equal(B, A[src]), (g, m)
            scattered = np.empty_like(A)
            scattered[dst] = A
            assert np.array_equal(scattered, B), (g, m)
