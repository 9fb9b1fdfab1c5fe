#!/usr/bin/env python3
"""pyref.py -- pure-Python big-integer restatement of the hot path, used to
(a) cross-check the C oracle and (b) generate the committed golden vectors in
tests/golden/ (the reference's own tests hold none -- SURVEY.md 8c).

TEST INFRASTRUCTURE ONLY.  Written from the mathematics + the reference's
call sites, independently of oracle/*.c: exact modular arithmetic with Python
ints, NTT by definition-checked iterative butterflies, key switching / mod-down /
rescale / BEHZ following the reference kernels step by step
(src/lib/kernel/{multiplication,switchkey}.cu, src/lib/host/*/operator.cu).

    python oracle/pyref.py            # regenerate tests/golden/*.json
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
M64 = (1 << 64) - 1


# ----------------------------------------------------------------- number theory
def is_prime(v):
    if v < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if v == p:
            return True
        if v % p == 0:
            return False
    d, r = v - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, v)
        if x in (1, v - 1):
            continue
        for _ in range(r - 1):
            x = x * x % v
            if x == v - 1:
                break
        else:
            return False
    return True


def generate_primes(n, bit_sizes):
    """util.cu:219-276: per bit size scan down from floor((2^b-1)/2N)*2N+1."""
    need = {}
    for b in bit_sizes:
        need[b] = need.get(b, 0) + 1
    pool = {}
    for b, cnt in need.items():
        v = ((1 << b) - 1) // (2 * n) * (2 * n) + 1
        got = []
        while len(got) < cnt and v > (1 << (b - 1)):
            if is_prime(v):
                got.append(v)
            v -= 2 * n
        assert len(got) == cnt
        pool[b] = got
    return [pool[b].pop() for b in bit_sizes]


def min_primitive_root(degree, q):
    """util.cu:312-380: minimum over all primitive degree-th roots."""
    assert (q - 1) % degree == 0
    g = 2
    while True:
        c = pow(g, (q - 1) // degree, q)
        if pow(c, degree // 2, q) == q - 1:
            break
        g += 1
    best, cur, sq = c, c, c * c % q
    for _ in range(degree // 2 - 1):
        cur = cur * sq % q
        best = min(best, cur)
    return best


def bitrev(x, bits):
    return int(bin(x)[2:].zfill(bits)[::-1], 2) if bits else 0


def power_table(base, q, n_power):
    n = 1 << n_power
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * base % q
    return [pw[bitrev(j, n_power)] for j in range(n)]


# ----------------------------------------------------------------- NTT
def ntt(a, table, q):
    a = list(a)
    n = len(a)
    t, m = n, 1
    while m < n:
        t >>= 1
        for i in range(m):
            w = table[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], a[j + t] * w % q
                a[j], a[j + t] = (u + v) % q, (u - v) % q
        m <<= 1
    return a


def intt(a, itable, q, n_inv):
    a = list(a)
    n = len(a)
    t, m = 1, n >> 1
    while m >= 1:
        for i in range(m):
            w = itable[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], a[j + t]
                a[j], a[j + t] = (u + v) % q, (u - v) * w % q
        t <<= 1
        m >>= 1
    return [x * n_inv % q for x in a]


def ntt_by_definition(a, psi, q, n_power, slots):
    """slot j = sum_i a[i] * psi^((2*bitrev(j)+1)*i)  (switchkey.cu:1461-1476)."""
    out = {}
    for j in slots:
        w = pow(psi, 2 * bitrev(j, n_power) + 1, q)
        acc, p = 0, 1
        for x in a:
            acc = (acc + x * p) % q
            p = p * w % q
        out[j] = acc
    return out


# ----------------------------------------------------------------- synthetic data
def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def fill_poly(seed, limb, n, q):
    return [splitmix64((seed + (limb << 32) + i) & M64) % q for i in range(n)]


def synth_ct(primes, limb_ids, parts, n, seed):
    return [[fill_poly(seed * 1000 + p, lid, n, primes[lid]) for lid in limb_ids] for p in range(parts)]


def synth_key(primes, Q, Qp, n, seed):
    return [[[fill_poly(seed * 100000 + i * 2 + c, j, n, primes[j]) for j in range(Qp)] for c in range(2)]
            for i in range(Q)]


def flat(x):
    if isinstance(x, int):
        return [x]
    out = []
    for y in x:
        out.extend(flat(y))
    return out


def digest(x):
    import array
    return hashlib.sha256(array.array("Q", flat(x)).tobytes()).hexdigest()


# ----------------------------------------------------------------- context
class Ctx:
    def __init__(self, scheme, n_power, primes, Q, P, t=0):
        self.scheme, self.np, self.n = scheme, n_power, 1 << n_power
        self.primes, self.Q, self.P, self.Qp, self.t = primes, Q, P, Q + P, t
        n = self.n
        self.psi = [min_primitive_root(2 * n, q) for q in primes]
        self.tab = [power_table(r, q, n_power) for r, q in zip(self.psi, primes)]
        self.itab = [power_table(pow(r, -1, q), q, n_power) for r, q in zip(self.psi, primes)]
        self.ninv = [pow(n, -1, q) for q in primes]
        Qp = self.Qp
        self.half, self.half_mod, self.lqm, self.factor = [], [], [], []
        for i in range(P):
            p = primes[Qp - 1 - i]
            self.half.append(p >> 1)
            for j in range(Qp - 1 - i):
                self.lqm.append(pow(p % primes[j], -1, primes[j]))
                self.half_mod.append((p >> 1) % primes[j])
            self.factor += [p % primes[j] for j in range(Q)]
        if scheme == "ckks":
            self.r_half, self.r_half_mod, self.r_lqm = [], [], []
            for d in range(Q - 1):
                last = Q - 1 - d
                ql = primes[last]
                self.r_half.append(ql >> 1)
                for i in range(last):
                    self.r_lqm.append(pow(ql % primes[i], -1, primes[i]))
                    self.r_half_mod.append((ql >> 1) % primes[i])
            self.prime_loc, self.input_loc = [], []
            for d in range(Q):
                self.prime_loc += list(range(Q - d)) + [Q + j for j in range(P)]
            for i in range(Qp - 1):
                c = Qp - i
                self.input_loc += [c - 1, 2 * c - 1]
        if scheme == "bfv":
            self.behz()

    def behz(self):
        """bfv/context.cu:510-671, 939-1347."""
        n, Q, Qp, q, t = self.n, self.Q, self.Qp, self.primes, self.t
        mt = 1 << 32
        total_bits = sum(x.bit_length() for x in q)
        bsk = Qp + (1 if t.bit_length() + total_bits + 32 >= 61 * Q + 61 else 0)
        ip = generate_primes(n, [61] * (bsk + 1))
        B, self.gamma = ip[:bsk], ip[bsk]
        msk = B[-1]
        self.bsk, self.B, self.mt = bsk, B, mt
        prod = lambda xs, m: __import__("functools").reduce(lambda a, b: a * b % m, xs, 1)
        self.m_q_Bsk = [prod([q[j] for j in range(Q) if j != i], B[k]) for k in range(bsk) for i in range(Q)]
        self.inv_punct = [pow(prod([q[j] for j in range(Q) if j != i], q[i]), -1, q[i]) for i in range(Q)]
        self.m_mt = [prod([q[j] for j in range(Q) if j != i], mt) for i in range(Q)]
        self.inv_prod_q_mt = pow(prod(q[:Q], mt), -1, mt)
        self.inv_mt_B = [pow(mt, -1, b) for b in B]
        self.prod_q_B = [prod(q[:Q], b) for b in B]
        self.inv_prod_q_B = [pow(x, -1, b) for x, b in zip(self.prod_q_B, B)]
        self.m_B_q = [prod([B[j] for j in range(bsk - 1) if j != i], q[k]) for k in range(Q) for i in range(bsk - 1)]
        self.m_msk = [prod([B[j] for j in range(bsk - 1) if j != i], msk) for i in range(bsk - 1)]
        self.inv_punct_B = [pow(prod([B[j] for j in range(bsk - 1) if j != i], B[i]), -1, B[i])
                            for i in range(bsk - 1)]
        self.inv_prod_B_msk = pow(prod(B[:-1], msk), -1, msk)
        self.prod_B_q = [prod(B[:-1], q[i]) for i in range(Q)]
        self.mm = q[:Q] + B
        self.mpsi = self.psi[:Q] + [min_primitive_root(2 * n, b) for b in B]
        self.mtab = [power_table(r, m, self.np) for r, m in zip(self.mpsi, self.mm)]
        self.mitab = [power_table(pow(r, -1, m), m, self.np) for r, m in zip(self.mpsi, self.mm)]
        self.mninv = [pow(n, -1, m) for m in self.mm]


# ----------------------------------------------------------------- key switch (method I)
def keyswitch(c, poly_coeff, key, depth):
    """decompose -> NTT -> inner product with key  (switchkey.cu:11-285).
    poly_coeff: l limbs in COEFFICIENT domain.  Returns [2][rc][N] NTT domain
    (rows ordered q_0..q_{l-1}, P)."""
    Q, Qp, n, q = c.Q, c.Qp, c.n, c.primes
    l, rc = Q - depth, Qp - depth
    mods = list(range(l)) + [Qp - 1]
    out = [[[0] * n for _ in range(rc)] for _ in range(2)]
    for i in range(l):
        for r, j in enumerate(mods):
            d = ntt([x % q[j] for x in poly_coeff[i]], c.tab[j], q[j])
            for part in range(2):
                k = key[i][part][j]
                o = out[part][r]
                for x in range(n):
                    o[x] = (o[x] + d[x] * k[x]) % q[j]
    return out, mods


def moddown_value(x, last_plus_half, half_mod, inv, q):
    """(x - ((x_P + half) mod q - half mod q)) * P^-1 mod q  (switchkey.cu:411-428)."""
    r = (last_plus_half % q - half_mod) % q
    return (x - r) * inv % q


def ckks_relinearize(c, ct3, key, depth):
    """ckks/operator.cu:899-1023; ct3 = [3][l][N] NTT domain -> [2][l][N]."""
    Q, n, q = c.Q, c.n, c.primes
    l = Q - depth
    c2 = [intt(ct3[2][j], c.itab[j], q[j], c.ninv[j]) for j in range(l)]
    ks, mods = keyswitch(c, c2, key, depth)
    P = q[c.Qp - 1]
    out = []
    for part in range(2):
        lastc = intt(ks[part][l], c.itab[c.Qp - 1], P, c.ninv[c.Qp - 1])
        lastc = [(x + c.half[0]) % P for x in lastc]
        rows = []
        for j in range(l):
            corr = ntt([(x % q[j] - c.half_mod[j]) % q[j] for x in lastc], c.tab[j], q[j])
            rows.append([((ks[part][j][x] - corr[x]) * c.lqm[j] + ct3[part][j][x]) % q[j] for x in range(n)])
        out.append(rows)
    return out


def ckks_rescale(c, ct, depth):
    """ckks/operator.cu:1156-1244; [2][l][N] -> [2][l-1][N]."""
    Q, n, q = c.Q, c.n, c.primes
    l = Q - depth
    loc = sum(Q - 1 - i for i in range(depth))
    ql = q[l - 1]
    out = []
    for part in range(2):
        lastc = intt(ct[part][l - 1], c.itab[l - 1], ql, c.ninv[l - 1])
        lastc = [(x + c.r_half[depth]) % ql for x in lastc]
        rows = []
        for j in range(l - 1):
            corr = ntt([(x % q[j] - c.r_half_mod[loc + j]) % q[j] for x in lastc], c.tab[j], q[j])
            rows.append([(ct[part][j][x] - corr[x]) * c.r_lqm[loc + j] % q[j] for x in range(n)])
        out.append(rows)
    return out


def permute(poly, g, q):
    """coefficient-domain automorphism (switchkey.cu:1687-1699); q - x without zero test."""
    n = len(poly)
    out = [0] * n
    for i, x in enumerate(poly):
        r = i * g
        out[r % n] = (q - x) if (r // n) & 1 else x
    return out


def ckks_apply_galois(c, ct, key, g, depth):
    """ckks/operator.cu:1422-1559."""
    Q, n, q = c.Q, c.n, c.primes
    l = Q - depth
    coeff = [[intt(ct[p][j], c.itab[j], q[j], c.ninv[j]) for j in range(l)] for p in range(2)]
    ks, mods = keyswitch(c, coeff[1], key, depth)
    P = q[c.Qp - 1]
    out = []
    for part in range(2):
        ksc = [intt(ks[part][r], c.itab[j], q[j], c.ninv[j]) for r, j in enumerate(mods)]
        lastc = [(x + c.half[0]) % P for x in ksc[l]]
        rows = []
        for j in range(l):
            vals = [moddown_value(ksc[j][x], lastc[x], c.half_mod[j], c.lqm[j], q[j]) for x in range(n)]
            if part == 0:
                vals = [(coeff[0][j][x] + vals[x]) % q[j] for x in range(n)]
            rows.append(ntt([v % q[j] for v in permute(vals, g, q[j])], c.tab[j], q[j]))
        out.append(rows)
    return out


def bfv_relinearize(c, ct3, key):
    """bfv/operator.cu:505-583; coefficient domain."""
    Q, n, q = c.Q, c.n, c.primes
    ks, mods = keyswitch(c, ct3[2], key, 0)
    P = q[c.Qp - 1]
    out = []
    for part in range(2):
        ksc = [intt(ks[part][r], c.itab[j], q[j], c.ninv[j]) for r, j in enumerate(mods)]
        lastc = [(x + c.half[0]) % P for x in ksc[Q]]
        out.append([[(ct3[part][j][x] + moddown_value(ksc[j][x], lastc[x], c.half_mod[j], c.lqm[j], q[j])) % q[j]
                     for x in range(n)] for j in range(Q)])
    return out


def bfv_apply_galois(c, ct, key, g):
    """bfv/operator.cu:771-864."""
    Q, n, q = c.Q, c.n, c.primes
    ks, mods = keyswitch(c, ct[1], key, 0)
    P = q[c.Qp - 1]
    out = []
    for part in range(2):
        ksc = [intt(ks[part][r], c.itab[j], q[j], c.ninv[j]) for r, j in enumerate(mods)]
        lastc = [(x + c.half[0]) % P for x in ksc[Q]]
        rows = []
        for j in range(Q):
            vals = [moddown_value(ksc[j][x], lastc[x], c.half_mod[j], c.lqm[j], q[j]) for x in range(n)]
            if part == 0:
                vals = [(ct[0][j][x] + vals[x]) % q[j] for x in range(n)]
            rows.append(permute(vals, g, q[j]))
        out.append(rows)
    return out


def cross_multiplication(a, b, mods):
    """multiplication.cu:102-126."""
    L, n = len(mods), len(a[0][0])
    o0 = [[a[0][j][x] * b[0][j][x] % mods[j] for x in range(n)] for j in range(L)]
    o1 = [[(a[0][j][x] * b[1][j][x] + a[1][j][x] * b[0][j][x]) % mods[j] for x in range(n)] for j in range(L)]
    o2 = [[a[1][j][x] * b[1][j][x] % mods[j] for x in range(n)] for j in range(L)]
    return [o0, o1, o2]


def bfv_multiply(c, ct1, ct2):
    """bfv/operator.cu:336-430 (BEHZ): fast_convertion -> NTT -> tensor -> INTT -> fast_floor."""
    Q, n, q, B, bsk, mt, t = c.Q, c.n, c.primes, c.B, c.bsk, c.mt, c.t
    L = Q + bsk
    msk = B[-1]

    def fast_convertion(poly):  # [Q][N] -> [L][N]   (multiplication.cu:10-100)
        out = [list(poly[i]) for i in range(Q)] + [[0] * n for _ in range(bsk)]
        for x in range(n):
            temp = [poly[i][x] * mt % q[i] * c.inv_punct[i] % q[i] for i in range(Q)]
            t2 = [sum(temp[j] * c.m_q_Bsk[j + i * Q] for j in range(Q)) % B[i] for i in range(bsk)]
            tmt = sum((temp[j] % mt) * c.m_mt[j] for j in range(Q)) % mt
            r = (mt - tmt * c.inv_prod_q_mt % mt)  # may equal mt (kept as in the kernel)
            for i in range(bsk):
                t3 = r
                if t3 >= (mt >> 1):
                    t3 = (B[i] - mt + r) % B[i]
                t3 = t3 * c.prod_q_B[i] % B[i]
                out[Q + i][x] = (t2[i] + t3) * c.inv_mt_B[i] % B[i]
        return out

    ext = [[ntt(row, c.mtab[j], c.mm[j]) for j, row in enumerate(fast_convertion(ct[p]))]
           for ct in (ct1, ct2) for p in range(2)]
    prod3 = cross_multiplication(ext[0:2], ext[2:4], c.mm)
    prod3 = [[intt(prod3[p][j], c.mitab[j], c.mm[j], c.mninv[j]) for j in range(L)] for p in range(3)]

    def fast_floor(poly):  # [L][N] -> [Q][N]   (multiplication.cu:128-272)
        out = [[0] * n for _ in range(Q)]
        for x in range(n):
            reg_q = [poly[i][x] * t % q[i] * c.inv_punct[i] % q[i] for i in range(Q)]
            reg_B = [poly[Q + i][x] * t % B[i] for i in range(bsk)]
            tmp = [sum(reg_q[j] * c.m_q_Bsk[j + i * Q] for j in range(Q)) % B[i] for i in range(bsk)]
            reg_B = [(reg_B[i] - tmp[i]) * c.inv_prod_q_B[i] % B[i] for i in range(bsk)]
            temp3 = [reg_B[i] * c.inv_punct_B[i] % B[i] for i in range(bsk - 1)]
            temp4 = [sum((temp3[j] % q[i]) * c.m_B_q[j + i * (bsk - 1)] for j in range(bsk - 1)) % q[i]
                     for i in range(Q)]
            t4sk = sum(temp3[j] * c.m_msk[j] for j in range(bsk - 1)) % msk
            alpha = (t4sk - reg_B[bsk - 1]) * c.inv_prod_B_msk % msk
            for i in range(Q):
                if alpha > (msk >> 1):
                    inner = (msk % q[i] - alpha % q[i]) * c.prod_B_q[i] % q[i]
                else:
                    inner = (q[i] - c.prod_B_q[i]) * (alpha % q[i]) % q[i]
                out[i][x] = (temp4[i] + inner) % q[i]
        return out

    return [fast_floor(prod3[p]) for p in range(3)]


# ----------------------------------------------------------------- golden generation
def head(x, k=4):
    return [int(v) for v in flat(x)[:k]]


def summarize(x):
    return {"sha256": digest(x), "head": head(x), "count": len(flat(x))}


DEFAULT_4096 = [0x800004001, 0x800008001, 0x1000002001]  # defaultmodulus.cpp:18-20


def golden_c1():
    """BASELINE config C1: BFV N=2^12 default chain, t=1032193."""
    n_power, n, t = 12, 4096, 1032193
    c = Ctx("bfv", n_power, DEFAULT_4096, 2, 1, t)
    assert c.psi == [6071469, 18291550, 28979647]  # SURVEY.md 8a-a6 probe of the reference generator
    g = {"config": "C1: BFV N=4096, default 128-bit chain (Q=2,P=1), t=1032193", "primes": c.primes,
         "psi": c.psi, "n_inverse": c.ninv, "last_q_modinv": c.lqm, "half": c.half, "half_mod": c.half_mod,
         "factor": c.factor, "base_Bsk": c.B, "gamma": c.gamma, "bsk_modulus": c.bsk,
         "tables": {
             "ntt_table": summarize(c.tab), "intt_table": summarize(c.itab),
             "base_change_matrix_Bsk": c.m_q_Bsk, "inv_punctured_prod_mod_base_array": c.inv_punct,
             "base_change_matrix_m_tilde": c.m_mt, "inv_prod_q_mod_m_tilde": [c.inv_prod_q_mt],
             "inv_m_tilde_mod_Bsk": c.inv_mt_B, "prod_q_mod_Bsk": c.prod_q_B, "inv_prod_q_mod_Bsk": c.inv_prod_q_B,
             "base_change_matrix_q": c.m_B_q, "base_change_matrix_msk": c.m_msk,
             "inv_punctured_prod_mod_B_array": c.inv_punct_B, "inv_prod_B_mod_m_sk": [c.inv_prod_B_msk],
             "prod_B_mod_q": c.prod_B_q, "q_Bsk_merge_modulus": c.mm,
             "q_Bsk_merge_ntt_tables": summarize(c.mtab), "q_Bsk_merge_intt_tables": summarize(c.mitab),
             "q_Bsk_n_inverse": c.mninv}}
    # NTT of a seeded limb, checked against the definition at a few slots
    x = fill_poly(7, 0, n, c.primes[0])
    y = ntt(x, c.tab[0], c.primes[0])
    for j, v in ntt_by_definition(x, c.psi[0], c.primes[0], n_power, [0, 1, 2, 1234, 4095]).items():
        assert y[j] == v, "iterative NTT disagrees with the definition"
    assert intt(y, c.itab[0], c.primes[0], c.ninv[0]) == x
    g["ntt"] = {"input": "fill_poly(seed=7, limb=0, q=primes[0])", "forward": summarize(y),
                "inverse_of_input": summarize(intt(x, c.itab[0], c.primes[0], c.ninv[0]))}
    ct1 = synth_ct(c.primes, range(2), 2, n, 1)
    ct2 = synth_ct(c.primes, range(2), 2, n, 2)
    key = synth_key(c.primes, 2, 3, n, 3)
    add = [[[(a + b) % c.primes[j] for a, b in zip(ct1[p][j], ct2[p][j])] for j in range(2)] for p in range(2)]
    sub = [[[(a - b) % c.primes[j] for a, b in zip(ct1[p][j], ct2[p][j])] for j in range(2)] for p in range(2)]
    mul = bfv_multiply(c, ct1, ct2)
    rel = bfv_relinearize(c, mul, key)
    gal = 3  # steps_to_galois_elt(1, n, 3)
    rot = bfv_apply_galois(c, ct1, key, gal)
    g["ops"] = {"inputs": "ct1=synth_ct(seed 1), ct2=synth_ct(seed 2), key=synth_key(seed 3) (tests/helpers.py)",
                "add": summarize(add), "sub": summarize(sub), "multiply": summarize(mul),
                "multiply_relinearize": summarize(rel), "galois_elt": gal, "rotate_rows_1": summarize(rot)}
    return g


def golden_ckks_small():
    """CKKS N=2^12 {40,30,30}|{40} (benchmark_ckks.cpp:17-22), depths 0 and 1."""
    n_power, n = 12, 4096
    primes = generate_primes(n, [40, 30, 30, 40])
    c = Ctx("ckks", n_power, primes, 3, 1)
    g = {"config": "CKKS N=4096 Q{40,30,30} P{40}", "primes": primes, "psi": c.psi, "n_inverse": c.ninv,
         "last_q_modinv": c.lqm, "half": c.half, "half_mod": c.half_mod, "factor": c.factor,
         "rescaled_half": c.r_half, "rescaled_half_mod": c.r_half_mod, "rescaled_last_q_modinv": c.r_lqm,
         "new_prime_locations": c.prime_loc, "new_input_locations": c.input_loc,
         "tables": {"ntt_table": summarize(c.tab), "intt_table": summarize(c.itab)}, "ops": {}}
    key = synth_key(primes, 3, 4, n, 3)
    for depth in (0, 1):
        l = 3 - depth
        ct1 = synth_ct(primes, range(l), 2, n, 1)
        ct2 = synth_ct(primes, range(l), 2, n, 2)
        mul = cross_multiplication(ct1, ct2, primes[:l])
        rel = ckks_relinearize(c, mul, key, depth)
        res = ckks_rescale(c, rel, depth)
        g5 = 5  # steps_to_galois_elt(1, n, 5)
        rot = ckks_apply_galois(c, ct1, key, g5, depth)
        g["ops"]["depth%d" % depth] = {"multiply": summarize(mul), "relinearize": summarize(rel),
                                       "rescale": summarize(res), "galois_elt": g5, "rotate_1": summarize(rot)}
    return g


def golden_params():
    """prime chains / psi for the benchmark configs (deterministic derivation)."""
    out = {}
    for name, n, bits in (("C2 CKKS N=2^14 {50,40x7}|{50}", 16384, [50] + [40] * 7 + [50]),
                          ("C4 CKKS N=2^16 {60,50x15}|{60}", 65536, [60] + [50] * 15 + [60])):
        primes = generate_primes(n, bits)
        out[name] = {"n": n, "bits": bits, "primes": primes,
                     "psi": [min_primitive_root(2 * n, q) for q in primes[:2]] + ["..."]}
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    for name, fn in (("c1_bfv_4096.json", golden_c1), ("ckks_4096.json", golden_ckks_small),
                     ("params.json", golden_params)):
        data = fn()
        with open(os.path.join(GOLD, name), "w") as f:
            json.dump(data, f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    sys.exit(main())
