"""Textbook RLWE helpers (python big ints + the oracle's NTT) used by the
semantic round-trip tests: keygen in the reference's key layout
(keygeneration.cu:145-185), coefficient-level encrypt/decrypt, CRT compose.
Randomness is numpy's -- the reference's RNG can never be bit-matched
(SURVEY.md 8f next-1); only distributions are mirrored (ternary secret,
Gaussian sigma=3.2 error, uniform a)."""
import numpy as np


class RLWE:
    def __init__(self, octx, seed=0):
        self.o = octx
        self.n = octx.n
        self.Q, self.Qp = octx.Q, octx.Qp
        self.primes = octx.primes
        self.rng = np.random.default_rng(seed)
        self.s = self.rng.integers(-1, 2, self.n)  # ternary
        self.s_ntt = self.to_ntt(self.s, range(self.Qp))

    # --- representation changes
    def to_rns(self, poly, limb_ids):
        return np.stack([np.array([int(v) % self.primes[j] for v in poly], dtype=np.uint64) for j in limb_ids])

    def ntt_limbs(self, rns, limb_ids, inverse=False):
        out = rns.copy()
        for r, j in enumerate(limb_ids):
            row = np.ascontiguousarray(out[r])
            self.o.ntt(row, 1, 1, mod_offset=j, inverse=inverse)
            out[r] = row
        return out

    def to_ntt(self, poly, limb_ids):
        limb_ids = list(limb_ids)
        return self.ntt_limbs(self.to_rns(poly, limb_ids), limb_ids)

    def mulmod(self, a, b, q):
        return np.array([(int(x) * int(y)) % q for x, y in zip(a, b)], dtype=np.uint64)

    def addmod(self, a, b, q):
        return np.array([(int(x) + int(y)) % q for x, y in zip(a, b)], dtype=np.uint64)

    def negmod(self, a, q):
        return np.array([(q - int(x)) % q for x in a], dtype=np.uint64)

    def error(self):
        return np.rint(self.rng.normal(0, 3.2, self.n)).astype(np.int64)

    def uniform_ntt(self, limb_ids):
        return np.stack([np.array([int(self.rng.integers(0, self.primes[j])) for _ in range(self.n)],
                                  dtype=np.uint64) for j in limb_ids])

    # --- keys (method I layout [digit i][c][limb j][n], NTT domain)
    def switch_key(self, target_ntt, secret_ntt=None):
        """key[i][0] = -(a_i*secret + e_i) + [j==i] (P mod q_j) target, key[i][1] = a_i.
        relin (keygeneration.cu:145-185): secret = s, target = s^2;
        galois (keygeneration.cu:757-805): secret = sigma_{g^-1}(s), target = s."""
        if secret_ntt is None:
            secret_ntt = self.s_ntt
        Q, Qp, n = self.Q, self.Qp, self.n
        P = self.primes[Qp - 1]
        key = np.zeros((Q, 2, Qp, n), dtype=np.uint64)
        for i in range(Q):
            a = self.uniform_ntt(range(Qp))
            e = self.to_ntt(self.error(), range(Qp))
            for j in range(Qp):
                q = self.primes[j]
                k0 = self.negmod(self.addmod(self.mulmod(a[j], secret_ntt[j], q), e[j], q), q)
                if j == i:
                    f = P % q
                    k0 = self.addmod(k0, self.mulmod(target_ntt[j], np.full(n, f, dtype=np.uint64), q), q)
                key[i, 0, j] = k0
                key[i, 1, j] = a[j]
        return key.reshape(-1)

    def switch_key_II(self, target_ntt, m, secret_ntt=None):
        """method II (hybrid) key: digit i covers limbs [i*m, min((i+1)*m, Q));
        key[i][0][j] = -(a_i*secret + e_i) + [j in digit i] (P mod q_j) target with
        P = product of ALL special primes (keygeneration.cu:584-629); layout [d][2][Q'][N]."""
        if secret_ntt is None:
            secret_ntt = self.s_ntt
        Q, Qp, n = self.Q, self.Qp, self.n
        P = 1
        for j in range(Q, Qp):
            P *= self.primes[j]
        d = (Q + m - 1) // m
        key = np.zeros((d, 2, Qp, n), dtype=np.uint64)
        for i in range(d):
            a = self.uniform_ntt(range(Qp))
            e = self.to_ntt(self.error(), range(Qp))
            for j in range(Qp):
                q = self.primes[j]
                k0 = self.negmod(self.addmod(self.mulmod(a[j], secret_ntt[j], q), e[j], q), q)
                if j < Q and j // m == i:
                    k0 = self.addmod(k0, self.mulmod(target_ntt[j], np.full(n, P % q, dtype=np.uint64), q), q)
                key[i, 0, j] = k0
                key[i, 1, j] = a[j]
        return key.reshape(-1)

    def relin_key_II(self, m):
        s2 = np.stack([self.mulmod(self.s_ntt[j], self.s_ntt[j], self.primes[j]) for j in range(self.Qp)])
        return self.switch_key_II(s2, m)

    def galois_key_II(self, g, m):
        g_inv = pow(g, -1, 2 * self.n)
        sg = self.apply_galois_poly(self.s.astype(object), g_inv)
        return self.switch_key_II(self.s_ntt, m, self.to_ntt(sg, range(self.Qp)))

    def relin_key(self):
        s2 = np.stack([self.mulmod(self.s_ntt[j], self.s_ntt[j], self.primes[j]) for j in range(self.Qp)])
        return self.switch_key(s2)

    def apply_galois_poly(self, poly, g):
        n = self.n
        out = np.zeros(n, dtype=object)
        for i in range(n):
            r = i * g
            idx = r % n
            out[idx] = -poly[i] if (r // n) & 1 else poly[i]
        return out

    def galois_key(self, g):
        g_inv = pow(g, -1, 2 * self.n)
        sg = self.apply_galois_poly(self.s.astype(object), g_inv)
        return self.switch_key(self.s_ntt, self.to_ntt(sg, range(self.Qp)))

    # --- encryption of an integer polynomial `m` on limbs 0..l-1
    def encrypt(self, m, l, ntt_domain):
        ids = list(range(l))
        a = self.uniform_ntt(ids)
        e_m = self.to_ntt(self.error().astype(object) + np.array(m, dtype=object), ids)
        c0 = np.stack([self.addmod(self.negmod(self.mulmod(a[j], self.s_ntt[j], self.primes[j]), self.primes[j]),
                                   e_m[j], self.primes[j]) for j in ids])
        ct = np.stack([c0, a])
        if not ntt_domain:
            ct = np.stack([self.ntt_limbs(ct[p], ids, inverse=True) for p in range(2)])
        return ct.reshape(-1)

    def decrypt(self, ct, l, parts, ntt_domain):
        """returns the centered integer polynomial c0 + c1 s (+ c2 s^2) mod q_0..q_{l-1}."""
        ids = list(range(l))
        ct = ct.reshape(parts, l, self.n)
        if not ntt_domain:
            ct = np.stack([self.ntt_limbs(ct[p], ids) for p in range(parts)])
        acc = ct[0].copy()
        spow = [self.s_ntt[j].copy() for j in ids]
        for p in range(1, parts):
            for j in ids:
                q = self.primes[j]
                acc[j] = self.addmod(acc[j], self.mulmod(ct[p][j], spow[j], q), q)
                spow[j] = self.mulmod(spow[j], self.s_ntt[j], q)
        coeff = self.ntt_limbs(acc, ids, inverse=True)
        return self.crt_centered(coeff, ids)

    def crt_centered(self, rns, ids):
        M = 1
        for j in ids:
            M *= self.primes[j]
        out = np.zeros(self.n, dtype=object)
        for r, j in enumerate(ids):
            q = self.primes[j]
            Mi = M // q
            f = Mi * pow(Mi % q, -1, q)
            for k in range(self.n):
                out[k] = (out[k] + int(rns[r][k]) * f) % M
        for k in range(self.n):
            if out[k] > M // 2:
                out[k] -= M
        return out, M


def negacyclic_mul(a, b):
    n = len(a)
    a = [int(x) for x in a]
    b = [int(x) for x in b]
    full = np.convolve(np.array(a, dtype=object), np.array(b, dtype=object))
    out = np.zeros(n, dtype=object)
    for i, v in enumerate(full):
        if i < n:
            out[i] += v
        else:
            out[i - n] -= v
    return out
