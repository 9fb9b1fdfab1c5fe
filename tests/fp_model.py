"""Host-side exact model of the FP64 modular arithmetic of heongpu_amd/csrc/fpmod.cuh -- TEST INFRASTRUCTURE.

The forward / inverse transforms of moduli below 2^50, the fused key-switch inner product (ntt.hip) and the TFHE
blind rotate (tfhe.hip) keep INTEGERS in doubles.  Their bit-exactness rests on three statements, made in comments
next to the code (fpmod.cuh:5-21, ntt.hip FP_STAGE_GROW / fp_ct_radix16_tb8 / ks_row_mac_fp_body, tfhe.hip
fwave_ntt1024_l):

  (L1) fp_mul(y, w, w') returns t = y w - k q EXACTLY for an integer k, whenever |k| < 2^53 and |t| + ulp(y w)/2 < 2^53;
  (L2) |t| <= (1/2 + e) q, where e bounds the error of the quotient estimate rint(RN(y w')):
         table companion w' = RN(w/q), |w| < q, |y| <= b q:            e <= 0.25 b    (q < 2^50)
         recomputed companion w' = RN(w RN(1/q)):                      e <= 0.375 b
         product of the key switch (y < q, |w| <= 5.22 q, recomputed): |t| <= 2.46 q
  (L3) hence the magnitudes along each schedule of butterflies / reductions stay below 2^53.

This module (a) restates the arithmetic for a binary format with P significand bits (P = 53: Python floats and
exact integers; small P: numpy, exact in float64), (b) derives RIGOROUS bounds for (L2) in exact rationals for a
concrete modulus, (c) evaluates every schedule the kernels use with the claimed and with the rigorous recurrences,
(d) checks (L1)/(L2) EXHAUSTIVELY -- every (y, w) pair -- in reduced formats (P = 11..14, q < 2^(P-3), where the
statements scale: 2^53 -> 2^P, 2^50 -> 2^(P-3)), and (e) searches the double-precision arithmetic adversarially
(constructed worst cases + random restarts) for inputs that come closest to the bounds.  tests/test_fp_model.py runs
(b)-(e) on the CPU; tests/test_gpu_fp_audit.py compares the device's measured maxima (tests/audit) with (c).
"""
import math
import random
from fractions import Fraction

import numpy as np

# ------------------------------------------------------------------ (a) the arithmetic, P = 53 (exact, scalar)


def _rint_half_even(x):
    """rint() of a float as a Python int (ties to even, as __builtin_rint in the default rounding mode)."""
    return int(round(x))  # Python's round(float) is round-half-even and exact


def fp_mul(y, wx, wy, q):
    """fpmod.cuh fp_mul.  y, wx: integer-valued floats; wy: float; q: int.  Returns (t, k, exact) with `exact` True
    iff both FMAs and the final sum were exact, i.e. t == y*wx - k*q as integers."""
    yi, wi = int(y), int(wx)
    assert float(yi) == y and float(wi) == wx, "operands must be integers"
    h = y * wx  # RN(y w)
    if math.isinf(h):
        return float("nan"), 0, False
    hi = int(h)
    l = float(yi * wi - hi)  # fma(y, w, -h): the exact difference, then one rounding
    k = _rint_half_even(y * wy)
    v1 = hi - k * q
    f1 = float(v1)  # fma(-k, q, h)
    t = f1 + l
    exact = (int(l) == yi * wi - hi) and (int(f1) == v1) and (int(t) == yi * wi - k * q) and abs(k) < 2 ** 53
    return t, k, exact


def fp_reduce(x, q, qi):
    """fpmod.cuh fp_reduce: x - rint(x * qi) * q through one FMA."""
    xi = int(x)
    k = _rint_half_even(x * qi)
    v = xi - k * q
    r = float(v)
    return r, int(r) == v


def companion_table(w, q):
    """RN(w / q) -- the (w, w') pairs of the twiddle tables (context.cpp)."""
    return w / q if isinstance(w, int) else float(Fraction(int(w)) / q)


def companion_recomputed(w, q):
    """RN(w * RN(1/q)) -- ntt.hip fp_ct_radix16_tb8 / ks_row_mac_fp_body / tfhe.hip f_ct_l."""
    return float(w) * (1.0 / float(q))


# ------------------------------------------------------------------ (b) rigorous bounds, exact rationals
def _half_ulp(bound, p):
    """|RN(v) - v| <= this for every |v| <= bound (binary format, p significand bits, no underflow)."""
    bound = Fraction(bound)
    if bound <= 0:
        return Fraction(0)
    e = bound.numerator.bit_length() - bound.denominator.bit_length()
    if Fraction(2) ** e > bound:
        e -= 1  # 2^e <= bound < 2^(e+1)
    return Fraction(2) ** (e - p)


def quotient_error(b_y, b_w, q, companion, p=53):
    """Rigorous bound on |k - y w / q| for |y| <= b_y q, |w| <= b_w q (b_w < 1 for a twiddle: pass (q-1)/q).
    companion: 'table' (w' = RN(w/q)) or 'recomputed' (w' = RN(w RN(1/q)))."""
    b_y, b_w, q = Fraction(b_y), Fraction(b_w), Fraction(q)
    if companion == "table":
        d1 = _half_ulp(b_w, p)  # |w' - w/q|
    else:
        qi_err = _half_ulp(1 / q, p)  # |RN(1/q) - 1/q|
        qi_max = 1 / q + qi_err
        d1 = b_w * q * qi_err + _half_ulp(b_w * q * qi_max, p)
    wp_max = b_w + d1
    prod_max = b_y * q * wp_max  # |y w'|
    return Fraction(1, 2) + _half_ulp(prod_max, p) + b_y * q * d1


def mul_is_exact(b_y, b_w, q, e, p=53):
    """The conditions of (L1) for |y| <= b_y q, |w| <= b_w q and a quotient error bound e: |k| < 2^p and
    |h - k q| <= e q + ulp(y w)/2 < 2^p (an integer: representable, so both FMAs and the final sum are exact)."""
    b_y, b_w, q = Fraction(b_y), Fraction(b_w), Fraction(q)
    k_max = b_y * b_w * q + e
    h_err = _half_ulp(b_y * b_w * q * q * (1 + Fraction(1, 2 ** (p - 1))), p)
    return k_max < 2 ** p and e * q + h_err < 2 ** p


def reduce_out(q, p=53):
    """|fp_reduce(x)| / q for any integer |x| < 2^p (quotient estimate RN(x RN(1/q)))."""
    q = Fraction(q)
    x = Fraction(2) ** p
    qi_err = _half_ulp(1 / q, p)
    e = Fraction(1, 2) + x * qi_err + _half_ulp(x * (1 / q + qi_err), p)
    return e


class Track:
    """Magnitude bound (units of q) along a schedule; records the bound after every step and checks (L1) at every
    product.  rigorous=True uses quotient_error() for the concrete q, False the recurrences claimed in the comments."""

    CLAIM = {"table": Fraction(1, 4), "recomputed": Fraction(3, 8)}

    def __init__(self, q, b, rigorous, p=53, claim_e=None):
        self.q, self.b, self.rig, self.p = q, Fraction(b), rigorous, p
        self.claim_e = claim_e  # a schedule whose comments claim a constant |t| / q per butterfly (tfhe.hip: 0.51)
        self.rows = []  # (label, bound after)
        self.ok = True
        self.tw = Fraction(q - 1, q)

    def _e(self, b_y, b_w, companion):
        if self.rig:
            e = quotient_error(b_y, b_w, self.q, companion, self.p)
        elif self.claim_e is not None:
            e = Fraction(self.claim_e)
        else:
            assert b_w <= 1
            e = Fraction(1, 2) + self.CLAIM[companion] * b_y
        if not mul_is_exact(b_y, b_w, self.q, e, self.p):
            self.ok = False
        return e

    def note(self, label):
        if self.b * self.q >= 2 ** self.p:
            self.ok = False
        self.rows.append((label, self.b))

    def ct(self, label, companion="table"):  # x' = x +- t, t = fp_mul(y, w)
        self.b = self.b + self._e(self.b, self.tw, companion)
        self.note(label)

    def gs(self, label, companion="table"):  # x' = x + y, y' = fp_mul(x - y, w)
        self._e(2 * self.b, self.tw, companion)
        self.b = 2 * self.b
        self.note(label)

    def reduce(self, label):
        self.b = reduce_out(self.q, self.p) if self.rig else Fraction(1, 2) * (1 + Fraction(1, 2 ** 40))
        self.note(label)

    def product(self, label, b_key, e_claim):
        """t = fp_mul(key, x, RN(x RN(1/q))): x (this track's bound) plays the twiddle"""
        if self.rig:
            e = quotient_error(b_key, self.b, self.q, "recomputed", self.p)
        else:
            e = Fraction(e_claim)
        if not mul_is_exact(b_key, self.b, self.q, e, self.p):
            self.ok = False
        return e


# the schedule of the forward column stages: ntt.hip fp_sched / FpColSched (claimed recurrence, limit 7.9)
FP_BOUND_LIMIT = Fraction(79, 10)
FP_HANDOVER = Fraction(51, 100)
FP_UNREDUCED_IN = Fraction(105, 100)


def fp_sched(stages, b_in, b_out_max):
    grow = lambda b: Fraction(5, 4) * b + Fraction(1, 2)
    before, b = 0, Fraction(b_in)
    for s in range(stages):
        if grow(b) > FP_BOUND_LIMIT:
            before |= 1 << s
            b = Fraction(1, 2)
        b = grow(b)
    return before, b > b_out_max


def sched_forward(q, n_power, rigorous, decomp_unreduced=True, p=53):
    """Forward transform of one limb as the two passes / the single pass run it: column stages 0..S1-1 (FpColSched),
    first row round (table companions, reduction), second row round (recomputed companions, canonical reduction)."""
    s1 = n_power - 8
    before, at_end = fp_sched(s1, FP_UNREDUCED_IN, FP_HANDOVER)
    t = Track(q, FP_UNREDUCED_IN if decomp_unreduced else 1, rigorous, p)
    t.note("in")
    for s in range(s1):
        if (before >> s) & 1:
            t.reduce("red<%d" % s)
        t.ct("s%d" % s)
    if at_end:
        t.reduce("red<%d" % s1)
    for s in range(4):
        t.ct("s%d" % (s1 + s))
    t.reduce("red<%d" % (s1 + 4))
    for s in range(4):
        t.ct("s%d" % (s1 + 4 + s), "recomputed")
    return t


def sched_keyswitch(q, n_power, digits, rigorous, p=53):
    """Row stages + inner product of ks_row_mac_fp: column hand-over |x| <= q/2 (1 + 2^-40), four stages (recomputed
    companions), reduction, four stages, x un-reduced into the product with the key (< q), sums re-centred after
    every third digit."""
    s1 = n_power - 8
    t = Track(q, 0, rigorous, p)
    t.reduce("in")
    for s in range(4):
        t.ct("s%d" % (s1 + s), "recomputed")
    t.reduce("red<%d" % (s1 + 4))
    for s in range(4):
        t.ct("s%d" % (s1 + 4 + s), "recomputed")
    e = t.product("product", Fraction(q - 1, q), Fraction(246, 100))
    t.rows.append(("product |t|", e))
    acc, since, worst = Fraction(0), 0, Fraction(0)
    red = reduce_out(q, p) if rigorous else Fraction(1, 2) * (1 + Fraction(1, 2 ** 40))
    for _ in range(digits):
        acc += e
        worst = max(worst, acc)
        since += 1
        if since == 3:
            since, acc = 0, red
    t.rows.append(("sums", worst))
    if worst * q >= 2 ** p:
        t.ok = False
    return t


def sched_inverse(q, n_power, rigorous, p=53):
    """ArFp: canonical input, Gentleman-Sande stages with a reduction after every second one (ntt.hip ArFp::radix /
    radix16_tb / radix_last), n^-1 in the last."""
    s1 = n_power - 8
    nsa = s1 - 4
    t = Track(q, 1, rigorous, p)
    t.note("in")
    for s in (3, 2, 1, 0):  # radix16_tb
        t.gs("s%d" % (s1 + 4 + s))
        if s in (2, 0):
            t.reduce("red>%d" % (s1 + 4 + s))
    done = 0
    for s in (3, 2, 1, 0):  # radix<4>, row part
        t.gs("s%d" % (s1 + s))
        done += 1
        if done % 2 == 0 or s == 0:
            t.reduce("red>%d" % (s1 + s))
    if nsa > 0:
        done = 0
        for s in (3, 2, 1, 0):  # radix<4>, column part round one
            t.gs("s%d" % (nsa + s))
            done += 1
            if done % 2 == 0 or s == 0:
                t.reduce("red>%d" % (nsa + s))
    last = nsa if nsa > 0 else 4
    done = 0
    for s in range(last - 1, 0, -1):  # radix_last
        t.gs("s%d" % s)
        done += 1
        if done % 2 == 0:
            t.reduce("red>%d" % s)
    t._e(2 * t.b, t.tw, "table")  # (x +- y) n^-1
    t.b = 2 * t.b
    t.note("s0 sum")
    return t


def sched_tfhe(q, rigorous, p=53):
    """Blind rotate (tfhe.hip fwave_ntt1024_l / external product / fwave_intt1024_l), prime q = p' (44 bits): digits
    |d| <= 2^9, ten stages without a reduction (four with table, six with recomputed companions), the four products
    key x digit-transform (|key| <= p'/2), nine inverse stages before the first reduction."""
    t = Track(q, Fraction(512, q), rigorous, p, claim_e=Fraction(51, 100))  # "at most 0.51 p' per stage" (fwave_ntt1024)
    t.note("in")
    for s in range(4):
        t.ct("s%d" % s)
    for s in range(4, 10):
        t.ct("s%d" % s, "recomputed")
    e = t.product("product", Fraction(1, 2), Fraction(53, 100))
    t.rows.append(("product |t|", e))
    t.b = 4 * e
    t.note("sums")
    for s in (9, 8, 7, 6, 5, 4):
        t.gs("i%d" % s, "recomputed")
    t.reduce("red>4")
    for s in (3, 2, 1):
        t.gs("i%d" % s)
    t._e(2 * t.b, t.tw, "table")
    t.b = 2 * t.b
    t.note("i0 sum")
    return t


# ------------------------------------------------------------------ (d) exhaustive check in reduced formats (numpy)
def _rn_p(x, p):
    """round-to-nearest-even to p significand bits; exact for float64 inputs (p <= 26)"""
    m, e = np.frexp(x)
    return np.ldexp(np.rint(np.ldexp(m, p)), e - p)


def exhaustive_fp_mul(p, q, b_max, companion, b_w=None, chunk=1 << 22):
    """Every pair (y, w): companion 'table'/'recomputed': |y| <= b_max q, w in [0, q) (a butterfly);
    companion 'product': y in [0, q), |w| <= b_w q with w' = RN(w RN(1/q)) (the key-switch product).
    Returns dict(pairs, inexact, worst=(max over pairs of |t|/q - 1/2) / (|y|/q or 1), worst_t)."""
    assert p <= 26 and q < 2 ** (p - 3)
    qf = float(q)
    qi = _rn_p(np.array([1.0 / qf]), p)[0]
    if companion == "product":
        ys = np.arange(0, q, dtype=np.float64)
        wb = int(b_w * q)
        ws = np.arange(-wb, wb + 1, dtype=np.float64)
        wps = _rn_p(ws * qi, p)
    else:
        yb = int(b_max * q)
        ys = np.arange(-yb, yb + 1, dtype=np.float64)
        ws = np.arange(0, q, dtype=np.float64)
        wps = _rn_p(ws / qf, p) if companion == "table" else _rn_p(ws * qi, p)
    inexact, worst_slope, worst_t, pairs = 0, 0.0, 0.0, 0
    rows = max(1, chunk // len(ws))
    for i0 in range(0, len(ys), rows):
        y = ys[i0:i0 + rows, None]
        w, wp = ws[None, :], wps[None, :]
        yw = y * w  # exact: <= 2^(2p) <= 2^52
        h = _rn_p(yw, p)
        l = _rn_p(yw - h, p)
        k = np.rint(_rn_p(y * wp, p))
        f1e = h - k * qf
        f1 = _rn_p(f1e, p)
        te = f1 + l
        t = _rn_p(te, p)
        bad = (l != yw - h) | (f1 != f1e) | (t != te) | (t != yw - k * qf) | (np.abs(k) >= 2.0 ** p)
        inexact += int(bad.sum())
        r = np.abs(yw - k * qf) / qf  # the true |y w - k q| / q: bounds hold for the exact value
        worst_t = max(worst_t, float(r.max()))
        if companion == "product":
            denom = 1.0
        else:
            denom = np.maximum(np.abs(y) / qf, 1e-300)
        slope = (r - 0.5) / denom
        worst_slope = max(worst_slope, float(slope.max()))
        pairs += y.size * w.size
    return dict(pairs=pairs, inexact=inexact, worst_slope=worst_slope, worst_t=worst_t)


def exhaustive_fp_reduce(p, q):
    """every integer |x| < 2^p: r == x (mod q) exactly, returns max |r| / q"""
    qf = float(q)
    qi = _rn_p(np.array([1.0 / qf]), p)[0]
    x = np.arange(-(2 ** p) + 1, 2 ** p, dtype=np.float64)
    k = np.rint(_rn_p(x * qi, p))
    re = x - k * qf
    r = _rn_p(re, p)
    return dict(count=int(x.size), inexact=int((r != re).sum()), worst=float(np.abs(re).max() / qf))


# ------------------------------------------------------------------ (e) adversarial search, P = 53
def worst_twiddles(q, count, rng, companion, samples=200000):
    """twiddle values whose companion is furthest from w/q (signed), both directions"""
    best = []
    for _ in range(samples):
        w = rng.randrange(1, q)
        wp = companion_table(w, q) if companion == "table" else companion_recomputed(w, q)
        d = Fraction(wp) - Fraction(w, q)
        best.append((d, w))
    best.sort()
    return [w for _, w in best[:count]] + [w for _, w in best[-count:]]


def search_fp_mul(q, b, companion, rng, trials=20000, tws=None):
    """maximise |y w - k q| / q over |y| <= b q, w in [0, q): candidates = worst twiddles x y's whose exact quotient
    sits next to a rounding boundary, plus random pairs.  Returns (worst |t|/q, all_exact)."""
    tws = tws or worst_twiddles(q, 16, rng, companion, 20000)
    worst, all_exact = Fraction(0), True
    ymax = int(Fraction(b) * q)
    for i in range(trials):
        w = tws[i % len(tws)] if i % 4 else rng.randrange(1, q)
        if i % 2:
            # y close to ymax with y w / q close to a half-integer: y = round((m + 1/2) q / w)
            m = (ymax * w) // q - rng.randrange(0, 1 << 12)
            y = ((2 * m + 1) * q) // (2 * w)
            y = min(max(y, -ymax), ymax)
        else:
            y = ymax - rng.randrange(0, 1 << 20)
        if rng.random() < 0.5:
            y = -y
        wp = companion_table(w, q) if companion == "table" else companion_recomputed(w, q)
        t, k, exact = fp_mul(float(y), float(w), wp, q)
        all_exact &= exact and float(y) == y
        worst = max(worst, Fraction(abs(y * w - k * q), q))
    return worst, all_exact


def search_product(q, b_x, rng, trials=20000):
    """the key-switch product: y (key) in [0, q), |x| <= b_x q un-reduced, companion RN(x RN(1/q))"""
    worst, all_exact = Fraction(0), True
    xmax = int(Fraction(b_x) * q)
    for i in range(trials):
        x = xmax - rng.randrange(0, 1 << 24)
        if i % 2:
            y = q - 1 - rng.randrange(0, 1 << 16)
        else:
            m = rng.randrange(1, 5 * (q - 1))
            y = min(q - 1, max(1, ((2 * m + 1) * q) // (2 * x)))
        if rng.random() < 0.5:
            x = -x
        t, k, exact = fp_mul(float(y), float(x), companion_recomputed(x, q), q)
        all_exact &= exact
        worst = max(worst, Fraction(abs(y * x - k * q), q))
    return worst, all_exact


def greedy_chain(q, b0, stages, companion, rng, trials=4000):
    """an adversary stronger than any real data flow: at every stage it picks x = +-b q and the (y, w) that the
    search found worst, so the sums add up; returns the magnitudes reached after each stage (units of q)"""
    tws = worst_twiddles(q, 16, rng, companion, 20000)
    b, out = Fraction(b0), []
    for _ in range(stages):
        e, ok = search_fp_mul(q, b, companion, rng, trials, tws)
        assert ok
        b = b + e
        out.append(b)
    return out


def format_rows(track):
    return ", ".join("%s %.3f" % (k, float(v)) for k, v in track.rows)


if __name__ == "__main__":
    import argparse
    import json

    ap = argparse.ArgumentParser(description="exhaustive reduced-format check / bound tables of the FP64 arithmetic")
    ap.add_argument("--exhaustive", type=int, default=0, help="P (11..14): every (y, w) pair for the largest prime below 2^(P-3)")
    ap.add_argument("--table", action="store_true", help="print the schedule bounds for the C4 primes' size")
    a = ap.parse_args()
    if a.exhaustive:
        p = a.exhaustive
        q = next(v for v in range(2 ** (p - 3) - 1, 2, -2) if all(v % d for d in range(3, int(v ** 0.5) + 1, 2)))
        for comp, kw in (("table", dict(b_max=7.9)), ("recomputed", dict(b_max=5.3)), ("product", dict(b_max=None, b_w=5.3))):
            r = exhaustive_fp_mul(p, q, companion=comp, **kw)
            print(json.dumps(dict(P=p, q=q, companion=comp, **r)))
        print(json.dumps(dict(P=p, q=q, reduce=exhaustive_fp_reduce(p, q))))
    if a.table:
        q = 2 ** 50 - 27  # any modulus just below 2^50 gives the same bounds to three digits
        for name, mk in (("forward N=2^%d" % n, lambda r, n=n: sched_forward(q, n, r)) for n in range(12, 17)):
            print(name, "| claimed:", format_rows(mk(False)), "| rigorous:", format_rows(mk(True)))
        print("keyswitch N=2^16 l=16 | claimed:", format_rows(sched_keyswitch(q, 16, 16, False)), "| rigorous:",
              format_rows(sched_keyswitch(q, 16, 16, True)))
        print("inverse N=2^16 | claimed:", format_rows(sched_inverse(q, 16, False)), "| rigorous:", format_rows(sched_inverse(q, 16, True)))
