"""CPU checks of the exactness argument behind the FP64 paths (tests/fp_model.py): the bounds claimed in the kernels'
comments against rigorous ones for the concrete moduli, every (y, w) pair in reduced binary formats, and an
adversarial search of the double-precision arithmetic itself.  (The device side: tests/test_gpu_fp_audit.py.)"""
import random
from fractions import Fraction

import pytest

import fp_model as fm


def _c4_fp_primes():
    """the 50-bit primes of config C4's chain ({60, 50 x 15} | {60}, N = 2^16) from the library's own prime search"""
    import heongpu_amd as hg
    c = hg.Context.from_bit_sizes(hg.CKKS, 65536, [60] + [50] * 15, [60])
    return [int(v) for v in c.table("modulus") if int(v) < 2 ** 50]


@pytest.fixture(scope="module")
def primes50():
    p = _c4_fp_primes()
    assert len(p) == 15 and all(2 ** 49 < v < 2 ** 50 for v in p)
    return p


def test_every_schedule_stays_exact(primes50):
    """(L3): for every schedule of butterflies and reductions the kernels run, with the recurrences CLAIMED in the
    comments and with the RIGOROUS per-modulus ones: all products satisfy the exactness conditions of (L1), every
    magnitude is below 2^53, and the rigorous bound never exceeds the claimed one (so the claimed schedule -- where
    the reductions sit -- is safe)."""
    worst_q = [max(primes50), min(primes50), 2 ** 50 - 1]  # 2^50 - 1: the limit the plan builder admits (bit <= 50)
    for q in worst_q:
        tracks = []
        for n_power in range(12, 17):
            for unreduced in (True, False):
                tracks.append(("forward 2^%d" % n_power, lambda r, n=n_power, u=unreduced: fm.sched_forward(q, n, r, u)))
            tracks.append(("inverse 2^%d" % n_power, lambda r, n=n_power: fm.sched_inverse(q, n, r)))
            for digits in (1, 2, 3, 4, 16, 17, 32, 64):
                tracks.append(("keyswitch 2^%d l=%d" % (n_power, digits),
                               lambda r, n=n_power, d=digits: fm.sched_keyswitch(q, n, d, r)))
        for name, mk in tracks:
            claimed, rigorous = mk(False), mk(True)
            assert claimed.ok, (name, q, fm.format_rows(claimed))
            assert rigorous.ok, (name, q, fm.format_rows(rigorous))
            for (lc, bc), (lr, br) in zip(claimed.rows, rigorous.rows):
                assert lc == lr
                # (a reduction's own output: the rigorous 1/2 + 2^-49 is inside the claimed 1/2 (1 + 2^-40))
                assert br <= bc, (name, q, lc, float(br), float(bc))
                assert bc * q < 2 ** 53, (name, lc)
            if name.startswith("forward") or name.startswith("keyswitch"):
                assert max(b for l, b in claimed.rows if l.startswith("s") and l != "sums") <= fm.FP_BOUND_LIMIT
    # the numbers quoted in ntt.hip
    ks = dict(fm.sched_keyswitch(2 ** 50 - 1, 16, 16, False).rows)
    assert abs(float(ks["s15"]) - 5.22) < 0.01 and float(ks["product |t|"]) == 2.46 and abs(float(ks["sums"]) - 7.88) < 0.001
    assert float(dict(fm.sched_keyswitch(2 ** 50 - 1, 16, 16, True).rows)["sums"]) < 6.7


def test_tfhe_schedule_stays_exact():
    q = 2 ** 44 - 1  # the blind rotate's prime is the largest 44-bit one = 1 (mod 2048)
    claimed, rigorous = fm.sched_tfhe(q, False), fm.sched_tfhe(q, True)
    assert claimed.ok and rigorous.ok
    c, r = dict(claimed.rows), dict(rigorous.rows)
    assert float(r["s9"]) < 5.1 < 7.0  # "below 7 p'" (tfhe.hip fwave_ntt1024_l)
    assert float(r["product |t|"]) <= 0.53 and float(r["sums"]) <= 2.2
    assert float(r["i4"]) * q < 2 ** 53 and float(c["i4"]) * q < 2 ** 53  # 64 x the sums, the one place far above 8 q
    for k in r:
        if k != "in" and not k.startswith("red"):
            assert r[k] <= c[k], k


@pytest.mark.parametrize("p", [11, 12, 13])
def test_exhaustive_in_reduced_formats(p):
    """(L1) + (L2) for EVERY input pair of a format with p significand bits and the largest primes below 2^(p-3):
    2^53 -> 2^p, 2^50 -> 2^(p-3) -- the argument has no other constant in it."""
    primes = [v for v in range(2 ** (p - 3) - 1, 2, -2) if all(v % d for d in range(3, int(v ** 0.5) + 1, 2))]
    for q in primes[:2 if p < 13 else 1]:
        r = fm.exhaustive_fp_mul(p, q, 7.9, "table")
        assert r["inexact"] == 0 and r["pairs"] > 0
        rig = fm.quotient_error(Fraction(79, 10), Fraction(q - 1, q), q, "table", p)
        assert r["worst_slope"] <= 0.25 and r["worst_t"] <= float(rig)
        r = fm.exhaustive_fp_mul(p, q, 5.3, "recomputed")
        assert r["inexact"] == 0 and r["worst_slope"] <= 0.375
        assert r["worst_t"] <= float(fm.quotient_error(Fraction(53, 10), Fraction(q - 1, q), q, "recomputed", p))
        r = fm.exhaustive_fp_mul(p, q, None, "product", b_w=5.22)
        assert r["inexact"] == 0 and r["worst_t"] <= 2.46
        assert r["worst_t"] <= float(fm.quotient_error(Fraction(q - 1, q), Fraction(522, 100), q, "recomputed", p))
        r = fm.exhaustive_fp_reduce(p, q)
        assert r["inexact"] == 0 and r["worst"] <= float(fm.reduce_out(q, p))


def test_adversarial_search_in_double_precision(primes50):
    """The double-precision arithmetic itself, searched for inputs that come closest to the bounds: worst twiddles
    (companion furthest from w / q), operands at the largest admitted magnitude whose exact quotient sits next to a
    rounding boundary, random restarts.  Every product exact; nothing found above the rigorous bound."""
    rng = random.Random(2026)
    found = {}
    for q in (max(primes50), min(primes50)):
        for companion, claim in (("table", Fraction(1, 4)), ("recomputed", Fraction(3, 8))):
            tws = fm.worst_twiddles(q, 16, rng, companion, 20000)
            for b in (Fraction(105, 100), Fraction(41, 10), Fraction(73, 10)):
                worst, exact = fm.search_fp_mul(q, b, companion, rng, 6000, tws)
                assert exact
                rig = fm.quotient_error(b, Fraction(q - 1, q), q, companion)
                assert worst <= rig <= Fraction(1, 2) + claim * b, (q, companion, float(b), float(worst), float(rig))
                found[(companion, float(b))] = max(found.get((companion, float(b)), 0.0), float(worst))
        worst, exact = fm.search_product(q, Fraction(522, 100), rng, 6000)
        assert exact and worst <= fm.quotient_error(Fraction(q - 1, q), Fraction(522, 100), q, "recomputed") <= Fraction(246, 100)
        found["product"] = max(found.get("product", 0.0), float(worst))
    # the search is not vacuous: it gets well beyond the 1/2 of an exact quotient.  (At b = 7.3 the true worst case is
    # 1/4 + 1/2 + 7.3 / 16 = 1.206: y w' just below 2^52 -- half an ulp is 1/4 there, the rint adds 1/2 -- with a
    # twiddle just above q / 2, whose companion is off by up to 2^-54.)
    assert found[("table", 7.3)] > 1.19 and found["product"] > 1.0, found
    # a greedy adversary (free choice of both operands at every stage: stronger than any data flow) against the
    # claimed recurrence of the column stages, five stages from an un-reduced input
    q = max(primes50)
    reached = fm.greedy_chain(q, Fraction(105, 100), 5, "table", rng, 2000)
    claimed = [b for l, b in fm.sched_forward(q, 16, False).rows if l in ("s0", "s1", "s2", "s3", "s4")]
    assert all(r <= c for r, c in zip(reached, claimed)) and reached[-1] > 4.5, [float(r) for r in reached]
