#!/usr/bin/env python3
"""One parameterised driver for profiler runs (rocprofv3 --kernel-trace --stats -- python tools/prof.py ...; or under
tools/prof_all.sh for the counter passes).  It sets a workload up with random residues and runs it --reps times; results
are not checked here (the tests and bench.py do that).  Replaces the per-workload scripts of rounds 1-5.

  --workload c4_relin | c2 | c3 | bfv_mul | bfv_relin | m2 | hoist | rescale | scan | probe
  --logn N   --limbs Q   --batch B   --reps R   (defaults per workload: the shapes bench.py's `secondary` block quotes)

The eight workloads the bench line itself quotes are `python bench.py --profile-workload <name>` (tools/profile.sh)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import heongpu_amd as hg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", required=True)
ap.add_argument("--logn", type=int, default=0)
ap.add_argument("--limbs", type=int, default=0)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--reps", type=int, default=0)
a = ap.parse_args()
W = a.workload
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")


def ckks(logn, log_q, log_p):
    c = hg.Context.from_bit_sizes(hg.CKKS, 1 << logn, log_q, log_p, sec=hg.SEC_NONE)
    c.upload()
    return c, 1 << logn, c.Q_size, c.Q_prime_size


def bfv(logn):
    c = hg.Context.from_default(hg.BFV, 1 << logn, 1, plain_modulus=786433)
    c.upload()
    return c, 1 << logn, c.Q_size, c.Q_prime_size


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.3f ms per repetition" % (W, e0.elapsed_time(e1) / reps))


if W == "c4_relin":      # relinearize_inplace on a {60, 50 x (Q-1)} | {60} chain (default: config C4's, 64 ciphertexts)
    logn, nq, B = a.logn or 16, a.limbs or 16, a.batch or 64
    c, n, Q, Qp = ckks(logn, [60] + [50] * (nq - 1), [60])
    ob, key, ws = r(3 * Q * n * B), r(Q * 2 * Qp * n), c.workspace(hg.OP_CKKS_RELIN, 0, B)
    timed(lambda: c.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws), a.reps or 20)
elif W == "c2":          # config C2: CKKS N = 2^14 {50, 40 x 7} | {50}: multiply + relinearize + rescale
    B = a.batch or 1
    c, n, Q, Qp = ckks(14, [50] + [40] * 7, [50])
    key, c1, c2 = r(Q * 2 * Qp * n), r(2 * Q * n * B), r(2 * Q * n * B)
    ob = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    ws, ws2 = c.workspace(hg.OP_CKKS_RELIN, 0, B), c.workspace(hg.OP_CKKS_RESCALE, 0, B)

    def f():
        c.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, ob, 3 * Q * n, 0, B)
        c.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
        c.ckks_rescale_inplace(ob, 3 * Q * n, 0, B, ws2)
    timed(f, a.reps or 20)
elif W == "c3":          # config C3: BFV N = 2^15 default chain, rotate_rows, 64 ciphertexts
    B = a.batch or 64
    c, n, Q, Qp = bfv(a.logn or 15)
    ct, key = r(2 * Q * n * B), r(Q * 2 * Qp * n)
    out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
    ws, gal = c.workspace(hg.OP_BFV_GALOIS, 0, B), hg.steps_to_galois_elt(1, n, 3)
    timed(lambda: c.bfv_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, B, ws), a.reps or 6)
elif W in ("bfv_mul", "bfv_relin"):   # BFV default chain (default N = 2^14, 256 ciphertexts)
    B = a.batch or 256
    c, n, Q, Qp = bfv(a.logn or 14)
    c1, c2, o3 = r(2 * Q * n * B), r(2 * Q * n * B), r(3 * Q * n * B)
    if W == "bfv_mul":
        ws = c.workspace(hg.OP_BFV_MULTIPLY, 0, B)
        timed(lambda: c.bfv_multiply(c1, 2 * Q * n, c2, 2 * Q * n, o3, 3 * Q * n, B, ws), a.reps or 6)
    else:
        key, ws = r(Q * 2 * Qp * n), c.workspace(hg.OP_BFV_RELIN, 0, B)
        timed(lambda: c.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, ws), a.reps or 6)
elif W == "m2":          # method II relinearize: CKKS N = 2^16, Q = 16 x 50 bits, P = 4 x 50 bits
    B = a.batch or 64
    c, n, Q, Qp = ckks(16, [50] * 16, [50] * 4)
    out, key, ws = r(3 * Q * n * B), r(4 * 2 * Qp * n), c.workspace(hg.OP_CKKS_RELIN, 0, B)
    timed(lambda: c.ckks_relinearize_inplace(out, 3 * Q * n, key, 0, B, ws), a.reps or 5)
elif W == "hoist":       # hoisted rotations on config C4's chain: B ciphertexts x 8 Galois elements
    B, K = a.batch or 16, 8
    c, n, Q, Qp = ckks(16, [60] + [50] * 15, [60])
    ct, keys = r(2 * Q * n * B), [r(Q * 2 * Qp * n) for _ in range(K)]
    elts, words = [hg.steps_to_galois_elt(s + 1, n, 5) for s in range(K)], 2 * Q * n
    out = torch.empty(B * K * words, dtype=torch.int64, device="cuda")
    ws = c.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, B)
    timed(lambda: c.ckks_rotate_hoisted(ct, words, out, K * words, keys, elts, 0, B, ws), a.reps or 4)
elif W == "rescale":     # rescale_inplace on config C4's chain
    B = a.batch or 64
    c, n, Q, Qp = ckks(16, [60] + [50] * 15, [60])
    ct, ws = r(3 * Q * n * B), c.workspace(hg.OP_CKKS_RESCALE, 0, B)
    timed(lambda: c.ckks_rescale_inplace(ct, 3 * Q * n, 0, B, ws), a.reps or 5)
elif W == "probe":       # the launch groups of the C4 relinearize one by one (hegpu_probe_ckks_relinearize phases 1..16)
    B = a.batch or 64
    c, n, Q, Qp = ckks(16, [60] + [50] * 15, [60])
    out, key, ws = r(3 * Q * n * B), r(Q * 2 * Qp * n), c.workspace(hg.OP_CKKS_RELIN, 0, B)
    for ph in (1, 2, 4, 8, 16):
        W = "phase %d" % ph
        timed(lambda: c.probe_ckks_relinearize(out, 3 * Q * n, key, 0, B, ws, ph), a.reps or 5)
elif W == "scan":        # a mixed operator sequence on one parameter set: looking for kernels out of proportion
    logn, B = a.logn or 14, a.batch or 16
    nq = a.limbs or {12: 2, 13: 4, 14: 8, 15: 14, 16: 16}[logn]
    c, n, Q, Qp = ckks(logn, [60] + [40] * (nq - 1), [60])
    key, gkey, c1, c2 = r(Q * 2 * Qp * n), r(Q * 2 * Qp * n), r(2 * Q * n * B), r(2 * Q * n * B)
    ob = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    rot = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
    ws, ws2, ws3 = c.workspace(hg.OP_CKKS_RELIN, 0, B), c.workspace(hg.OP_CKKS_RESCALE, 0, B), c.workspace(hg.OP_CKKS_GALOIS, 0, B)
    g = hg.steps_to_galois_elt(1, n, 5)

    def f():
        c.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, ob, 3 * Q * n, 0, B)
        c.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
        c.ckks_apply_galois(ob, 3 * Q * n, rot, 2 * Q * n, gkey, g, 0, B, ws3)
        c.ckks_rescale_inplace(ob, 3 * Q * n, 0, B, ws2)
    timed(f, a.reps or 5)
else:
    raise SystemExit("unknown workload " + W)
