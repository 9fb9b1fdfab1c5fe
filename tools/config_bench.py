#!/usr/bin/env python3
"""Timing probe for the BASELINE.json configs that bench.py does not quote:
C2 (CKKS N=2^14 L=8, one ciphertext, multiply+relinearize+rescale latency) and
C3 (BFV N=2^15 default chain, rotate_rows, 64 ciphertexts).  Synthetic data."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import heongpu_amd as hg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()


def rnd(n_elems, bound=1 << 30):
    return torch.randint(0, bound, (n_elems,), dtype=torch.int64, device="cuda")


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# ---- C2: CKKS N=2^14, Q = {60, 40 x 7} | P = {60}, batch 1
n = 1 << 14
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [40] * 7, [60], sec=hg.SEC_NONE)
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
ct1, ct2 = rnd(2 * Q * n), rnd(2 * Q * n)
out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
key = rnd(Q * 2 * Qp * n)
ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, 1)
ws2 = ctx.workspace(hg.OP_CKKS_RESCALE, 0, 1)


def c2():
    ctx.ckks_multiply(ct1, 2 * Q * n, ct2, 2 * Q * n, out, 3 * Q * n, 0, 1)
    ctx.ckks_relinearize_inplace(out, 3 * Q * n, key, 0, 1, ws)
    ctx.ckks_rescale_inplace(out, 3 * Q * n, 0, 1, ws2)


dt = timeit(c2, args.reps)
print(f"C2 CKKS N=2^14 L=8 mul+relin+rescale, 1 ciphertext: {dt*1e6:.0f} us  ({1/dt:.0f} op/s)")
g = torch.cuda.CUDAGraph()
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c2()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            c2()
    dtg = timeit(g.replay, args.reps)
    print(f"   same sequence replayed as a hipGraph: {dtg*1e6:.0f} us  ({1/dtg:.0f} op/s)")
except Exception as e:  # noqa: BLE001
    print("   hipGraph capture failed:", e)
for b in (8, 64):
    c1b, c2b = rnd(2 * Q * n * b), rnd(2 * Q * n * b)
    ob = torch.empty(3 * Q * n * b, dtype=torch.int64, device="cuda")
    wsb = ctx.workspace(hg.OP_CKKS_RELIN, 0, b)
    wsb2 = ctx.workspace(hg.OP_CKKS_RESCALE, 0, b)

    def c2b_():
        ctx.ckks_multiply(c1b, 2 * Q * n, c2b, 2 * Q * n, ob, 3 * Q * n, 0, b)
        ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, b, wsb)
        ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, b, wsb2)
    dt = timeit(c2b_, args.reps)
    print(f"   batch {b}: {dt*1e3:.3f} ms  ({b/dt:.0f} op/s)")
ctx.close()

# ---- C3: BFV N=2^15 default chain, rotate, batch 64
n, t, B = 1 << 15, 786433, 64
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
ct = rnd(2 * Q * n * B)
out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
key = rnd(Q * 2 * Qp * n)
ws = ctx.workspace(hg.OP_BFV_GALOIS, 0, B)
gal = hg.steps_to_galois_elt(1, n, 2 * n)
dt = timeit(lambda: ctx.bfv_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, B, ws), max(3, args.reps // 4))
print(f"C3 BFV N=2^15 Q={Q} P=1 rotate_rows, {B} ciphertexts: {dt*1e3:.2f} ms  ({B/dt:.0f} rotations/s)")
ct2 = rnd(2 * Q * n * B)
o3 = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
wsm = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B)
wsr = ctx.workspace(hg.OP_BFV_RELIN, 0, B)


def bm():
    ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm)
    ctx.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, wsr)


dt = timeit(bm, max(3, args.reps // 4))
print(f"   BFV N=2^15 multiply+relinearize, {B} ciphertexts: {dt*1e3:.2f} ms  ({B/dt:.0f} op/s)")
dtm = timeit(lambda: ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm), max(3, args.reps // 4))
print(f"   (multiply alone: {dtm*1e3:.2f} ms)")
ctx.close()
del ct, ct2, out, o3, key, ws, wsm, wsr

# ---- north_star's second target: BFV N=2^14 (default 128-bit chain) homomorphic multiplications per second
n, t, B = 1 << 14, 786433, 256
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
ct, ct2 = rnd(2 * Q * n * B), rnd(2 * Q * n * B)
o3 = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
key = rnd(Q * 2 * Qp * n)
wsm = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B)
wsr = ctx.workspace(hg.OP_BFV_RELIN, 0, B)


def bm14():
    ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm)
    ctx.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, wsr)


dt = timeit(bm14, max(3, args.reps // 4))
print(f"BFV N=2^14 Q={Q} P=1 multiply+relinearize, {B} ciphertexts: {dt*1e3:.2f} ms  ({B/dt:.0f} op/s)")
dtm = timeit(lambda: ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm), max(3, args.reps // 4))
print(f"   (multiply alone: {dtm*1e3:.2f} ms = {B/dtm:.0f} mult/s)")
