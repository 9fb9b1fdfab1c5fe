#!/usr/bin/env python3
"""CKKS N=2^16 {60,50x15}|{60} rotate (apply_galois, method I), 16 ciphertexts: rotations/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, B = 1 << 16, int(os.environ.get("ROT_BATCH", "16"))
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * 15, [60])
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
ct = r(2 * Q * n * B)
out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
key = r(Q * 2 * Qp * n)
ws = ctx.workspace(hg.OP_CKKS_GALOIS, 0, B)
gal = hg.steps_to_galois_elt(1, n, 5)
f = lambda: ctx.ckks_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, 0, B, ws)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("batch %d: %.3f ms, %.0f rotations/s" % (B, ms, B / ms * 1e3))
