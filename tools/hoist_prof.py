#!/usr/bin/env python3
"""Hoisted rotations, C4 chain, 16 ciphertexts x 8 Galois elements, for rocprofv3 --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, B, K = 1 << 16, 16, 8
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * 15, [60])
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
ct = r(2 * Q * n * B)
keys = [r(Q * 2 * Qp * n) for _ in range(K)]
elts = [hg.steps_to_galois_elt(s + 1, n, 5) for s in range(K)]
words = 2 * Q * n
out = torch.empty(B * K * words, dtype=torch.int64, device="cuda")
ws = ctx.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, B)
for _ in range(4):
    ctx.ckks_rotate_hoisted(ct, words, out, K * words, keys, elts, 0, B, ws)
torch.cuda.synchronize()
