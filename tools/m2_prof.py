#!/usr/bin/env python3
"""Method II relinearize (CKKS N=2^16, Q = 16 x 50-bit, P = 4 x 50-bit), 64 pairs, for rocprofv3 --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, B = 1 << 16, 64
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] * 16, [50] * 4, sec=hg.SEC_NONE)
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
out = r(3 * Q * n * B)
key = r(4 * 2 * Qp * n)
ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)
for _ in range(5):
    ctx.ckks_relinearize_inplace(out, 3 * Q * n, key, 0, B, ws)
torch.cuda.synchronize()
