#!/usr/bin/env python3
"""C2 (CKKS N=2^14, {50,40x7}|{50}, multiply + relinearize + rescale, batch 1): the launch-bound case of
bench.py's secondary.c2_ckks_n14, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, B = 1 << 14, int(os.environ.get("C2_BATCH", "1"))
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50])
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
key = r(Q * 2 * Qp * n)
c1, c2 = r(2 * Q * n * B), r(2 * Q * n * B)
ob = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
ws, ws2 = ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B)
for _ in range(20):
    ctx.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, ob, 3 * Q * n, 0, B)
    ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
    ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, B, ws2)
torch.cuda.synchronize()
