#!/usr/bin/env python3
"""C2 latency (CKKS N=2^14, {50,40x7}|{50}, multiply + relinearize + rescale) at small batches, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n = 1 << 14
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50])
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
key = r(Q * 2 * Qp * n)
for B in (1, 2, 4, 8, 16, 64):
    c1, c2 = r(2 * Q * n * B), r(2 * Q * n * B)
    ob = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    ws, ws2 = ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B)
    def seq():
        ctx.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, ob, 3 * Q * n, 0, B)
        ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
        ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, B, ws2)
    for _ in range(5): seq()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): seq()
    e1.record(); torch.cuda.synchronize()
    print("batch %3d: %8.1f us per sequence, %8.1f us per ciphertext" % (B, e0.elapsed_time(e1) * 1e3 / 20, e0.elapsed_time(e1) * 1e3 / 20 / B))
