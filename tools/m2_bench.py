#!/usr/bin/env python3
"""CKKS N=2^16 multiply + relinearize with key-switching method II (P_size > 1): Q = 16 x 50-bit, P = 4 x 50-bit
(d = 4 digits of 4 primes), 64 ciphertext pairs; next to method I on {60,50x15}|{60} for comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, B = 1 << 16, 64
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
for name, lq, lp in (("method I  {60,50x15}|{60}", [60] + [50] * 15, [60]), ("method II {50x16}|{50x4}", [50] * 16, [50] * 4),
                     ("method II {60,50x15}|{60x4}", [60] + [50] * 15, [60] * 4)):
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, lq, lp, sec=hg.SEC_NONE)
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    d = Q if len(lp) == 1 else -(-Q // len(lp))
    c1, c2 = r(2 * Q * n * B), r(2 * Q * n * B)
    out = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    key = r(d * 2 * Qp * n)
    ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)
    def step():
        ctx.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, out, 3 * Q * n, 0, B)
        ctx.ckks_relinearize_inplace(out, 3 * Q * n, key, 0, B, ws)
    for _ in range(3): step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-30s %7.3f ms per %d pairs  %7.0f mul+relin/s" % (name, ms, B, B / ms * 1e3))
    ctx.close()
