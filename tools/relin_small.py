#!/usr/bin/env python3
"""One- and two-ciphertext relinearize (method I) under combinations of the launch-form options.
usage: relin_small.py logn nq"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
logn, nq = int(sys.argv[1]), int(sys.argv[2])
n = 1 << logn
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
for opts in ({}, {"fused_row_mac": 0}, {"fused_row_mac": 1}, {"fused_row_mac": 1, "digit_split": 4},
             {"fused_row_mac": 1, "digit_split": 4, "col_multi": 0}, {"fused_row_mac": 1, "digit_split": 8, "col_multi": 0},
             {"fused_row_mac": 1, "digit_split": 2, "col_multi": 0}, {"fused_row_mac": 1, "col_multi": 0}):
    with hg.default_options(**opts):
        ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * (nq - 1), [60], sec=hg.SEC_NONE)
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    key = r(Q * 2 * Qp * n)
    res = []
    for B in (1, 2, 4):
        ob = r(3 * Q * n * B)
        ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)
        f = lambda: ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append("%d: %.1f" % (B, e0.elapsed_time(e1) / 20 * 1e3))
    print("N=2^%d Q=%d %-70s us: %s" % (logn, Q, opts, "  ".join(res)), flush=True)
