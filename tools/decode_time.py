#!/usr/bin/env python3
"""CKKS decode time by degree on the chains of the reference's benchmark_ckks.cpp:17-24."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import heongpu_amd as hg
for logn, q, p, sc in ((12, [40, 30, 30], [40], 30), (13, [40] + [35] * 4, [40], 35), (14, [50] + [40] * 8, [50], 40),
                       (15, [60] + [45] * 16, [60], 45)):
    n = 1 << logn
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, q, p, sec=hg.SEC_NONE)
    ctx.upload()
    msg = torch.rand(n // 2, dtype=torch.float64, device="cuda")
    pl = ctx.ckks_encode(msg, 2.0 ** sc)
    for _ in range(3): ctx.ckks_decode(pl, 2.0 ** sc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): out = ctx.ckks_decode(pl, 2.0 ** sc)
    e1.record(); torch.cuda.synchronize()
    print("N=2^%d decode %.1f us  max err %.2e" % (logn, e0.elapsed_time(e1) / 20 * 1e3, float((out - msg).abs().max())))
