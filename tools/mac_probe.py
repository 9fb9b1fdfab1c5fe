#!/usr/bin/env python3
"""Time of the row pass + key inner product group of the C4 step (hegpu_probe_ckks_relinearize, phase 4),
for experiment builds of libhegpu (results are not checked)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, B = 1 << 16, 64
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * 15, [60])
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
out = r(3 * Q * n * B)
key = r(Q * 2 * Qp * n)
ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)
for ph in (2, 4):
    f = lambda: ctx.probe_ckks_relinearize(out, 3 * Q * n, key, 0, B, ws, ph)
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print("phase %d: %.3f ms" % (ph, e0.elapsed_time(e1) / 5))
