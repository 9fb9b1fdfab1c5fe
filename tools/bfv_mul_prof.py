#!/usr/bin/env python3
"""BFV multiply (default chain) of B ciphertext pairs, 10 times: for rocprofv3 --kernel-trace --stats.
usage: bfv_mul_prof.py logn B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
logn, B = int(sys.argv[1]), int(sys.argv[2])
n = 1 << logn
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=786433)
ctx.upload()
Q = ctx.Q_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
c1, c2 = r(2 * Q * n * B), r(2 * Q * n * B)
out = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
ws = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B)
for _ in range(10):
    ctx.bfv_multiply(c1, 2 * Q * n, c2, 2 * Q * n, out, 3 * Q * n, B, ws)
torch.cuda.synchronize()
