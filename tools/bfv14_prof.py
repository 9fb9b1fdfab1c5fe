#!/usr/bin/env python3
"""BFV N=2^14 (default chain) multiply only, 256 pairs: the workload of bench.py's secondary.bfv_n14_multiply,
for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, t, B = 1 << 14, 786433, 256
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
ctx.upload()
Q = ctx.Q_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
ct, ct2 = r(2 * Q * n * B), r(2 * Q * n * B)
o3 = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
ws = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B)
for _ in range(6):
    ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, ws)
torch.cuda.synchronize()
