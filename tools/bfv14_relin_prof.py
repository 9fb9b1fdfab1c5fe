#!/usr/bin/env python3
"""BFV N=2^14 (default chain) relinearize only, 256 ciphertexts, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, t, B = 1 << 14, 786433, 256
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
o3 = r(3 * Q * n * B)
key = r(Q * 2 * Qp * n)
ws = ctx.workspace(hg.OP_BFV_RELIN, 0, B)
for _ in range(6):
    ctx.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, ws)
torch.cuda.synchronize()
