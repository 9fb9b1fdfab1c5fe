#!/usr/bin/env python3
"""C3 (BFV N=2^15 default chain, rotate_rows, 64 ciphertexts): the workload of bench.py's
secondary.c3_bfv_n15_rotate, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, t, B = 1 << 15, 786433, 64
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
ct = r(2 * Q * n * B)
out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
key = r(Q * 2 * Qp * n)
ws = ctx.workspace(hg.OP_BFV_GALOIS, 0, B)
gal = hg.steps_to_galois_elt(1, n, 3)
for _ in range(6):
    ctx.bfv_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, B, ws)
torch.cuda.synchronize()
