#!/usr/bin/env python3
"""CKKS rescale_inplace at the C4 chain (N=2^16, 16 limbs -> 15), 64 ciphertexts, for rocprofv3 --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
n, B = 1 << 16, 64
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * 15, [60])
ctx.upload()
Q = ctx.Q_size
ct = torch.randint(0, 1 << 30, (3 * Q * n * B,), dtype=torch.int64, device="cuda")
ws = ctx.workspace(hg.OP_CKKS_RESCALE, 0, B)
for _ in range(5):
    ctx.ckks_rescale_inplace(ct, 3 * Q * n, 0, B, ws)
torch.cuda.synchronize()
