#!/usr/bin/env python3
"""relinearize_inplace of B ciphertexts on a {60,50x(nq-1)}|{60} chain, 20 times: for rocprofv3 --kernel-trace --stats.
usage: relin_prof.py logn nq B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
logn, nq, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 1 << logn
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * (nq - 1), [60], sec=hg.SEC_NONE)
ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
key = r(Q * 2 * Qp * n)
ob = r(3 * Q * n * B)
ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)
for _ in range(20):
    ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
torch.cuda.synchronize()
