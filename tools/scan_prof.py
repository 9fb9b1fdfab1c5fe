#!/usr/bin/env python3
"""A mixed operator sequence on one parameter set, for rocprofv3 --kernel-trace --stats: looking for kernels whose time
is out of proportion.  usage: scan_prof.py ckks|bfv logn batch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
scheme, logn, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = 1 << logn
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
if scheme == "ckks":
    nq = {12: 2, 13: 4, 14: 8, 15: 14, 16: 16}[logn]
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [40] * (nq - 1), [60], sec=hg.SEC_NONE)
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    key, gkey = r(Q * 2 * Qp * n), r(Q * 2 * Qp * n)
    c1, c2 = r(2 * Q * n * B), r(2 * Q * n * B)
    ob = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    rot = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
    ws, ws2, ws3 = ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B), ctx.workspace(hg.OP_CKKS_GALOIS, 0, B)
    g = hg.steps_to_galois_elt(1, n, 5)
    for _ in range(5):
        ctx.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, ob, 3 * Q * n, 0, B)
        ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
        ctx.ckks_apply_galois(ob, 3 * Q * n, rot, 2 * Q * n, gkey, g, 0, B, ws3)
        ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, B, ws2)
else:
    t = 786433
    ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    key, gkey = r(Q * 2 * Qp * n), r(Q * 2 * Qp * n)
    c1, c2 = r(2 * Q * n * B), r(2 * Q * n * B)
    o3 = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
    wm, wr, wg = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B), ctx.workspace(hg.OP_BFV_RELIN, 0, B), ctx.workspace(hg.OP_BFV_GALOIS, 0, B)
    g = hg.steps_to_galois_elt(1, n, 3)
    for _ in range(5):
        ctx.bfv_multiply(c1, 2 * Q * n, c2, 2 * Q * n, o3, 3 * Q * n, B, wm)
        ctx.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, wr)
        ctx.bfv_apply_galois(o3, 3 * Q * n, out, 2 * Q * n, gkey, g, B, wg)
torch.cuda.synchronize()
