#!/usr/bin/env python3
"""One-ciphertext calls (the reference's own benchmark regime) under combinations of the launch-form options:
C2 (CKKS N=2^14 {50,40x7}|{50}) multiply + relinearize + rescale at batch 1, and relinearize alone at N = 2^15
(Q = 16 + 1, benchmark_ckks.cpp's chain) and N = 2^16 (C4's chain).  us per call, HIP events over 20 calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")


def timed(f, reps=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


combos = [{}, {"single_pass": 1}, {"single_pass": 0}, {"fused_row_mac": 1}, {"fused_row_mac": 1, "single_pass": 1},
          {"fused_row_mac": 1, "digit_split": 2}, {"fused_row_mac": 1, "digit_split": 4}]
if len(sys.argv) > 1:
    combos = [eval(a) for a in sys.argv[1:]]
for opts in combos:
    res = []
    with hg.default_options(**opts):
        ctx = hg.Context.from_bit_sizes(hg.CKKS, 1 << 14, [50] + [40] * 7, [50])
    ctx.upload()
    n, Q, Qp = 1 << 14, ctx.Q_size, ctx.Q_prime_size
    key = r(Q * 2 * Qp * n); c1, c2 = r(2 * Q * n), r(2 * Q * n)
    ob = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
    ws, ws2 = ctx.workspace(hg.OP_CKKS_RELIN, 0, 1), ctx.workspace(hg.OP_CKKS_RESCALE, 0, 1)

    def seq():
        ctx.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, ob, 3 * Q * n, 0, 1)
        ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, 1, ws)
        ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, 1, ws2)
    res.append("C2 mul+relin+rescale %.1f" % timed(seq))
    res.append("relin %.1f" % timed(lambda: ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, 1, ws)))
    res.append("rescale %.1f" % timed(lambda: ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, 1, ws2)))
    ctx.close()
    for logn, lq, lp in ((15, [60] + [50] * 15, [60]), (16, [60] + [50] * 15, [60])):
        n = 1 << logn
        with hg.default_options(**opts):
            ctx = hg.Context.from_bit_sizes(hg.CKKS, n, lq, lp, sec=hg.SEC_NONE)
        ctx.upload()
        Q, Qp = ctx.Q_size, ctx.Q_prime_size
        key = r(Q * 2 * Qp * n); ob = r(3 * Q * n)
        ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, 1)
        res.append("N=2^%d relin %.1f" % (logn, timed(lambda: ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, 1, ws))))
        ctx.close()
    print("%-60s %s" % (opts, " | ".join(res)), flush=True)
