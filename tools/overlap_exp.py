#!/usr/bin/env python3
"""Does running two halves of the C4 batch on two streams overlap the HBM-bound and the VALU-bound kernels of
multiply + relinearize?  Times the 64-pair step on one stream against 2 x 32 and 4 x 16 on separate streams."""
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
c1, c2 = r(2 * Q * n * B), r(2 * Q * n * B)
out = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
key = r(Q * 2 * Qp * n)
cs, os_ = 2 * Q * n, 3 * Q * n


def run(parts, skew=False):
    per = B // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    wss = [ctx.workspace(hg.OP_CKKS_RELIN, 0, per) for _ in range(parts)]
    def step():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
        for i, s in enumerate(streams):
            a, b, o = c1[i * per * cs:], c2[i * per * cs:], out[i * per * os_:]
            ctx.ckks_multiply(a, cs, b, cs, o, os_, 0, per, stream=s.cuda_stream)
            ctx.ckks_relinearize_inplace(o, os_, key, 0, per, wss[i], stream=s.cuda_stream)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(3): step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10

for parts in (1, 2, 4, 8):
    print("%d stream(s) x %2d ciphertexts: %.3f ms per 64-pair step" % (parts, B // parts, run(parts)))
