#!/usr/bin/env python3
"""Determinism soak: the same operator call repeated on the same inputs must give the same bits every time
(a race between workgroups, streams or the workspace would show as a difference).  C4 multiply + relinearize,
BFV N=2^15 rotate, BFV N=2^14 multiply + relinearize, CKKS rescale + rotate at N=2^14, TFHE NAND."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import heongpu_amd as hg
REPS = int(os.environ.get("SOAK_REPS", "60"))
r = lambda k, hi=1 << 30: torch.randint(0, hi, (k,), dtype=torch.int64, device="cuda")
bad = 0

def soak(name, fn, out):
    global bad
    fn(); torch.cuda.synchronize()
    ref = out.clone()
    diff = 0
    for _ in range(REPS):
        out.fill_(-1)
        fn()
        diff += int((out != ref).any())
    torch.cuda.synchronize()
    print("%-40s %d / %d repetitions differ" % (name, diff, REPS))
    bad += diff

n, B = 1 << 16, 32
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * 15, [60]); ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
c1, c2, key = r(2 * Q * n * B), r(2 * Q * n * B), r(Q * 2 * Qp * n)
out = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)
def f():
    ctx.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, out, 3 * Q * n, 0, B)
    ctx.ckks_relinearize_inplace(out, 3 * Q * n, key, 0, B, ws)
soak("C4 multiply + relinearize, 32 pairs", f, out)
for b1 in (1, 3):
    o1 = torch.empty(3 * Q * n * b1, dtype=torch.int64, device="cuda")
    w1 = ctx.workspace(hg.OP_CKKS_RELIN, 0, b1)
    def f1():
        ctx.ckks_multiply(c1, 2 * Q * n, c2, 2 * Q * n, o1, 3 * Q * n, 0, b1)
        ctx.ckks_relinearize_inplace(o1, 3 * Q * n, key, 0, b1, w1)
    soak("C4 multiply + relinearize, %d pair(s)" % b1, f1, o1)
del ctx, c1, c2, key, out, ws

n, B = 1 << 15, 32
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=786433); ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
ct, key = r(2 * Q * n * B), r(Q * 2 * Qp * n)
out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
ws = ctx.workspace(hg.OP_BFV_GALOIS, 0, B)
gal = hg.steps_to_galois_elt(1, n, 3)
soak("BFV N=2^15 rotate, 32 ciphertexts", lambda: ctx.bfv_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, B, ws), out)
del ctx

n, B = 1 << 14, 64
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=786433); ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
c1, c2, key = r(2 * Q * n * B), r(2 * Q * n * B), r(Q * 2 * Qp * n)
out = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
ws, ws2 = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B), ctx.workspace(hg.OP_BFV_RELIN, 0, B)
def g():
    ctx.bfv_multiply(c1, 2 * Q * n, c2, 2 * Q * n, out, 3 * Q * n, B, ws)
    ctx.bfv_relinearize_inplace(out, 3 * Q * n, key, B, ws2)
soak("BFV N=2^14 multiply + relinearize, 64", g, out)
del ctx

n, B = 1 << 14, 8
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50]); ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
c1, key = r(3 * Q * n * B), r(Q * 2 * Qp * n)
work = torch.empty_like(c1)
rot = torch.empty(2 * (Q - 1) * n * B, dtype=torch.int64, device="cuda")
ws, ws2, ws3 = ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B), ctx.workspace(hg.OP_CKKS_GALOIS, 1, B)
gal = hg.steps_to_galois_elt(1, n, 5)
def h():
    work.copy_(c1)
    ctx.ckks_relinearize_inplace(work, 3 * Q * n, key, 0, B, ws)
    ctx.ckks_rescale_inplace(work, 3 * Q * n, 0, B, ws2)
    ctx.ckks_apply_galois(work, 3 * Q * n, rot, 2 * (Q - 1) * n, key, gal, 1, B, ws3)
soak("CKKS N=2^14 relin + rescale + rotate, 8", h, rot)
del ctx

# two streams on ONE context at the same time, each with its own data and workspace: results must equal the serial ones
n, B = 1 << 14, 4
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50]); ctx.upload()
Q, Qp = ctx.Q_size, ctx.Q_prime_size
key = r(Q * 2 * Qp * n)
gal = hg.steps_to_galois_elt(1, n, 5)
srcs = [r(3 * Q * n * B) for _ in range(2)]
works = [torch.empty_like(srcs[0]) for _ in range(2)]
rots = [torch.empty(2 * (Q - 1) * n * B, dtype=torch.int64, device="cuda") for _ in range(2)]
wss = [(ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B), ctx.workspace(hg.OP_CKKS_GALOIS, 1, B)) for _ in range(2)]
def seq(i, stream=None):
    works[i].copy_(srcs[i])
    ctx.ckks_relinearize_inplace(works[i], 3 * Q * n, key, 0, B, wss[i][0], stream=stream)
    ctx.ckks_rescale_inplace(works[i], 3 * Q * n, 0, B, wss[i][1], stream=stream)
    ctx.ckks_apply_galois(works[i], 3 * Q * n, rots[i], 2 * (Q - 1) * n, key, gal, 1, B, wss[i][2], stream=stream)
for i in range(2):
    seq(i)
torch.cuda.synchronize()
refs = [x.clone() for x in rots]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
diff = 0
for _ in range(REPS):
    for i in range(2):
        rots[i].fill_(-1)
    torch.cuda.synchronize()
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            seq(i, stream=streams[i].cuda_stream)
    torch.cuda.synchronize()
    diff += int(any(bool((rots[i] != refs[i]).any()) for i in range(2)))
print("%-40s %d / %d repetitions differ" % ("two streams on one context (CKKS N=2^14)", diff, REPS))
bad += diff
del ctx

t = hg.TfheContext()
S = 300
bk = r(t.int("bootkey_elems"), 1 << 31)
prepared = t.prepare_bootkey(bk)
i32 = lambda k: torch.randint(-2**31, 2**31, (k,), dtype=torch.int64, device="cuda").to(torch.int32)
ks_a, ks_b = i32(t.int("kskey_a_elems")), i32(t.int("kskey_b_elems"))
a1, a2, b1, b2 = i32(S * 512), i32(S * 512), i32(S), i32(S)
oa, ob = torch.empty(S * 512, dtype=torch.int32, device="cuda"), torch.empty(S, dtype=torch.int32, device="cuda")
wst = torch.empty((512 + 1024 + 2) * S, dtype=torch.int32, device="cuda")
soak("TFHE NAND, 300 gates", lambda: t.gate(hg.GATE_NAND, a1, b1, a2, b2, oa, ob, prepared, ks_a, ks_b, S, wst), oa)
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
