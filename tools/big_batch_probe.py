import sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import heongpu_amd as hg
n = 4096
c = hg.Context.from_bit_sizes(hg.CKKS, n, [40, 30], [40], sec=hg.SEC_NONE); c.upload()
Q, Qp = c.Q_size, c.Q_prime_size
B, D = 70000, 10
r = lambda k: torch.randint(0, 1 << 29, (k,), dtype=torch.int64, device="cuda")
def rep(x, per): return x.reshape(D, per).repeat((B + D - 1) // D, 1)[:B].contiguous().reshape(-1)
def twins(name, out, per):
    o = out.reshape(B, per)
    ok = bool((o == o[:D].repeat((B + D - 1) // D, 1)[:B]).all())
    print("%-28s twins equal: %s   nonzero words in the last item: %d of %d" % (name, ok, int((o[B - 1] != 0).sum()), per))
a, b = rep(r(D * 2 * Q * n), 2 * Q * n), rep(r(D * 2 * Q * n), 2 * Q * n)
key = r(Q * 2 * Qp * n)
for name, fn in (
    ("addition", lambda out: c.addition(a, b, out, Q, 2, B, 0)),
):
    out = torch.zeros_like(a)
    try:
        fn(out); torch.cuda.synchronize(); twins(name, out, 2 * Q * n)
    except Exception as e:
        print("%-28s raised: %s" % (name, str(e)[:100]))
out3 = torch.zeros(3 * Q * n * B, dtype=torch.int64, device="cuda")
try:
    c.ckks_multiply(a, 2 * Q * n, b, 2 * Q * n, out3, 3 * Q * n, 0, B); torch.cuda.synchronize(); twins("ckks_multiply", out3, 3 * Q * n)
except Exception as e:
    print("ckks_multiply raised:", str(e)[:100])
try:
    ws = c.workspace(hg.OP_CKKS_RELIN, 0, B)
    c.ckks_relinearize_inplace(out3, 3 * Q * n, key, 0, B, ws); torch.cuda.synchronize(); twins("ckks_relinearize", out3, 3 * Q * n)
    del ws
except Exception as e:
    print("ckks_relinearize raised:", str(e)[:100])
try:
    ws = c.workspace(hg.OP_CKKS_RESCALE, 0, B)
    c.ckks_rescale_inplace(out3, 3 * Q * n, 0, B, ws); torch.cuda.synchronize(); twins("ckks_rescale", out3, 3 * Q * n)
    del ws
except Exception as e:
    print("ckks_rescale raised:", str(e)[:100])
try:
    rot = torch.zeros(2 * Q * n * B, dtype=torch.int64, device="cuda")
    ws = c.workspace(hg.OP_CKKS_GALOIS, 0, B)
    c.ckks_apply_galois(a, 2 * Q * n, rot, 2 * Q * n, key, hg.steps_to_galois_elt(1, n, 5), 0, B, ws); torch.cuda.synchronize(); twins("ckks_apply_galois", rot, 2 * Q * n)
    del ws, rot
except Exception as e:
    print("ckks_apply_galois raised:", str(e)[:100])
del c, out3
c = hg.Context.from_default(hg.BFV, n, 1, 1032193); c.upload()
Q, Qp = c.Q_size, c.Q_prime_size
a, b = rep(r(D * 2 * Q * n), 2 * Q * n), rep(r(D * 2 * Q * n), 2 * Q * n)
out3 = torch.zeros(3 * Q * n * B, dtype=torch.int64, device="cuda")
try:
    ws = c.workspace(hg.OP_BFV_MULTIPLY, 0, B)
    c.bfv_multiply(a, 2 * Q * n, b, 2 * Q * n, out3, 3 * Q * n, B, ws); torch.cuda.synchronize(); twins("bfv_multiply", out3, 3 * Q * n)
except Exception as e:
    print("bfv_multiply raised:", str(e)[:100])
