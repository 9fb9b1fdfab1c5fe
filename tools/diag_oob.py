"""Canary check: which operator writes outside its ciphertext buffer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import heongpu_amd as hg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 21
c = hg.Context.from_bit_sizes(hg.CKKS, n, [59] + [45] * (Q - 1), [59], sec=hg.SEC_NONE)
c.upload()
rg = hg.Rng(3)
sk = c.generate_secret_key(rg); pk = c.generate_public_key(rg, sk); rk = c.generate_relin_key(rg, sk)
p1 = c.ckks_encode(torch.rand(n // 2, dtype=torch.float64, device="cuda"), 2.0 ** 45)
ct1 = c.ckks_encrypt(rg, pk, p1); ct2 = c.ckks_encrypt(rg, pk, p1)
G = 1 << 20
CAN = -6148914691236517206  # 0xAAAA...
def fresh(words):
    t = torch.full((G + words + G,), CAN, dtype=torch.int64, device="cuda")
    return t, t[G:G + words]
def check(t, words, what):
    torch.cuda.synchronize()
    lo = (t[:G] != CAN).nonzero(); hi = (t[G + words:] != CAN).nonzero()
    print(what, "before:", lo.numel(), "after:", hi.numel(), ("first after offset %d last %d" % (int(hi[0]), int(hi[-1]))) if hi.numel() else "")
buf, out = fresh(3 * Q * n)
c.ckks_multiply(ct1, 2 * Q * n, ct2, 2 * Q * n, out, 3 * Q * n, 0, 1)
check(buf, 3 * Q * n, "multiply")
wsb, ws = fresh(c.workspace(hg.OP_CKKS_RELIN, 0, 1).numel())
c.ckks_relinearize_inplace(out, 3 * Q * n, rk, 0, 1, ws)
check(buf, 3 * Q * n, "relinearize ct"); check(wsb, ws.numel(), "relinearize ws")
wsb, ws = fresh(c.workspace(hg.OP_CKKS_RESCALE, 0, 1).numel())
c.ckks_rescale_inplace(out, 3 * Q * n, 0, 1, ws)
check(buf, 3 * Q * n, "rescale ct"); check(wsb, ws.numel(), "rescale ws")
l = Q - 1
dbuf, dec = fresh(l * n)
c._lib.hegpu_ckks_decrypt(c._h, out.data_ptr(), sk.data_ptr(), 1, dec.data_ptr(), torch.cuda.current_stream().cuda_stream)
check(dbuf, l * n, "decrypt")
wsb, ws = fresh(c.workspace(hg.OP_CKKS_DECODE, 1, 1).numel())
ob, o = fresh(n // 2)
c._lib.hegpu_ckks_decode(c._h, dec.data_ptr(), 1, 2.0 ** 45, o.data_ptr(), ws.data_ptr(), ws.numel() * 8, torch.cuda.current_stream().cuda_stream)
check(wsb, ws.numel(), "decode ws"); check(ob, n // 2, "decode out")
gal = hg.steps_to_galois_elt(1, n, 5)
gk = c.generate_galois_key(rg, sk, gal)
rb, rot = fresh(2 * Q * n)
wsb, ws = fresh(c.workspace(hg.OP_CKKS_GALOIS, 0, 1).numel())
c.ckks_apply_galois(ct1, 2 * Q * n, rot, 2 * Q * n, gk, gal, 0, 1, ws)
check(rb, 2 * Q * n, "galois out"); check(wsb, ws.numel(), "galois ws")
