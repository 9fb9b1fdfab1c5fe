import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import heongpu_amd as hg
from oracle import binding as ob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
log_q = [59] + [45] * (int(sys.argv[2]) if len(sys.argv) > 2 else 36)
c = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, [59], sec=hg.SEC_NONE)
primes = [int(x) for x in c.table("modulus")]
Q = len(log_q)
o = ob.OracleContext(ob.CKKS, c.n_power, primes, Q, 1)
c.upload()
g = np.random.default_rng(0)
m1, m2 = g.random(n // 2), g.random(n // 2)
scale = 2.0 ** 45
p1 = c.ckks_encode(torch.from_numpy(m1).cuda(), scale)
p1_o = o.ckks_encode(m1, scale)
print("encode == oracle", np.array_equal(hg.to_host(p1), p1_o))
d = c.ckks_decode(p1, scale, 0).cpu().numpy()
print("decode(encode) err", np.max(np.abs(d - m1)))
d_o = o.ckks_decode(p1_o, scale, 0)
print("oracle decode err", np.max(np.abs(d_o - m1)), "bit-equal", np.array_equal(d, d_o))
p2 = c.ckks_encode(torch.from_numpy(m2).cuda(), scale)
rg = hg.Rng(3)
sk = c.generate_secret_key(rg); pk = c.generate_public_key(rg, sk); rk = c.generate_relin_key(rg, sk)
ct1 = c.ckks_encrypt(rg, pk, p1); ct2 = c.ckks_encrypt(rg, pk, p2)
dd = c.ckks_decode(c.ckks_decrypt(ct1, sk, 0), scale, 0).cpu().numpy()
print("decrypt(encrypt) err", np.max(np.abs(dd - m1)))
out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
c.ckks_multiply(ct1, 2 * Q * n, ct2, 2 * Q * n, out, 3 * Q * n, 0, 1)
c.ckks_relinearize_inplace(out, 3 * Q * n, rk, 0, 1, c.workspace(hg.OP_CKKS_RELIN, 0, 1))
dd = c.ckks_decode(c.ckks_decrypt(out[:2 * Q * n].contiguous(), sk, 0), scale * scale, 0).cpu().numpy()
print("after mul+relin err", np.max(np.abs(dd - m1 * m2)))
c.ckks_rescale_inplace(out, 3 * Q * n, 0, 1, c.workspace(hg.OP_CKKS_RESCALE, 0, 1))
l = Q - 1
dec = c.ckks_decrypt(out[:2 * l * n].contiguous(), sk, 1)
dd = c.ckks_decode(dec, scale * scale / primes[Q - 1], 1).cpu().numpy()
print("after rescale err", np.max(np.abs(dd - m1 * m2)))
