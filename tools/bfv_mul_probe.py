"""BFV N=2^14 multiply (default chain), 256 ciphertext pairs: a few launches for profilers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heongpu_amd as hg
n, t, B = 1 << 14, 786433, 256
ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t); ctx.upload()
Q = ctx.Q_size
rnd = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
ct, ct2 = rnd(2 * Q * n * B), rnd(2 * Q * n * B)
o3 = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
wsm = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B)
for _ in range(4):
    ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm)
torch.cuda.synchronize()
