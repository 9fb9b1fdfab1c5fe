#!/usr/bin/env python3
"""Relinearize (method I) time per ciphertext against the batch size, fused row pass + inner product
(HEGPU_FUSED_ROW_MAC=1) against the reference's sequence (=0): where the automatic choice should switch."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import heongpu_amd as hg
    logn, nq = int(sys.argv[2]), abs(int(sys.argv[3]))
    n = 1 << logn
    if int(sys.argv[3]) < 0:  # a chain of 59-bit primes: integer butterflies throughout (the BFV default chains)
        ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [59] * nq, [59], sec=hg.SEC_NONE)
    else:
        ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [60] + [50] * (nq - 1), [60], sec=hg.SEC_NONE)
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    r = lambda k: torch.randint(0, 1 << 30, (k,), dtype=torch.int64, device="cuda")
    key = r(Q * 2 * Qp * n)
    res = []
    for B in (1, 2, 4, 8, 16):
        ob = r(3 * Q * n * B)
        ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)
        f = lambda: ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, ws)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append("%d: %.1f" % (B, e0.elapsed_time(e1) / 20 * 1e3))
    print("N=2^%d Q=%d  us per launch by batch  " % (logn, Q) + "  ".join(res))
    sys.exit(0)
for logn, nq in [(int(a), int(b)) for a, b in (x.split(':') for x in os.environ.get('SWEEP', '14:8,15:15,16:16,16:30').split(','))]:
    for mode in os.environ.get("MODES", "1,0,split2,split4,auto").split(","):
        env = dict(os.environ)
        if mode in ("1", "0"): env["HEGPU_FUSED_ROW_MAC"] = mode
        if mode.startswith("split"): env["HEGPU_DIGIT_SPLIT"] = mode[5:]
        if mode.startswith("multi"): env["HEGPU_COL_MULTI"] = mode[5:]
        out = subprocess.run([sys.executable, __file__, "child", str(logn), str(nq)], env=env, capture_output=True, text=True)
        print("fused=%s " % mode + out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
