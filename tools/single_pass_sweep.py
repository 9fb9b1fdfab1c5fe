#!/usr/bin/env python3
"""Plain NTT time against the number of limbs of the launch, LDS-resident single pass (HEGPU_SINGLE_PASS=1) against
the two passes (=0), N = 2^12..2^14: where NTT_SINGLE_PASS_MIN_LIMBS should sit."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import heongpu_amd as hg
    logn = int(sys.argv[2])
    n = 1 << logn
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50], sec=hg.SEC_NONE)
    ctx.upload()
    Qp = ctx.Q_prime_size
    res = []
    for mult in (1, 2, 4, 8, 16, 32, 64):
        limbs = Qp * mult
        x = torch.randint(0, 1 << 30, (limbs * n,), dtype=torch.int64, device="cuda")
        ts = []
        for inv in (0, 1):
            f = lambda: ctx.ntt(x, x, inv, limbs, Qp)
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res.append("%d: %.1f/%.1f" % (limbs, ts[0], ts[1]))
    print("N=2^%d  us forward/inverse by limbs  " % logn + "  ".join(res))
    sys.exit(0)
for logn in (12, 13, 14):
    for mode in ("1", "0"):
        env = dict(os.environ); env["HEGPU_SINGLE_PASS"] = mode
        out = subprocess.run([sys.executable, __file__, "child", str(logn)], env=env, capture_output=True, text=True)
        print("single=%s " % mode + (out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]))
