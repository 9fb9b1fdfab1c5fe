#!/usr/bin/env python3
"""N = 2^14 forward / inverse transform, the forms side by side (option single_pass: 0 two passes, 1 one workgroup per
limb, 2 persistent + prefetch), on bench.py's ntt_by_degree shape (eight 50-bit primes + one 60-bit, ~1 GiB of limbs) and
on an all-integer chain (BFV's Bsk side: 61-bit primes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import heongpu_amd as hg  # noqa: E402

n = 1 << 14
for name, log_q, log_p in (("8x50+60", [50] * 8, [60]), ("9x60 (integer)", [60] * 8, [60]), ("9x55 (lazy integer)", [55] * 8, [55])):
    for form in (0, 1, 2):
        ctx = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, log_p, sec=hg.SEC_NONE)
        ctx.set_option("single_pass", form)
        ctx.upload()
        rc = ctx.Q_prime_size
        polys = ((1 << 27) // n) // rc * rc
        x = torch.randint(0, 1 << 49, (polys * n,), dtype=torch.int64, device="cuda")
        y = torch.empty_like(x)
        res = []
        for inverse in (False, True):
            ctx.ntt(x, y, inverse, polys, rc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.ntt(x, y, inverse, polys, rc)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            res.append("%s %.3f ms %.0f GB/s (%.3f of 8 TB/s)" % ("inv" if inverse else "fwd", ms, polys * 2 * n * 8 / ms / 1e6,
                                                                    polys * 2 * n * 8 / ms / 1e6 / 8000))
        print("%-22s single_pass=%d  %d limbs: %s" % (name, form, polys, "; ".join(res)), flush=True)
        ctx.close()
