#!/usr/bin/env python3
"""Quick NTT throughput probe (forward/inverse GB/s = batch*2*N*8 B / time)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import heongpu_amd as hg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-power", type=int, default=16)
ap.add_argument("--polys", type=int, default=17 * 512)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--all-fp", action="store_true", help="a chain of 50-bit primes only (every limb on the FP64 butterflies)")
args = ap.parse_args()
n = 1 << args.n_power
log_q = [50] * 16 if args.all_fp else [60] + [50] * 15
ctx = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, [50 if args.all_fp else 60], sec=hg.SEC_NONE)
ctx.upload()
rc = 17
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randint(0, 1 << 49, (args.polys * n,), dtype=torch.int64, device="cuda", generator=g)
y = torch.empty_like(x)
for inverse in (False, True):
    ctx.ntt(x, y, inverse, args.polys, rc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ctx.ntt(x, y, inverse, args.polys, rc)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    gbps = args.polys * 2 * n * 8 / (ms * 1e-3) / 1e9
    print(f"N=2^{args.n_power} polys={args.polys} {'inv' if inverse else 'fwd'}: {ms:.3f} ms  "
          f"{gbps:.1f} GB/s  {args.polys / ms * 1e-3:.3f} M limb-NTT/s")
