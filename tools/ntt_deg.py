#!/usr/bin/env python3
"""bench.py's ntt_by_degree shape (eight 50-bit + one 60-bit prime, 1 GiB of limbs) for N = 2^12 .. 2^14, forward / inverse,
plus the BFV N = 2^14 multiply of the secondary block: one line each (A/B runs: tools/ntt_ab.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import heongpu_amd as hg  # noqa: E402

timer = bench.Timer(torch)
r = bench.ntt_sweep(torch, hg, timer)
print(" ".join("%s fwd %.3f inv %.3f" % (k, v["forward_frac"], v["inverse_GBps"] / 8000) for k, v in r.items() if k in ("2^12", "2^13", "2^14")))
s = bench.sec_bfv14(torch, hg, None)
run, units = s.runs["bfv_n14_multiply"]
ms = timer.ms(run, 5)
print("BFV N=2^14 multiply: %.3f ms per 256 pairs, %.0f /s" % (ms, units / ms * 1e3))
