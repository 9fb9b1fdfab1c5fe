#!/usr/bin/env python3
"""C4 step (multiply + relinearize, 64 pairs) with one context option toggled: usage c4_ab.py option v0 v1 [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import heongpu_amd as hg  # noqa: E402

opt, vals, reps = sys.argv[1], [int(v) for v in sys.argv[2:4]], int(sys.argv[4]) if len(sys.argv) > 4 else 20
work = bench.C4(torch, hg, torch.device("cuda", 0), 0, 64)
work.make_keys()
st = torch.cuda.current_stream().cuda_stream
ref = None
for rnd in range(2):
    for v in vals:
        work.ctx.set_option(opt, v)
        for _ in range(3):
            work.step(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            work.step(st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        cur = work.out.clone()
        same = True if ref is None else bool(torch.equal(cur.view(64, -1)[:, :work.ct_elems], ref.view(64, -1)[:, :work.ct_elems]))
        ref = cur if ref is None else ref
        print("%s=%d: %.3f ms per step, %.0f mul+relin/s, output equal to the first run's: %s" % (opt, v, ms, 64 / ms * 1e3, same), flush=True)
