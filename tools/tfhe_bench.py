#!/usr/bin/env python3
"""TFHE gate-bootstrap throughput probe (config C5 per-GPU share: 1024 gates)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import heongpu_amd as hg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gates", type=int, default=1024)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
t = hg.TfheContext()
rng = np.random.default_rng(1)
S = args.gates
if os.environ.get("TFHE_BENCH_KEY", "torus32") == "torus32":
    # a real key has torus32 coefficients.  The NTT image of a constant polynomial v is v in
    # every slot, so per-polynomial constants give a valid key without needing a transform here.
    polys = t.int("bootkey_elems") // 1024
    v = rng.integers(-2**31, 2**31, polys, dtype=np.int64)
    lifted = np.where(v < 0, v + t.prime, v).astype(np.uint64)
    bk = torch.from_numpy(np.repeat(lifted, 1024).view(np.int64)).cuda()
else:
    bk = torch.from_numpy(rng.integers(0, t.prime, t.int("bootkey_elems"), dtype=np.uint64).view(np.int64)).cuda()
ks_a = torch.randint(-2**31, 2**31, (t.int("kskey_a_elems"),), dtype=torch.int64, device="cuda").to(torch.int32)
ks_b = torch.randint(-2**31, 2**31, (t.int("kskey_b_elems"),), dtype=torch.int64, device="cuda").to(torch.int32)
a1 = torch.randint(-2**31, 2**31, (S * 512,), dtype=torch.int64, device="cuda").to(torch.int32)
a2 = torch.randint(-2**31, 2**31, (S * 512,), dtype=torch.int64, device="cuda").to(torch.int32)
b1 = torch.randint(-2**31, 2**31, (S,), dtype=torch.int64, device="cuda").to(torch.int32)
b2 = torch.randint(-2**31, 2**31, (S,), dtype=torch.int64, device="cuda").to(torch.int32)
prepared = t.prepare_bootkey(bk)
print('prepared key layout:', 'FP64' if t.prepared_is_fp64(prepared) else 'integer')
out_a = torch.empty(S * 512, dtype=torch.int32, device="cuda")
out_b = torch.empty(S, dtype=torch.int32, device="cuda")
ws = torch.empty((512 + 1024 + 2) * S, dtype=torch.int32, device="cuda")
t.gate(hg.GATE_NAND, a1, b1, a2, b2, out_a, out_b, prepared, ks_a, ks_b, S, ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    t.gate(hg.GATE_NAND, a1, b1, a2, b2, out_a, out_b, prepared, ks_a, ks_b, S, ws)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
print(f"NAND gate bootstrap: {S} gates in {dt*1e3:.2f} ms  -> {S/dt:.0f} gates/s")
# split: blind rotate vs key switching
ea = torch.empty(S * 1024, dtype=torch.int32, device="cuda")
eb = torch.empty(S, dtype=torch.int32, device="cuda")
for name, fn in (("blind_rotate", lambda: t.bootstrapping(a1, b1, prepared, ea, eb, S)),
                 ("key_switching", lambda: t.key_switching(ea, eb, out_a, out_b, ks_a, ks_b, S))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        fn()
    torch.cuda.synchronize()
    print(f"  {name}: {(time.perf_counter()-t0)/args.reps*1e3:.2f} ms")
