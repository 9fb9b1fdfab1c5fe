#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 4(a): do the HBM-bound launch groups of one half-batch (tensor product, INTT of c2, mod-down:
2.0 ms at 0.8-1.0 of the copy ceiling, <= 0.5 of issue) run beside the VALU-bound groups of the other half (column pass,
row pass + inner product: 5.7 ms at 0.75-0.78 of issue, <= 0.7 of copy)?

Two FREE-RUNNING streams (no synchronisation between them until the end -- tools/overlap_exp.py of round 2 joined the
streams after every step, which keeps the halves in lockstep on the same kind of kernel), each K steps of 32 pairs, the
second started half a step late; plain streams and streams created with hipExtStreamCreateWithCUMask for several splits
of the 256 CUs.  Reference: K steps of 64 pairs on one stream.  Prints ms per 64 pairs and the package power / clock."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import heongpu_amd as hg  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")  # torch's copy is already in the process


def masked_stream(mask_words):
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(mask_words)), arr)
    assert rc == 0, "hipExtStreamCreateWithCUMask: %d" % rc
    return st.value


def plain_stream():
    st = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(st), ctypes.c_uint(1)) == 0  # non-blocking
    return st.value


def sync(st):
    assert hip.hipStreamSynchronize(ctypes.c_void_p(st)) == 0


def power():
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        import json
        card = next(iter(json.loads(out).values()))
        return {k: v for k, v in card.items() if "Power" in k or "sclk" in k}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:80]}


full = bench.C4(torch, hg, dev, 0, 64)
full.make_keys()
st0 = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    full.step(st0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    full.step(st0)
torch.cuda.synchronize()
base = (time.perf_counter() - t0) / K * 1e3
print("one stream, 64 pairs per step: %.3f ms per 64 pairs   %s" % (base, power()), flush=True)
ref = full.out.view(64, -1)[:, :full.ct_elems].clone()

halves = [bench.C4(torch, hg, dev, 32 * i, 32, ctx=full.ctx) for i in range(2)]
for h in halves:
    h.key = full.key


def run(streams, offset_steps=0.5, label=""):
    for h, s in zip(halves, streams):
        for _ in range(2):
            h.step(s)
    for s in streams:
        sync(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # interleave the submissions so that neither stream's queue runs dry; stream 1 gets a head start of half a step
    halves[0].step(streams[0])
    if offset_steps:
        time.sleep(offset_steps * base / 2 * 1e-3)
    for k in range(K):
        halves[1].step(streams[1])
        if k + 1 < K:
            halves[0].step(streams[0])
    for s in streams:
        sync(s)
    ms = (time.perf_counter() - t0) / K * 1e3
    got = torch.cat([h.out.view(32, -1)[:, :h.ct_elems] for h in halves])
    print("%-58s %.3f ms per 64 pairs (%.1f %% of one stream)  equal: %s  %s"
          % (label, ms, 100 * ms / base, bool(torch.equal(got, ref)), power()), flush=True)


run([plain_stream(), plain_stream()], 0.5, "two plain streams, 32 + 32, half a step apart")
run([plain_stream(), plain_stream()], 0.0, "two plain streams, 32 + 32, started together")
ALL = 0xFFFFFFFF
for name, m0, m1 in (
        ("CU masks: words 0-3 | words 4-7 (128 | 128)", [ALL] * 4 + [0] * 4, [0] * 4 + [ALL] * 4),
        ("CU masks: even bits | odd bits (128 | 128)", [0x55555555] * 8, [0xAAAAAAAA] * 8),
        ("CU masks: 192 | 64 (words 0-5 | 6-7)", [ALL] * 6 + [0] * 2, [0] * 6 + [ALL] * 2),
        ("CU masks: both full (256 | 256)", [ALL] * 8, [ALL] * 8)):
    try:
        run([masked_stream(m0), masked_stream(m1)], 0.5, name)
    except AssertionError as e:
        print(name, "failed:", e)
