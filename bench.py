#!/usr/bin/env python3
"""bench.py -- homomorphic mults/sec (CKKS N=2^16, L=16) + NTT GB/s on MI355X.

One "step" = one pass of the hot path (multiply + relinearize_inplace, reference
benchmark/benchmark_ckks.cpp:123-137) over one batch of independent synthetic ciphertext pairs
that are already resident in HBM.  BASELINE.json config C4: 512 pairs sharded over 8 GPUs =
64 pairs per GPU; weak scaling: the global batch is 64 x n_gpus pairs, every rank (one process
per GPU) owns its `sharding.shard_range` slice, the relinearization key is produced on rank 0
and broadcast once with RCCL (torch.distributed "nccl").  There is no collective on the data path.

`python bench.py --gpus N` launches its N ranks itself (torch.distributed.run on 127.0.0.1) when
it is not already running under a launcher; under `torchrun --nproc-per-node N` it checks that the
world size equals --gpus.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (forward NTT
launch pair, HBM-bound), `in_step` (every launch group of the step timed on its own with HIP events
through the hegpu_probe_ckks_relinearize seam, with its algorithmic bytes), `secondary` (the other
BASELINE.json configurations, timed in the same run) and `cpu_baseline` (the CPU oracle timed on
this host, rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N = 65536
LOG_Q = [60] + [50] * 15
LOG_P = [60]
PAIRS_PER_GPU = 64      # config C4: 512 pairs over 8 GPUs
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=PAIRS_PER_GPU, help="ciphertext pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE.json configurations")
    ap.add_argument("--step-only", action="store_true",
                    help="warm-up + timed steps only (the PMC passes of tools/profile.sh: every dispatch belongs to a step)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs for the CPU baseline (0 = 2 x cores)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: exercise launch, sharding, key broadcast and the max-over-ranks reduction (CPU, gloo)")
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """--gpus N > 1 outside a launcher: start N ranks of this script, one per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def launcher_selftest(args):
    """What the multi-GPU path does around the kernels, without a GPU: rendezvous, shard the global batch,
    broadcast a key-shaped tensor from rank 0, reduce the elapsed time with MAX, gather per-rank rates."""
    import torch
    import torch.distributed as dist

    from heongpu_amd import sharding
    rank, world, _ = sharding.init_distributed(args.backend or "gloo")
    assert world == args.gpus, "world size %d != --gpus %d" % (world, args.gpus)
    start, count = sharding.shard_range(args.batch * world, world, rank)
    key = torch.arange(1 << 16, dtype=torch.int64) * 7 + 3 if rank == 0 else torch.zeros(1 << 16, dtype=torch.int64)
    sharding.broadcast_eval_key(key, src=0)
    ok = bool((key == torch.arange(1 << 16, dtype=torch.int64) * 7 + 3).all())
    elapsed = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    rates = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_gather(rates, torch.tensor([count / (0.001 * (rank + 1))], dtype=torch.float64))
    owned = torch.tensor([start, count], dtype=torch.int64)
    slices = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    if world > 1:
        dist.all_gather(slices, owned)
    else:
        slices = [owned]
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "key_broadcast_ok": ok,
                          "max_elapsed_s": float(elapsed.item()), "global_batch": args.batch * world,
                          "slices": [[int(v) for v in s] for s in slices]}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def synth_pairs(primes, Q, n, uniq):
    from helpers import synth_ct
    a = [synth_ct(primes, range(Q), 2, n, 1 + 10 * b) for b in range(uniq)]
    b = [synth_ct(primes, range(Q), 2, n, 2 + 10 * b) for b in range(uniq)]
    return a, b


class Timer:
    """HIP events on the launch stream (torch's current stream = the stream handed to the C ABI)."""

    def __init__(self, torch, reps=5):
        self.torch, self.reps = torch, reps
        self.e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        self.e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]

    def ms(self, fn, reps=None):
        reps = reps or self.reps
        fn()  # warm
        for i in range(reps):
            self.e0[i].record()
            fn()
            self.e1[i].record()
        self.torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in zip(self.e0[:reps], self.e1[:reps])) / reps


def rnd(torch, n_elems, bound=1 << 30):
    return torch.randint(0, bound, (n_elems,), dtype=torch.int64, device="cuda")


def ntt_sweep(torch, hg, timer):
    """NTT GB/s (2W per limb NTT) for N = 2^12 .. 2^16, forward and inverse, on a chain
    of eight 50-bit primes + one 60-bit special prime, ~1 GiB of polynomials per launch."""
    out = {}
    for n_power in range(12, 17):
        n = 1 << n_power
        ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] * 8, [60], sec=hg.SEC_NONE)
        ctx.upload()
        rc = ctx.Q_prime_size
        polys = max(rc, ((1 << 27) // n) // rc * rc)
        x = rnd(torch, polys * n, 1 << 49)
        y = torch.empty_like(x)
        stream = torch.cuda.current_stream().cuda_stream
        f = timer.ms(lambda: ctx.ntt(x, y, False, polys, rc, stream=stream), 3)
        i = timer.ms(lambda: ctx.ntt(x, y, True, polys, rc, stream=stream), 3)
        b = polys * 2 * n * 8
        out["2^%d" % n_power] = {"limbs": polys, "forward_GBps": b / (f * 1e-3) / 1e9, "inverse_GBps": b / (i * 1e-3) / 1e9,
                                 "forward_frac": b / (f * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        ctx.close()
        del x, y
    return out


def hoisted_rotation_block(torch, hg, timer, ctx, B=16):
    """k rotations of each of B ciphertexts at the C4 chain, the decomposition and the digit
    NTT shared (hegpu_ckks_rotate_hoisted) against k separate hegpu_ckks_apply_galois calls."""
    n, Q, Qp = ctx.n, ctx.Q_size, ctx.Q_prime_size
    stream = torch.cuda.current_stream().cuda_stream
    words = 2 * Q * n
    ct = rnd(torch, B * words, 1 << 49)
    ws = ctx.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, B)  # room for four accumulators (>= OP_CKKS_GALOIS)
    kmax = 8
    keys = [rnd(torch, Q * 2 * Qp * n, 1 << 49) for _ in range(kmax)]
    elts = [hg.steps_to_galois_elt(i + 1, n, 5) for i in range(kmax)]
    out = torch.empty(B * kmax * words, dtype=torch.int64, device="cuda")
    res = {"workload": "CKKS N=2^16, Q=16 | P=1 (method I), depth 0, %d ciphertexts, k Galois elements each" % B, "by_k": {}}
    for k in (1, 2, 4, 8):
        h = timer.ms(lambda: ctx.ckks_rotate_hoisted(ct, words, out, k * words, keys[:k], elts[:k], 0, B, ws, stream=stream), 3)

        def separate():
            for i in range(k):
                ctx.ckks_apply_galois(ct, words, out[i * B * words:], words, keys[i], elts[i], 0, B, ws, stream=stream)
        s_ = timer.ms(separate, 3)
        res["by_k"][str(k)] = {"hoisted_rotations_per_s": B * k / (h * 1e-3), "separate_rotations_per_s": B * k / (s_ * 1e-3),
                               "hoisted_ms": h, "separate_ms": s_}
    return res


def secondary_block(torch, hg, timer):
    """The other BASELINE.json configurations, synthetic data, each with the algorithmic bytes of the
    reference's kernel sequence and the fraction of the 8 TB/s HBM peak that rate means."""
    sec = {}
    stream = torch.cuda.current_stream().cuda_stream

    # ---- north_star target 2: BFV N=2^14, default 128-bit chain (Q=8, P=1), 256 pairs
    n, t, B = 1 << 14, 786433, 256
    ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
    ctx.upload()
    Q, Qp, L = ctx.Q_size, ctx.Q_prime_size, len(ctx.table("q_Bsk_merge_modulus"))
    W = 8 * n
    ct, ct2 = rnd(torch, 2 * Q * n * B), rnd(torch, 2 * Q * n * B)
    o3 = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    key = rnd(torch, Q * 2 * Qp * n)
    wsm, wsr = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B), ctx.workspace(hg.OP_BFV_RELIN, 0, B)
    m = timer.ms(lambda: ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm, stream=stream), 3)
    r = timer.ms(lambda: ctx.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, wsr, stream=stream), 3)
    mul_bytes = (28 * L + 7 * Q) * W
    sec["bfv_n14_multiply"] = {
        "workload": "BFV N=2^14 default 128-bit chain (Q=%d, P=1, Bsk=%d), t=786433, %d pairs resident in HBM" % (Q, L - Q, B),
        "multiplications_per_s": B / (m * 1e-3), "ms_per_batch": m,
        "multiply_relinearize_per_s": B / ((m + r) * 1e-3),
        "reference_sequence_bytes_per_op": mul_bytes, "frac_of_hbm_peak": mul_bytes * B / (m * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    ctx.close()
    del ct, ct2, o3, key, wsm, wsr

    # ---- C3: BFV N=2^15 default chain (Q=14, P=1), rotate_rows by one step, 64 ciphertexts
    n, t, B = 1 << 15, 786433, 64
    ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    W = 8 * n
    ct = rnd(torch, 2 * Q * n * B)
    out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
    key = rnd(torch, Q * 2 * Qp * n)
    ws = ctx.workspace(hg.OP_BFV_GALOIS, 0, B)
    gal = hg.steps_to_galois_elt(1, n, 3)
    g = timer.ms(lambda: ctx.bfv_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, B, ws, stream=stream), 3)
    rot_bytes = (6 * Q * Qp + 6 * Q + 8 * Qp) * W
    sec["c3_bfv_n15_rotate"] = {
        "workload": "BFV N=2^15 default chain (Q=%d, P=1), rotate_rows (Galois key switch method I), %d ciphertexts" % (Q, B),
        "rotations_per_s": B / (g * 1e-3), "ms_per_batch": g,
        "reference_sequence_bytes_per_op": rot_bytes, "frac_of_hbm_peak": rot_bytes * B / (g * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    ctx.close()
    del ct, out, key, ws

    # ---- C2: CKKS N=2^14, {50, 40 x 7} | {50}, multiply + relinearize + rescale
    n = 1 << 14
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50])
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    W = 8 * n
    key = rnd(torch, Q * 2 * Qp * n)
    c2 = {}
    for B in (1, 64):
        c1b, c2b = rnd(torch, 2 * Q * n * B), rnd(torch, 2 * Q * n * B)
        ob = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
        wsb, wsb2 = ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B)

        def seq():
            ctx.ckks_multiply(c1b, 2 * Q * n, c2b, 2 * Q * n, ob, 3 * Q * n, 0, B, stream=stream)
            ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, wsb, stream=stream)
            ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, B, wsb2, stream=stream)
        c2[B] = timer.ms(seq, 5)
        if B == 1:
            # the same sequence captured once and replayed as one hipGraph launch (the operator entries
            # allocate nothing and never synchronise): the launch-bound batch-1 case
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gs = torch.cuda.current_stream().cuda_stream
                ctx.ckks_multiply(c1b, 2 * Q * n, c2b, 2 * Q * n, ob, 3 * Q * n, 0, B, stream=gs)
                ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, wsb, stream=gs)
                ctx.ckks_rescale_inplace(ob, 3 * Q * n, 0, B, wsb2, stream=gs)
            c2["graph"] = timer.ms(graph.replay, 5)
    op_bytes = (6 * Q * Q + 32 * Q + 8 + 6 + 16 * (Q - 1)) * W
    sec["c2_ckks_n14"] = {
        "workload": "CKKS N=2^14, Q=8 {50,40x7} | P=1 {50}, multiply + relinearize + rescale",
        "latency_us_batch1": c2[1] * 1e3, "latency_us_batch1_hipgraph_replay": c2["graph"] * 1e3,
        "ops_per_s_batch64": 64 / (c2[64] * 1e-3),
        "reference_sequence_bytes_per_op": op_bytes,
        "frac_of_hbm_peak_batch64": op_bytes * 64 / (c2[64] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    ctx.close()

    # ---- key-switching method II (selected by the reference whenever P_size > 1): the C4 shape with four special
    # primes, Q = 16 x 50 bits | P = 4 x 50 bits (d = 4 digits of 4 primes), multiply + relinearize, 64 pairs
    n, B = 1 << 16, 64
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] * 16, [50] * 4, sec=hg.SEC_NONE)
    ctx.upload()
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    d = -(-Q // 4)
    c1b, c2b = rnd(torch, 2 * Q * n * B), rnd(torch, 2 * Q * n * B)
    ob = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    key = rnd(torch, d * 2 * Qp * n)
    wsb = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)

    def seq2():
        ctx.ckks_multiply(c1b, 2 * Q * n, c2b, 2 * Q * n, ob, 3 * Q * n, 0, B, stream=stream)
        ctx.ckks_relinearize_inplace(ob, 3 * Q * n, key, 0, B, wsb, stream=stream)
    m2 = timer.ms(seq2, 3)
    sec["ckks_n16_method_II"] = {
        "workload": "CKKS N=2^16, Q=16 x 50 bits | P=4 x 50 bits (hybrid key switching, 4 digits), multiply + relinearize, "
                    "%d pairs" % B,
        "multiply_relinearize_per_s": B / (m2 * 1e-3), "ms_per_batch": m2}
    ctx.close()
    del c1b, c2b, ob, key, wsb

    # ---- C5: TFHE STD128 NAND gate bootstrap, 8192 concurrent gates (the whole config on one GPU;
    # its 8-GPU share is 1024)
    t = hg.TfheContext()
    rng = np.random.default_rng(1)
    S = 8192
    polys = t.int("bootkey_elems") // 1024
    v = rng.integers(-2**31, 2**31, polys, dtype=np.int64)  # constant polynomials: NTT image = the constant
    lifted = np.where(v < 0, v + t.prime, v).astype(np.uint64)
    bk = torch.from_numpy(np.repeat(lifted, 1024).view(np.int64)).cuda()
    i32 = lambda k: torch.randint(-2**31, 2**31, (k,), dtype=torch.int64, device="cuda").to(torch.int32)
    ks_a, ks_b = i32(t.int("kskey_a_elems")), i32(t.int("kskey_b_elems"))
    a1, a2, b1, b2 = i32(S * 512), i32(S * 512), i32(S), i32(S)
    prepared = t.prepare_bootkey(bk)
    out_a = torch.empty(S * 512, dtype=torch.int32, device="cuda")
    out_b = torch.empty(S, dtype=torch.int32, device="cuda")
    ws = torch.empty((512 + 1024 + 2) * S, dtype=torch.int32, device="cuda")
    g = timer.ms(lambda: t.gate(hg.GATE_NAND, a1, b1, a2, b2, out_a, out_b, prepared, ks_a, ks_b, S, ws, stream=stream), 2)
    sec["c5_tfhe_gates"] = {
        "workload": "TFHE STD128 NAND gate bootstrap (pre-computation, blind rotate n=512, sample extraction, key switch), "
                    "%d concurrent gates, torus32 boot key (FP64 blind rotate)" % S,
        "gates_per_s": S / (g * 1e-3), "ms_per_batch": g,
        "reference_sequence_bytes_per_gate": 72 * (1 << 20),
        "frac_of_hbm_peak": 72 * (1 << 20) * S / (g * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    t.close()
    return sec


def power_sample(torch, step):
    """Package power and shader clock while the step loops (outside the timed region): the kernels of the step are
    bound by FP64 issue at the clock the package power limit allows, which this records next to the
    throughput.  rocm-smi is polled from the host while ~3 s of steps sit in the stream; None if it is not there."""
    import re
    import subprocess
    smi = "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    try:
        for _ in range(350):
            step()
        watts, mhz, limit = [], [], None
        for _ in range(3):
            txt = subprocess.run([smi, "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True,
                                 timeout=20).stdout
            m = re.search(r"Current Socket Graphics Package Power \(W\): ([0-9.]+)", txt)
            if m:
                watts.append(float(m.group(1)))
            m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            if m:
                mhz.append(int(m.group(1)))
            m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", txt)
            if m:
                limit = float(m.group(1))
        torch.cuda.synchronize()
        if not watts:
            return None
        return {"package_w": max(watts), "package_limit_w": limit, "sclk_mhz": (min(mhz) if mhz else None),
                "how": "rocm-smi polled three times while 350 extra steps run (not in the timed region); "
                       "highest power / lowest clock of the polls that fell inside the run"}
    except Exception as e:  # a missing or slow tool must not cost the line
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return {"error": str(e)[:120]}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.launcher_selftest:
        raise SystemExit(launcher_selftest(args))

    import torch
    import torch.distributed as dist

    import heongpu_amd as hg
    from heongpu_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # one process per GPU; on a box with fewer devices than ranks (a functional check of the multi-rank path
    # with --backend gloo on one GPU) ranks share devices
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    rank, world, local_rank = sharding.init_distributed(args.backend or "nccl", device_index=dev_index)
    if world != args.gpus:
        raise SystemExit("bench.py: world size %d (WORLD_SIZE) does not match --gpus %d" % (world, args.gpus))
    dev = torch.device("cuda", dev_index)

    ctx = hg.Context.from_bit_sizes(hg.CKKS, N, LOG_Q, LOG_P)
    ctx.upload()
    primes = [int(v) for v in ctx.table("modulus")]
    Q, Qp, n = ctx.Q_size, ctx.Q_prime_size, N
    l, rc = Q, Qp
    # global batch = args.batch pairs per GPU (C4: 512 over 8); this rank's contiguous slice of it
    first, B = sharding.shard_range(args.batch * world, world, rank)
    W = 8 * n  # bytes of one limb polynomial

    # ---- evaluation key: rank 0 generates, RCCL broadcast over xGMI
    key = torch.empty(2 * Q * Qp * n, dtype=torch.int64, device=dev)
    key_host = None
    if rank == 0:
        from helpers import synth_key
        key_host = synth_key(primes, Q, Qp, n, 3)
        key.copy_(torch.from_numpy(key_host.view(np.int64)))
    bcast_ms = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        sharding.broadcast_eval_key(key, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3

    # ---- inputs: a few distinct seeded pairs, repeated to fill the batch
    uniq = min(B, 4)
    a_h, b_h = synth_pairs(primes, Q, n, uniq)
    ct_elems = 2 * l * n
    ct1 = torch.empty(B * ct_elems, dtype=torch.int64, device=dev)
    ct2 = torch.empty(B * ct_elems, dtype=torch.int64, device=dev)
    for b in range(B):
        ct1[b * ct_elems:(b + 1) * ct_elems].copy_(torch.from_numpy(a_h[(first + b) % uniq].view(np.int64)))
        ct2[b * ct_elems:(b + 1) * ct_elems].copy_(torch.from_numpy(b_h[(first + b) % uniq].view(np.int64)))
    out_elems = 3 * l * n
    out = torch.empty(B * out_elems, dtype=torch.int64, device=dev)
    ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.ckks_multiply(ct1, ct_elems, ct2, ct_elems, out, out_elems, 0, B, stream=stream)
        ctx.ckks_relinearize_inplace(out, out_elems, key, 0, B, ws, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = [B * args.steps / own_elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rates = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(rates, torch.tensor(per_rank, dtype=torch.float64, device=dev))
        per_rank = [float(r.item()) for r in rates]
    result0 = hg.to_host(out[:out_elems]) if rank == 0 else None
    total_ops = args.batch * world * args.steps
    value = total_ops / elapsed

    line = {
        "metric": "homomorphic mults/sec (CKKS N=2^16, L=16) + NTT GB/s vs HBM roofline",
        "value": value,
        "unit": "multiply+relinearize/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "CKKS N=2^16, Q=16 {60,50x15} | P=1 {60}, depth 0: multiply + relinearize_inplace "
                        "(key-switch method I), %d independent ciphertext pairs per GPU per step (global batch %d "
                        "sharded by contiguous slices), inputs resident in HBM" % (args.batch, args.batch * world),
            "poly_modulus_degree": N,
            "Q_size": Q,
            "P_size": 1,
            "batch_per_gpu": args.batch,
            "global_batch": args.batch * world,
            "parallelism": "batch-sharded x%d, one process per GPU, RCCL key broadcast, no data-path collective" % world,
            "algorithmic_bytes_per_op": (6 * l * l + 32 * l + 8) * W,
        },
        "per_rank_ops_per_s": per_rank,
    }
    if bcast_ms is not None:
        line["key_broadcast_ms"] = bcast_ms
        line["key_bytes"] = key.numel() * 8

    if args.step_only:
        if rank == 0:
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the transform: the forward NTT launch pair at the key-switch
    # shape (l*rc limb NTTs per ciphertext).  Algorithmic bytes = 2W per limb NTT; HIP events on the
    # launch stream.
    timer = Timer(torch)
    polys = B * l * rc
    ntt_ms = {inv: timer.ms(lambda: ctx.ntt(ws, ws, inv, polys, rc, stream=stream)) for inv in (False, True)}
    ntt_bytes = polys * 2 * W
    fwd_gbps = ntt_bytes / (ntt_ms[False] * 1e-3) / 1e9
    inv_gbps = ntt_bytes / (ntt_ms[True] * 1e-3) / 1e9

    # ---- every launch group of a step on its own (hegpu_probe_ckks_relinearize), against the bytes the
    # group has to move as it is built (fused kernels: fewer than the reference's sequence):
    #   W = one limb; per ciphertext at depth 0 (l = 16, Q' = 17)
    def probe(ph):
        return timer.ms(lambda: ctx.probe_ckks_relinearize(out, out_elems, key, 0, B, ws, ph, stream=stream))
    groups = [
        ("ckks_multiply: k_cross_multiplication", timer.ms(
            lambda: ctx.ckks_multiply(ct1, ct_elems, ct2, ct_elems, out, out_elems, 0, B, stream=stream)),
         7 * l * W * B, "read 4l, write 3l limbs"),
        ("INTT of c2: ntt_inv_row + ntt_inv_col", probe(1), 2 * l * W * B,
         "l limb INTTs, 2W each (the column stages of the FP64 limbs run inside the next group's kernel)"),
        ("decomposing column pass: ntt_fwd_col_multi + ntt_fwd_col<8,true>", probe(2),
         (l + l * rc - l) * W * B, "read l source limbs once, write l*Q' - l half-transformed digits"),
        ("row pass + key inner product: ks_row_mac_fp + ks_row_mac", probe(4),
         ((l * rc - l) + l + 2 * rc) * W * B + 2 * l * rc * W,
         "read l*Q' - l digits + l identity limbs, write 2Q' limbs per ciphertext; the key (2 l Q' limbs) once per batch"),
        ("INTT of the two P limbs", probe(8), 2 * 2 * W * B, "2 limb INTTs"),
        ("mod-down NTT with stage one / two fused: ntt_fwd_col_multi + ntt_fwd_row", probe(16),
         (2 + 2 * l + 2 * l + 3 * 2 * l + 2 * l) * W * B,
         "column pass: read 2 P limbs, write 2l; row pass: read 2l + the 2l accumulated limbs + 2l of ct, write 2l"),
    ]
    in_step = [{"launches": name, "ms": ms, "algorithmic_bytes": by, "accounting": note,
                "achieved_GBps": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
               for name, ms, by, note in groups]
    dominant = max(in_step, key=lambda g: g["ms"])

    line["roofline"] = {
        "bound": "hbm",
        "kernel": "forward NTT of the key-switch digits (ntt_fwd_col<8,false> + ntt_fwd_row, one launch pair)",
        "achieved": fwd_gbps,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": fwd_gbps / HBM_PEAK_GBPS,
        "traffic": None,
        "launch_ms": ntt_ms[False],
        "limb_ntts_per_launch": polys,
        "algorithmic_bytes_per_launch": ntt_bytes,
        "in_step_dominant": {"launches": dominant["launches"], "ms": dominant["ms"], "frac": dominant["frac"],
                             "achieved": dominant["achieved_GBps"], "algorithmic_bytes": dominant["algorithmic_bytes"]},
    }
    line["ntt"] = {"forward_GBps": fwd_gbps, "inverse_GBps": inv_gbps, "n": N, "limbs": polys}
    line["in_step"] = in_step
    # the whole step against (i) the reference sequence's bytes (what the unfused kernels would
    # move) and (ii) the bytes this implementation actually moves (PMC, profiles/traffic.json)
    line["hbm_fraction_reference_equivalent"] = ((6 * l * l + 32 * l + 8) * W * value / world) / 1e9 / HBM_PEAK_GBPS
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("limb_ntts_per_launch") == polys:
            line["roofline"]["traffic"] = tj["bytes_per_launch"]
            line["roofline"]["traffic_source"] = tj.get("source", "profiles/traffic.json")
        if tj.get("step_bytes") and tj.get("step_batch") == B:
            line["roofline"]["step_traffic"] = tj["step_bytes"]
            line["hbm_fraction_moved"] = tj["step_bytes"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS
        if tj.get("copy_ceiling_GBps"):
            line["roofline"]["copy_ceiling_GBps"] = tj["copy_ceiling_GBps"]

    if rank == 0 and world == 1 and not args.no_secondary:
        line["power"] = power_sample(torch, step)
        line["ntt_by_degree"] = ntt_sweep(torch, hg, timer)
        line["secondary"] = secondary_block(torch, hg, timer)
        del ct1, ct2, out, ws
        line["secondary"]["hoisted_rotations"] = hoisted_rotation_block(torch, hg, timer, ctx)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        o = ob.OracleContext(ob.CKKS, 16, primes, Q, 1)
        cores = ob.lib().o_omp_threads()
        sample = args.cpu_sample or max(2 * cores, 2)
        sample = min(sample, B)
        c1 = np.concatenate([a_h[b % uniq] for b in range(sample)])
        c2 = np.concatenate([b_h[b % uniq] for b in range(sample)])
        o3 = np.zeros(sample * out_elems, dtype=np.uint64)
        t0 = time.perf_counter()
        ob.lib().o_ckks_mul_relin_batch(o.h, c1.ctypes.data, c2.ctypes.data, o3.ctypes.data,
                                        key_host.ctypes.data, 0, sample)
        cpu_s = time.perf_counter() - t0
        same = bool(np.array_equal(o3[:2 * l * n], result0[:2 * l * n]))
        line["cpu_baseline"] = {
            "value": sample / cpu_s,
            "unit": "multiply+relinearize/s",
            "cores": cores,
            "kind": "port",
            "sample": "%d ciphertext pairs of the same workload (CKKS N=2^16 L=16 mul+relin), CPU oracle, "
                      "OpenMP over pairs, %.1f s" % (sample, cpu_s),
            "gpu_matches_cpu_bit_exact": same,
        }
    print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
