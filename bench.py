#!/usr/bin/env python3
"""bench.py -- homomorphic mults/sec (CKKS N=2^16, L=16) + NTT GB/s on MI355X.

One "step" = one pass of the hot path (multiply + relinearize_inplace,
reference benchmark/benchmark_ckks.cpp:123-137) over one batch of independent
synthetic ciphertext pairs that are already resident in HBM.  Weak scaling:
every rank (one process per GPU) owns `--batch` pairs; the relinearization key
is produced on rank 0 and broadcast with RCCL (torch.distributed "nccl").

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (forward NTT, HBM-bound) and `cpu_baseline` (the CPU oracle timed
on this host, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N = 65536
LOG_Q = [60] + [50] * 15
LOG_P = [60]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def synth_pairs(primes, Q, n, uniq):
    from helpers import synth_ct
    a = [synth_ct(primes, range(Q), 2, n, 1 + 10 * b) for b in range(uniq)]
    b = [synth_ct(primes, range(Q), 2, n, 2 + 10 * b) for b in range(uniq)]
    return a, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="ciphertext pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs for the CPU baseline (0 = 2 x cores)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import heongpu_amd as hg
    from heongpu_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    rank, world, local_rank = sharding.init_distributed("nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    ctx = hg.Context.from_bit_sizes(hg.CKKS, N, LOG_Q, LOG_P)
    ctx.upload()
    primes = [int(v) for v in ctx.table("modulus")]
    Q, Qp, n = ctx.Q_size, ctx.Q_prime_size, N
    l, rc = Q, Qp
    B = args.batch
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
        ct1[b * ct_elems:(b + 1) * ct_elems].copy_(torch.from_numpy(a_h[b % uniq].view(np.int64)))
        ct2[b * ct_elems:(b + 1) * ct_elems].copy_(torch.from_numpy(b_h[b % uniq].view(np.int64)))
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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    result0 = hg.to_host(out[:out_elems]) if rank == 0 else None

    # ---- roofline of the dominant kernel: the key-switch forward NTT
    # (l*rc limb NTTs per ciphertext).  Algorithmic bytes = 2W per limb NTT
    # (SURVEY.md 8d); HIP events on the launch stream.
    polys = B * l * rc
    reps = 5
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    ntt_ms = {}
    for inverse in (False, True):
        ctx.ntt(ws, ws, inverse, polys, rc, stream=stream)  # warm
        for i in range(reps):
            e0[i].record()
            ctx.ntt(ws, ws, inverse, polys, rc, stream=stream)
            e1[i].record()
        torch.cuda.synchronize()
        ntt_ms[inverse] = sum(a.elapsed_time(b) for a, b in zip(e0, e1)) / reps
    ntt_bytes = polys * 2 * W
    fwd_gbps = ntt_bytes / (ntt_ms[False] * 1e-3) / 1e9
    inv_gbps = ntt_bytes / (ntt_ms[True] * 1e-3) / 1e9

    # ---- the two operators of a step on their own (HIP events on the launch stream), against
    # their algorithmic bytes: multiply reads 4l and writes 3l limbs per pair, the
    # rest of (6 l^2 + 32 l + 8) W belongs to relinearize_inplace
    op_ms = {}
    for name, fn in (("ckks_multiply", lambda: ctx.ckks_multiply(ct1, ct_elems, ct2, ct_elems, out, out_elems, 0, B,
                                                                 stream=stream)),
                     ("ckks_relinearize_inplace", lambda: ctx.ckks_relinearize_inplace(out, out_elems, key, 0, B, ws,
                                                                                       stream=stream))):
        fn()
        for i in range(reps):
            e0[i].record()
            fn()
            e1[i].record()
        torch.cuda.synchronize()
        op_ms[name] = sum(a.elapsed_time(b) for a, b in zip(e0, e1)) / reps
    op_bytes = {"ckks_multiply": 7 * l * W * B, "ckks_relinearize_inplace": (6 * l * l + 25 * l + 8) * W * B}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_ops = B * args.steps * world
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
                        "(key-switch method I), %d independent ciphertext pairs per GPU per step, inputs "
                        "resident in HBM" % B,
            "poly_modulus_degree": N,
            "Q_size": Q,
            "P_size": 1,
            "batch_per_gpu": B,
            "parallelism": "batch-sharded x%d, RCCL key broadcast" % world,
            "algorithmic_bytes_per_op": (6 * l * l + 32 * l + 8) * W,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "forward NTT of the key-switch digits (ntt_fwd_col<8> + ntt_fwd_row, one launch pair)",
            "achieved": fwd_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": fwd_gbps / HBM_PEAK_GBPS,
            "traffic": None,
            "launch_ms": ntt_ms[False],
            "limb_ntts_per_launch": polys,
            "algorithmic_bytes_per_launch": ntt_bytes,
        },
        "ntt": {"forward_GBps": fwd_gbps, "inverse_GBps": inv_gbps, "n": N, "limbs": polys},
        # per operator: SURVEY 8d bytes = what the reference's kernel sequence for that operator reads and
        # writes; the fused kernels here move fewer bytes, so this is "reference-equivalent" bandwidth
        "operators": {k: {"ms_per_launch": op_ms[k], "reference_sequence_bytes_per_launch": op_bytes[k],
                          "reference_equivalent_GBps": op_bytes[k] / (op_ms[k] * 1e-3) / 1e9,
                          "frac_of_hbm_peak": op_bytes[k] / (op_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS} for k in op_ms},
        "hbm_fraction_end_to_end": ((6 * l * l + 32 * l + 8) * W * value / world) / 1e9 / HBM_PEAK_GBPS,
    }
    # HBM traffic of the same launch from the PMC counters (separate rocprofv3
    # --pmc passes, tools/profile.sh): committed next to the kernel stats.
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("limb_ntts_per_launch") == polys:
            line["roofline"]["traffic"] = tj["bytes_per_launch"]
            line["roofline"]["traffic_source"] = tj.get("source", "profiles/traffic.json")
    if bcast_ms is not None:
        line["key_broadcast_ms"] = bcast_ms

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
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
