#!/usr/bin/env python3
"""bench.py -- homomorphic mults/sec (CKKS N=2^16, L=16) + NTT GB/s on MI355X.

One "step" = one pass of the hot path (multiply + relinearize_inplace, reference
benchmark/benchmark_ckks.cpp:123-137) over one batch of independent synthetic ciphertext pairs
that are already resident in HBM.  BASELINE.json config C4: 512 pairs sharded over 8 GPUs =
64 pairs per GPU; weak scaling: the global batch is 64 x n_gpus pairs, every rank owns its
`sharding.shard_range` slice, the relinearization key is produced on rank 0 and replicated once.
There is no collective on the data path.

Multi-GPU (`--gpus N`), in this order of preference -- the line says which one ran (`config.parallelism`):
  1. one process per GPU (torch.distributed.run, launched by bench.py itself when it is not already under a
     launcher): control plane on gloo, the key broadcast on RCCL (backend "nccl") over xGMI;
  2. the same, with the key staged through the host over gloo when RCCL cannot be brought up within 60 s;
  3. one process, one thread and one context per device (hegpu_context_upload_device), the key replicated with
     hegpu_broadcast_key (peer copies) -- when the launch of the ranks itself fails, or with --single-process.
Ranks beyond the number of visible devices share devices (a functional check on a 1-GPU box).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (forward NTT launch pair),
`in_step` (every launch group of the step timed on its own with HIP events, with its algorithmic bytes, its
vector-ALU issue figures and the ceiling that binds it), `checked_items` (every output of the timed batch
verified: against the CPU oracle and against its twin), `secondary` (the other BASELINE.json configurations,
timed and verified in the same run) and `cpu_baseline` (the CPU oracle timed on this host, rank 0, N=1 only).
The oracle is imported only by the verification / cpu_baseline legs: inputs come from heongpu_amd.synth.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 65536
LOG_Q = [60] + [50] * 15
LOG_P = [60]
PAIRS_PER_GPU = 64      # config C4: 512 pairs over 8 GPUs
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
UNIQ = 4                # distinct seeded pairs, repeated to fill a batch
SIMDS, LANES = 1024, 16  # 256 CUs x 4 SIMDs, 16 lanes each: a wave64 vector instruction holds its SIMD for 4 cycles
NCCL_TIMEOUT_S = 60


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
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs for the CPU baseline (0 = 2 x cores, at most the batch)")
    ap.add_argument("--backend", default=None, help="backend of the key broadcast (default nccl = RCCL; gloo stages through the host)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in one process: a thread and a context per device, hegpu_broadcast_key")
    ap.add_argument("--force-launch-failure", action="store_true",
                    help="testing: pretend the launch of the ranks failed (exercises the single-process fallback)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: exercise launch, sharding, key replication and the max-over-ranks reduction on CPU")
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """--gpus N > 1 outside a launcher: start N ranks of this script, one per GPU.  Returns the launcher's exit code."""
    if args.force_launch_failure:
        return 97
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------ distributed plumbing
class Dist:
    """Control plane on gloo (CPU tensors: barriers, timing reductions -- nothing that can hang on a GPU fabric), the
    one data movement -- the key broadcast -- on RCCL, brought up under a watchdog; host staging over gloo otherwise."""

    def __init__(self, torch, world, rank, want_backend):
        self.torch, self.world, self.rank = torch, world, rank
        self.key_path = "none (one rank)"
        self.hung = False
        if world == 1:
            return
        import datetime
        import torch.distributed as dist
        self.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        self.want = want_backend or "nccl"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_float(self, v):
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, v):
        if self.world == 1:
            return [v]
        out = [self.torch.zeros(1, dtype=self.torch.float64) for _ in range(self.world)]
        self.dist.all_gather(out, self.torch.tensor([v], dtype=self.torch.float64))
        return [float(t.item()) for t in out]

    def _nccl_broadcast(self, key, dev, result):
        try:
            import datetime
            g = self.dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=NCCL_TIMEOUT_S), device_id=dev)
            probe = self.torch.ones(1, device=dev)
            self.dist.all_reduce(probe, group=g)
            self.torch.cuda.synchronize(dev)
            if int(probe.item()) != self.world:
                raise RuntimeError("RCCL all-reduce probe returned %r" % probe.item())
            flat = key.view(-1)
            for off in range(0, flat.numel(), 1 << 27):  # <= 1 GiB per message
                self.dist.broadcast(flat[off:off + (1 << 27)], src=0, group=g)
            self.torch.cuda.synchronize(dev)
            result["ok"] = True
        except Exception as e:  # noqa: BLE001 -- any failure selects the fallback
            result["error"] = "%s: %s" % (type(e).__name__, str(e)[:200])

    def broadcast_key(self, key, dev):
        """key: int64 tensor on `dev`, valid on rank 0.  Returns the elapsed milliseconds."""
        if self.world == 1:
            return None
        torch, dist = self.torch, self.dist
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        ok = False
        why = "requested"
        if self.want == "nccl":
            res = {}
            th = threading.Thread(target=self._nccl_broadcast, args=(key, dev, res), daemon=True)
            th.start()
            th.join(NCCL_TIMEOUT_S + 30)
            self.hung = th.is_alive()   # a collective that never returned: leave through os._exit at the end
            ok = bool(res.get("ok"))
            why = res.get("error", "no answer within %d s" % (NCCL_TIMEOUT_S + 30))
        # every rank must take the same path: one failure anywhere sends all of them through the host
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            self.key_path = "RCCL broadcast (torch.distributed nccl) over xGMI"
        else:
            host = key.cpu() if self.rank == 0 else torch.empty(key.numel(), dtype=torch.int64)
            for off in range(0, host.numel(), 1 << 27):
                dist.broadcast(host[off:off + (1 << 27)], src=0)
            if self.rank != 0:
                key.copy_(host)
            torch.cuda.synchronize(dev)
            self.key_path = "staged through the host over gloo (RCCL: %s)" % why
        return (time.perf_counter() - t0) * 1e3

    def close(self, rc=0):
        if self.world > 1:
            try:
                self.dist.barrier()
                if self.hung:
                    sys.stdout.flush()
                    os._exit(rc)
                self.dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass


def launcher_selftest(args):
    """What the multi-GPU paths do around the kernels, without a GPU.  Under a launcher (or self-launched): rendezvous,
    shard the global batch, replicate a key-shaped tensor from rank 0, reduce the elapsed time with MAX, gather
    per-rank rates (gloo).  With the launch failing (--force-launch-failure): the single-process path -- one thread
    per "device", the same sharding, a barrier on both sides of the timed region, the maximum over the threads."""
    import torch

    from heongpu_amd import sharding
    ref = torch.arange(1 << 16, dtype=torch.int64) * 7 + 3
    if "WORLD_SIZE" not in os.environ:  # single-process fallback, on CPU
        world = args.gpus
        keys = [ref.clone()] + [torch.zeros_like(ref) for _ in range(world - 1)]
        bar = threading.Barrier(world)
        elapsed, slices = [0.0] * world, [None] * world

        def worker(r):
            slices[r] = list(sharding.shard_range(args.batch * world, world, r))
            bar.wait()
            t0 = time.perf_counter()
            time.sleep(0.001 * (r + 1))
            bar.wait()
            elapsed[r] = time.perf_counter() - t0
        span = 1
        while span < world:  # the fan-out order of hegpu_broadcast_key
            for i in range(min(span, world - span)):
                keys[i + span].copy_(keys[i])
            span <<= 1
        ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        ok = all(bool((k == ref).all()) for k in keys)
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "key_broadcast_ok": ok,
                          "max_elapsed_s": max(elapsed), "global_batch": args.batch * world, "slices": slices,
                          "parallelism": "single process, one thread per device (fallback: launch failed)"}))
        return 0 if ok else 1
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0"))
    assert world == args.gpus, "world size %d != --gpus %d" % (world, args.gpus)
    d = Dist(torch, world, rank, "gloo")
    start, count = sharding.shard_range(args.batch * world, world, rank)
    key = ref.clone() if rank == 0 else torch.zeros_like(ref)
    if world > 1:
        d.dist.broadcast(key, src=0)
    ok = bool((key == ref).all())
    mx = d.max_float(0.001 * (rank + 1))
    d.gather_floats(count / (0.001 * (rank + 1)))
    slices = [[start, count]]
    if world > 1:
        own = torch.tensor([start, count], dtype=torch.int64)
        sl = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        d.dist.all_gather(sl, own)
        slices = [[int(v) for v in s] for s in sl]
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "key_broadcast_ok": ok, "max_elapsed_s": mx,
                          "global_batch": args.batch * world, "slices": slices,
                          "parallelism": "one process per GPU (gloo control plane)"}))
    d.close()
    return 0 if ok else 1


# ------------------------------------------------------------------ timing helpers
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


def tile_items(torch, uniq_items, count, first=0):
    """`count` items, item b = uniq_items[(first + b) % len(uniq_items)], back to back in one tensor"""
    u = len(uniq_items)
    return torch.cat([uniq_items[(first + b) % u] for b in range(count)])


def twins_equal(torch, out, item_elems, count, uniq, first=0, used_elems=None):
    """every item equals the first item of the batch that had the same input: number of items compared, all equal?"""
    used = used_elems or item_elems
    view = out[:count * item_elems].view(count, item_elems)[:, :used]
    first_of = {}
    compared, ok = 0, True
    for b in range(count):
        u = (first + b) % uniq
        if u in first_of:
            compared += 1
            ok = ok and bool(torch.equal(view[b], view[first_of[u]]))
        else:
            first_of[u] = b
    return compared, ok


# ------------------------------------------------------------------ the C4 workload on one device
class C4:
    """CKKS N=2^16, Q=16 {60,50x15} | P=1 {60}, depth 0: `B` ciphertext pairs of one rank resident on `dev`."""

    def __init__(self, torch, hg, ctx, dev, first, B, key=None):
        from heongpu_amd import synth
        self.torch, self.hg, self.ctx, self.dev, self.first, self.B = torch, hg, ctx, dev, first, B
        self.primes = [int(v) for v in ctx.table("modulus")]
        self.Q, self.Qp, self.n = ctx.Q_size, ctx.Q_prime_size, N
        l, n = self.Q, self.n
        self.ct_elems, self.out_elems = 2 * l * n, 3 * l * n
        need = (2 * B * self.ct_elems + B * self.out_elems) * 8 + ctx.workspace_bytes(hg.OP_CKKS_RELIN, 0, B) + \
            2 * self.Q * self.Qp * n * 8
        free_b, _ = torch.cuda.mem_get_info(dev)
        if free_b < need + (1 << 30):
            raise SystemExit("bench.py: device %s has %.1f GiB free, the workload needs %.1f GiB"
                             % (dev, free_b / 2**30, need / 2**30))
        self.uniq = min(B, UNIQ)
        a = [synth.synth_ct_t(torch, self.primes, range(l), 2, n, 1 + 10 * u, dev) for u in range(self.uniq)]
        b = [synth.synth_ct_t(torch, self.primes, range(l), 2, n, 2 + 10 * u, dev) for u in range(self.uniq)]
        self.ct1, self.ct2 = tile_items(torch, a, B, first), tile_items(torch, b, B, first)
        self.out = torch.empty(B * self.out_elems, dtype=torch.int64, device=dev)
        self.ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B, device=dev)
        self.key = key if key is not None else torch.empty(2 * self.Q * self.Qp * n, dtype=torch.int64, device=dev)

    def make_key(self):
        from heongpu_amd import synth
        self.key.copy_(synth.synth_key_t(self.torch, self.primes, self.Q, self.Qp, self.n, 3, self.dev))

    def step(self, stream):
        c = self.ctx
        c.ckks_multiply(self.ct1, self.ct_elems, self.ct2, self.ct_elems, self.out, self.out_elems, 0, self.B, stream=stream)
        c.ckks_relinearize_inplace(self.out, self.out_elems, self.key, 0, self.B, self.ws, stream=stream)

    def twins(self):
        return twins_equal(self.torch, self.out, self.out_elems, self.B, self.uniq, self.first, used_elems=self.ct_elems)


def c4_host_inputs(primes, l, n, uniq, first, sample):
    """numpy copies of the first `sample` pairs of a rank's batch (the CPU checker's input)"""
    from heongpu_amd import synth
    a = [synth.synth_ct_np(primes, range(l), 2, n, 1 + 10 * u) for u in range(uniq)]
    b = [synth.synth_ct_np(primes, range(l), 2, n, 2 + 10 * u) for u in range(uniq)]
    c1 = np.concatenate([a[(first + s) % uniq] for s in range(sample)])
    c2 = np.concatenate([b[(first + s) % uniq] for s in range(sample)])
    return c1, c2


def line_skeleton(args, world, value, elapsed, per_rank, Q, parallelism):
    W = 8 * N
    l = Q
    return {
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
            "parallelism": parallelism,
            "algorithmic_bytes_per_op": (6 * l * l + 32 * l + 8) * W,
        },
        "per_rank_ops_per_s": per_rank,
    }


# ------------------------------------------------------------------ one process, N devices (fallback 3)
def run_single_process(args, reason):
    import torch

    import heongpu_amd as hg
    from heongpu_amd import sharding
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    world = args.gpus
    ndev = torch.cuda.device_count()
    devs = [torch.device("cuda", r % ndev) for r in range(world)]
    ctx0 = hg.Context.from_bit_sizes(hg.CKKS, N, LOG_Q, LOG_P)
    ctxs = [ctx0] + [ctx0.clone() for _ in range(world - 1)]
    works, streams = [], []
    for r in range(world):
        torch.cuda.set_device(devs[r])
        ctxs[r].upload_device(devs[r].index)
        first, B = sharding.shard_range(args.batch * world, world, r)
        works.append(C4(torch, hg, ctxs[r], devs[r], first, B))
        streams.append(torch.cuda.Stream(device=devs[r]))
    with torch.cuda.stream(streams[0]):
        works[0].make_key()
    t0 = time.perf_counter()
    hg.broadcast_key(ctxs, [w.key for w in works], works[0].key.numel(), [s.cuda_stream for s in streams])
    for s in streams:
        s.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3
    bar = threading.Barrier(world)
    own, total = [0.0] * world, [0.0] * world
    errors = []

    def worker(r):
        try:
            torch.cuda.set_device(devs[r])
            st = streams[r].cuda_stream
            for _ in range(args.warmup):
                works[r].step(st)
            streams[r].synchronize()
            bar.wait()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                works[r].step(st)
            streams[r].synchronize()
            own[r] = time.perf_counter() - t0
            bar.wait()
            total[r] = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            bar.abort()
    ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if errors:
        raise SystemExit("bench.py: a device thread failed: " + "; ".join(errors))
    elapsed = max(total)
    value = args.batch * world * args.steps / elapsed
    line = line_skeleton(args, world, value, elapsed, [w.B * args.steps / o for w, o in zip(works, own)], ctx0.Q_size,
                         "batch-sharded x%d in ONE process: a thread, a stream and a context per device "
                         "(hegpu_context_upload_device), key replicated with hegpu_broadcast_key (peer copies); "
                         "fallback because %s; %d visible device(s)" % (world, reason, ndev))
    line["key_broadcast_ms"] = bcast_ms
    line["key_bytes"] = works[0].key.numel() * 8
    tw = [w.twins() for w in works]
    same_key = all(bool(torch.equal(works[0].key.to(w.dev), w.key)) for w in works[1:])
    line["checked_items"] = {"twin_compared": sum(t[0] for t in tw), "twins_equal": all(t[1] for t in tw),
                             "key_replicas_equal": same_key, "oracle_compared": 0,
                             "note": "multi-GPU run: the oracle comparison is part of the N=1 line"}
    print(json.dumps(line))
    return 0 if (all(t[1] for t in tw) and same_key) else 1


# ------------------------------------------------------------------ secondary workloads (N=1 only)
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
        x = torch.randint(0, 1 << 49, (polys * n,), dtype=torch.int64, device="cuda")
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
    NTT shared (hegpu_ckks_rotate_hoisted) against k separate hegpu_ckks_apply_galois calls; every hoisted entry
    compared with the separate call's on the device."""
    n, Q, Qp = ctx.n, ctx.Q_size, ctx.Q_prime_size
    stream = torch.cuda.current_stream().cuda_stream
    words = 2 * Q * n
    rnd = lambda k: torch.randint(0, 1 << 49, (k,), dtype=torch.int64, device="cuda")
    ct = rnd(B * words)
    ws = ctx.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, B)  # room for four accumulators (>= OP_CKKS_GALOIS)
    kmax = 8
    keys = [rnd(Q * 2 * Qp * n) for _ in range(kmax)]
    elts = [hg.steps_to_galois_elt(i + 1, n, 5) for i in range(kmax)]
    out = torch.empty(B * kmax * words, dtype=torch.int64, device="cuda")
    out2 = torch.empty(B * kmax * words, dtype=torch.int64, device="cuda")
    res = {"workload": "CKKS N=2^16, Q=16 | P=1 (method I), depth 0, %d ciphertexts, k Galois elements each" % B, "by_k": {}}
    for k in (1, 2, 4, 8):
        h = timer.ms(lambda: ctx.ckks_rotate_hoisted(ct, words, out, k * words, keys[:k], elts[:k], 0, B, ws, stream=stream), 3)

        def separate():
            for i in range(k):
                ctx.ckks_apply_galois(ct, words, out2[i * B * words:], words, keys[i], elts[i], 0, B, ws, stream=stream)
        s_ = timer.ms(separate, 3)
        hv = out[:B * k * words].view(B, k, words)
        sv = out2[:B * k * words].view(k, B, words)
        same = all(bool(torch.equal(hv[:, i], sv[i])) for i in range(k))
        res["by_k"][str(k)] = {"hoisted_rotations_per_s": B * k / (h * 1e-3), "separate_rotations_per_s": B * k / (s_ * 1e-3),
                               "hoisted_ms": h, "separate_ms": s_, "hoisted_equals_separate": same,
                               "checked_items": B * k}
    return res


def secondary_block(torch, hg, timer):
    """The other BASELINE.json configurations, synthetic data, each with the algorithmic bytes of the
    reference's kernel sequence, the fraction of the 8 TB/s HBM peak that rate means, and a check of
    the batch that was timed: `checked` distinct items against the CPU oracle, every other item against its twin."""
    from heongpu_amd import synth
    from oracle import binding as ob  # the checker; nothing below is timed through it
    sec = {}
    stream = torch.cuda.current_stream().cuda_stream
    dev = torch.device("cuda")
    U = 2  # distinct inputs per workload, all of them compared with the oracle

    def check_line(twins, oracle_ok, n_oracle):
        return {"oracle_compared": n_oracle, "oracle_equal": bool(oracle_ok), "twin_compared": twins[0],
                "twins_equal": bool(twins[1])}

    # ---- north_star target 2: BFV N=2^14, default 128-bit chain (Q=8, P=1), 256 pairs
    n, t, B = 1 << 14, 786433, 256
    ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
    ctx.upload()
    primes = [int(v) for v in ctx.table("modulus")]
    Q, Qp, L = ctx.Q_size, ctx.Q_prime_size, len(ctx.table("q_Bsk_merge_modulus"))
    W = 8 * n
    a_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 1 + 10 * u, dev) for u in range(U)]
    b_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 2 + 10 * u, dev) for u in range(U)]
    ct, ct2 = tile_items(torch, a_u, B), tile_items(torch, b_u, B)
    o3 = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    key = synth.synth_key_t(torch, primes, Q, Qp, n, 3, dev)
    wsm, wsr = ctx.workspace(hg.OP_BFV_MULTIPLY, 0, B), ctx.workspace(hg.OP_BFV_RELIN, 0, B)
    m = timer.ms(lambda: ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm, stream=stream), 3)
    o = ob.OracleContext(ob.BFV, 14, primes, Q, 1, t)
    got = hg.to_host(o3[:U * 3 * Q * n]).reshape(U, -1)
    ok = all(np.array_equal(got[u], o.bfv_multiply(hg.to_host(a_u[u]), hg.to_host(b_u[u]))) for u in range(U))
    chk = check_line(twins_equal(torch, o3, 3 * Q * n, B, U), ok, U)
    r = timer.ms(lambda: ctx.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, wsr, stream=stream), 3)
    mul_bytes = (28 * L + 7 * Q) * W
    sec["bfv_n14_multiply"] = {
        "workload": "BFV N=2^14 default 128-bit chain (Q=%d, P=1, Bsk=%d), t=786433, %d pairs resident in HBM" % (Q, L - Q, B),
        "multiplications_per_s": B / (m * 1e-3), "ms_per_batch": m,
        "multiply_relinearize_per_s": B / ((m + r) * 1e-3),
        "reference_sequence_bytes_per_op": mul_bytes, "frac_of_hbm_peak": mul_bytes * B / (m * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "checked_items": chk}
    ctx.close()
    del ct, ct2, o3, key, wsm, wsr, o

    # ---- C3: BFV N=2^15 default chain (Q=14, P=1), rotate_rows by one step, 64 ciphertexts
    n, t, B = 1 << 15, 786433, 64
    ctx = hg.Context.from_default(hg.BFV, n, 1, plain_modulus=t)
    ctx.upload()
    primes = [int(v) for v in ctx.table("modulus")]
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    W = 8 * n
    a_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 1 + 10 * u, dev) for u in range(U)]
    ct = tile_items(torch, a_u, B)
    out = torch.empty(2 * Q * n * B, dtype=torch.int64, device="cuda")
    key = synth.synth_key_t(torch, primes, Q, Qp, n, 3, dev)
    ws = ctx.workspace(hg.OP_BFV_GALOIS, 0, B)
    gal = hg.steps_to_galois_elt(1, n, 3)
    g = timer.ms(lambda: ctx.bfv_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, B, ws, stream=stream), 3)
    o = ob.OracleContext(ob.BFV, 15, primes, Q, 1, t)
    got = hg.to_host(out[:U * 2 * Q * n]).reshape(U, -1)
    key_h = hg.to_host(key)
    ok = all(np.array_equal(got[u], o.bfv_apply_galois(hg.to_host(a_u[u]), key_h, gal)) for u in range(U))
    rot_bytes = (6 * Q * Qp + 6 * Q + 8 * Qp) * W
    sec["c3_bfv_n15_rotate"] = {
        "workload": "BFV N=2^15 default chain (Q=%d, P=1), rotate_rows (Galois key switch method I), %d ciphertexts" % (Q, B),
        "rotations_per_s": B / (g * 1e-3), "ms_per_batch": g,
        "reference_sequence_bytes_per_op": rot_bytes, "frac_of_hbm_peak": rot_bytes * B / (g * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "checked_items": check_line(twins_equal(torch, out, 2 * Q * n, B, U), ok, U)}
    ctx.close()
    del ct, out, key, ws, o, key_h

    # ---- C2: CKKS N=2^14, {50, 40 x 7} | {50}, multiply + relinearize + rescale
    n = 1 << 14
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50])
    ctx.upload()
    primes = [int(v) for v in ctx.table("modulus")]
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    W = 8 * n
    key = synth.synth_key_t(torch, primes, Q, Qp, n, 3, dev)
    key_h = hg.to_host(key)
    a_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 1 + 10 * u, dev) for u in range(U)]
    b_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 2 + 10 * u, dev) for u in range(U)]
    o = ob.OracleContext(ob.CKKS, 14, primes, Q, 1)
    want = []
    for u in range(U):
        w3 = o.ckks_multiply(hg.to_host(a_u[u]), hg.to_host(b_u[u]), 0)
        o.ckks_relinearize(w3, key_h, 0)
        w2 = w3[:2 * Q * n].copy()
        o.ckks_rescale(w2, 0)
        want.append(w2[:2 * (Q - 1) * n])
    c2, c2chk = {}, {}
    for B in (1, 64):
        c1b, c2b = tile_items(torch, a_u, B), tile_items(torch, b_u, B)
        ob_ = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
        wsb, wsb2 = ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B)

        def seq(s=stream):
            ctx.ckks_multiply(c1b, 2 * Q * n, c2b, 2 * Q * n, ob_, 3 * Q * n, 0, B, stream=s)
            ctx.ckks_relinearize_inplace(ob_, 3 * Q * n, key, 0, B, wsb, stream=s)
            ctx.ckks_rescale_inplace(ob_, 3 * Q * n, 0, B, wsb2, stream=s)
        c2[B] = timer.ms(seq, 5)
        nu = min(U, B)
        got = hg.to_host(ob_[:nu * 3 * Q * n]).reshape(nu, -1)
        ok = all(np.array_equal(got[u][:2 * (Q - 1) * n], want[u]) for u in range(nu))
        c2chk["batch%d" % B] = check_line(twins_equal(torch, ob_, 3 * Q * n, B, U, used_elems=2 * (Q - 1) * n), ok, nu)
        if B == 1:
            # the same sequence captured once and replayed as one hipGraph launch (the operator entries
            # allocate nothing and never synchronise): the launch-bound batch-1 case
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                seq(torch.cuda.current_stream().cuda_stream)
            c2["graph"] = timer.ms(graph.replay, 5)
    op_bytes = (6 * Q * Q + 32 * Q + 8 + 6 + 16 * (Q - 1)) * W
    sec["c2_ckks_n14"] = {
        "workload": "CKKS N=2^14, Q=8 {50,40x7} | P=1 {50}, multiply + relinearize + rescale",
        "latency_us_batch1": c2[1] * 1e3, "latency_us_batch1_hipgraph_replay": c2["graph"] * 1e3,
        "ops_per_s_batch64": 64 / (c2[64] * 1e-3),
        "reference_sequence_bytes_per_op": op_bytes,
        "frac_of_hbm_peak_batch64": op_bytes * 64 / (c2[64] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "checked_items": c2chk}
    ctx.close()
    del o, key, key_h

    # ---- key-switching method II (selected by the reference whenever P_size > 1): the C4 shape with four special
    # primes, Q = 16 x 50 bits | P = 4 x 50 bits (d = 4 digits of 4 primes), multiply + relinearize, 64 pairs
    n, B = 1 << 16, 64
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] * 16, [50] * 4, sec=hg.SEC_NONE)
    ctx.upload()
    primes = [int(v) for v in ctx.table("modulus")]
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    d = -(-Q // 4)
    a_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 1 + 10 * u, dev) for u in range(U)]
    b_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 2 + 10 * u, dev) for u in range(U)]
    c1b, c2b = tile_items(torch, a_u, B), tile_items(torch, b_u, B)
    ob_ = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
    key = synth.synth_key_t(torch, primes, d, Qp, n, 3, dev)
    wsb = ctx.workspace(hg.OP_CKKS_RELIN, 0, B)

    def seq2():
        ctx.ckks_multiply(c1b, 2 * Q * n, c2b, 2 * Q * n, ob_, 3 * Q * n, 0, B, stream=stream)
        ctx.ckks_relinearize_inplace(ob_, 3 * Q * n, key, 0, B, wsb, stream=stream)
    m2 = timer.ms(seq2, 3)
    o = ob.OracleContext(ob.CKKS, 16, primes, Q, 4)
    key_h = hg.to_host(key)
    got = hg.to_host(ob_[:U * 3 * Q * n]).reshape(U, -1)
    ok = True
    for u in range(U):
        w3 = o.ckks_multiply(hg.to_host(a_u[u]), hg.to_host(b_u[u]), 0)
        o.ckks_relinearize_II(w3, key_h, 0)
        ok = ok and np.array_equal(got[u][:2 * Q * n], w3[:2 * Q * n])
    sec["ckks_n16_method_II"] = {
        "workload": "CKKS N=2^16, Q=16 x 50 bits | P=4 x 50 bits (hybrid key switching, 4 digits), multiply + relinearize, "
                    "%d pairs" % B,
        "multiply_relinearize_per_s": B / (m2 * 1e-3), "ms_per_batch": m2,
        "checked_items": check_line(twins_equal(torch, ob_, 3 * Q * n, B, U, used_elems=2 * Q * n), ok, U)}
    ctx.close()
    del c1b, c2b, ob_, key, wsb, o, key_h

    # ---- C5: TFHE STD128 NAND gate bootstrap, 8192 concurrent gates (the whole config on one GPU;
    # its 8-GPU share is 1024)
    t = hg.TfheContext()
    rng = np.random.default_rng(1)
    S, TU, TCHK = 8192, 64, 2
    polys = t.int("bootkey_elems") // 1024
    v = rng.integers(-2**31, 2**31, polys, dtype=np.int64)  # constant polynomials: NTT image = the constant
    lifted = np.where(v < 0, v + t.prime, v).astype(np.uint64)
    bk_h = np.repeat(lifted, 1024)
    bk = torch.from_numpy(bk_h.view(np.int64)).cuda()
    r32 = lambda k: rng.integers(-2**31, 2**31, k, dtype=np.int64).astype(np.int32)
    ks_a_h, ks_b_h = r32(t.int("kskey_a_elems")), r32(t.int("kskey_b_elems"))
    a1u, a2u, b1u, b2u = r32(TU * 512), r32(TU * 512), r32(TU), r32(TU)
    rep = S // TU
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a1, a2, b1, b2 = cu(np.tile(a1u, rep)), cu(np.tile(a2u, rep)), cu(np.tile(b1u, rep)), cu(np.tile(b2u, rep))
    ks_a, ks_b = cu(ks_a_h), cu(ks_b_h)
    prepared = t.prepare_bootkey(bk)
    out_a = torch.empty(S * 512, dtype=torch.int32, device="cuda")
    out_b = torch.empty(S, dtype=torch.int32, device="cuda")
    ws = torch.empty((512 + 1024 + 2) * S, dtype=torch.int32, device="cuda")
    g = timer.ms(lambda: t.gate(hg.GATE_NAND, a1, b1, a2, b2, out_a, out_b, prepared, ks_a, ks_b, S, ws, stream=stream), 2)
    ga, gb = out_a.view(S, 512), out_b
    tw_ok = bool(torch.equal(ga, ga[:TU].repeat(rep, 1))) and bool(torch.equal(gb, gb[:TU].repeat(rep)))
    ot = ob.OracleTfhe()
    want_a, want_b = ot.gate(hg.GATE_NAND, a1u[:TCHK * 512], b1u[:TCHK], a2u[:TCHK * 512], b2u[:TCHK], bk_h, ks_a_h, ks_b_h)
    ok = np.array_equal(ga[:TCHK].cpu().numpy().reshape(-1), want_a) and np.array_equal(gb[:TCHK].cpu().numpy(), want_b)
    sec["c5_tfhe_gates"] = {
        "workload": "TFHE STD128 NAND gate bootstrap (pre-computation, blind rotate n=512, sample extraction, key switch), "
                    "%d concurrent gates (%d distinct inputs), torus32 boot key (FP64 blind rotate)" % (S, TU),
        "gates_per_s": S / (g * 1e-3), "ms_per_batch": g,
        "reference_sequence_bytes_per_gate": 72 * (1 << 20),
        "frac_of_hbm_peak": 72 * (1 << 20) * S / (g * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "checked_items": check_line((S - TU, tw_ok), ok, TCHK)}
    t.close()
    return sec


def power_sample(torch, step):
    """Package power and shader clock while the step loops (outside the timed region): the kernels of the step are
    bound by FP64 issue at the clock the package power limit allows, which this records next to the
    throughput.  rocm-smi is polled from the host while ~3 s of steps sit in the stream; None if it is not there."""
    import re
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


def issue_figures(tj, names, ms):
    """Vector-ALU issue figures of a launch group from the committed profile (profiles/traffic.json, written by
    tools/profile.sh from rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE passes of this very
    command): lane_instructions = wave-level vector instructions x 64; valu_busy = the share of the kernels' own
    cycles in which their SIMDs issued a vector instruction; frac_of_issue_ceiling = lane_instructions over what 1024
    SIMDs x 16 lanes issue in the group's LIVE duration at the profile's measured shader clock."""
    ks = (tj or {}).get("step_kernels_sq") or {}
    hit = [ks[k] for k in names if k in ks]
    if len(hit) != len(names) or not hit:
        return None
    wave_insts = sum(h["valu_wave_insts"] for h in hit)
    busy_cyc = sum(h["valu_busy_cycles_per_simd"] for h in hit)
    cyc = sum(h["cycles"] for h in hit)
    prof_s = sum(h["seconds"] for h in hit)
    sclk = cyc / prof_s if prof_s > 0 else None
    out = {"lane_instructions": wave_insts * 64, "valu_busy": busy_cyc / cyc if cyc else None,
           "sclk_GHz_in_kernel": sclk / 1e9 if sclk else None, "kernels": names}
    if sclk:
        out["frac_of_issue_ceiling"] = wave_insts * 64 / (SIMDS * LANES * sclk * ms * 1e-3)
    return out


# ------------------------------------------------------------------ one rank (N = 1, or one process per GPU)
def run_rank(args):
    import torch

    import heongpu_amd as hg
    from heongpu_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: world size %d (WORLD_SIZE) does not match --gpus %d" % (world, args.gpus))
    # one process per GPU; on a box with fewer devices than ranks (a functional check on one GPU) ranks share devices
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = Dist(torch, world, rank, args.backend)

    ctx = hg.Context.from_bit_sizes(hg.CKKS, N, LOG_Q, LOG_P)
    ctx.upload()
    Q, Qp, n = ctx.Q_size, ctx.Q_prime_size, N
    l, rc = Q, Qp
    first, B = sharding.shard_range(args.batch * world, world, rank)
    W = 8 * n  # bytes of one limb polynomial
    work = C4(torch, hg, ctx, dev, first, B)
    if rank == 0:
        work.make_key()   # produced on the device, 272 MiB
    bcast_ms = dist.broadcast_key(work.key, dev)
    stream = torch.cuda.current_stream().cuda_stream
    step = lambda: work.step(stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = dist.max_float(time.perf_counter() - t0)
    per_rank = dist.gather_floats(B * args.steps / own_elapsed)
    tw_n, tw_ok = work.twins()
    tw_all = dist.gather_floats(float(tw_n if tw_ok else -1))
    value = args.batch * world * args.steps / elapsed

    parallelism = "batch-sharded x%d, one process per GPU (gloo control plane), key broadcast: %s; no data-path collective" \
        % (world, dist.key_path)
    line = line_skeleton(args, world, value, elapsed, per_rank, Q, parallelism)
    if bcast_ms is not None:
        line["key_broadcast_ms"] = bcast_ms
        line["key_bytes"] = work.key.numel() * 8
    line["checked_items"] = {"distinct_inputs": work.uniq, "twin_compared": int(sum(max(t, 0) for t in tw_all)),
                             "twins_equal": all(t >= 0 for t in tw_all), "oracle_compared": 0}

    if args.step_only or rank != 0 or world > 1:
        if rank == 0:
            if world > 1:
                line["checked_items"]["note"] = "multi-GPU run: the oracle comparison is part of the N=1 line"
            print(json.dumps(line))
        rc = 0 if line["checked_items"]["twins_equal"] else 1
        dist.close(rc)
        return rc

    out, key, ws, ct1, ct2 = work.out, work.key, work.ws, work.ct1, work.ct2
    ct_elems, out_elems = work.ct_elems, work.out_elems
    result = hg.to_host(out.view(B, out_elems)[:, :ct_elems]) if not args.no_cpu_baseline else None

    # ---- roofline of the transform: the forward NTT launch pair at the key-switch
    # shape (l*rc limb NTTs per ciphertext).  Algorithmic bytes = 2W per limb NTT; HIP events on the
    # launch stream.
    timer = Timer(torch)
    polys = B * l * rc
    ntt_ms = {inv: timer.ms(lambda: ctx.ntt(ws, ws, inv, polys, rc, stream=stream)) for inv in (False, True)}
    ntt_bytes = polys * 2 * W
    fwd_gbps = ntt_bytes / (ntt_ms[False] * 1e-3) / 1e9
    inv_gbps = ntt_bytes / (ntt_ms[True] * 1e-3) / 1e9
    tj = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)

    # ---- every launch group of a step on its own (hegpu_probe_ckks_relinearize), against the bytes the
    # group has to move as it is built (fused kernels: fewer than the reference's sequence):
    #   W = one limb; per ciphertext at depth 0 (l = 16, Q' = 17)
    def probe(ph):
        return timer.ms(lambda: ctx.probe_ckks_relinearize(out, out_elems, key, 0, B, ws, ph, stream=stream))
    g256 = lambda polys_: "grid %d" % (16 * polys_ * 256)   # N / 4096 = 16 tiles (or column tiles) per polynomial
    ksg = "grid %d" % (((16 * rc + 7) // 8) * 8 * B * 256)
    groups = [
        ("ckks_multiply: k_cross_multiplication", timer.ms(
            lambda: ctx.ckks_multiply(ct1, ct_elems, ct2, ct_elems, out, out_elems, 0, B, stream=stream)),
         7 * l * W * B, "read 4l, write 3l limbs",
         ["hegpu::k_cross_multiplication grid %d" % (B * l * n // 2)], "int"),
        ("INTT of c2: ntt_inv_row + ntt_inv_col", probe(1), 2 * l * W * B,
         "l limb INTTs, 2W each (the column stages of the FP64 limbs run inside the next group's kernel)",
         ["hegpu::ntt_inv_row " + g256(l * B), "hegpu::ntt_inv_col<8, false> " + g256(l * B)], "fp64"),
        ("decomposing column pass: ntt_fwd_col_multi + ntt_fwd_col<8,true>", probe(2),
         (l + l * rc - l) * W * B, "read l source limbs once, write l*Q' - l half-transformed digits",
         ["hegpu::ntt_fwd_col_multi<8> " + g256(l * B), "hegpu::ntt_fwd_col<8, true> " + g256(2 * l * B)], "fp64"),
        ("row pass + key inner product: ks_row_mac_fp + ks_row_mac", probe(4),
         ((l * rc - l) + l + 2 * rc) * W * B + 2 * l * rc * W,
         "read l*Q' - l digits + l identity limbs, write 2Q' limbs per ciphertext; the key (2 l Q' limbs) once per batch",
         ["hegpu::ks_row_mac_fp " + ksg, "hegpu::ks_row_mac " + ksg], "fp64"),
        ("INTT of the two P limbs", probe(8), 2 * 2 * W * B, "2 limb INTTs",
         ["hegpu::ntt_inv_row " + g256(2 * B), "hegpu::ntt_inv_col<8, false> " + g256(2 * B)], "int"),
        ("mod-down NTT with stage one / two fused: ntt_fwd_col_multi + ntt_fwd_row", probe(16),
         (2 + 2 * l + 2 * l + 3 * 2 * l + 2 * l) * W * B,
         "column pass: read 2 P limbs, write 2l; row pass: read 2l + the 2l accumulated limbs + 2l of ct, write 2l",
         ["hegpu::ntt_fwd_col_multi<8> " + g256(2 * B), "hegpu::ntt_fwd_col<8, true> " + g256(2 * B),
          "hegpu::ntt_fwd_row " + g256(2 * l * B)], "fp64"),
    ]
    copy_frac = ((tj or {}).get("copy_ceiling_GBps") or 5230.0) / HBM_PEAK_GBPS
    in_step = []
    for name, ms, by, note, kernels, alu in groups:
        g = {"launches": name, "ms": ms, "algorithmic_bytes": by, "accounting": note,
             "achieved_GBps": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        g["frac_of_copy_ceiling"] = g["frac"] / copy_frac
        isu = issue_figures(tj, kernels, ms)
        if isu:
            g.update(isu)
            # the ceiling that binds: the larger of (bytes over what a read+write stream sustains) and (vector
            # instructions over what the SIMDs can issue)
            g["bound"] = ("valu-" + alu) if isu.get("frac_of_issue_ceiling", 0) > g["frac_of_copy_ceiling"] else "hbm"
            g["frac_of_binding_ceiling"] = max(isu.get("frac_of_issue_ceiling", 0), g["frac_of_copy_ceiling"])
        else:
            g["bound"] = "hbm (no committed SQ counters for this group)"
        in_step.append(g)
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
        "in_step_dominant": {k: dominant.get(k) for k in ("launches", "ms", "frac", "achieved_GBps", "algorithmic_bytes", "bound",
                                                            "valu_busy", "frac_of_issue_ceiling", "frac_of_binding_ceiling")},
    }
    line["ntt"] = {"forward_GBps": fwd_gbps, "inverse_GBps": inv_gbps, "n": N, "limbs": polys}
    line["in_step"] = in_step
    # the reference's own kernel sequence would have to move its bytes (SURVEY 8d, 1028 MiB per op) at this rate to
    # keep up; above the 8 TB/s peak it only says that the fused path moves fewer bytes -- not an efficiency
    line["reference_sequence_bytes_rate_GBps"] = (6 * l * l + 32 * l + 8) * W * value / 1e9
    if tj:
        if tj.get("limb_ntts_per_launch") == polys:
            line["roofline"]["traffic"] = tj["bytes_per_launch"]
            line["roofline"]["traffic_source"] = tj.get("source", "profiles/traffic.json")
            nsq = (tj.get("roofline_pair_sq") or {})
            if nsq.get("valu_wave_insts") and nsq.get("cycles") and nsq.get("seconds"):
                sclk = nsq["cycles"] / nsq["seconds"]
                line["roofline"]["issue"] = {
                    "lane_instructions": nsq["valu_wave_insts"] * 64, "valu_busy": nsq["valu_busy_cycles_per_simd"] / nsq["cycles"],
                    "frac_of_issue_ceiling": nsq["valu_wave_insts"] * 64 / (SIMDS * LANES * sclk * ntt_ms[False] * 1e-3),
                    "frac_of_copy_ceiling_two_passes": 2 * fwd_gbps / ((tj.get("copy_ceiling_GBps") or 5230.0))}
        if tj.get("step_bytes") and tj.get("step_batch") == B:
            line["roofline"]["step_traffic"] = tj["step_bytes"]
            line["hbm_fraction_moved"] = tj["step_bytes"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS
        if tj.get("copy_ceiling_GBps"):
            line["roofline"]["copy_ceiling_GBps"] = tj["copy_ceiling_GBps"]

    if not args.no_secondary:
        line["power"] = power_sample(torch, step)
        line["ntt_by_degree"] = ntt_sweep(torch, hg, timer)
    del work, ct1, ct2, out, ws
    torch.cuda.empty_cache()
    if not args.no_secondary:
        line["secondary"] = secondary_block(torch, hg, timer)
        line["secondary"]["hoisted_rotations"] = hoisted_rotation_block(torch, hg, timer, ctx)

    if not args.no_cpu_baseline:
        from heongpu_amd import synth
        from oracle import binding as ob
        primes = [int(v) for v in ctx.table("modulus")]
        o = ob.OracleContext(ob.CKKS, 16, primes, Q, 1)
        cores = ob.lib().o_omp_threads()
        sample = min(args.cpu_sample or max(2 * cores, 2), B)
        c1, c2 = c4_host_inputs(primes, l, n, min(B, UNIQ), first, sample)
        key_h = synth.synth_key_np(primes, Q, Qp, n, 3)
        o3 = np.zeros(sample * out_elems, dtype=np.uint64)
        t0 = time.perf_counter()
        ob.lib().o_ckks_mul_relin_batch(o.h, c1.ctypes.data, c2.ctypes.data, o3.ctypes.data, key_h.ctypes.data, 0, sample)
        cpu_s = time.perf_counter() - t0
        o3 = o3.reshape(sample, out_elems)[:, :ct_elems]
        equal = [bool(np.array_equal(o3[s], result[s])) for s in range(sample)]
        line["checked_items"]["oracle_compared"] = sample
        line["checked_items"]["oracle_equal"] = all(equal)
        line["checked_items"]["total"] = min(B, sample + line["checked_items"]["twin_compared"])
        line["cpu_baseline"] = {
            "value": sample / cpu_s,
            "unit": "multiply+relinearize/s",
            "cores": cores,
            "kind": "port",
            "sample": "%d ciphertext pairs of the same workload (CKKS N=2^16 L=16 mul+relin), CPU oracle, "
                      "OpenMP over pairs, %.1f s" % (sample, cpu_s),
            "gpu_matches_cpu_bit_exact": all(equal),
            "gpu_items_compared": sample,
        }
    print(json.dumps(line))
    dist.close()
    ok = line["checked_items"]["twins_equal"] and line["checked_items"].get("oracle_equal", True)
    return 0 if ok else 1


def main():
    args = parse_args()
    if args.launcher_selftest:
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
            rc = self_launch(args)
            if rc == 0:
                raise SystemExit(0)
            print("bench.py: the launch of %d ranks failed (rc %d): single-process path" % (args.gpus, rc), file=sys.stderr)
        raise SystemExit(launcher_selftest(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        reason = "--single-process"
        if not args.single_process:
            rc = self_launch(args)
            if rc == 0:
                raise SystemExit(0)
            reason = "the launch of the ranks (torch.distributed.run) failed with exit code %d" % rc
            print("bench.py: " + reason + ": one process, one thread per device", file=sys.stderr)
        raise SystemExit(run_single_process(args, reason))
    raise SystemExit(run_rank(args))


if __name__ == "__main__":
    main()
