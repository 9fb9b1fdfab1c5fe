#!/usr/bin/env python3
"""bench.py -- homomorphic mults/sec (CKKS N=2^16, L=16) + NTT GB/s on MI355X.

One "step" = one pass of the hot path (multiply + relinearize_inplace, reference
benchmark/benchmark_ckks.cpp:123-137) over one batch of independent synthetic ciphertext pairs
that are already resident in HBM.  BASELINE.json config C4: 512 pairs sharded over 8 GPUs =
64 pairs per GPU; weak scaling: the global batch is 64 x n_gpus pairs, every rank owns its
`sharding.shard_range` slice, the relinearization key is produced on rank 0 and replicated once.
There is no collective on the data path.  `--workload c5` runs BASELINE.json config C5 the same way: TFHE NAND gate
bootstraps, 1024 gates per GPU (8192 over 8), sharded along `shape`, the prepared boot key (64 MiB) and the
key-switch key (48 MiB) replicated once.

Multi-GPU (`--gpus N`), in this order of preference -- the line says which one ran (`config.parallelism`):
  1. one process per GPU (torch.distributed.run, launched by bench.py itself when it is not already under a
     launcher): control plane on gloo, the key broadcast on RCCL (backend "nccl") over xGMI;
  2. the same, with the keys staged through the host over gloo when RCCL cannot be brought up within 60 s;
  3. one process, one thread and one context per device, the keys replicated with hegpu_broadcast_bytes (peer copies:
     peer access checked per edge, flat fan-out or chunked tree -- the line names the path) -- when the launch of the
     ranks itself fails, or with --single-process.
Ranks beyond the number of visible devices share devices (a functional check on a 1-GPU box).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (forward NTT launch pair),
`in_step` (every launch group of the step timed on its own with HIP events), `checked_items` (every output of the timed
batch verified: against the CPU oracle and against its twin), `secondary` (the other BASELINE.json configurations, timed
and verified in the same run) and `cpu_baseline` (the CPU oracle timed on this host, rank 0, N=1 only).

WHERE EACH NUMBER WAS MEASURED.  Milliseconds, rates and checks are LIVE (this run, HIP events on the launch stream).
Byte counts and vector-instruction counts per launch do not depend on the box; they are read from the committed
counter passes (`profiles/profile.json`, built by tools/build_profile_json.py from `tools/prof_all.sh` runs of this very
script's `--profile-workload` modes) and every field that comes from there sits under a key named `from_profile`,
together with the directory, the profile run's own milliseconds and shader clock; fractions of the issue ceiling are
the profile run's own (its instructions over its cycles), the live time is printed next to the profile's with their
ratio.  The oracle is imported only by the verification / cpu_baseline legs: inputs come from heongpu_amd.synth.
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
GATES_PER_GPU = 1024    # config C5: 8192 gates over 8 GPUs
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
UNIQ = 64               # distinct seeded pairs of the global batch (pair g has the inputs of index g % UNIQ): a rank's 64 pairs are
#                         all different, every rank of an N-GPU run computes the same 64 -- what the cross-rank check compares
TFHE_UNIQ = 64          # distinct seeded gate inputs
SIMDS, LANES = 1024, 16  # 256 CUs x 4 SIMDs, 16 lanes each: a wave64 vector instruction holds its SIMD for 4 cycles
NCCL_TIMEOUT_S = 60
PROFILE_JSON = os.path.join(ROOT, "profiles", "profile.json")
PROFILE_WORKLOADS = ("c4_step", "ntt_pair", "bfv_n14_multiply", "c3_bfv_n15_rotate", "c2_ckks_n14_b1", "c2_ckks_n14_b64",
                     "ckks_n16_method_II", "c5_tfhe_gates")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("c4", "c5"), default="c4",
                    help="c4: CKKS N=2^16 multiply + relinearize (the headline); c5: TFHE NAND gate bootstraps")
    ap.add_argument("--batch", type=int, default=0,
                    help="units per GPU per step: ciphertext pairs (c4, default 64) or gates (c5, default 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE.json configurations")
    ap.add_argument("--step-only", action="store_true",
                    help="warm-up + timed steps only (counter passes: every dispatch belongs to a step)")
    ap.add_argument("--profile-workload", choices=PROFILE_WORKLOADS, default=None,
                    help="counter passes (tools/prof_all.sh): set the named workload up and run it --reps times, nothing else")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs for the CPU baseline (0 = one per host core)")
    ap.add_argument("--backend", default=None, help="backend of the key broadcast (default nccl = RCCL; gloo stages through the host)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in one process: a thread and a context per device, hegpu_broadcast_bytes")
    ap.add_argument("--force-launch-failure", action="store_true",
                    help="testing: pretend the launch of the ranks failed (exercises the single-process fallback)")
    ap.add_argument("--allow-shared-devices", action="store_true",
                    help="let --gpus N run with fewer visible devices than ranks (functional check on a 1-GPU box): the line then "
                         "reports n_gpus = distinct devices and ranks = N")
    ap.add_argument("--preflight", action="store_true",
                    help="no timing: print the device count, the peer-access matrix, the RCCL version and which of the three "
                         "multi-GPU tiers --gpus N would take on this node")
    ap.add_argument("--detail", default=None,
                    help="where the full detail of the run goes (default gpurun_out/bench_detail.json); stdout carries only the "
                         "compact contract line")
    ap.add_argument("--compact-from", default=None,
                    help="no GPU work: read a detail file and print the compact contract line built from it")
    ap.add_argument("--selftest-corrupt-rank", type=int, default=-1,
                    help="testing: one bit of this rank's key replica is flipped after the broadcast: the line must say "
                         "key_digest_equal / cross_rank_equal false and the run must fail")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: exercise launch, sharding, key replication and the max-over-ranks reduction on CPU")
    a = ap.parse_args(argv)
    if a.batch <= 0:
        a.batch = PAIRS_PER_GPU if a.workload == "c4" else GATES_PER_GPU
    return a


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
    # rank 0 touches this file when it has printed its line: a non-zero exit AFTER that is a failed check of a run that
    # took place (a wrong replica, a mismatch) and must not be mistaken for a launch failure and retried in one process
    import tempfile
    fd, marker = tempfile.mkstemp(prefix="hegpu_bench_ran_")
    os.close(fd)
    os.unlink(marker)
    env["HEGPU_BENCH_RAN_MARKER"] = marker
    rc = subprocess.call(cmd, env=env)
    if os.path.exists(marker):
        os.unlink(marker)
        if rc != 0:
            raise SystemExit(rc)
    return rc


def mark_ran():
    m = os.environ.get("HEGPU_BENCH_RAN_MARKER")
    if m:
        try:
            open(m, "w").close()
        except OSError:
            pass


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

    def gather_i64(self, values):
        """every rank's list of 64-bit values (bit patterns as Python ints), rank-major"""
        vals = [v - (1 << 64) if v >> 63 else v for v in (int(x) & ((1 << 64) - 1) for x in values)]
        if self.world == 1:
            return [[v & ((1 << 64) - 1) for v in vals]]
        own = self.torch.tensor(vals, dtype=self.torch.int64)
        out = [self.torch.zeros_like(own) for _ in range(self.world)]
        self.dist.all_gather(out, own)
        return [[int(v) & ((1 << 64) - 1) for v in t.tolist()] for t in out]

    def _nccl_broadcast(self, tensors, dev, result):
        try:
            import datetime
            g = self.dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=NCCL_TIMEOUT_S), device_id=dev)
            probe = self.torch.ones(1, device=dev)
            self.dist.all_reduce(probe, group=g)
            self.torch.cuda.synchronize(dev)
            if int(probe.item()) != self.world:
                raise RuntimeError("RCCL all-reduce probe returned %r" % probe.item())
            for key in tensors:
                flat = key.view(-1)
                for off in range(0, flat.numel(), 1 << 27):  # <= 1 GiB per message
                    self.dist.broadcast(flat[off:off + (1 << 27)], src=0, group=g)
            self.torch.cuda.synchronize(dev)
            result["ok"] = True
        except Exception as e:  # noqa: BLE001 -- any failure selects the fallback
            result["error"] = "%s: %s" % (type(e).__name__, str(e)[:200])

    def broadcast_keys(self, tensors, dev):
        """tensors: device tensors on `dev`, valid on rank 0.  Returns the elapsed milliseconds."""
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
            th = threading.Thread(target=self._nccl_broadcast, args=(tensors, dev, res), daemon=True)
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
            for key in tensors:
                host = key.cpu().view(-1) if self.rank == 0 else torch.empty(key.numel(), dtype=key.dtype)
                for off in range(0, host.numel(), 1 << 27):
                    dist.broadcast(host[off:off + (1 << 27)], src=0)
                if self.rank != 0:
                    key.view(-1).copy_(host)
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


def selftest_keys(torch, workload):
    """key-shaped stand-ins (CPU): one tensor for C4's relinearization key, three for C5 (prepared boot key, key-switch a / b)"""
    if workload == "c4":
        return [torch.arange(1 << 16, dtype=torch.int64) * 7 + 3]
    return [torch.arange(1 << 14, dtype=torch.int64) * 5 + 1, (torch.arange(1 << 15, dtype=torch.int64) % 65521).to(torch.int32),
            (torch.arange(3000, dtype=torch.int64) % 251).to(torch.int32)]


def selftest_outputs(torch, keys, slice_):
    """stand-in "outputs" of a rank (CPU): unit g of its slice = a function of its input index g % 64 and of the rank's
    own key replica -- so a broken replica shows in the outputs exactly as on the device.  Returns (digests, present)."""
    start, count = slice_
    dig, present = [0] * 64, [0] * 64
    k = keys[0].reshape(-1).to(torch.int64)
    for g in range(start, start + count):
        u = g % 64
        if not present[u]:
            row = (k[:16384] * (2 * u + 1) + u).reshape(1, -1)
            dig[u], present[u] = row_digests(torch, row)[0], 1
    return dig, present


def launcher_selftest(args):
    """What the multi-GPU paths do around the kernels, without a GPU.  Under a launcher (or self-launched): rendezvous,
    shard the global batch, replicate the key-shaped tensors from rank 0, reduce the elapsed time with MAX, gather
    per-rank rates (gloo).  With the launch failing (--force-launch-failure): the single-process path -- one thread
    per "device", the same sharding, a barrier on both sides of the timed region, the maximum over the threads."""
    import torch

    from heongpu_amd import sharding
    refs = selftest_keys(torch, args.workload)
    if "WORLD_SIZE" not in os.environ:  # single-process fallback, on CPU
        world = args.gpus
        keys = [[r.clone() for r in refs]] + [[torch.zeros_like(r) for r in refs] for _ in range(world - 1)]
        bar = threading.Barrier(world)
        elapsed, slices = [0.0] * world, [None] * world

        def worker(r):
            slices[r] = list(sharding.shard_range(args.batch * world, world, r))
            bar.wait()
            t0 = time.perf_counter()
            time.sleep(0.001 * (r + 1))
            bar.wait()
            elapsed[r] = time.perf_counter() - t0
        for j in range(1, world):  # hegpu_broadcast_bytes on a fully connected node: a flat fan-out from device 0
            for a, b in zip(keys[0], keys[j]):
                b.copy_(a)
        if 0 <= args.selftest_corrupt_rank < world:
            keys[args.selftest_corrupt_rank][0][12345 % keys[0][0].numel()] ^= 1
        xc = cross_check([[digest64(torch, k) for k in ks] for ks in keys],
                         *zip(*[selftest_outputs(torch, ks, slices_r) for ks, slices_r in
                                zip(keys, [sharding.shard_range(args.batch * world, world, r) for r in range(world)])]))
        ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        ok = all(bool((k == ref).all()) for ks in keys for k, ref in zip(ks, refs))
        print(json.dumps({"launcher_selftest": True, "workload": args.workload, "n_gpus": world, "key_broadcast_ok": ok,
                          "replicated_tensors": len(refs), "max_elapsed_s": max(elapsed),
                          "global_batch": args.batch * world, "slices": slices,
                          "key_digest_equal": xc["key_digest_equal"], "cross_rank_equal": xc["cross_rank_equal"],
                          "distinct_inputs": xc["distinct_inputs"], "ranks_disagreeing_on_key": xc["ranks_disagreeing_on_key"],
                          "parallelism": "single process, one thread per device (fallback: launch failed)"}))
        return 0 if (ok and xc["key_digest_equal"] and xc["cross_rank_equal"]) else 1
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0"))
    assert world == args.gpus, "world size %d != --gpus %d" % (world, args.gpus)
    d = Dist(torch, world, rank, "gloo")
    start, count = sharding.shard_range(args.batch * world, world, rank)
    keys = [r.clone() if rank == 0 else torch.zeros_like(r) for r in refs]
    if world > 1:
        for k in keys:
            d.dist.broadcast(k, src=0)
    if args.selftest_corrupt_rank == rank:
        keys[0][12345 % keys[0].numel()] ^= 1
    ok = all(bool((k == r).all()) for k, r in zip(keys, refs)) or args.selftest_corrupt_rank == rank
    od, op = selftest_outputs(torch, keys, (start, count))
    xc = cross_check(d.gather_i64([digest64(torch, k) for k in keys]), d.gather_i64(od), d.gather_i64(op))
    mx = d.max_float(0.001 * (rank + 1))
    d.gather_floats(count / (0.001 * (rank + 1)))
    slices = [[start, count]]
    if world > 1:
        own = torch.tensor([start, count], dtype=torch.int64)
        sl = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        d.dist.all_gather(sl, own)
        slices = [[int(v) for v in s] for s in sl]
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "workload": args.workload, "n_gpus": world, "key_broadcast_ok": ok,
                          "replicated_tensors": len(refs), "max_elapsed_s": mx,
                          "global_batch": args.batch * world, "slices": slices,
                          "key_digest_equal": xc["key_digest_equal"], "cross_rank_equal": xc["cross_rank_equal"],
                          "distinct_inputs": xc["distinct_inputs"], "ranks_disagreeing_on_key": xc["ranks_disagreeing_on_key"],
                          "parallelism": "one process per GPU (gloo control plane)"}))
        mark_ran()
    rc = 0 if (ok and xc["key_digest_equal"] and xc["cross_rank_equal"]) else 1
    d.close(rc)
    return rc


# ------------------------------------------------------------------ what makes an N > 1 line prove itself
_M64 = (1 << 64) - 1


def _weights(torch, n, device, offset=0):
    """n odd pseudo-random 64-bit weights (splitmix64 of the element index), as int64 bit patterns"""
    from heongpu_amd import synth
    idx = torch.arange(offset, offset + n, dtype=torch.int64, device=device)
    return synth._splitmix64_t(torch, idx) | 1


def digest64(torch, t, chunk=1 << 24):
    """64-bit digest of a tensor: sum of x_i * w_i modulo 2^64 with odd pseudo-random w_i.  One changed element always
    changes it (w_i is a unit modulo 2^64); computed where the tensor lives."""
    x = t.reshape(-1)
    if x.dtype != torch.int64:
        x = x.to(torch.int64)
    acc = 0
    for off in range(0, x.numel(), chunk):
        part = x[off:off + chunk]
        acc = (acc + int((part * _weights(torch, part.numel(), part.device, off)).sum().item())) & _M64
    return acc


def row_digests(torch, rows):
    """one digest per row of a 2-D integer tensor (the same weights for every row)"""
    x = rows if rows.dtype == torch.int64 else rows.to(torch.int64)
    w = _weights(torch, x.shape[1], x.device)
    return [int((x[i] * w).sum().item()) & _M64 for i in range(x.shape[0])]


def cross_check(key_digests, out_digests, out_present):
    """Pure.  key_digests[r]: the digests of rank r's replicated tensors; out_digests[r][u] / out_present[r][u]: rank r's
    output digest for distinct input u (0 / absent if its slice holds no item with that input).
      key_digest_equal: every rank holds the bits rank 0 holds;
      cross_rank_equal: whenever two ranks computed on the same input they produced the same output."""
    key_ok = all(k == key_digests[0] for k in key_digests)
    disagree, seen = [], 0
    for u in range(len(out_digests[0])):
        got = {out_digests[r][u] for r in range(len(out_digests)) if out_present[r][u]}
        seen += 1 if got else 0
        if len(got) > 1:
            disagree.append(u)
    return {"key_digest_equal": key_ok, "cross_rank_equal": not disagree, "distinct_inputs": seen,
            "ranks_disagreeing_on_key": [r for r, k in enumerate(key_digests) if k != key_digests[0]],
            "inputs_with_differing_outputs": disagree[:8]}


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


# ------------------------------------------------------------------ counters from the committed profile
def load_profile():
    if not os.path.exists(PROFILE_JSON):
        return None
    with open(PROFILE_JSON) as f:
        return json.load(f)


def prof_group(prof, workload, live_ms, match=None, algorithmic_bytes=None):
    """As-built figures of a workload (or of the kernels of it whose name contains one of `match`) per batch, from the
    committed counter passes, next to the live time.  Everything that is not live sits under `from_profile`.
      from_profile.hbm_bytes            (2 FETCH_SIZE + WRITE_SIZE) KiB of the kernels, per batch
      from_profile.ms / sclk_GHz        the kernels' own durations and cycles in the SQ pass of the profile run
      from_profile.frac_of_issue_ceiling  wave-level vector instructions x 4 cycles / 1024 SIMDs over those cycles
      from_profile.frac_of_copy_ceiling   hbm_bytes over from_profile.ms at the read+write stream ceiling of that box
      achieved_GBps (live)              from_profile.hbm_bytes over the LIVE milliseconds
      live_over_profile_ms              how far this box / run is from the profiled one
      bound                             the larger of the two fractions names it; both below 0.5: neither"""
    w = ((prof or {}).get("workloads") or {}).get(workload)
    if not w:
        return {"from_profile": None, "bound": "unknown (no committed counter pass for %s)" % workload}
    ks = {k: v for k, v in w["kernels"].items() if match is None or any(m in k for m in match)}
    if not ks:
        return {"from_profile": None, "bound": "unknown (kernels %r not in the profile of %s)" % (match, workload)}
    per = lambda v, f: v.get(f, 0.0) * v["per_batch"]
    ms = sum(per(v, "ms") for v in ks.values())
    cyc = sum(per(v, "cycles") for v in ks.values())
    by = sum(per(v, "hbm_bytes") for v in ks.values())
    insts = sum(per(v, "valu_wave_insts") for v in ks.values())
    busy = sum(v.get("valu_busy", 0.0) * per(v, "cycles") for v in ks.values())
    copy = prof.get("copy_ceiling_GBps") or 5230.0
    # share of the group's cycles that are not a counter of the dispatch but duration x an assumed clock (short dispatches,
    # tools/prof_all.sh `cycles_estimated`): the issue fractions of such a group are estimates and say so
    est = sum(per(v, "cycles") for v in ks.values() if v.get("cycles_estimated"))
    fp = {"dir": prof.get("dir"), "workload": workload, "kernels": sorted(ks), "ms": ms, "sclk_GHz": cyc / (ms * 1e-3) / 1e9 if ms else None,
          "cycles_estimated_share": est / cyc if cyc else None,
          "hbm_bytes": by, "lane_instructions": insts * 64,
          "valu_busy": busy / cyc if cyc else None,
          "frac_of_issue_ceiling": insts * 4 / SIMDS / cyc if cyc else None,
          "frac_of_copy_ceiling": by / (ms * 1e-3) / 1e9 / copy if ms else None,
          "copy_ceiling_GBps": copy}
    out = {"from_profile": fp, "achieved_GBps": by / (live_ms * 1e-3) / 1e9, "frac_of_hbm_peak_as_built": by / (live_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "live_ms": live_ms, "live_over_profile_ms": live_ms / ms if ms else None}
    if algorithmic_bytes:
        out["traffic_over_algorithmic"] = by / algorithmic_bytes
    fi, fc = fp["frac_of_issue_ceiling"] or 0.0, fp["frac_of_copy_ceiling"] or 0.0
    if max(fi, fc) < 0.5:
        out["bound"] = "neither ceiling (issue %.2f, copy %.2f): latency of dependent phases / launch size" % (fi, fc)
    else:
        out["bound"] = "valu (vector-ALU issue)" if fi > fc else "hbm (read+write stream ceiling)"
    out["frac_of_binding_ceiling"] = max(fi, fc)
    return out


# ------------------------------------------------------------------ the C4 workload on one device
class C4:
    """CKKS N=2^16, Q=16 {60,50x15} | P=1 {60}, depth 0: `B` ciphertext pairs of one rank resident on `dev`."""
    name, unit = "c4", "multiply+relinearize/s"

    def __init__(self, torch, hg, dev, first, B, ctx=None):
        from heongpu_amd import synth
        if ctx is None:
            ctx = hg.Context.from_bit_sizes(hg.CKKS, N, LOG_Q, LOG_P)
            ctx.upload()
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
        # pair g of the global batch has the inputs of distinct index g % UNIQ (seeds 1 + 10 u / 2 + 10 u)
        need_u = sorted({(first + b) % UNIQ for b in range(B)})
        self.uniq = len(need_u)
        a = {u: synth.synth_ct_t(torch, self.primes, range(l), 2, n, 1 + 10 * u, dev) for u in need_u}
        b = {u: synth.synth_ct_t(torch, self.primes, range(l), 2, n, 2 + 10 * u, dev) for u in need_u}
        self.ct1 = torch.cat([a[(first + i) % UNIQ] for i in range(B)])
        self.ct2 = torch.cat([b[(first + i) % UNIQ] for i in range(B)])
        del a, b
        self.out = torch.empty(B * self.out_elems, dtype=torch.int64, device=dev)
        self.ws = ctx.workspace(hg.OP_CKKS_RELIN, 0, B, device=dev)
        self.key = torch.empty(2 * self.Q * self.Qp * n, dtype=torch.int64, device=dev)

    @staticmethod
    def contexts(hg, world):
        ctx0 = hg.Context.from_bit_sizes(hg.CKKS, N, LOG_Q, LOG_P)
        return [ctx0] + [ctx0.clone() for _ in range(world - 1)]

    def replicated(self):
        return [self.key]

    def make_keys(self):
        from heongpu_amd import synth
        self.key.copy_(synth.synth_key_t(self.torch, self.primes, self.Q, self.Qp, self.n, 3, self.dev))

    def step(self, stream):
        c = self.ctx
        c.ckks_multiply(self.ct1, self.ct_elems, self.ct2, self.ct_elems, self.out, self.out_elems, 0, self.B, stream=stream)
        c.ckks_relinearize_inplace(self.out, self.out_elems, self.key, 0, self.B, self.ws, stream=stream)

    def twins(self):
        return twins_equal(self.torch, self.out, self.out_elems, self.B, UNIQ, self.first, used_elems=self.ct_elems)

    def output_digests(self):
        """(digest, present) per distinct input index: the first item of this rank's slice that has it"""
        d = row_digests(self.torch, self.out.view(self.B, self.out_elems)[:, :self.ct_elems])
        dig, present = [0] * UNIQ, [0] * UNIQ
        for b in range(self.B):
            u = (self.first + b) % UNIQ
            if not present[u]:
                dig[u], present[u] = d[b], 1
        return dig, present

    def describe(self, args, world):
        W, l = 8 * N, self.Q
        return ("homomorphic mults/sec (CKKS N=2^16, L=16) + NTT GB/s vs HBM roofline", {
            "workload": "CKKS N=2^16, Q=16 {60,50x15} | P=1 {60}, depth 0: multiply + relinearize_inplace "
                        "(key-switch method I), %d independent ciphertext pairs per GPU per step (global batch %d "
                        "sharded by contiguous slices), inputs resident in HBM" % (args.batch, args.batch * world),
            "poly_modulus_degree": N, "Q_size": l, "P_size": 1, "batch_per_gpu": args.batch,
            "global_batch": args.batch * world, "algorithmic_bytes_per_op": (6 * l * l + 32 * l + 8) * W})


def c4_host_inputs(primes, l, n, indices):
    """numpy copies of the pairs with the given distinct indices (the CPU checker's input)"""
    from heongpu_amd import synth
    a = np.concatenate([synth.synth_ct_np(primes, range(l), 2, n, 1 + 10 * u) for u in indices])
    b = np.concatenate([synth.synth_ct_np(primes, range(l), 2, n, 2 + 10 * u) for u in indices])
    return a, b


def c4_oracle_item(hg, work, b):
    """item b of this rank's slice against the CPU oracle (the checker): multiply + relinearize of its input pair"""
    from heongpu_amd import synth
    from oracle import binding as ob
    u = (work.first + b) % UNIQ
    l, n = work.Q, work.n
    c1, c2 = c4_host_inputs(work.primes, l, n, [u])
    key_h = synth.synth_key_np(work.primes, work.Q, work.Qp, n, 3)
    o = ob.OracleContext(ob.CKKS, 16, work.primes, work.Q, 1)
    want = o.ckks_multiply(c1, c2, 0)
    o.ckks_relinearize(want, key_h, 0)
    got = hg.to_host(work.out.view(work.B, work.out_elems)[b, :work.ct_elems])
    return bool(np.array_equal(got, want[:work.ct_elems]))


# ------------------------------------------------------------------ the C5 workload on one device
TFHE_SEED = 2025   # DRBG seed of the C5 key material and inputs (the secret key is re-derived from it on every rank)


class C5:
    """TFHE STD128 NAND gate bootstrap (pre-computation, blind rotate n=512, sample extraction, key switch): `S` gates of
    one rank resident on `dev`; gate b of the global batch has the inputs of distinct index (first + b) % TFHE_UNIQ.
    REAL key material from the backend's own key generator (hegpu_tfhe_generate_secret_key / _bootstrapping_key, seeded
    DRBG): a torus32 boot key with random polynomials in every slot (round 4's constant polynomials could not
    have shown a slot-order bug of the prepared layout), inputs are real encryptions of TFHE_UNIQ random bit pairs, so every
    output of the timed batch is also DECRYPTED and compared with NAND of its input bits."""
    name, unit = "c5", "gate bootstraps/s"

    def __init__(self, torch, hg, dev, first, S, ctx=None):
        self.torch, self.hg, self.dev, self.first, self.S = torch, hg, dev, first, S
        self.B = S
        with torch.cuda.device(dev):
            self.t = ctx if ctx is not None else hg.TfheContext()
            t = self.t
            U = TFHE_UNIQ
            # the same secret key on every rank (deterministic from the seed: the first two DRBG streams); the boot key and
            # the key-switch key -- the big, random part -- are made on rank 0 only and replicated
            self.rng = hg.Rng(TFHE_SEED)
            self.lwe, self.tlwe = t.generate_secret_key(self.rng)
            bits = np.random.default_rng(TFHE_SEED)
            self.x_u, self.y_u = bits.integers(0, 2, U), bits.integers(0, 2, U)
            mu = 1 << 29
            enc = lambda v: torch.from_numpy(np.where(v == 1, mu, -mu).astype(np.int32)).to(dev)
            in_rng = hg.Rng(TFHE_SEED + 1)   # inputs: their own streams, identical on every rank
            a1u, b1u = t.encrypt(in_rng, self.lwe, enc(self.x_u))
            a2u, b2u = t.encrypt(in_rng, self.lwe, enc(self.y_u))
            self.a1u, self.a2u = a1u.cpu().numpy(), a2u.cpu().numpy()
            self.b1u, self.b2u = b1u.cpu().numpy(), b2u.cpu().numpy()
            idx = (first + np.arange(S)) % U
            self.idx = idx
            it = torch.from_numpy(idx).to(dev)
            self.a1, self.a2 = a1u.view(U, 512)[it].reshape(-1).contiguous(), a2u.view(U, 512)[it].reshape(-1).contiguous()
            self.b1, self.b2 = b1u[it].contiguous(), b2u[it].contiguous()
            self.prepared = torch.empty(t.int("prepared_bootkey_elems"), dtype=torch.int64, device=dev)
            self.ks_a = torch.empty(t.int("kskey_a_elems"), dtype=torch.int32, device=dev)
            self.ks_b = torch.empty(t.int("kskey_b_elems"), dtype=torch.int32, device=dev)
            self.out_a = torch.empty(S * 512, dtype=torch.int32, device=dev)
            self.out_b = torch.empty(S, dtype=torch.int32, device=dev)
            self.ws = torch.empty((512 + 1024 + 2) * S, dtype=torch.int32, device=dev)
            self.bk_host = None   # reference-layout boot key on the host (rank 0, for the CPU checker)

    @staticmethod
    def contexts(hg, world):
        return [hg.TfheContext() for _ in range(world)]

    def replicated(self):
        return [self.prepared, self.ks_a, self.ks_b]

    def make_keys(self):
        torch, t = self.torch, self.t
        with torch.cuda.device(self.dev):
            bk, ks_a, ks_b = t.generate_bootstrapping_key(self.rng, self.lwe, self.tlwe)
            self.prepared.copy_(t.prepare_bootkey(bk))
            self.ks_a.copy_(ks_a)
            self.ks_b.copy_(ks_b)
            self.bk_host = bk.cpu().numpy().view(np.uint64)
            self.ks_a_h, self.ks_b_h = ks_a.cpu().numpy(), ks_b.cpu().numpy()

    def decrypt_check(self):
        """every gate of the timed batch decrypted with the secret key: NAND of its two input bits?"""
        torch = self.torch
        with torch.cuda.device(self.dev):
            got = (self.t.decrypt_phase(self.lwe, self.out_a, self.out_b).cpu().numpy() > 0).astype(np.int64)
        want = 1 - (self.x_u[self.idx] & self.y_u[self.idx])
        return int(self.S), bool(np.array_equal(got, want))

    def step(self, stream):
        self.t.gate(self.hg.GATE_NAND, self.a1, self.b1, self.a2, self.b2, self.out_a, self.out_b, self.prepared, self.ks_a,
                    self.ks_b, self.S, self.ws, stream=stream)

    def twins(self):
        """every gate against the first gate of this rank's slice with the same inputs"""
        torch = self.torch
        first_pos = {}
        for b, u in enumerate(self.idx):
            first_pos.setdefault(int(u), b)
        ref = torch.tensor([first_pos[int(u)] for u in self.idx], device=self.dev)
        ga, gb = self.out_a.view(self.S, 512), self.out_b
        ok = bool(torch.equal(ga, ga[ref])) and bool(torch.equal(gb, gb[ref]))
        return self.S - len(first_pos), ok

    def output_digests(self):
        """(digest, present) per distinct input index: the first gate of this rank's slice that has it"""
        torch = self.torch
        rows = torch.cat([self.out_a.view(self.S, 512), self.out_b.view(self.S, 1)], dim=1)
        first_pos = {}
        for b, u in enumerate(self.idx):
            first_pos.setdefault(int(u), b)
        us = sorted(first_pos)
        d = row_digests(torch, rows[torch.tensor([first_pos[u] for u in us], device=self.dev)])
        dig, present = [0] * TFHE_UNIQ, [0] * TFHE_UNIQ
        for u, v in zip(us, d):
            dig[u], present[u] = v, 1
        return dig, present

    def oracle_check(self, count):
        """the first `count` gates of this rank's slice against the CPU oracle (the checker; not timed here)"""
        from oracle import binding as ob
        ot = ob.OracleTfhe()
        sel = self.idx[:count]
        a1 = self.a1u.reshape(-1, 512)[sel].reshape(-1)
        a2 = self.a2u.reshape(-1, 512)[sel].reshape(-1)
        t0 = time.perf_counter()
        want_a, want_b = ot.gate(self.hg.GATE_NAND, a1, self.b1u[sel], a2, self.b2u[sel], self.bk_host, self.ks_a_h,
                                 self.ks_b_h)
        cpu_s = time.perf_counter() - t0
        ga = self.out_a.view(self.S, 512)[:count].cpu().numpy().reshape(-1)
        ok = np.array_equal(ga, want_a) and np.array_equal(self.out_b[:count].cpu().numpy(), want_b)
        return bool(ok), cpu_s

    def describe(self, args, world):
        return ("TFHE STD128 gate bootstraps/sec (NAND, blind-rotate PBS, n=512, N=1024)", {
            "workload": "TFHE STD128 NAND gate bootstrap (pre-computation, blind rotate n=512, sample extraction, key switch), "
                        "%d concurrent gates per GPU per step (global batch %d sharded along `shape` by contiguous slices, %d "
                        "distinct encrypted bit pairs), generated torus32 boot key (FP64 blind rotate), keys resident in HBM"
                        % (args.batch, args.batch * world, TFHE_UNIQ),
            "gates_per_gpu": args.batch, "global_batch": args.batch * world,
            "replicated_bytes": sum(t.numel() * t.element_size() for t in self.replicated())})


WORKLOADS = {"c4": C4, "c5": C5}


def line_skeleton(args, work, world, value, elapsed, per_rank, parallelism, distinct_devices):
    """`n_gpus` is the number of DISTINCT devices the ranks ran on; `ranks` the number of ranks (they differ only under
    --allow-shared-devices, where the line is a functional check and not an N-GPU point)"""
    metric, config = work.describe(args, world)
    config["parallelism"] = parallelism
    return {
        "metric": metric, "value": value, "unit": work.unit, "n_gpus": distinct_devices, "ranks": world,
        "distinct_devices": distinct_devices, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64" if work.name == "c4" else "f64 (exact integer arithmetic modulo a 44-bit prime) / i32 torus",
        "data": "synthetic", "config": config, "per_rank_units_per_s": per_rank,
    }


# ------------------------------------------------------------------ what goes to stdout: one compact contract line
COMPACT_LIMIT = 4096   # bytes; the driver keeps the last 8 KB of stdout and parses its last line (a 21 KB line was lost)
DETAIL_DEFAULT = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _r(v, sig=5):
    """floats to `sig` significant digits (the compact line is for reading and for the driver's checks, the detail file
    keeps every digit)"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float("%.*g" % (sig, v))


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and d.get(k) is not None}


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


# per secondary workload: (key of the one rate that goes into the compact line, where its as-built block lives)
SECONDARY_RATE = {"bfv_n14_multiply": ("multiplications_per_s", "as_built"), "c3_bfv_n15_rotate": ("rotations_per_s", "as_built"),
                  "c2_ckks_n14": ("ops_per_s_batch64", "as_built_batch64"), "ckks_n16_method_II": ("multiply_relinearize_per_s", "as_built"),
                  "c5_tfhe_gates": ("gates_per_s", "as_built")}


def _all_true(chk, key):
    """checked_items blocks are flat or one level deep (per batch size): every `key` in there is true"""
    if not isinstance(chk, dict):
        return None
    if key in chk:
        return bool(chk[key])
    vals = [v[key] for v in chk.values() if isinstance(v, dict) and key in v]
    return all(vals) if vals else None


def compact_line(d, detail_path=None):
    """The contract line (task statement + VERDICT r4 item 1) from the full detail of a run: the headline, `roofline`,
    `cpu_baseline`, `checked_items` and ONE rate + binding ceiling + verification flags per secondary workload.  Kernel
    lists, replayed counters and per-group accounting stay in the detail file and in profiles/profile.json."""
    c = {k: _r(d[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data") if k in d}
    for k in ("ranks", "distinct_devices", "key_broadcast_ms", "key_bytes"):
        if k in d:
            c[k] = _r(d[k])
    cfg = d.get("config") or {}
    c["config"] = {k: (_clip(v, 260) if k in ("workload", "parallelism") else v) for k, v in cfg.items()}
    if "per_rank_units_per_s" in d and len(d["per_rank_units_per_s"]) > 1:
        c["per_rank_units_per_s"] = [_r(v, 4) for v in d["per_rank_units_per_s"]]
    rf = d.get("roofline")
    if rf:
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_bytes_per_launch"))
        r["traffic_source"] = _clip(rf.get("traffic_source"), 90)
        pair = rf.get("launch_pair") or {}
        r["traffic_over_algorithmic"] = _r(pair.get("traffic_over_algorithmic"))
        r["frac_of_copy_ceiling"] = _r(((pair.get("from_profile") or {}).get("frac_of_copy_ceiling")))
        st = rf.get("step") or {}
        r["step"] = _pick(st, ("achieved_GBps", "frac_of_hbm_peak_as_built", "bound"))
        r["in_step_dominant"] = _pick(rf.get("in_step_dominant") or {}, ("launches", "ms", "bound", "frac_of_binding_ceiling"))
        r["profile_dir"] = ((pair.get("from_profile") or st.get("from_profile") or {}).get("dir"))
        c["roofline"] = r
    if "as_built" in d:  # the C5 headline
        ab = d["as_built"]
        c["as_built"] = _pick(ab, ("bound", "frac_of_binding_ceiling", "achieved_GBps", "live_over_profile_ms"))
    if "ntt" in d:
        c["ntt"] = _pick(d["ntt"], ("forward_GBps", "inverse_GBps", "n", "limbs"))
    if "ntt_by_degree" in d:
        c["ntt_forward_frac_by_degree"] = {k: _r(v.get("forward_frac"), 3) for k, v in d["ntt_by_degree"].items()}
    if d.get("power"):
        c["power"] = _pick(d["power"], ("package_w", "package_limit_w", "sclk_mhz"))
    if "checked_items" in d:
        c["checked_items"] = {k: v for k, v in d["checked_items"].items() if k != "note"}
    sec = {}
    for name, e in (d.get("secondary") or {}).items():
        if name == "hoisted_rotations":
            k8 = (e.get("by_k") or {}).get("8") or {}
            sec[name] = {"hoisted_rotations_per_s_k8": _r(k8.get("hoisted_rotations_per_s")),
                         "separate_rotations_per_s_k8": _r(k8.get("separate_rotations_per_s")),
                         "hoisted_equals_separate": all(v.get("hoisted_equals_separate") for v in (e.get("by_k") or {}).values())}
            continue
        rate_key, ab_key = SECONDARY_RATE.get(name, (None, "as_built"))
        ab = e.get(ab_key) or {}
        s = {rate_key: _r(e.get(rate_key))} if rate_key else {}
        if name == "c2_ckks_n14":
            s["latency_us_batch1"] = _r(e.get("latency_us_batch1"))
        s["bound"] = _clip(ab.get("bound"), 48)
        s["frac_of_binding_ceiling"] = _r(ab.get("frac_of_binding_ceiling"), 3)
        s["oracle_equal"] = _all_true(e.get("checked_items"), "oracle_equal")
        s["twins_equal"] = _all_true(e.get("checked_items"), "twins_equal")
        if _all_true(e.get("checked_items"), "decrypt_equal") is not None:
            s["decrypt_equal"] = _all_true(e.get("checked_items"), "decrypt_equal")
        sec[name] = s
    if sec:
        c["secondary"] = sec
    cb = d.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "threads_busy", "kind", "gpu_matches_cpu_bit_exact", "gpu_items_compared"))
        c["cpu_baseline"]["sample"] = _clip(cb.get("sample"), 200)
    if detail_path:
        c["detail"] = os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path
    text = json.dumps(c)
    if len(text) > COMPACT_LIMIT:  # never again a line the driver cannot hold: drop what is optional, keep the contract
        for k in ("secondary", "ntt_forward_frac_by_degree", "power", "ntt", "per_rank_units_per_s"):
            c.pop(k, None)
            text = json.dumps(c)
            if len(text) <= COMPACT_LIMIT:
                break
    return text


def emit(line, args):
    """Full detail to a file (merged back from the GPU box under gpurun_out/), the compact contract line -- and nothing
    else -- as the LAST line of stdout."""
    path = args.detail or DETAIL_DEFAULT
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(line, f, indent=1)
    except OSError as e:   # a read-only tree must not cost the line
        print("bench.py: could not write the detail file %s: %s" % (path, e), file=sys.stderr)
        path = None
    sys.stdout.flush()
    print(compact_line(line, path), flush=True)
    mark_ran()


# ------------------------------------------------------------------ which path a command line takes (pure; CPU-testable)
def plan_run(gpus, env, device_count, single_process=False, allow_shared=False):
    """What `bench.py --gpus N` does given the launcher's environment and the visible devices -- no side effects.
      mode: "rank" (this process is one rank: plain N=1, or under torch.distributed.run), "self_launch" (start N ranks),
            "single_process" (one thread per device), "refuse" (with `why`)
      full_line: the rank-0 process also runs the roofline / secondary / cpu_baseline legs (N=1 only -- with or without a launcher)
      distinct_devices / n_gpus: ranks beyond the visible devices share devices; the line reports the DISTINCT devices as
      n_gpus and the rank count as `ranks`, and only with --allow-shared-devices."""
    under = "WORLD_SIZE" in env
    world = int(env.get("WORLD_SIZE", "1")) if under else (1 if gpus == 1 else gpus)
    rank = int(env.get("RANK", "0")) if under else 0
    local = int(env.get("LOCAL_RANK", "0")) if under else 0
    p = {"mode": "rank", "world": world, "rank": rank, "under_launcher": under, "why": None}
    if under and world != gpus:
        return dict(p, mode="refuse", why="world size %d (WORLD_SIZE) does not match --gpus %d" % (world, gpus))
    if device_count is not None:
        if device_count < 1:
            return dict(p, mode="refuse", why="bench.py needs a HIP device (there is no CPU fallback)")
        distinct = min(gpus, device_count)
        if distinct < gpus and not allow_shared:
            return dict(p, mode="refuse", why="--gpus %d but %d visible device(s): ranks would share devices and the line would not be a "
                                              "%d-GPU point; pass --allow-shared-devices for a functional check" % (gpus, device_count, gpus))
        p.update(distinct_devices=distinct, n_gpus=distinct, ranks=gpus, dev_index=local % device_count)
    if not under and gpus > 1:
        p["mode"] = "single_process" if single_process else "self_launch"
    p["full_line"] = (world == 1 and rank == 0)
    return p


def preflight(args):
    """`--preflight`: what a multi-GPU run would find on this node, without timing anything."""
    import torch
    rep = {"preflight": True, "gpus_requested": args.gpus, "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    rep["visible_devices"] = ndev
    rep["device_names"] = [torch.cuda.get_device_name(i) for i in range(ndev)]
    rep["peer_access"] = [[(i == j) or bool(torch.cuda.can_device_access_peer(i, j)) for j in range(ndev)] for i in range(ndev)]
    try:
        rep["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        rep["rccl_version"] = "unavailable (%s)" % str(e)[:80]
    import torch.distributed as dist
    rep["torch_distributed_backends"] = {"nccl": bool(dist.is_nccl_available()), "gloo": bool(dist.is_gloo_available())}
    plan = plan_run(args.gpus, {}, ndev, args.single_process, args.allow_shared_devices)
    rep["plan"] = plan
    if plan["mode"] == "refuse":
        rep["tier"] = "none: " + plan["why"]
    elif args.gpus == 1:
        rep["tier"] = "one rank, no replication"
    elif plan["mode"] == "single_process":
        all_peer = all(all(r) for r in rep["peer_access"])
        rep["tier"] = "3: one process, a thread per device, hegpu_broadcast_bytes (%s)" % ("flat peer fan-out" if all_peer else "chunked tree / staged edges")
    else:
        rep["tier"] = ("1: one process per GPU, RCCL broadcast of the keys over xGMI (falls to tier 2, gloo host staging, if the RCCL "
                       "probe all-reduce does not answer within %d s; to tier 3 if the launch itself fails)" % NCCL_TIMEOUT_S)
        if plan.get("distinct_devices", args.gpus) < args.gpus:
            rep["tier"] += "; ranks share devices: RCCL refuses two ranks on one device (ncclInvalidUsage) -> tier 2"
    print(json.dumps(rep))
    return 0


# ------------------------------------------------------------------ one process, N devices (fallback 3)
def run_single_process(args, reason):
    import torch

    import heongpu_amd as hg
    from heongpu_amd import sharding
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    W = WORKLOADS[args.workload]
    world = args.gpus
    ndev = torch.cuda.device_count()
    plan = plan_run(args.gpus, {}, ndev, True, args.allow_shared_devices)
    if plan["mode"] == "refuse":
        raise SystemExit("bench.py: " + plan["why"])
    devs = [torch.device("cuda", r % ndev) for r in range(world)]
    ctxs = W.contexts(hg, world)
    works, streams = [], []
    for r in range(world):
        torch.cuda.set_device(devs[r])
        if args.workload == "c4":
            ctxs[r].upload_device(devs[r].index)
        first, B = sharding.shard_range(args.batch * world, world, r)
        works.append(W(torch, hg, devs[r], first, B, ctx=ctxs[r]))
        streams.append(torch.cuda.Stream(device=devs[r]))
    torch.cuda.set_device(devs[0])
    with torch.cuda.stream(streams[0]):
        works[0].make_keys()
    t0 = time.perf_counter()
    paths = []
    for i in range(len(works[0].replicated())):
        bufs = [w.replicated()[i] for w in works]
        paths.append(hg.broadcast_bytes([d.index for d in devs], bufs, bufs[0].numel() * bufs[0].element_size(),
                                        [s.cuda_stream for s in streams]))
    for s in streams:
        s.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3
    if 0 < args.selftest_corrupt_rank < world:   # testing: a replica that arrived with one bit wrong
        with torch.cuda.device(devs[args.selftest_corrupt_rank]):
            works[args.selftest_corrupt_rank].replicated()[0].view(-1)[12345] ^= 1
            torch.cuda.synchronize()
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
    line = line_skeleton(args, works[0], world, value, elapsed, [w.B * args.steps / o for w, o in zip(works, own)],
                         "sharded x%d in ONE process: a thread, a stream and a context per device, keys replicated with "
                         "hegpu_broadcast_bytes -- %s; fallback because %s; %d visible device(s)"
                         % (world, hg.broadcast_path_name(paths[0]), reason, ndev), plan["distinct_devices"])
    line["key_broadcast_ms"] = bcast_ms
    line["key_broadcast_path"] = paths
    line["key_bytes"] = sum(t.numel() * t.element_size() for t in works[0].replicated())
    tw = [w.twins() for w in works]
    same_key = all(bool(torch.equal(a.to(w.dev), b)) for w in works[1:] for a, b in zip(works[0].replicated(), w.replicated()))
    # the same self-proof as the one-process-per-GPU tiers (run_rank): repeat, key digests, cross-device outputs
    digs, repeat_ok, key_digs = [], True, []
    for r, w in enumerate(works):
        with torch.cuda.device(devs[r]), torch.cuda.stream(streams[r]):
            d0 = w.output_digests()
            w.step(streams[r].cuda_stream)
            streams[r].synchronize()
            repeat_ok &= w.output_digests()[0] == d0[0]
            digs.append(d0)
            key_digs.append([digest64(torch, t) for t in w.replicated()])
    xc = cross_check(key_digs, [d[0] for d in digs], [d[1] for d in digs])
    line["checked_items"] = {"distinct_inputs": xc["distinct_inputs"], "twin_compared": sum(t[0] for t in tw),
                             "twins_equal": all(t[1] for t in tw), "repeat_equal": repeat_ok,
                             "key_replicas_equal": same_key, "key_digest_equal": xc["key_digest_equal"],
                             "cross_rank_equal": xc["cross_rank_equal"], "oracle_compared": 0,
                             "note": "multi-GPU run: the oracle comparison is part of the N=1 line"}
    if args.workload == "c4" and not args.no_cpu_baseline:
        got = []
        for r, w in enumerate(works):
            with torch.cuda.device(devs[r]):
                got.append(c4_oracle_item(hg, w, r % w.B))
        line["checked_items"].update(oracle_compared=len(got), oracle_equal=all(got),
                                     note="one item per device against the CPU oracle; every other output through cross_rank_equal")
    dec_ok = True
    if hasattr(works[0], "decrypt_check"):
        dec = [w.decrypt_check() for w in works]
        dec_ok = all(d[1] for d in dec)
        line["checked_items"].update(decrypt_compared=sum(d[0] for d in dec), decrypt_equal=dec_ok)
    emit(line, args)
    chk = line["checked_items"]
    return 0 if (all(t[1] for t in tw) and same_key and dec_ok and repeat_ok and chk["key_digest_equal"] and
                 chk["cross_rank_equal"] and chk.get("oracle_equal", True)) else 1


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


def check_line(twins, oracle_ok, n_oracle):
    return {"oracle_compared": n_oracle, "oracle_equal": bool(oracle_ok), "twin_compared": twins[0],
            "twins_equal": bool(twins[1])}


class Sec:
    """One secondary workload: `runs` = {profile-workload name: (callable, units per call)}, `entry(ms)` builds its part of
    the line (timing LIVE, verification of the timed batch against the oracle, the as-built figures from the profile)."""

    def __init__(self, runs, entry, close):
        self.runs, self.entry, self.close = runs, entry, close


U_SEC = 2  # distinct inputs per secondary workload, all of them compared with the oracle


def sec_bfv14(torch, hg, prof):
    """north_star target 2: BFV N=2^14, default 128-bit chain (Q=8, P=1), 256 pairs: multiply"""
    from heongpu_amd import synth
    stream, dev, U = torch.cuda.current_stream().cuda_stream, torch.device("cuda"), U_SEC
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
    mul = lambda: ctx.bfv_multiply(ct, 2 * Q * n, ct2, 2 * Q * n, o3, 3 * Q * n, B, wsm, stream=stream)

    def entry(timer):
        from oracle import binding as ob  # the checker; nothing timed goes through it
        m = timer.ms(mul, 3)
        o = ob.OracleContext(ob.BFV, 14, primes, Q, 1, t)
        got = hg.to_host(o3[:U * 3 * Q * n]).reshape(U, -1)
        ok = all(np.array_equal(got[u], o.bfv_multiply(hg.to_host(a_u[u]), hg.to_host(b_u[u]))) for u in range(U))
        chk = check_line(twins_equal(torch, o3, 3 * Q * n, B, U), ok, U)
        r = timer.ms(lambda: ctx.bfv_relinearize_inplace(o3, 3 * Q * n, key, B, wsr, stream=stream), 3)
        mul_bytes = (28 * L + 7 * Q) * W
        e = {"workload": "BFV N=2^14 default 128-bit chain (Q=%d, P=1, Bsk=%d), t=786433, %d pairs resident in HBM" % (Q, L - Q, B),
             "multiplications_per_s": B / (m * 1e-3), "ms_per_batch": m, "multiply_relinearize_per_s": B / ((m + r) * 1e-3),
             "reference_sequence_bytes_per_op": mul_bytes,
             "reference_sequence_bytes_rate_GBps": mul_bytes * B / (m * 1e-3) / 1e9,
             "as_built": prof_group(prof, "bfv_n14_multiply", m), "checked_items": chk}
        return "bfv_n14_multiply", e
    return Sec({"bfv_n14_multiply": (mul, B)}, entry, ctx.close)


def sec_c3(torch, hg, prof):
    """C3: BFV N=2^15 default chain (Q=14, P=1), rotate_rows by one step, 64 ciphertexts"""
    from heongpu_amd import synth
    stream, dev, U = torch.cuda.current_stream().cuda_stream, torch.device("cuda"), U_SEC
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
    rot = lambda: ctx.bfv_apply_galois(ct, 2 * Q * n, out, 2 * Q * n, key, gal, B, ws, stream=stream)

    def entry(timer):
        from oracle import binding as ob
        g = timer.ms(rot, 3)
        o = ob.OracleContext(ob.BFV, 15, primes, Q, 1, t)
        got = hg.to_host(out[:U * 2 * Q * n]).reshape(U, -1)
        key_h = hg.to_host(key)
        ok = all(np.array_equal(got[u], o.bfv_apply_galois(hg.to_host(a_u[u]), key_h, gal)) for u in range(U))
        rot_bytes = (6 * Q * Qp + 6 * Q + 8 * Qp) * W
        e = {"workload": "BFV N=2^15 default chain (Q=%d, P=1), rotate_rows (Galois key switch method I), %d ciphertexts" % (Q, B),
             "rotations_per_s": B / (g * 1e-3), "ms_per_batch": g, "reference_sequence_bytes_per_op": rot_bytes,
             "reference_sequence_bytes_rate_GBps": rot_bytes * B / (g * 1e-3) / 1e9,
             "as_built": prof_group(prof, "c3_bfv_n15_rotate", g),
             "checked_items": check_line(twins_equal(torch, out, 2 * Q * n, B, U), ok, U)}
        return "c3_bfv_n15_rotate", e
    return Sec({"c3_bfv_n15_rotate": (rot, B)}, entry, ctx.close)


def sec_c2(torch, hg, prof):
    """C2: CKKS N=2^14, {50, 40 x 7} | {50}, multiply + relinearize + rescale, one ciphertext and 64"""
    from heongpu_amd import synth
    stream, dev, U = torch.cuda.current_stream().cuda_stream, torch.device("cuda"), U_SEC
    n = 1 << 14
    ctx = hg.Context.from_bit_sizes(hg.CKKS, n, [50] + [40] * 7, [50])
    ctx.upload()
    primes = [int(v) for v in ctx.table("modulus")]
    Q, Qp = ctx.Q_size, ctx.Q_prime_size
    W = 8 * n
    key = synth.synth_key_t(torch, primes, Q, Qp, n, 3, dev)
    a_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 1 + 10 * u, dev) for u in range(U)]
    b_u = [synth.synth_ct_t(torch, primes, range(Q), 2, n, 2 + 10 * u, dev) for u in range(U)]
    seqs, bufs = {}, {}
    for B in (1, 64):
        c1b, c2b = tile_items(torch, a_u, B), tile_items(torch, b_u, B)
        ob_ = torch.empty(3 * Q * n * B, dtype=torch.int64, device="cuda")
        wsb, wsb2 = ctx.workspace(hg.OP_CKKS_RELIN, 0, B), ctx.workspace(hg.OP_CKKS_RESCALE, 0, B)

        def seq(s=None, B=B, c1b=c1b, c2b=c2b, ob_=ob_, wsb=wsb, wsb2=wsb2):
            s = stream if s is None else s
            ctx.ckks_multiply(c1b, 2 * Q * n, c2b, 2 * Q * n, ob_, 3 * Q * n, 0, B, stream=s)
            ctx.ckks_relinearize_inplace(ob_, 3 * Q * n, key, 0, B, wsb, stream=s)
            ctx.ckks_rescale_inplace(ob_, 3 * Q * n, 0, B, wsb2, stream=s)
        seqs[B], bufs[B] = seq, ob_

    def entry(timer):
        from oracle import binding as ob
        key_h = hg.to_host(key)
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
            c2[B] = timer.ms(seqs[B], 5)
            nu = min(U, B)
            got = hg.to_host(bufs[B][:nu * 3 * Q * n]).reshape(nu, -1)
            ok = all(np.array_equal(got[u][:2 * (Q - 1) * n], want[u]) for u in range(nu))
            c2chk["batch%d" % B] = check_line(twins_equal(torch, bufs[B], 3 * Q * n, B, U, used_elems=2 * (Q - 1) * n), ok, nu)
        # the same sequence captured once and replayed as one hipGraph launch (the operator entries allocate nothing
        # and never synchronise): the launch-bound batch-1 case
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            seqs[1](torch.cuda.current_stream().cuda_stream)
        c2["graph"] = timer.ms(graph.replay, 5)
        op_bytes = (6 * Q * Q + 32 * Q + 8 + 6 + 16 * (Q - 1)) * W
        e = {"workload": "CKKS N=2^14, Q=8 {50,40x7} | P=1 {50}, multiply + relinearize + rescale",
             "latency_us_batch1": c2[1] * 1e3, "latency_us_batch1_hipgraph_replay": c2["graph"] * 1e3,
             "ops_per_s_batch64": 64 / (c2[64] * 1e-3), "reference_sequence_bytes_per_op": op_bytes,
             "reference_sequence_bytes_rate_GBps_batch64": op_bytes * 64 / (c2[64] * 1e-3) / 1e9,
             "as_built_batch1": prof_group(prof, "c2_ckks_n14_b1", c2[1]),
             "as_built_batch64": prof_group(prof, "c2_ckks_n14_b64", c2[64]), "checked_items": c2chk}
        return "c2_ckks_n14", e
    return Sec({"c2_ckks_n14_b1": (seqs[1], 1), "c2_ckks_n14_b64": (seqs[64], 64)}, entry, ctx.close)


def sec_m2(torch, hg, prof):
    """key-switching method II (selected by the reference whenever P_size > 1): the C4 shape with four special primes,
    Q = 16 x 50 bits | P = 4 x 50 bits (d = 4 digits of 4 primes), multiply + relinearize, 64 pairs"""
    from heongpu_amd import synth
    stream, dev, U = torch.cuda.current_stream().cuda_stream, torch.device("cuda"), U_SEC
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

    def entry(timer):
        from oracle import binding as ob
        m2 = timer.ms(seq2, 3)
        o = ob.OracleContext(ob.CKKS, 16, primes, Q, 4)
        key_h = hg.to_host(key)
        got = hg.to_host(ob_[:U * 3 * Q * n]).reshape(U, -1)
        ok = True
        for u in range(U):
            w3 = o.ckks_multiply(hg.to_host(a_u[u]), hg.to_host(b_u[u]), 0)
            o.ckks_relinearize_II(w3, key_h, 0)
            ok = ok and np.array_equal(got[u][:2 * Q * n], w3[:2 * Q * n])
        e = {"workload": "CKKS N=2^16, Q=16 x 50 bits | P=4 x 50 bits (hybrid key switching, 4 digits), multiply + relinearize, "
                         "%d pairs" % B,
             "multiply_relinearize_per_s": B / (m2 * 1e-3), "ms_per_batch": m2,
             "as_built": prof_group(prof, "ckks_n16_method_II", m2),
             "checked_items": check_line(twins_equal(torch, ob_, 3 * Q * n, B, U, used_elems=2 * Q * n), ok, U)}
        return "ckks_n16_method_II", e
    return Sec({"ckks_n16_method_II": (seq2, B)}, entry, ctx.close)


def sec_c5(torch, hg, prof):
    """C5: TFHE STD128 NAND gate bootstrap, 8192 concurrent gates (the whole config on one GPU; its 8-GPU share is 1024)"""
    S, TCHK = 8192, 2
    work = C5(torch, hg, torch.device("cuda", torch.cuda.current_device()), 0, S)
    work.make_keys()
    stream = torch.cuda.current_stream().cuda_stream
    run = lambda: work.step(stream)

    def entry(timer):
        g = timer.ms(run, 2)
        tw = work.twins()
        ok, _ = work.oracle_check(TCHK)
        dn, dok = work.decrypt_check()
        ab = prof_group(prof, "c5_tfhe_gates", g)
        ab["blind_rotate"] = prof_group(prof, "c5_tfhe_gates", g, match=["k_tfhe_blind_rotate_fp"])
        ab["blind_rotate"].pop("achieved_GBps", None)  # the live time is the whole gate's, not this kernel's
        e = {"workload": "TFHE STD128 NAND gate bootstrap (pre-computation, blind rotate n=512, sample extraction, key switch), "
                         "%d concurrent gates (%d distinct encrypted bit pairs), generated torus32 boot key (FP64 blind rotate)" % (S, TFHE_UNIQ),
             "gates_per_s": S / (g * 1e-3), "ms_per_batch": g,
             "reference_sequence_bytes_per_gate": 72 * (1 << 20),
             "reference_sequence_bytes_rate_GBps": 72 * (1 << 20) * S / (g * 1e-3) / 1e9,
             "note": "the accumulator of a gate never leaves LDS: the reference-sequence rate is the rate the reference's "
                     "1024 launches per gate would have to move THEIR 72 MiB at, not bytes this kernel moves",
             "as_built": ab, "checked_items": dict(check_line(tw, ok, TCHK), decrypt_compared=dn, decrypt_equal=dok)}
        return "c5_tfhe_gates", e
    return Sec({"c5_tfhe_gates": (run, S)}, entry, work.t.close)


SECONDARY = (sec_bfv14, sec_c3, sec_c2, sec_m2, sec_c5)


def secondary_block(torch, hg, timer, prof):
    """The other BASELINE.json configurations, synthetic data: each timed LIVE, its timed batch checked (`U_SEC` distinct
    items against the CPU oracle, every other item against its twin), and -- instead of round 3's reference-bytes
    "fraction of HBM peak", which is no efficiency for a fused path -- the bytes and vector instructions it moves and
    issues AS BUILT (committed counter passes, `from_profile`) with the ceiling that binds it."""
    sec = {}
    for make in SECONDARY:
        s = make(torch, hg, prof)
        name, e = s.entry(timer)
        sec[name] = e
        s.close()
        del s
        torch.cuda.empty_cache()
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


# ------------------------------------------------------------------ counter passes: one workload, nothing else
def run_profile_workload(args):
    """`--profile-workload NAME --reps R` (tools/prof_all.sh wraps this in rocprofv3): set the workload up exactly as the
    line does, run it R times.  The JSON it prints tells tools/build_profile_json.py how many batches the dispatch
    counts of the trace belong to (kernels that ran fewer times than that are set-up, not workload)."""
    import torch

    import heongpu_amd as hg
    name = args.profile_workload
    stream = torch.cuda.current_stream().cuda_stream
    units = None
    if name in ("c4_step", "ntt_pair"):
        work = C4(torch, hg, torch.device("cuda", 0), 0, args.batch)
        work.make_keys()
        if name == "c4_step":
            run, units = (lambda: work.step(stream)), work.B
        else:
            polys = work.B * work.Q * work.Qp
            run, units = (lambda: work.ctx.ntt(work.ws, work.ws, False, polys, work.Qp, stream=stream)), polys
    else:
        makers = {"bfv_n14_multiply": sec_bfv14, "c3_bfv_n15_rotate": sec_c3, "c2_ckks_n14_b1": sec_c2, "c2_ckks_n14_b64": sec_c2,
                  "ckks_n16_method_II": sec_m2, "c5_tfhe_gates": sec_c5}
        run, units = makers[name](torch, hg, None).runs[name]
    torch.cuda.synchronize()
    for _ in range(args.reps):
        run()
    torch.cuda.synchronize()
    print(json.dumps({"profile_workload": name, "batches": args.reps, "units_per_batch": units}))
    return 0


# ------------------------------------------------------------------ one rank (N = 1, or one process per GPU)
def run_rank(args):
    import torch

    import heongpu_amd as hg
    from heongpu_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    plan = plan_run(args.gpus, os.environ, torch.cuda.device_count(), False, args.allow_shared_devices)
    if plan["mode"] == "refuse":
        raise SystemExit("bench.py: " + plan["why"])
    world, rank = plan["world"], plan["rank"]
    # one process per GPU; with --allow-shared-devices on a box with fewer devices than ranks, ranks share devices
    dev_index = plan["dev_index"]
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = Dist(torch, world, rank, args.backend)

    first, B = sharding.shard_range(args.batch * world, world, rank)
    work = WORKLOADS[args.workload](torch, hg, dev, first, B)
    if rank == 0:
        work.make_keys()   # produced on the device (C4: 272 MiB; C5: 64 + 48 MiB)
    bcast_ms = dist.broadcast_keys(work.replicated(), dev)
    if args.selftest_corrupt_rank == rank and world > 1:   # testing: a replica that arrived with one bit wrong
        work.replicated()[0].view(-1)[12345] ^= 1
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
    # ---- the line proves itself: (a) the same step once more gives the same bits (determinism without twins: all inputs
    # of a slice are distinct), (b) every rank holds the key bits rank 0 made, (c) ranks that computed on the same input
    # produced the same output.  64-bit digests, gathered over gloo; a mismatch makes the run fail.
    out_dig, out_present = work.output_digests()
    repeat_ok = True
    if not args.step_only:   # (counter passes: every dispatch belongs to a timed step)
        step()
        torch.cuda.synchronize()
        repeat_ok = work.output_digests()[0] == out_dig
    key_dig = [digest64(torch, t) for t in work.replicated()]
    xc = cross_check(dist.gather_i64(key_dig), dist.gather_i64(out_dig), dist.gather_i64(out_present))
    rep_all = dist.gather_floats(1.0 if repeat_ok else 0.0)
    dec_all = None
    if hasattr(work, "decrypt_check"):   # C5: every output decrypted with the secret key, on every rank
        dn, dok = work.decrypt_check()
        dec_all = dist.gather_floats(float(dn if dok else -1))
    value = args.batch * world * args.steps / elapsed

    parallelism = "sharded x%d, one process per GPU (gloo control plane), key broadcast: %s; no data-path collective" \
        % (world, dist.key_path)
    if plan["distinct_devices"] < world:
        parallelism += "; %d ranks SHARE %d device(s) (--allow-shared-devices: functional check, not an N-GPU point)" \
            % (world, plan["distinct_devices"])
    line = line_skeleton(args, work, world, value, elapsed, per_rank, parallelism, plan["distinct_devices"])
    if bcast_ms is not None:
        line["key_broadcast_ms"] = bcast_ms
        line["key_bytes"] = sum(t.numel() * t.element_size() for t in work.replicated())
    line["checked_items"] = {"distinct_inputs": xc["distinct_inputs"], "twin_compared": int(sum(max(t, 0) for t in tw_all)),
                             "twins_equal": all(t >= 0 for t in tw_all), "repeat_equal": all(v == 1.0 for v in rep_all),
                             "key_digest_equal": xc["key_digest_equal"], "cross_rank_equal": xc["cross_rank_equal"],
                             "oracle_compared": 0}
    if not (xc["key_digest_equal"] and xc["cross_rank_equal"]):
        line["checked_items"].update(ranks_disagreeing_on_key=xc["ranks_disagreeing_on_key"],
                                     inputs_with_differing_outputs=xc["inputs_with_differing_outputs"])
    if dec_all is not None:
        line["checked_items"].update(decrypt_compared=int(sum(max(t, 0) for t in dec_all)), decrypt_equal=all(t >= 0 for t in dec_all))

    if args.step_only or not plan["full_line"]:
        chk = line["checked_items"]
        if world > 1 and args.workload == "c4" and not args.no_cpu_baseline and not args.step_only:
            # one item per rank against the CPU oracle (rank r: the item of distinct index r of its slice) -- together with
            # cross_rank_equal every rank's every output is tied to an oracle-checked one of the same input
            # (the ranks share the host: each takes its share of the cores for the checker -- set before the oracle loads)
            os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or world) // world)))
            ok_r = c4_oracle_item(hg, work, rank % B)
            got = dist.gather_floats(1.0 if ok_r else 0.0)
            chk.update(oracle_compared=len(got), oracle_equal=all(v == 1.0 for v in got),
                       note="one item per rank against the CPU oracle; every other output through cross_rank_equal")
        elif world > 1:
            chk["note"] = "multi-GPU run: the oracle comparison is part of the N=1 line"
        if rank == 0:
            emit(line, args)
        rc = 0 if (chk["twins_equal"] and chk["repeat_equal"] and chk["key_digest_equal"] and chk["cross_rank_equal"] and
                   chk.get("oracle_equal", True) and chk.get("decrypt_equal", True)) else 1
        dist.close(rc)
        return rc

    prof = load_profile()
    if args.workload == "c5":
        ms = elapsed / args.steps * 1e3
        line["as_built"] = prof_group(prof, "c5_tfhe_gates", ms * 8192 / B) if B != 8192 else prof_group(prof, "c5_tfhe_gates", ms)
        line["as_built"]["note"] = "counter pass taken at 8192 gates per call; live time scaled to 8192 gates" if B != 8192 else ""
        if not args.no_cpu_baseline:
            from oracle import binding as ob
            n_chk = 4
            ok, cpu_s = work.oracle_check(n_chk)
            line["checked_items"].update(oracle_compared=n_chk, oracle_equal=ok)
            line["cpu_baseline"] = {"value": n_chk / cpu_s, "unit": work.unit, "cores": ob.lib().o_omp_threads(), "kind": "port",
                                    "sample": "%d gates of the same workload, CPU oracle (includes its own key preparation), %.1f s" % (n_chk, cpu_s),
                                    "gpu_matches_cpu_bit_exact": ok}
        emit(line, args)
        dist.close()
        chk = line["checked_items"]
        return 0 if (chk["twins_equal"] and chk["repeat_equal"] and chk["key_digest_equal"] and chk["cross_rank_equal"] and
                     chk.get("oracle_equal", True) and chk.get("decrypt_equal", True)) else 1

    ctx = work.ctx
    Q, Qp, n = ctx.Q_size, ctx.Q_prime_size, N
    l, rc = Q, Qp
    W = 8 * n  # bytes of one limb polynomial
    out, key, ws, ct1, ct2 = work.out, work.key, work.ws, work.ct1, work.ct2
    ct_elems, out_elems = work.ct_elems, work.out_elems
    sample_gpu = None
    if not args.no_cpu_baseline:
        # the outputs the CPU checker will be compared with: the first occurrence of every distinct input
        first_pos = {}
        for b in range(B):
            first_pos.setdefault((first + b) % work.uniq, b)
        sample_gpu = {u: hg.to_host(out.view(B, out_elems)[p, :ct_elems]) for u, p in first_pos.items()}

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
    g256 = lambda polys_: "grid %d" % (16 * polys_ * 256)   # N / 4096 = 16 tiles (or column tiles) per polynomial
    groups = [
        ("ckks_multiply: k_cross_multiplication", timer.ms(
            lambda: ctx.ckks_multiply(ct1, ct_elems, ct2, ct_elems, out, out_elems, 0, B, stream=stream)),
         7 * l * W * B, "read 4l, write 3l limbs", ["k_cross_multiplication grid"]),
        ("INTT of c2: ntt_inv_row (+ ntt_inv_col of the integer limbs)", probe(1), 2 * l * W * B,
         "l limb INTTs, 2W each (the column stages of the FP64 limbs run inside the next group's kernel)",
         ["ntt_inv_row " + g256(l * B), "ntt_inv_col<8, false> " + g256(l * B)]),
        ("decomposing column pass: ntt_fwd_col_multi + ntt_fwd_col<8,true>", probe(2),
         (l + l * rc - l) * W * B, "read l source limbs once, write l*Q' - l half-transformed digits",
         ["ntt_fwd_col_multi<8> " + g256(l * B), "ntt_fwd_col<8, true> " + g256(2 * l * B)]),
        ("row pass + key inner product: ks_row_mac_fp + ks_row_mac", probe(4),
         ((l * rc - l) + l + 2 * rc) * W * B + 2 * l * rc * W,
         "read l*Q' - l digits + l identity limbs, write 2Q' limbs per ciphertext; the key (2 l Q' limbs) once per batch",
         ["ks_row_mac"]),
        ("INTT of the two P limbs", probe(8), 2 * 2 * W * B, "2 limb INTTs",
         ["ntt_inv_row " + g256(2 * B), "ntt_inv_col<8, false> " + g256(2 * B)]),
        ("mod-down NTT with stage one / two fused: ntt_fwd_col_multi + ntt_fwd_row", probe(16),
         (2 + 2 * l + 2 * l + 3 * 2 * l + 2 * l) * W * B,
         "column pass: read 2 P limbs, write 2l; row pass: read 2l + the 2l accumulated limbs + 2l of ct, write 2l",
         ["ntt_fwd_col_multi<8> " + g256(2 * B), "ntt_fwd_col<8, true> " + g256(2 * B), "ntt_fwd_row " + g256(2 * l * B)]),
    ]
    in_step = []
    for name, ms, by, note, kernels in groups:
        g = {"launches": name, "ms": ms, "algorithmic_bytes": by, "accounting": note,
             "achieved_GBps_algorithmic": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        g.update(prof_group(prof, "c4_step", ms, match=kernels, algorithmic_bytes=by))
        in_step.append(g)
    dominant = max(in_step, key=lambda g: g["ms"])

    pair = prof_group(prof, "ntt_pair", ntt_ms[False], algorithmic_bytes=ntt_bytes)
    step_all = prof_group(prof, "c4_step", elapsed / args.steps * 1e3)
    line["roofline"] = {
        "bound": "hbm",
        "kernel": "forward NTT of the key-switch digits (ntt_fwd_col<8,false> + ntt_fwd_row, one launch pair)",
        "achieved": fwd_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fwd_gbps / HBM_PEAK_GBPS,
        "traffic": (pair.get("from_profile") or {}).get("hbm_bytes"),
        "traffic_source": "from_profile (PMC FETCH_SIZE / WRITE_SIZE passes of `bench.py --profile-workload ntt_pair`; not measured in this run)",
        "launch_ms": ntt_ms[False], "limb_ntts_per_launch": polys, "algorithmic_bytes_per_launch": ntt_bytes,
        "launch_pair": pair,
        "step": {k: step_all.get(k) for k in ("from_profile", "achieved_GBps", "frac_of_hbm_peak_as_built", "live_over_profile_ms", "bound")},
        "in_step_dominant": {k: dominant.get(k) for k in ("launches", "ms", "frac", "achieved_GBps_algorithmic", "algorithmic_bytes", "bound",
                                                            "frac_of_binding_ceiling", "live_over_profile_ms")},
    }
    line["ntt"] = {"forward_GBps": fwd_gbps, "inverse_GBps": inv_gbps, "n": N, "limbs": polys}
    line["in_step"] = in_step
    # the reference's own kernel sequence would have to move its bytes (SURVEY 8d, 1028 MiB per op) at this rate to
    # keep up; above the 8 TB/s peak it only says that the fused path moves fewer bytes -- not an efficiency
    line["reference_sequence_bytes_rate_GBps"] = (6 * l * l + 32 * l + 8) * W * value / 1e9

    if not args.no_secondary:
        line["power"] = power_sample(torch, step)
        line["ntt_by_degree"] = ntt_sweep(torch, hg, timer)
    uniq, primes = work.uniq, work.primes
    del work, ct1, ct2, out, ws
    torch.cuda.empty_cache()
    if not args.no_secondary:
        line["secondary"] = secondary_block(torch, hg, timer, prof)
        line["secondary"]["hoisted_rotations"] = hoisted_rotation_block(torch, hg, timer, ctx)

    if not args.no_cpu_baseline:
        from heongpu_amd import synth
        from oracle import binding as ob
        o = ob.OracleContext(ob.CKKS, 16, primes, Q, 1)
        cores = ob.lib().o_omp_threads()
        # every host core busy: as many pairs of the same workload as there are cores (the batch's distinct inputs, tiled
        # exactly as the GPU's batch is), OpenMP over the pairs -- ~16 s on a 128-core host (2 x cores: 33 s, same rate:
        # the host's memory system, not its core count, bounds the restatement)
        sample = args.cpu_sample or max(cores, 2)
        c1u, c2u = c4_host_inputs(primes, l, n, range(uniq))
        key_h = synth.synth_key_np(primes, Q, Qp, n, 3)
        o3 = np.zeros(sample * out_elems, dtype=np.uint64)
        t0 = time.perf_counter()
        threads = ob.lib().o_ckks_mul_relin_batch_tiled(o.h, c1u.ctypes.data, c2u.ctypes.data, uniq, first, o3.ctypes.data,
                                                        key_h.ctypes.data, 0, sample)
        cpu_s = time.perf_counter() - t0
        o3 = o3.reshape(sample, out_elems)[:, :ct_elems]
        equal = [bool(np.array_equal(o3[s], sample_gpu[(first + s) % uniq])) for s in range(sample)]
        line["checked_items"]["oracle_compared"] = min(sample, uniq)   # DISTINCT (input, output) pairs verified by the CPU
        line["checked_items"]["oracle_equal"] = all(equal)
        line["checked_items"]["total"] = B  # every item of the timed batch: against the oracle's result for its input (directly for
        #                                      the first occurrence, through its twin otherwise)
        line["cpu_baseline"] = {
            "value": sample / cpu_s, "unit": "multiply+relinearize/s", "cores": cores, "threads_busy": min(threads, sample),
            "kind": "port",
            "sample": "%d ciphertext pairs of the same workload (CKKS N=2^16 L=16 mul+relin; the batch's %d distinct inputs tiled "
                      "as on the GPU), CPU oracle, OpenMP over the pairs on %d threads, %.1f s" % (sample, uniq, threads, cpu_s),
            "gpu_matches_cpu_bit_exact": all(equal), "gpu_items_compared": sample,
        }
    emit(line, args)
    dist.close()
    chk = line["checked_items"]
    ok = chk["twins_equal"] and chk["repeat_equal"] and chk["key_digest_equal"] and chk["cross_rank_equal"] and chk.get("oracle_equal", True)
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
    if args.compact_from:
        with open(args.compact_from) as f:
            print(compact_line(json.load(f), args.compact_from))
        raise SystemExit(0)
    if args.preflight:
        raise SystemExit(preflight(args))
    if args.profile_workload:
        raise SystemExit(run_profile_workload(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch
        plan = plan_run(args.gpus, {}, torch.cuda.device_count() if torch.cuda.is_available() else 0, args.single_process,
                        args.allow_shared_devices)
        if plan["mode"] == "refuse":   # before any rank starts: fewer devices than ranks is not an N-GPU point
            raise SystemExit("bench.py: " + plan["why"])
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
