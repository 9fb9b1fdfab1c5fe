"""CPU, world_size 2, gloo: the N>1 path of bench.py -- batch sharding +
evaluation-key broadcast -- with the oracle standing in for the GPU compute
(the sharding code is identical; only the backend name differs on MI355X)."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import synth_ct, synth_key

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from heongpu_amd import sharding
    from oracle import binding as ob
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = sharding.init_distributed("gloo")
    assert (r, w) == (rank, world)
    import ctypes
    n, Q, Qp, total = 4096, 3, 4, 5
    bits = (ctypes.c_int * 4)(40, 30, 30, 40)
    out = (ctypes.c_uint64 * 4)()
    ob.lib().o_generate_primes(n, bits, 4, out)
    primes = [int(v) for v in out]
    o = ob.OracleContext(ob.CKKS, 12, primes, Q, 1)
    key = torch.zeros(2 * Q * Qp * n, dtype=torch.int64)
    if rank == 0:  # only rank 0 owns the key before the broadcast
        key.copy_(torch.from_numpy(synth_key(primes, Q, Qp, n, 3).view(np.int64)))
    sharding.broadcast_eval_key(key, src=0, chunk_elems=50000)  # several chunks
    key_np = key.numpy().view(np.uint64)
    start, count = sharding.shard_range(total, world, rank)
    res = []
    for b in range(start, start + count):
        ct1 = synth_ct(primes, range(Q), 2, n, 1 + 10 * b)
        ct2 = synth_ct(primes, range(Q), 2, n, 2 + 10 * b)
        m = o.ckks_multiply(ct1, ct2, 0)
        o.ckks_relinearize(m, key_np, 0)
        res.append(m[:2 * Q * n].copy())
    gathered = [None] * world
    dist.all_gather_object(gathered, (start, count, [r_.tobytes() for r_ in res]))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from heongpu_amd import sharding
    for total in (0, 1, 5, 64, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = sharding.shard_range(total, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(total))


@pytest.mark.timeout(300)
def test_two_rank_sharded_mul_relin_matches_single_process(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference for all 5 ciphertext pairs
    import ctypes
    n, Q, Qp, total = 4096, 3, 4, 5
    bits = (ctypes.c_int * 4)(40, 30, 30, 40)
    out = (ctypes.c_uint64 * 4)()
    oracle.lib().o_generate_primes(n, bits, 4, out)
    primes = [int(v) for v in out]
    o = oracle.OracleContext(oracle.CKKS, 12, primes, Q, 1)
    key = synth_key(primes, Q, Qp, n, 3)
    covered = []
    for start, count, blobs in gathered:
        assert len(blobs) == count
        for i, blob in enumerate(blobs):
            b = start + i
            m = o.ckks_multiply(synth_ct(primes, range(Q), 2, n, 1 + 10 * b),
                                synth_ct(primes, range(Q), 2, n, 2 + 10 * b), 0)
            o.ckks_relinearize(m, key, 0)
            assert m[:2 * Q * n].tobytes() == blob, f"ciphertext {b} differs"
            covered.append(b)
    assert sorted(covered) == list(range(total))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself (torch.distributed.run on
    127.0.0.1); --launcher-selftest runs everything bench.py does around the kernels -- rendezvous,
    contiguous sharding of the global batch, key broadcast from rank 0, MAX reduction of the elapsed
    time -- on CPU with gloo and reports n_gpus = 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["key_broadcast_ok"]
    assert line["slices"] == [[0, 64], [64, 64]] and line["global_batch"] == 128


def test_bench_falls_back_to_one_process_when_the_launch_fails():
    """The third multi-GPU path of bench.py: when the launch of the ranks fails (forced here), one process drives the
    N devices itself -- a thread per device, the same contiguous sharding, the key replicated in the fan-out order of
    hegpu_broadcast_key, a barrier on both sides of the timed region and the maximum over the threads -- and the line
    still reports n_gpus = 2 and says which path ran."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launcher-selftest",
                        "--force-launch-failure"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "single-process path" in r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["key_broadcast_ok"] and "single process" in line["parallelism"]
    assert line["slices"] == [[0, 64], [64, 64], [128, 64]] and line["global_batch"] == 192
    assert line["max_elapsed_s"] >= 0.003


def test_bench_refuses_a_world_size_mismatch():
    """under a launcher the world size must equal --gpus (the driver passes both)"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "world size" in (r.stdout + r.stderr)


def _tfhe_material():
    from oracle import binding as ob
    ot = ob.OracleTfhe()
    rng = np.random.default_rng(5)
    polys = 512 * 2 * 2 * 2
    v = rng.integers(-2**31, 2**31, polys, dtype=np.int64)  # constant polynomials: the NTT image of a constant is the constant
    bk = np.repeat(np.where(v < 0, v + ot.prime, v).astype(np.uint64), 1024)
    r32 = lambda k: rng.integers(-2**31, 2**31, k, dtype=np.int64).astype(np.int32)
    ks_a, ks_b = r32(1024 * 8 * 3 * 512), r32(1024 * 8 * 3)
    total = 3
    return ot, bk, ks_a, ks_b, r32(total * 512), r32(total), r32(total * 512), r32(total), total


def _worker_c5(rank, world, port, q):
    """config C5's split: gates sharded along `shape`, the boot key and the key-switch key replicated from rank 0"""
    import torch
    import torch.distributed as dist

    from heongpu_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sharding.init_distributed("gloo")
    ot, bk, ks_a, ks_b, a1, b1, a2, b2, total = _tfhe_material()
    keys = [torch.from_numpy(bk.view(np.int64).copy()), torch.from_numpy(ks_a.copy()), torch.from_numpy(ks_b.copy())]
    if rank != 0:  # only rank 0 owns the keys before the broadcast
        for k in keys:
            k.zero_()
    for k in keys:
        sharding.broadcast_eval_key(k, src=0, chunk_elems=1 << 20)
    start, count = sharding.shard_range(total, world, rank)
    sl = slice(start, start + count)
    out_a, out_b = ot.gate(0, a1[start * 512:(start + count) * 512], b1[sl], a2[start * 512:(start + count) * 512], b2[sl],
                           keys[0].numpy().view(np.uint64), keys[1].numpy(), keys[2].numpy())
    gathered = [None] * world
    dist.all_gather_object(gathered, (start, count, out_a.tobytes(), out_b.tobytes()))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_tfhe_gates_match_single_process(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c5, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ot, bk, ks_a, ks_b, a1, b1, a2, b2, total = _tfhe_material()
    want_a, want_b = ot.gate(0, a1, b1, a2, b2, bk, ks_a, ks_b)
    covered = 0
    for start, count, ba, bb in gathered:
        assert ba == want_a[start * 512:(start + count) * 512].tobytes(), (start, count)
        assert bb == want_b[start:start + count].tobytes(), (start, count)
        covered += count
    assert covered == total and [g[0] for g in gathered] == [0, 2] and [g[1] for g in gathered] == [2, 1]


def test_bench_c5_selftest_shards_gates_and_replicates_three_keys():
    """`bench.py --workload c5 --gpus 2`: 1024 gates per GPU, contiguous slices, the prepared boot key and both parts of
    the key-switch key replicated (gloo ranks, then the single-process fall-back)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for extra, where in (([], "one process per GPU"), (["--force-launch-failure"], "single process")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5",
                            "--launcher-selftest"] + extra, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["workload"] == "c5" and line["key_broadcast_ok"] and line["replicated_tensors"] == 3
        assert line["slices"] == [[0, 1024], [1024, 1024]] and line["global_batch"] == 2048 and where in line["parallelism"]


def _selftest(extra, timeout=600):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-selftest"] + extra,
                       capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_a_corrupted_key_replica_fails_the_multi_gpu_line():
    """an N > 1 line must detect a wrong key replica.  Every rank digests each replicated key tensor
    and its output per distinct input; the digests are gathered over gloo; rank 0 checks that every rank holds the bits
    it made and that equal inputs gave equal outputs on every rank.  Here rank 1's replica has one bit flipped after
    the broadcast (world 2, gloo, CPU stand-ins whose "outputs" depend on the rank's own replica as on the device):
    the line says so, names the rank, the run exits non-zero -- and is NOT retried as a launch failure."""
    r, line = _selftest(["--gpus", "2"])
    assert r.returncode == 0 and line["key_digest_equal"] and line["cross_rank_equal"] and line["distinct_inputs"] == 64
    r, line = _selftest(["--gpus", "2", "--selftest-corrupt-rank", "1"])
    assert r.returncode != 0, r.stdout[-1500:]
    assert line["key_digest_equal"] is False and line["cross_rank_equal"] is False and line["ranks_disagreeing_on_key"] == [1]
    assert "one process per GPU" in line["parallelism"] and "single-process path" not in r.stderr
    # the single-process tier (the launch of the ranks failing) runs the same check
    r, line = _selftest(["--gpus", "3", "--force-launch-failure", "--selftest-corrupt-rank", "2"])
    assert r.returncode != 0 and line["key_digest_equal"] is False and line["ranks_disagreeing_on_key"] == [2]
    # C5: three replicated tensors
    r, line = _selftest(["--gpus", "2", "--workload", "c5", "--selftest-corrupt-rank", "1"])
    assert r.returncode != 0 and line["key_digest_equal"] is False


def test_launcher_selftest_at_world_8():
    """the driver's 8-GPU shape: eight ranks, 512 pairs in contiguous slices of 64, every rank's digests gathered"""
    r, line = _selftest(["--gpus", "8"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert line["n_gpus"] == 8 and line["global_batch"] == 512 and line["slices"] == [[64 * i, 64] for i in range(8)]
    assert line["key_digest_equal"] and line["cross_rank_equal"] and line["distinct_inputs"] == 64


def test_cross_check_is_a_pure_function_of_the_digests():
    import bench
    keys = [[1, 2], [1, 2], [1, 2]]
    outs = [[5, 6, 0], [5, 0, 7], [0, 6, 7]]
    pres = [[1, 1, 0], [1, 0, 1], [0, 1, 1]]
    x = bench.cross_check(keys, outs, pres)
    assert x["key_digest_equal"] and x["cross_rank_equal"] and x["distinct_inputs"] == 3
    x = bench.cross_check([[1, 2], [1, 3], [1, 2]], outs, pres)
    assert not x["key_digest_equal"] and x["ranks_disagreeing_on_key"] == [1] and x["cross_rank_equal"]
    outs[2][1] = 9
    x = bench.cross_check(keys, outs, pres)
    assert x["key_digest_equal"] and not x["cross_rank_equal"] and x["inputs_with_differing_outputs"] == [1]


def test_digest64_sees_every_single_element_change():
    import torch
    import bench
    t = (torch.arange(100000, dtype=torch.int64) * 2654435761) % 1000003
    d0 = bench.digest64(torch, t, chunk=1 << 12)
    assert d0 == bench.digest64(torch, t.clone()) and 0 <= d0 < 2 ** 64
    for i in (0, 4095, 4096, 99999):
        u = t.clone()
        u[i] ^= 1
        assert bench.digest64(torch, u, chunk=1 << 12) != d0
    swapped = t.clone()
    swapped[[3, 5]] = swapped[[5, 3]]
    assert bench.digest64(torch, swapped) != d0     # position-dependent
    assert bench.digest64(torch, t.to(torch.int32)) == bench.digest64(torch, t.to(torch.int32).to(torch.int64))
