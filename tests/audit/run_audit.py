"""TEST INFRASTRUCTURE: drives the INSTRUMENTED build of the library (tests/audit/lib/libhegpu_audit.so, the product's
sources with the hooks of heongpu_amd/csrc/fpmod.cuh filled in by tests/audit/fp_audit.cuh) through the workloads whose
bit-exactness rests on the FP64 arithmetic, with inputs at their extremes, and writes what the device recorded:

  * per workload, translation unit, (site, stage, metric): the largest |value| / q seen;
  * the violation counters (non-integral value, |v| >= 2^53, fp_mul result != y w - k q in 128-bit integers,
    fp_reduce not congruent / not centred, ...) -- every one must be zero;
  * whether the outputs equal the CPU oracle's (the instrumented build computes with the same instructions).

Usage (GPU box):  python tests/audit/run_audit.py out.json [--quick]
The process re-executes itself with HEGPU_AUDIT_LIB set, so that heongpu_amd binds the instrumented library; the
product never reads that variable for anything else (heongpu_amd/_lib.py)."""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
AUDIT_LIB = os.path.join(HERE, "lib", "libhegpu_audit.so")

if os.environ.get("HEGPU_AUDIT_LIB") != AUDIT_LIB:
    if not os.path.exists(AUDIT_LIB):
        sys.exit("build it first: make -C tests/audit")
    env = dict(os.environ, HEGPU_AUDIT_LIB=AUDIT_LIB)
    os.execve(sys.executable, [sys.executable] + sys.argv, env)

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import heongpu_amd as hg  # noqa: E402
from heongpu_amd import _lib  # noqa: E402
from helpers import backend_switches, extreme_limbs, synth_key  # noqa: E402
from oracle import binding as ob  # noqa: E402

KINDS, STAGES, METRICS = 10, 32, 6
SITES = KINDS * 8
KIND_NAMES = ["none", "fwd_col", "fwd_col_decomp", "fwd_row", "fwd_single", "ks_row", "ks_row_split", "inv", "tfhe_prep",
              "tfhe_br"]
METRIC_NAMES = ["mul_y", "mul_w", "mul_t", "sum", "red_in", "abs"]
VIOL_NAMES = ["nonintegral", "range", "mul_inexact", "reduce", "canon", "from_u64", "to_u64", "unused"]


def read_tables(reset=True):
    lib = _lib.load()
    out = {}
    for tu in ("ntt", "tfhe"):
        fn = getattr(lib, "hegpu_fp_audit_read_" + tu)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int]
        mx = np.zeros(SITES * STAGES * METRICS, dtype=np.uint64)
        viol = np.zeros(8, dtype=np.uint64)
        calls = np.zeros(4, dtype=np.uint64)
        first = np.zeros(6, dtype=np.uint64)
        rc = fn(mx.ctypes.data, viol.ctypes.data, calls.ctypes.data, first.ctypes.data, 1 if reset else 0)
        assert rc == mx.size, rc
        vals = mx.view(np.float64).reshape(SITES, STAGES, METRICS)
        rows = []
        for site in range(SITES):
            for stage in range(STAGES):
                if vals[site, stage].any():
                    rows.append(dict(kind=KIND_NAMES[site // 8], sub=site % 8, stage=stage,
                                     **{m: float(vals[site, stage, i]) for i, m in enumerate(METRIC_NAMES)}))
        out[tu] = dict(rows=rows, violations={n: int(v) for n, v in zip(VIOL_NAMES, viol)},
                       calls=dict(fp_mul=int(calls[0]), fp_reduce=int(calls[1]), values=int(calls[2]), conversions=int(calls[3])),
                       first_violation=None if first[0] == 0 else dict(
                           kind=VIOL_NAMES[int(first[0]) - 1], site=int(first[1]), stage=int(first[2]),
                           values=[float(v) for v in first[3:].view(np.float64)]))
    return out


def ckks(n, log_q, log_p, **switches):
    with backend_switches(**switches):
        c = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, log_p, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    o = ob.OracleContext(ob.CKKS, c.n_power, primes, len(log_q), len(log_p))
    c.upload()
    return c, o, primes


def run_keyswitch(label, n, log_q, log_p, depth, patterns, keys, results, **switches):
    """relinearize (inverse transform, decomposition, forward transforms, inner product, mod-down) on extreme inputs"""
    c, o, primes = ckks(n, log_q, log_p, **switches)
    Q, Qp = len(log_q), len(log_q) + len(log_p)
    l = Q - depth
    ok = True
    detail = []
    for key_kind in keys:
        if key_kind == "max":
            key = np.concatenate([np.full(n, primes[j] - 1, dtype=np.uint64) for _ in range(Q) for _c in range(2) for j in range(Qp)])
        else:
            key = synth_key(primes, Q, Qp, n, 3)
        cts = []
        for i, pat in enumerate(patterns):
            parts = [extreme_limbs(c, primes, range(l), n, pat, 31 * i + p) for p in range(3)]
            cts.append(np.concatenate(parts))
        batch = len(cts)
        d = hg.to_device(np.concatenate(cts))
        c.ckks_relinearize_inplace(d, 3 * l * n, hg.to_device(key), depth, batch, c.workspace(hg.OP_CKKS_RELIN, depth, batch))
        torch.cuda.synchronize()
        got = hg.to_host(d).reshape(batch, -1)
        for b in range(batch):
            want = o.ckks_relinearize(cts[b].copy(), key, depth)
            same = bool(np.array_equal(got[b][:2 * l * n], want[:2 * l * n]))
            ok &= same
            detail.append(dict(key=key_kind, pattern=patterns[b], equal_to_oracle=same))
    results[label] = dict(tables=read_tables(), equal_to_oracle=ok, detail=detail, batch=len(patterns))
    print(label, "oracle-equal:", ok, flush=True)


def run_transforms(label, n, bits, results, batches, **switches):
    """plain forward + inverse transforms of FP64 moduli, extremes in, compared with the oracle"""
    c, o, primes = ckks(n, bits, [bits[0]], **switches)
    Q = len(bits)
    ok = True
    for batch in batches:
        for pat in ("max", "alt", "spike", "half", "random"):
            x = np.concatenate([extreme_limbs(c, primes, range(Q), n, pat, 5 + i) for i in range(batch)])
            d = hg.to_device(x)
            c.ntt(d, d, False, batch * Q, Q)
            torch.cuda.synchronize()
            f = hg.to_host(d)
            want = o.ntt(x.copy(), batch * Q, Q)
            ok &= bool(np.array_equal(f, want))
            c.ntt(d, d, True, batch * Q, Q)
            torch.cuda.synchronize()
            ok &= bool(np.array_equal(hg.to_host(d), x))  # round trip: size-independent property
    results[label] = dict(tables=read_tables(), equal_to_oracle=ok)
    print(label, "ok:", ok, flush=True)


def run_c2(label, results):
    """config C2: CKKS N = 2^14 {50, 40 x 7} | {50}: multiply, relinearize, rescale"""
    n, log_q, log_p = 16384, [50] + [40] * 7, [50]
    c, o, primes = ckks(n, log_q, log_p)
    Q, Qp = 8, 9
    key = synth_key(primes, Q, Qp, n, 3)
    ok = True
    for pat in ("random", "max", "alt_coeff"):
        ct1 = np.concatenate([extreme_limbs(c, primes, range(Q), n, pat, 1 + p) for p in range(2)])
        ct2 = np.concatenate([extreme_limbs(c, primes, range(Q), n, pat, 9 + p) for p in range(2)])
        out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
        c.ckks_multiply(hg.to_device(ct1), 2 * Q * n, hg.to_device(ct2), 2 * Q * n, out, 3 * Q * n, 0, 1)
        c.ckks_relinearize_inplace(out, 3 * Q * n, hg.to_device(key), 0, 1, c.workspace(hg.OP_CKKS_RELIN, 0, 1))
        c.ckks_rescale_inplace(out, 3 * Q * n, 0, 1, c.workspace(hg.OP_CKKS_RESCALE, 0, 1))
        torch.cuda.synchronize()
        want = o.ckks_multiply(ct1, ct2, 0)
        o.ckks_relinearize(want, key, 0)
        w2 = want[:2 * Q * n].copy()
        o.ckks_rescale(w2, 0)
        ok &= bool(np.array_equal(hg.to_host(out)[:2 * (Q - 1) * n], w2[:2 * (Q - 1) * n]))
    results[label] = dict(tables=read_tables(), equal_to_oracle=ok)
    print(label, "oracle-equal:", ok, flush=True)


def run_tfhe(label, results, shape):
    """blind rotate with a torus32 boot key whose rows sit at the corners of the lo / hi split"""
    t = hg.TfheContext()
    o = ob.OracleTfhe()
    rng = np.random.default_rng(11)
    polys = t.int("bootkey_elems") // 1024
    coeff = rng.integers(-2**31, 2**31, (polys, 1024), dtype=np.int64).astype(np.int32)
    coeff[0, :4] = [-2**31, 2**31 - 1, 0, -1]
    coeff[1, :] = -2**31
    coeff[2, :] = 2**31 - 1
    coeff[3, ::2] = -2**31
    coeff[3, 1::2] = 2**31 - 1
    for r in range(4, min(polys, 40)):  # many rows at the extremes: the convolution's magnitude bound is what is probed
        coeff[r, :] = np.where(rng.integers(0, 2, 1024) == 1, 2**31 - 1, -2**31)
    bk = np.concatenate([o.to_ntt(coeff[i]) for i in range(polys)])
    prepared = t.prepare_bootkey(hg.to_device(bk))
    assert t.prepared_is_fp64(prepared)
    a = rng.integers(-2**31, 2**31, shape * 512, dtype=np.int64).astype(np.int32)
    b = rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)
    a[0], a[1], a[2], a[3] = 0, -2**31, -1, 2**20
    b[0], b[1] = 0, -2**31
    want_a, want_b = o.bootstrapping(a, b, bk)
    out_a = torch.empty(shape * 1024, dtype=torch.int32, device="cuda")
    out_b = torch.empty(shape, dtype=torch.int32, device="cuda")
    t.bootstrapping(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), prepared, out_a, out_b, shape)
    torch.cuda.synchronize()
    ok = bool(np.array_equal(out_b.cpu().numpy(), want_b) and np.array_equal(out_a.cpu().numpy(), want_a))
    results[label] = dict(tables=read_tables(), equal_to_oracle=ok)
    print(label, "oracle-equal:", ok, flush=True)


def main():
    out_path = sys.argv[1]
    quick = "--quick" in sys.argv
    assert torch.cuda.is_available(), "needs a HIP device"
    assert os.path.basename(_lib.LIB_PATH) == "libhegpu_audit.so", _lib.LIB_PATH
    results = {}
    read_tables()  # clear
    c4 = dict(n=65536, log_q=[60] + [50] * 15, log_p=[60])
    pats = ["max", "alt", "spike", "max_coeff", "alt_coeff", "alt3_coeff", "half_coeff", "random"]
    # config C4's shape, eight ciphertexts: the multi-modulus column pass + ks_row_mac_fp, the kernels of the bench
    run_keyswitch("c4_keyswitch_batch8", depth=0, patterns=pats, keys=["max"] if quick else ["max", "random"], results=results, **c4)
    # two ciphertexts: the launch-size rules take the per-polynomial column pass and ks_row_mac_split
    run_keyswitch("c4_keyswitch_batch2", depth=0, patterns=["max_coeff", "alt_coeff"], keys=["max"], results=results, **c4)
    if not quick:
        run_keyswitch("c4_keyswitch_depth3", depth=3, patterns=["max_coeff", "half_coeff", "random"], keys=["max"], results=results, **c4)
        # the other degrees: FpColSched<4..7>, single pass and two passes
        for n_power in (12, 13, 14, 15):
            n = 1 << n_power
            run_keyswitch("keyswitch_n%d" % n_power, n=n, log_q=[60] + [50] * 5, log_p=[60], depth=0,
                          patterns=["max_coeff", "alt_coeff", "half_coeff", "random"], keys=["max"], results=results)
            run_keyswitch("keyswitch_n%d_col_multi" % n_power, n=n, log_q=[60] + [50] * 5, log_p=[60], depth=0,
                          patterns=["max_coeff", "alt_coeff", "half_coeff", "random"], keys=["max"], results=results,
                          HEGPU_COL_MULTI=1, HEGPU_FUSED_ROW_MAC=1)
        for n_power in (12, 13, 14, 15, 16):
            run_transforms("transforms_n%d" % n_power, 1 << n_power, [50, 50, 49, 36], results, batches=(1, 40))
            if n_power <= 14:
                run_transforms("transforms_n%d_two_pass" % n_power, 1 << n_power, [50, 50, 49, 36], results, batches=(3,),
                               HEGPU_SINGLE_PASS=0)
                run_transforms("transforms_n%d_single_pass" % n_power, 1 << n_power, [50, 50, 49, 36], results, batches=(3,),
                               HEGPU_SINGLE_PASS=1)
        run_c2("c2_mul_relin_rescale", results)
        run_tfhe("tfhe_blind_rotate", results, 12)
    else:
        run_transforms("transforms_n16", 65536, [50, 50, 49, 36], results, batches=(1,))
        run_tfhe("tfhe_blind_rotate", results, 4)
    with open(out_path, "w") as f:
        json.dump(results, f)
    bad = [k for k, v in results.items() if not v["equal_to_oracle"] or
           any(sum(t["violations"].values()) for t in v["tables"].values())]
    print("audit written to", out_path, "| workloads:", len(results), "| failing:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
