"""The FP64 exactness bounds, measured on the device.

tests/audit/run_audit.py drives the INSTRUMENTED build of the library (tests/audit: the product's sources, every
fp_mul compared with exact 128-bit integer arithmetic, every value checked for integrality and |v| < 2^53) through the
key switch at config C4's exact shape (N = 2^16, {60, 50 x 15 | 60}, 16 digits) and the other degrees, the plain
transforms and the blind rotate, with inputs at their extremes (every residue q - 1, 0 / q - 1 patterns, spikes, q / 2,
in the NTT domain and in the coefficient domain, an all-(q - 1) key).  This test asserts

  * no violation anywhere (the arithmetic was exact at every executed operation),
  * outputs equal to the CPU oracle's,
  * the largest |value| / q recorded per (kernel body, stage) <= the bound CLAIMED in the kernel's comments
    (tests/fp_model.py restates those schedules; tests/test_fp_model.py proves claimed >= rigorous on the CPU),

and writes the table "claimed / rigorous / observed" (copied to profiles/r6_fp_audit/ by tools/final_run.sh)."""
import json
import os
import subprocess
import sys
from fractions import Fraction

import pytest

import fp_model as fm

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q50 = 2 ** 50 - 1  # the largest modulus the plan builder puts on the FP64 path (context.cpp: bit <= 50)
PRODUCT, SUMS, INPUT, OUT = 24, 25, 29, 30


def _bounds(kind, sub, rigorous):
    """{stage: bound on |butterfly output| / q} of the schedule the body runs, plus named extras"""
    n_power = sub + 12
    s1 = n_power - 8
    if kind in ("fwd_col", "fwd_col_decomp", "fwd_single", "fwd_row"):
        rows = dict(fm.sched_forward(Q50, n_power, rigorous).rows)
        out = {s: rows["s%d" % s] for s in range(n_power)}
        out["input"] = rows["in"]
        return out
    if kind in ("ks_row", "ks_row_split"):
        rows = dict(fm.sched_keyswitch(Q50, n_power, 64, rigorous).rows)
        out = {s: rows["s%d" % s] for s in range(s1, n_power)}
        out["product"] = rows["product |t|"]
        out["sums"] = rows["sums"]
        return out
    if kind == "inv":
        rows = dict(fm.sched_inverse(Q50, n_power, rigorous).rows)
        out = {s: rows["s%d" % s] for s in range(1, n_power)}
        out[0] = rows["s0 sum"]
        return out
    if kind in ("tfhe_br", "tfhe_prep"):
        rows = dict(fm.sched_tfhe(2 ** 44 - 1, rigorous).rows)
        out = {s: rows["s%d" % s] for s in range(10)}
        out.update({10 + s: rows["i%d" % s] for s in range(1, 10)})
        out[10] = rows["i0 sum"]
        out["product"] = rows["product |t|"]
        out["sums"] = rows["sums"]
        return out
    raise KeyError(kind)


def _check_rows(workload, tu, rows, report):
    for r in rows:
        kind, sub, stage = r["kind"], r["sub"], r["stage"]
        where = (workload, tu, kind, sub, stage)
        assert kind != "none", ("a value recorded without a site", where, r)
        assert r["abs"] < 1.0, ("a value reached 2^53", where, r)
        claimed, rig = _bounds(kind, sub, False), _bounds(kind, sub, True)
        recomputed = kind in ("ks_row", "ks_row_split") or (kind in ("fwd_row", "fwd_single") and stage >= sub + 8) or \
            (kind == "tfhe_br" and isinstance(stage, int) and (4 <= stage <= 9 or 14 <= stage <= 19))
        slope = 0.375 if recomputed else 0.25
        if stage == PRODUCT:
            assert r["mul_t"] <= float(claimed["product"]), where
            report.append((workload, kind, sub, "product |t|/q", claimed["product"], rig["product"], r["mul_t"]))
            if kind.startswith("ks_row"):
                assert r["mul_y"] < 1.0 and r["mul_w"] <= float(claimed[sub + 11]), where  # key < q, digit un-reduced
                report.append((workload, kind, sub, "product: digit |x|/q", claimed[sub + 11], rig[sub + 11], r["mul_w"]))
        elif SUMS <= stage <= SUMS + 3:
            v = max(r["sum"], r["red_in"])
            assert v <= float(claimed["sums"]), where
            report.append((workload, kind, sub, "sums (%d since re-centring)" % (stage - SUMS + 1), claimed["sums"], rig["sums"], v))
        elif stage == INPUT:
            assert r["sum"] <= float(fm.FP_UNREDUCED_IN), where
            report.append((workload, kind, sub, "input", fm.FP_UNREDUCED_IN, fm.FP_UNREDUCED_IN, r["sum"]))
        elif stage == OUT:
            if kind == "tfhe_br":
                assert r["sum"] * (2 ** 44) <= 2 ** 37, where  # the convolution's own coefficients (tfhe.hip: 2^37)
                report.append((workload, kind, sub, "external product coefficient / p'", Fraction(2 ** 37, 2 ** 44 - 1),
                               Fraction(2 ** 37, 2 ** 44 - 1), r["sum"]))
            else:
                assert r["red_in"] <= float(claimed["sums"]), where
        else:
            # a transform stage: butterfly outputs, the product inside the butterfly, a reduction next to it
            last = (sub + 12) if kind != "tfhe_br" and kind != "tfhe_prep" else 20
            if stage >= last:  # the canonical / final reduction after the last stage
                b = claimed[last - 1 if kind not in ("tfhe_prep",) else 9]
                assert r["red_in"] <= float(b), where
                continue
            b, br = claimed[stage], rig[stage]
            assert r["sum"] <= float(b), (where, r["sum"], float(b))
            # a reduction that precedes (forward) / follows (inverse) the stage sees at most the neighbouring bound
            neighbours = [claimed[s] for s in (stage - 1, stage, stage + 1) if s in claimed] + [claimed.get("input", 0)]
            assert r["red_in"] <= float(max(neighbours)), (where, r["red_in"])
            if r["mul_w"] <= 1.0:
                assert r["mul_t"] <= 0.5 + slope * r["mul_y"] + 1e-9, (where, r)
            if r["sum"] > 0:
                report.append((workload, kind, sub, "stage %d" % stage, b, br, r["sum"]))


@pytest.fixture(scope="module")
def audit(tmp_path_factory):
    lib = os.path.join(ROOT, "tests", "audit", "lib", "libhegpu_audit.so")
    assert os.path.exists(lib), "the instrumented build is missing: make -C tests/audit (done by __graft_entry__.build())"
    out = str(tmp_path_factory.mktemp("audit") / "audit.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "audit", "run_audit.py"), out], cwd=ROOT,
                       capture_output=True, text=True, timeout=3000)
    sys.stdout.write(p.stdout[-4000:])
    sys.stderr.write(p.stderr[-4000:])
    assert os.path.exists(out), "the audit run did not finish"
    with open(out) as f:
        data = json.load(f)
    keep = os.path.join(ROOT, "gpurun_out", "fp_audit")
    os.makedirs(keep, exist_ok=True)
    with open(os.path.join(keep, "audit.json"), "w") as f:
        json.dump(data, f)
    return data, p.returncode


def test_no_violation_and_oracle_equal(audit):
    data, rc = audit
    assert len(data) >= 20
    for workload, res in data.items():
        assert res["equal_to_oracle"], (workload, res.get("detail"))
        for tu, t in res["tables"].items():
            assert sum(t["violations"].values()) == 0, (workload, tu, t["violations"], t["first_violation"])
    assert rc == 0
    # the audit saw the work: the C4-shape key switch alone is > 10^9 checked products
    c4 = data["c4_keyswitch_batch8"]["tables"]["ntt"]
    assert c4["calls"]["fp_mul"] > 10 ** 9 and c4["calls"]["fp_reduce"] > 10 ** 8, c4["calls"]
    assert data["tfhe_blind_rotate"]["tables"]["tfhe"]["calls"]["fp_mul"] > 10 ** 7


def test_observed_maxima_inside_the_claimed_bounds(audit):
    data, _ = audit
    report = []
    for workload, res in data.items():
        for tu, t in res["tables"].items():
            _check_rows(workload, tu, t["rows"], report)
    # every schedule was exercised: FpColSched<4..8> (decomposing and plain), rows, the single pass, both forms of the
    # fused key switch at 2^16, the inverse at every degree, the blind rotate
    seen = {(k, s) for _, k, s, _, _, _, _ in report}
    for sub in range(5):
        assert ("inv", sub) in seen and ("fwd_col_decomp", sub) in seen, (sub, sorted(seen))
        assert ("fwd_col", sub) in seen or ("fwd_single", sub) in seen
    for need in (("ks_row", 4), ("ks_row_split", 4), ("fwd_row", 4), ("fwd_col", 4), ("fwd_single", 0), ("fwd_single", 2),
                 ("tfhe_br", 0), ("tfhe_prep", 0)):
        assert need in seen, (need, sorted(seen))
    # the table: per (body, degree, stage) the largest observation over all workloads
    best = {}
    for workload, kind, sub, what, claimed, rig, obs in report:
        k = (kind, sub, what)
        if k not in best or obs > best[k][2]:
            best[k] = (claimed, rig, obs, workload)
    lines = ["| kernel body | log2 N | where | claimed (comments) | rigorous (q = 2^50 - 1) | largest observed | in workload |",
             "|---|---|---|---|---|---|---|"]
    order = {"input": -1}
    for (kind, sub, what), (claimed, rig, obs, workload) in sorted(
            best.items(), key=lambda kv: (kv[0][0], kv[0][1], int(kv[0][2].split()[1]) if kv[0][2].startswith("stage") else 99, kv[0][2])):
        n = "-" if kind.startswith("tfhe") else str(sub + 12)
        lines.append("| %s | %s | %s | %.4g | %.4g | %.4g | %s |" % (kind, n, what, float(claimed), float(rig), obs, workload))
        assert obs <= float(claimed) and float(rig) <= float(claimed)
    with open(os.path.join(ROOT, "gpurun_out", "fp_audit", "table.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
