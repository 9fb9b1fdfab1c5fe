"""bench.py's committed counter passes (profiles/profile.json): the file exists, names a tracked directory, holds every
workload the line quotes, and bench.py's reader turns it into the `from_profile` blocks -- CPU only, no GPU work."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_profile_json_is_complete_and_tracked():
    b = _bench()
    prof = b.load_profile()
    assert prof is not None, "profiles/profile.json is missing"
    d = os.path.join(ROOT, prof["dir"])
    assert os.path.isdir(d), "the directory the profile names (%s) is not in the tree" % prof["dir"]
    assert json.load(open(os.path.join(d, "profile.json"))) == prof, "profiles/profile.json is not the copy of its directory's file"
    assert prof.get("copy_ceiling_GBps", 0) > 4000
    for w in b.PROFILE_WORKLOADS:
        assert w in prof["workloads"], w
        wl = prof["workloads"][w]
        assert wl["batches"] >= 1 and wl["kernels"], w
        assert os.path.exists(os.path.join(d, w, "summary.txt")) and os.path.exists(os.path.join(d, w, "kernel_stats.csv")), w
        for k, e in wl["kernels"].items():
            assert k.startswith("hegpu::") and e["per_batch"] >= 1 and e["ms"] > 0 and e["cycles"] > 0, (w, k)
            # cycles from the counter WINDOW gave dispatches of a few microseconds "clocks" of 2.6 - 6.9 GHz
            # and fractions divided by them.  Since round 5 short dispatches take their cycles from SQ_BUSY_CYCLES
            # (tools/prof_all.sh): every kernel's in-kernel clock has to be one this part can run at.
            ghz = e["cycles"] / (e["ms"] * 1e-3) / 1e9
            assert 1.2 <= ghz <= 2.6, (w, k, ghz, e.get("cycles_source"))
            assert e.get("cycles_source"), (w, k)
            # a clock the heuristic ASSUMED passes the range check by construction -- such kernels must say so
            # (schema v3: cycles_estimated), and the kernels the line's roofline rests on must be measured ones
            if "v3" in prof.get("schema", ""):
                assert e.get("cycles_estimated") == e["cycles_source"].startswith("ESTIMATED"), (w, k)
                if w in ("c4_step", "ntt_pair", "c5_tfhe_gates") and e["ms"] >= 0.1:
                    assert e["cycles_estimated"] is False, (w, k)
            for f in ("valu_busy", "frac_of_issue_ceiling"):
                assert e.get(f) is None or 0.0 <= e[f] <= 1.6, (w, k, f, e[f])  # (simple 32-bit ops issue faster than one per 4 cycles: BEHZ > 1)


def test_from_profile_blocks():
    b = _bench()
    prof = b.load_profile()
    g = b.prof_group(prof, "c5_tfhe_gates", 100.0, match=["k_tfhe_blind_rotate_fp"])
    fp = g["from_profile"]
    assert fp["dir"] == prof["dir"] and len(fp["kernels"]) == 1
    assert 0.5 < fp["frac_of_issue_ceiling"] < 1.0 and fp["frac_of_copy_ceiling"] < 0.1 and g["bound"].startswith("valu")
    assert abs(g["live_over_profile_ms"] - 100.0 / fp["ms"]) < 1e-9
    step = b.prof_group(prof, "c4_step", 8.5)
    assert 25e9 < step["from_profile"]["hbm_bytes"] < 40e9
    pair = b.prof_group(prof, "ntt_pair", 8.0, algorithmic_bytes=17408 * 2 * 8 * 65536)
    assert 1.9 < pair["traffic_over_algorithmic"] < 2.1 and pair["bound"].startswith("hbm")
    one = b.prof_group(prof, "c2_ckks_n14_b1", 0.1)
    assert one["bound"].startswith("neither")
    assert b.prof_group(None, "c4_step", 1.0)["from_profile"] is None
    assert b.prof_group(prof, "c4_step", 1.0, match=["no such kernel"])["from_profile"] is None
