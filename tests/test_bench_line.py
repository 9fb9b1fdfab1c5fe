"""bench.py's stdout contract, CPU only: the LAST stdout line is a compact (< 4 KB) JSON line the driver can hold in the
8 KB tail it keeps (a 21 KB line left BENCH_r04.json with parsed = null), carrying the headline, `roofline`,
`cpu_baseline` and one rate per secondary workload; and the decision which path a command line takes (`plan_run`):
ranks that would share devices are refused unless asked for and then reported as what they are, and `--gpus 1` takes
the same path with and without a launcher."""
import glob
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "checked_items")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _details():
    """the committed full-detail records of N=1 C4 runs in the current schema (round 4's 21 KB line, and later rounds')"""
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[4-9]*_final", "bench_full_line.json")) +
                glob.glob(os.path.join(ROOT, "profiles", "r[4-9]*_final", "bench_detail.json")))
    assert fs, "no committed bench detail under profiles/r*_final/"
    return fs


def driver_parse(stdout):
    """what the driver does with bench.py's stdout: keep the last 8000 characters, take the last line that starts with `{`"""
    tail = stdout[-8000:]
    lines = [ln for ln in tail.splitlines() if ln.startswith("{")]
    assert lines, "no JSON line in the last 8000 characters of stdout"
    return lines[-1], json.loads(lines[-1])


@pytest.mark.parametrize("detail", _details(), ids=lambda p: os.path.relpath(p, ROOT))
def test_compact_line_fits_the_drivers_tail(detail):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--compact-from", detail], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    text, line = driver_parse(r.stdout)
    assert len(text) < 4096, len(text)
    for k in CONTRACT:
        assert k in line, k
    full = json.load(open(detail))
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-4 and abs(line["ms_per_step"] - full["ms_per_step"]) < 1e-3
    assert line["config"]["workload"].startswith("CKKS N=2^16") and "parallelism" in line["config"]
    rf = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["bound"] == "hbm" and rf["unit"] == "GB/s"
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    for name, s in line["secondary"].items():
        rates = [k for k, v in s.items() if k.endswith("_per_s") or k.endswith("_per_s_k8") or k == "ops_per_s_batch64"]
        assert rates, name
        if name != "hoisted_rotations":
            assert s["oracle_equal"] is True and s["twins_equal"] is True and "bound" in s and "frac_of_binding_ceiling" in s, name
    assert "kernels" not in text  # kernel-name lists stay in profiles/profile.json


def test_compact_line_never_exceeds_the_limit_even_with_bloated_detail():
    b = _bench()
    d = json.load(open(_details()[-1]))
    d["config"]["workload"] = "w" * 5000
    d["config"]["parallelism"] = "p" * 5000
    d["cpu_baseline"]["sample"] = "s" * 5000
    d["roofline"]["traffic_source"] = "t" * 5000
    d["secondary"] = {("workload_%d" % i): {"as_built": {"bound": "b" * 300, "frac_of_binding_ceiling": 0.5},
                                             "checked_items": {"oracle_equal": True, "twins_equal": True}} for i in range(40)}
    text = b.compact_line(d, "gpurun_out/bench_detail.json")
    line = json.loads(text)
    assert len(text) <= b.COMPACT_LIMIT
    for k in CONTRACT:
        assert k in line, k


def test_emit_prints_only_the_compact_line_last_and_writes_the_detail(tmp_path, capsys):
    b = _bench()
    d = json.load(open(_details()[-1]))

    class A:
        detail = str(tmp_path / "d" / "bench_detail.json")
    b.emit(d, A)
    out = capsys.readouterr().out
    text, line = driver_parse("x" * 30000 + "\n" + out)
    assert out.count("\n") == 1 and len(text) < 4096 and line["value"] == pytest.approx(d["value"], rel=1e-4)
    assert json.load(open(A.detail)) == d


def test_shared_devices_are_refused_unless_asked_for_and_reported_as_what_they_are():
    b = _bench()
    # --gpus 2 on a 1-GPU box, outside a launcher and on each rank under one
    assert b.plan_run(2, {}, 1)["mode"] == "refuse"
    assert "--allow-shared-devices" in b.plan_run(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, 1)["why"]
    p = b.plan_run(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, 1, allow_shared=True)
    assert p["mode"] == "rank" and p["distinct_devices"] == 1 and p["n_gpus"] == 1 and p["ranks"] == 2 and p["dev_index"] == 0
    assert not p["full_line"]
    # a real 8-GPU node: every rank its own device
    for r in range(8):
        p = b.plan_run(8, {"WORLD_SIZE": "8", "RANK": str(r), "LOCAL_RANK": str(r)}, 8)
        assert p["mode"] == "rank" and p["distinct_devices"] == 8 and p["n_gpus"] == 8 and p["dev_index"] == r
    assert b.plan_run(4, {}, 8)["mode"] == "self_launch" and b.plan_run(4, {}, 8, single_process=True)["mode"] == "single_process"
    assert b.plan_run(2, {"WORLD_SIZE": "4", "RANK": "0"}, 8)["mode"] == "refuse"
    assert b.plan_run(1, {}, 0)["mode"] == "refuse"
    # the line itself
    class W:
        unit, name = "u", "c4"

        def describe(self, args, world):
            return "m", {"workload": "w"}

    class A:
        steps, warmup = 3, 1
    ln = b.line_skeleton(A, W(), 2, 10.0, 1.0, [5.0, 5.0], "shared", 1)
    assert ln["n_gpus"] == 1 and ln["ranks"] == 2 and ln["distinct_devices"] == 1
    c = json.loads(b.compact_line(ln))
    assert c["n_gpus"] == 1 and c["ranks"] == 2 and c["distinct_devices"] == 1


def test_gpus_1_takes_the_same_path_with_and_without_a_launcher():
    """SCALE's N=1 point (python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1) and BENCH's (python
    bench.py --gpus 1) must be the same measurement: same mode, same device, the full line on rank 0, no process group."""
    b = _bench()
    plain = b.plan_run(1, {}, 1)
    under = b.plan_run(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"}, 1)
    for k in ("mode", "world", "rank", "full_line", "distinct_devices", "n_gpus", "ranks", "dev_index"):
        assert plain[k] == under[k], k
    assert plain["mode"] == "rank" and plain["full_line"] and plain["world"] == 1

    class T:  # Dist with one rank never touches torch.distributed
        pass
    d = b.Dist(T, 1, 0, None)
    assert d.max_float(1.5) == 1.5 and d.gather_floats(2.0) == [2.0] and d.broadcast_keys([], None) is None
    d.barrier()
    d.close()


def test_preflight_runs_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight", "--gpus", "8"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["preflight"] and "visible_devices" in rep and "peer_access" in rep and "rccl_version" in rep and "tier" in rep
    if rep["visible_devices"] == 0:
        assert rep["tier"].startswith("none")
