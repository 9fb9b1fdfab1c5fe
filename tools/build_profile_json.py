#!/usr/bin/env python3
"""profiles/profile.json from tools/prof_all.sh runs of `bench.py --profile-workload NAME --reps R`.
usage: build_profile_json.py <profiles dir to describe, e.g. profiles/r4_final> <out.json> NAME=<prof_all output dir> ...
                             [copy_bw=<copy_bw.txt>] [box=<free text>]
Per workload: `batches` (from the JSON line the run printed), per kernel (name + grid) the number of dispatches per batch
and the per-dispatch averages of the counter passes: ms / cycles (SQ pass), hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB,
valu_wave_insts, valu_busy, waits, LDS figures.  Kernels that ran fewer times than `batches` are set-up and dropped."""
import json
import os
import sys


def main():
    desc, outp = sys.argv[1], sys.argv[2]
    res = {"dir": desc, "schema": "profile.json v3 (tools/build_profile_json.py): cycles of dispatches < 0.1 ms from SQ_BUSY_CYCLES over the device's shader-engine count, not the counter window; cycles_estimated marks kernels whose cycles are duration x an assumed clock", "workloads": {},
           "how": "tools/prof_all.sh: rocprofv3 --kernel-trace [--stats | --pmc <one counter set>] -- python bench.py "
                  "--profile-workload NAME --reps R; FETCH_SIZE, WRITE_SIZE and three SQ sets each in a pass of its own; "
                  "hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: wide coalesced reads count at 1/2, profiles/README.md)"}
    for a in sys.argv[3:]:
        k, v = a.split("=", 1)
        if k == "copy_bw":
            try:
                rows = [l.split() for l in open(v) if l.split()[:1] == ["lin16"] and "nt-" not in l]
                res["copy_ceiling_GBps"] = float(rows[-1][-2])
                res["copy_ceiling_note"] = "tools/exp/copy_bw lin16: contiguous 16 B/lane read+write stream, 8 GiB each way, same box"
            except Exception as e:  # noqa: BLE001
                print("no copy ceiling:", e)
            continue
        if k == "box":
            res["box"] = v
            continue
        summ = json.load(open(os.path.join(v, "summary.json")))
        meta = None
        for rf in ("run_sq1.txt", "run_stats.txt", "run_fetch.txt"):
            try:
                for line in open(os.path.join(v, rf)):
                    if line.startswith('{"profile_workload"'):
                        meta = json.loads(line)
            except OSError:
                pass
            if meta:
                break
        if not meta:
            print("no batches line for", k, "in", v)
            continue
        batches = meta["batches"]
        w = {"batches": batches, "units_per_batch": meta.get("units_per_batch"), "kernels": {}}
        for name, e in summ.items():
            n = e.get("dispatches", 0)
            if n < batches or n % batches or "ms" not in e:
                continue
            w["kernels"][name] = {
                "per_batch": int(n // batches), "ms": e["ms"], "cycles": e.get("cycles", e.get("cycles_sq1")), "GHz": e.get("GHz"),
                "cycles_source": e.get("cycles_source"), "cycles_estimated": bool(e.get("cycles_estimated", False)),
                "hbm_bytes": e.get("hbm_bytes", 0.0), "valu_wave_insts": e.get("SQ_INSTS_VALU", 0.0),
                "valu_busy": e.get("valu_busy"), "frac_of_issue_ceiling": e.get("frac_of_issue_ceiling"),
                "waves": e.get("SQ_WAVES"), "waves_resident_per_simd": e.get("waves_resident_per_simd"),
                "wait_any_frac_of_wave_cycles": e.get("wait_any_frac_of_wave_cycles"), "lds_busy": e.get("lds_busy"),
                "lds_bank_conflict_cycles": e.get("SQ_LDS_BANK_CONFLICT")}
        w["ms_per_batch"] = sum(x["ms"] * x["per_batch"] for x in w["kernels"].values())
        w["hbm_bytes_per_batch"] = sum(x["hbm_bytes"] * x["per_batch"] for x in w["kernels"].values())
        res["workloads"][k] = w
        print("%-22s %3d kernels, %9.3f ms and %8.2f GB per batch" % (k, len(w["kernels"]), w["ms_per_batch"], w["hbm_bytes_per_batch"] / 1e9))
    json.dump(res, open(outp, "w"), indent=1)


if __name__ == "__main__":
    main()
