#!/usr/bin/env python3
"""The numbers the docs may quote, side by side: what the DRIVER measured at the end of each round (BENCH_rNN.json, a fresh
box of the pool) and what the builder's own leases measured (profiles/*/bench_compact_line.json and older full lines).
VERDICT r5 item 6: a headline in the docs has to be one a driver record carries; leases give the range.
usage: python tools/driver_vs_lease.py"""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def row(tag, d):
    if not isinstance(d, dict) or "value" not in d:
        return None
    rf = d.get("roofline") or {}
    deg = d.get("ntt_forward_frac_by_degree") or {k: v.get("forward_frac") for k, v in (d.get("ntt_by_degree") or {}).items()}
    pw = d.get("power") or {}
    return (tag, d.get("value"), d.get("ms_per_step"), rf.get("frac"), " ".join("%.3f" % deg[k] for k in sorted(deg)) if deg else "-",
            pw.get("package_w"), pw.get("sclk_mhz"))


rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json"))):
    j = json.load(open(f))
    r = row("DRIVER " + os.path.basename(f)[6:9], j.get("parsed"))
    rows.append(r or ("DRIVER " + os.path.basename(f)[6:9], None, None, None, "(no parsed line)", None, None))
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "bench*line*.json")) + glob.glob(os.path.join(ROOT, "profiles", "*", "bench_detail.json"))):
    try:
        txt = open(f).read().strip()
        d = json.loads(txt.splitlines()[-1]) if not txt.startswith("{\n") else json.loads(txt)
    except Exception:  # noqa: BLE001
        continue
    if d.get("n_gpus", 1) != 1 or d.get("ranks", 1) != 1 or "mults" not in str(d.get("metric", "")):
        continue
    r = row("lease  " + os.path.relpath(f, os.path.join(ROOT, "profiles")), d)
    if r:
        rows.append(r)
print("%-58s %9s %8s %7s  %-34s %7s %6s" % ("source", "op/s", "ms/step", "frac", "NTT frac 2^12..2^16", "W", "MHz"))
for t, v, ms, fr, deg, w, mhz in rows:
    f = lambda x, p: ("%" + p) % x if isinstance(x, (int, float)) else "-"
    print("%-58s %9s %8s %7s  %-34s %7s %6s" % (t[:58], f(v, ".1f"), f(ms, ".3f"), f(fr, ".4f"), deg, f(w, ".0f"), f(mhz, ".0f")))
