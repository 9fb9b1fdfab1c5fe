#!/bin/bash
# Every evidence pass for an arbitrary command, one summary per kernel:
#   tools/prof_all.sh <tag> <cmd...>   ->  gpurun_out/<tag>/{kernel_stats.csv, summary.txt, summary.json, run_*.txt}
# Passes (each its own run; --pmc never together with anything but --kernel-trace, as the guide prescribes):
#   1 kernel-trace + stats          -> kernel_stats.csv
#   2 FETCH_SIZE   3 WRITE_SIZE     -> HBM bytes per kernel: (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: wide coalesced
#                                      reads count at 1/2; calibrated in profiles/README.md)
#   4 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES
#   5 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
#   6 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE   (PROF_SKIP_LDS=1 skips 6)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
# shader engines of the device (SQ_BUSY_CYCLES sums one counter per engine): read, not assumed
SES=$(rocminfo 2>/dev/null | awk '/Shader Engines:/ {if ($3 > m) m = $3} END {print m + 0}'); [ "$SES" -gt 0 ] 2>/dev/null || SES=32
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p0 -o x -- "$@" > $OUT/run_stats.txt 2>&1
find $OUT/p0 -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/p0
pass() { # name, counters
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT/p_$name -o x -- "${CMD[@]}" > $OUT/run_$name.txt 2>&1
  find $OUT/p_$name -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$name.csv \;
  find $OUT/p_$name -name '*kernel_trace.csv' -exec cp {} $OUT/trace_$name.csv \;
  rm -rf $OUT/p_$name
}
CMD=("$@")
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
pass sq1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES"
pass sq2 "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"   # (always: SQ_BUSY_CYCLES gives short dispatches their cycles)
if [ -z "${PROF_SKIP_LDS:-}" ]; then
  pass sq3 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
fi
python3 - <<PY
import csv, collections, json, re, os
out = "$OUT"
def norm(k):
    k = re.sub(r"\(.*", "", k); k = re.sub(r"^void ", "", k)
    return k
def load(name):
    """per (kernel, grid): dispatch count, summed counters, summed seconds of the same run"""
    dur = {}
    try:
        for r in csv.DictReader(open(f"{out}/trace_{name}.csv")):
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    except Exception as e:
        print("trace", name, "missing", e)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    try:
        for r in csv.DictReader(open(f"{out}/pmc_{name}.csv")):
            key = norm(r["Kernel_Name"]) + " grid " + r["Grid_Size"]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            d = r["Dispatch_Id"]
            if d not in seen[key]:
                seen[key].add(d); agg[key]["_n"] += 1; agg[key]["_s"] += dur.get(d, 0.0)
    except Exception as e:
        print("pmc", name, "missing", e)
    return agg
res = collections.defaultdict(dict)
for name in ("fetch", "write", "sq1", "sq2", "sq3"):
    if not os.path.exists(f"{out}/pmc_{name}.csv"): continue
    for key, a in load(name).items():
        if "hegpu::" not in key: continue
        n = a["_n"]; e = res[key]
        e.setdefault("dispatches", n)
        for c, v in a.items():
            if c.startswith("_"): continue
            if c == "GRBM_GUI_ACTIVE":
                e.setdefault("cycles_" + name, v / 8 / n); e.setdefault("seconds_" + name, a["_s"] / n)
            else:
                e[c] = v / n
SHORT_MS = 0.1
for key, e in res.items():
    if "FETCH_SIZE" in e:
        e["hbm_bytes"] = (2 * e["FETCH_SIZE"] + e.get("WRITE_SIZE", 0.0)) * 1024
    if "SQ_BUSY_CYCLES" in e and e.get("cycles_sq2"):
        # SQ_BUSY_CYCLES sums the busy cycles of the 32 shader engines' SQs; per engine it is the kernel's own span on that
        # engine (no launch window around it, unlike GRBM_GUI_ACTIVE): the clock it implies over the dispatch's duration
        e["sq_busy_over_gui"] = e["SQ_BUSY_CYCLES"] / e["cycles_sq2"]
        e["GHz_sq_busy"] = e["SQ_BUSY_CYCLES"] / $SES / e["seconds_sq2"] / 1e9
# Cycles of a dispatch.  GRBM_GUI_ACTIVE / 8 is the counter window, which for a dispatch of a few microseconds is several
# times wider than the dispatch ("clocks" of 2.6 - 6.9 GHz).  Dispatches of >= 0.1 ms keep it (window
# error < 3 %); shorter ones take SQ_BUSY_CYCLES / 32 engines scaled to this pass's duration when the launch is big enough to
# keep every engine busy (>= 1024 waves), else duration x the median clock of the workload's other short kernels.
short_clk = sorted(e["GHz_sq_busy"] for e in res.values()
                   if "GHz_sq_busy" in e and e.get("seconds_sq1", 1) * 1e3 < SHORT_MS and e.get("SQ_WAVES", 0) >= 1024)
long_clk = sorted(e["cycles_sq1"] / e["seconds_sq1"] / 1e9 for e in res.values()
                  if e.get("cycles_sq1") and e.get("seconds_sq1", 0) * 1e3 >= SHORT_MS)
ref_clk = (short_clk[len(short_clk) // 2] if short_clk else (long_clk[len(long_clk) // 2] if long_clk else 2.0))
for key, e in res.items():
    if "SQ_INSTS_VALU" in e and e.get("cycles_sq1"):
        e["ms"] = e["seconds_sq1"] * 1e3
        e["GHz_gui_window"] = e["cycles_sq1"] / e["seconds_sq1"] / 1e9
        # cycles_estimated: the cycles are NOT a counter of this dispatch but duration x a clock taken from elsewhere -- every
        # fraction divided by them (valu_busy, frac_of_issue_ceiling, waves_resident) is an estimate and is marked as one
        if e["ms"] >= SHORT_MS:
            e["cycles"], e["cycles_source"], e["cycles_estimated"] = e["cycles_sq1"], "GRBM_GUI_ACTIVE / 8", False
        elif "GHz_sq_busy" in e and e.get("SQ_WAVES", 0) >= 1024:
            e["cycles"], e["cycles_source"] = e["GHz_sq_busy"] * 1e9 * e["seconds_sq1"], "SQ_BUSY_CYCLES / $SES engines (own pass), scaled to this pass's duration"
            e["cycles_estimated"] = False
        else:
            e["cycles"], e["cycles_source"] = ref_clk * 1e9 * e["seconds_sq1"], "ESTIMATED: duration x %.2f GHz (median of the workload's short kernels)" % ref_clk
            e["cycles_estimated"] = True
        e["GHz"] = e["cycles"] / e["seconds_sq1"] / 1e9
        e["valu_busy"] = e["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / e["cycles"]
        e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / max(e["SQ_WAVES"], 1)
        # issue ceiling: one wave64 VALU instruction per SIMD per 4 cycles
        e["frac_of_issue_ceiling"] = e["SQ_INSTS_VALU"] * 4 / 1024 / e["cycles"]
        if "hbm_bytes" in e: e["hbm_GBps"] = e["hbm_bytes"] / e["seconds_sq1"] / 1e9
    if "SQ_WAVE_CYCLES" in e and e.get("cycles_sq2"):
        # SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY count in units of 4 cycles per wave (guide); the ratio is unit-free
        e["wait_any_frac_of_wave_cycles"] = e["SQ_WAIT_INST_ANY"] / max(e["SQ_WAVE_CYCLES"], 1)
        c2 = e["cycles_sq2"] if e["seconds_sq2"] * 1e3 >= SHORT_MS else e.get("GHz", ref_clk) * 1e9 * e["seconds_sq2"]
        e["waves_resident_per_simd"] = e["SQ_WAVE_CYCLES"] * 4 / 1024 / c2
    if "SQ_ACTIVE_INST_LDS" in e and e.get("cycles_sq3"):
        c3 = e["cycles_sq3"] if e["seconds_sq3"] * 1e3 >= SHORT_MS else e.get("GHz", ref_clk) * 1e9 * e["seconds_sq3"]
        e["lds_busy"] = e["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / c3
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
with open(f"{out}/summary.txt", "w") as f:
    f.write("command: $*\n")
    for key, e in sorted(res.items(), key=lambda kv: -kv[1].get("ms", 0) * kv[1].get("dispatches", 1)):
        f.write("%s  x%d\n" % (key, e.get("dispatches", 0)))
        if "cycles_source" in e: f.write("    %-30s %s%s\n" % ("cycles from", e["cycles_source"], "  (fractions below are estimates)" if e.get("cycles_estimated") else ""))
        for c in ("ms", "GHz", "GHz_gui_window", "GHz_sq_busy", "hbm_bytes", "hbm_GBps", "valu_busy", "frac_of_issue_ceiling", "valu_insts_per_wave", "SQ_WAVES",
                  "waves_resident_per_simd", "wait_any_frac_of_wave_cycles", "lds_busy", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            if c in e: f.write("    %-30s %16.4f\n" % (c, e[c]))
print(open(f"{out}/summary.txt").read()[:6000])
PY
rm -f $OUT/pmc_*.csv $OUT/trace_*.csv
