#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes for bench.py; summaries land
# in gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   stats pass : python bench.py --no-cpu-baseline --no-secondary   -> kernel_stats.csv, bench_line.json
#   PMC passes : FETCH_SIZE and WRITE_SIZE, each in its own run with --kernel-trace only (guide), twice:
#                (a) the full line above  -> traffic of the roofline launch pair (largest forward-NTT dispatches)
#                (b) --step-only          -> every dispatch belongs to a step: bytes moved per step
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
    python $REPO/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/bench_stats_run.txt 2>&1
grep '^{' $OUT/bench_stats_run.txt > $OUT/bench_line.json
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
STEPS=4; WARM=1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- \
      python $REPO/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/pmc_${C}_run.txt 2>&1
  find $OUT/pmc_$C -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$C.csv \;
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmcs_$C -o bench -- \
      python $REPO/bench.py --step-only --steps $STEPS --warmup $WARM > $OUT/pmcs_${C}_run.txt 2>&1
  find $OUT/pmcs_$C -name '*counter_collection.csv' -exec cp {} $OUT/pmcs_$C.csv \;
done
# vector-ALU issue counters, their own passes (SQ counters never share a pass with the traffic counters): the full
# line (-> the roofline launch pair) and --step-only (-> every kernel of the step)
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/sq -o bench -- \
    python $REPO/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/sq_run.txt 2>&1
find $OUT/sq -name '*counter_collection.csv' -exec cp {} $OUT/sq.csv \;
find $OUT/sq -name '*kernel_trace.csv' -exec cp {} $OUT/sq_trace.csv \;
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/sqs -o bench -- \
    python $REPO/bench.py --step-only --steps $STEPS --warmup $WARM > $OUT/sqs_run.txt 2>&1
find $OUT/sqs -name '*counter_collection.csv' -exec cp {} $OUT/sqs.csv \;
find $OUT/sqs -name '*kernel_trace.csv' -exec cp {} $OUT/sqs_trace.csv \;
[ -x $REPO/tools/exp/copy_bw ] && $REPO/tools/exp/copy_bw 16384 5 > $OUT/copy_bw.txt 2>&1
python3 - <<PY
import csv, collections, json, re
out = "$OUT"
def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    try:
        rows = list(csv.DictReader(open(path)))
    except Exception as e:
        print(path, "missing", e); return agg
    for r in rows:
        k = (r.get("Kernel_Name", "?"), int(r.get("Grid_Size", 0)))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0))
    return agg
per, step = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = load(f"{out}/pmc_{c}.csv")
    with open(f"{out}/pmc_{c}_summary.csv", "w") as f:
        f.write("kernel,grid_size,dispatches,sum_KiB,avg_KiB_per_dispatch\n")
        for (k, g), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{k}",{g},{n},{s},{s/n}\n')
            per[(c, k, g)] = s / n
    agg = load(f"{out}/pmcs_{c}.csv")
    with open(f"{out}/pmc_step_{c}_summary.csv", "w") as f:
        f.write("kernel,grid_size,dispatches,sum_KiB,avg_KiB_per_dispatch\n")
        for (k, g), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if "hegpu::" not in k: continue   # input synthesis / verification (torch kernels, copies) outside the timed region
            f.write(f'"{k}",{g},{n},{s},{s/n}\n')
            step[(c, k, g)] = s
# HBM traffic per launch of the roofline kernel pair (largest forward-NTT dispatches):
# bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts coalesced reads at 1/2)
def biggest(prefix):
    c = [(g, k) for (cc, k, g) in per if cc == "FETCH_SIZE" and k.startswith(prefix)]
    return max(c) if c else None
res = {"unit": "bytes", "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024", "kernels": {}}
total = 0.0
for prefix in ("void hegpu::ntt_fwd_col<8, false>", "hegpu::ntt_fwd_row"):
    b = biggest(prefix)
    if not b: continue
    g, k = b
    by = (2 * per[("FETCH_SIZE", k, g)] + per.get(("WRITE_SIZE", k, g), 0.0)) * 1024
    res["kernels"][k] = {"grid_size": g, "fetch_KiB": per[("FETCH_SIZE", k, g)],
                         "write_KiB": per.get(("WRITE_SIZE", k, g)), "bytes": by}
    total += by
res["bytes_per_launch"] = total
res["limb_ntts_per_launch"] = 17408  # bench.py default workload: 64 pairs x 16 digits x 17 limbs
steps = $STEPS + $WARM
sb = sum((2 if c == "FETCH_SIZE" else 1) * s * 1024 for (c, k, g), s in step.items()) / steps
res["step_bytes"] = sb
res["step_batch"] = 64
res["step_kernels"] = {}
for (c, k, g), s in step.items():
    e = res["step_kernels"].setdefault(re.sub(r"^void ", "", re.sub(r"\(.*", "", k)) + f" grid {g}", {"bytes_per_step": 0.0})
    e["bytes_per_step"] += (2 if c == "FETCH_SIZE" else 1) * s * 1024 / steps
# ---- SQ counters per (kernel, grid): wave-level vector instructions, the cycles in which a SIMD issued one, the
# kernels' own cycles (GRBM_GUI_ACTIVE summed over the 8 XCDs -> / 8) and their durations in the same run
def norm(k):
    k = re.sub(r"\(.*", "", k)
    k = re.sub(r"^void ", "", k)
    return re.sub(r"(ks_row_mac(?:_fp)?)<(?:false|true)>", r"\\1", k)
def load_sq(cpath, tpath):
    dur = {}
    try:
        for r in csv.DictReader(open(tpath)):
            dur[r.get("Dispatch_Id")] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    except Exception as e:
        print(tpath, "missing", e)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    try:
        for r in csv.DictReader(open(cpath)):
            key = norm(r.get("Kernel_Name", "?")) + " grid " + r.get("Grid_Size", "0")
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            d = r.get("Dispatch_Id")
            if d not in seen[key]:
                seen[key].add(d)
                agg[key]["seconds"] += dur.get(d, 0.0)
                agg[key]["dispatches"] += 1
    except Exception as e:
        print(cpath, "missing", e)
    res_ = {}
    for key, a in agg.items():
        if "hegpu::" not in key or not a.get("GRBM_GUI_ACTIVE"): continue
        n = a["dispatches"]
        res_[key] = {"dispatches": n, "valu_wave_insts": a["SQ_INSTS_VALU"] / n,
                     "valu_busy_cycles_per_simd": a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / n,
                     "cycles": a["GRBM_GUI_ACTIVE"] / 8 / n, "seconds": a["seconds"] / n, "waves": a["SQ_WAVES"] / n}
        res_[key]["valu_busy"] = res_[key]["valu_busy_cycles_per_simd"] / res_[key]["cycles"]
    return res_
sqs = load_sq(f"{out}/sqs.csv", f"{out}/sqs_trace.csv")
res["step_kernels_sq"] = sqs
res["sq_note"] = ("per dispatch: valu_wave_insts = SQ_INSTS_VALU, valu_busy = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / cycles, "
                   "cycles = GRBM_GUI_ACTIVE / 8 XCDs, seconds from the kernel trace of the same (counter) run")
sqf = load_sq(f"{out}/sq.csv", f"{out}/sq_trace.csv")
pair = [v for k, v in sqf.items() if (k.startswith("hegpu::ntt_fwd_col<8, false>") or k.startswith("hegpu::ntt_fwd_row"))
        and k.endswith("grid 71303168")]
if len(pair) == 2:
    res["roofline_pair_sq"] = {f: sum(p[f] for p in pair) for f in ("valu_wave_insts", "valu_busy_cycles_per_simd", "cycles", "seconds")}
with open(f"{out}/pmc_sq_summary.txt", "w") as f:
    for name, d in (("step", sqs), ("full line", sqf)):
        f.write("== %s\n" % name)
        for k, v in sorted(d.items(), key=lambda kv: -kv[1]["seconds"] * kv[1]["dispatches"]):
            f.write("%-64s x%-4d %8.3f ms  %12.0f wave-insts  busy %.3f  %.3f GHz\n" % (
                k, v["dispatches"], v["seconds"] * 1e3, v["valu_wave_insts"], v["valu_busy"],
                v["cycles"] / v["seconds"] / 1e9 if v["seconds"] else 0))
try:
    rows = [l.split() for l in open(f"{out}/copy_bw.txt") if l.split()[:1] == ["lin16"] and "nt-" not in l]
    res["copy_ceiling_GBps"] = float(rows[-1][-2])
    res["copy_ceiling_note"] = "tools/exp/copy_bw lin16: contiguous 16 B/lane read+write stream, 8 GiB each way"
except Exception as e:
    print("no copy_bw", e)
res["source"] = "profiles/$TAG (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; roofline pair from bench.py --no-cpu-baseline --no-secondary $*, step bytes from bench.py --step-only)"
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcs_FETCH_SIZE $OUT/pmcs_WRITE_SIZE $OUT/pmc_*.csv.tmp $OUT/sq $OUT/sqs
rm -f $OUT/sq.csv $OUT/sqs.csv $OUT/sq_trace.csv $OUT/sqs_trace.csv
rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmcs_FETCH_SIZE.csv $OUT/pmcs_WRITE_SIZE.csv
ls -la $OUT
head -14 $OUT/kernel_stats.csv
