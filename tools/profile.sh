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
            if "rocclr" in k: continue   # input upload before the timed region
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
    e = res["step_kernels"].setdefault(re.sub(r"\(.*", "", k) + f" grid {g}", {"bytes_per_step": 0.0})
    e["bytes_per_step"] += (2 if c == "FETCH_SIZE" else 1) * s * 1024 / steps
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
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcs_FETCH_SIZE $OUT/pmcs_WRITE_SIZE $OUT/pmc_*.csv.tmp
rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmcs_FETCH_SIZE.csv $OUT/pmcs_WRITE_SIZE.csv
ls -la $OUT
head -14 $OUT/kernel_stats.csv
