#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats (+ optional PMC
# passes) for bench.py; summaries land in gpurun_out/<tag>/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
    python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench_stats_run.txt 2>&1
grep '^{' $OUT/bench_stats_run.txt > $OUT/bench_line.json
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
# PMC passes (own runs, kernel-trace only, as the guide prescribes)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- \
      python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/pmc_${C}_run.txt 2>&1
  find $OUT/pmc_$C -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$C.csv \;
done
python3 - <<PY
import csv, collections, json
out = "$OUT"
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        rows = list(csv.DictReader(open(f"{out}/pmc_{c}.csv")))
    except Exception as e:
        print(c, "missing", e); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = (r.get("Kernel_Name", "?"), int(r.get("Grid_Size", 0)))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0))
    with open(f"{out}/pmc_{c}_summary.csv", "w") as f:
        f.write("kernel,grid_size,dispatches,sum_KiB,avg_KiB_per_dispatch\n")
        for (k, g), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{k}",{g},{n},{s},{s/n}\n')
            per[(c, k, g)] = s / n
# HBM traffic per launch of the roofline kernel pair (largest forward-NTT dispatches):
# bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts coalesced reads at 1/2)
def biggest(prefix):
    c = [(g, k) for (cc, k, g) in per if cc == "FETCH_SIZE" and k.startswith(prefix)]
    return max(c) if c else None
res = {"unit": "bytes per launch", "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024", "kernels": {}}
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
res["source"] = "profiles/$TAG (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bench.py --no-cpu-baseline $*)"
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res)[:400])
PY
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv
ls -la $OUT
head -20 $OUT/kernel_stats.csv
