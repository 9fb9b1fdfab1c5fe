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
import csv, collections, sys
out = "$OUT"
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        rows = list(csv.DictReader(open(f"{out}/pmc_{c}.csv")))
    except Exception as e:
        print(c, "missing", e); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = r.get("Kernel_Name", "?")
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0))
    with open(f"{out}/pmc_{c}_summary.csv", "w") as f:
        f.write("kernel,dispatches,sum,avg_per_dispatch\n")
        for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{k}",{n},{s},{s/n}\n')
PY
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv
ls -la $OUT
head -20 $OUT/kernel_stats.csv
