#!/bin/bash
# rocprofv3 kernel statistics of an arbitrary command: tools/kstats.sh <tag> <cmd...>  ->  gpurun_out/<tag>/kernel_stats.csv
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$1; mkdir -p $OUT; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o x -- "$@" > $OUT/run.txt 2>&1
find $OUT/p -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/p
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    print("%-72s %5s %10.1f us %6s %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
