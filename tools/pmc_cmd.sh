#!/bin/bash
# one PMC pass over an arbitrary command: tools/pmc_cmd.sh <tag> "<counters>" <cmd...>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$1; mkdir -p $OUT; shift
SET="$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p -o x -- "$@" > $OUT/run.txt 2>&1
find $OUT/p -name '*counter_collection.csv' -exec cp {} $OUT/pmc.csv \;
rm -rf $OUT/p
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open("$OUT/pmc.csv")):
    k = r["Kernel_Name"].split("(")[0][-40:]
    a = agg[k][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in agg:
    print("==", k)
    for c, (n, s) in agg[k].items():
        print(f"   {c:32s} {s/n:16.1f}  (x{n})")
PY
