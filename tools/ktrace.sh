#!/bin/bash
# per-dispatch kernel sequence of the LAST repetition of a command: tools/ktrace.sh <tag> <kernels per repetition> <cmd...>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$1; mkdir -p $OUT; shift
PER=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o x -- "$@" > $OUT/run.txt 2>&1
find $OUT/p -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/p
python3 - <<PY
import csv
rows = sorted(csv.DictReader(open("$OUT/kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "hegpu::" in r["Kernel_Name"]][-$PER:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f .. %8.1f us  %7.1f us  grid %-8s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Grid_Size", "?"), r["Kernel_Name"][:70]))
PY
