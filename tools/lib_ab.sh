#!/bin/bash
# A/B of two library builds on one box under any timing command; the product's library is restored at the end.
#   tools/lib_ab.sh <lib A> <lib B> [cmd...]        default cmd: python tools/c4_ab.py fused_tensor 1 1 20  (the C4 step)
#   e.g. tools/lib_ab.sh heongpu_amd/lib_a/a.so heongpu_amd/lib_a/b.so python tools/tfhe_bench.py --gates 8192
#        tools/lib_ab.sh a.so b.so python tools/ntt_deg.py
# With the default command it also prints the per-kernel times of both builds (rocprofv3 --kernel-trace --stats).
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
DEFAULT=0; if [ $# -eq 0 ]; then DEFAULT=1; set -- python $R/tools/c4_ab.py fused_tensor 1 1 20; fi
cp $R/heongpu_amd/lib/libhegpu.so /tmp/keep.so
trap 'cp /tmp/keep.so $R/heongpu_amd/lib/libhegpu.so' EXIT
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in $A $B; do cp $R/$v $R/heongpu_amd/lib/libhegpu.so; echo "== $(basename $v): $("$@" 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"; done; done
if [ $DEFAULT = 1 ]; then
  for v in $A $B; do cp $R/$v $R/heongpu_amd/lib/libhegpu.so; n=$(basename $v .so); rm -rf /tmp/p_$n
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -o x -- python $R/tools/c4_ab.py fused_tensor 1 1 5 > /dev/null 2>&1
    echo "-- $n"; python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/p_$n/**/*kernel_stats.csv", recursive=True)[0])):
    if float(r["Percentage"]) >= 2.0: print("   %-64s x%-4s %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
fi
