#!/bin/bash
# A/B of library builds on one box: tools/tfhe_ab.sh <lib A> <lib B> [gates...] -- alternates the two libraries under
# tools/tfhe_bench.py (the product's library path is restored at the end)
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/heongpu_amd/lib/libhegpu.so /tmp/keep.so
for g in "${@:-8192}"; do
  for rep in 1 2; do
    for v in $A $B; do
      cp $v $R/heongpu_amd/lib/libhegpu.so
      echo "$(basename $v) gates=$g: $(python $R/tools/tfhe_bench.py --gates $g 2>&1 | grep -E 'blind_rotate' )"
    done
  done
done
cp /tmp/keep.so $R/heongpu_amd/lib/libhegpu.so
