#!/bin/bash
# A/B of two library builds on one box with tools/c4_ab.py + per-kernel times (rocprofv3 --stats):
#   tools/lib_ab.sh <lib A> <lib B> [kernel-name regex]
A=$1; B=$2; RE=${3:-ks_row_mac|col_multi}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/heongpu_amd/lib/libhegpu.so /tmp/keep.so
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in $A $B; do cp $R/$v $R/heongpu_amd/lib/libhegpu.so; echo "== $(basename $v) $(python $R/tools/c4_ab.py fused_tensor 1 1 20 | tail -1)"; done; done
for v in $A $B; do cp $R/$v $R/heongpu_amd/lib/libhegpu.so; n=$(basename $v .so); rm -rf /tmp/p_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -o x -- python $R/tools/c4_ab.py fused_tensor 1 1 5 > /dev/null 2>&1
  echo "-- $n"; grep -E "$RE" $(find /tmp/p_$n -name "*kernel_stats.csv") | cut -d, -f1-4; done
cp /tmp/keep.so $R/heongpu_amd/lib/libhegpu.so
