#!/bin/bash
# A/B of two library builds under a timing script (e.g. tools/ntt_deg.py): tools/ntt_ab.sh <lib A> <lib B> <python script + args>
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/heongpu_amd/lib/libhegpu.so /tmp/keep.so
for rep in 1 2; do for v in $A $B; do cp $R/$v $R/heongpu_amd/lib/libhegpu.so; echo "== $(basename $v)"; python "$@" 2>&1 | grep -v amdgpu.ids; done; done
cp /tmp/keep.so $R/heongpu_amd/lib/libhegpu.so
