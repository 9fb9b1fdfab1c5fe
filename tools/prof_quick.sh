#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$1; mkdir -p $OUT; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/run.txt 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats
grep '^{' $OUT/run.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ntt'])"
head -14 $OUT/kernel_stats.csv | cut -c1-150
