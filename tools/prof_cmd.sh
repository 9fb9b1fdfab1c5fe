#!/bin/bash
# rocprofv3 kernel stats for an arbitrary command: tools/prof_cmd.sh <tag> <cmd...>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$1; mkdir -p $OUT; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o x -- "$@" > $OUT/run.txt 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats
tail -8 $OUT/run.txt
head -16 $OUT/kernel_stats.csv | cut -c1-160
