#!/bin/bash
# PMC passes over the standalone NTT harness (tools/exp/ntt_exp0); prints
# per-kernel averages.  usage: tools/pmc_ntt.sh <tag> [ntt_exp args]
TAG=${1:-pmc}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o x -- ${BIN:-$REPO/tools/exp/ntt_exp0} "$@" > $OUT/run$i.txt 2>&1
  find $OUT/p$i -name '*counter_collection.csv' -exec cp {} $OUT/pmc$i.csv \;
  rm -rf $OUT/p$i
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("$OUT/pmc*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k in agg:
    print("==", k)
    for c, (n, s) in agg[k].items():
        print(f"   {c:32s} {s/n:16.1f}  (x{n})")
PY
