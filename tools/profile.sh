#!/bin/bash
# Every counter pass the bench line quotes (profiles/profile.json), on the GPU box:  tools/profile.sh <tag>
# -> gpurun_out/<tag>/<workload>/{summary.txt, summary.json, kernel_stats.csv}, gpurun_out/<tag>/profile.json
TAG=${1:-r5prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
ARGS=""
for W in c4_step ntt_pair bfv_n14_multiply c3_bfv_n15_rotate c2_ckks_n14_b1 c2_ckks_n14_b64 ckks_n16_method_II c5_tfhe_gates; do
  REPS=4; [ $W = c5_tfhe_gates ] && REPS=2; [ $W = c2_ckks_n14_b1 ] && REPS=20
  SKIP=1; [ $W = c4_step ] && SKIP=""; [ $W = c5_tfhe_gates ] && SKIP=""
  PROF_SKIP_LDS=$SKIP tools/prof_all.sh $TAG/$W python $R/bench.py --profile-workload $W --reps $REPS > $O/$W.log 2>&1
  ARGS="$ARGS $W=$O/$W"
done
[ -x $R/tools/exp/copy_bw ] && $R/tools/exp/copy_bw 16384 5 > $O/copy_bw.txt 2>&1
python $R/tools/build_profile_json.py profiles/$TAG $O/profile.json $ARGS copy_bw=$O/copy_bw.txt box="$(hostname) $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Series' | sed 's/.*: *//')"
