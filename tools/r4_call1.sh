R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4a
tools/prof_all.sh r4a/c5_8192 python $R/tools/tfhe_bench.py --gates 8192 --reps 2 > gpurun_out/r4a/c5_8192.log 2>&1
tools/prof_all.sh r4a/c5_8 python $R/tools/tfhe_bench.py --gates 8 --reps 5 > gpurun_out/r4a/c5_8.log 2>&1
tools/prof_all.sh r4a/step python $R/bench.py --step-only --steps 4 --warmup 1 > gpurun_out/r4a/step.log 2>&1
tail -30 gpurun_out/r4a/c5_8192/summary.txt
