R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_tfhe.py -x -q > gpurun_out/r4b/pytest_tfhe.txt 2>&1; tail -3 gpurun_out/r4b/pytest_tfhe.txt
for f in 2 3 4; do for g in 8192 2048 512 8; do
  echo "== br_form $f gates $g"; HEGPU_TFHE_BR_FORM=$f python tools/tfhe_bench.py --gates $g --reps 3 | grep -v prepared
done; done > gpurun_out/r4b/forms.txt 2>&1
cat gpurun_out/r4b/forms.txt
HEGPU_TFHE_BR_FORM=3 tools/prof_all.sh r4b/c5_8192_form3 python $R/tools/tfhe_bench.py --gates 8192 --reps 2 > gpurun_out/r4b/c5_form3.log 2>&1
head -16 gpurun_out/r4b/c5_8192_form3/summary.txt
