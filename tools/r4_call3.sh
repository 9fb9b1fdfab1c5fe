R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4c
for f in 3 5 6 2; do for g in 8192; do
  echo "== br_form $f gates $g"; HEGPU_TFHE_BR_FORM=$f python tools/tfhe_bench.py --gates $g --reps 3 2>&1 | grep -v "prepared\|amdgpu.ids"
done; done > gpurun_out/r4c/forms.txt 2>&1
cat gpurun_out/r4c/forms.txt
python tools/small_sweep.py > gpurun_out/r4c/small_sweep.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4c/small_sweep.txt
