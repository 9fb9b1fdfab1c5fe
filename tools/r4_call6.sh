mkdir -p gpurun_out/r4f
python bench.py > gpurun_out/r4f/bench_default.json 2> gpurun_out/r4f/bench_default.err; echo "rc=$?"; tail -3 gpurun_out/r4f/bench_default.err | grep -v amdgpu
python -m pytest tests -m gpu -x -q > gpurun_out/r4f/pytest_gpu.txt 2>&1; tail -5 gpurun_out/r4f/pytest_gpu.txt
