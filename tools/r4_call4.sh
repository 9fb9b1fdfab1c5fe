mkdir -p gpurun_out/r4d
python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r4d/b1.json 2> gpurun_out/r4d/b1.err; echo "b1 rc=$?"; tail -c 600 gpurun_out/r4d/b1.json; tail -3 gpurun_out/r4d/b1.err
python bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/r4d/c5_1.json 2> gpurun_out/r4d/c5_1.err; echo "c5 rc=$?"; tail -c 1500 gpurun_out/r4d/c5_1.json; tail -3 gpurun_out/r4d/c5_1.err
timeout 600 python bench.py --gpus 2 --workload c5 --steps 2 --warmup 1 > gpurun_out/r4d/c5_g2.json 2> gpurun_out/r4d/c5_g2.err; echo "c5 g2 rc=$?"; tail -c 1500 gpurun_out/r4d/c5_g2.json; tail -3 gpurun_out/r4d/c5_g2.err
python bench.py --gpus 2 --single-process --workload c5 --steps 2 --warmup 1 > gpurun_out/r4d/c5_g2sp.json 2> gpurun_out/r4d/c5_g2sp.err; echo "c5 g2sp rc=$?"; tail -c 1500 gpurun_out/r4d/c5_g2sp.json; tail -3 gpurun_out/r4d/c5_g2sp.err
python bench.py --gpus 2 --single-process --steps 2 --warmup 1 > gpurun_out/r4d/c4_g2sp.json 2> gpurun_out/r4d/c4_g2sp.err; echo "c4 g2sp rc=$?"; tail -c 1200 gpurun_out/r4d/c4_g2sp.json; tail -3 gpurun_out/r4d/c4_g2sp.err
python bench.py --profile-workload c2_ckks_n14_b1 --reps 3 2>&1 | tail -2
python bench.py --profile-workload c5_tfhe_gates --reps 1 2>&1 | tail -2
