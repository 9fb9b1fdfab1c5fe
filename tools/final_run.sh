set -x
O=gpurun_out/r5_runs; rm -rf $O gpurun_out/r5_final; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
bash tools/profile.sh r5_final > $O/profile.log 2>&1; tail -9 $O/profile.log
python bench.py --detail $O/bench_detail.json > $O/bench_compact_line.json 2> $O/bench_stderr.txt; echo rc=$?
python bench.py --workload c5 --detail $O/bench_c5_detail.json > $O/bench_c5_compact_line.json 2>>$O/bench_stderr.txt
python bench.py --gpus 2 > $O/bench_g2_refused.txt 2>&1; echo "refusal rc=$?"
python bench.py --gpus 2 --allow-shared-devices --no-secondary --no-cpu-baseline --detail $O/bench_c4_g2_shared_detail.json > $O/bench_c4_g2_shared_line.json 2>>$O/bench_stderr.txt
python bench.py --gpus 2 --allow-shared-devices --workload c5 --detail $O/bench_c5_g2_shared_detail.json > $O/bench_c5_g2_shared_line.json 2>>$O/bench_stderr.txt
python bench.py --gpus 2 --allow-shared-devices --single-process --workload c5 --detail $O/bench_c5_g2_single_process_detail.json > $O/bench_c5_g2_single_process_line.json 2>>$O/bench_stderr.txt
python bench.py --preflight --gpus 8 2>/dev/null | tail -1 > $O/preflight.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --detail $O/bench_under_launcher_detail.json 2>>$O/bench_stderr.txt | tail -1 > $O/bench_under_launcher_line.json
(for b in ckks bfv tfhe; do echo "######## benchmark_$b.cpp (reference source, compiled unchanged)"; ./heongpu_amd/lib/ref_benchmark_$b 2>&1; done) > $O/reference_benchmarks_unchanged.txt
SOAK_REPS=40 python tools/soak.py > $O/soak.txt 2>&1; tail -1 $O/soak.txt
for g in 8192 1024 8; do python tools/tfhe_bench.py --gates $g 2>&1 | grep -E "NAND|blind|key_sw"; done > $O/tfhe_bench.txt; cat $O/tfhe_bench.txt
tail -c 3600 $O/bench_compact_line.json
