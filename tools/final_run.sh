#!/bin/bash
# Everything profiles/r6_final/ holds, in one gpurun call:  gpurun --timeout 3000 -- 'bash tools/final_run.sh'
set -x
TAG=${1:-r6_final}
O=gpurun_out/${TAG}_runs; rm -rf $O gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
cp gpurun_out/fp_audit/table.md gpurun_out/fp_audit/audit.json $O/ 2>/dev/null
bash tools/profile.sh $TAG > $O/profile.log 2>&1; tail -9 $O/profile.log
cp gpurun_out/$TAG/profile.json profiles/profile.json   # the lines below quote THIS run's counter passes
python bench.py --detail $O/bench_detail.json > $O/bench_compact_line.json 2> $O/bench_stderr.txt; echo rc=$?
python bench.py --workload c5 --detail $O/bench_c5_detail.json > $O/bench_c5_compact_line.json 2>>$O/bench_stderr.txt
python bench.py --gpus 2 > $O/bench_g2_refused.txt 2>&1; echo "refusal rc=$?"
python bench.py --gpus 2 --allow-shared-devices --no-secondary --detail $O/bench_c4_g2_shared_detail.json > $O/bench_c4_g2_shared_line.json 2>>$O/bench_stderr.txt
python bench.py --gpus 2 --allow-shared-devices --no-secondary --no-cpu-baseline --selftest-corrupt-rank 1 --detail $O/bench_c4_g2_corrupt_detail.json > $O/bench_c4_g2_corrupt_line.json 2>>$O/bench_stderr.txt; echo "corrupted replica rc=$? (must be non-zero)"
python bench.py --gpus 2 --allow-shared-devices --workload c5 --detail $O/bench_c5_g2_shared_detail.json > $O/bench_c5_g2_shared_line.json 2>>$O/bench_stderr.txt
python bench.py --gpus 2 --allow-shared-devices --single-process --no-secondary --detail $O/bench_c4_g2_single_process_detail.json > $O/bench_c4_g2_single_process_line.json 2>>$O/bench_stderr.txt
python bench.py --preflight --gpus 8 2>/dev/null | tail -1 > $O/preflight.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --detail $O/bench_under_launcher_detail.json 2>>$O/bench_stderr.txt | tail -1 > $O/bench_under_launcher_line.json
(for b in ckks bfv tfhe; do echo "######## benchmark_$b.cpp (reference source, compiled unchanged)"; ./heongpu_amd/lib/ref_benchmark_$b 2>&1; done) > $O/reference_benchmarks_unchanged.txt
(for t in heongpu_amd/lib/ref_test_*; do echo "######## $(basename $t)"; $t 2>&1 | tail -4; done) > $O/reference_tests_unchanged.txt
SOAK_REPS=40 python tools/soak.py > $O/soak.txt 2>&1; tail -1 $O/soak.txt
for g in 8192 1024 8; do python tools/tfhe_bench.py --gates $g 2>&1 | grep -E "NAND|blind|key_sw"; done > $O/tfhe_bench.txt; cat $O/tfhe_bench.txt
python tools/exp/overlap_free.py 20 2>&1 | grep -v amdgpu.ids > $O/overlap_free.txt
# the big GPU test files under four sets of forced options (the environment seeds the defaults of every context created)
(for SET in "HEGPU_FP_NTT=0" "HEGPU_FUSED_ROW_MAC=1 HEGPU_COL_MULTI=1 HEGPU_DIGIT_SPLIT=0" "HEGPU_SINGLE_PASS=1 HEGPU_FUSED_ROW_MAC=0" \
            "HEGPU_FUSED_MODDOWN=0 HEGPU_NTT_GALOIS=0 HEGPU_FUSE_INVERSE=0 HEGPU_FUSED_TENSOR=0"; do
   echo "== $SET"; env $SET python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_edges.py tests/test_gpu_keygen.py tests/test_gpu_encode.py \
       -m gpu -q --deselect tests/test_gpu_edges.py::test_options_change_the_launches_not_the_result 2>&1 | tail -2; done) > $O/switch_soak.txt 2>&1; grep -E "==|passed|failed" $O/switch_soak.txt
tail -c 3700 $O/bench_compact_line.json
