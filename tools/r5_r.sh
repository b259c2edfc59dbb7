#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/ubench/gemm_bench check > gpurun_out/r_gemm_check.log 2>&1; tail -2 gpurun_out/r_gemm_check.log
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_gemm.py tests/test_gpu_segan.py tests/test_gpu_dnn_gan.py -x -q -m gpu -p no:cacheprovider -k "batched_kernel or gemm or segan or dnn" > gpurun_out/r_tests.log 2>&1; tail -3 gpurun_out/r_tests.log
for i in 1 2 3; do
for f in 1 0; do
RSRGAN_GEMM_BATCH=$f timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/r_bench$f.log 2>&1; echo "GEMM_BATCH=$f: $(tail -1 gpurun_out/r_bench$f.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
timeout 300 python bench.py --net segan --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r_segan.log 2>&1; echo "segan: $(tail -1 gpurun_out/r_segan.log | grep -o '"ms_per_step": [0-9.]*')"
bash tools/r5_e.sh | grep -A8 "k_gemm_fixup<1" | head -12
