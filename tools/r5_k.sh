#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_fullsize.py tests/test_gpu_gemm.py -x -q -m gpu -p no:cacheprovider -k "batched_kernel or step_against_oracle or gemm" > gpurun_out/k_tests.log 2>&1; tail -4 gpurun_out/k_tests.log
for i in 1 2 3; do
timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/k_bench.log 2>&1; echo "batched: $(tail -1 gpurun_out/k_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
RSRGAN_GEMM_BATCH=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/k_bench0.log 2>&1; echo "per layer: $(tail -1 gpurun_out/k_bench0.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
