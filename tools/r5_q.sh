#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_placement.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider -k "batched_kernel or (step_against_oracle and 64)" > gpurun_out/q_tests.log 2>&1; tail -3 gpurun_out/q_tests.log
for i in 1 2 3; do
for f in 1 2 0; do
RSRGAN_GEMM_BATCH=$f timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/q_bench$f.log 2>&1; echo "GEMM_BATCH=$f: $(tail -1 gpurun_out/q_bench$f.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
bash tools/r5_e.sh | grep -A8 "k_gemm_fixup<1" | head -12
