#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --net segan --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/o_segan.log 2>&1; echo "segan: $(tail -1 gpurun_out/o_segan.log | grep -o '"ms_per_step": [0-9.]*')"
timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/o_bench.log 2>&1; echo "headline: $(tail -1 gpurun_out/o_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
RSRGAN_GEMM_BATCH=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/o_bench0.log 2>&1; echo "headline, per-layer dK: $(tail -1 gpurun_out/o_bench0.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_placement.py tests/test_gpu_segan.py -x -q -m gpu -p no:cacheprovider -k "gemm or segan_at or batched" > gpurun_out/o_tests.log 2>&1; tail -3 gpurun_out/o_tests.log
bash tools/r5_e.sh | grep -A12 "k_glstm_bwd" | head -16
