#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider -k "unprojected or baseline_named" > gpurun_out/v_tests.log 2>&1; tail -6 gpurun_out/v_tests.log
for f in 1 0; do
RSRGAN_GP_NOPROJ=$f timeout 300 python bench.py --net baseline_named --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/v_bench$f.log 2>&1; echo "NOPROJ=$f: $(tail -1 gpurun_out/v_bench$f.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
RSRGAN_GP_NOPROJ=$f timeout 300 python bench.py --net baseline_named --batch 32 --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/v_bench32_$f.log 2>&1; echo "NOPROJ=$f B=32: $(tail -1 gpurun_out/v_bench32_$f.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
