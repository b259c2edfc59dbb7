#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider -k "unprojected or baseline_named" > gpurun_out/v_tests.log 2>&1; tail -6 gpurun_out/v_tests.log
for f in "RSRGAN_GP_NP_BWD=1" "RSRGAN_GP_NP_BWD=0" "RSRGAN_GP_NOPROJ=0"; do
env $f timeout 300 python bench.py --net baseline_named --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/v_bench.log 2>&1; echo "$f: $(tail -1 gpurun_out/v_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
env $f timeout 300 python bench.py --net baseline_named --batch 32 --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/v_bench32.log 2>&1; echo "$f B=32: $(tail -1 gpurun_out/v_bench32.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
bash tools/prof.sh bn3 --net baseline_named --steps 5 --warmup 2 --no-variants --no-kernel-timing >/dev/null 2>&1; head -6 gpurun_out/prof_bn3/r_kernel_stats.csv | cut -c1-130
