#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_placement.py -m gpu -x -q -k "persistent_generator or trailing_discriminator_forward or pipelined" 2>&1 | tail -3
for i in 1 2 3; do for g in 0 3; do
RSRGAN_GP_SCHED=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t21_bench.log 2>&1
echo "sched=$g: $(tail -1 gpurun_out/t21_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
