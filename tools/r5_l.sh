#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/l_bench.log 2>&1; echo "default: $(tail -1 gpurun_out/l_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
RSRGAN_DIN0_SIDE=1 timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/l_bench1.log 2>&1; echo "din0 on side: $(tail -1 gpurun_out/l_bench1.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
RSRGAN_DIN0_SIDE=1 RSRGAN_GEMM_BATCH=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/l_bench2.log 2>&1; echo "din0 on side, per-layer dK: $(tail -1 gpurun_out/l_bench2.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
