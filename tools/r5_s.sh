#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
for w in 256 248 240 232 216; do
RSRGAN_GEMM_BATCH_W=$w timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/s_bench$w.log 2>&1; echo "W=$w: $(tail -1 gpurun_out/s_bench$w.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
