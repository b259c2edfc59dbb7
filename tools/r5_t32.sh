#!/bin/bash
# the input-gradient ring between two layers 8 steps deep instead of 6 (-DGP_XR_STEPS=8): does back-pressure pace the four-layer residual stack?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "--net res_lstm_l --batch 8 --gen-updates 2" "--net res_lstm_l --batch 32" "--net res_lstm_l --batch 64" "--net lstm --batch 64"; do for i in 1 2; do
timeout 300 python bench.py $cfg --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t32_bench.log 2>&1
echo "XR=8 $cfg: $(tail -1 gpurun_out/t32_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
