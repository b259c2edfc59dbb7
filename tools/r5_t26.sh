#!/bin/bash
# DPersistArgs::nrt: the discriminator's halves of the fused launches drop the padding tile as well
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_placement.py tests/test_gpu_padrows.py -m gpu -x -q -k "single_tile or padded or shipped or trailing" 2>&1 | tail -3
for i in 1 2; do for g in 0 1; do
for cfg in "--net res_lstm_l --batch 8 --gen-updates 2" "--net lstm --batch 8"; do
RSRGAN_DP_NRT=$g timeout 300 python bench.py $cfg --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t26_bench.log 2>&1
echo "dp_nrt=$g $cfg: $(tail -1 gpurun_out/t26_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done; done
for i in 1 2 3; do timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t26_bench.log 2>&1
echo "headline: $(tail -1 gpurun_out/t26_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"; done
