#!/bin/bash
# round 6: validation of the in-launch discriminator weight gradients (full-size oracle cases, padded batches, DP sequence) + A/B
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_padrows.py tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_placement.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -8
for i in 1 2; do for g in 0 1; do
RSRGAN_DW_INKERNEL=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/e_$g.log 2>&1
echo "dw_inkernel=$g: $(tail -1 gpurun_out/e_$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
for g in 0 1; do
RSRGAN_DW_INKERNEL=$g timeout 300 python bench.py --net res_lstm_l --batch 8 --gen-updates 2 --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/e8_$g.log 2>&1
echo "recipe dw_inkernel=$g: $(tail -1 gpurun_out/e8_$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
