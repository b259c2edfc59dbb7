#!/bin/bash
# res_lstm_l: the layers' kernel gradients as ONE batched stream-K launch over the zero-padded input width (dk_tmp) against two M = 257 GEMMs per layer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_placement.py tests/test_gpu_padrows.py -m gpu -x -q -k "res or residual or padrows" 2>&1 | tail -3
for i in 1 2; do for g in 0 1; do
for cfg in "--batch 64" "--batch 32" "--batch 8 --gen-updates 2"; do
RSRGAN_DK_PAD=$g timeout 300 python bench.py --net res_lstm_l $cfg --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t13_bench.log 2>&1
echo "dk_pad=$g $cfg: $(tail -1 gpurun_out/t13_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done; done
