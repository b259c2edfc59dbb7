#!/bin/bash
# res_lstm_l BPTT: the reducers' sum of the layer above's input-gradient partials with ten loads in flight and no leading poll (gp_sum_all<TAG, 10, false>)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_padrows.py -m gpu -x -q -k "residual or shipped or (single_tile and res) or (trailing and res)" 2>&1 | tail -2
for cfg in "--net res_lstm_l --batch 8 --gen-updates 2" "--net res_lstm_l --batch 32" "--net res_lstm_l --batch 64"; do for i in 1 2; do
timeout 300 python bench.py $cfg --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t33_bench.log 2>&1
echo "NB=10+y $cfg: $(tail -1 gpurun_out/t33_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
