#!/bin/bash
# GPersistArgs::nrt: a row group whose second 16-row tile holds padding rows only (Bt <= 16) runs one tile lane
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nl in 3 4; do for m in "" b; do for g in 0 1; do echo "nrt=$g: $(GP_TAGS=1 GP_NRT=$g timeout 60 tools/ubench/gpersist_trace_nt 32 100 $nl $m | head -1 | cut -c1-150)"; done; done; done
timeout 1500 python -m pytest tests/test_gpu_padrows.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do for g in 0 1; do
for cfg in "--net res_lstm_l --batch 8 --gen-updates 2" "--net lstm --batch 8" "--net lstm --batch 16"; do
RSRGAN_GP_NRT=$g timeout 300 python bench.py $cfg --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t14_bench.log 2>&1
echo "nrt=$g $cfg: $(tail -1 gpurun_out/t14_bench.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done; done
