#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_placement.py tests/test_gpu_padrows.py -x -q -m gpu -p no:cacheprovider -k "residual or shipped_recipe" > gpurun_out/d_tests.log 2>&1; tail -15 gpurun_out/d_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider -k "res_lstm_l" > gpurun_out/d_tests_full.log 2>&1; tail -5 gpurun_out/d_tests_full.log
for b in 8 32; do
  timeout 300 python bench.py --net res_lstm_l --batch $b --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/d_bench_res_b$b.log 2>&1; echo "res b=$b: $(tail -1 gpurun_out/d_bench_res_b$b.log | cut -c1-200)"
  RSRGAN_GP_RES=0 RSRGAN_PAD_ROWS=0 timeout 300 python bench.py --net res_lstm_l --batch $b --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/d_bench_res_b${b}_off.log 2>&1; echo "res b=$b launch path: $(tail -1 gpurun_out/d_bench_res_b${b}_off.log | cut -c1-200)"
done
timeout 300 python bench.py --net res_lstm_l --batch 8 --gen-updates 2 --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/d_bench_res_b8_g2.log 2>&1; echo "res b=8 1D+2G: $(tail -1 gpurun_out/d_bench_res_b8_g2.log | cut -c1-200)"
RSRGAN_GP_RES=0 RSRGAN_PAD_ROWS=0 timeout 300 python bench.py --net res_lstm_l --batch 8 --gen-updates 2 --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/d_bench_res_b8_g2_off.log 2>&1; echo "res b=8 1D+2G launch path: $(tail -1 gpurun_out/d_bench_res_b8_g2_off.log | cut -c1-200)"
