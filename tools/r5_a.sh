#!/bin/bash
# round 5, first GPU call: the new tests (padded batches, resident probe / CU masks, DP sequence at the reference's size), a regression
# slice, and the headline + padded-batch bench lines of this build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_padrows.py tests/test_gpu_dist.py -x -q -m gpu -p no:cacheprovider > gpurun_out/a_tests_new.log 2>&1; tail -15 gpurun_out/a_tests_new.log
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_parity.py tests/test_gpu_capi_errors.py -x -q -m gpu -p no:cacheprovider -k "not rced and not segan" > gpurun_out/a_tests_reg.log 2>&1; tail -5 gpurun_out/a_tests_reg.log
timeout 300 python bench.py --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity > gpurun_out/a_bench.log 2>&1; tail -1 gpurun_out/a_bench.log | cut -c1-400
for b in 8 32; do
  timeout 300 python bench.py --batch $b --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/a_bench_b$b.log 2>&1; tail -1 gpurun_out/a_bench_b$b.log | cut -c1-300
  RSRGAN_PAD_ROWS=0 timeout 300 python bench.py --batch $b --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/a_bench_b${b}_nopad.log 2>&1; tail -1 gpurun_out/a_bench_b${b}_nopad.log | cut -c1-300
done
timeout 300 python bench.py --batch 8 --gen-updates 2 --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/a_bench_b8_g2.log 2>&1; tail -1 gpurun_out/a_bench_b8_g2.log | cut -c1-300
