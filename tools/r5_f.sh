#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_placement.py tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu -p no:cacheprovider -k "fusions or parity or dp_sequence or two_ranks or graph" > gpurun_out/f_tests.log 2>&1; tail -6 gpurun_out/f_tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/f_bench.log 2>&1; echo "new: $(tail -1 gpurun_out/f_bench.log | cut -c90-200)"
RSRGAN_FUSED_SEG=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/f_bench0.log 2>&1; echo "FUSED_SEG=0: $(tail -1 gpurun_out/f_bench0.log | cut -c90-200)"
done
