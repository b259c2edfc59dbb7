#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_placement.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_padrows.py -x -q -m gpu -p no:cacheprovider -k "not rced and not segan_col" > gpurun_out/u_tests.log 2>&1; tail -4 gpurun_out/u_tests.log
for i in 1 2 3; do
for f in 1 0; do
RSRGAN_PTR_GRAPHS=$f timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/u_bench$f.log 2>&1; echo "PTR_GRAPHS=$f: $(tail -1 gpurun_out/u_bench$f.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
