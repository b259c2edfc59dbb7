#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/h_bench.log 2>&1; echo "nt stash: $(tail -1 gpurun_out/h_bench.log | cut -c150-230)"
done
timeout 600 python -m pytest tests/test_gpu_placement.py -x -q -m gpu -p no:cacheprovider -k "persistent_generator or residual" > gpurun_out/h_tests.log 2>&1; tail -3 gpurun_out/h_tests.log
