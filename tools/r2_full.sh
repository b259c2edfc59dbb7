#!/bin/bash
# full GPU test suite + default bench line (as the driver runs them) + ablations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}
(cd tools/ubench && timeout 120 ./ub t7 > ../../gpurun_out/ub_t7_$tag.txt 2>&1; cat ../../gpurun_out/ub_t7_$tag.txt | tail -22)
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu_$tag.log 2>&1; echo "pytest -m gpu rc=$?"; tail -8 gpurun_out/t_gpu_$tag.log
timeout 900 python bench.py > gpurun_out/bench_default_$tag.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default_$tag.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
