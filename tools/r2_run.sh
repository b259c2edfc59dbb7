#!/bin/bash
# parity tests + headline bench + kernel profile of one build: tools/r2_run.sh <tag> [l2]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/t_parity_$tag.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/t_parity_$tag.log
timeout 300 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-variants > gpurun_out/bench_$tag.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_$tag.log | cut -c1-260
bash tools/prof.sh r2$tag --steps 8 --warmup 3 --no-variants > /dev/null 2>&1
head -22 gpurun_out/prof_r2$tag/r_kernel_stats.csv | cut -c1-150
if [ "$2" = "l2" ]; then (cd tools/ubench && timeout 120 ./l2 > ../../gpurun_out/l2_$tag.txt 2>&1); fi
