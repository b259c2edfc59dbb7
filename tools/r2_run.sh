#!/bin/bash
# one GPU session: parity tests of the hot path, A/B bench (panel vs round-1 kernels), kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/t_parity_$tag.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/t_parity_$tag.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/bench_panel_$tag.log 2>&1; echo "bench panel rc=$?"; tail -2 gpurun_out/bench_panel_$tag.log | cut -c1-600
RSRGAN_PANEL=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/bench_r1k_$tag.log 2>&1; echo "bench r1 kernels rc=$?"; tail -1 gpurun_out/bench_r1k_$tag.log | cut -c1-300
bash tools/prof.sh r2$tag --steps 8 --warmup 2 --no-variants > /dev/null 2>&1
head -16 gpurun_out/prof_r2$tag/r_kernel_stats.csv | cut -c1-200
