#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}
(cd tools/ubench && timeout 120 ./l2 > ../../gpurun_out/l2_$tag.txt 2>&1; cat ../../gpurun_out/l2_$tag.txt)
timeout 300 python tools/dbg_rced.py 2 > gpurun_out/dbg_rced_$tag.txt 2>&1; tail -40 gpurun_out/dbg_rced_$tag.txt
PYTHONFAULTHANDLER=1 timeout -s ABRT 500 python bench.py > gpurun_out/bench_default_$tag.log 2>&1; echo "bench rc=$?"; tail -30 gpurun_out/bench_default_$tag.log | cut -c1-400
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/t_gpu_$tag.log 2>&1; echo "pytest -m gpu rc=$?"; tail -8 gpurun_out/t_gpu_$tag.log
