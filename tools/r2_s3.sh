#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}
(cd tools/ubench && timeout 120 ./l2 > ../../gpurun_out/l2_$tag.txt 2>&1; sed -n '/== C/,/== D/p' ../../gpurun_out/l2_$tag.txt)
timeout 600 python tools/two_stream.py > gpurun_out/two_stream_$tag.txt 2>&1; tail -6 gpurun_out/two_stream_$tag.txt
timeout 600 python -m pytest tests/test_gpu_trainers.py -q -m gpu -k rced > gpurun_out/t_rced_$tag.log 2>&1; echo "pytest rced rc=$?"; tail -5 gpurun_out/t_rced_$tag.log
