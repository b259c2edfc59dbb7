#!/bin/bash
# two-launch trailing form: where does the full suite fail?
cd $GRAFT_REPO_ROOT
export RSRGAN_TRAIL=2
for i in 1 2; do
timeout 1500 python -m pytest tests/test_gpu_capi_errors.py tests/test_gpu_dist.py tests/test_gpu_dnn_gan.py tests/test_gpu_fullsize.py -m gpu -q > gpurun_out/t4_$i.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/t4_$i.log | grep -E "^E  |FAILED|passed|failed" | head -40
done
